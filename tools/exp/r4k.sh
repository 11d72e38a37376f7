# round 4: BASELINE configs[4]'s per-GPU shape with a network that learns: 15x15, 10 blocks, 800 sims, 1024 concurrent games per iteration,
# trained by the engine for 32 minutes; then forward error / self-play evidence on the result
python tools/train_omok.py --out gpurun_out/r4k_train15 --minutes 32 --board 15 --blocks 10 --planes 128 --sims 800 \
    --games 1024 --steps 400 --batch 512 --eval-every 100 --eval-dense-until 8 --eval-matches 32 --yardstick puct:800 --ckpt-every 1000 > gpurun_out/r4k_train15.log 2>&1
grep '"kind": "elo"' gpurun_out/r4k_train15/log.jsonl | cut -c1-230 | tail -14
grep '"kind": "iter"' gpurun_out/r4k_train15/log.jsonl | cut -c1-520 | tail -8
python tools/check_trained_net.py --ckpt gpurun_out/r4k_train15/final.pt --board 15 --blocks 10 --sims 800 --games 1024 --boards 1024 --position-games 128 --position-sims 100 --out gpurun_out/r4k_trained_net15.json 2>&1 | grep -v "amdgpu.ids\|WARNING" | tail -10
rm -f gpurun_out/r4k_train15/ckpt_0.pt gpurun_out/r4k_train15/final.pt
