# round 4, run B: GPU suite (new: trained-net deep-tree parity, 8-rank launch shape, per-net fp16 fallback), then the
# headline network (9x9, 4 blocks, 128 planes, 400 sims) trained by the engine for 45 minutes, then the evidence on the
# result: forward error of every kernel family on real positions, one self_play(4096) with the trained net, bench leg
python -m pytest tests -m gpu -x -q > gpurun_out/r4b_pytest.log 2>&1; tail -4 gpurun_out/r4b_pytest.log
python tools/train_omok.py --out gpurun_out/r4b_train --minutes 45 --board 9 --blocks 4 --planes 128 --sims 400 \
    --games 2048 --steps 800 --batch 512 --eval-every 5 --eval-matches 64 --ckpt-every 10 --max-ckpts 5 > gpurun_out/r4b_train.log 2>&1
grep '"kind": "elo"' gpurun_out/r4b_train/log.jsonl | tail -6 | cut -c1-300
grep '"kind": "iter"' gpurun_out/r4b_train/log.jsonl | tail -3 | cut -c1-600
python tools/check_trained_net.py --ckpt gpurun_out/r4b_train/final.pt --blocks 4 --out gpurun_out/r4b_trained_net.json 2>&1 | grep -v "amdgpu.ids\|WARNING" | tail -12
python bench.py --trained-weights gpurun_out/r4b_train/final.pt --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err
python -c "import json; d=json.load(open('gpurun_out/r4b_bench.json')); print(d['value'], json.dumps(d.get('trained_net'))[:1500])"
