# round 4, run A: GPU suite on the round's host-side changes, then the small (2-block, 128-plane) network trained by the
# engine itself for a few minutes -- pipeline shake-out + the checkpoint behind tests/golden/trained_2block.npz
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/train_omok.py --out gpurun_out/r4a_train --minutes 9 --board 9 --blocks 2 --planes 128 --sims 200 \
    --games 1024 --steps 200 --batch 512 --eval-every 4 --eval-matches 64 --ckpt-every 8 --max-ckpts 3 2>&1 | grep -v "amdgpu.ids\|WARNING\|^$" | tail -80
ls -la gpurun_out/r4a_train
