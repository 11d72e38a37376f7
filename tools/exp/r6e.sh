#!/bin/bash
# round 6: does the loop learn with the conv weights kept on the fp16 grid (tools/train_omok.py --fp16-grid-weights: two-product kernels in every
# search)? From scratch, 9x9 / 4 blocks / 400 sims / 2048 games per iteration, evaluated after every iteration up to 6 against iteration 0 and PUCT@400.
# The same command without the flag runs beside it for the rate (three-product kernels), same box.
fmt='
import sys, json
tag = sys.argv[1]
mv = sp = tr = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        ph = d["self_play_phases_s"]
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
        print("%s iter %2d: %d games, self-play %.2f s (searches %.2f, waited for the pass %.2f) + train call %.2f s, mean game %.1f plies, depth %.2f, terminal leaves %.3f, loss %s, products %s, skipped %s | cumulative %.0f move decisions/s of self-play, %.0f /s with training" % (
            tag, d["iter"], d["games"], d["self_play_s"], ph["play"], ph.get("train_wait", 0.0), d["train_s"], d["mean_game_len"], d["mean_select_depth"], d["terminal_share"], d["loss"], d.get("mfma_products"), d.get("skipped_steps"), mv / sp, mv / (sp + tr)))
    elif d.get("kind") == "elo":
        print("%s elo after iteration %d vs %s: %s" % (tag, d["iter"], d["vs"], d["result"]))
'
python tools/train_omok.py --out gpurun_out/r6e_G --minutes ${MINUTES:-7} --iters ${ITERS:-8} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 2 --eval-dense-until 6 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 --fp16-grid-weights > gpurun_out/r6e_G.log 2>&1
python -c "$fmt" G < gpurun_out/r6e_G/log.jsonl | tee gpurun_out/r6e_fp16grid_learning_check.txt
grep -i "error\|Traceback\|non-finite" gpurun_out/r6e_G.log | head -5
python - <<'PY' | tee -a gpurun_out/r6e_fp16grid_learning_check.txt
import torch
sd = torch.load("gpurun_out/r6e_G/final.pt", map_location="cpu", weights_only=True)
convs = [k for k, v in sd.items() if v.dim() == 4 and v.shape[2] == 3]
ok = all(torch.equal(sd[k].half().float(), sd[k]) for k in convs)
print("final.pt: %d conv tensors, all fp16 numbers: %s; dtype %s" % (len(convs), ok, sd[convs[0]].dtype))
PY
python tools/train_omok.py --out gpurun_out/r6e_T --minutes 3 --iters 4 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 100 --ckpt-every 1000 > gpurun_out/r6e_T.log 2>&1
python -c "$fmt" T < gpurun_out/r6e_T/log.jsonl | tee -a gpurun_out/r6e_fp16grid_learning_check.txt
cp gpurun_out/r6e_G/final.pt gpurun_out/r6e_fp16grid_final.pt
rm -f gpurun_out/r6e_G/*.pt gpurun_out/r6e_T/*.pt
