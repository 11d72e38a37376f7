#!/bin/bash
# round 6: a checkpoint TRAINED on the fp16 grid from scratch (tools/train_omok.py --fp16-grid-weights; 9x9, 4 blocks, 400 sims, 2048 games + 800 mini-batches of
# 512 per iteration, the settings of the round-4 run), ${MINUTES:-40} minutes; evaluated every 5 iterations against iteration 0 and PUCT@400, and at the end against
# the round-4 checkpoint (three products, 45 minutes of training). The final state_dict is kept: profiles/r6j_trained_fp16grid_9x9_4block.pt
python tools/train_omok.py --out gpurun_out/r6j --minutes ${MINUTES:-40} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 5 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 --fp16-grid-weights > gpurun_out/r6j.log 2>&1
grep -i "error\|Traceback\|non-finite" gpurun_out/r6j.log | head -5
python - <<'PY' | tee gpurun_out/r6j_trained_fp16grid_summary.txt
import json, sys, torch, numpy as np
sys.path.insert(0, ".")
it = [json.loads(l) for l in open("gpurun_out/r6j/log.jsonl")]
iters = [d for d in it if d.get("kind") == "iter"]
mv = sum(d["moves"] for d in iters); sp = sum(d["self_play_s"] for d in iters); tr = sum(d["train_s"] for d in iters)
print("%d iterations, %d games, %d move decisions, %d optimiser steps; cumulative %.0f move decisions/s of self-play, %.0f /s with training; products %s, skipped steps %s, trims %s, fp16-range events %s" % (
    len(iters), sum(d["games"] for d in iters), mv, iters[-1]["opt_step"], mv / sp, mv / (sp + tr), sorted(set(d.get("mfma_products") for d in iters)), iters[-1].get("skipped_steps"),
    iters[-1]["trims"]["reroots_trimmed"], iters[-1]["fp16_range_events"]))
sub = iters[2:-1]
m = sum(d["moves"] for d in sub); s = sum(d["self_play_s"] for d in sub); t = sum(d["train_s"] for d in sub)
print("steady state (iterations 2 .. %d): self-play %.0f move decisions/s, with training %.0f /s" % (sub[-1]["iter"], m / s, m / (s + t)))
print("loss first / last:", iters[1]["loss"], iters[-1]["loss"], "| mean game length first / last: %.1f / %.1f" % (iters[0]["mean_game_len"], iters[-1]["mean_game_len"]),
      "| depth last %.2f, terminal share %.3f" % (iters[-1]["mean_select_depth"], iters[-1]["terminal_share"]))
for d in it:
    if d.get("kind") == "elo":
        print("elo after iteration %d vs %s: %s" % (d["iter"], d["vs"], d["result"]))
from alpha_omok_amd import evaluate
from alpha_omok_amd.pvnet import PVNet
def load(p):
    m = PVNet(4, 5, 128, 9); m.load_state_dict(torch.load(p, map_location="cpu", weights_only=True)); return m.cuda().eval()
g, base = load("gpurun_out/r6j/final.pt"), load("profiles/r4_trained_9x9_4block.pt")
sd = g.state_dict()
convs = [k for k, v in sd.items() if v.dim() == 4 and v.shape[2] == 3]
print("final.pt: %d conv tensors, all fp16 numbers: %s, dtypes %s" % (len(convs), all(torch.equal(sd[k].half().float(), sd[k]) for k in convs), sorted(set(str(v.dtype) for v in sd.values()))))
res, (pe, ee), games = evaluate.evaluate_batched(g, base, 9, 400, n_match=64, seed=777, device=0)
print("fp16-grid checkpoint vs the round-4 checkpoint (64 matches, 400 sims, noise off, tau 0):", res, "mean plies %.1f" % np.mean([len(x[1]) for x in games]))
PY
cp gpurun_out/r6j/final.pt gpurun_out/r6j_trained_fp16grid_9x9_4block.pt
cp gpurun_out/r6j/log.jsonl gpurun_out/r6j_train_log.jsonl
rm -f gpurun_out/r6j/*.pt
