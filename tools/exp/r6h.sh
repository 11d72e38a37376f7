#!/bin/bash
# round 6: the loop (tools/train_omok.py defaults: carry-over, 5120 slots on 4096 rows, device sample emission, overlapped training + play-ahead) resumed
# from the round-4 checkpoint, 6 minutes WITH --fp16-grid-weights (two-product kernels in every search) and 6 minutes without (three products), same box;
# then the fp16-grid run's final network against the checkpoint it started from (64 matches) and its conv weights checked
fmt='
import sys, json
tag = sys.argv[1]
it = [json.loads(l) for l in sys.stdin if "\"kind\": \"iter\"" in l]
mv = sp = tr = 0.0
for d in it:
    ph = d["self_play_phases_s"]
    mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
    print("%s iter %2d: %d games, self-play %.2f s (waited for the pass %.2f) + train call %.2f s, depth %.2f, terminal %.3f, trims %s, loss %s, products %s, skipped %s | cumulative %.0f /s of self-play, %.0f /s with training" % (
        tag, d["iter"], d["games"], d["self_play_s"], ph.get("train_wait", 0.0), d["train_s"], d["mean_select_depth"], d["terminal_share"], d["trims"]["reroots_trimmed"], d["loss"], d.get("mfma_products"), d.get("skipped_steps"), mv / sp, mv / (sp + tr)))
sub = it[2:-1]
m = sum(d["moves"] for d in sub); s = sum(d["self_play_s"] for d in sub); t = sum(d["train_s"] for d in sub)
print("%s steady state (iterations 2 .. %d): self-play %.0f move decisions/s, with training %.0f /s" % (tag, sub[-1]["iter"], m / s, m / (s + t)))
'
for mode in G T; do
  flag=""; [ $mode = G ] && flag="--fp16-grid-weights"
  python tools/train_omok.py --out gpurun_out/r6h_$mode --minutes ${MINUTES:-6} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
      --eval-every 1000 --ckpt-every 1000 $flag > gpurun_out/r6h_$mode.log 2>&1
  python -c "$fmt" $mode < gpurun_out/r6h_$mode/log.jsonl
  grep -i "error\|Traceback\|non-finite" gpurun_out/r6h_$mode.log | head -5
done | tee gpurun_out/r6h_loop_fp16grid_vs_three_products.txt
python - <<'PY' | tee -a gpurun_out/r6h_loop_fp16grid_vs_three_products.txt
import sys, torch, numpy as np
sys.path.insert(0, ".")
from alpha_omok_amd import evaluate
from alpha_omok_amd.pvnet import PVNet
def load(p):
    m = PVNet(4, 5, 128, 9); m.load_state_dict(torch.load(p, map_location="cpu", weights_only=True)); return m.cuda().eval()
g, base = load("gpurun_out/r6h_G/final.pt"), load("profiles/r4_trained_9x9_4block.pt")
sd = g.state_dict()
convs = [k for k, v in sd.items() if v.dim() == 4 and v.shape[2] == 3]
print("fp16-grid run final.pt: %d conv tensors, all fp16 numbers: %s" % (len(convs), all(torch.equal(sd[k].half().float(), sd[k]) for k in convs)))
res, (pe, ee), games = evaluate.evaluate_batched(g, base, 9, 400, n_match=64, seed=4242, device=0)
print("fp16-grid run's final network vs the checkpoint it started from (64 matches, 400 sims, noise off, tau 0):", res, "mean plies %.1f" % np.mean([len(x[1]) for x in games]))
PY
rm -f gpurun_out/r6h_*/*.pt
