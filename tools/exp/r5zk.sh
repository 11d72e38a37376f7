# round 5: the loop (tools/train_omok.py defaults: carry-over, 5120 slots on 4096 rows, device sample emission, overlapped training + play-ahead) under the
# FINAL sources (k_order in), 6 minutes from the round-4 checkpoint
fmt='
import sys, json
it = [json.loads(l) for l in sys.stdin if "\"kind\": \"iter\"" in l]
mv = sp = tr = 0.0
for d in it:
    ph = d["self_play_phases_s"]
    mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
    print("iter %2d: %d games, self-play %.2f s (waited for the pass %.2f) + train call %.2f s, depth %.2f, terminal %.3f, trims %s, loss %s | cumulative %.0f /s of self-play, %.0f /s with training" % (
        d["iter"], d["games"], d["self_play_s"], ph.get("train_wait", 0.0), d["train_s"], d["mean_select_depth"], d["terminal_share"], d["trims"]["reroots_trimmed"], d["loss"], mv / sp, mv / (sp + tr)))
sub = it[2:-1]
m = sum(d["moves"] for d in sub); s = sum(d["self_play_s"] for d in sub); t = sum(d["train_s"] for d in sub)
print("steady state (iterations 2 .. %d): self-play %.0f move decisions/s, with training %.0f /s" % (sub[-1]["iter"], m / s, m / (s + t)))
'
python tools/train_omok.py --out gpurun_out/r5zk --minutes ${MINUTES:-6} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
    --eval-every 1000 --ckpt-every 1000 > gpurun_out/r5zk.log 2>&1
python -c "$fmt" < gpurun_out/r5zk/log.jsonl
grep -i "error\|Traceback\|non-finite" gpurun_out/r5zk.log | head -5
rm -f gpurun_out/r5zk/*.pt
