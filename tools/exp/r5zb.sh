# round 5: overlapped training (main.train_async / tools/train_omok.py --overlap-train): parity test, then the training loop as a user gets it,
# (S) strict self-play -> train alternation per iteration (carry-over on, 5120 slots) against (O) the pass beside the next iteration's games; same box,
# resumed from the committed trained checkpoint
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_replay.py -x -q 2>&1 | tail -5
fmt='
import sys, json
tag = sys.argv[1]
mv = sp = tr = wt = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        ph = d["self_play_phases_s"]
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]; wt += ph.get("train_wait", 0.0)
        print("%s iter %2d: %d games, self-play %.2f s (searches %.2f, waited for the pass %.2f) + train call %.2f s, loss %s | cumulative %.0f move decisions/s of self-play, %.0f /s with training" % (
            tag, d["iter"], d["games"], d["self_play_s"], ph["play"], ph.get("train_wait", 0.0), d["train_s"], d["loss"], mv / sp, mv / (sp + tr)))
'
for tag in S O; do
  extra="--no-overlap-train"; [ $tag = O ] && extra="--overlap-train"
  python tools/train_omok.py --out gpurun_out/r5zb_$tag --minutes ${MINUTES:-4} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
      --eval-every 1000 --ckpt-every 1000 $extra > gpurun_out/r5zb_$tag.log 2>&1
  python -c "$fmt" $tag < gpurun_out/r5zb_$tag/log.jsonl
  grep -i "error\|Traceback\|non-finite" gpurun_out/r5zb_$tag.log | head -5
  rm -f gpurun_out/r5zb_$tag/*.pt
done
