# round 5: device-side sample emission (ao_replay_extend_moves): parity tests, then the training loop as a user gets it with the samples'
# states built (Q) on the device / (H) on the host and uploaded, same box, resumed from the committed trained checkpoint
python -m pytest tests/test_gpu_replay.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
fmt='
import sys, json
tag = sys.argv[1]
mv = sp = tr = em = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        ph = d["self_play_phases_s"]
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]; em += ph["emit"]
        print("%s iter %2d: %d games on %d slots, self-play %.0f move decisions/s (%.2f s: searches %.2f + samples %.3f) + train %.1f s | cumulative %.0f /s of self-play, %.0f /s with training; samples %.2f s" % (
            tag, d["iter"], d["games"], d.get("slots", 0), d["moves_per_s"], d["self_play_s"], ph["play"], ph["emit"], d["train_s"], mv / sp, mv / (sp + tr), em))
'
for tag in Q H; do
  extra=""; [ $tag = H ] && extra="--host-states"
  python tools/train_omok.py --out gpurun_out/r5q_$tag --minutes 3 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
      --eval-every 1000 --ckpt-every 1000 $extra > gpurun_out/r5q_$tag.log 2>&1
  python -c "$fmt" $tag < gpurun_out/r5q_$tag/log.jsonl
  grep -i "error\|Traceback\|non-finite" gpurun_out/r5q_$tag.log | head -5
  rm -f gpurun_out/r5q_$tag/*.pt
done
