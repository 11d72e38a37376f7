# round 5: the final loop (carry-over, 5120 slots on 4096 rows, samples emitted on the device, training overlapped + play-ahead) for 18 minutes, resumed from
# the round-4 checkpoint: sustained rate with training time included, arena trims / fp16-range events, and does the player still get better
# (64 games against the checkpoint it started from -- "iter0" here -- every 15 iterations, and against the network of the previous evaluation)
fmt='
import sys, json
tag = sys.argv[1]
mv = sp = tr = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        ph = d["self_play_phases_s"]
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
        print("%s iter %2d: %d games, self-play %.2f s (waited for the pass %.2f) + train call %.2f s, game %.1f plies, depth %.2f, terminal %.3f, trims %s, fp16 %s, loss %s | cumulative %.0f /s of self-play, %.0f /s with training" % (
            tag, d["iter"], d["games"], d["self_play_s"], ph.get("train_wait", 0.0), d["train_s"], d["mean_game_len"], d["mean_select_depth"], d["terminal_share"], d["trims"]["reroots_trimmed"], d["fp16_range_events"], d["loss"], mv / sp, mv / (sp + tr)))
    elif d.get("kind") == "elo":
        print("%s elo after iteration %d vs %s: %s" % (tag, d["iter"], d["vs"], d["result"]))
'
python tools/train_omok.py --out gpurun_out/r5ze --minutes ${MINUTES:-18} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
    --eval-every 15 --eval-matches 64 --ckpt-every 1000 --overlap-train > gpurun_out/r5ze.log 2>&1
python -c "$fmt" F < gpurun_out/r5ze/log.jsonl
grep -i "error\|Traceback\|non-finite" gpurun_out/r5ze.log | head -5
rm -f gpurun_out/r5ze/*.pt
