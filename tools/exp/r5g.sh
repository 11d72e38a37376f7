# round 5: the training loop as a user gets it now -- carry-over (default in the tool) + 5120 slots on 4096 rows -- (A) resumed from the committed
# trained checkpoint: the self-play rate under real training; (A0) the same with the round-4 schedule (2048 slots, no carry-over) on the same box;
# (B) from scratch: does it still learn (vs iteration 0 and vs PUCT@400, 64 games each)
fmt='
import sys, json
tag = sys.argv[1]
mv = sp = tr = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        print("%s iter %2d: %d games on %d slots, self-play %.0f move decisions/s (%.1f s) + train %.1f s, mean game %.1f plies, depth %.2f, terminal leaves %.3f, trims %s, fp16 %s, loss %s" % (tag, d["iter"], d["games"], d.get("slots", 0), d["moves_per_s"], d["self_play_s"], d["train_s"], d["mean_game_len"], d["mean_select_depth"], d["terminal_share"], d["trims"]["reroots_trimmed"], d["fp16_range_events"], d["loss"]))
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
        print("%s      cumulative: %.0f move decisions in %.1f s of self-play = %.0f /s (with the %.1f s of training: %.0f /s)" % (tag, mv, sp, mv / sp, tr, mv / (sp + tr)))
    elif d.get("kind") == "elo":
        print("%s elo after iteration %d vs %s: %s" % (tag, d["iter"], d["vs"], d["result"]))
'
python tools/train_omok.py --out gpurun_out/r5g_a --minutes 5 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
    --eval-every 1000 --ckpt-every 1000 > gpurun_out/r5g_a.log 2>&1
python -c "$fmt" A < gpurun_out/r5g_a/log.jsonl
[ -n "$SKIP_A0" ] || python tools/train_omok.py --out gpurun_out/r5g_a0 --minutes 2.5 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
    --eval-every 1000 --ckpt-every 1000 --no-carry-over --oversubscribe 1 --rows static > gpurun_out/r5g_a0.log 2>&1
[ -n "$SKIP_A0" ] || python -c "$fmt" A0 < gpurun_out/r5g_a0/log.jsonl
python tools/train_omok.py --out gpurun_out/r5g_b --minutes 5 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 100 --eval-dense-until 7 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 > gpurun_out/r5g_b.log 2>&1
python -c "$fmt" B < gpurun_out/r5g_b/log.jsonl
grep -i "error\|Traceback\|non-finite" gpurun_out/r5g_a.log gpurun_out/r5g_b.log | head -5
rm -f gpurun_out/r5g_a/*.pt gpurun_out/r5g_a0/*.pt gpurun_out/r5g_b/*.pt
