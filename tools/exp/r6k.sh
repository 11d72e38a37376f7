# round 6 (the same run as tools/exp/r5p.sh WITH --fp16-grid-weights: k_boardh_w16, two products): BASELINE configs[4]'s per-GPU shape under real training with this round's engine: 15x15, 10 blocks, 800 sims, 1024 rows per simulation
# (k_boardh), 1280 game slots (over-subscribed), carry-over, 1024 games per iteration -- 18 minutes from scratch; round 4 (profiles/r4k_*): 313 - 356
# move decisions/s per iteration on 1024 slots with the per-layer kernels
fmt='
import sys, json
mv = sp = tr = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
        r = d.get("rows", {})
        fill = r.get("rows_live", 0) / max(r.get("rows_launched", 1), 1)
        print("iter %2d: %d games on %d slots, self-play call %.1f s + train %.1f s, mean game %.1f plies, depth %.2f, terminal leaves %.3f, trims %s, fp16 %s, loss %s, batch fill so far %.3f | cumulative %.0f move decisions / %.1f s of self-play = %.1f /s (training included: %.1f /s)" % (
            d["iter"], d["games"], d.get("slots", 0), d["self_play_s"], d["train_s"], d["mean_game_len"], d["mean_select_depth"], d["terminal_share"], d["trims"]["reroots_trimmed"], d["fp16_range_events"], d["loss"], fill, mv, sp, mv / sp, mv / (sp + tr)))
    elif d.get("kind") == "elo":
        print("elo after iteration %d vs %s: %s" % (d["iter"], d["vs"], d["result"]))
'
python tools/train_omok.py --out gpurun_out/r6k_train15 --minutes 18 --board 15 --blocks 10 --planes 128 --sims 800 --games 1024 --steps 400 --batch 512 \
    --rows-per-sim 1024 --oversubscribe 1.25 --eval-every 4 --eval-matches 32 --yardstick puct:800 --ckpt-every 1000 --fp16-grid-weights > gpurun_out/r6k_train15.log 2>&1
python -c "$fmt" < gpurun_out/r6k_train15/log.jsonl
grep -i "error\|Traceback\|non-finite" gpurun_out/r6k_train15.log | head -5
rm -f gpurun_out/r6k_train15/*.pt
