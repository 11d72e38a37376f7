# round 5, last set (one box): GPU suite + smoke under the final sources, the driver-style bench record, and (L) does the loop still learn with
# the training pass overlapped (tools/train_omok.py --overlap-train, from scratch, evaluated every 2 iterations against iteration 0 and PUCT@400)
python -m pytest tests -m gpu -x -q > gpurun_out/r5zz_pytest.log 2>&1; tail -3 gpurun_out/r5zz_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r5zz_bench.json 2> gpurun_out/r5zz_bench.err; tail -c 600 gpurun_out/r5zz_bench.json; echo
fmt='
import sys, json
tag = sys.argv[1]
mv = sp = tr = 0.0
for l in sys.stdin:
    d = json.loads(l)
    if d.get("kind") == "iter":
        ph = d["self_play_phases_s"]
        mv += d["moves"]; sp += d["self_play_s"]; tr += d["train_s"]
        print("%s iter %2d: %d games, self-play %.2f s (searches %.2f, waited for the pass %.2f) + train call %.2f s, mean game %.1f plies, depth %.2f, terminal leaves %.3f, loss %s | cumulative %.0f move decisions/s of self-play, %.0f /s with training" % (
            tag, d["iter"], d["games"], d["self_play_s"], ph["play"], ph.get("train_wait", 0.0), d["train_s"], d["mean_game_len"], d["mean_select_depth"], d["terminal_share"], d["loss"], mv / sp, mv / (sp + tr)))
    elif d.get("kind") == "elo":
        print("%s elo after iteration %d vs %s: %s" % (tag, d["iter"], d["vs"], d["result"]))
'
python tools/train_omok.py --out gpurun_out/r5zz_L --minutes ${MINUTES:-6} --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 2 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 --overlap-train > gpurun_out/r5zz_L.log 2>&1
python -c "$fmt" L < gpurun_out/r5zz_L/log.jsonl
grep -i "error\|Traceback\|non-finite" gpurun_out/r5zz_L.log | head -5
rm -f gpurun_out/r5zz_L/*.pt
