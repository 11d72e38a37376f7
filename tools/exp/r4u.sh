# round 4: the engine as it stands at the end of the round, under real training: (A) 5 minutes from the committed trained checkpoint (deep trees from
# the first move: depth, trims, fp16-range events, self-play rate), (B) 5 minutes from scratch (does it still learn: vs iteration 0 and vs PUCT@400)
python tools/train_omok.py --out gpurun_out/r4u_a --minutes 5 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 --resume profiles/r4_trained_9x9_4block.pt \
    --eval-every 1000 --ckpt-every 1000 > gpurun_out/r4u_a.log 2>&1
grep '"kind": "iter"' gpurun_out/r4u_a/log.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('A iter %d: %d games %.0f moves/s len %.1f depth %.2f terminal %.3f trims %s fp16 %s loss %s' % (d['iter'], d['games'], d['moves_per_s'], d['mean_game_len'], d['mean_select_depth'], d['terminal_share'], d['trims'], d['fp16_range_events'], d['loss']))"
python tools/train_omok.py --out gpurun_out/r4u_b --minutes 5 --board 9 --blocks 4 --sims 400 --games 2048 --steps 800 --batch 512 \
    --eval-every 100 --eval-dense-until 6 --eval-matches 64 --yardstick puct:400 --ckpt-every 1000 > gpurun_out/r4u_b.log 2>&1
grep '"kind": "iter"' gpurun_out/r4u_b/log.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('B iter %d: %d games %.0f moves/s len %.1f depth %.2f terminal %.3f trims %s fp16 %s loss %s' % (d['iter'], d['games'], d['moves_per_s'], d['mean_game_len'], d['mean_select_depth'], d['terminal_share'], d['trims'], d['fp16_range_events'], d['loss']))"
grep '"kind": "elo"' gpurun_out/r4u_b/log.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('B elo iter %d vs %s: %s' % (d['iter'], d['vs'], d['result']))"
rm -f gpurun_out/r4u_a/*.pt gpurun_out/r4u_b/*.pt
