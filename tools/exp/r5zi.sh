# round 5: launch order of k_expand_select's slots in over-subscribed searches (k_order: deepest descents of the previous move first); parity of the
# over-subscribed paths with it on, then the trained-net leg of bench.py with AO_TREE_ORDER=0 / 1 alternating on one box
python -m pytest tests/test_gpu_fused_parity.py -x -q 2>&1 | tail -2
python -m pytest tests/test_gpu_dropin.py -x -q -k "over_subscribed or carry_over or plays_ahead" 2>&1 | tail -2
for rep in 1 2; do
for sw in 0 1; do
  AO_TREE_ORDER=$sw python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-wide-board > gpurun_out/r5zi_$sw.json 2>/dev/null
  python - $sw <<'P'
import json, sys
d = json.load(open('gpurun_out/r5zi_%s.json' % sys.argv[1])); t = d['trained_net']; s = t['static_rows']
print('AO_TREE_ORDER=%s: trained net over-subscribed %.0f move decisions/s (tree %.1f us, trunk %.3f ms, %.1f launches/move, fill %.3f, waits %s) | static rows %.0f (tree %.1f us) | ratio %.3f' % (
    sys.argv[1], t['value'], t['roofline_tree']['avg_launch_ms'] * 1e3, t['trunk_avg_launch_ms'], t['network_launches_per_move'], t['batch_fill'], t['leaves_that_waited_a_launch'], s['value'], s['roofline_tree']['avg_launch_ms'] * 1e3, t['vs_static_rows']))
P
done
done
