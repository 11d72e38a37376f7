# round 5: speculative prefetch of the most visited child's record in the PUCT descent (AO_TREE_PREFETCH=<min visits>, 0 = off): parity with it on,
# then the trained-net leg of bench.py (tree kernel time, move decisions/s) for several thresholds on one box, off / on alternating
AO_TREE_PREFETCH=4 python -m pytest tests/test_gpu_tree_parity.py tests/test_gpu_fused_parity.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for pf in 0 4 16 64; do
  AO_TREE_PREFETCH=$pf python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-wide-board > gpurun_out/r5zd_$pf.json 2>/dev/null
  python - $pf <<'P'
import json, sys
d = json.load(open('gpurun_out/r5zd_%s.json' % sys.argv[1])); t = d['trained_net']; s = t['static_rows']
print('prefetch >= %3s visits: headline %.0f (tree %.1f us) | trained net over-subscribed %.0f (tree %.1f us, trunk %.3f ms) | static rows %.0f (tree %.1f us)' % (
    sys.argv[1], d['value'], d['roofline_tree']['avg_launch_ms'] * 1e3, t['value'], t['roofline_tree']['avg_launch_ms'] * 1e3, t['trunk_avg_launch_ms'], s['value'], s['roofline_tree']['avg_launch_ms'] * 1e3))
P
done
done
