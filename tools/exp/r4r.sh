# (A/B script of a variant that was measured and NOT kept: AO_TREE_PREFETCH is not in the tree -- profiles/r4r_tree_touch_ab.txt)
# round 4: select_game touching the majority child's node record a level ahead (LDS-direct, nothing reads the copy): parity, then the
# tree kernel with the trained and the random-init network and one game alone, AO_TREE_PREFETCH=0 / 1 back to back, twice
AO_TREE_PREFETCH=1 python -m pytest tests/test_gpu_tree_parity.py tests/test_gpu_dropin.py tests/test_gpu_edges.py tests/test_gpu_soak.py -x -q > gpurun_out/r4r_pytest.log 2>&1; tail -3 gpurun_out/r4r_pytest.log
for rep in 1 2; do for pf in 0 1; do
  AO_TREE_PREFETCH=$pf python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r4r_bench_pf$pf.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4r_bench_pf$pf.json')); t=d['trained_net']
print('AO_TREE_PREFETCH=$pf  random-init: value %.0f tree %.1f us | trained: value %.0f depth %.2f trunk %.4f ms tree %.1f us share %.3f' % (d['value'], d['roofline_tree']['avg_launch_ms']*1e3, t['value'], t['mean_select_depth'], t['trunk_avg_launch_ms'], t['roofline_tree']['avg_launch_ms']*1e3, t['roofline_tree']['time_share']))"
done; done
