# round 4: the tree kernels' memory round trips made explicit (unconditional clustered loads + pins, path in registers, MT word window, DPP wave
# reductions): parity suites first, then the tree kernel with the trained and the random-init network and the single-game path
python -m pytest tests/test_gpu_tree_parity.py tests/test_gpu_dropin.py tests/test_gpu_edges.py tests/test_gpu_soak.py tests/test_gpu_rollout.py -x -q > gpurun_out/r4n_pytest.log 2>&1; tail -5 gpurun_out/r4n_pytest.log
for rep in 1 2; do
  python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r4n_bench.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4n_bench.json')); t=d['trained_net']
print('random-init: value %.0f tree %.1f us | trained: value %.0f depth %.2f trunk %.4f ms tree %.1f us share %.3f' % (d['value'], d['roofline_tree']['avg_launch_ms']*1e3, t['value'], t['mean_select_depth'], t['trunk_avg_launch_ms'], t['roofline_tree']['avg_launch_ms']*1e3, t['roofline_tree']['time_share']))"
done
python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"
python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"
