# round 5: over-subscribed trained-net leg with the device-side sit-out controller: parity of the row assignments, then the A/B
python -m pytest tests/test_gpu_fused_parity.py -x -q 2>&1 | tail -5
AO_ROW_TRACE=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r5e_bench.json 2> gpurun_out/r5e_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r5e_bench.json')); t=d['trained_net']; s=t['static_rows']
print('headline %.0f | static %.0f (%.1f ms/step) | over %.0f (%.1f ms/step, %.1f launches/move, fill %.3f, waits %d, tree %.1f us, trunk %.3f ms) ratio %.3f' % (d['value'], s['value'], s['ms_per_step'], t['value'], t['ms_per_step'], t['network_launches_per_move'], t['batch_fill'], t['leaves_that_waited_a_launch'], t['roofline_tree']['avg_launch_ms']*1e3, t['trunk_avg_launch_ms'], t['vs_static_rows']))"
grep AO_ROW_TRACE gpurun_out/r5e_bench.err | tail -2 | cut -c1-6000
