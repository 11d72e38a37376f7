# round 5: descent budget per launch of the tree kernel (AO_DESCENT_BUDGET=n: a descent pauses after n levels and resumes in the next launch), trained-net
# leg of the bench, over-subscribed; parity first (the row-assignment test runs its dynamic engines with the budget too)
AO_DESCENT_BUDGET=6 python -m pytest tests/test_gpu_fused_parity.py -x -q -k "row_assignments" 2>&1 | tail -3
for b in 0 48 32 24 16; do
AO_DESCENT_BUDGET=$b python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r5l_bench.json 2> gpurun_out/r5l_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r5l_bench.json')); t=d['trained_net']; s=t['static_rows']
print('budget $b: static %.0f | over %.0f (%.1f ms/step, %.1f launches/move, fill %.3f, waits %d, tree %.1f us, trunk %.3f ms) ratio %.3f' % (s['value'], t['value'], t['ms_per_step'], t['network_launches_per_move'], t['batch_fill'], t['leaves_that_waited_a_launch'], t['roofline_tree']['avg_launch_ms']*1e3, t['trunk_avg_launch_ms'], t['vs_static_rows']))"
done
