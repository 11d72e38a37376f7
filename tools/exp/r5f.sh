# round 5: trained-net leg: static rows / rows per simulation without over-subscription (is the clock effect of 11 % fewer groups worth the atomics?) / over-subscribed
for rep in 1 2; do
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r5f_bench.json')); t=d['trained_net']; s=t['static_rows']
print('headline %.0f | static %.0f (%.1f ms/step, trunk %.3f ms, tree %.1f us) | over %.0f (%.1f ms/step, %.1f launches/move, fill %.3f, waits %d, tree %.1f us, trunk %.3f ms) ratio %.3f' % (d['value'], s['value'], s['ms_per_step'], s['trunk_avg_launch_ms'], s['roofline_tree']['avg_launch_ms']*1e3, t['value'], t['ms_per_step'], t['network_launches_per_move'], t['batch_fill'], t['leaves_that_waited_a_launch'], t['roofline_tree']['avg_launch_ms']*1e3, t['trunk_avg_launch_ms'], t['vs_static_rows']))"
AO_DYNAMIC_ROWS=1 python bench.py --steps 3 --warmup 2 --oversubscribe 1 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r5f_bench_dyn.json 2> gpurun_out/r5f_bench_dyn.err
python -c "
import json; d=json.load(open('gpurun_out/r5f_bench_dyn.json')); t=d['trained_net']
print('AO_DYNAMIC_ROWS=1, 4096 games: headline %.0f (tree %.1f us) | trained %.0f (%.1f ms/step, trunk %.3f ms, tree %.1f us)' % (d['value'], d['roofline_tree']['avg_launch_ms']*1e3, t['value'], t['ms_per_step'], t['trunk_avg_launch_ms'], t['roofline_tree']['avg_launch_ms']*1e3))"
done
