# round 5: k_boardh<15> (one board resident in LDS through all trunk convs, cells as MFMA N) against k_layer16h<15> (AO_BOARDK=0): numerics, then the
# configs[4] per-GPU shape (15x15, 10 blocks, 800 sims, 1024 games), same box, alternating
python -m pytest tests/test_gpu_net.py -x -q -k "board_resident" 2>&1 | tail -8
for rep in 1; do
for bk in 128 0; do
AO_BOARDK=$bk python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 2 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net > gpurun_out/r5i_bench15_$bk.json 2>gpurun_out/r5i_err_$bk.txt
python -c "
import json; d=json.load(open('gpurun_out/r5i_bench15_$bk.json')); r=d['roofline']
print('AO_BOARDK=$bk: %.1f move-decisions/s, %.1f ms/step | %s: %.4f ms per launch, %.1f TFLOP/s algorithmic = %.3f of peak, conv share %.3f' % (d['value'], d['ms_per_step'], r['kernel'].split(' (')[0], r['avg_launch_ms'], r['achieved'], r['frac'], r['conv_time_share']))"
done
done
