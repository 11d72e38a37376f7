# round 4: write-through activation stores (AO_AUX_ST = 16: sc1, 17: sc0 sc1) in the RESIDENT trunk: does the step get shorter (less dirty L2 to
# write back at the end of the launch)? bench.py --steps 10, variants interleaved, three times
for rep in 1 2 3; do for tag in "" st16 st17; do
  AO_LIB_TAG=$tag python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game --no-trained-net > gpurun_out/r4t_b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4t_b.json')); r=d['roofline']
print('lib %-5s value %.0f  ms/step %.2f  trunk (events) %.4f ms  tree %.1f us' % ('$tag' or '-', d['value'], d['ms_per_step'], r['avg_launch_ms'], d['roofline_tree']['avg_launch_ms']*1e3))"
done; done
