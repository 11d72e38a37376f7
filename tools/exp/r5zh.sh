# round 5: rocprofv3 evidence for the TRAINED-net workload (deep trees, over-subscribed rows) under the final sources: kernel stats of bench.py's trained_net
# leg, and FETCH_SIZE / WRITE_SIZE of the tree and trunk kernels there (separate passes)
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-wide-board"
rocprofv3 --kernel-trace --stats -d /tmp/r5zh -o s -- $B > /tmp/r5zh_bench.json 2>/dev/null
python /root/repo/tools/rocpd_summary.py stats $(find /tmp/r5zh -name "*.db" | head -1) > /root/repo/gpurun_out/r5zh_kernel_stats_with_trained_net.txt 2>&1; head -14 /root/repo/gpurun_out/r5zh_kernel_stats_with_trained_net.txt | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d /tmp/r5zh_$c -o p -- $B > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py pmc $(find /tmp/r5zh_$c -name "*.db" | head -1) 2>&1 | grep -i "expand_select\|trunk16hb\|k_play\|k_reroot" | head -6
done > /root/repo/gpurun_out/r5zh_pmc_with_trained_net.txt 2>&1; cat /root/repo/gpurun_out/r5zh_pmc_with_trained_net.txt | cut -c1-170
python -c "
import json
d=json.load(open('/tmp/r5zh_bench.json')); t=d['trained_net']
print('under rocprofv3: headline', round(d['value']), '| trained net', round(t['value']), 'static', round(t['static_rows']['value']), 'depth', round(t['mean_select_depth'],2), 'terminal', round(t['terminal_leaf_fraction'],3))
"
