# round 5: the 15x15 (configs[4] per-GPU shape) kernel stats and HBM traffic of k_boardh under the final source hash (the r5z set's were taken before the
# tree / replay changes; bench.match_traffic keys on the hash) -> profiles/r5za_traffic.json "other_workloads"
cd /tmp; export TMPDIR=/tmp
B="python /root/repo/bench.py --board 15 --games 1024 --blocks 10 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net"
rocprofv3 --kernel-trace --stats -d /tmp/r5zg15 -o s15 -- $B --sims 40 --steps 2 --warmup 1 > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py stats $(find /tmp/r5zg15 -name "*.db" | head -1) > /root/repo/gpurun_out/r5za_kernel_stats_15x15.txt 2>&1; head -8 /root/repo/gpurun_out/r5za_kernel_stats_15x15.txt | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d /tmp/r5zg15_$c -o p -- $B --sims 20 --steps 1 --warmup 0 > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py pmc $(find /tmp/r5zg15_$c -name "*.db" | head -1) 2>&1 | grep -i "boardh\|layer16h" | head -4
done > /root/repo/gpurun_out/r5za_pmc_15x15.txt 2>&1; cat /root/repo/gpurun_out/r5za_pmc_15x15.txt
cd /root/repo
python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net > gpurun_out/r5za_bench_15x15.json 2>/dev/null; tail -c 500 gpurun_out/r5za_bench_15x15.json; echo
