# round 5 final measurement set (one box): GPU suite + smoke, driver-style bench (twice: cpu_baseline stability), rocprof stats + PMC (profile_round /
# profile_deep), 15x15 (bench + kernel stats + traffic), forward by batch size (9x9 and 15x15)
python -m pytest tests -m gpu -x -q > gpurun_out/r5z_pytest.log 2>&1; tail -3 gpurun_out/r5z_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r5z_bench.json 2> gpurun_out/r5z_bench.err; tail -c 900 gpurun_out/r5z_bench.json; echo
python bench.py --no-trained-net --no-ten-block --no-fp32-compare --no-single-game --no-tictactoe --no-wide-board > gpurun_out/r5z_bench_again.json 2> /dev/null
python -c "
import json
for f in ('r5z_bench.json', 'r5z_bench_again.json'):
    d = json.load(open('gpurun_out/' + f)); c = d['cpu_baseline']
    print(f, 'value %.0f' % d['value'], '| cpu_baseline %.3f /s on %d threads, calibration %s' % (c['value'], c['cores'], c.get('calibration', {}).get('threads')))"
python tools/profile_round.py r5z > gpurun_out/r5z_profile_round.log 2>&1; tail -2 gpurun_out/r5z_profile_round.log
python tools/profile_deep.py r5z > gpurun_out/r5z_profile_deep.log 2>&1; tail -3 gpurun_out/r5z_profile_deep.log
rm -rf gpurun_out/profiles_r5z/raw_*
python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net > gpurun_out/r5z_bench_15x15.json 2>/dev/null; tail -c 400 gpurun_out/r5z_bench_15x15.json; echo
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/r5z15 -o s15 -- python /root/repo/bench.py --board 15 --games 1024 --sims 40 --blocks 10 --steps 2 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py stats $(find /tmp/r5z15 -name "*.db" | head -1) > /root/repo/gpurun_out/r5z_kernel_stats_15x15.txt 2>&1; head -12 /root/repo/gpurun_out/r5z_kernel_stats_15x15.txt | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d /tmp/r5z15_$c -o p -- python /root/repo/bench.py --board 15 --games 1024 --sims 20 --blocks 10 --steps 1 --warmup 0 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe --no-trained-net > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py pmc $(find /tmp/r5z15_$c -name "*.db" | head -1) 2>&1 | grep -i "boardh\|layer16h" | head -4
done > /root/repo/gpurun_out/r5z_pmc_15x15.txt 2>&1; cat /root/repo/gpurun_out/r5z_pmc_15x15.txt
cd /root/repo
for b in 512 768 1024 2048 4096; do python tools/time_net.py $b 4 9 0 2>&1 | grep forward; done > gpurun_out/r5z_forward_by_batch.txt
for b in 64 128 256 512 1024 2048; do python tools/time_net.py $b 10 15 0 2>&1 | grep forward; done >> gpurun_out/r5z_forward_by_batch.txt; cut -c1-140 gpurun_out/r5z_forward_by_batch.txt
