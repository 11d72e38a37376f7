# round 4 final measurement set (one box): GPU suite + smoke, driver-style bench, rocprof stats + PMC (profile_round / profile_deep), 15x15,
# main.self_play end to end (synchronous; random-init and the trained net), forward at medium batches
python -m pytest tests -m gpu -x -q > gpurun_out/r4z_pytest.log 2>&1; tail -3 gpurun_out/r4z_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err; tail -c 700 gpurun_out/r4z_bench.json; echo
python tools/profile_round.py r4z > gpurun_out/r4z_profile_round.log 2>&1; tail -2 gpurun_out/r4z_profile_round.log
python tools/profile_deep.py r4z > gpurun_out/r4z_profile_deep.log 2>&1; tail -3 gpurun_out/r4z_profile_deep.log
rm -rf gpurun_out/profiles_r4z/raw_*
python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe > gpurun_out/r4z_bench_15x15.json 2>/dev/null; tail -c 300 gpurun_out/r4z_bench_15x15.json; echo
python tools/time_self_play.py 4096 400 4 1 0 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r4z_self_play.txt
cat gpurun_out/r4z_self_play.txt
for b in 512 768 1024 2048 4096; do python tools/time_net.py $b 4 9 0 2>&1 | grep forward; done > gpurun_out/r4z_forward_by_batch.txt; cat gpurun_out/r4z_forward_by_batch.txt | cut -c1-140
