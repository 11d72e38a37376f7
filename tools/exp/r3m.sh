# round 3 measurement set (one box): driver-style bench, rocprof stats + PMC (profile_round / profile_deep), 15x15, self-play API
python bench.py > gpurun_out/r3m_bench.json 2> gpurun_out/r3m_bench.err; tail -c 600 gpurun_out/r3m_bench.json
python tools/profile_round.py r3m > gpurun_out/r3m_profile_round.log 2>&1; tail -2 gpurun_out/r3m_profile_round.log
python tools/profile_deep.py r3m > gpurun_out/r3m_profile_deep.log 2>&1; tail -3 gpurun_out/r3m_profile_deep.log
rm -rf gpurun_out/profiles_r3m/raw_*
python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe > gpurun_out/r3m_bench_15x15.json 2>/dev/null; tail -c 400 gpurun_out/r3m_bench_15x15.json
python tools/time_self_play.py 4096 400 4 1 0 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r3m_self_play.txt
python tools/time_self_play.py 8192 400 4 1 0 2>&1 | grep -v amdgpu | tail -8 >> gpurun_out/r3m_self_play.txt
cat gpurun_out/r3m_self_play.txt
