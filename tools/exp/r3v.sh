# round 3 final measurement set (one box): driver-style bench, rocprof stats + PMC (profile_round / profile_deep), 15x15, self-play API (also carry-over), fused per-game step on / off
python bench.py > gpurun_out/r3v_bench.json 2> gpurun_out/r3v_bench.err; tail -c 600 gpurun_out/r3v_bench.json
python tools/profile_round.py r3v > gpurun_out/r3v_profile_round.log 2>&1; tail -2 gpurun_out/r3v_profile_round.log
python tools/profile_deep.py r3v > gpurun_out/r3v_profile_deep.log 2>&1; tail -3 gpurun_out/r3v_profile_deep.log
rm -rf gpurun_out/profiles_r3v/raw_*
python bench.py --board 15 --games 1024 --sims 800 --blocks 10 --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe > gpurun_out/r3v_bench_15x15.json 2>/dev/null; tail -c 400 gpurun_out/r3v_bench_15x15.json
python tools/time_self_play.py 4096 400 4 1 0 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r3v_self_play.txt
python tools/time_self_play.py 8192 400 4 1 0 2>&1 | grep -v amdgpu | tail -8 >> gpurun_out/r3v_self_play.txt
cat gpurun_out/r3v_self_play.txt
python tools/time_self_play.py 4096 400 4 1 -3 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/r3v_self_play.txt
for g in 1 8 24 48; do for f in 1 0; do echo -n "games $g AO_FUSED_STEP=$f  "; AO_FUSED_STEP=$f python tools/time_single_game.py --moves 8 --games $g 2>&1 | grep "us/sim"; done; done > gpurun_out/r3v_fused_step.txt
tail -3 gpurun_out/r3v_self_play.txt; cat gpurun_out/r3v_fused_step.txt
