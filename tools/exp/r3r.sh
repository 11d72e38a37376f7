cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3r_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --board 15 --games 1024 --sims 100 --blocks 10 --steps 1 --warmup 0 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find $GRAFT_REPO_ROOT/gpurun_out/r3r_prof -name "*.db" | head -1) | head -7
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3r_prof
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -2
