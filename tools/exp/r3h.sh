# round 3: single-game latency path after the prefetch changes (conv epilogue operands, head FC weights)
python tools/time_single_game.py 2>&1 | tail -1
python tools/time_single_game.py --games 16 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3h_prof -o p -- python $GRAFT_REPO_ROOT/tools/time_single_game.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find $GRAFT_REPO_ROOT/gpurun_out/r3h_prof -name "*.db" | head -1) | head -8
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3h_prof
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_net.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
