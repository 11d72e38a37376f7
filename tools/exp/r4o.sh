# round 4: heads_board_one_round (every memory request of the heads at the top of the kernel) on top of r4n: parity suites, single-game timing
python -m pytest tests/test_gpu_net.py tests/test_gpu_tree_parity.py tests/test_gpu_dropin.py tests/test_gpu_edges.py -x -q > gpurun_out/r4o_pytest.log 2>&1; tail -5 gpurun_out/r4o_pytest.log
python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"
python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"
python tools/time_single_game.py --moves 10 --games 8 2>&1 | grep "us/sim"
python tools/time_single_game.py --moves 10 --games 32 2>&1 | grep "us/sim"
