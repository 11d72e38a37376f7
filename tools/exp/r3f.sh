python -m pytest tests/test_gpu_replay.py tests/test_gpu_dropin.py tests/test_gpu_tree_parity.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -8
python tools/time_self_play.py 4096 400 4 1 2>&1 | tail -2
