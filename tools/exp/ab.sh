for rep in 1 2; do for t in "" hard hardko9 splitko9 splitko4; do echo "== tag=[$t] rep $rep"; AO_LIB_TAG=$t python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -5
