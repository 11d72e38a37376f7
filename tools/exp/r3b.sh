# round 3: the 3-byte activation format of the resident trunk -- correctness, error against fp64, launch time
python tools/check_trunk_fmt.py 2>&1 | tail -20
for rep in 1 2 3; do for f in 0 1; do echo "== AO_TRUNK_FMT=$f rep $rep"; AO_TRUNK_FMT=$f python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -8
