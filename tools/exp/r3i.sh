# round 3: heads 1x1 loop with batched loads + conv1 bit-plane staging split across slabs
echo "== AO_PROF phase cycles (wave 0..7), default format"; AO_LIB_TAG=prof AO_PROF_PRINT=1 python tools/time_net.py 4096 4 9 5 2>&1 | grep -E "AO_PROF|boards" | tail -13
for rep in 1 2 3; do python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -3
