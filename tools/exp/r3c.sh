# round 3: where does the 3-byte format lose what the traffic knock-out promised? phase timing + no-expansion knock-out
for f in 0 1; do echo "== AO_PROF phase cycles (wave 0..7), AO_TRUNK_FMT=$f"; AO_TRUNK_FMT=$f AO_LIB_TAG=prof AO_PROF_PRINT=1 python tools/time_net.py 4096 4 9 5 2>&1 | grep -E "AO_PROF|boards" | tail -13; done
for rep in 1 2; do for t in "0:" "1:" "1:ko11"; do f=${t%%:*}; tag=${t##*:}; echo "== AO_TRUNK_FMT=$f tag=[$tag] rep $rep"; AO_TRUNK_FMT=$f AO_LIB_TAG=$tag python tools/time_net.py 4096 4 9 5 2>&1 | tail -1; done; done
python -m pytest tests/test_gpu_net.py -x -q -k "fp16_range or full_size" 2>&1 | tail -5
