#!/bin/bash
# round 6: heads' 1x1 convs inside k_boardh, planes up to 512, failure-mode tests of bench.py
python -m pytest tests/test_gpu_net.py tests/test_gpu_w16.py tests/test_gpu_realnet_drift.py tests/test_gpu_multirank.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r6d_tests.txt
python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game --no-trained-net > gpurun_out/r6d_bench.json 2> gpurun_out/r6d_bench_err.txt
tail -n 12 gpurun_out/r6d_tests.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6d_bench.json').read().strip().splitlines()[-1])
w=d['wide_board']; print(w.get('value'), w.get('roofline',{}).get('avg_launch_ms'), w.get('fp16grid',{}).get('value'), w.get('fp16grid',{}).get('roofline',{}).get('avg_launch_ms'), w.get('error'))
PY
