#!/bin/bash
# round 6: two-product kernels -- new tests first, then the whole GPU suite, then the bench legs that compare 2 and 3 products
mkdir -p gpurun_out
python -m pytest tests/test_gpu_w16.py tests/test_gpu_production_loop.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r6b_new_tests.txt
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r6b_suite.txt
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare --no-single-game > gpurun_out/r6b_bench.json 2> gpurun_out/r6b_bench_err.txt
tail -n 5 gpurun_out/r6b_new_tests.txt gpurun_out/r6b_suite.txt
