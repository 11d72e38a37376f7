python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -8
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_tree']['avg_launch_ms'], d['roofline_tree']['achieved'])"
