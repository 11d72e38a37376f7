# conv1 bit-plane staging split (A/B on one box) through the engine's fused search: avg launch of k_trunk16hb
for rep in 1 2; do for t in "" c1old; do echo "== tag=[$t] rep $rep"; AO_LIB_TAG=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block --no-tictactoe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:24])"; done; done
python -m pytest tests/test_gpu_net.py -x -q 2>&1 | tail -2
