# round 4: k_layer16hk v3 (resident weights made visible to the wait-count pass; KS = 2 without the wave-dependent exchange position),
# the game header handed from expansion to selection in registers (single game, tree kernel), GPU suite
python -m pytest tests -m gpu -x -q > gpurun_out/r4e_pytest.log 2>&1; tail -4 gpurun_out/r4e_pytest.log
for b in 640 768 896 1024 1280 1536 2048; do
  for ks in "0,0,0" "1,64,128"; do
    echo -n "boards $b AO_KSPLIT=$ks: "; AO_KSPLIT=$ks python tools/time_net.py $b 4 9 0 2>&1 | grep forward
  done
done > gpurun_out/r4e_ksplit.txt
cat gpurun_out/r4e_ksplit.txt | cut -c1-150
for i in 1 2 3; do python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"; done > gpurun_out/r4e_single.txt
for g in 8 48; do python tools/time_single_game.py --moves 6 --games $g 2>&1 | grep "us/sim"; done >> gpurun_out/r4e_single.txt
cat gpurun_out/r4e_single.txt
python bench.py --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
python -c "import json; d=json.load(open('gpurun_out/r4e_bench.json')); print(d['value'], d['roofline']['avg_launch_ms'] if 'avg_launch_ms' in d['roofline'] else d['roofline'], d['roofline_tree']['avg_launch_ms'], d.get('single_game'), json.dumps(d.get('trained_net'))[:900])"
