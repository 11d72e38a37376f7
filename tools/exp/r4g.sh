# round 4: the tree arena as one record per node (was six arrays): GPU suite, the tree kernel with the random-init and the trained
# network, single game; then the no-epilogue knock-outs again with their accumulators kept alive (AO_KO 14 / 15)
python -m pytest tests -m gpu -x -q > gpurun_out/r4g_pytest.log 2>&1; tail -3 gpurun_out/r4g_pytest.log
python bench.py --no-cpu-baseline --no-tictactoe --no-ten-block --no-fp32-compare > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4g_bench.json')); t=d['trained_net']
print('random-init: value %.0f  trunk %.4f ms  tree %.1f us (%.2f of algorithmic traffic n/a)  single %.1f' % (d['value'], d['roofline']['avg_launch_ms'] if 'avg_launch_ms' in d['roofline'] else -1, d['roofline_tree']['avg_launch_ms']*1e3, 0, d['single_game']['value']))
print('trained: value %.0f depth %.2f trunk %.4f ms tree %.1f us share %.3f GB/s %.0f trims %s' % (t['value'], t['mean_select_depth'], t['trunk_avg_launch_ms'], t['roofline_tree']['avg_launch_ms']*1e3, t['roofline_tree']['time_share'], t['roofline_tree']['achieved'], t['arena_trims']))
"
AO_LIB_TAG=prof AO_PROF_TREE=1 python /dev/stdin <<'PY' 2>&1 | grep -v amdgpu | tail -6 > gpurun_out/r4g_tree_deep_phases.txt
import sys, numpy as np, torch
sys.path.insert(0, '.')
from alpha_omok_amd.engine import Engine
from alpha_omok_amd.pvnet import PVNet
m = PVNet(4, 5, 128, 9); m.load_state_dict(torch.load('profiles/r4_trained_9x9_4block.pt', map_location='cpu')); m.eval()
net = m.to_native(0)
G = 4096
eng = Engine(9, 400, 5, games=G, noise=True); eng.seed_all(np.arange(G, dtype=np.uint32) + 7)
for t in range(9):
    eng.search(net, tau=np.full(G, 1 if t < 6 else 0, np.int8)); st = eng.search_stats(); eng.play()
    print("ply", t, "depth %.2f" % (st['levels'] / max(st['evaluated'] + st['terminal'], 1)), flush=True)
PY
cat gpurun_out/r4g_tree_deep_phases.txt | cut -c1-330
for i in 1 2; do python tools/time_single_game.py --moves 10 2>&1 | grep "us/sim"; done
run() {  # $1 boards $2 blocks $3 board
python - "$1" "$2" "$3" <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from alpha_omok_amd.pvnet import PVNet
boards, nb, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
net = PVNet(nb, 5, 128, B).eval().to_native(0)
net.set_mode(5)
x = (torch.rand(boards, 5, B, B, device="cuda") < 0.3).float()
for _ in range(30): net(x)
torch.cuda.synchronize()
net.conv_timing(True)
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    for _ in range(100): net(x)
    torch.cuda.synchronize(); n += 100
ms, cnt = net.conv_timing(False)
print("  forward avg ms %.4f   %s: %.4f ms per launch" % ((time.time() - t0) / n * 1e3, net.dominant_kernel(boards)[0].split(" (")[0], ms / max(cnt, 1)))
PY
sleep 4
for i in 1 2 3; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done
wait
}
for t in "" ko14 ko15; do
  echo "== 9x9 resident trunk, 4096 boards, tag=[$t]"; AO_TRUNK_FMT=0 AO_LIB_TAG=$t run 4096 4 9
done
for t in "" ko14 ko15; do
  echo "== 15x15 per-layer, 1024 boards, 10 blocks, tag=[$t]"; AO_LIB_TAG=$t run 1024 10 15
done
