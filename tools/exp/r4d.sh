# round 4: k_layer16hk v2 (rotating window, de-phased exchange, KS = 2 for 65..128 groups) against k_layer16h; tree-kernel phases with the trained net
python -m pytest tests/test_gpu_net.py -x -q -k "ksplit" > gpurun_out/r4d_pytest.log 2>&1; tail -3 gpurun_out/r4d_pytest.log
for b in 640 768 896 1024 1280 1536 2048 2560; do
  for ks in "0,0,0" "1,64,128" "1,64,160"; do
    echo -n "boards $b AO_KSPLIT=$ks: "; AO_KSPLIT=$ks python tools/time_net.py $b 4 9 0 2>&1 | grep forward
  done
done > gpurun_out/r4d_ksplit.txt
cat gpurun_out/r4d_ksplit.txt | cut -c1-150
cat > /tmp/deep.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '.')
from alpha_omok_amd.engine import Engine
from alpha_omok_amd.pvnet import PVNet
m = PVNet(4, 5, 128, 9); m.load_state_dict(torch.load('profiles/r4_trained_9x9_4block.pt', map_location='cpu')); m.eval()
net = m.to_native(0)
G = 4096
eng = Engine(9, 400, 5, games=G, noise=True); eng.seed_all(np.arange(G, dtype=np.uint32) + 7)
ply = 0
for t in range(11):
    eng.search(net, tau=np.full(G, 1 if t < 6 else 0, np.int8)); st = eng.search_stats(); eng.play()
    print("ply", t, "depth %.2f" % (st['levels'] / max(st['evaluated'] + st['terminal'], 1)), flush=True)
PY
AO_LIB_TAG=prof AO_PROF_TREE=1 python /tmp/deep.py 2>&1 | grep -v amdgpu | tail -24 > gpurun_out/r4d_tree_deep_phases.txt
cat gpurun_out/r4d_tree_deep_phases.txt | cut -c1-330
