#!/bin/bash
# round 6: k_boardh<15> (three products) with a block's nine taps in TWO passes (AO_BOARDH_WPASS: 36 weight registers live instead of 72): 1 = one register
# set reloaded between the passes, 2 = two sets of 36 loaded while the other multiplies. Results are correct in all three (checked against torch below).
for i in 1 2; do
for tag in "" wp1 wp2; do
  AO_LIB_TAG=$tag python tools/time_net.py 1024 10 15 0 2>/dev/null | sed "s/^/[${tag:-product}] /"
done
done | tee gpurun_out/r6i_boardh_weight_passes.txt
for tag in wp1 wp2; do
AO_LIB_TAG=$tag python - <<'PY' 2>&1 | tail -1 | tee -a gpurun_out/r6i_boardh_weight_passes.txt
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pvnet_weights
from alpha_omok_amd.pvnet import PVNet
sd = pvnet_weights.make_state_dict(2, 5, 128, 15, 9)
ref = PVNet(2, 5, 128, 15); ref.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); ref.eval()
x = (np.random.RandomState(0).rand(96, 5, 15, 15) < 0.3).astype(np.float32)
with torch.no_grad(): rp, rv = ref(torch.from_numpy(x))
net = ref.to_native(0); p, v = net(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
print("[%s] %s: max |dp| %.2e, max |dv| %.2e vs torch fp32" % (os.environ["AO_LIB_TAG"], net.dominant_kernel(96)[0][:22], float((p.cpu() - rp).abs().max()), float((v.cpu() - rv).abs().max())))
PY
done
