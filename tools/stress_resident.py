"""Long stress of the group-resident split-fp16 trunk: 30000 back-to-back 4096-board forwards, 8 inputs in random
order, compared on the device with the first result of the same input (no host sync inside the loop).
    python tools/stress_resident.py      -> "long stress: 30000 forwards, 0 mismatching" (MI355X, 51 s)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from alpha_omok_amd.pvnet import PVNet
torch.manual_seed(11)
ref = PVNet(4, 5, 128, 9).eval()
net = ref.to_native(0)
rs = np.random.RandomState(5)
xs = [torch.from_numpy((rs.rand(4096, 5, 9, 9) < (0.1 + 0.1 * k)).astype(np.float32)).cuda() for k in range(8)]
net.set_mode(5)
first = []
for x in xs:
    p, v = net(x); torch.cuda.synchronize(); first.append((p.clone(), v.clone()))
bad = torch.zeros((), dtype=torch.int64, device="cuda")
order = rs.randint(0, 8, size=30000)
for k in order:
    p, v = net(xs[k])
    bad += (p != first[k][0]).any().long() + (v != first[k][1]).any().long()
torch.cuda.synchronize()
print("long stress: %d forwards, %d mismatching" % (len(order), int(bad.item())))
