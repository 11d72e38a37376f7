"""Time the weight re-export (what main.train / bench.py --train-step do after every optimiser step):
    python tools/time_export.py [blocks] [board]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alpha_omok_amd.pvnet import PVNet
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 9
model = PVNet(nb, 5, 128, B).cuda()
net = model.to_native(0)
x = (torch.rand(64, 5, B, B, device="cuda") < 0.3).float()
net(x)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    sd = model.state_dict()
    t1 = time.perf_counter()
    net.load_state_dict(sd)
    t2 = time.perf_counter()
    net(x); torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("state_dict %.2f ms, load_state_dict (copy + repack + upload) %.2f ms, first forward after %.2f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
