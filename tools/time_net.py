"""Time the native PVNet forward alone (no search): python tools/time_net.py [boards] [blocks] [board] [mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alpha_omok_amd.pvnet import PVNet

boards = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 9
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.manual_seed(0)
net = PVNet(nb, 5, 128, B).eval().to_native(0)
net.set_mode(mode)
x = (torch.rand(boards, 5, B, B, device="cuda") < 0.3).float()
for _ in range(3):
    net(x)
torch.cuda.synchronize()
net.conv_timing(True)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    net(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
ms, cnt = net.conv_timing(False)
name, flop = net.dominant_kernel(boards)
print("%d boards: forward %.3f ms; %s: %.3f ms/launch, %.1f TFLOP/s" % (boards, dt * 1e3, name[:40], ms / max(cnt, 1), flop / (ms / max(cnt, 1) * 1e-3) / 1e12))
