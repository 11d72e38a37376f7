"""Time the native PVNet forward alone (no search): python tools/time_net.py [boards] [blocks] [board] [mode] [--fp16-grid] [--weights state_dict.pt]
--fp16-grid rounds the 3x3 conv weights to fp16 numbers first (the library then plans the two-product kernels: ao_net_products)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alpha_omok_amd.pvnet import PVNet

grid = "--fp16-grid" in sys.argv
wpath = sys.argv[sys.argv.index("--weights") + 1] if "--weights" in sys.argv else None
sys.argv = [a for i, a in enumerate(sys.argv) if a != "--fp16-grid" and a != "--weights" and (i == 0 or sys.argv[i - 1] != "--weights")]
boards = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 9
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
torch.manual_seed(0)
model = PVNet(nb, 5, 128, B).eval()
if wpath:
    model.load_state_dict(torch.load(wpath, map_location="cpu", weights_only=True))
if grid:
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() == 4 and p_.shape[2] == 3:
                p_.copy_(p_.half().float())
net = model.to_native(0)
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
print("%d boards: forward %.3f ms; %s: %.3f ms/launch, %.1f TFLOP/s, %d products" % (boards, dt * 1e3, name[:40], ms / max(cnt, 1), flop / (ms / max(cnt, 1) * 1e-3) / 1e12, net.products()[0]))
