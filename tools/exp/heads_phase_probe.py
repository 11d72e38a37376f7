import os, sys
sys.path.insert(0, "/root/repo")
import torch
from alpha_omok_amd.pvnet import PVNet
torch.manual_seed(0)
net = PVNet(4, 5, 128, 9).eval().to_native(0)
x = (torch.rand(1, 5, 9, 9, device="cuda") < 0.3).float()
for _ in range(20): net(x)
torch.cuda.synchronize()
os.environ["AO_PROF_PRINT"] = "1"
for _ in range(4): net(x)
