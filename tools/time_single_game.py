#!/usr/bin/env python3
"""Latency path timing: G concurrent games (default 1), S sims/move, native PVNet.
    python tools/time_single_game.py [--games 1] [--moves 8] [--board 9] [--blocks 4] [--sims 400] [--fp16-grid]
--fp16-grid rounds the 3x3 conv weights to fp16 numbers first (two-product kernels, ao_net_products)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from alpha_omok_amd.engine import Engine, Net

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=1)
ap.add_argument("--moves", type=int, default=8)
ap.add_argument("--board", type=int, default=9)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--planes", type=int, default=128)
ap.add_argument("--sims", type=int, default=400)
ap.add_argument("--mode", type=int, default=0, help="ao_net_set_mode: 0 auto, 1, 2, 3, 4")
ap.add_argument("--fp16-grid", action="store_true")
a = ap.parse_args()
import torch
from alpha_omok_amd.pvnet import PVNet
torch.manual_seed(0)
model = PVNet(a.blocks, 5, a.planes, a.board)   # PyTorch default init, as bench.py
model.eval()
if a.fp16_grid:
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() == 4 and p_.shape[2] == 3:
                p_.copy_(p_.half().float())
net = model.to_native(0)
net.set_mode(a.mode)
eng = Engine(a.board, a.sims, 5, games=a.games, noise=True)
eng.seed_all(np.arange(a.games))
tau = np.ones(a.games, np.int8)
eng.search(net, tau=tau); eng.play(); eng.sync()
t0 = time.perf_counter()
for _ in range(a.moves):
    eng.search(net, tau=tau)
    eng.play()
eng.sync()
dt = time.perf_counter() - t0
print("mode %d games %d: %.2f ms/move, %.1f us/sim, %.1f move-decisions/s, %d products" % (
    a.mode, a.games, dt / a.moves * 1e3, dt / a.moves / a.sims * 1e6, a.games * a.moves / dt, net.products()[0]))
