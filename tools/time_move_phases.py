#!/usr/bin/env python3
"""Where one move decision of the headline workload (4096 games, 9x9, 400 sims) spends its time OUTSIDE the simulation loop:
   python tools/time_move_phases.py [--games 4096] [--steps 6]     (AO_LAUNCH_TIMING=1 prints the loop's own share from inside ao_search)"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_omok_amd.engine import Engine
from alpha_omok_amd.pvnet import PVNet

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--sims", type=int, default=400)
ap.add_argument("--weights", default=None, help="state_dict of a 4-block / 128-plane 9x9 PVNet (e.g. profiles/r4_trained_9x9_4block.pt): deep trees, big subtrees to re-root")
ap.add_argument("--warm-plies", type=int, default=2, help="untimed move decisions before the timed ones")
ap.add_argument("--events", type=int, default=0, help="1: HIP-event timing of the trunk and tree launches on, as bench.py has it")
a = ap.parse_args()
torch.manual_seed(0)
model = PVNet(4, 5, 128, 9)
if a.weights:
    model.load_state_dict(torch.load(a.weights, map_location="cpu", weights_only=True))
model = model.cuda().eval()
net = model.to_native(0)
G = a.games
eng = Engine(9, a.sims, 5, games=G, noise=True, device=0)
eng.seed_all(np.arange(G, dtype=np.uint32))
ply = np.zeros(G, np.int64)
if a.events:
    eng.tree_timing(True); net.conv_timing(True)
nxt = G
T = dict(search=0.0, stats=0.0, play=0.0, refill=0.0)
def tick():
    eng.sync()
    return time.perf_counter()
for s in range(a.steps + a.warm_plies):
    t0 = tick()
    eng.search(net, tau=(ply < 6).astype(np.int8))
    t1 = tick()
    eng.search_stats()
    t2 = tick()
    act, win = eng.play()
    t3 = tick()
    ply += 1
    done = win != 0
    if done.any():
        eng.reset(done.astype(np.uint8))
        for g in np.nonzero(done)[0]:
            eng.seed(int(g), nxt); nxt += 1
        ply[done] = 0
    t4 = tick()
    if s >= a.warm_plies:
        T["search"] += t1 - t0; T["stats"] += t2 - t1; T["play"] += t3 - t2; T["refill"] += t4 - t3
n = a.steps
print("per move decision of %d games (ms): search %.2f  search_stats %.2f  play %.2f  reset+seed of finished games %.2f" % (
    G, T["search"] / n * 1e3, T["stats"] / n * 1e3, T["play"] / n * 1e3, T["refill"] / n * 1e3))
