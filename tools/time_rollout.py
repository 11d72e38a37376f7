#!/usr/bin/env python3
"""Throughput of the rollout agents (PUCTAgent / UCTAgent searches, one kernel per get_pi) on the GPU.
    python tools/time_rollout.py [--board 9] [--sims 400] [--games 4096] [--mode 0]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_omok_amd.rollout import RolloutEngine

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=9)
ap.add_argument("--sims", type=int, default=400)
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--mode", type=int, default=0)
a = ap.parse_args()
eng = RolloutEngine(a.board, a.sims, a.mode, games=a.games)
for g in range(a.games):
    eng.seed(g, g)
roots = [(0,)] * a.games
eng.search(roots)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    pi, stat, act = eng.search(roots)
dt = (time.perf_counter() - t0) / reps
print("GPU  mode %d %dx%d %d sims: %d searches in %.1f ms -> %.0f searches/s, %.2f M playouts/s" % (
    a.mode, a.board, a.board, a.sims, a.games, dt * 1e3, a.games / dt, a.games * a.sims / dt / 1e6))
