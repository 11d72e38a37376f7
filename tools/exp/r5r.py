"""round 5: where do the 0.15 - 0.25 s of DeviceReplay.extend_augmented_moves / extend_augmented_arrays go (71 k samples of 2048 games)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from alpha_omok_amd import utils
from alpha_omok_amd.replay import DeviceReplay

rng = np.random.default_rng(0)
E, A = 2048, 81
lens = rng.integers(20, 50, E)
moves = np.full((E, A), -1, np.int32)
for e in range(E):
    moves[e, :lens[e]] = rng.permutation(A)[:lens[e]]
ep_of = np.repeat(np.arange(E), lens)
ply_of = np.concatenate([np.arange(l) for l in lens])
n = ep_of.size
pis = rng.random((n, A))
z = rng.choice([-1.0, 0.0, 1.0], n)
for cap in (int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 400000):
    mem = DeviceReplay(9, 17, cap)
    for rep in range(4):
        t0 = time.perf_counter()
        mem.extend_augmented_moves(moves, ep_of, ply_of, pis, z)
        t1 = time.perf_counter()
        st = utils.states_of_episodes(moves, ep_of, ply_of, 9, 17)
        t2 = time.perf_counter()
        mem.extend_augmented_arrays(st, pis, z)
        t3 = time.perf_counter()
        print("capacity %d, %d samples: from moves %.3f s | host states %.3f s + upload and write %.3f s" % (cap, n, t1 - t0, t2 - t1, t3 - t2), flush=True)
    mem.close()
