"""round 5: the README's "training loop at scale" recipe, at a size that takes seconds (smoke check of main.run with every switch on)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import alpha_omok_amd.main as m
m.MEMORY_SIZE, m.BATCH_SIZE, m.TRAIN_STEPS, m.GAMES_PER_ITER = 200_000, 128, 40, 256
m.MAX_CONCURRENT = 256
m.configure(board_size=9, n_mcts=64, n_blocks=2, device_replay=True, oversubscribe=1.25, carry_over=True, overlap_train=True)
t0 = time.time()
n = m.run(total_iter=5, n_selfplay=256, save_every=2, directory="/tmp/r5zf_data")
print("iterations", n, "optimiser steps", m.step, "replay", len(m.rep_memory), "played ahead", m.played_ahead[0], "train_wait %.2f s" % m.phase_seconds['train_wait'],
      "slots", m._engine.G, "seconds %.1f" % (time.time() - t0), sorted(os.listdir("/tmp/r5zf_data")))
assert n == 5 and m.step == 4 * 40 and m._train_job is None
