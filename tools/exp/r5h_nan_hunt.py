"""round 5: where does the NaN loss of the from-scratch training run (tools/exp/r5g.sh B, iteration 3) come from?
Same loop at the same sizes, every call's samples and every mini-batch loss checked."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import alpha_omok_amd.main as m

rows = sys.argv[1] if len(sys.argv) > 1 else "auto"
carry = (sys.argv[2] == "carry") if len(sys.argv) > 2 else True
over = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
m.BATCH_SIZE, m.LR, m.L2, m.MEMORY_SIZE, m.TRAIN_STEPS = 512, 1e-3, 1e-4, 2_000_000, 800
m.configure(board_size=9, n_mcts=400, n_blocks=4, out_planes=128, seed=0, device_replay=True, carry_over=carry, oversubscribe=over, rows=rows)
for it in range(6):
    t0 = time.time()
    out = m.self_play(2048)
    cm = list(m.cur_memory)
    pis = np.stack([c[1] for c in cm]); zs = np.array([c[2] for c in cm]); st = np.stack([c[0] for c in cm])
    bad_pi = np.flatnonzero(~np.isfinite(pis).all(axis=1))
    psum = pis.sum(axis=1)
    print("iter %d: %d samples in %.1f s, rows %s cap %s, terminal share %.3f | non-finite pi rows %d, pi sums in [%.6f, %.6f], z in %s, states in [%g, %g]" % (
        it, len(cm), time.time() - t0, m.ROWS, getattr(m._engine, "row_cap", None), m._terminal_share, bad_pi.size, psum.min(), psum.max(), sorted(set(zs.tolist())), st.min(), st.max()), flush=True)
    if bad_pi.size:
        print("  first bad rows:", bad_pi[:8], pis[bad_pi[0]])
    losses = m.train(m.N_EPOCHS, it)
    L = np.array(losses)
    bad = np.flatnonzero(~np.isfinite(L).all(axis=1))
    print("  train: %d steps, mean loss %s, non-finite steps %d (first %s), max loss %.3f" % (len(L), np.nanmean(L, axis=0).round(4), bad.size, bad[:3], np.nanmax(L[:, 0])), flush=True)
    pbad = sum(int((~torch.isfinite(p)).sum()) for p in m.Agent.model.parameters())
    print("  non-finite parameters: %d" % pbad, flush=True)
    m.reset_iter(m.result, m.cur_memory)
    if bad.size or pbad:
        break
