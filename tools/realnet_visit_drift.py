#!/usr/bin/env python3
"""What `north_star`'s "visit counts and chosen moves bit-exact under a fixed seed" becomes when the evaluations come from the
NATIVE network instead of a replay of the reference's (agents.py:170-221: the search consumes the network's (p, v) as they are).

Tree parity is defined -- and held bit for bit -- on identical (p, v) bits (gv5 / gv6). The native forward is within 1e-5 of
torch's; PUCT compares q + u in float64 and takes an exact arg-max, so two evaluations that differ in the last bits can order two
nearly equal children differently, and from that simulation on the two searches are different (equally valid) searches of the
same position. This script measures how soon that happens: the reference's own torch-CPU searches are recorded in
tests/golden/gv14_realnet_visits.npz (tools/gen_golden.py gv14: 3 seeds x 6 plies with the random-init 4-block network of gv6,
2 seeds x 6 plies with the trained 2-block fixture -- 9x9, 400 simulations -- and 3 plies of a 15x15 game with a random-init
10-block network at 200 simulations; np.random.seed(seed) before the game), and the engine
searches the same positions under the same seeds with its own forward. Runs ON THE GPU BOX:

    python tools/realnet_visit_drift.py            # JSON on stdout

Per case and network mode (0 = the default split-fp16 path, 2 = fp32 MFMAs): per ply whether the visit vector, the chosen action
and the MT19937 position equal the reference's, the share of the visit mass that sits on the same moves, and the first ply where
the two part."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))


NAMES = {-1: "random-init 4-block (gv6)", -2: "trained 2-block fixture", -3: "random-init 10-block 15x15 (configs[4]'s shape)"}


def _network(kind):
    import torch
    from alpha_omok_amd.pvnet import PVNet
    if kind == -1:                                        # gv6's generator: torch's default init under manual_seed(0)
        torch.manual_seed(0)
        m = PVNet(4, 5, 128, 9)
    elif kind == -3:
        torch.manual_seed(1)
        m = PVNet(10, 5, 128, 15)
    else:
        from make_trained_fixture import load
        m = PVNet(2, 5, 128, 9)
        m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in load(os.path.join(REPO, "tests", "golden", "trained_2block_9x9.npz")).items()})
    m.eval()
    return m


def measure(modes=(0, 2)):
    from alpha_omok_amd.engine import Engine
    g = np.load(os.path.join(REPO, "tests", "golden", "gv14_realnet_visits.npz"))
    out = []
    nets = {}
    for ci, (B, S, kind, seed, plies, tau_thres, noise, nrec, win) in enumerate(g["meta"].tolist()):
        for mode in modes:
            if (kind, mode) not in nets:
                nets[(kind, mode)] = _network(kind).to_native(0)
                nets[(kind, mode)].set_mode(mode)
            net = nets[(kind, mode)]
            eng = Engine(B, S, 5, games=1, noise=bool(noise))
            eng.seed(0, seed)
            rec = dict(case=ci, network=NAMES[kind], seed=seed, net_mode=mode, sims=S, recorded_plies=nrec, plies=[], first_ply_parted=None)
            for t in range(nrec):
                root = [int(x) for x in g["c%d_root" % ci][t] if x >= 0]
                eng.set_root(0, root)
                pi, vis, pol = eng.search(net, tau=np.array([1 if t < tau_thres else 0], np.int8))
                act, w = eng.play()
                ref_vis = g["c%d_visit" % ci][t]
                same_vis = bool(np.array_equal(vis[0], ref_vis))
                same_act = int(act[0]) == int(g["c%d_action" % ci][t])
                _, pos, _, _ = eng.get_rng_state(0)
                same_pos = int(pos) == int(g["c%d_mt_pos" % ci][t])
                overlap = float(np.minimum(vis[0], ref_vis).sum() / max(ref_vis.sum(), 1.0))
                rec["plies"].append(dict(ply=t, visits_equal=same_vis, action_equal=same_act, mt_pos_equal=same_pos,
                                         visit_mass_on_the_same_moves=round(overlap, 4), visits=int(vis[0].sum()), ref_visits=int(ref_vis.sum()),
                                         argmax_equal=int(np.argmax(vis[0])) == int(np.argmax(ref_vis))))
                if rec["first_ply_parted"] is None and not (same_vis and same_act and same_pos):
                    rec["first_ply_parted"] = t
                if not (same_act and same_pos):
                    break                                 # another game from here on: nothing left to compare ply by ply
            eng.close()
            out.append(rec)
    for n in nets.values():
        n.close()
    return out


if __name__ == "__main__":
    print(json.dumps(measure(), indent=1))
