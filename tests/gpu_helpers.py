"""Shared helpers of the -m gpu parity tests: drive the HIP engine through its C ABI with a
host-side evaluator (the oracle's exact-arithmetic stub or a replay of recorded evaluations)."""
import numpy as np


class HostEvalRunner:
    """Runs move decisions on an alpha_omok_amd.engine.Engine with p/v supplied from the host."""

    def __init__(self, engine):
        import torch
        self.torch = torch
        self.e = engine
        dev = torch.device("cuda", engine.device)
        G, C, B, A = engine.G, engine.inplanes, engine.board_size, engine.A
        self.planes = torch.zeros((G, C, B, B), dtype=torch.float32, device=dev)
        self.policy = torch.zeros((G, A), dtype=torch.float32, device=dev)
        self.value = torch.zeros((G,), dtype=torch.float32, device=dev)
        self.h_policy = np.zeros((G, A), np.float32)
        self.h_value = np.zeros((G,), np.float32)

    def move(self, eval_fn, tau=None, active=None):
        """eval_fn(game, sim, planes[C,B,B]) -> (policy[A] f32, value f32). Returns end_move()."""
        e, torch = self.e, self.torch
        e.begin_move(active)
        sim = 0
        while e.sims_left() > 0:
            e.collect_leaves(self.planes.data_ptr())
            e.sync()
            pl = self.planes.cpu().numpy()
            for g in range(e.G):
                if active is not None and not active[g]:
                    continue
                p, v = eval_fn(g, sim, pl[g])
                self.h_policy[g] = p
                self.h_value[g] = v
            self.policy.copy_(torch.from_numpy(self.h_policy))
            self.value.copy_(torch.from_numpy(self.h_value))
            torch.cuda.synchronize()
            e.apply_evals(self.policy.data_ptr(), self.value.data_ptr())
            sim += 1
        return e.end_move(tau)


def children_by_action(ch, A):
    """root_children() dict -> dense [A] arrays like the oracle's Agent.children()."""
    out = {k: np.zeros(A) for k in ("n", "w", "q", "p")}
    for i, a in enumerate(ch["action"].tolist()):
        for k in out:
            out[k][a] = ch[k][i]
    out["order"] = ch["action"].copy()
    return out
