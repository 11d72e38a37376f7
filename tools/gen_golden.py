#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the unmodified reference.

Runs only in the build container (needs /root/reference). Nothing of the reference is copied:
the fixtures are inputs and expected outputs (data). The GPU box and the test-suite only read
the .npz files.

    python tools/gen_golden.py            # all fixtures
    python tools/gen_golden.py gv5 gv7    # a subset

Fixture ids follow SURVEY.md section 8(c): GV1 check_win, GV2 legal_actions order, GV3 state
planes / board / turn, GV4 numpy RNG stream, GV5 tree parity with the exact-arithmetic stub
evaluator, GV6 tree parity with the real PVNet (evaluations recorded for replay), GV7 PVNet
forward, GV8 augment_dataset order, GV9 main.self_play memory order + z, GV10 one train step,
GV11 rollout agents (PUCTAgent / UCTAgent.get_pi: one-hot, child visits / q, stream position),
GV12 the 3x3 UCT search of 1_tictactoe_MCTS/mcts_vs.py under Python's `random` (BASELINE configs[0]),
GV13 head-to-head matches through the reference's eval_main.Evaluator.get_action / GameState.step / del_parents / elo.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/2_AlphaOmok"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

# pygame is only needed to import env/*.py and main.py (rendering is never exercised)
_pg = types.ModuleType("pygame")
_pgl = types.ModuleType("pygame.locals")
_pgl.QUIT = 0
_pg.locals = _pgl
sys.modules.setdefault("pygame", _pg)
sys.modules.setdefault("pygame.locals", _pgl)

sys.path.insert(0, REF)
import agents as ref_agents  # noqa: E402
import model as ref_model  # noqa: E402
import utils as ref_utils  # noqa: E402

from oracle import oracle_py as O  # noqa: E402  (only its pure stub function is used here)
import pvnet_weights  # noqa: E402  (tests/pvnet_weights.py: deterministic weight generator)

ref_agents.PRINT_MCTS = False
torch.set_num_threads(1)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


class StubModel:
    """Agent.model stand-in: exact-arithmetic hash of the input planes (oracle stub)."""

    def __init__(self, mode):
        self.mode = mode
        self.log_p, self.log_v = [], []

    def eval(self):
        return self

    def __call__(self, x):
        planes = x[0].cpu().numpy().astype(np.float32)
        p, v = O.stub_eval(planes, self.mode)
        return torch.from_numpy(p[None].copy()), torch.from_numpy(np.array([v], np.float32))


class RecordingModel:
    """Wraps a real torch module and records every (p, v) it returns (GV6 replay)."""

    def __init__(self, net):
        self.net = net
        self.p, self.v = [], []

    def eval(self):
        self.net.eval()
        return self

    def __call__(self, x):
        p, v = self.net(x)
        self.p.append(p[0].numpy().copy())
        self.v.append(np.float32(v[0].item()))
        return p, v


# --------------------------------------------------------------------------------------
def gv1():
    rng = np.random.RandomState(11)
    boards, marks, wins = [], [], []

    def add(b, k):
        boards.append(np.pad(b, ((0, 15 - b.shape[0]), (0, 15 - b.shape[1]))).astype(np.int8))
        marks.append((b.shape[0], k))
        wins.append(ref_utils.check_win(b.astype(float), k))

    for B, k in ((3, 3), (9, 5), (15, 5)):
        # hand-made lines: every direction, both colours, at edges/corners, plus overline
        for colour in (1, -1):
            for r0, c0, dr, dc in ((0, 0, 0, 1), (B - 1, B - k, 0, 1), (0, B - 1, 1, 0),
                                   (B - k, 0, 1, 0), (0, 0, 1, 1), (B - k, B - k, 1, 1),
                                   (k - 1, 0, -1, 1), (B - 1, B - k, -1, 1)):
                b = np.zeros((B, B))
                for i in range(k):
                    b[r0 + i * dr, c0 + i * dc] = colour
                add(b, k)
                b2 = b.copy()
                b2[r0 + (k - 1) * dr, c0 + (k - 1) * dc] = 0  # broken line
                add(b2, k)
                b3 = b.copy()
                b3[r0 + 2 * dr, c0 + 2 * dc] = -colour  # blocked line
                add(b3, k)
        if B > 5:
            b = np.zeros((B, B))
            b[2, 1:7] = 1  # overline (six)
            add(b, k)
            b = np.zeros((B, B))
            b[1:7, 3] = -1
            add(b, k)
        # full board without a line -> draw ; full board with a line
        for t in range(40):
            b = rng.choice([1, -1], size=(B, B)).astype(float)
            add(b, k)
        # random sparse / dense boards
        for t in range(300):
            dens = rng.uniform(0.1, 1.0)
            b = rng.choice([0, 1, -1], p=[1 - dens, dens / 2, dens / 2], size=(B, B)).astype(float)
            add(b, k)
    save("gv1_check_win", boards=np.stack(boards), size_mark=np.array(marks, np.int32),
         win=np.array(wins, np.int32))


def gv2():
    rng = np.random.RandomState(22)
    recs = []
    for B in (3, 9, 15):
        A = B * B
        counts = list(range(0, A)) * 3
        for ns in counts:
            mv = rng.permutation(A)[:ns].tolist()
            order = ref_utils.legal_actions((0,) + tuple(mv), B)
            recs.append((B, mv, order))
    mvs = np.full((len(recs), 225), -1, np.int16)
    ords = np.full((len(recs), 225), -1, np.int16)
    bs = np.zeros(len(recs), np.int16)
    for i, (B, mv, order) in enumerate(recs):
        bs[i] = B
        mvs[i, :len(mv)] = mv
        ords[i, :len(order)] = order
    save("gv2_legal_order", board=bs, moves=mvs, order=ords)


def gv3():
    rng = np.random.RandomState(33)
    out = {}
    idx = 0
    for B in (3, 9, 15):
        A = B * B
        for k in list(range(0, min(A, 12))) + [A // 2, A - 1]:
            for C in (5, 3, 7):
                mv = rng.permutation(A)[:k].tolist()
                nid = (0,) + tuple(mv)
                out["m%d" % idx] = np.array([B, C] + mv, np.int32)
                out["s%d" % idx] = ref_utils.get_state_pt(nid, B, C).astype(np.float32)
                out["b%d" % idx] = ref_utils.get_board(nid, B).astype(np.int8)
                out["t%d" % idx] = np.array(ref_utils.get_turn(nid), np.int32)
                idx += 1
    out["count"] = np.array(idx)
    save("gv3_state_planes", **out)


def gv4():
    out = {}
    for seed in (0, 1, 12345, 4294967295):
        np.random.seed(seed)
        st = np.random.get_state()
        out["init_%d" % seed] = st[1][:8].copy()
        ks = [1, 2, 3, 5, 17, 64, 65, 81, 225, 1, 7]
        out["choice_k"] = np.array(ks)
        out["choice_%d" % seed] = np.array([np.random.choice(k) for k in ks * 5])
        out["dbl_%d" % seed] = np.array([np.random.random_sample() for _ in range(8)])
        out["dir81_%d" % seed] = np.random.dirichlet(10 / 81 * np.ones(81))
        out["dir17_%d" % seed] = np.random.dirichlet(10 / 81 * np.ones(17))
        out["dir225_%d" % seed] = np.random.dirichlet(10 / 225 * np.ones(225))
        out["dir9_%d" % seed] = np.random.dirichlet(10 / 9 * np.ones(9))
        out["dir4_%d" % seed] = np.random.dirichlet(10 / 9 * np.ones(4))
        p = np.random.dirichlet(np.ones(81))
        out["p_%d" % seed] = p
        out["choicep_%d" % seed] = np.array([np.random.choice(81, p=p) for _ in range(16)])
        onehot = np.zeros(81)
        onehot[37] = 1.0
        out["choice1h_%d" % seed] = np.array(np.random.choice(81, p=onehot))
        st = np.random.get_state()
        out["end_pos_%d" % seed] = np.array(st[2])
        out["end_state_%d" % seed] = st[1].copy()
    save("gv4_numpy_rng", **out)


def _search_record(agent, root_id, tau):
    """One get_pi + get_action, recording everything the parity tests compare."""
    pi = agent.get_pi(root_id, tau)
    A = agent.board_size ** 2
    cw = np.zeros(A)
    cq = np.zeros(A)
    for a in agent.tree[agent.root_id]["child"]:
        nd = agent.tree[agent.root_id + (a,)]
        cw[a] = nd["w"]
        cq[a] = nd["q"]
    order = np.full(A, -1, np.int32)
    ch = agent.tree[agent.root_id]["child"]
    order[:len(ch)] = ch
    _, action = ref_utils.get_action(pi)
    st = np.random.get_state()
    return dict(pi=pi.copy(), visit=agent.visit.copy(), policy=agent.policy.copy(), w=cw, q=cq,
                order=order, action=int(action), mt_pos=int(st[2]),
                mt_sum=int(st[1].astype(np.uint64).sum()), tree_size=len(agent.tree))


def _play(B, S, mode, seed, plies, tau_thres=6, start=(0,), noise=True, model=None):
    agent = ref_agents.ZeroAgent(B, S, 5, noise=noise)
    agent.model = model if model is not None else StubModel(mode)
    np.random.seed(seed)
    root = tuple(start)
    recs = []
    win = 0
    t = 0
    while win == 0 and (plies == 0 or t < plies):
        tau = 1 if t < tau_thres else 0
        r = _search_record(agent, root, tau)
        r["root"] = np.array(root[1:], np.int32)
        recs.append(r)
        root = root + (r["action"],)
        win = ref_utils.check_win(ref_utils.get_board(root, B), 3 if B == 3 else 5)
        t += 1
    return recs, win


def _pack(cases):
    out = {}
    meta = []
    for ci, (cfg, recs, win) in enumerate(cases):
        meta.append(list(cfg) + [len(recs), win])
        for k in ("pi", "visit", "policy", "w", "q", "order"):
            out["c%d_%s" % (ci, k)] = np.stack([r[k] for r in recs])
        out["c%d_action" % ci] = np.array([r["action"] for r in recs], np.int32)
        out["c%d_mt_pos" % ci] = np.array([r["mt_pos"] for r in recs], np.int64)
        out["c%d_mt_sum" % ci] = np.array([r["mt_sum"] for r in recs], np.uint64)
        out["c%d_tree_size" % ci] = np.array([r["tree_size"] for r in recs], np.int64)
        L = max(len(r["root"]) for r in recs)
        roots = np.full((len(recs), max(L, 1)), -1, np.int32)
        for i, r in enumerate(recs):
            roots[i, :len(r["root"])] = r["root"]
        out["c%d_root" % ci] = roots
    # meta columns: board, sims, stub mode, seed, max plies, tau_thres, noise, n_records, win
    out["meta"] = np.array(meta, np.int64)
    return out


def gv5():
    cases = []

    def run(B, S, mode, seed, plies, tau_thres=6, noise=1, start=(0,)):
        recs, win = _play(B, S, mode, seed, plies, tau_thres, start, bool(noise))
        cases.append(((B, S, mode, seed, plies, tau_thres, noise), recs, win))
        print("  gv5 case", (B, S, mode, seed, plies), "->", len(recs), "plies, win", win)
        return recs

    # BASELINE shape: 9x9, 400 sims, several plies, all three stub flavours
    for mode, seed in ((0, 0), (1, 1), (2, 2)):
        run(9, 400, mode, seed, 5)
    # full 9x9 games at 60 sims (reach draws / wins, deep endgames with set-order quirks)
    for mode, seed in ((0, 3), (1, 4), (2, 5)):
        run(9, 60, mode, seed, 0)
    run(9, 60, 1, 6, 0, tau_thres=0)
    run(9, 48, 0, 7, 8, noise=0)
    # 15x15 at 80 sims
    run(15, 80, 2, 8, 5)
    run(15, 80, 1, 9, 4)
    # 3x3 full games (terminal + draw leaves, non-ascending child order, Marsaglia gamma)
    for seed in (10, 11, 12, 13):
        run(3, 50, seed % 3, seed, 0)
    run(3, 400, 0, 14, 0)
    out = _pack(cases)
    save("gv5_tree_stub", **out)

    # searches from deep 9x9 roots (>= 63 stones: non-ascending child order at the root).
    rng = np.random.RandomState(55)
    cases = []
    tries = 0
    while len(cases) < 4 and tries < 2000:
        tries += 1
        mv = rng.permutation(81)[:66].tolist()
        nid = (0,) + tuple(mv)
        if ref_utils.check_win(ref_utils.get_board(nid, 9), 5) != 0:
            continue
        recs, win = _play(9, 120, len(cases) % 3, 100 + len(cases), 2, start=nid)
        cases.append(((9, 120, len(cases) % 3, 100 + len(cases), 2, 6, 1), recs, win))
    out = _pack(cases)
    save("gv5_tree_stub_deeproot", **out)

    # searches from deep 15x15 roots (>= 149 stones: the 128-slot table case of CPython's set difference).
    # A random 152-stone position on 15x15 almost surely holds a five, so the stones are drawn from a five-free
    # colouring of the whole board (runs of two along rows, alternating along columns; checked with the reference's
    # check_win): any subset of a five-free position is five-free.
    B15 = 15
    full = np.zeros((B15, B15), np.int64)
    for r in range(B15):
        for c in range(B15):
            full[r, c] = 1 if ((c // 2) + r) % 2 == 0 else -1
    assert ref_utils.check_win(full.astype(float), 5) in (0, 3), "the colouring holds a five"
    black = [r * B15 + c for r in range(B15) for c in range(B15) if full[r, c] == 1]
    white = [r * B15 + c for r in range(B15) for c in range(B15) if full[r, c] == -1]
    rng = np.random.RandomState(77)
    cases = []
    for k, n_stones in enumerate((150, 152, 160)):
        bsel = rng.permutation(black)[:n_stones // 2].tolist()
        wsel = rng.permutation(white)[:n_stones // 2].tolist()
        mv = [x for pair in zip(bsel, wsel) for x in pair]       # black moves first (utils.get_board: even index = +1)
        nid = (0,) + tuple(mv)
        assert ref_utils.check_win(ref_utils.get_board(nid, B15), 5) == 0
        legal = ref_utils.legal_actions(nid, B15)
        assert legal != sorted(legal), "expected a non-ascending child order at %d stones" % n_stones
        recs, win = _play(B15, 40, k % 3, 200 + k, 2, start=nid)
        cases.append(((B15, 40, k % 3, 200 + k, 2, 6, 1), recs, win))
        print("  gv5 deep 15x15 root,", n_stones, "stones ->", len(recs), "plies, win", win)
    out = _pack(cases)
    save("gv5_tree_stub_deeproot15", **out)


def gv6():
    torch.manual_seed(0)
    net = ref_model.PVNet(4, 5, 128, 9)
    net.eval()
    # persist the weights as a plain npz so the tests do not depend on torch's init stream
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    rec = RecordingModel(net)
    recs, win = _play(9, 400, 0, 0, 2, model=rec)
    out = _pack([((9, 400, -1, 0, 2, 6, 1), recs, win)])
    out["eval_p"] = np.stack(rec.p).astype(np.float32)
    out["eval_v"] = np.array(rec.v, np.float32)
    save("gv6_tree_realnet", **out)
    # 4-block net weights are 4.8 MB; keep only a checksum + the generator (torch seed 0) out of
    # the repo. Tests replay eval_p/eval_v, so the weights themselves are not needed.
    save("gv6_net_digest", keys=np.array(sorted(sd.keys())),
         sums=np.array([float(np.abs(sd[k].astype(np.float64)).sum()) for k in sorted(sd)]))


def gv7():
    out = {}
    idx = 0
    rng = np.random.RandomState(77)
    for (nb, B, planes, wseed) in ((1, 9, 32, 1), (2, 9, 128, 2), (4, 9, 128, 3), (2, 15, 64, 4),
                                   (1, 3, 32, 5)):
        net = ref_model.PVNet(nb, 5, planes, B)
        sd = pvnet_weights.make_state_dict(nb, 5, planes, B, wseed)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        A = B * B
        xs = []
        for t in range(6):
            k = [0, 1, 2, 7, A // 2, A - 2][t]
            mv = rng.permutation(A)[:k].tolist()
            xs.append(ref_utils.get_state_pt((0,) + tuple(mv), B, 5))
        x = torch.tensor(np.stack(xs)).float()
        with torch.no_grad():
            p, v = net(x)
        out["cfg%d" % idx] = np.array([nb, B, planes, wseed], np.int32)
        out["x%d" % idx] = x.numpy()
        out["p%d" % idx] = p.numpy()
        out["v%d" % idx] = v.numpy()
        idx += 1
    out["count"] = np.array(idx)
    save("gv7_pvnet_forward", **out)


def gv8():
    rng = np.random.RandomState(88)
    out = {}
    for i, B in enumerate((3, 9)):
        s = rng.rand(5, B, B)
        pi = rng.rand(B * B)
        aug = ref_utils.augment_dataset([(s, pi, 1.0)], B)
        out["s%d" % i] = s
        out["pi%d" % i] = pi
        out["as%d" % i] = np.stack([a[0] for a in aug])
        out["api%d" % i] = np.stack([a[1] for a in aug])
    save("gv8_augment", **out)


def gv9():
    """main.self_play(1) with the stub evaluator: memory order, z assignment, augmentation."""
    cwd = os.getcwd()
    scratch = "/tmp/gen_golden_scratch"
    os.makedirs(os.path.join(scratch, "logs"), exist_ok=True)
    os.chdir(scratch)
    try:
        import main as ref_main
    finally:
        os.chdir(cwd)
    ref_main.PRINT_SELFPLAY = False
    ref_agents.PRINT_MCTS = False
    out = {}
    # (sims, stub, seed, episodes): the third case is main.py:136-142's loop over several episodes on ONE stream
    for ci, (S, mode, seed, n_ep) in enumerate(((30, 1, 5, 1), (24, 0, 9, 1), (20, 1, 31, 3))):
        ref_main.Agent = ref_agents.ZeroAgent(9, S, 5, noise=True)
        ref_main.Agent.model = StubModel(mode)
        ref_main.cur_memory.clear()
        ref_main.rep_memory.clear()
        for k in ref_main.result:
            ref_main.result[k] = 0
        np.random.seed(seed)
        ref_main.self_play(n_ep)
        cm = list(ref_main.cur_memory)
        out["c%d_cfg" % ci] = np.array([9, S, mode, seed], np.int32)
        out["c%d_episodes" % ci] = np.array(n_ep, np.int32)
        out["c%d_mt_pos" % ci] = np.array(np.random.get_state()[2], np.int64)
        out["c%d_state" % ci] = np.stack([m[0] for m in cm]).astype(np.float32)
        out["c%d_pi" % ci] = np.stack([m[1] for m in cm])
        out["c%d_z" % ci] = np.array([m[2] for m in cm])
        out["c%d_result" % ci] = np.array([ref_main.result[k] for k in ("Black", "White", "Draw")])
        out["c%d_rep_len" % ci] = np.array(len(ref_main.rep_memory))
        rm = list(ref_main.rep_memory)
        out["c%d_rep_pi_head" % ci] = np.stack([m[1] for m in rm[:16]])
        out["c%d_rep_state_head" % ci] = np.stack([m[0] for m in rm[:16]]).astype(np.float32)
    out["ncases"] = np.array(3)
    save("gv9_self_play_memory", **out)


def gv10():
    """One training mini-batch of main.train's loss + Adam(lr 2e-4, eps 1e-6) (main.py:85,294-305)."""
    nb, B, planes = 2, 9, 32
    net = ref_model.PVNet(nb, 5, planes, B)
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, 10)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, weight_decay=0, eps=1e-6)
    rng = np.random.RandomState(1010)
    xs = []
    for t in range(32):
        k = rng.randint(0, 40)
        mv = rng.permutation(81)[:k].tolist()
        xs.append(ref_utils.get_state_pt((0,) + tuple(mv), B, 5))
    s = torch.tensor(np.stack(xs)).float()
    pi = torch.tensor(rng.dirichlet(np.ones(81), size=32)).float()
    z = torch.tensor(rng.choice([-1.0, 0.0, 1.0], size=32)).float()
    p, v = net(s)
    v_loss = (v - z).pow(2).mean()
    p_loss = -(pi * p.log()).sum(dim=-1).mean()
    loss = v_loss + p_loss
    opt.zero_grad()
    loss.backward()
    grads = {k: p_.grad.numpy().copy() for k, p_ in net.named_parameters()}
    opt.step()
    after = {k: v_.detach().numpy().copy() for k, v_ in net.state_dict().items()}
    keys = ["conv1.weight", "layers.1.conv2.weight", "policy_head.policy_fc.bias",
            "value_head.value_fc2.weight", "bn1.running_mean", "layers.0.bn1.running_var"]
    out = dict(cfg=np.array([nb, B, planes, 10], np.int32), s=s.numpy(), pi=pi.numpy(),
               z=z.numpy(), v_loss=np.array(v_loss.item()), p_loss=np.array(p_loss.item()))
    for k in keys:
        out["after__" + k] = after[k]
        if k in grads:
            out["grad__" + k] = grads[k]
    save("gv10_train_step", **out)


def gv11():
    """Rollout agents: PUCTAgent / UCTAgent.get_pi (agents.py:263-614) under np.random.seed, several
    consecutive calls per agent object (each call is a fresh search; stale dict entries must not matter)."""
    import contextlib
    import io
    cases = [  # (mode, B, num_mcts, seed, start moves, plies)
        (0, 3, 200, 1, (), 4), (1, 3, 200, 2, (), 4), (0, 3, 60, 3, (4, 0, 8), 3), (1, 3, 60, 4, (4, 0, 8, 2), 3),
        (0, 9, 60, 5, (), 3), (1, 9, 60, 6, (), 3), (0, 9, 40, 7, (40, 41, 31, 32, 22, 23, 13), 3),
        (1, 9, 40, 8, (40, 41, 31, 32, 22, 23, 13), 3), (0, 15, 16, 9, (112, 113), 2), (1, 15, 16, 10, (112, 113), 2),
    ]
    # a crowded, still undecided 9x9 position (short playouts, many terminal leaves, draws possible)
    for sd in range(1000):
        dense = tuple(np.random.RandomState(sd).permutation(81)[:66].tolist())
        if ref_utils.check_win(ref_utils.get_board((0,) + dense, 9), 5) == 0:
            break
    cases += [(0, 9, 30, 11, dense, 3), (1, 9, 30, 12, dense, 3)]
    out = dict(ncases=np.array(len(cases)))
    for ci, (mode, B, S, seed, start, plies) in enumerate(cases):
        agent = (ref_agents.PUCTAgent if mode == 0 else ref_agents.UCTAgent)(B, S)
        np.random.seed(seed)
        root = (0,) + tuple(int(a) for a in start)
        recs = dict(pi=[], stat=[], action=[], pos=[])
        nrec = 0
        for t in range(plies):
            board = ref_utils.get_board(root, B)
            if ref_utils.check_win(board, agent.win_mark) != 0:
                break
            with contextlib.redirect_stdout(io.StringIO()):
                pi = agent.get_pi(root, board, ref_utils.get_turn(root), 0)
            stat = np.zeros(B * B) if mode == 0 else np.full(B * B, -np.inf)
            for a in agent.tree[root]['child']:
                stat[a] = agent.tree[root + (a,)]['n' if mode == 0 else 'q']
            recs["pi"].append(pi.copy())
            recs["stat"].append(stat)
            recs["action"].append(int(np.argmax(pi)))
            recs["pos"].append(int(np.random.get_state()[2]))
            root = root + (int(np.argmax(pi)),)
            nrec += 1
        out["c%d_cfg" % ci] = np.array([mode, B, S, seed, nrec], np.int64)
        out["c%d_start" % ci] = np.array(start, np.int64)
        for k, v in recs.items():
            out["c%d_%s" % (ci, k)] = np.array(v)
    save("gv11_rollout_agents", **out)


def gv12():
    """BASELINE configs[0]: the UCT search of 1_tictactoe_MCTS/mcts_vs.py (its __main__ loop, lines 153-183)
    driven through the reference's MCTS methods under random.seed; records q / n of the root children,
    max_action and the final `random` state."""
    import importlib.util
    import random
    ttt = "/root/reference/1_tictactoe_MCTS"
    saved = list(sys.path)
    sys.path.insert(0, ttt)
    for m in ("utils", "env"):
        sys.modules.pop(m, None)
    spec = importlib.util.spec_from_file_location("ref_mcts_vs", os.path.join(ttt, "mcts_vs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # __main__ guard: only the MCTS class is defined
    sys.path[:] = saved
    for m in ("utils", "env"):
        sys.modules.pop(m, None)
    agent = mod.MCTS(3)
    boards = [np.zeros((3, 3)),
              np.array([[1., 0, 0], [0, -1, 0], [0, 0, 0]]),
              np.array([[1., -1, 1], [0, -1, 0], [0, 1, 0]]),
              np.array([[1., -1, 0], [0, 1, 0], [0, 0, -1]]),
              np.array([[1., -1, 1], [-1, 1, 1], [0, 0, -1]]) * 1.0,
              np.array([[0., 0, 0], [0, 1, 0], [0, 0, 0]])]
    cases = []
    for bi, gb in enumerate(boards):
        turn = 0 if np.count_nonzero(gb) % 2 == 0 else 1
        for num_mcts, seed in ((200, 10 + bi), (1500, 20 + bi), (37, 30 + bi)):
            random.seed(seed)
            tree = {(0,): {'state': gb.copy(), 'player': turn, 'child': [], 'parent': None, 'n': 0, 'w': None, 'q': None}}
            for _ in range(num_mcts):
                leaf_id = agent.selection(tree)
                tree, child_id = agent.expansion(tree, leaf_id)
                sim_result = agent.simulation(tree, child_id)
                tree = agent.backup(tree, child_id, sim_result)
            q = np.full(9, -np.inf)
            n = np.zeros(9)
            q_list = {}
            for i in tree[(0,)]['child']:
                q[i] = tree[(0, i)]['q']
                n[i] = tree[(0, i)]['n']
                q_list[(0, i)] = tree[(0, i)]['q']
            max_action = max(q_list, key=q_list.get)[1]
            st = random.getstate()
            cases.append(dict(board=gb.astype(np.int8), cfg=np.array([turn, num_mcts, seed, max_action, st[1][624]], np.int64),
                              q=q, n=n, mt=np.array(st[1][:624], np.uint32)))
    out = dict(ncases=np.array(len(cases)))
    for ci, c in enumerate(cases):
        for k, v in c.items():
            out["c%d_%s" % (ci, k)] = v
    save("gv12_tictactoe_uct", **out)


def gv13():
    """eval_main.Evaluator.get_action call pattern (eval_main.py:137-151, 229-333): two ZeroAgents (noise off,
    tau 0) alternate on N matches with the colours swapped every match; after each move the loop forms root_id
    from the MOVER's root_id, steps the reference env, calls the opponent's del_parents and, at the end of a
    match, Evaluator.reset() and elo(). The reference's own Evaluator / GameState / elo objects are driven by
    this loop (main() itself cannot run: it loads checkpoints that do not ship)."""
    import contextlib
    import io
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import eval_main as ref_eval
    finally:
        os.chdir(cwd)
    out = {}
    cases = ((9, 36, 30, 1, 0, 21, 3), (9, 50, 24, 0, 1, 5, 2))   # board, S_player, S_enemy, stub_p, stub_e, seed, matches
    for ci, (B, SP, SE, mp, me, seed, n_match) in enumerate(cases):
        ev = ref_eval.Evaluator()
        ev.player = ref_agents.ZeroAgent(B, SP, 5, noise=False)
        ev.player.model = StubModel(mp)
        ev.enemy = ref_agents.ZeroAgent(B, SE, 5, noise=False)
        ev.enemy.model = StubModel(me)
        np.random.seed(seed)
        enemy_turn = 1
        player_elo, enemy_elo = 1500, 1500
        result = {'Player': 0, 'Enemy': 0, 'Draw': 0}
        moves_all, visit_all, pos_all, win_all, elo_all = [], [], [], [], []
        for i in range(n_match):
            env = ref_eval.game.GameState('text')
            board = np.zeros([B, B])
            root_id = (0,)
            win_index = 0
            turn = 0
            moves, visits, poss = [], [], []
            while win_index == 0:
                with contextlib.redirect_stdout(io.StringIO()):
                    action, action_index = ev.get_action(root_id, board, turn, enemy_turn)
                mover = ev.player if turn != enemy_turn else ev.enemy
                root_id = mover.root_id + (action_index,)
                visits.append(mover.get_visit().copy())
                with contextlib.redirect_stdout(io.StringIO()):
                    board, check_valid_pos, win_index, turn, _ = env.step(action)
                    (ev.enemy if turn == enemy_turn else ev.player).del_parents(root_id)
                moves.append(int(action_index))
                poss.append(int(np.random.get_state()[2]))
            if win_index == 3:
                result['Draw'] += 1
                player_elo, enemy_elo = ref_eval.elo(player_elo, enemy_elo, 0.5, 0.5)
            elif turn == enemy_turn:           # the player made the last move (eval_main.py:288-299)
                result['Player'] += 1
                player_elo, enemy_elo = ref_eval.elo(player_elo, enemy_elo, 1, 0)
            else:
                result['Enemy'] += 1
                player_elo, enemy_elo = ref_eval.elo(player_elo, enemy_elo, 0, 1)
            enemy_turn = abs(enemy_turn - 1)
            ev.reset()
            mv = np.full(B * B, -1, np.int32)
            mv[:len(moves)] = moves
            moves_all.append(mv)
            vis = np.zeros((B * B, B * B))
            vis[:len(visits)] = np.stack(visits)
            visit_all.append(vis)
            ps = np.full(B * B, -1, np.int64)
            ps[:len(poss)] = poss
            pos_all.append(ps)
            win_all.append(win_index)
            elo_all.append((player_elo, enemy_elo))
        out["c%d_cfg" % ci] = np.array([B, SP, SE, mp, me, seed, n_match], np.int32)
        out["c%d_moves" % ci] = np.stack(moves_all)
        out["c%d_visit" % ci] = np.stack(visit_all).astype(np.float32)
        out["c%d_mt_pos" % ci] = np.stack(pos_all)
        out["c%d_win" % ci] = np.array(win_all, np.int32)
        out["c%d_elo" % ci] = np.array(elo_all, np.float64)
        out["c%d_result" % ci] = np.array([result[k] for k in ("Player", "Enemy", "Draw")], np.int32)
    # mixed matches (eval_main.py:137-151): the player is a rollout agent -> get_pi(root_id, board, turn, tau) and the
    # MONITOR ZeroAgent searches the same root right after it (its draws and evaluations sit in the shared stream);
    # the enemy is a ZeroAgent. One match per case: (player kind, S_player, S_enemy, S_monitor, stub_e, stub_m, seed, enemy_turn)
    mixed = (("puct", 40, 24, 16, 1, 0, 77, 1), ("uct", 30, 20, 12, 0, 1, 78, 0))
    for mi, (kind, SP, SE, SM, me, mm, seed, enemy_turn) in enumerate(mixed):
        B = 9
        ev = ref_eval.Evaluator()
        ev.player = (ref_agents.PUCTAgent if kind == "puct" else ref_agents.UCTAgent)(B, SP)
        ev.enemy = ref_agents.ZeroAgent(B, SE, 5, noise=False)
        ev.enemy.model = StubModel(me)
        ev.monitor = ref_agents.ZeroAgent(B, SM, 5, noise=False)
        ev.monitor.model = StubModel(mm)
        np.random.seed(seed)
        env = ref_eval.game.GameState('text')
        board = np.zeros([B, B])
        root_id = (0,)
        win_index = 0
        turn = 0
        moves, poss, mvis = [], [], []
        while win_index == 0:
            with contextlib.redirect_stdout(io.StringIO()):
                action, action_index = ev.get_action(root_id, board, turn, enemy_turn)
            mover = ev.player if turn != enemy_turn else ev.enemy
            if turn != enemy_turn:
                mvis.append(ev.monitor.get_visit().copy())
            root_id = mover.root_id + (action_index,)
            with contextlib.redirect_stdout(io.StringIO()):
                board, check_valid_pos, win_index, turn, _ = env.step(action)
                (ev.enemy if turn == enemy_turn else ev.player).del_parents(root_id)
            moves.append(int(action_index))
            poss.append(int(np.random.get_state()[2]))
        out["m%d_cfg" % mi] = np.array([B, 0 if kind == "puct" else 1, SP, SE, SM, me, mm, seed, enemy_turn], np.int32)
        out["m%d_moves" % mi] = np.array(moves, np.int32)
        out["m%d_mt_pos" % mi] = np.array(poss, np.int64)
        out["m%d_monitor_visit" % mi] = np.stack(mvis).astype(np.float32)
        out["m%d_win" % mi] = np.array(win_index, np.int32)
    out["nmixed"] = np.array(len(mixed))
    out["ncases"] = np.array(len(cases))
    save("gv13_eval_head_to_head", **out)


def gv14():
    """The reference's searches with the REAL network end to end (torch CPU, one thread): what `north_star`'s "visit counts and
    chosen moves bit-exact" has to be measured against once the evaluations come from the native MI355X forward instead of a
    replay (tests/test_gpu_realnet_drift.py). Visits / actions / stream positions only -- no recorded evaluations: 3 seeds x 6
    plies with the random-init 4-block network of gv6 (torch.manual_seed(0)), 2 seeds x 6 plies with the trained 2-block fixture, and one 15x15 game
    of 3 plies at 200 simulations with a random-init 10-block network (configs[4]'s shape)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_trained_fixture import load
    cases = []
    torch.manual_seed(0)
    net = ref_model.PVNet(4, 5, 128, 9)
    net.eval()
    for seed in (0, 1, 2):
        recs, win = _play(9, 400, 0, seed, 6, model=net)
        cases.append(((9, 400, -1, seed, 6, 6, 1), recs, win))
        print("  gv14 random-init seed", seed, [r["action"] for r in recs])
    tnet = ref_model.PVNet(2, 5, 128, 9)
    tnet.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in load(os.path.join(OUT, "trained_2block_9x9.npz")).items()})
    tnet.eval()
    for seed in (0, 1):
        recs, win = _play(9, 400, 0, seed, 6, model=tnet)
        cases.append(((9, 400, -2, seed, 6, 6, 1), recs, win))
        print("  gv14 trained seed", seed, [r["action"] for r in recs])
    # BASELINE configs[4]'s shape: 15x15, the reference's default 10 blocks (torch.manual_seed(1) init), 200 simulations, 3 plies
    torch.manual_seed(1)
    wnet = ref_model.PVNet(10, 5, 128, 15)
    wnet.eval()
    recs, win = _play(15, 200, 0, 5, 3, model=wnet)
    cases.append(((15, 200, -3, 5, 3, 6, 1), recs, win))
    print("  gv14 15x15 10-block seed 5", [r["action"] for r in recs])
    out = _pack(cases)
    for k in [k for k in out if k.split("_", 1)[-1] in ("w", "q", "policy", "pi", "order")]:
        del out[k]
    save("gv14_realnet_visits", **out)


ALL = dict(gv14=gv14, gv13=gv13, gv12=gv12, gv11=gv11, gv1=gv1, gv2=gv2, gv3=gv3, gv4=gv4, gv5=gv5, gv6=gv6, gv7=gv7, gv8=gv8, gv9=gv9,
           gv10=gv10)

if __name__ == "__main__":
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print("==", w)
        ALL[w]()
