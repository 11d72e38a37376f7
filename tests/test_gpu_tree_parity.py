"""-m gpu: the HIP tree kernels (through the C ABI) against the CPU oracle and against the golden
vectors captured from the reference. Bar: bit-exact visit counts, priors, w/q, chosen moves and
MT19937 stream position (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from conftest import load_golden
from gpu_helpers import HostEvalRunner, children_by_action

pytestmark = pytest.mark.gpu


def _engine(*a, **k):
    from alpha_omok_amd.engine import Engine
    return Engine(*a, **k)


def _compare_move(eng, g, ag, root, pi, vis, pol, tag):
    opi, ovis, opol = ag
    np.testing.assert_array_equal(vis[g], ovis, err_msg="visit " + tag)
    np.testing.assert_array_equal(pol[g], opol, err_msg="policy " + tag)
    np.testing.assert_array_equal(pi[g], opi, err_msg="pi " + tag)


@pytest.mark.parametrize("board,sims,mode,plies,tau_thres", [
    (9, 400, 0, 3, 6), (9, 400, 2, 3, 6), (9, 120, 1, 12, 4), (3, 50, 0, 9, 2), (3, 50, 1, 9, 0),
    (15, 80, 2, 4, 6), (15, 60, 1, 5, 2), (5, 40, 0, 25, 3)])
def test_multi_game_parity_with_oracle(oracle, board, sims, mode, plies, tau_thres):
    """G games advance in lock-step on the GPU; every game must reproduce the oracle's
    sequential search move by move (stub evaluator = exact arithmetic, so p/v are identical)."""
    G = 6
    eng = _engine(board, sims, 5, games=G, noise=True)
    run = HostEvalRunner(eng)
    seeds = [1000 + 17 * g for g in range(G)]
    eng.seed_all(seeds)
    agents = [oracle.Agent(board, sims, 5, noise=True, evaluator="stub%d" % mode) for _ in range(G)]
    for g in range(G):
        agents[g].seed(seeds[g])
    roots = [(0,) for _ in range(G)]
    alive = np.ones(G, np.uint8)
    wm = 3 if board == 3 else 5

    def ev(g, sim, planes):
        return oracle.stub_eval(planes, mode)

    for t in range(plies):
        if not alive.any():
            break
        tau = np.array([1 if t < tau_thres else 0] * G, np.int8)
        pi, vis, pol = run.move(ev, tau=tau, active=alive)
        act, win = eng.play()
        for g in range(G):
            if not alive[g]:
                continue
            tag = "board %d game %d ply %d" % (board, g, t)
            o = agents[g].get_pi(roots[g], int(tau[g]))
            _compare_move(eng, g, o, roots[g], pi, vis, pol, tag)
            oa = agents[g].rng.choice_p(o[0])
            assert act[g] == oa, tag
            roots[g] = roots[g] + (int(oa),)
            mt, pos, _, _ = eng.get_rng_state(g)
            assert pos == agents[g].rng.pos, tag
            np.testing.assert_array_equal(mt, agents[g].rng.state_words(), err_msg="mt " + tag)
            ow = oracle.check_win(oracle.get_board(list(roots[g])[1:], board), wm)
            assert win[g] == ow, tag
            if ow != 0:
                alive[g] = 0
    eng.close()


def _run_golden_cases(oracle, g, evaluator_for_case, only=None):
    meta = g["meta"].tolist()
    for ci, (B, S, mode, seed, plies, tau_thres, noise, nrec, win) in enumerate(meta):
        if only is not None and ci not in only:
            continue
        eng = _engine(B, S, 5, games=1, noise=bool(noise))
        run = HostEvalRunner(eng)
        eng.seed(0, seed)
        ev = evaluator_for_case(ci, mode)
        roots = g["c%d_root" % ci]
        for t in range(nrec):
            root = [int(x) for x in roots[t] if x >= 0]
            st = eng.set_root(0, root)
            assert st == (0 if t == 0 else 2)
            tau = 1 if t < tau_thres else 0
            pi, vis, pol = run.move(ev, tau=np.array([tau], np.int8))
            tag = "case %d ply %d" % (ci, t)
            np.testing.assert_array_equal(vis[0], g["c%d_visit" % ci][t], err_msg="visit " + tag)
            np.testing.assert_array_equal(pol[0], g["c%d_policy" % ci][t], err_msg="policy " + tag)
            np.testing.assert_array_equal(pi[0], g["c%d_pi" % ci][t], err_msg="pi " + tag)
            ch = children_by_action(eng.root_children(0), B * B)
            np.testing.assert_array_equal(ch["w"], g["c%d_w" % ci][t], err_msg="w " + tag)
            np.testing.assert_array_equal(ch["q"], g["c%d_q" % ci][t], err_msg="q " + tag)
            order = g["c%d_order" % ci][t]
            assert ch["order"].tolist() == order[order >= 0].tolist(), tag
            act, w = eng.play()
            assert act[0] == int(g["c%d_action" % ci][t]), tag
            mt, pos, _, _ = eng.get_rng_state(0)
            assert pos == int(g["c%d_mt_pos" % ci][t]), tag
            assert int(mt.astype(np.uint64).sum()) == int(g["c%d_mt_sum" % ci][t]), tag
        eng.close()


def test_golden_gv5_stub_tree(oracle):
    """Direct check against the reference's own outputs (tests/golden/gv5_tree_stub.npz)."""
    g = load_golden("gv5_tree_stub")
    _run_golden_cases(oracle, g, lambda ci, mode: (lambda gi, sim, pl: oracle.stub_eval(pl, mode)))


def test_golden_gv5_deep_roots(oracle):
    """Roots with >= 63 stones on 9x9: child order follows CPython's set iteration."""
    g = load_golden("gv5_tree_stub_deeproot")
    _run_golden_cases(oracle, g, lambda ci, mode: (lambda gi, sim, pl: oracle.stub_eval(pl, mode)))


def test_golden_gv5_deep_roots_15x15(oracle):
    """Roots with 150 / 152 / 160 stones on 15x15: the 128-slot-table case of the child order on the device."""
    g = load_golden("gv5_tree_stub_deeproot15")
    _run_golden_cases(oracle, g, lambda ci, mode: (lambda gi, sim, pl: oracle.stub_eval(pl, mode)))


def test_golden_gv6_real_net_replay():
    """Search driven by the reference PVNet's recorded (p, v): same visits and moves."""
    g = load_golden("gv6_tree_realnet")
    ep, ev = g["eval_p"], g["eval_v"]
    cursor = [0]

    def replay(gi, sim, pl):
        i = cursor[0]
        cursor[0] += 1
        return ep[i], ev[i]

    # The reference evaluates the net on terminal leaves too; none occur in the first plies of a
    # 9x9 game, so the recording is consumed one entry per simulation.
    _run_golden_cases(None, g, lambda ci, mode: replay)
    assert cursor[0] == len(ev)


def test_planes_match_oracle(oracle):
    """Leaf planes written by k_select == utils.get_state_pt of the leaf id."""
    import torch
    for board in (3, 9, 15):
        eng = _engine(board, 30, 5, games=4, noise=True)
        run = HostEvalRunner(eng)
        eng.seed_all([5, 6, 7, 8])
        seen = []

        def ev(g, sim, planes):
            seen.append(planes.copy())
            # planes must be a valid get_state_pt output: rebuild the id from the planes' stones
            return oracle.stub_eval(planes, 0)

        run.move(ev)
        # cross-check a handful against the oracle encoder through the stub's own hash: the
        # multi-game parity test already proves equality indirectly; here check structure.
        for pl in seen[:50]:
            assert set(np.unique(pl)).issubset({0.0, 1.0})
            assert pl[4].min() == pl[4].max()
            own, opp = pl[2], pl[3]
            assert (own * opp).sum() == 0
            assert np.all(pl[0] <= pl[2]) and np.all(pl[1] <= pl[3])
        eng.close()


def test_device_planes_match_reference_get_state_pt():
    """The device encoder against gv3 (the reference's utils.get_state_pt outputs) directly: a fresh root's first
    leaf is the root itself, so ao_set_roots + ao_begin_move + ao_collect_leaves on every gv3 id hands back the
    planes of that id -- both the NCHW batch an external evaluator sees and, through the native network's plane
    input, nothing else in between. Boards 3 / 9 / 15, C in {3, 5, 7}."""
    import torch
    from alpha_omok_amd import utils
    g = load_golden("gv3_state_planes")
    groups = {}
    for i in range(int(g["count"])):
        m = g["m%d" % i]
        B, C, nid = int(m[0]), int(m[1]), (0,) + tuple(int(x) for x in m[2:])
        if utils.check_win(utils.get_board(nid, B), 3 if B == 3 else 5) != 0:
            continue                      # a finished game has no leaf to evaluate
        groups.setdefault((B, C), []).append((nid, g["s%d" % i]))
    assert len(groups) == 9
    checked = 0
    for (B, C), items in sorted(groups.items()):
        G = len(items)
        eng = _engine(B, 4, C, games=G, noise=False)
        st = eng.set_roots([nid for nid, _ in items])
        assert (st == 0).all()            # nothing was known: fresh roots
        planes = torch.full((G, C, B, B), -7.0, dtype=torch.float32, device="cuda")
        eng.begin_move()
        eng.collect_leaves(planes.data_ptr())
        eng.sync()
        got = planes.cpu().numpy()
        for k, (nid, want) in enumerate(items):
            np.testing.assert_array_equal(got[k], want, err_msg="B=%d C=%d id=%r" % (B, C, nid))
            checked += 1
        eng.close()
    assert checked > 90


def test_set_roots_batched_equals_one_by_one(oracle):
    """ao_set_roots (one launch, every masked game) leaves the same trees as G calls of ao_set_root: statuses,
    kept subtrees and the next search are identical; an id that does not extend the kept root restarts that game
    only; an illegal id is reported and resets that game."""
    G, B, S = 6, 9, 24
    engs = [_engine(B, S, 5, games=G, noise=True) for _ in range(2)]
    for e in engs:
        e.seed_all(list(range(3, 3 + G)))
    runs = [HostEvalRunner(e) for e in engs]
    ev = lambda g, sim, planes: oracle.stub_eval(planes, 1)  # noqa: E731
    outs = [r.move(ev) for r in runs]
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    vis = outs[0][1]
    # every game jumps two plies: most-visited child, then (games 0-2) a visited grandchild / (3-4) the lowest
    # legal cell, which the search may never have reached; game 5 gets an unrelated id
    ids = []
    for g in range(G):
        a = int(np.argmax(vis[g]))
        b = next(c for c in range(B * B) if c != a)
        ids.append((0, a, b) if g < 5 else (0, 40, 41, 42))
    st_b = engs[0].set_roots(ids)
    st_1 = np.array([engs[1].set_root(g, list(ids[g])[1:]) for g in range(G)])
    np.testing.assert_array_equal(st_b, st_1)
    for g in range(G):
        assert engs[0].tree_nodes(g) == engs[1].tree_nodes(g)
        assert engs[0].get_moves(g) == list(ids[g])[1:]
    outs = [r.move(ev) for r in runs]
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    # masked call: only game 2 moves
    mask = np.zeros(G, np.uint8)
    mask[2] = 1
    more = [tuple(i) + (80,) for i in ids]
    st = engs[0].set_roots(more, mask)
    assert st[2] >= 0 and (np.delete(st, 2) == -2).all()
    assert engs[0].get_moves(2) == list(more[2])[1:] and engs[0].get_moves(1) == list(ids[1])[1:]
    # an id that does not extend the kept one restarts that game only
    mask[:] = 0
    mask[1] = 1
    other = list(more)
    other[1] = (0, (ids[1][1] + 1) % (B * B))
    st = engs[0].set_roots(other, mask)
    assert st[1] == 0 and engs[0].get_moves(1) == [other[1][1]] and engs[0].tree_nodes(1)[0] == 0
    assert engs[0].get_moves(0) == list(ids[0])[1:]
    mask[:] = 0
    mask[2] = 1
    # an occupied cell in the id: error, that game restarts
    bad = list(more)
    bad[2] = more[2] + (80,)
    with pytest.raises(Exception):
        engs[0].set_roots(bad, mask)
    assert engs[0].get_moves(2) == []
    for e in engs:
        e.close()


def test_set_root_semantics(oracle):
    """ZeroAgent.get_pi with ids that jump two plies (eval_main's usage): known/unknown roots."""
    B, S = 9, 60
    eng = _engine(B, S, 5, games=1, noise=True)
    run = HostEvalRunner(eng)
    ag = oracle.Agent(B, S, 5, noise=True, evaluator="stub1")
    eng.seed(0, 77)
    ag.seed(77)

    def ev(g, sim, pl):
        return oracle.stub_eval(pl, 1)

    root = (0,)
    rs = np.random.RandomState(3)
    for t in range(8):
        st = eng.set_root(0, list(root)[1:])
        pi, vis, pol = run.move(ev, tau=np.array([1], np.int8))
        opi, ovis, opol = ag.get_pi(root, 1)
        np.testing.assert_array_equal(vis[0], ovis, err_msg="ply %d status %d" % (t, st))
        np.testing.assert_array_equal(pol[0], opol)
        # our move: most visited; opponent's reply: a random legal cell (often unvisited)
        a = int(np.argmax(ovis))
        legal = [c for c in range(B * B) if c not in root[1:] and c != a]
        b = int(legal[rs.randint(len(legal))])
        root = root + (a, b)
        # keep both RNG streams aligned (the engine's stream is per game)
        mt, pos, hg, gs = eng.get_rng_state(0)
        assert pos == ag.rng.pos
    eng.close()


# a full-board draw on 9x9: 81 moves, colours alternating, never five in a row (found by a randomised greedy search)
_DRAW_ORDER = [9, 43, 55, 14, 78, 32, 11, 23, 30, 3, 29, 47, 10, 17, 65, 20, 64, 40, 44, 56, 2, 80, 50, 25, 79, 68, 37, 76, 77, 63, 69,
               34, 46, 49, 4, 74, 62, 61, 31, 13, 53, 75, 18, 59, 72, 42, 24, 28, 12, 35, 7, 5, 60, 73, 27, 15, 22, 45, 33, 6, 66, 0, 39,
               1, 67, 21, 58, 48, 54, 19, 71, 57, 36, 51, 16, 70, 26, 38, 41, 8, 52]


@pytest.mark.parametrize("B,S,plies", [(9, 200, 3), (15, 320, 3)])
def test_descents_longer_than_64_levels(oracle, B, S, plies):
    """A policy that puts 97 % on the next move of one fixed line and a value of 0 makes every simulation extend ONE chain; with
    win_mark above the board size (ZeroAgent.win_mark is a plain attribute in the reference; the oracle with such a mark was
    checked against the reference itself) no line wins, so the chain runs until the board is full: descents of 81 levels on
    9x9, 225 on 15x15, ending at a drawn terminal leaf. select_game keeps the path in registers -- entry d in lane d & 63 of
    chunk d >> 6 -- and expand_backup_game walks it chunk by chunk: these are the cases with two and four chunks. Several moves
    (the later ones search the inherited chain down to its end), visits / priors / pi / action / MT19937 state against the
    oracle, bit for bit."""
    G = 3
    A = B * B
    eng = _engine(B, S, 5, games=G, noise=True, win_mark=B + 1)
    run = HostEvalRunner(eng)
    order = _DRAW_ORDER if B == 9 else np.random.RandomState(1).permutation(A).tolist()
    rank = {c: i for i, c in enumerate(order)}

    def chain_eval(planes):
        occupied = (planes[:4].sum(axis=0) > 0).reshape(-1)   # planes 0..3: the two players' stones now and a move ago
        empty = [c for c in range(A) if not occupied[c]]
        p = np.full(A, np.float32(0.03 / (A - 1)), np.float32)
        if empty:
            p[min(empty, key=lambda c: rank[c])] = np.float32(0.97)
        return p, np.float32(0.0)

    agents = [oracle.Agent(B, S, 5, noise=True, evaluator=lambda mv, pl, sim: chain_eval(pl)) for _ in range(G)]
    for g in range(G):
        eng.seed(g, 500 + g)
        agents[g].seed(500 + g)
        agents[g].set_win_mark(B + 1)
    roots = [(0,)] * G
    deep = 0
    for ply in range(plies):
        pi, vis, pol = run.move(lambda g, sim, pl: chain_eval(pl), tau=np.zeros(G, np.int8))
        st = eng.search_stats()
        deep = max(deep, st["levels"] / max(st["evaluated"] + st["terminal"], 1))
        if ply > 0 and B == 9:
            assert st["terminal"] > 0
        act, win = eng.play()
        for g in range(G):
            opi, ovis, opol = agents[g].get_pi(roots[g], 0)
            np.testing.assert_array_equal(vis[g], ovis, err_msg="ply %d game %d" % (ply, g))
            np.testing.assert_array_equal(pol[g], opol)
            np.testing.assert_array_equal(pi[g], opi)
            oa = agents[g].rng.choice_p(opi)   # utils.get_action on the game's stream (eng.play did the same)
            assert act[g] == oa
            mt, pos, _, _ = eng.get_rng_state(g)
            assert pos == agents[g].rng.pos
            np.testing.assert_array_equal(mt, agents[g].rng.state_words())
            roots[g] = roots[g] + (int(oa),)
    # (9x9: terminal > 0 from the second move on, asserted above -- with no winning line the only terminal leaf is the full board,
    # so descents of 81 - ply levels did happen. The mean over all simulations of a move; 15x15: chains of 100 - 200 levels)
    assert deep > (30 if B == 9 else 100), deep
    eng.close()


@pytest.mark.parametrize("B,S,G,blocks", [(9, 400, 4096, 4), (15, 800, 1024, 10)])
def test_full_size_properties_and_sampled_oracle_parity(oracle, B, S, G, blocks):
    """BASELINE configs[2] size (9x9, 4096 games x 400 sims, 4-block net, group-resident trunk) and the per-GPU
    shape of configs[4] (15x15, 800 sims, 10-block net, all 1024 games of one GPU), FOUR plies with re-rooting
    (eng.play) between them -- tau = 1, 1, 0, 0 (arg-max + tie-break stream) -- and a refill before the fourth (a
    handful of slots reset and re-seeded, so fresh roots with S + 1 simulations search beside inherited ones):
    size-independent properties for every game at every ply + bit-exact oracle replay of the sampled games
    (visits, post-noise priors, pi, chosen action, MT19937 position and state) at every ply."""
    import torch
    import pvnet_weights
    from alpha_omok_amd.engine import Engine, Net
    from alpha_omok_amd.pvnet import PVNet
    if blocks == 4:
        net = Net(4, 5, 128, B, 0)
        net.load_state_dict(pvnet_weights.make_state_dict(4, 5, 128, B, 77))
    else:  # PyTorch default init (the deterministic generator saturates a 10-block stack)
        torch.manual_seed(7)
        model = PVNet(blocks, 5, 128, B)
        model.eval()
        net = model.to_native(0)
    A = B * B
    eng = Engine(B, S, 5, games=G, noise=True)
    seeds = np.arange(9000, 9000 + G, dtype=np.uint32)
    eng.seed_all(seeds)
    sample = [0, 1, 17, G // 2, G - 1]
    refill = [1, 5, 6, G - 2]                                # game 1 is sampled: its oracle restarts too
    planes = torch.zeros((G, 5, B, B), dtype=torch.float32, device="cuda")
    cursors = {g: [0] for g in sample}
    recs = {g: [] for g in sample}

    def make_agent(g, seed):
        def replay(moves, pl, sim, g=g):
            i = cursors[g][0]
            cursors[g][0] += 1
            return recs[g][i]
        ag = oracle.Agent(B, S, 5, noise=True, evaluator=replay)
        ag.seed(int(seed))
        return ag

    agents = {g: make_agent(g, seeds[g]) for g in sample}
    roots = {g: (0,) for g in sample}
    ply = np.zeros(G, np.int64)
    inherited = np.zeros(G)                                  # visits the root's children hold before the search
    first_vis = None
    for t in range(4):
        if t == 3:
            mask = np.zeros(G, np.uint8)
            mask[refill] = 1
            eng.reset(mask)
            for k, g in enumerate(refill):
                eng.seed(g, 50000 + k)
            ply[refill] = 0
            inherited[refill] = 0
            agents[1] = make_agent(1, 50000)
            roots[1] = (0,)
        tau = (ply < 2).astype(np.int8)
        fresh = ply == 0
        for g in sample:
            recs[g] = []
            cursors[g][0] = 0
        eng.begin_move()
        assert eng.sims_left() == (S + 1 if fresh.any() else S)
        while eng.sims_left() > 0:
            eng.collect_leaves(planes.data_ptr())
            eng.sync()
            p, v = net(planes)                       # ao_net_forward: same kernels as ao_search
            torch.cuda.synchronize()
            hp, hv = p[sample].cpu().numpy(), v[sample].cpu().numpy()
            for i, g in enumerate(sample):
                recs[g].append((hp[i].copy(), hv[i].copy()))
            eng.apply_evals(p.data_ptr(), v.data_ptr())
        pi, vis, pol = eng.end_move(tau)
        if t == 0:
            first_vis, first_pol = vis.copy(), pol.copy()
        # properties that hold for every game regardless of size
        np.testing.assert_array_equal(vis.sum(axis=1), inherited + S)   # fresh root: S+1 sims, S child visits; else on top of the inherited ones
        assert np.all(vis == np.round(vis)) and np.all(vis >= 0)
        assert np.abs(pi.sum(axis=1) - 1).max() < 1e-12
        assert np.abs(pol.sum(axis=1) - 1).max() < 1e-9          # renormalised priors mixed with Dirichlet noise
        npos = (pol > 0).sum(axis=1)                             # the legal cells carry a prior (a Dirichlet component may underflow)
        assert np.all(npos <= A - ply) and np.all(npos >= A - ply - 1)
        onehot = tau == 0
        assert np.all(pi[onehot].max(axis=1) == 1.0) and np.all((pi[onehot] > 0).sum(axis=1) == 1)
        assert np.all(vis[onehot, pi[onehot].argmax(axis=1)] == vis[onehot].max(axis=1))   # the arg-max is a maximum
        st = eng.search_stats()
        assert st["evaluated"] + st["terminal"] == int((S + fresh).sum()) and st["terminal"] == 0
        assert 1.0 < st["levels"] / st["evaluated"] < 4.0
        act, win = eng.play()
        assert np.all(win == 0) and np.all((act >= 0) & (act < A))
        assert np.all(vis[np.arange(G), act] > 0)                # the chosen move was visited
        assert np.all(act[onehot] == pi[onehot].argmax(axis=1))
        # sampled games against the oracle, bit for bit
        for g in sample:
            tag = "game %d ply %d (search %d)" % (g, ply[g], t)
            opi, ovis, opol = agents[g].get_pi(roots[g], int(tau[g]))
            assert cursors[g][0] == S + (1 if ply[g] == 0 else 0), tag
            np.testing.assert_array_equal(vis[g], ovis, err_msg=tag)
            np.testing.assert_array_equal(pol[g], opol, err_msg=tag)
            np.testing.assert_array_equal(pi[g], opi, err_msg=tag)
            oa = agents[g].rng.choice_p(opi)
            assert act[g] == oa, tag
            roots[g] = roots[g] + (int(oa),)
            mt, pos, _, _ = eng.get_rng_state(g)
            assert pos == agents[g].rng.pos, tag
            np.testing.assert_array_equal(mt, agents[g].rng.state_words(), err_msg="mt " + tag)
        inherited = vis[np.arange(G), act] - 1                   # the chosen child's own first visit expanded it
        ply += 1
    assert eng.trim_stats() == (0, 0)                            # nothing was forgotten at re-rooting
    # determinism: a second engine with the same seeds and the fused path gives the same first move
    eng2 = Engine(B, S, 5, games=G, noise=True)
    eng2.seed_all(seeds)
    pi2, vis2, pol2 = eng2.search(net, tau=np.ones(G, np.int8))
    np.testing.assert_array_equal(vis2, first_vis)
    np.testing.assert_array_equal(pol2, first_pol)
    eng.close()
    eng2.close()
    net.close()


def test_trained_network_deep_trees_and_sampled_oracle_parity(oracle):
    """The regime a TRAINED network puts the tree kernels in, at G = 1024: sharp priors, so the PUCT descents go 6 - 25
    levels deep, meet terminal leaves inside the batch, and the played child keeps most of the root's visits (inherited
    counts of several times S; the arena default must hold them without a trim). Network: tests/golden/trained_2block_9x9.npz
    -- trained by THIS engine on the MI355X (tools/train_omok.py, tools/exp/r4a.sh: 113 k self-play games, 64 : 0 against
    its iteration 0; tools/make_trained_fixture.py made the fixture). Fourteen plies with re-rooting, tau = 1 for six plies
    then 0 (main.py:150-153); five sampled games are replayed through the oracle at every ply with the recorded (p, v):
    visits, post-noise priors, pi, action, MT19937 position and state bit for bit; games that end stay ended."""
    import sys
    import torch
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_trained_fixture import load
    from alpha_omok_amd.engine import Engine, Net
    B, S, G, PLIES = 9, 400, 1024, 14
    A = B * B
    sd = load(os.path.join(REPO, "tests", "golden", "trained_2block_9x9.npz"))
    net = Net(2, 5, 128, B, 0)
    net.load_state_dict(sd)
    eng = Engine(B, S, 5, games=G, noise=True)
    seeds = np.arange(31000, 31000 + G, dtype=np.uint32)
    eng.seed_all(seeds)
    sample = [0, 1, 17, G // 2, G - 1]
    planes = torch.zeros((G, 5, B, B), dtype=torch.float32, device="cuda")
    cursors = {g: [0] for g in sample}
    recs = {g: [] for g in sample}

    def make_agent(g, seed):
        def replay(moves, pl, sim, g=g):
            i = cursors[g][0]
            cursors[g][0] += 1
            return recs[g][i]
        ag = oracle.Agent(B, S, 5, noise=True, evaluator=replay)
        ag.seed(int(seed))
        return ag

    agents = {g: make_agent(g, seeds[g]) for g in sample}
    roots = {g: (0,) for g in sample}
    alive = np.ones(G, np.uint8)
    inherited = np.zeros(G)
    depth, terminal_total, max_inherited, checked = [], 0, 0.0, 0
    for t in range(PLIES):
        tau = np.full(G, 1 if t < 6 else 0, np.int8)
        for g in sample:
            recs[g] = []
            cursors[g][0] = 0
        eng.begin_move(alive)
        while eng.sims_left() > 0:
            eng.collect_leaves(planes.data_ptr())
            eng.sync()
            p, v = net(planes)                       # ao_net_forward: the kernels ao_search uses
            torch.cuda.synchronize()
            hp, hv = p[sample].cpu().numpy(), v[sample].cpu().numpy()
            for i, g in enumerate(sample):
                recs[g].append((hp[i].copy(), hv[i].copy()))
            eng.apply_evals(p.data_ptr(), v.data_ptr())
        pi, vis, pol = eng.end_move(tau)
        on = alive != 0
        st = eng.search_stats()
        n_sims = int(on.sum()) * S + (int(on.sum()) if t == 0 else 0)
        assert st["evaluated"] + st["terminal"] == n_sims
        depth.append(st["levels"] / n_sims)
        terminal_total += st["terminal"]
        max_inherited = max(max_inherited, float(inherited[on].max()))
        np.testing.assert_array_equal(vis[on].sum(axis=1), inherited[on] + S)
        assert np.abs(pi[on].sum(axis=1) - 1).max() < 1e-12 and np.abs(pol[on].sum(axis=1) - 1).max() < 1e-9
        act, win = eng.play()
        assert np.all(vis[on, act[on]] > 0)
        for g in sample:
            if not alive[g]:
                continue
            tag = "game %d ply %d" % (g, t)
            opi, ovis, opol = agents[g].get_pi(roots[g], int(tau[g]))
            np.testing.assert_array_equal(vis[g], ovis, err_msg=tag)
            np.testing.assert_array_equal(pol[g], opol, err_msg=tag)
            np.testing.assert_array_equal(pi[g], opi, err_msg=tag)
            oa = agents[g].rng.choice_p(opi)
            assert act[g] == oa, tag
            roots[g] = roots[g] + (int(oa),)
            mt, pos, _, _ = eng.get_rng_state(g)
            assert pos == agents[g].rng.pos, tag
            np.testing.assert_array_equal(mt, agents[g].rng.state_words(), err_msg="mt " + tag)
            assert win[g] == oracle.check_win(oracle.get_board(list(roots[g][1:]), B), 5), tag
            checked += 1
        inherited = np.where(on, vis[np.arange(G), act] - 1, 0.0)
        alive[on & (win != 0)] = 0
        if not alive.any():
            break
    # the regime was the one the test is for
    assert max(depth) >= 6.0, depth
    assert terminal_total > 0
    assert max_inherited >= 2 * S, max_inherited
    assert checked >= 3 * 10, checked                         # >= 3 sampled games alive through >= 10 plies
    assert eng.trim_stats() == (0, 0) and eng.node_cap()[0] >= 8 * (S + 1)
    eng.close()
    net.close()
