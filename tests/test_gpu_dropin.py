"""-m gpu: the reference-shaped Python surface (agents.ZeroAgent, main.self_play) on the HIP engine
against golden vectors captured from the reference and against the oracle."""
import os

import numpy as np
import pytest

import pvnet_weights
from conftest import load_golden

pytestmark = pytest.mark.gpu


class StubModel:
    """Agent.model stand-in (any callable works): the oracle's exact-arithmetic stub, batch-capable."""

    def __init__(self, oracle, mode):
        self.oracle, self.mode = oracle, mode

    def eval(self):
        return self

    def __call__(self, x):
        import torch
        xs = x.detach().cpu().numpy().astype(np.float32)
        ps, vs = zip(*(self.oracle.stub_eval(xs[i], self.mode) for i in range(xs.shape[0])))
        return torch.from_numpy(np.stack(ps)), torch.from_numpy(np.array(vs, np.float32))


def test_zero_agent_matches_reference_under_numpy_seed(oracle):
    """np.random.seed(s); ZeroAgent.get_pi; utils.get_action -- the reference's own call sequence
    (main.py:155-171) -- reproduces the reference's pi / visit / policy / move and leaves the
    process-global numpy stream where the reference leaves it (gv5 captured from the reference)."""
    from alpha_omok_amd import agents, utils
    agents.PRINT_MCTS = False
    g = load_golden("gv5_tree_stub")
    for ci in (0, 4, 10, 13):
        B, S, mode, seed, plies, tau_thres, noise, nrec, win = g["meta"][ci].tolist()
        agent = agents.ZeroAgent(B, S, 5, noise=bool(noise))
        agent.model = StubModel(oracle, mode)
        np.random.seed(seed)
        root_id = (0,)
        for t in range(nrec):
            pi = agent.get_pi(root_id, 1 if t < tau_thres else 0)
            np.testing.assert_array_equal(pi, g["c%d_pi" % ci][t])
            np.testing.assert_array_equal(agent.get_visit(), g["c%d_visit" % ci][t])
            np.testing.assert_array_equal(agent.get_policy(), g["c%d_policy" % ci][t])
            assert agent.is_real_root == (t == 0)
            _, a = utils.get_action(pi)
            assert a == int(g["c%d_action" % ci][t])
            assert np.random.get_state()[2] == int(g["c%d_mt_pos" % ci][t])
            root_id = root_id + (int(a),)
        assert agent.get_name() == "ZeroAgent" and agent.root_id == root_id[:-1]
        agent.reset()
        assert agent.root_id is None


def test_zero_agent_native_pvnet_path():
    """A PVNet-shaped Agent.model runs on the MFMA forward; get_pv matches the torch module."""
    import torch
    from alpha_omok_amd import agents
    from alpha_omok_amd.pvnet import PVNet
    agents.PRINT_MCTS = False
    B, S = 9, 48
    model = PVNet(2, 5, 64, B)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(2, 5, 64, B, 9).items()})
    model = model.cuda().eval()
    agent = agents.ZeroAgent(B, S, 5, noise=True)
    agent.model = model
    np.random.seed(3)
    root = (0,)
    inherited = 0
    for t in range(3):
        pi = agent.get_pi(root, 1)
        # agents.py:105-132: S simulations on top of what the root brings along -- the child that becomes the next root was
        # expanded by the first of its n visits, so it inherits n - 1 (visit.sum() = S, then inherited + S)
        vis = agent.get_visit()
        assert vis.sum() == inherited + S, (t, vis.sum(), inherited)
        assert abs(pi.sum() - 1) < 1e-12
        a = int(np.argmax(pi))
        inherited = max(int(vis[a]) - 1, 0)
        root = root + (a,)
    assert agent._evaluator.native_net(model, B, 5) is not None
    p, v = agent.get_pv(root)
    with torch.no_grad():
        from alpha_omok_amd import utils
        rp, rv = model(torch.from_numpy(utils.get_state_pt(root, B, 5)[None]).float().cuda())
    assert np.abs(p - rp.cpu().numpy()[0]).max() < 1e-4 and abs(float(v) - float(rv[0])) < 1e-4
    assert p.dtype == np.float32


def test_self_play_single_episode_matches_reference_memory(oracle):
    """np.random.seed(s); main.self_play(1) -> cur_memory / rep_memory / result as the reference
    produced them (tests/golden/gv9, captured from main.self_play with the same stub model)."""
    import alpha_omok_amd.main as main
    g = load_golden("gv9_self_play_memory")
    for ci in range(int(g["ncases"])):
        B, S, mode, seed = g["c%d_cfg" % ci].tolist()
        n_ep = int(g["c%d_episodes" % ci])
        main.PRINT_SELFPLAY = False
        main.configure(board_size=B, n_mcts=S, model=StubModel(oracle, mode), seed=0)
        main.cur_memory.clear()
        main.rep_memory.clear()
        main.reset_iter(main.result, main.cur_memory)
        np.random.seed(seed)
        # one episode: the default path; three episodes: the reference's sequential one-stream schedule (opt-in)
        main.self_play(n_ep, single_stream=(n_ep > 1))
        assert np.random.get_state()[2] == int(g["c%d_mt_pos" % ci])
        cm = list(main.cur_memory)
        assert len(cm) == len(g["c%d_z" % ci])
        np.testing.assert_array_equal(np.stack([m[0] for m in cm]).astype(np.float32), g["c%d_state" % ci])
        np.testing.assert_array_equal(np.stack([m[1] for m in cm]), g["c%d_pi" % ci])
        np.testing.assert_array_equal(np.array([m[2] for m in cm]), g["c%d_z" % ci])
        assert [main.result[k] for k in ("Black", "White", "Draw")] == g["c%d_result" % ci].tolist()
        assert len(main.rep_memory) == int(g["c%d_rep_len" % ci])
        rm = list(main.rep_memory)[:16]
        np.testing.assert_array_equal(np.stack([m[1] for m in rm]), g["c%d_rep_pi_head" % ci])
        np.testing.assert_array_equal(np.stack([m[0] for m in rm]).astype(np.float32), g["c%d_rep_state_head" % ci])
        assert cm[0][0].dtype == np.float64 and cm[0][1].dtype == np.float64


def test_self_play_concurrent_episodes_match_oracle_games(oracle):
    """n episodes side by side (with slot refill) == n sequential oracle games with the same seeds."""
    import alpha_omok_amd.main as main
    B, S, mode, n = 9, 40, 1, 7
    main.MAX_CONCURRENT = 4                      # forces refills: 7 episodes through 4 slots
    main.configure(board_size=B, n_mcts=S, model=StubModel(oracle, mode), seed=123)
    main.cur_memory.clear()
    main.rep_memory.clear()
    main.reset_iter(main.result, main.cur_memory)
    main.self_play(n)
    main.MAX_CONCURRENT = 4096
    cm = list(main.cur_memory)
    ag = oracle.Agent(B, S, 5, noise=True, evaluator="stub%d" % mode)
    off = 0
    res = {1: 0, 2: 0, 3: 0}
    for ep in range(n):
        moves, pis, vis, win = ag.self_play_game(123 + ep, main.TAU_THRES)
        res[win] += 1
        zb = {1: 1.0, 2: -1.0, 3: 0.0}[win]
        root = (0,)
        for t in range(len(moves)):
            s, pi, z = cm[off + t]
            np.testing.assert_array_equal(pi, pis[t], err_msg="episode %d ply %d" % (ep, t))
            np.testing.assert_array_equal(s.astype(np.float32), oracle.get_state_pt(list(root)[1:], B, 5))
            assert z == (zb if t % 2 == 0 else -zb)
            root = root + (int(moves[t]),)
        off += len(moves)
    assert off == len(cm)
    assert [main.result[k] for k in ("Black", "White", "Draw")] == [res[1], res[2], res[3]]
    assert len(main.rep_memory) == min(8 * len(cm), main.MEMORY_SIZE)


def test_reproducible_mode_makes_an_episode_independent_of_its_neighbours():
    """configure(reproducible=True) = ao_net_set_mode 6: the per-layer split-fp16 kernels evaluate every batch size, so an
    episode's samples depend on its seed only. The same three seeds are played alone (one slot each), two at a time and
    among 150 other episodes with a real (default-initialised) network: identical samples, bit for bit. (In the default
    mode the kernel family follows the number of active games -- per-board path, per-layer kernels, resident trunk --
    and their fp32 roundings differ by ~1e-7, which a long game eventually turns into another move.)"""
    import torch
    import alpha_omok_amd.main as main
    B, S = 9, 24
    seeds3 = [901, 902, 903]

    def play(n, seeds, conc):
        torch.manual_seed(3)
        main.MAX_CONCURRENT = conc
        main.configure(board_size=B, n_mcts=S, n_blocks=2, in_planes=5, out_planes=128, seed=0, reproducible=True, node_cap=0, strict=True)
        main.cur_memory.clear()
        main.rep_memory.clear()
        main.self_play(n, seeds=seeds)
        main.MAX_CONCURRENT = 4096
        mem = list(main.cur_memory)
        eps = []
        for s, p, z in mem:
            if not s[:4].any():
                eps.append([])
            eps[-1].append((s.copy(), p.copy(), z))
        return eps

    alone = [play(1, [sd], 1)[0] for sd in seeds3]
    pair = play(3, seeds3, 2)
    crowd = play(153, seeds3 + list(range(5000, 5150)), 4096)[:3]
    for e in range(3):
        for other in (pair[e], crowd[e]):
            assert len(other) == len(alone[e]), "episode %d: %d plies alone, %d with neighbours" % (e, len(alone[e]), len(other))
            for (s0, p0, z0), (s1, p1, z1) in zip(alone[e], other):
                assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
    main.configure(board_size=B, n_mcts=S, n_blocks=2, out_planes=128, seed=0, reproducible=False, strict=False)
    main.release_engine()


def test_carry_over_self_play_returns_each_calls_own_episodes():
    """configure(carry_over=True): slots freed near the end of a call start the NEXT calls' episodes, the engine never runs
    down to one game. With an unchanged network and one kernel family (reproducible=True) every episode must still be the game
    its seed fixes: three calls of 7 episodes on 4 slots give call by call the same cur_memory / results / replay as the
    synchronous schedule, and games of later calls are in flight when a call returns."""
    import torch
    import alpha_omok_amd.main as main
    B, S, N = 9, 16, 7

    def run(carry):
        torch.manual_seed(5)
        main.MAX_CONCURRENT = 4
        main.configure(board_size=B, n_mcts=S, n_blocks=2, in_planes=5, out_planes=128, seed=40, reproducible=True, node_cap=0,
                       strict=True, carry_over=carry)
        main.result.update(Black=0, White=0, Draw=0)
        main.rep_memory.clear()
        out = []
        in_flight = []
        for _ in range(3):
            main.cur_memory.clear()
            ret = main.self_play(N)
            assert ret['episodes'] == N and ret['moves'] == len(main.cur_memory)
            out.append([(s.copy(), p.copy(), z) for s, p, z in main.cur_memory])
            in_flight.append(int(main._pool.active.sum()) if main._pool is not None else 0)
        res = dict(main.result)
        rep = [(s.copy(), p.copy(), z) for s, p, z in list(main.rep_memory)]
        main.MAX_CONCURRENT = 4096
        return out, res, rep, in_flight

    sync, res0, rep0, fl0 = run(False)
    carry, res1, rep1, fl1 = run(True)
    assert fl0 == [0, 0, 0] and all(f > 0 for f in fl1), (fl0, fl1)
    assert res0 == res1
    for c in range(3):
        assert len(sync[c]) == len(carry[c])
        for (s0, p0, z0), (s1, p1, z1) in zip(sync[c], carry[c]):
            assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
    assert len(rep0) == len(rep1)
    for (s0, p0, z0), (s1, p1, z1) in zip(rep0, rep1):
        assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
    # a call with another n_selfplay cannot use the games in flight: they are dropped, the call is still complete
    main.cur_memory.clear()
    ret = main.self_play(3)
    assert ret['episodes'] == 3 and sum(1 for s, _, _ in main.cur_memory if not s[:4].any()) == 3
    with pytest.raises(ValueError):
        main.self_play(3, seeds=[1, 2, 3])
    main.configure(board_size=B, n_mcts=S, n_blocks=2, out_planes=128, seed=0, reproducible=False, strict=False, carry_over=False)
    main.release_engine()


def test_carry_over_plays_ahead_while_an_overlapped_pass_is_running(monkeypatch):
    """carry_over + overlap_train: a call whose own episodes are complete does not sit in train_join while the pass is still at
    work -- it keeps playing the games of the calls to come (same frozen weights) and joins when the pass ends. With a pass that
    leaves the weights alone (patched: it only takes time) every call must still return exactly its own episodes, sample for sample
    what the plain carry-over schedule returns, and searches must have been made during the wait."""
    import time
    import torch
    import alpha_omok_amd.main as main
    B, S, N = 9, 16, 7
    monkeypatch.setattr(main, "_train_plan", lambda: (0, [], []))

    def fake_pass(plan, n_epochs, publish=True):           # ends once five searches were made for later calls' games (or gives up)
        t0 = time.time()
        base = main.played_ahead[0]
        while main.played_ahead[0] < base + 5 and time.time() - t0 < 20.0:
            time.sleep(0.002)
        return []
    monkeypatch.setattr(main, "_train_execute", fake_pass)

    def run(overlap):
        torch.manual_seed(5)
        main.MAX_CONCURRENT = 4
        main.configure(board_size=B, n_mcts=S, n_blocks=2, in_planes=5, out_planes=128, seed=40, reproducible=True, node_cap=0,
                       strict=True, carry_over=True, overlap_train=overlap)
        main.result.update(Black=0, White=0, Draw=0)
        main.rep_memory.clear()
        main.played_ahead[0] = 0
        out, waited = [], []
        for c in range(3):
            main.cur_memory.clear()
            if overlap and c > 0:
                main.train_async(1, c)
                assert main._train_job['thread'] is not None
            w0 = main.phase_seconds['train_wait']
            ret = main.self_play(N)
            waited.append(main.phase_seconds['train_wait'] - w0)
            assert ret['episodes'] == N and ret['moves'] == len(main.cur_memory) and main._train_job is None
            out.append([(s.copy(), p.copy(), z) for s, p, z in main.cur_memory])
        return out, main.played_ahead[0], waited

    try:
        plain, ahead0, _ = run(False)
        over, ahead1, waited = run(True)
        assert ahead0 == 0 and ahead1 >= 10, (ahead0, ahead1)
        assert max(waited) < 1.0, waited                   # (the passes' time was spent playing, not waiting)
        for c in range(3):
            assert len(plain[c]) == len(over[c])
            for (s0, p0, z0), (s1, p1, z1) in zip(plain[c], over[c]):
                assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
    finally:
        main.MAX_CONCURRENT = 4096
        monkeypatch.undo()
        main.configure(board_size=B, n_mcts=S, n_blocks=2, out_planes=128, seed=0, reproducible=False, strict=False, carry_over=False,
                       overlap_train=False)
        main.release_engine()


def test_over_subscribed_self_play_plays_the_same_episodes():
    """configure(oversubscribe=1.5): 72 game slots on 48 rows of the evaluation batch -- the tree kernel hands out the rows per
    simulation (terminal leaves take none), a share of the games sits out every launch, a leaf that finds the batch full is
    evaluated one launch later -- and configure(rows='dynamic') without over-subscription. Every game still runs the reference's
    strictly sequential search, so with one kernel family (reproducible=True) each episode must be the game its seed fixes: the
    same cur_memory, results and replay as the per-move packing on 48 slots, with refills (100 episodes), on a TRAINED network
    (terminal leaves inside the batches: tests/golden/trained_2block_9x9.npz)."""
    import sys
    import torch
    import alpha_omok_amd.main as main
    from alpha_omok_amd.pvnet import PVNet
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_trained_fixture import load
    sd = load(os.path.join(REPO, "tests", "golden", "trained_2block_9x9.npz"))
    B, S, N = 9, 64, 100

    def run(over, rows):
        model = PVNet(2, 5, 128, B)
        model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
        main.MAX_CONCURRENT = 48
        main.configure(board_size=B, n_mcts=S, n_blocks=2, in_planes=5, out_planes=128, seed=60, model=model.cuda().eval(), reproducible=True,
                       node_cap=0, strict=True, carry_over=False, oversubscribe=over, rows=rows)
        main.result.update(Black=0, White=0, Draw=0)
        main.rep_memory.clear()
        main.cur_memory.clear()
        ret = main.self_play(N)
        assert ret['episodes'] == N
        eng = main._engine
        info = (eng.G, eng.row_stats(), dict(main.search_totals))
        out = [(s.copy(), p.copy(), z) for s, p, z in main.cur_memory]
        res = dict(main.result)
        main.MAX_CONCURRENT = 4096
        return out, res, info

    base, res0, (g0, rs0, st0) = run(1.0, 'static')
    over, res1, (g1, rs1, st1) = run(1.5, 'auto')
    dyn, res2, (g2, rs2, st2) = run(1.0, 'dynamic')
    assert (g0, g1, g2) == (48, 72, 48)
    assert rs0['launches'] == 0 and rs1['launches'] > 0 and rs2['launches'] > 0
    shape = lambda st: tuple(st[k] for k in ('levels', 'ties', 'terminal', 'evaluated'))   # ('searches' = engine calls: follows the slot count)
    assert st0['terminal'] > 0 and shape(st0) == shape(st1) == shape(st2)          # the same simulations, level for level
    assert rs1['rows_live'] == st1['evaluated'] and rs2['rows_live'] == st2['evaluated'] and rs2['waits'] == 0
    assert res0 == res1 == res2
    for other in (over, dyn):
        assert len(base) == len(other)
        for (s0, p0, z0), (s1, p1, z1) in zip(base, other):
            assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
    main.configure(board_size=B, n_mcts=S, n_blocks=2, out_planes=128, seed=0, reproducible=False, strict=False, carry_over=False,
                   oversubscribe=1.0, rows='auto')
    main.release_engine()


def test_overlapped_training_keeps_the_searches_on_the_weights_exported_before_it():
    """main.train_async (configure(overlap_train=True); the loop: main.py:377-414): the pass runs on a worker thread and a side stream
    while the next self_play call is being played. That call's searches must see the weights exported BEFORE the pass -- its samples
    equal the samples of a run that never trains --, the pass must be the one a serial execution of the same schedule makes
    (overlap_train='serial': same plan, executed inside train_join), the samples are appended after the join, and the call after
    that plays with the NEW weights."""
    import threading
    import torch
    import alpha_omok_amd.main as main
    B, S, N = 9, 16, 24

    def run(mode):
        torch.manual_seed(5)
        import random
        main.MAX_CONCURRENT = 32
        main.configure(board_size=B, n_mcts=S, n_blocks=1, in_planes=5, out_planes=128, seed=40, reproducible=True, device_replay=True,
                       overlap_train=mode if mode != 'none' else False)
        main.TRAIN_STEPS, main.BATCH_SIZE = 12, 32
        main.result.update(Black=0, White=0, Draw=0)
        main.rep_memory.clear(); main.cur_memory.clear()
        main.step = 0
        random.seed(9)
        out = {}
        main.self_play(N)
        main.reset_iter(main.result, main.cur_memory)
        if mode != 'none':
            main.train_async(1, 1)
            out['threaded'] = main._train_job['thread'] is not None
            out['frozen'] = main._evaluator._frozen
        n_rep = len(main.rep_memory)
        main.self_play(N)                                  # (joins the pass before it appends)
        out['joined'] = main._train_job is None and not main._evaluator._frozen and not main.Agent.model.training
        out['call1'] = [(s.copy(), p.copy(), z) for s, p, z in main.cur_memory]
        out['appended'] = len(main.rep_memory) - n_rep
        out['losses'] = main.last_train_losses
        out['step'] = main.step
        out['weights'] = {k: v.detach().cpu().clone() for k, v in main.Agent.model.state_dict().items()}
        main.reset_iter(main.result, main.cur_memory)
        main.self_play(8)
        out['call2'] = [(s.copy(), p.copy(), z) for s, p, z in main.cur_memory]
        main.reset_iter(main.result, main.cur_memory)
        return out

    try:
        none, serial, thread = run('none'), run('serial'), run(True)
        assert thread['threaded'] and not serial['threaded'] and thread['frozen'] and serial['frozen']
        assert thread['joined'] and serial['joined']
        assert not any(t.name == "alpha_omok_amd.train" for t in threading.enumerate())
        assert none['step'] == 0 and serial['step'] == 12 and thread['step'] == 12
        assert thread['appended'] == serial['appended'] == none['appended'] == 8 * len(none['call1'])
        # the call played beside the pass = the call of a run that never trained, sample for sample
        for other in (serial, thread):
            assert len(other['call1']) == len(none['call1'])
            for (s0, p0, z0), (s1, p1, z1) in zip(none['call1'], other['call1']):
                assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
        # the pass itself: same mini-batches from the same weights, next to a running search or alone
        assert len(thread['losses']) == len(serial['losses']) == 12
        np.testing.assert_allclose(np.array(thread['losses'][0]), np.array(serial['losses'][0]), rtol=0, atol=2e-6)   # same weights, same batch
        np.testing.assert_allclose(np.array(thread['losses']), np.array(serial['losses']), rtol=0, atol=2e-2)   # (then torch's atomics, see below)
        # (the weights: torch's conv backward sums with atomics, and Adam turns the sign of a near-zero gradient into +-LR per step --
        # elements differ between ANY two runs of the same pass; what holds is Adam's bound on a parameter's travel, and BatchNorm's
        # running statistics agreeing to a few per cent)
        params = {k for k, _ in main.Agent.model.named_parameters()}
        for k, v in serial['weights'].items():
            if not v.dtype.is_floating_point:
                assert torch.equal(thread['weights'][k], v), k                    # (num_batches_tracked)
            elif k in params:
                d = (thread['weights'][k] - v).abs()
                assert float(d.max()) <= 2 * main.LR * 12, (k, float(d.max()))
            else:
                np.testing.assert_allclose(thread['weights'][k].numpy(), v.numpy(), rtol=5e-2, atol=1e-2, err_msg=k)
        moved = max(float((serial['weights'][k] - none['weights'][k]).abs().max()) for k in none['weights']
                    if serial['weights'][k].dtype.is_floating_point)
        assert moved > 1e-4                                 # (it did train)
        # ... and the call after the join plays with the new weights
        same = len(none['call2']) == len(thread['call2']) and all(np.array_equal(a[1], b[1]) for a, b in zip(none['call2'], thread['call2']))
        assert not same
    finally:
        main.TRAIN_STEPS, main.BATCH_SIZE, main.MAX_CONCURRENT = None, 32, 4096
        main.configure(board_size=B, n_mcts=S, n_blocks=2, out_planes=128, seed=0, reproducible=False, overlap_train=False)
        main.release_engine()


def test_self_play_reports_arena_trims_and_strict_mode_raises(oracle):
    """A tree arena too small for what the searches keep from move to move makes re-rooting forget subtrees -- a
    divergence from the reference's never-pruned dict (agents.py:52) that must not pass silently: self_play returns /
    main.trim_stats carry the counters, configure(strict=True) raises; a roomy arena reports zeros."""
    import alpha_omok_amd.main as main
    B, S = 9, 60
    main.configure(board_size=B, n_mcts=S, model=StubModel(oracle, 1), seed=5, node_cap=S + 2, strict=False)
    main.cur_memory.clear()
    main.rep_memory.clear()
    out = main.self_play(3)
    assert out["episodes"] == 3 and out["moves"] == len(main.cur_memory)
    assert out["reroots_trimmed"] > 0 and out["subtrees_dropped"] > 0 and main.trim_stats["reroots_trimmed"] == out["reroots_trimmed"]
    main.configure(board_size=B, n_mcts=S, model=StubModel(oracle, 1), seed=5, node_cap=S + 2, strict=True)
    with pytest.raises(main.TreeTrimmed):
        main.self_play(3)
    main.configure(board_size=B, n_mcts=S, model=StubModel(oracle, 1), seed=5, node_cap=0, strict=True)
    main.cur_memory.clear()
    out = main.self_play(3)
    assert out["reroots_trimmed"] == 0 and out["subtrees_dropped"] == 0
    main.configure(node_cap=0, strict=False)
    main.release_engine()


def test_head_to_head_two_agents_match_oracle(oracle):
    """eval_main's call pattern: two ZeroAgents (noise off, tau 0) alternate on one game, each
    re-rooting two plies down after the opponent's reply -- against two oracle agents fed the same way."""
    from alpha_omok_amd import agents, evaluate, utils
    agents.PRINT_MCTS = False
    B, S = 9, 36
    pa, pb = agents.ZeroAgent(B, S, 5, noise=False), agents.ZeroAgent(B, S, 5, noise=False)
    pa.model, pb.model = StubModel(oracle, 1), StubModel(oracle, 0)
    oa = oracle.Agent(B, S, 5, noise=False, evaluator="stub1")
    ob = oracle.Agent(B, S, 5, noise=False, evaluator="stub0")
    np.random.seed(11)
    win, moves = evaluate.play_match(pa, pb, B, enemy_turn=1, max_plies=30)
    # oracle replay of the same protocol on one shared stream (both oracle agents share one Rng
    # the way the reference's players share np.random)
    shared = oracle.Rng(11)
    root = (0,)
    for t, mv in enumerate(moves):
        ag = ob if t % 2 == 1 else oa
        ag.rng.set_state(shared.state_words(), shared.pos)
        pi, vis, pol = ag.get_pi(root, 0)
        shared.set_state(ag.rng.state_words(), ag.rng.pos)
        k = int((pi == pi.max()).sum())
        a = int(np.flatnonzero(pi == pi.max())[shared.choice(k)])
        assert a == mv, "ply %d" % t
        root = root + (a,)
    assert np.random.get_state()[2] == shared.pos
    assert win == oracle.check_win(oracle.get_board(list(root)[1:], B), 5)
    pe, ee = evaluate.elo(1500.0, 1500.0, 1.0, 0.0)
    assert abs(pe - 1516.0) < 1e-9 and abs(ee - 1484.0) < 1e-9


def test_evaluate_matches_reference_eval_main(oracle):
    """evaluate.evaluate (sequential matches through two drop-in ZeroAgents on the process-global stream) against
    gv13: moves, the mover's visit counts, stream position after every move, winners, result tally and ELO
    as the reference's eval_main.Evaluator.get_action / GameState.step / elo produced them."""
    from alpha_omok_amd import agents, evaluate
    agents.PRINT_MCTS = False
    g = load_golden("gv13_eval_head_to_head")
    for ci in range(int(g["ncases"])):
        B, SP, SE, mp, me, seed, n_match = g["c%d_cfg" % ci].tolist()
        pa, pb = agents.ZeroAgent(B, SP, 5, noise=False), agents.ZeroAgent(B, SE, 5, noise=False)
        pa.model, pb.model = StubModel(oracle, mp), StubModel(oracle, me)
        np.random.seed(seed)
        log = []

        class Spy:
            def __init__(self, ag):
                self.ag = ag

            def get_pi(self, root_id, tau):
                pi = self.ag.get_pi(root_id, tau)
                log.append((self.ag.get_visit().copy(), int(np.random.get_state()[2])))
                return pi

            def reset(self):
                self.ag.reset()

        result, (pe, ee), games = evaluate.evaluate(Spy(pa), Spy(pb), B, n_match=n_match, return_games=True)
        k = 0
        for i, (win, moves) in enumerate(games):
            want = g["c%d_moves" % ci][i]
            assert moves == want[want >= 0].tolist(), (ci, i)
            assert win == int(g["c%d_win" % ci][i])
            for t in range(len(moves)):
                np.testing.assert_array_equal(log[k][0], g["c%d_visit" % ci][i][t])
                assert log[k][1] == int(g["c%d_mt_pos" % ci][i][t])
                k += 1
        assert [result[x] for x in ("Player", "Enemy", "Draw")] == g["c%d_result" % ci].tolist()
        np.testing.assert_allclose([pe, ee], g["c%d_elo" % ci][-1], rtol=0, atol=1e-9)


def test_mixed_match_rollout_player_monitor_and_zero_enemy_match_reference(oracle):
    """eval_main.py:137-151 with a rollout PLAYER (PUCTAgent / UCTAgent), the monitor ZeroAgent that searches the same
    root right after it, and a ZeroAgent enemy -- three engines taking turns on the process-global stream. Moves,
    winner, the monitor's visit counts and the stream position after the match equal the reference's (gv13 m-cases)."""
    from alpha_omok_amd import agents, evaluate
    agents.PRINT_MCTS = False
    g = load_golden("gv13_eval_head_to_head")
    for mi in range(int(g["nmixed"])):
        B, mode, SP, SE, SM, me, mm, seed, enemy_turn = g["m%d_cfg" % mi].tolist()
        player = (agents.PUCTAgent if mode == 0 else agents.UCTAgent)(B, SP)
        enemy = agents.ZeroAgent(B, SE, 5, noise=False)
        enemy.model = StubModel(oracle, me)
        monitor = agents.ZeroAgent(B, SM, 5, noise=False)
        monitor.model = StubModel(oracle, mm)
        seen = []
        orig = monitor.get_pi

        def spy(root_id, tau, orig=orig, seen=seen, monitor=monitor):
            pi = orig(root_id, tau)
            seen.append(monitor.get_visit().copy())
            return pi

        monitor.get_pi = spy
        np.random.seed(seed)
        win, moves = evaluate.play_match(player, enemy, B, enemy_turn, monitor=monitor)
        assert moves == g["m%d_moves" % mi].tolist(), mi
        assert win == int(g["m%d_win" % mi])
        assert np.random.get_state()[2] == int(g["m%d_mt_pos" % mi][-1])
        np.testing.assert_array_equal(np.stack(seen), g["m%d_monitor_visit" % mi])


def test_evaluate_batched_equals_sequential_matches(oracle):
    """All matches of eval_main.py:204-333 concurrently (two G = n_match engines, one ao_set_roots launch per
    side and ply): match i must be exactly `np.random.seed(seed + i); play_match(...)` with the colours of
    match i -- moves, winner, and from those the tally and ELO."""
    import torch
    from alpha_omok_amd import agents, evaluate
    from alpha_omok_amd.pvnet import PVNet
    agents.PRINT_MCTS = False
    B, n_match, seed = 9, 6, 40
    for kind in ("stub", "native"):
        if kind == "stub":
            mp, me = StubModel(oracle, 1), StubModel(oracle, 0)
            SP, SE = 30, 24
        else:
            torch.manual_seed(5)
            mp, me = PVNet(1, 5, 32, B).cuda().eval(), PVNet(2, 5, 32, B).cuda().eval()
            SP, SE = 20, 28
        res_b, elo_b, games_b = evaluate.evaluate_batched(mp, me, B, SP, SE, n_match=n_match, seed=seed, max_plies=40)
        pa, pb = agents.ZeroAgent(B, SP, 5, noise=False), agents.ZeroAgent(B, SE, 5, noise=False)
        pa.model, pb.model = mp, me
        enemy_turn = 1
        for i in range(n_match):
            np.random.seed(seed + i)
            win, moves = evaluate.play_match(pa, pb, B, enemy_turn, max_plies=40)
            assert moves == games_b[i][1], (kind, i)
            assert win == games_b[i][0]
            enemy_turn ^= 1
        assert sum(res_b.values()) == n_match


def test_evaluate_batched_with_rollout_player_and_monitor(oracle):
    """evaluate_batched with a PUCT / UCT player ('puct' / 'uct'), a ZeroAgent enemy and the monitor: every match
    equals `np.random.seed(seed + i); play_match(rollout agent, ZeroAgent, monitor=...)`; match 0 of each kind with
    gv13's seeds reproduces the reference's own mixed match."""
    from alpha_omok_amd import agents, evaluate
    agents.PRINT_MCTS = False
    g = load_golden("gv13_eval_head_to_head")
    for mi in range(int(g["nmixed"])):
        B, mode, SP, SE, SM, me, mm, seed, enemy_turn = g["m%d_cfg" % mi].tolist()
        kind = 'puct' if mode == 0 else 'uct'
        n_match = 4
        res, elos, games = evaluate.evaluate_batched(kind, StubModel(oracle, me), B, SP, SE, n_match=n_match, seed=seed,
                                                     monitor_model=StubModel(oracle, mm), n_mcts_monitor=SM)
        player = (agents.PUCTAgent if mode == 0 else agents.UCTAgent)(B, SP)
        enemy = agents.ZeroAgent(B, SE, 5, noise=False)
        enemy.model = StubModel(oracle, me)
        monitor = agents.ZeroAgent(B, SM, 5, noise=False)
        monitor.model = StubModel(oracle, mm)
        et = 1
        for i in range(n_match):
            np.random.seed(seed + i)
            win, moves = evaluate.play_match(player, enemy, B, et, monitor=monitor)
            assert moves == games[i][1] and win == games[i][0], (kind, i)
            et ^= 1
        if enemy_turn == 1:        # gv13's match has the colours of batched match 0
            assert games[0][1] == g["m%d_moves" % mi].tolist() and games[0][0] == int(g["m%d_win" % mi])
        assert sum(res.values()) == n_match


def test_run_loop_iterations_save_and_train(tmp_path):
    """main.run = the reference's __main__ loop (main.py:377-414): iteration 0 only plays, later
    iterations play one game and train on it; checkpoints appear when n_iter % save_every == 0 and
    load_data resumes from their file names."""
    import os
    from alpha_omok_amd import main
    main.configure(board_size=9, n_mcts=8, n_blocks=1, seed=3)
    main.rep_memory.clear(); main.cur_memory.clear()
    main.step = 0; main.start_iter = 0
    n = main.run(total_iter=3, n_selfplay=12, save_every=2, directory=str(tmp_path))
    assert n == 3
    assert main.step > 0                                   # iterations 1 and 2 trained
    assert len(main.cur_memory) == 0 and main.result == {'Black': 0, 'White': 0, 'Draw': 0}
    files = sorted(os.listdir(tmp_path))
    models = [f for f in files if f.endswith('_step_model.pickle')]
    datasets = [f for f in files if f.endswith('_step_dataset.pickle')]
    assert len(models) == 2 and len(datasets) == 2         # n_iter 0 and 2, named 2 and 4
    assert {int(f.split('_')[1]) for f in models} == {2, 4}
    last = [f for f in models if f.split('_')[1] == '4'][0]
    main.load_data(os.path.join(str(tmp_path), last), os.path.join(str(tmp_path), last.replace('model', 'dataset')))
    assert main.start_iter == 5 and main.step == int(last.split('_')[2])
    assert len(main.rep_memory) > 0
