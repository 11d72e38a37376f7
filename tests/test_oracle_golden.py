"""Pins the CPU oracle (oracle/omok_oracle.c) against golden vectors captured from the
unmodified reference (tools/gen_golden.py). CPU-only; these tests are what make the oracle a
trustworthy checker for the HIP path."""
import random

import numpy as np
import pytest

from conftest import load_golden


def test_gv1_check_win(oracle):
    g = load_golden("gv1_check_win")
    for b, (n, k), w in zip(g["boards"], g["size_mark"], g["win"]):
        assert oracle.check_win(b[:n, :n], int(k)) == int(w)
    assert set(g["win"].tolist()) == {0, 1, 2, 3}


def test_gv2_legal_order(oracle):
    g = load_golden("gv2_legal_order")
    nonasc = 0
    for B, mv, order in zip(g["board"], g["moves"], g["order"]):
        mv = mv[mv >= 0].astype(np.int32)
        order = order[order >= 0]
        got = oracle.legal_actions(mv, int(B))
        assert got.tolist() == order.tolist()
        nonasc += int(order.tolist() != sorted(order.tolist()))
    assert nonasc > 50  # the CPython set-order quirk is exercised


def test_legal_order_matches_live_cpython(oracle):
    # Same property checked against this interpreter's own set implementation.
    rnd = random.Random(5)
    for B in (3, 9, 15):
        A = B * B
        for _ in range(400):
            mv = rnd.sample(range(A), rnd.randint(0, A - 1))
            ref = list({a for a in range(A)} - set(mv))
            assert oracle.legal_actions(mv, B).tolist() == ref


def test_gv3_state_planes(oracle):
    g = load_golden("gv3_state_planes")
    for i in range(int(g["count"])):
        m = g["m%d" % i]
        B, C, mv = int(m[0]), int(m[1]), m[2:]
        np.testing.assert_array_equal(oracle.get_state_pt(mv, B, C), g["s%d" % i])
        np.testing.assert_array_equal(oracle.get_board(mv, B), g["b%d" % i])
        assert (len(mv) + 1) % 2 == (1 if int(g["t%d" % i]) == 0 else 0)


def test_gv4_numpy_rng(oracle):
    g = load_golden("gv4_numpy_rng")
    ks = g["choice_k"].tolist()
    for seed in (0, 1, 12345, 4294967295):
        r = oracle.Rng(seed)
        np.testing.assert_array_equal(r.state_words()[:8], g["init_%d" % seed])
        got = [r.choice(k) for k in ks * 5]
        assert got == g["choice_%d" % seed].tolist()
        got = [r.random_sample() for _ in range(8)]
        assert got == g["dbl_%d" % seed].tolist()
        for name, alpha, k in (("dir81", 10 / 81, 81), ("dir17", 10 / 81, 17),
                               ("dir225", 10 / 225, 225), ("dir9", 10 / 9, 9),
                               ("dir4", 10 / 9, 4)):
            np.testing.assert_array_equal(r.dirichlet(alpha, k), g["%s_%d" % (name, seed)])
        r.dirichlet(1.0, 81)  # the generator drew p ~ Dirichlet(1) here
        p = g["p_%d" % seed]
        got = [r.choice_p(p) for _ in range(16)]
        assert got == g["choicep_%d" % seed].tolist()
        onehot = np.zeros(81)
        onehot[37] = 1.0
        assert r.choice_p(onehot) == int(g["choice1h_%d" % seed])
        assert r.pos == int(g["end_pos_%d" % seed])
        np.testing.assert_array_equal(r.state_words(), g["end_state_%d" % seed])


def test_pairwise_sum_matches_numpy(oracle):
    rng = np.random.RandomState(0)
    for n in (9, 81, 225, 128, 129, 7):
        for _ in range(200):
            a = rng.rand(n) * (rng.rand(n) < 0.7)
            assert oracle.pairwise_sum(a) == np.sum(a)


def _check_tree_cases(oracle, g, evaluator_for_case):
    meta = g["meta"]
    for ci, (B, S, mode, seed, plies, tau_thres, noise, nrec, win) in enumerate(meta.tolist()):
        ag = oracle.Agent(B, S, 5, noise=bool(noise), evaluator=evaluator_for_case(ci, mode))
        ag.seed(seed)
        roots = g["c%d_root" % ci]
        for t in range(nrec):
            root = (0,) + tuple(int(x) for x in roots[t] if x >= 0)
            tau = 1 if t < tau_thres else 0
            pi, visit, policy = ag.get_pi(root, tau)
            np.testing.assert_array_equal(visit, g["c%d_visit" % ci][t], err_msg="visit c%d t%d" % (ci, t))
            np.testing.assert_array_equal(policy, g["c%d_policy" % ci][t], err_msg="policy c%d t%d" % (ci, t))
            np.testing.assert_array_equal(pi, g["c%d_pi" % ci][t])
            ch = ag.children(root)
            np.testing.assert_array_equal(ch["w"], g["c%d_w" % ci][t])
            np.testing.assert_array_equal(ch["q"], g["c%d_q" % ci][t])
            order = g["c%d_order" % ci][t]
            assert ch["order"].tolist() == order[order >= 0].tolist()
            action = ag.rng.choice_p(pi)
            assert action == int(g["c%d_action" % ci][t])
            assert ag.rng.pos == int(g["c%d_mt_pos" % ci][t])
            assert int(ag.rng.state_words().astype(np.uint64).sum()) == int(g["c%d_mt_sum" % ci][t])
            assert ag.tree_size() == int(g["c%d_tree_size" % ci][t])


def test_gv5_tree_parity_stub(oracle):
    g = load_golden("gv5_tree_stub")
    _check_tree_cases(oracle, g, lambda ci, mode: "stub%d" % mode)
    # the fixture covers draws / both winners on 3x3 and full 9x9 games
    assert len(set(g["meta"][:, -1].tolist())) >= 2


def test_gv5_tree_parity_deep_roots(oracle):
    g = load_golden("gv5_tree_stub_deeproot")
    _check_tree_cases(oracle, g, lambda ci, mode: "stub%d" % mode)
    # at least one root lists its children in non-ascending order
    nonasc = 0
    for ci in range(len(g["meta"])):
        o = g["c%d_order" % ci][0]
        o = o[o >= 0].tolist()
        nonasc += int(o != sorted(o))
    assert nonasc >= 1


def test_gv5_tree_parity_deep_roots_15x15(oracle):
    """15x15 roots with 150 / 152 / 160 stones (>= 149: CPython's set difference ends in a 128-slot table smaller
    than the largest key, SURVEY Q5 / utils.py:22-27): every root lists its children in non-ascending order."""
    g = load_golden("gv5_tree_stub_deeproot15")
    _check_tree_cases(oracle, g, lambda ci, mode: "stub%d" % mode)
    for ci in range(len(g["meta"])):
        o = g["c%d_order" % ci][0]
        o = o[o >= 0].tolist()
        assert o != sorted(o) and int(g["c%d_root" % ci][0].shape[0]) >= 150


def test_gv6_tree_parity_real_net_replay(oracle):
    g = load_golden("gv6_tree_realnet")
    ep, ev = g["eval_p"], g["eval_v"]
    cursor = [0]

    def replay(moves, planes, sim):
        i = cursor[0]
        cursor[0] += 1
        return ep[i], ev[i]

    _check_tree_cases(oracle, g, lambda ci, mode: replay)
    assert cursor[0] == len(ev)


def test_self_play_game_matches_stepwise(oracle):
    # oo_self_play_game (main.self_play's loop) == stepping get_pi/get_action by hand
    g = load_golden("gv5_tree_stub")
    meta = g["meta"].tolist()
    for ci, (B, S, mode, seed, plies, tau_thres, noise, nrec, win) in enumerate(meta):
        if plies != 0 or not noise:
            continue
        ag = oracle.Agent(B, S, 5, noise=True, evaluator="stub%d" % mode)
        moves, pis, vis, w = ag.self_play_game(seed, tau_thres)
        assert moves.tolist() == g["c%d_action" % ci].tolist()
        assert w == win
        np.testing.assert_array_equal(pis, g["c%d_pi" % ci])
        np.testing.assert_array_equal(vis, g["c%d_visit" % ci])


def test_rollout_agents_match_reference(oracle):
    """PUCTAgent / UCTAgent.get_pi (agents.py:263-614) restated in oracle/rollout_oracle.c: one-hot pi,
    root-child visits (PUCT) / q (UCT), chosen move and np.random stream position after every call,
    for consecutive calls under one np.random.seed (gv11 captured from the reference)."""
    g = load_golden("gv11_rollout_agents")
    for ci in range(int(g["ncases"])):
        mode, B, S, seed, nrec = g["c%d_cfg" % ci].tolist()
        root = (0,) + tuple(int(a) for a in g["c%d_start" % ci])
        rng = oracle.Rng(seed)
        assert nrec >= 1
        for t in range(nrec):
            pi, stat, action, _ = oracle.rollout_search(mode, B, S, root, rng)
            np.testing.assert_array_equal(pi, g["c%d_pi" % ci][t], err_msg="case %d call %d" % (ci, t))
            np.testing.assert_array_equal(stat, g["c%d_stat" % ci][t], err_msg="case %d call %d" % (ci, t))
            assert action == int(g["c%d_action" % ci][t])
            assert rng.pos == int(g["c%d_pos" % ci][t])
            root = root + (action,)


def test_tictactoe_uct_matches_reference(oracle):
    """BASELINE configs[0]: the per-move UCT search of 1_tictactoe_MCTS/mcts_vs.py (selection / expansion /
    simulation / backup, lines 15-131, driven as its __main__ does) under random.seed: q and n of the root
    children, max_action and the complete final state of Python's `random` stream (gv12)."""
    g = load_golden("gv12_tictactoe_uct")
    for ci in range(int(g["ncases"])):
        turn, num_mcts, seed, max_action, pos = g["c%d_cfg" % ci].tolist()
        rng = oracle.PyRandom(seed)
        a, q, n = oracle.ttt_search(g["c%d_board" % ci], turn, num_mcts, rng)
        assert a == max_action, "case %d" % ci
        np.testing.assert_array_equal(q, g["c%d_q" % ci], err_msg="case %d" % ci)
        np.testing.assert_array_equal(n, g["c%d_n" % ci], err_msg="case %d" % ci)
        assert rng.pos == pos
        np.testing.assert_array_equal(rng.state_words(), g["c%d_mt" % ci])


def test_gv13_head_to_head_matches_reference(oracle):
    """Two oracle agents driven the way eval_main drives its players (shared stream, noise off, tau 0, the
    next root id formed from the mover's id + its move) against the reference's Evaluator.get_action /
    GameState.step / elo outputs."""
    g = load_golden("gv13_eval_head_to_head")
    for ci in range(int(g["ncases"])):
        B, SP, SE, mp, me, seed, n_match = g["c%d_cfg" % ci].tolist()
        pa = oracle.Agent(B, SP, 5, noise=False, evaluator="stub%d" % mp)
        pb = oracle.Agent(B, SE, 5, noise=False, evaluator="stub%d" % me)
        shared = oracle.Rng(seed)
        enemy_turn = 1
        for i in range(n_match):
            want = g["c%d_moves" % ci][i]
            want = want[want >= 0]
            root = (0,)
            for t, mv in enumerate(want):
                ag = pb if (t % 2) == enemy_turn else pa
                ag.rng.set_state(shared.state_words(), shared.pos)
                pi, vis, pol = ag.get_pi(root, 0)
                shared.set_state(ag.rng.state_words(), ag.rng.pos)
                assert int(np.argmax(pi)) == int(mv) and pi.sum() == 1.0, (ci, i, t)
                np.testing.assert_array_equal(vis, g["c%d_visit" % ci][i][t])
                assert shared.pos == int(g["c%d_mt_pos" % ci][i][t])
                root = root + (int(mv),)
            assert oracle.check_win(oracle.get_board(list(root)[1:], B), 5) == int(g["c%d_win" % ci][i])
            pa.reset()
            pb.reset()
            enemy_turn ^= 1


def test_gv9_self_play_memory_incl_several_episodes_on_one_stream(oracle):
    """main.self_play(n) of the reference (main.py:122-250): n episodes one after another on ONE np.random stream
    (the agent is reset between episodes, the stream is not). Oracle agent driven the same way against gv9's
    cur_memory (pi and z per ply, stream position at the end); c2 has three episodes."""
    g = load_golden("gv9_self_play_memory")
    for ci in range(int(g["ncases"])):
        B, S, mode, seed = g["c%d_cfg" % ci].tolist()
        ag = oracle.Agent(B, S, 5, noise=True, evaluator="stub%d" % mode)
        ag.seed(seed)
        want_pi, want_z = g["c%d_pi" % ci], g["c%d_z" % ci]
        k = 0
        res = [0, 0, 0]
        for ep in range(int(g["c%d_episodes" % ci])):
            root = (0,)
            start = k
            while True:
                pi, vis, pol = ag.get_pi(root, 1 if len(root) - 1 < 6 else 0)
                np.testing.assert_array_equal(pi, want_pi[k], err_msg="case %d episode %d ply %d" % (ci, ep, k - start))
                root = root + (int(ag.rng.choice_p(pi)),)
                k += 1
                win = oracle.check_win(oracle.get_board(list(root)[1:], B), 5)
                if win != 0:
                    break
            res[win - 1] += 1
            zb = {1: 1.0, 2: -1.0, 3: 0.0}[win]
            for t in range(start, k):
                assert want_z[t] == (zb if (t - start) % 2 == 0 else -zb)
            ag.reset()
        assert k == len(want_z) and res == g["c%d_result" % ci].tolist()
        assert ag.rng.pos == int(g["c%d_mt_pos" % ci])


def test_gv13_mixed_matches_rollout_player_with_monitor(oracle):
    """eval_main.py:137-151 with a rollout PLAYER: PUCT / UCT search, then the monitor ZeroAgent on the same root, then
    the ZeroAgent enemy's reply -- all on one stream. Oracle rollout search + two oracle agents against the reference."""
    g = load_golden("gv13_eval_head_to_head")
    for mi in range(int(g["nmixed"])):
        B, mode, SP, SE, SM, me, mm, seed, enemy_turn = g["m%d_cfg" % mi].tolist()
        enemy = oracle.Agent(B, SE, 5, noise=False, evaluator="stub%d" % me)
        monitor = oracle.Agent(B, SM, 5, noise=False, evaluator="stub%d" % mm)
        shared = oracle.Rng(seed)
        root = (0,)
        k = 0
        for t, mv in enumerate(g["m%d_moves" % mi]):
            if (t % 2) != enemy_turn:
                pi, stat, act, nodes = oracle.rollout_search(mode, B, SP, root, shared)
                monitor.rng.set_state(shared.state_words(), shared.pos)
                _, mvis, _ = monitor.get_pi(root, 0)
                shared.set_state(monitor.rng.state_words(), monitor.rng.pos)
                np.testing.assert_array_equal(mvis, g["m%d_monitor_visit" % mi][k])
                k += 1
            else:
                enemy.rng.set_state(shared.state_words(), shared.pos)
                pi, _, _ = enemy.get_pi(root, 0)
                shared.set_state(enemy.rng.state_words(), enemy.rng.pos)
            assert int(np.argmax(pi)) == int(mv), (mi, t)
            assert shared.pos == int(g["m%d_mt_pos" % mi][t]), (mi, t)
            root = root + (int(mv),)
        assert oracle.check_win(oracle.get_board(list(root)[1:], B), 5) == int(g["m%d_win" % mi])


def test_trained_fixture_loads_and_makes_the_oracle_search_deep(oracle):
    """tests/golden/trained_2block_9x9.npz (a network trained by the engine itself, tools/make_trained_fixture.py): the
    state_dict wire format loads into PVNet, and under the ORACLE's sequential search its sharp priors give the deep,
    terminal-hitting descents the GPU test of the same name relies on (random-init networks stay at depth ~1.9)."""
    import os
    import sys
    import torch
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_trained_fixture import load
    from alpha_omok_amd.pvnet import PVNet
    sd = load(os.path.join(REPO, "tests", "golden", "trained_2block_9x9.npz"))
    net = PVNet(2, 5, 128, 9)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net.eval()
    torch.set_num_threads(2)

    def ev(moves, planes, sim):
        with torch.no_grad():
            p, v = net(torch.from_numpy(np.array(planes, np.float32))[None])
        return p[0].numpy(), float(v[0])

    S = 100
    ag = oracle.Agent(9, S, 5, noise=True, evaluator=ev)
    ag.seed(123)
    root, depths = (0,), []
    for ply in range(8):
        pi, vis, pol = ag.get_pi(root, 1 if ply < 6 else 0)
        assert vis.sum() >= S and abs(pi.sum() - 1) < 1e-12
        depths.append(ag.last_stats()["levels"] / (S + (ply == 0)))
        root = root + (int(ag.rng.choice_p(pi)),)
        if oracle.check_win(oracle.get_board(list(root[1:]), 9), 5):
            break
    assert pol.max() > 0.3                                   # a sharp prior at the root (random init: ~1 / 81 + noise)
    assert max(depths) >= 4.0, depths
