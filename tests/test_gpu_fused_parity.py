"""GPU: the FUSED search path -- ao_search = k_select / k_expand_select (game header forwarded in registers, four games per
workgroup, bit planes, the planner switching kernels as games end) -- held to the oracle in the regime a TRAINED network puts
it in (descents of 6 - 25 levels, terminal leaves inside the batch, inherited roots of several times S), and the two ways the
evaluation-batch rows can be assigned (packed per move by the host; handed out per simulation by the tree kernel, with or
without over-subscription) held to each other, bit for bit.

Round-4 review, "what's weak" 1: both deep-regime tests of test_gpu_tree_parity.py drive the STEP-WISE protocol (k_select +
k_expand_backup); the path the bench and main.self_play run was oracle-checked on shallow trees only. What makes the checks
below possible is ao_set_eval_log: the (policy, value) a listed game's leaf was evaluated with, recorded straight out of the
fused loop's evaluation batch, so the oracle (agents.py:60-239 restated in C) replays exactly what the engine saw.

Network: tests/golden/trained_2block_9x9.npz, trained by this engine (tools/make_trained_fixture.py)."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

B, S = 9, 400
A = B * B


def _trained_state_dict():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_trained_fixture import load
    return load(os.path.join(REPO, "tests", "golden", "trained_2block_9x9.npz"))


def _evals_of(rec, k):
    """Records of listed game k (rows [launch, n, A + 3]) -> the (policy, value) sequence of its simulations: a record counts
    when the game's leaf was waiting for exactly this evaluation (status 1 / 2) or was terminal (3: the reference evaluates and
    discards -- any values do). The simulation counter restarts with every move and must count up without gaps."""
    out, last_done = [], None
    for r in rec[:, k]:
        st, done = int(r[A + 2]), int(r[A + 1])
        if st not in (1, 2, 3):
            continue
        if last_done is not None and done != 0:
            assert done == last_done + 1, (done, last_done)
        last_done = done
        out.append((r[:A].copy(), np.float32(r[A])))
    return out


def _kernel_family(n_block, boards):
    from alpha_omok_amd.engine import plan_kernel
    for kind in (2, 1):
        try:
            return plan_kernel(n_block, 5, 128, B, int(boards), in_kind=kind)[0].split("<")[0]
        except Exception:
            continue
    return "?"


# retire (non-sampled) games so that the number of active games falls through every boundary of the planner on the way:
# 1024 boards k_layer16hk | < 768 k_row16hk | <= 32 the per-board path with the fused per-game step
_KEEP_AT_PLY = {6: 900, 8: 700, 9: 400, 10: 120, 11: 40, 12: 24, 13: 9}


def _retire(alive, keep, protect):
    on = np.flatnonzero(alive)
    extra = [g for g in on[::-1] if g not in protect]
    for g in extra[:max(0, on.size - keep)]:
        alive[g] = 0


def test_fused_search_trained_net_through_the_planner_boundaries_vs_oracle(oracle):
    """Engine.search (= ao_search) with the trained network, G = 1024, S = 400, 15 plies, games ENDING and not refilled -- and
    retired by the test -- so the active count falls 1024 -> 9 and the planner goes k_layer16hk -> k_row16hk -> per-board path.
    Six sampled games are replayed through the oracle at every ply from the evaluations the fused loop itself recorded: visits,
    post-noise priors, pi, action, MT19937 position and state, win index, bit for bit; size-independent properties for all games."""
    import torch
    from alpha_omok_amd.engine import Engine, Net
    G, PLIES = 1024, 15
    net = Net(2, 5, 128, B, 0)
    net.load_state_dict(_trained_state_dict())
    eng = Engine(B, S, 5, games=G, noise=True)
    seeds = np.arange(47000, 47000 + G, dtype=np.uint32)
    eng.seed_all(seeds)
    sample = [0, 1, 17, 300, G // 2, G - 1]
    log = torch.zeros(((S + 8) * len(sample) * (A + 3),), dtype=torch.float32, device="cuda")
    cursors = {g: [0] for g in sample}
    recs = {g: [] for g in sample}

    def make_agent(g):
        def replay(moves, pl, sim, g=g):
            i = cursors[g][0]
            cursors[g][0] += 1
            return recs[g][i]
        ag = oracle.Agent(B, S, 5, noise=True, evaluator=replay)
        ag.seed(int(seeds[g]))
        return ag

    agents = {g: make_agent(g) for g in sample}
    roots = {g: (0,) for g in sample}
    alive = np.ones(G, np.uint8)
    inherited = np.zeros(G)
    families, depth, terminal_total, checked = [], [], 0, 0
    for t in range(PLIES):
        if t in _KEEP_AT_PLY:
            _retire(alive, _KEEP_AT_PLY[t], set(g for g in sample if alive[g]))
        if not alive.any():
            break
        on = alive != 0
        families.append(_kernel_family(2, int(on.sum())))
        tau = np.full(G, 1 if t < 6 else 0, np.int8)
        eng.set_eval_log(sample, log.data_ptr(), log.numel())          # (re-armed: records of this move from 0)
        pi, vis, pol = eng.search(net, tau=tau, active=alive)
        n_rec = eng.eval_log_count()
        rec = log[:n_rec * len(sample) * (A + 3)].cpu().numpy().reshape(n_rec, len(sample), A + 3)
        st = eng.search_stats()
        n_sims = int(on.sum()) * S + (int(on.sum()) if t == 0 else 0)
        assert st["evaluated"] + st["terminal"] == n_sims
        assert n_rec == S + (1 if t == 0 else 0)
        depth.append(st["levels"] / n_sims)
        terminal_total += st["terminal"]
        np.testing.assert_array_equal(vis[on].sum(axis=1), inherited[on] + S)
        assert np.abs(pi[on].sum(axis=1) - 1).max() < 1e-12 and np.abs(pol[on].sum(axis=1) - 1).max() < 1e-9
        act, win = eng.play()
        assert np.all(vis[on, act[on]] > 0)
        for k, g in enumerate(sample):
            if not alive[g]:
                continue
            tag = "game %d ply %d (%d active, %s)" % (g, t, int(on.sum()), families[-1])
            recs[g] = _evals_of(rec, k)
            cursors[g][0] = 0
            assert len(recs[g]) == S + (1 if t == 0 else 0), tag
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
    # the regime and the kernels were the ones the test is for
    assert max(depth) >= 6.0, depth
    assert terminal_total > 0
    assert {"k_layer16hk", "k_row16hk", "k_conv_cells_h"} <= set(families), families
    assert checked >= 5 * 10, checked
    assert eng.trim_stats() == (0, 0)
    eng.close()
    net.close()


@pytest.mark.parametrize("BB,SS,G,PLIES,CAP,mode", [(9, 400, 1024, 13, 640, 6), (9, 400, 4096, 9, 3072, 0), (10, 48, 160, 44, 112, 0)])
def test_row_assignments_agree_for_every_game_fused_vs_stepwise_vs_per_simulation_rows(BB, SS, G, PLIES, CAP, mode):
    """One arithmetic for every batch size (ao_net_set_mode 6) makes a game's evaluations independent of who else is in the
    batch, so FOUR ways of running the same 1024 games must agree for EVERY game at every ply, bit for bit:
      A  ao_search, rows packed per move by the host (rounds 3 - 4),
      B  the step-wise protocol (begin_move / collect_leaves / net / apply_evals) -- the path the older deep-tree tests check,
      C  ao_search, rows handed out per simulation by the tree kernel (ao_set_row_cap(G): terminal leaves take no row),
      D  the same over-subscribed (640 rows for up to 1024 games: a share of the games sits out each launch, leaves that find
         the batch full wait one launch; more launches per move; its k_expand_select slots are dispatched in k_order's order --
         the games with the deepest descents of the previous move first -- which must not matter to any game either).
    13 plies of the trained network with games ending and being retired.
    Second case: 4096 games in the default mode, where all four run the RESIDENT trunk (k_trunk16hb / k_trunk16h, >= 192 groups;
    its result for a board does not depend on the board's row or neighbours either): the kernel of the headline, reading the live-row
    count (C, D: 3072 rows = 192 groups for 4096 games).
    Third case: a WIDE board (10x10, default-initialised 2-block network, 160 games, 44 plies): all four run k_boardh -- one board per
    workgroup, conv1 on bit planes (A, C, D) or on float planes through k_layer16h (B): the same bits --, C and D with the kernel
    reading the live-row count (112 rows for 160 games)."""
    import torch
    from alpha_omok_amd.engine import Engine, Net
    B, S = BB, SS
    if B == 9:
        net = Net(2, 5, 128, B, 0)
        net.load_state_dict(_trained_state_dict())
    else:
        from alpha_omok_amd.engine import plan_kernel
        from alpha_omok_amd.pvnet import PVNet
        torch.manual_seed(10)
        net = PVNet(2, 5, 128, B).eval().to_native(0)
        assert plan_kernel(2, 5, 128, B, CAP, in_kind=2)[0].startswith("k_boardh<%d, 2>" % B)
    net.set_mode(mode)
    seeds = np.arange(52000, 52000 + G, dtype=np.uint32)
    engs = {}
    for name in "ABCD":
        engs[name] = Engine(B, S, 5, games=G, noise=True, node_cap=(8 if G <= 1024 else 5) * (S + 1))   # (four engines side by side)
        engs[name].seed_all(seeds)
    engs["C"].set_row_cap(G)
    engs["D"].set_row_cap(CAP)
    planes = torch.zeros((G, 5, B, B), dtype=torch.float32, device="cuda")
    alive = np.ones(G, np.uint8)
    evaluated = {n: 0 for n in "ABCD"}
    terminal = 0
    sims_total = 0
    for t in range(PLIES):
        if G == 1024 and t in _KEEP_AT_PLY:
            _retire(alive, _KEEP_AT_PLY[t], set())
        if not alive.any():
            break
        on = alive != 0
        if mode == 0 and B == 9:
            assert int(on.sum()) >= 3072 + 16, "the resident trunk needs 192 groups: shorten the test"
        if B != 9 and int(on.sum()) < 64:
            break                                                         # (below 64 boards the planner leaves k_boardh)
        tau = np.full(G, 1 if t < (6 if B == 9 else 30) else 0, np.int8)
        out = {}
        for name in "ACD":
            out[name] = engs[name].search(net, tau=tau, active=alive)
            st = engs[name].search_stats()
            evaluated[name] += st["evaluated"]
            if name == "A":
                terminal += st["terminal"]
                sims_total += st["evaluated"] + st["terminal"]
        eb = engs["B"]
        eb.begin_move(alive)
        while eb.sims_left() > 0:
            eb.collect_leaves(planes.data_ptr())
            eb.sync()
            p, v = net(planes)
            torch.cuda.synchronize()
            eb.apply_evals(p.data_ptr(), v.data_ptr())
        out["B"] = eb.end_move(tau)
        evaluated["B"] += eb.search_stats()["evaluated"]
        plays = {name: engs[name].play() for name in "ABCD"}
        for name in "BCD":
            tag = "%s vs A, ply %d (%d active)" % (name, t, int(on.sum()))
            for i, what in enumerate(("pi", "visits", "priors")):
                np.testing.assert_array_equal(out[name][i][on], out["A"][i][on], err_msg=what + " " + tag)
            np.testing.assert_array_equal(plays[name][0][on], plays["A"][0][on], err_msg="action " + tag)
            np.testing.assert_array_equal(plays[name][1][on], plays["A"][1][on], err_msg="win " + tag)
        for g in (0, 5, G - 1):
            if alive[g]:
                ref = engs["A"].get_rng_state(g)
                for name in "BCD":
                    got = engs[name].get_rng_state(g)
                    assert got[1] == ref[1] and np.array_equal(got[0], ref[0]), "MT19937 of game %d, %s, ply %d" % (g, name, t)
        act, win = plays["A"]
        alive[on & (win != 0)] = 0
    assert terminal > 0 and len(set(evaluated.values())) == 1, (terminal, evaluated)
    rc, rd = engs["C"].row_stats(), engs["D"].row_stats()
    # per-simulation rows: every evaluated leaf took exactly one row, terminal leaves none
    assert rc["rows_live"] == evaluated["C"] and rc["waits"] == 0, rc
    assert rd["rows_live"] == evaluated["D"], rd
    assert rc["rows_live"] < sims_total                                   # (the per-move packing evaluates sims_total rows)
    # over-subscribed: more launches than simulations per move while more games than rows were alive, batches nearly full
    assert rd["launches"] > rc["launches"], (rc, rd)
    assert engs["A"].row_stats()["launches"] == 0
    for e in engs.values():
        e.close()
    net.close()


def test_over_subscribed_handful_of_games_on_the_per_board_path_vs_oracle(oracle):
    """Over-subscription at the small end: 40 games on 24 rows. The planner takes the per-board network path for 24 boards
    (k_conv_cells_h + k_heads_board, not the fused per-game step: that one keeps the per-move packing), the tree kernel hands
    out the rows, games sit out / leaves wait. Five sampled games against the oracle from the recorded evaluations (records of
    launches in which a game did nothing are skipped by their leaf status), every game against the size-independent properties."""
    import torch
    from alpha_omok_amd.engine import Engine, Net
    G, CAP, S_, PLIES = 40, 24, 48, 8
    net = Net(2, 5, 128, B, 0)
    net.load_state_dict(_trained_state_dict())
    assert _kernel_family(2, CAP) == "k_conv_cells_h"
    eng = Engine(B, S_, 5, games=G, noise=True)
    eng.set_row_cap(CAP)
    seeds = np.arange(61000, 61000 + G, dtype=np.uint32)
    eng.seed_all(seeds)
    sample = [0, 7, 23, 24, G - 1]
    log = torch.zeros((4 * (S_ + 8) * len(sample) * (A + 3),), dtype=torch.float32, device="cuda")
    cur = {g: [0] for g in sample}
    recs = {g: [] for g in sample}

    def make_agent(g):
        def replay(moves, pl, sim, g=g):
            i = cur[g][0]
            cur[g][0] += 1
            return recs[g][i]
        ag = oracle.Agent(B, S_, 5, noise=True, evaluator=replay)
        ag.seed(int(seeds[g]))
        return ag

    agents = {g: make_agent(g) for g in sample}
    roots = {g: (0,) for g in sample}
    inherited = np.zeros(G)
    for t in range(PLIES):
        tau = np.full(G, 1, np.int8)
        eng.set_eval_log(sample, log.data_ptr(), log.numel())
        pi, vis, pol = eng.search(net, tau=tau)
        n_rec = eng.eval_log_count()
        assert n_rec > S_ + (1 if t == 0 else 0), "40 games on 24 rows need more launches than simulations"
        rec = log[:n_rec * len(sample) * (A + 3)].cpu().numpy().reshape(n_rec, len(sample), A + 3)
        np.testing.assert_array_equal(vis.sum(axis=1), inherited + S_)
        act, win = eng.play()
        assert np.all(win == 0)
        for k, g in enumerate(sample):
            tag = "game %d ply %d" % (g, t)
            recs[g] = _evals_of(rec, k)
            cur[g][0] = 0
            assert len(recs[g]) == S_ + (1 if t == 0 else 0), tag
            opi, ovis, opol = agents[g].get_pi(roots[g], 1)
            np.testing.assert_array_equal(vis[g], ovis, err_msg=tag)
            np.testing.assert_array_equal(pol[g], opol, err_msg=tag)
            np.testing.assert_array_equal(pi[g], opi, err_msg=tag)
            oa = agents[g].rng.choice_p(opi)
            assert act[g] == oa, tag
            roots[g] = roots[g] + (int(oa),)
            mt, pos, _, _ = eng.get_rng_state(g)
            assert pos == agents[g].rng.pos, tag
            np.testing.assert_array_equal(mt, agents[g].rng.state_words(), err_msg="mt " + tag)
        inherited = vis[np.arange(G), act] - 1
    rs = eng.row_stats()
    assert rs["launches"] > PLIES * S_ and rs["rows_live"] <= rs["rows_launched"]
    eng.set_eval_log([])
    eng.close()
    net.close()


def test_self_play_with_the_trained_network_replayed_by_the_oracle(oracle):
    """main.self_play(n) -- the whole drop-in loop: Evaluator export of the trained PVNet, ao_search with the planner following
    the shrinking number of active games (k_row16hk -> per-board path), get_action, env step, re-rooting -- with three
    sampled episodes replayed by oracle.self_play_game from the evaluations the engine recorded: every move, every pi."""
    import torch
    import alpha_omok_amd.main as main
    from alpha_omok_amd.pvnet import PVNet
    n = 64
    model = PVNet(2, 5, 128, B)
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in _trained_state_dict().items()})
    model.eval()
    main.MAX_CONCURRENT = 4096
    main.configure(board_size=B, n_mcts=S, n_blocks=2, in_planes=5, out_planes=128, seed=777, model=model.cuda(), node_cap=0, strict=True)
    main.cur_memory.clear()
    main.rep_memory.clear()
    main.reset_iter(main.result, main.cur_memory)
    eng = main._get_engine(n)                          # the engine self_play(n) will use: n episodes on n slots, slot = episode
    sample = [0, 31, 63]
    log = torch.zeros((A * (S + 1) * len(sample) * (A + 3),), dtype=torch.float32, device="cuda")   # a whole game of records
    eng.set_eval_log(sample, log.data_ptr(), log.numel())
    summary = main.self_play(n)
    n_rec = eng.eval_log_count()
    rec = log[:n_rec * len(sample) * (A + 3)].cpu().numpy().reshape(n_rec, len(sample), A + 3)
    eng.set_eval_log([])
    cm = list(main.cur_memory)
    assert summary["episodes"] == n and summary["moves"] == len(cm)
    # samples arrive in episode order: find each sampled episode's block
    lengths = []
    off = 0
    while off < len(cm):          # an episode starts at an empty board
        ln = 1
        while off + ln < len(cm) and cm[off + ln][0][:4].sum() > 0:
            ln += 1
        lengths.append(ln)
        off += ln
    assert len(lengths) == n
    starts = np.concatenate([[0], np.cumsum(lengths)])
    for k, ep in enumerate(sample):
        evs = _evals_of(rec, k)
        cur = [0]

        def replay(moves, pl, sim):
            i = cur[0]
            cur[0] += 1
            return evs[i]
        ag = oracle.Agent(B, S, 5, noise=True, evaluator=replay)
        moves, pis, vis, win = ag.self_play_game(777 + ep, main.TAU_THRES)
        assert len(moves) == lengths[ep], "episode %d: %d plies, the oracle plays %d" % (ep, lengths[ep], len(moves))
        assert cur[0] == len(evs)
        root = (0,)
        zb = {1: 1.0, 2: -1.0, 3: 0.0}[win]
        for t in range(len(moves)):
            s, pi, z = cm[starts[ep] + t]
            np.testing.assert_array_equal(pi, pis[t], err_msg="episode %d ply %d" % (ep, t))
            np.testing.assert_array_equal(s.astype(np.float32), oracle.get_state_pt(list(root)[1:], B, 5))
            assert z == (zb if t % 2 == 0 else -zb)
            root = root + (int(moves[t]),)
    main.release_engine()
