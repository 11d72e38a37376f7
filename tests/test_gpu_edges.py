"""-m gpu: edge cases of the hot path -- other history depths (IN_PLANES = 3, 7), noise off,
error paths that must fail loudly instead of corrupting a search."""
import numpy as np
import pytest

from gpu_helpers import HostEvalRunner

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("inplanes,board,noise", [(3, 9, True), (7, 9, True), (9, 5, True), (5, 9, False)])
def test_other_history_depths_and_noise_off(oracle, inplanes, board, noise):
    """get_state_pt is generic in the channel count (main.py:34 IN_PLANES = 2*history+1); the stub
    evaluator hashes every plane, so a wrong plane changes the visit counts."""
    from alpha_omok_amd.engine import Engine
    S, G = 50, 3
    eng = Engine(board, S, inplanes, games=G, noise=noise)
    run = HostEvalRunner(eng)
    seeds = [31, 32, 33]
    eng.seed_all(seeds)
    ags = [oracle.Agent(board, S, inplanes, noise=noise, evaluator="stub1") for _ in range(G)]
    for g in range(G):
        ags[g].seed(seeds[g])
    roots = [(0,)] * G
    for t in range(7):
        pi, vis, pol = run.move(lambda g, sim, pl: oracle.stub_eval(pl, 1), tau=np.ones(G, np.int8))
        act, win = eng.play()
        for g in range(G):
            opi, ovis, opol = ags[g].get_pi(roots[g], 1)
            np.testing.assert_array_equal(vis[g], ovis, err_msg="C=%d game %d ply %d" % (inplanes, g, t))
            np.testing.assert_array_equal(pol[g], opol)
            assert act[g] == ags[g].rng.choice_p(opi)
            roots[g] = roots[g] + (int(act[g]),)
        if (win != 0).any():
            break
    eng.close()


def test_full_arena_degrades_one_game_at_a_time(oracle):
    """A kept tree that outgrows the arena no longer aborts the batch (round-1 behaviour: ERR_NODE_CAP failed all G
    games): re-rooting keeps node_cap - sims - 1 nodes breadth first and forgets the rest, per game, and counts it.
    With room for ONE search only (node_cap = sims + 2) every re-rooting keeps just the new root; with a large
    arena nothing is dropped and the same games are the oracle's (checked everywhere else)."""
    from alpha_omok_amd.engine import Engine
    S, G = 40, 3
    ev = lambda g, sim, pl: oracle.stub_eval(pl, 1)   # noqa: E731
    tight = Engine(9, S, 5, games=G, noise=True, node_cap=S + 2)
    roomy = Engine(9, S, 5, games=G, noise=True)
    for e in (tight, roomy):
        e.seed_all([3, 4, 5])
    rt, rr = HostEvalRunner(tight), HostEvalRunner(roomy)
    for t in range(6):
        pt, vt, _ = rt.move(ev)
        pr, vr, _ = rr.move(ev)
        if t == 0:
            np.testing.assert_array_equal(vt, vr)          # the first search never re-rooted
        assert (vt.sum(axis=1) >= S).all() and (vt.sum(axis=1) <= S + 1).all()   # nothing inherited: the subtree was dropped
        assert (vr.sum(axis=1) >= S).all()
        for g in range(G):
            assert tight.tree_nodes(g)[0] <= S + 2
        tight.play()
        roomy.play()
    dropped, trimmed = tight.trim_stats()
    assert trimmed > 0 and dropped >= trimmed
    assert roomy.trim_stats() == (0, 0)
    tight.close()
    roomy.close()


@pytest.mark.parametrize("cap_extra,with_oracle", [(None, True), (160, True), (60, True), (12, False), (1, False)])
def test_arena_compacted_only_when_needed_equals_a_copy_after_every_move(oracle, monkeypatch, cap_extra, with_oracle):
    """k_play moves the root to the played child IN PLACE while the next search still fits behind what the arena holds
    (nodes_used <= node_cap - sims - 1) and only otherwise copies the subtree into the other arena (k_reroot, one workgroup per
    game); AO_COMPACT_ALWAYS=1 copies after every move as rounds 1 - 5 did. Same visits, priors, pi, moves, reachable tree and
    trims, move by move, for a roomy arena (no copy at all in 10 moves), arenas that fill up every few moves, and arenas so
    tight that subtrees are dropped; where nothing is dropped the games are also the oracle's (agents.py:84-132)."""
    from alpha_omok_amd.engine import Engine
    S, G, B = 48, 4, 9
    cap = 0 if cap_extra is None else S + 1 + cap_extra            # 0 = the default arena
    seeds = [11, 12, 13, 14]
    engs = []
    for always in (False, True):
        if always:
            monkeypatch.setenv("AO_COMPACT_ALWAYS", "1")
        else:
            monkeypatch.delenv("AO_COMPACT_ALWAYS", raising=False)
        e = Engine(B, S, 5, games=G, noise=True, node_cap=cap)
        e.seed_all(seeds)
        engs.append(e)
    monkeypatch.delenv("AO_COMPACT_ALWAYS", raising=False)
    runs = [HostEvalRunner(e) for e in engs]
    ags = [oracle.Agent(B, S, 5, noise=True, evaluator="stub1") for _ in range(G)]
    for g in range(G):
        ags[g].seed(seeds[g])
    roots = [(0,)] * G
    ev = lambda g, sim, pl: oracle.stub_eval(pl, 1)   # noqa: E731
    alive = np.ones(G, bool)
    for t in range(10):
        outs = [r.move(ev, tau=np.ones(G, np.int8), active=alive.astype(np.uint8)) for r in runs]
        for a, b in zip(outs[0], outs[1]):
            np.testing.assert_array_equal(a, b, err_msg="ply %d" % t)
        for g in np.nonzero(alive)[0]:
            assert engs[0].tree_nodes(g) == engs[1].tree_nodes(g)
            c0, c1 = engs[0].root_children(g), engs[1].root_children(g)
            for k in c0:
                np.testing.assert_array_equal(c0[k], c1[k])
        plays = [e.play() for e in engs]
        np.testing.assert_array_equal(plays[0][0][alive], plays[1][0][alive])
        np.testing.assert_array_equal(plays[0][1][alive], plays[1][1][alive])
        assert engs[0].trim_stats() == engs[1].trim_stats()
        if with_oracle:
            pi, vis, pol = outs[0]
            for g in np.nonzero(alive)[0]:
                opi, ovis, opol = ags[g].get_pi(roots[g], 1)
                np.testing.assert_array_equal(vis[g], ovis, err_msg="game %d ply %d" % (g, t))
                assert plays[0][0][g] == ags[g].rng.choice_p(opi)
                roots[g] = roots[g] + (int(plays[0][0][g]),)
        alive &= plays[0][1] == 0
        if not alive.any():
            break
    dropped, trimmed = engs[0].trim_stats()
    assert trimmed > 0 or cap_extra != 1
    assert not with_oracle or trimmed == 0
    for e in engs:
        e.close()


def test_illegal_root_id_is_rejected():
    from alpha_omok_amd.engine import Engine, EngineError
    eng = Engine(9, 10, 5, games=1)
    with pytest.raises(EngineError, match="illegal move"):
        eng.set_root(0, [3, 4, 3])          # cell 3 twice
    with pytest.raises(EngineError, match="illegal move"):
        eng.set_root(0, [3, 81])            # outside the board
    assert eng.set_root(0, [3, 4]) == 0     # still usable afterwards (fresh tree)
    eng.close()


def test_protocol_misuse_is_rejected():
    from alpha_omok_amd.engine import Engine, EngineError
    eng = Engine(3, 5, 5, games=1)
    with pytest.raises(EngineError):
        eng.play()                          # ao_play without ao_end_move
    with pytest.raises(EngineError):
        eng.collect_leaves(None)            # outside begin/end
    with pytest.raises(EngineError):
        Engine(9, 10, 4, games=1)           # even IN_PLANES
    with pytest.raises(EngineError):
        Engine(16, 10, 5, games=1)          # board too large
    eng.close()
