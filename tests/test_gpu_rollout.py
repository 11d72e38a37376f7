"""-m gpu: rollout agents (csrc/rollout.hip behind ao_rollout_*; agents.PUCTAgent / UCTAgent) against
the reference's own outputs (tests/golden/gv11, captured from agents.py:263-614) and against the
oracle restatement (oracle/rollout_oracle.c) on many seeded positions."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_dropin_agents_reproduce_reference_under_numpy_seed():
    """np.random.seed(s); agent.get_pi(root_id, board, turn, tau) -- the reference's call -- returns the
    reference's one-hot and leaves np.random where the reference leaves it (gv11)."""
    from alpha_omok_amd import agents, utils
    agents.PRINT_MCTS = False
    g = load_golden("gv11_rollout_agents")
    for ci in range(int(g["ncases"])):
        mode, B, S, seed, nrec = g["c%d_cfg" % ci].tolist()
        agent = (agents.PUCTAgent if mode == 0 else agents.UCTAgent)(B, S)
        np.random.seed(seed)
        root = (0,) + tuple(int(a) for a in g["c%d_start" % ci])
        for t in range(nrec):
            pi = agent.get_pi(root, utils.get_board(root, B), utils.get_turn(root), 0)
            np.testing.assert_array_equal(pi, g["c%d_pi" % ci][t], err_msg="case %d call %d" % (ci, t))
            if mode == 0:
                np.testing.assert_array_equal(agent.get_visit(), g["c%d_stat" % ci][t])
            assert np.random.get_state()[2] == int(g["c%d_pos" % ci][t])
            root = root + (int(np.argmax(pi)),)
        assert agent.get_name() == ("PUCTAgent" if mode == 0 else "UCTAgent")
        agent.reset()
        assert agent.root_id is None and len(agent.tree) == 0


@pytest.mark.parametrize("mode,board,sims,games", [(0, 9, 120, 24), (1, 9, 120, 24), (0, 15, 40, 12), (1, 15, 40, 12),
                                                   (0, 3, 300, 16), (1, 3, 300, 16), (0, 7, 64, 8), (1, 7, 64, 8)])
def test_many_games_match_oracle(oracle, mode, board, sims, games):
    """G concurrent searches from different positions with different seeds == G oracle searches: child visit
    counts / q, chosen move, stream position -- two consecutive moves per game."""
    from alpha_omok_amd.rollout import RolloutEngine
    rs = np.random.RandomState(100 * board + mode)
    A = board * board
    eng = RolloutEngine(board, sims, mode, games=games)
    roots, rngs = [], []
    for g in range(games):
        for _ in range(100):
            k = int(rs.randint(0, max(1, A - 2))) if board > 3 else int(rs.randint(0, 4))
            mv = tuple(int(a) for a in rs.permutation(A)[:k])
            if oracle.check_win(oracle.get_board(list(mv), board), 3 if board == 3 else 5) == 0:
                break
        roots.append((0,) + mv)
        eng.seed(g, 1000 + g)
        rngs.append(oracle.Rng(1000 + g))
    active = np.ones(games, np.uint8)
    for step in range(2):
        pi, stat, act = eng.search(roots, active=active)
        for g in range(games):
            if not active[g]:
                continue
            opi, ostat, oact, _ = oracle.rollout_search(mode, board, sims, roots[g], rngs[g])
            np.testing.assert_array_equal(pi[g], opi, err_msg="game %d step %d" % (g, step))
            np.testing.assert_array_equal(stat[g], ostat, err_msg="game %d step %d" % (g, step))
            assert act[g] == oact
            assert eng.get_rng_state(g)[1] == rngs[g].pos
            roots[g] = roots[g] + (int(oact),)
            mv = list(roots[g])[1:]
            if len(mv) >= A - 1 or oracle.check_win(oracle.get_board(mv, board), 3 if board == 3 else 5) != 0:
                active[g] = 0
    eng.close()


def test_rollout_errors():
    from alpha_omok_amd.rollout import RolloutEngine, RolloutError
    eng = RolloutEngine(9, 10, 0, games=1)
    with pytest.raises(RolloutError):
        eng.search([(0, 5, 5)])                      # occupied cell
    with pytest.raises(RolloutError):
        eng.search([(0, 0, 9, 1, 10, 2, 11, 3, 12, 4)])   # black already has five in a row
    with pytest.raises(RolloutError):
        RolloutEngine(9, 10, 2)
    pi, stat, act = eng.search([(0, 40)])
    assert pi.sum() == 1 and stat[0, 40] == 0 and stat.sum() == 10
    eng.close()


def test_tictactoe_uct_dropin_matches_reference_golden():
    """BASELINE configs[0]: random.seed(s); uct_search(board, turn, num_mcts) == the reference's per-move search
    (mcts_vs.py:153-183): max_action, q of every root child, and the `random` state it leaves behind (gv12)."""
    import random
    from alpha_omok_amd.tictactoe import uct_search
    g = load_golden("gv12_tictactoe_uct")
    for ci in range(int(g["ncases"])):
        turn, num_mcts, seed, max_action, pos = g["c%d_cfg" % ci].tolist()
        random.seed(seed)
        a, q_list = uct_search(g["c%d_board" % ci].astype(np.float64), turn, num_mcts)
        assert a == max_action, "case %d" % ci
        gq = g["c%d_q" % ci]
        assert sorted(q_list) == [(0, i) for i in range(9) if gq[i] != -np.inf]
        for (_, i), v in q_list.items():
            assert v == gq[i], "case %d child %d" % (ci, i)
        st = random.getstate()[1]
        assert st[624] == pos
        np.testing.assert_array_equal(np.array(st[:624], np.uint32), g["c%d_mt" % ci])


@pytest.mark.parametrize("board,sims,games", [(3, 1500, 64), (3, 200, 256), (5, 120, 16), (9, 60, 8)])
def test_tictactoe_uct_many_boards_match_oracle(oracle, board, sims, games):
    from alpha_omok_amd.tictactoe import TttEngine
    rs = np.random.RandomState(board * 7 + sims)
    wm = 3 if board == 3 else 5
    eng = TttEngine(sims, games=games, board_size=board)
    boards = np.zeros((games, board, board), np.int8)
    turns = np.zeros(games, np.int32)
    for g in range(games):
        for _ in range(200):
            k = int(rs.randint(0, board * board - 1))
            cells = rs.permutation(board * board)[:k]
            b = np.zeros(board * board, np.int8)
            b[cells[0::2]] = 1
            b[cells[1::2]] = -1
            if oracle.check_win(b.reshape(board, board), wm) == 0:
                break
        boards[g] = b.reshape(board, board)
        turns[g] = k % 2
        eng.seed(g, 500 + g)
    act, q, n = eng.search(boards, turns)
    for g in range(games):
        rng = oracle.PyRandom(500 + g)
        oa, oq, on = oracle.ttt_search(boards[g], int(turns[g]), sims, rng, win_mark=wm)
        assert act[g] == oa, "game %d" % g
        np.testing.assert_array_equal(q[g], oq, err_msg="game %d" % g)
        np.testing.assert_array_equal(n[g], on, err_msg="game %d" % g)
        mt, pos = eng.get_rng_state(g)
        assert pos == rng.pos
        np.testing.assert_array_equal(mt, rng.state_words())
    eng.close()
