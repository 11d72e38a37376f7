"""-m gpu: many full games with slot refill on the native network -- invariants that must hold for
every game played to the end (win detection == host check_win, draws, sample bookkeeping)."""
import numpy as np
import pytest

import pvnet_weights

pytestmark = pytest.mark.gpu


def test_full_games_native_net_invariants():
    import torch
    import alpha_omok_amd.main as main
    from alpha_omok_amd import utils
    from alpha_omok_amd.pvnet import PVNet
    B, S, n = 9, 24, 300
    model = PVNet(1, 5, 32, B)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(1, 5, 32, B, 5).items()})
    main.MAX_CONCURRENT = 128                         # 300 episodes through 128 slots: two refill waves
    main.configure(board_size=B, n_mcts=S, model=model.cuda().eval(), seed=7)
    main.cur_memory.clear()
    main.rep_memory.clear()
    main.reset_iter(main.result, main.cur_memory)
    main.self_play(n)
    main.MAX_CONCURRENT = 4096
    cm = list(main.cur_memory)
    assert sum(main.result.values()) == n
    # split the flat memory back into episodes: an episode starts at the empty board
    starts = [i for i, m in enumerate(cm) if m[0][:4].sum() == 0 and m[0][4].min() == 1]
    assert len(starts) == n
    starts.append(len(cm))
    tally = {1: 0, 2: 0, 3: 0}
    for e in range(n):
        ep = cm[starts[e]:starts[e + 1]]
        board = np.zeros((B, B))
        for t, (s, pi, z) in enumerate(ep):
            assert abs(pi.sum() - 1) < 1e-9 and (pi >= 0).all()
            # the stored state is the position before the move: rebuild it from the planes
            own, opp = s[2], s[3]
            assert (s[4] == (1.0 if t % 2 == 0 else 0.0)).all()
            black, white = (own, opp) if t % 2 == 0 else (opp, own)
            np.testing.assert_array_equal(black - white, board)
            assert utils.check_win(board, 5) == 0           # never searched from a finished position
            if t + 1 < len(ep):
                nxt = ep[t + 1][0]
                nb, nw_ = (nxt[3], nxt[2]) if t % 2 == 0 else (nxt[2], nxt[3])
                new_board = nb - nw_
                diff = new_board - board
                assert np.count_nonzero(diff) == 1 and diff.sum() == (1 if t % 2 == 0 else -1)
                a = int(np.flatnonzero(diff.reshape(-1))[0])
                assert pi[a] > 0                           # the played move had visits
                board = new_board
        # outcome: z of the last mover tells who won; it must be consistent along the episode
        zs = np.array([m[2] for m in ep])
        assert set(np.abs(zs)) <= {0.0, 1.0}
        assert (zs[0::2] == zs[0]).all() and (zs[1::2] == -zs[0]).all()
        win = 3 if zs[0] == 0 else (1 if zs[0] > 0 else 2)
        tally[win] += 1
        assert len(ep) >= 9 and len(ep) <= B * B
        if win == 3:
            assert len(ep) == B * B
    assert [main.result[k] for k in ("Black", "White", "Draw")] == [tally[1], tally[2], tally[3]]
    assert len(main.rep_memory) == min(8 * len(cm), main.MEMORY_SIZE)
