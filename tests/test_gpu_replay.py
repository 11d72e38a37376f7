"""-m gpu: device-resident replay memory (alpha_omok_amd.replay, csrc/replay.hip) against the
reference's own data flow: deque(maxlen) + utils.augment_dataset (golden gv8 captured from the
reference) + the mini-batches and losses of main.train."""
import random
from collections import deque

import numpy as np
import pytest

import pvnet_weights
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _samples(rs, n, B, C=5):
    out = []
    for _ in range(n):
        s = (rs.rand(C, B, B) < 0.3).astype(np.float64)
        pi = rs.dirichlet(np.ones(B * B))
        out.append((s, pi, float(rs.choice([-1.0, 0.0, 1.0]))))
    return out


def _same(a, b):
    assert len(a) == len(b)
    for (s0, p0, z0), (s1, p1, z1) in zip(a, b):
        np.testing.assert_array_equal(s0, s1)
        np.testing.assert_array_equal(p0, p1)      # float64, bit for bit
        assert z0 == z1


def test_augment_on_device_matches_reference_golden():
    """extend_augmented == the reference's augment_dataset output, entry by entry (gv8 fixture)."""
    from alpha_omok_amd.replay import DeviceReplay
    g = load_golden("gv8_augment")
    for i, B in enumerate((3, 9)):
        mem = DeviceReplay(B, 5, 64)
        mem.extend_augmented([(g["s%d" % i], g["pi%d" % i], 1.0)])
        assert len(mem) == 8
        s, pi, z = mem.read(0, 8)
        # the fixture's states are arbitrary float64 (every cell distinct, so a wrong permutation cannot
        # pass); the memory holds planes as float32 -- exact for the engine's 0/1 planes
        np.testing.assert_array_equal(s, g["as%d" % i].astype(np.float32).astype(np.float64))
        np.testing.assert_array_equal(pi, g["api%d" % i])
        assert (z == 1.0).all()
        mem.close()


@pytest.mark.parametrize("board,cap", [(9, 100), (15, 37), (3, 8), (9, 30000)])
def test_ring_matches_deque_through_wraps(board, cap):
    """Several extends, some larger than the capacity: contents and order == deque(maxlen).extend(augment(...))."""
    from alpha_omok_amd import utils
    from alpha_omok_amd.replay import DeviceReplay
    rs = np.random.RandomState(board * 1000 + cap)
    mem = DeviceReplay(board, 5, cap)
    ref = deque(maxlen=cap)
    assert mem.maxlen == cap and len(mem) == 0
    for n in (3, 1, 7, 0, 11, 2):
        smp = _samples(rs, n, board)
        mem.extend_augmented(smp)
        ref.extend(utils.augment_dataset(smp, board))
        assert len(mem) == len(ref)
        if cap <= 200:
            _same(list(mem), list(ref))
    _same(list(mem)[-40:], list(ref)[-40:])
    _same([mem[0], mem[-1], mem[len(mem) // 2]], [ref[0], ref[-1], ref[len(ref) // 2]])
    plain = _samples(rs, 5, board)
    mem.extend(plain)
    ref.extend(plain)
    _same(list(mem)[-12:], list(ref)[-12:])
    mem.clear()
    assert len(mem) == 0
    with pytest.raises(IndexError):
        mem[0]
    mem.close()


@pytest.mark.parametrize("board,cap,n", [(9, 100, 40), (9, 96, 12), (9, 97, 13), (3, 30000, 5000), (15, 37, 9), (9, 1000, 20)])
def test_extend_augmented_arrays_uploads_only_what_survives(board, cap, n):
    """main.self_play hands its samples over as arrays; when they alone fill the memory only the newest are uploaded
    (ao_replay_extend_skip) -- contents, order and length == deque(maxlen).extend(augment(all of them)), also on top
    of older entries and followed by more."""
    from alpha_omok_amd import utils
    from alpha_omok_amd.replay import DeviceReplay
    rs = np.random.RandomState(cap + n)
    mem = DeviceReplay(board, 5, cap)
    ref = deque(maxlen=cap)
    for k in (3, n, 2, n):
        smp = _samples(rs, k, board)
        mem.extend_augmented_arrays(np.stack([m[0] for m in smp]), np.stack([m[1] for m in smp]), np.array([m[2] for m in smp]))
        ref.extend(utils.augment_dataset(smp, board))
        assert len(mem) == len(ref)
        _same(list(mem)[-50:], list(ref)[-50:])
        _same(list(mem)[:20], list(ref)[:20])
    mem.close()


@pytest.mark.parametrize("board,C,cap,E", [(9, 17, 100000, 40), (15, 17, 5000, 12), (9, 5, 300, 30), (3, 5, 64, 6), (10, 3, 4096, 25),
                                           (9, 1, 900, 7)])
def test_states_built_on_the_device_from_move_lists(board, C, cap, E):
    """ao_replay_extend_moves (device-side sample emission): the planes a kernel builds from the episodes' moves == the planes
    of utils.get_state_pt (utils.py:139-168) uploaded through extend_augmented_arrays -- same ring contents, order and length,
    also when the call alone overfills the memory (only the surviving samples are staged) and on top of older entries."""
    from alpha_omok_amd import utils
    from alpha_omok_amd.replay import DeviceReplay, ReplayError
    rs = np.random.RandomState(board * 100 + C)
    A = board * board
    mem_d, mem_h = DeviceReplay(board, C, cap), DeviceReplay(board, C, cap)
    for rnd in range(3):
        lens = rs.randint(1, A + 1, E)
        lens[0] = A                                        # a board played full
        moves = np.full((E, A), -1, np.int32)
        for e in range(E):
            moves[e, :lens[e]] = rs.permutation(A)[:lens[e]]
        ep_of = np.repeat(np.arange(E), lens)
        ply_of = np.concatenate([np.arange(l) for l in lens])
        if rnd == 1:                                       # samples need not be sorted, nor cover every ply
            pick = rs.permutation(ep_of.size)[:max(1, ep_of.size // 3)]
            ep_of, ply_of = ep_of[pick], ply_of[pick]
        n = ep_of.size
        pis = rs.dirichlet(np.ones(A), n)
        z = rs.choice([-1.0, 0.0, 1.0], n)
        states = utils.states_of_episodes(moves, ep_of, ply_of, board, C)
        for i in rs.randint(0, n, 5):                      # the vectorised host builder against the per-sample restatement
            node = (0,) + tuple(int(m) for m in moves[ep_of[i], :ply_of[i]])
            np.testing.assert_array_equal(states[i], utils.get_state_pt(node, board, C))
        mem_h.extend_augmented_arrays(states, pis, z)
        mem_d.extend_augmented_moves(moves, ep_of, ply_of, pis, z)
        assert len(mem_d) == len(mem_h) == min(cap, len(mem_h))
        m = len(mem_h)
        sd, pd, zd = mem_d.read(0, m)
        sh, ph, zh = mem_h.read(0, m)
        np.testing.assert_array_equal(sd, sh)
        np.testing.assert_array_equal(pd, ph)
        np.testing.assert_array_equal(zd, zh)
    with pytest.raises(ReplayError):
        mem_d.extend_augmented_moves(moves, np.array([E]), np.array([0]), pis[:1], z[:1])          # no such episode
    with pytest.raises(ReplayError):
        mem_d.extend_augmented_moves(moves, np.array([0]), np.array([A + 1]), pis[:1], z[:1])      # more plies than moves
    mem_d.close(); mem_h.close()


def test_batches_match_host_assembly():
    """batch(indices) == torch.tensor(np.stack(...)).float() of the same deque entries (main.py:283-290)."""
    import torch
    from alpha_omok_amd import utils
    from alpha_omok_amd.replay import DeviceReplay, ReplayError
    rs = np.random.RandomState(5)
    smp = _samples(rs, 20, 9)
    mem = DeviceReplay(9, 5, 120)
    mem.extend_augmented(smp)
    ref = deque(utils.augment_dataset(smp, 9), maxlen=120)
    random.seed(3)
    idx = random.sample(range(len(mem)), 64)
    random.seed(3)
    picked = random.sample(list(ref), 64)
    s, pi, z = mem.batch(idx)
    assert s.is_cuda and s.dtype == torch.float32 and tuple(s.shape) == (64, 5, 9, 9)
    np.testing.assert_array_equal(s.cpu().numpy(), np.stack([b[0] for b in picked]).astype(np.float32))
    np.testing.assert_array_equal(pi.cpu().numpy(), np.stack([b[1] for b in picked]).astype(np.float32))
    np.testing.assert_array_equal(z.cpu().numpy(), np.array([b[2] for b in picked], np.float32))
    with pytest.raises(ReplayError):
        mem.batch([len(mem)])
    with pytest.raises(ReplayError):
        mem.extend([(np.zeros((5, 7, 7)), smp[0][1], 0.0)])         # wrong board
    mem.close()


def test_train_with_device_replay_equals_host_path(oracle):
    """main.self_play + main.train with rep_memory in HBM: same memory contents, same mini-batches,
    same losses as the host deque path under the same seeds."""
    import torch
    import alpha_omok_amd.main as main
    from test_gpu_dropin import StubModel
    from alpha_omok_amd.pvnet import PVNet
    B = 9
    results = []
    for dev_replay in (False, True):
        main.PRINT_SELFPLAY = False
        main.configure(board_size=B, n_mcts=24, n_blocks=1, out_planes=32, seed=0, device_replay=dev_replay)
        net = PVNet(1, 5, 32, B)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(1, 5, 32, B, 2).items()})
        main.Agent.model = StubModel(oracle, 1)
        main.cur_memory.clear()
        main.rep_memory.clear()
        main.reset_iter(main.result, main.cur_memory)
        main.self_play(3, seeds=[5, 6, 7])
        mem = [(s.copy(), p.copy(), z) for s, p, z in main.rep_memory]
        main.Agent.model = net.to(main.device)
        main.optimizer = torch.optim.Adam(main.Agent.model.parameters(), lr=main.LR, weight_decay=main.L2, eps=1e-6)
        random.seed(11)
        torch.manual_seed(11)
        n_keep = len(main.cur_memory)
        while len(main.cur_memory) > 2:                # keeps the training pass short: 2 * 32 samples
            main.cur_memory.pop()
        losses = main.train(1, 0)
        results.append((mem, losses, n_keep))
    (mem_h, loss_h, n_h), (mem_d, loss_d, n_d) = results
    assert n_h == n_d and len(mem_h) == 8 * n_h
    _same(mem_h, mem_d)
    assert len(loss_h) == len(loss_d) == 2
    np.testing.assert_allclose(np.array(loss_h), np.array(loss_d), rtol=0, atol=2e-5)
    main.configure(board_size=B, n_mcts=24, n_blocks=1, out_planes=32, seed=0)   # back to the host deque
    assert isinstance(main.rep_memory, deque)
