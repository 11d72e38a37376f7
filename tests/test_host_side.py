"""CPU: host-side mirror of the reference interface (alpha_omok_amd.utils / pvnet / parallel /
main.train) against the golden vectors, plus the multi-process (gloo, world_size 2) gradient
all-reduce path."""
import os
import random
import sys

import numpy as np
import pytest

import pvnet_weights
from conftest import REPO, load_golden


def test_utils_check_win_golden():
    from alpha_omok_amd import utils
    g = load_golden("gv1_check_win")
    for b, (n, k), w in zip(g["boards"], g["size_mark"], g["win"]):
        assert utils.check_win(b[:n, :n].astype(float), int(k)) == int(w)


def test_utils_planes_board_turn_golden():
    from alpha_omok_amd import utils
    g = load_golden("gv3_state_planes")
    for i in range(int(g["count"])):
        m = g["m%d" % i]
        B, C, nid = int(m[0]), int(m[1]), (0,) + tuple(int(x) for x in m[2:])
        s = utils.get_state_pt(nid, B, C)
        assert s.dtype == np.float64 and s.shape == (C, B, B)
        np.testing.assert_array_equal(s.astype(np.float32), g["s%d" % i])
        np.testing.assert_array_equal(utils.get_board(nid, B).astype(np.int8), g["b%d" % i])
        assert utils.get_turn(nid) == int(g["t%d" % i])


def test_states_of_episodes_equals_get_state_pt_per_sample():
    """main.self_play builds the states of all samples of a call at once (utils.states_of_episodes, episodes in chunks):
    entry for entry get_state_pt (utils.py:139-168), for samples in any order, across chunk boundaries, empty input."""
    from alpha_omok_amd import utils
    rng = np.random.default_rng(0)
    B, A, E = 9, 81, 300
    moves = np.full((E, A), -1, np.int32)
    lengths = rng.integers(1, A + 1, E)
    for e in range(E):
        moves[e, :lengths[e]] = rng.permutation(A)[:lengths[e]]
    ep = np.concatenate([np.full(lengths[e], e) for e in range(E)])
    pl = np.concatenate([np.arange(lengths[e]) for e in range(E)])
    pick = rng.permutation(ep.size)[:1200]
    for chunk in (7, 256):
        out = utils.states_of_episodes(moves, ep[pick], pl[pick], B, 5, chunk=chunk)
        assert out.shape == (pick.size, 5, B, B) and out.dtype == np.float64
        for k, i in enumerate(pick):
            rid = (0,) + tuple(int(x) for x in moves[ep[i], :pl[i]])
            assert np.array_equal(out[k], utils.get_state_pt(rid, B, 5)), (chunk, k)
    assert utils.states_of_episodes(moves, ep[:0], pl[:0], B, 5).shape == (0, 5, B, B)


def test_utils_legal_order_golden():
    from alpha_omok_amd import utils
    g = load_golden("gv2_legal_order")
    for B, mv, order in list(zip(g["board"], g["moves"], g["order"]))[::7]:
        nid = (0,) + tuple(int(x) for x in mv[mv >= 0])
        assert utils.legal_actions(nid, int(B)) == order[order >= 0].tolist()


def test_utils_augment_golden():
    from alpha_omok_amd import utils
    g = load_golden("gv8_augment")
    for i, B in enumerate((3, 9)):
        aug = utils.augment_dataset([(g["s%d" % i], g["pi%d" % i], 1.0)], B)
        assert len(aug) == 8
        np.testing.assert_array_equal(np.stack([a[0] for a in aug]), g["as%d" % i])
        np.testing.assert_array_equal(np.stack([a[1] for a in aug]), g["api%d" % i])


def test_utils_sampling_uses_numpy_stream_like_reference(oracle):
    from alpha_omok_amd import utils
    rs = np.random.RandomState(4)
    for seed in (0, 5, 99):
        pi = rs.dirichlet(np.ones(81))
        np.random.seed(seed)
        r = oracle.Rng(seed)
        _, a = utils.get_action(pi)
        assert a == r.choice_p(pi)
        vis = rs.randint(0, 4, 81).astype(float)
        _, b = utils.argmax_onehot(vis / vis.sum())
        k = int((vis == vis.max()).sum())
        assert b == np.flatnonzero(vis == vis.max())[r.choice(k)]
        assert np.random.get_state()[2] == r.pos


def test_pvnet_state_dict_wire_format_and_forward_golden():
    import torch
    from alpha_omok_amd.pvnet import PVNet, looks_like_pvnet
    g = load_golden("gv7_pvnet_forward")
    for i in range(int(g["count"])):
        nb, B, planes, wseed = g["cfg%d" % i].tolist()
        sd = pvnet_weights.make_state_dict(nb, 5, planes, B, wseed)
        net = PVNet(nb, 5, planes, B)
        assert set(net.state_dict().keys()) == set(sd.keys())
        for k, v in net.state_dict().items():
            assert tuple(v.shape) == sd[k].shape, k
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net.eval()
        assert looks_like_pvnet(net) == (nb, 5, planes, B)
        with torch.no_grad():
            p, v = net(torch.from_numpy(g["x%d" % i]))
        assert np.abs(p.numpy() - g["p%d" % i]).max() < 1e-5
        assert np.abs(v.numpy() - g["v%d" % i]).max() < 1e-5


def test_train_step_golden():
    """main.train's loss and Adam step (main.py:85,294-305) against the reference's own numbers."""
    import torch
    from alpha_omok_amd.pvnet import PVNet
    g = load_golden("gv10_train_step")
    nb, B, planes, wseed = g["cfg"].tolist()
    net = PVNet(nb, 5, planes, B)
    sd = pvnet_weights.make_state_dict(nb, 5, planes, B, wseed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, weight_decay=0, eps=1e-6)
    s, pi, z = (torch.from_numpy(g[k]) for k in ("s", "pi", "z"))
    p, v = net(s)
    v_loss = (v - z).pow(2).mean()
    p_loss = -(pi * p.log()).sum(dim=-1).mean()
    assert abs(v_loss.item() - float(g["v_loss"])) < 1e-5
    assert abs(p_loss.item() - float(g["p_loss"])) < 1e-5
    opt.zero_grad()
    (v_loss + p_loss).backward()
    grads = dict(net.named_parameters())
    for key in g.files:
        if key.startswith("grad__"):
            assert np.abs(grads[key[6:]].grad.numpy() - g[key]).max() < 1e-5, key
    opt.step()
    after = net.state_dict()
    for key in g.files:
        if key.startswith("after__"):
            assert np.abs(after[key[7:]].numpy() - g[key]).max() < 1e-5, key


def _ddp_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import pvnet_weights as pw
    from alpha_omok_amd import parallel
    from alpha_omok_amd.pvnet import PVNet
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    torch.manual_seed(100 + rank)                      # deliberately different initial weights
    net = PVNet(1, 5, 32, 9)
    net.eval()                                         # BN in eval: grads add up across shards
    ref_sd = {k: torch.from_numpy(v) for k, v in pw.make_state_dict(1, 5, 32, 9, 3).items()}
    if rank == 0:
        net.load_state_dict(ref_sd)
    parallel.broadcast_parameters(net)                 # rank 0's weights everywhere
    same = all(torch.equal(net.state_dict()[k], ref_sd[k]) for k in ref_sd if ref_sd[k].is_floating_point())
    rs = np.random.RandomState(7)
    x = torch.from_numpy((rs.rand(8, 5, 9, 9) < 0.3).astype(np.float32))
    pi = torch.from_numpy(rs.dirichlet(np.ones(81), size=8).astype(np.float32))
    z = torch.from_numpy(rs.choice([-1.0, 0.0, 1.0], size=8).astype(np.float32))
    sl = slice(rank * 4, rank * 4 + 4)
    p, v = net(x[sl])
    loss = (v - z[sl]).pow(2).mean() - (pi[sl] * p.log()).sum(-1).mean()
    loss.backward()
    n, contributors = parallel.allreduce_gradients(net)
    assert contributors == world
    flat = torch.cat([q.grad.reshape(-1) for q in net.parameters()])
    # unequal shards (5 + 3 samples): each rank's mean gradient counts for the samples behind it
    net.zero_grad()
    su = slice(0, 5) if rank == 0 else slice(5, 8)
    p, v = net(x[su])
    ((v - z[su]).pow(2).mean() - (pi[su] * p.log()).sum(-1).mean()).backward()
    n2, wsum = parallel.allreduce_gradients(net, weight=su.stop - su.start)
    assert wsum == 8 and n2 == n
    flat_w = torch.cat([q.grad.reshape(-1) for q in net.parameters()])
    torch.save(dict(same=same, n=n, grad=flat, grad_w=flat_w, shard=parallel.shard_games(10, rank, world)), out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce(tmp_path):
    """N>1 path on CPU: mean of the per-rank gradients == gradient of the full batch; broadcast makes
    the weights identical; games shard g % world."""
    import torch
    import torch.multiprocessing as mp
    from alpha_omok_amd.pvnet import PVNet
    port = 29500 + random.randint(0, 2000)
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0["same"] and r1["same"]
    assert torch.equal(r0["grad"], r1["grad"])
    assert r0["shard"] == [0, 2, 4, 6, 8] and r1["shard"] == [1, 3, 5, 7, 9]
    net = PVNet(1, 5, 32, 9)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in pvnet_weights.make_state_dict(1, 5, 32, 9, 3).items()})
    net.eval()
    rs = np.random.RandomState(7)
    x = torch.from_numpy((rs.rand(8, 5, 9, 9) < 0.3).astype(np.float32))
    pi = torch.from_numpy(rs.dirichlet(np.ones(81), size=8).astype(np.float32))
    z = torch.from_numpy(rs.choice([-1.0, 0.0, 1.0], size=8).astype(np.float32))
    p, v = net(x)
    ((v - z).pow(2).mean() - (pi * p.log()).sum(-1).mean()).backward()
    full = torch.cat([q.grad.reshape(-1) for q in net.parameters()])
    assert r0["n"] == full.numel()
    assert (r0["grad"] - full).abs().max().item() < 1e-5
    # 5 + 3 samples, weighted by their counts: still the gradient of the mean loss over all 8
    assert torch.equal(r0["grad_w"], r1["grad_w"])
    assert (r0["grad_w"] - full).abs().max().item() < 1e-5


def _train_worker(rank, world, port, out, n_cur, n_rep):
    """main.train under torch.distributed with rank-local memories of DIFFERENT sizes (the round-1
    deadlock: the mini-batch count was rank-local). n_cur / n_rep: per-rank len(cur_memory) /
    len(rep_memory)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from alpha_omok_amd import main, parallel
    from alpha_omok_amd.pvnet import PVNet
    parallel.init_from_env("gloo")
    torch.manual_seed(50 + rank)                       # different initial weights: configure() must broadcast
    main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9))
    rs = np.random.RandomState(100 + rank)
    random.seed(200 + rank)                            # every rank draws its own batches
    def sample():
        return ((rs.rand(5, 9, 9) < 0.3).astype(np.float64), rs.dirichlet(np.ones(81)), float(rs.choice([-1, 0, 1])))
    main.cur_memory.clear(); main.rep_memory.clear()
    main.cur_memory.extend(sample() for _ in range(n_cur[rank]))
    main.rep_memory.extend(sample() for _ in range(n_rep[rank]))
    main.step = 0
    losses = main.train(1, 1)
    sd = {k: v.clone() for k, v in main.Agent.model.state_dict().items()}
    torch.save(dict(sd=sd, step=main.step, n_losses=len(losses)), out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_cur,n_rep,steps,n_losses", [
    ((3, 5), (400, 400), 4, (4, 4)),        # unequal new samples: ceil(8 / 2) steps on both ranks
    ((6, 0), (150, 0), 3, (3, 0)),          # rank 1 holds nothing at all (run() with fewer games than ranks)
    ((4, 4), (70, 400), 4, (3, 4)),         # rank 0's shard runs out after 2 full + 1 partial batch
])
def test_gloo_world2_train_with_unequal_memories(tmp_path, n_cur, n_rep, steps, n_losses):
    """Every rank issues the same number of all-reduces whatever its local sample counts are, the
    optimiser steps are identical and the state_dict (BatchNorm buffers included) ends bit-identical."""
    import torch
    import torch.multiprocessing as mp
    port = 29500 + random.randint(0, 2000)
    out = str(tmp_path / "t%d.pt")
    mp.spawn(_train_worker, args=(2, port, out, n_cur, n_rep), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0["step"] == r1["step"] == steps
    assert (r0["n_losses"], r1["n_losses"]) == n_losses
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    assert int(r0["sd"]["bn1.num_batches_tracked"]) == max(n_losses)


def _shard_worker(rank, world, port, directory, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from alpha_omok_amd import main, parallel
    parallel.init_from_env("gloo")
    main.rep_memory = __import__("collections").deque(maxlen=main.MEMORY_SIZE)
    mine = [(np.full((5, 9, 9), float(rank)), np.ones(81) / 81, float(100 * rank + i)) for i in range(4 + rank)]
    main.rep_memory.extend(mine)
    path = main.save_dataset(main.rep_memory, 300, 77, directory=directory, datetime_now="180927")
    dist.barrier()
    main.rep_memory.clear()
    main.load_data(None, os.path.join(directory, "180927_300_77_step_dataset.pickle"))
    back = [m[2] for m in main.rep_memory]
    torch.save(dict(path=path, mine=[m[2] for m in mine], back=back), out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_dataset_shards_survive_save_and_resume(tmp_path):
    """Replay memories are rank-local: every rank writes its shard (rank 0 under the reference's file name), a resume
    of the same shape gives every rank its own samples back, and a single-process resume pools all of them."""
    import torch
    import torch.multiprocessing as mp
    from collections import deque
    from alpha_omok_amd import main
    port = 29500 + random.randint(0, 2000)
    out = str(tmp_path / "s%d.pt")
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert os.path.basename(r0["path"]) == "180927_300_77_step_dataset.pickle"
    assert os.path.basename(r1["path"]) == "180927_300_77_step_dataset.pickle.rank1of2"
    assert r0["back"] == r0["mine"] and r1["back"] == r1["mine"] and len(r1["mine"]) == 5
    main.rep_memory = deque(maxlen=main.MEMORY_SIZE)
    main.load_data(None, r0["path"])                   # one process: everything, nobody's samples lost
    assert sorted(m[2] for m in main.rep_memory) == sorted(r0["mine"] + r1["mine"])
    main.rep_memory.clear()


def test_train_raises_like_the_reference_when_replay_is_too_small():
    """main.py:263-264: random.sample(rep_memory, 32 * len(cur_memory)) raises ValueError."""
    from alpha_omok_amd import main
    from alpha_omok_amd.pvnet import PVNet
    main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9))
    smp = (np.zeros((5, 9, 9)), np.ones(81) / 81, 1.0)
    main.cur_memory.clear(); main.rep_memory.clear()
    main.cur_memory.extend([smp] * 2)
    main.rep_memory.extend([smp] * 63)
    with pytest.raises(ValueError):
        main.train(1, 1)
    main.cur_memory.clear(); main.rep_memory.clear()


def test_checkpoint_wire_format_roundtrip(tmp_path):
    """save_model / save_dataset / load_data (main.py:339-365): file naming carries iter and step;
    the model file is a plain state_dict with the reference's keys."""
    import torch
    import alpha_omok_amd.main as main
    from alpha_omok_amd.pvnet import PVNet

    class A:
        pass

    main.Agent = A()
    main.Agent.model = PVNet(1, 5, 32, 9)
    main.device = torch.device("cpu")
    main.rep_memory.clear()
    main.rep_memory.extend([(np.zeros((5, 9, 9)), np.ones(81) / 81, 1.0)] * 3)
    mp = main.save_model(main.Agent, 200, 1234, directory=str(tmp_path), datetime_now="180927")
    dp = main.save_dataset(main.rep_memory, 200, 1234, directory=str(tmp_path), datetime_now="180927")
    assert os.path.basename(mp) == "180927_200_1234_step_model.pickle"
    sd = torch.load(mp)
    assert set(sd.keys()) == set(pvnet_weights.make_state_dict(1, 5, 32, 9, 0).keys())
    fresh = PVNet(1, 5, 32, 9)
    main.Agent.model = fresh
    main.rep_memory.clear()
    main.load_data(mp, dp)
    assert main.step == 1234 and main.start_iter == 201
    assert len(main.rep_memory) == 3 and main.rep_memory.maxlen == main.MEMORY_SIZE
    for k, v in sd.items():
        assert torch.equal(fresh.state_dict()[k], v)


def test_carry_over_pool_numbers_episodes_across_calls_and_ranks():
    """main._CarryPool.take_next (carry-over self-play): a rank starts ITS episodes of the current call, then of the next calls,
    in global episode order, and stops at the look-ahead limit -- host logic only, no device."""
    from types import SimpleNamespace
    from alpha_omok_amd import main
    eng = SimpleNamespace(A=81, G=4)
    for rank, world, n_call, first in ((0, 1, 5, 0), (1, 2, 7, 14), (2, 3, 4, 8)):
        pool = main._CarryPool(eng, n_call, rank, world)
        pool.next_call = first
        limit = first + 3 * n_call
        got = []
        while True:
            gid = pool.take_next(limit)
            if gid < 0:
                break
            got.append(gid)
        want = [first + c * n_call + e for c in range(3) for e in range(rank, n_call, world)]
        assert got == want, (rank, world, n_call, got, want)
        assert pool.take_next(limit) == -1                      # stays stopped at the limit ...
        more = pool.take_next(limit + n_call)                   # ... and goes on when the limit moves (the next call)
        nxt = [first + 3 * n_call + e for e in range(rank, n_call, world)]
        assert more == (nxt[0] if nxt else -1)


def test_run_schedule_constants_scale_the_reference_loop(monkeypatch):
    """main.run is the reference's loop (main.py:377-414): N_SELFPLAY games, then ONE game + one training pass per iteration.
    GAMES_PER_ITER / TRAIN_STEPS (None = the reference) scale it to thousands of concurrent games: the games of the later
    iterations, and the mini-batches of a pass instead of one per new sample."""
    import random
    from alpha_omok_amd import main
    calls = []
    monkeypatch.setattr(main, "Agent", object())                     # (configured)
    monkeypatch.setattr(main, "self_play", lambda n: calls.append(("play", n)))
    monkeypatch.setattr(main, "train", lambda e, i: calls.append(("train", i)))
    monkeypatch.setattr(main, "load_data", lambda a, b: None)
    monkeypatch.setattr(main, "save_model", lambda *a, **k: None)
    monkeypatch.setattr(main, "save_dataset", lambda *a, **k: None)
    monkeypatch.setattr(main, "start_iter", 0)
    monkeypatch.setattr(main, "GAMES_PER_ITER", None)
    assert main.run(total_iter=3, n_selfplay=7) == 3
    assert calls == [("play", 7), ("play", 1), ("train", 1), ("play", 1), ("train", 2)]
    calls.clear()
    monkeypatch.setattr(main, "GAMES_PER_ITER", 2048)
    main.run(total_iter=2, n_selfplay=100)
    assert calls == [("play", 100), ("play", 2048), ("train", 1)]
    # TRAIN_STEPS replaces len(cur_memory) as the number of mini-batches of a pass
    import torch
    from alpha_omok_amd.pvnet import PVNet
    monkeypatch.undo()
    main.configure(board_size=3, n_mcts=4, n_blocks=1, out_planes=32, seed=1, gpu=0)
    rs = np.random.RandomState(0)
    main.rep_memory.clear(); main.cur_memory.clear()
    for _ in range(200):
        pi = rs.rand(9); pi /= pi.sum()
        main.rep_memory.append(((rs.rand(5, 3, 3) < 0.3).astype(np.float64), pi, float(rs.choice([-1.0, 0.0, 1.0]))))
    main.cur_memory.extend(list(main.rep_memory)[:3])
    random.seed(4)
    main.step = 0
    try:
        main.TRAIN_STEPS, main.BATCH_SIZE = 5, 16
        losses = main.train(1, 1)
        assert len(losses) == 5 and main.step == 5
        main.TRAIN_STEPS = None
        losses = main.train(1, 2)
        assert len(losses) == 3 and main.step == 8                   # the reference: one mini-batch per new sample
    finally:
        main.TRAIN_STEPS, main.BATCH_SIZE = None, 32
        main.rep_memory.clear(); main.cur_memory.clear()


def test_configure_oversubscribe_and_rows_host_logic():
    """Round 5 host logic of main.py, no GPU: configure(oversubscribe=, rows=) validation and persistence, the slot count of the
    self-play engine (MAX_CONCURRENT rows x OVERSUBSCRIBE, never more than there are episodes), the carry-over default (None =
    automatic inside run() with GAMES_PER_ITER) and the row mode `_set_rows` asks the engine for."""
    import alpha_omok_amd.main as main
    from alpha_omok_amd.pvnet import PVNet
    keep = (main.MAX_CONCURRENT, main.OVERSUBSCRIBE, main.ROWS, main.CARRY_OVER)
    try:
        main.MAX_CONCURRENT = 4096
        main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9), oversubscribe=1.25, rows='auto')
        assert main.OVERSUBSCRIBE == 1.25 and main.ROWS == 'auto'
        assert main._slots(10 ** 6) == 5120 and main._slots(3000) == 3000 and main._slots(1) == 1
        main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9))   # unchanged by a call that does not name them
        assert main.OVERSUBSCRIBE == 1.25
        for bad in (0.5, 2.5):
            with pytest.raises(ValueError):
                main.configure(board_size=9, n_blocks=1, out_planes=32, model=PVNet(1, 5, 32, 9), oversubscribe=bad)
        with pytest.raises(ValueError):
            main.configure(board_size=9, n_blocks=1, out_planes=32, model=PVNet(1, 5, 32, 9), rows='sometimes')

        class FakeEngine:
            def __init__(self, G):
                self.G, self.row_cap, self.calls = G, 0, []

            def set_row_cap(self, cap):
                self.row_cap = cap
                self.calls.append(cap)

        over, plain = FakeEngine(5120), FakeEngine(4096)
        main._terminal_share = 0.0
        main._set_rows(over)
        main._set_rows(plain)
        assert over.calls == [4096] and plain.calls == []            # over-subscribed: always per simulation; else static until leaves are terminal
        main._terminal_share = 0.12
        main._set_rows(plain)
        main._set_rows(plain)
        assert plain.calls == [4096]                                  # 'auto' switches once searches meet >= 3 % terminal leaves; no redundant calls
        main._terminal_share = 0.0
        main._set_rows(plain)
        assert plain.calls == [4096, 0]
        main.configure(board_size=9, n_blocks=1, out_planes=32, model=PVNet(1, 5, 32, 9), rows='static')
        main._terminal_share = 0.5
        main._set_rows(plain)
        assert plain.calls == [4096, 0]
        main.configure(board_size=9, n_blocks=1, out_planes=32, model=PVNet(1, 5, 32, 9), rows='dynamic', oversubscribe=1.0)
        main._terminal_share = 0.0
        main._set_rows(plain)
        assert plain.calls == [4096, 0, 4096] and main._slots(10 ** 6) == 4096
        assert main.CARRY_OVER is None or isinstance(main.CARRY_OVER, bool)
    finally:
        main.MAX_CONCURRENT = keep[0]
        main.configure(board_size=9, n_blocks=1, out_planes=32, model=PVNet(1, 5, 32, 9), oversubscribe=keep[1], rows=keep[2])
        main.CARRY_OVER = keep[3]


def test_pad_state_dict_keeps_the_function():
    """pvnet.pad_state_dict: a PVNet of any width (model.py:76-85) as the SAME function at the next multiple of 32 channels -- how
    widths the MFMA kernels are not built for reach the native forward (zero conv weights + identity BatchNorm statistics for the
    extra channels, zero columns / rows in the heads). torch on both: equal up to the order of summation."""
    import torch
    from alpha_omok_amd.pvnet import PVNet, native_width, pad_state_dict
    assert [native_width(p) for p in (20, 32, 33, 100, 128, 130, 250, 256, 300, 512, 513, 600)] == [32, 32, 64, 128, 128, 160, 256, 256, 320, 512, 513, 600]
    for nb, planes, B in ((2, 100, 9), (1, 40, 7), (3, 130, 5)):
        torch.manual_seed(planes)
        m = PVNet(nb, 5, planes, B)
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.uniform_(-0.2, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.uniform_(-0.2, 0.2)
        m.eval()
        width = native_width(planes)
        big = PVNet(nb, 5, width, B)
        sd = pad_state_dict(m.state_dict(), width)
        assert set(sd) == set(big.state_dict()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in big.state_dict().items())
        big.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        big.eval()
        x = (torch.rand(6, 5, B, B) < 0.3).float()
        with torch.no_grad():
            p0, v0 = m(x)
            p1, v1 = big(x)
        assert float((p0 - p1).abs().max()) < 1e-6 and float((v0 - v1).abs().max()) < 1e-6


def test_wide_networks_take_the_module_path_loudly():
    """model.PVNet takes any `planes` (model.py:76-85); the native forward stops at 512 (round 6; 256 before). A wider network --
    multiples of 32 included (round-5 advisor finding: 288 / 320 / 512 reached ao_net_create and raised when the limit was 256) -- is
    evaluated by its own torch module after ONE RuntimeWarning, or raises at once with strict_native. Decided before any GPU call."""
    import warnings
    from alpha_omok_amd.evaluator import Evaluator
    from alpha_omok_amd.pvnet import PVNet, native_supported
    assert [native_supported(p) for p in (1, 20, 256, 257, 288, 320, 512, 513, 544, 1024)] == [True, True, True, True, True, True, True, False, False, False]
    for planes in (513, 544, 1024):
        ev = Evaluator(0)
        m = PVNet(1, 5, planes, 5)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert ev.native_net(m, 5, 5) is None
            assert ev.native_net(m, 5, 5) is None
        assert len([x for x in w if issubclass(x.category, RuntimeWarning)]) == 1
        ev.strict_native = True
        with pytest.raises(ValueError, match="no native MI355X forward"):
            ev.native_net(m, 5, 5)
    # another board / plane count than the engine's: not a PVNet for this engine, no warning
    assert Evaluator(0).native_net(PVNet(1, 5, 544, 5), 9, 5) is None


def test_lazy_samples_unpack_like_the_tuples_they_stand_for():
    """main.self_play with rep_memory on the device appends utils.LazySamples to cur_memory: entries that count, unpack, index
    and stack like (state, pi, z) of main.py:159-166, with the state rebuilt (get_state_pt, utils.py:139-168) on first access."""
    from collections import deque
    from alpha_omok_amd import utils
    rs = np.random.RandomState(4)
    B, C, E = 5, 7, 6
    A = B * B
    lens = rs.randint(1, A + 1, E)
    moves = np.full((E, A), -1, np.int32)
    for e in range(E):
        moves[e, :lens[e]] = rs.permutation(A)[:lens[e]]
    ep_of = np.repeat(np.arange(E), lens)
    ply_of = np.concatenate([np.arange(l) for l in lens])
    pis = rs.dirichlet(np.ones(A), ep_of.size)
    z = rs.choice([-1.0, 0.0, 1.0], ep_of.size)
    blk = utils.LazySamples(moves, ep_of, ply_of, pis, z, B, C)
    mem = deque()
    mem.extend(blk)
    assert len(mem) == len(blk) == ep_of.size and blk._states is None        # counting builds nothing
    mem.pop()
    for i, (st, pi, zz) in enumerate(mem):
        node = (0,) + tuple(int(m) for m in moves[ep_of[i], :ply_of[i]])
        np.testing.assert_array_equal(st, utils.get_state_pt(node, B, C))
        np.testing.assert_array_equal(pi, pis[i])
        assert isinstance(zz, float) and zz == z[i]
    e0 = mem[0]
    assert len(e0) == 3 and e0[0].shape == (C, B, B) and e0[2] == z[0]
    assert np.stack([b[0] for b in list(mem)[:4]]).shape == (4, C, B, B)     # main.train_batch's host assembly


def test_sample_queue_behaves_like_the_deque_it_replaces():
    """main.cur_memory is a utils.SampleQueue: the deque operations the reference and its users apply to cur_memory (main.py:56,
    159-166, 229-231, 263, 374) on a mix of plain tuples and LazySamples blocks, compared with a real deque of the same entries."""
    from collections import deque
    from alpha_omok_amd import utils
    import alpha_omok_amd.main as main
    assert isinstance(main.cur_memory, utils.SampleQueue) and main.cur_memory.maxlen is None
    rs = np.random.RandomState(9)
    B, C = 4, 5
    A = B * B

    def block(E):
        lens = rs.randint(1, A + 1, E)
        moves = np.full((E, A), -1, np.int32)
        for e in range(E):
            moves[e, :lens[e]] = rs.permutation(A)[:lens[e]]
        ep_of = np.repeat(np.arange(E), lens)
        ply_of = np.concatenate([np.arange(l) for l in lens])
        return utils.LazySamples(moves, ep_of, ply_of, rs.dirichlet(np.ones(A), ep_of.size), rs.choice([-1.0, 0.0, 1.0], ep_of.size), B, C)

    def plain(n):
        return [(rs.rand(C, B, B), rs.dirichlet(np.ones(A)), float(rs.choice([-1.0, 1.0]))) for _ in range(n)]

    def same(q, d):
        assert len(q) == len(d) and bool(q) == bool(d)
        for i, (x, y) in enumerate(zip(q, d)):
            for u, v in zip(x, y):
                np.testing.assert_array_equal(u, v)
            for u, v in zip(q[i], y):
                np.testing.assert_array_equal(u, v)
        if len(d):
            np.testing.assert_array_equal(q[-1][1], d[-1][1])

    q, d = utils.SampleQueue(), deque()
    with pytest.raises(IndexError):
        q.pop()
    for step in range(3):
        b1 = block(3)
        q.extend(b1); d.extend(list(tuple(e) for e in b1))
        p = plain(2)
        q.extend(p); d.extend(p)
        q.append(p[0]); d.append(p[0])
        q.extend(utils.LazySamples(np.zeros((1, A), np.int32), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, A)), np.zeros(0), B, C))
        same(q, d)
        for _ in range(4):
            x, y = q.pop(), d.pop()
            np.testing.assert_array_equal(x[0], y[0]); assert x[2] == y[2]
        for _ in range(2):
            x, y = q.popleft(), d.popleft()
            np.testing.assert_array_equal(x[0], y[0]); assert x[2] == y[2]
        same(q, d)
    assert len(q[2:5]) == 3
    with pytest.raises(IndexError):
        q[len(q)]
    q.clear(); d.clear()
    same(q, d)


def test_overlapped_training_schedule_and_host_logic(monkeypatch):
    """configure(overlap_train=True): run() hands an iteration's training pass to train_async and only waits for it where somebody
    needs its result (the next self_play before it appends samples, a checkpoint, the end of run). Here: the call order of the
    loop, the synchronous fallback where the searches would have to call the module that is being trained (no GPU: no native
    export), train_join's bookkeeping, and 'serial' mode (same schedule, the pass runs inside train_join). main.py:377-414."""
    import random
    from alpha_omok_amd import main
    from alpha_omok_amd.evaluator import Evaluator
    calls = []
    monkeypatch.setattr(main, "Agent", object())
    monkeypatch.setattr(main, "self_play", lambda n: calls.append(("play", n)))
    monkeypatch.setattr(main, "train", lambda e, i: calls.append(("train", i)))
    monkeypatch.setattr(main, "train_async", lambda e, i: calls.append(("train_async", i)))
    monkeypatch.setattr(main, "train_join", lambda: calls.append(("join",)))
    monkeypatch.setattr(main, "load_data", lambda a, b: None)
    monkeypatch.setattr(main, "save_model", lambda *a, **k: calls.append(("save",)))
    monkeypatch.setattr(main, "save_dataset", lambda *a, **k: None)
    monkeypatch.setattr(main, "start_iter", 0)
    monkeypatch.setattr(main, "GAMES_PER_ITER", None)
    monkeypatch.setattr(main, "OVERLAP_TRAIN", True)
    assert main.run(total_iter=4, n_selfplay=7, save_every=2) == 4
    assert calls == [("play", 7), ("join",), ("save",), ("play", 1), ("train_async", 1), ("play", 1), ("train_async", 2), ("join",), ("save",),
                     ("play", 1), ("train_async", 3), ("join",)]
    monkeypatch.undo()

    with pytest.raises(ValueError):
        main.configure(board_size=3, n_mcts=4, n_blocks=1, out_planes=32, seed=1, gpu=0, overlap_train="yes")
    main.configure(board_size=3, n_mcts=4, n_blocks=1, out_planes=32, seed=1, gpu=0, overlap_train=True)
    try:
        assert main.OVERLAP_TRAIN is True and main.train_join() is None
        rs = np.random.RandomState(0)
        main.rep_memory.clear(); main.cur_memory.clear()
        for _ in range(200):
            pi = rs.rand(9); pi /= pi.sum()
            main.rep_memory.append(((rs.rand(5, 3, 3) < 0.3).astype(np.float64), pi, float(rs.choice([-1.0, 0.0, 1.0]))))
        main.cur_memory.extend(list(main.rep_memory)[:3])
        main.TRAIN_STEPS, main.BATCH_SIZE = 4, 16
        import copy
        import torch
        w0 = copy.deepcopy(main.Agent.model.state_dict())
        opt0 = copy.deepcopy(main.optimizer.state_dict())
        random.seed(4); torch.manual_seed(4); main.step = 0
        ref = main.train(1, 1)
        w_ref = copy.deepcopy(main.Agent.model.state_dict())
        # no GPU here: nothing native to freeze -> the pass runs at once, on this thread, and train_join only hands the losses over
        main.Agent.model.load_state_dict(w0); main.optimizer.load_state_dict(opt0)
        random.seed(4); torch.manual_seed(4); main.step = 0
        main.train_async(1, 1)
        assert main.step == 4 and main._train_job is not None and main._train_job['thread'] is None
        got = main.train_join()
        assert main._train_job is None and main.train_join() is None and main.last_train_losses == got
        np.testing.assert_array_equal(np.array(got), np.array(ref))
        for k, v in main.Agent.model.state_dict().items():
            assert torch.equal(v, w_ref[k]), k

        # Evaluator.freeze / thaw: a frozen evaluator hands out the copy exported last without looking at the module
        ev = Evaluator(0)
        sentinel = object()
        ev._net = sentinel
        ev._frozen = True
        assert ev.native_net(main.Agent.model, 3, 5) is sentinel
        ev._net = None                                  # (nothing exported: freeze() reports that it cannot, and stays thawed)
        assert ev.freeze(object(), 3, 5) is False and ev._frozen is False
        ev._key = ("x",)
        ev.thaw()
        assert ev._frozen is False and ev._key is None

        # 'serial': the pass is planned by train_async (the `random` stream moves there) and executed inside train_join
        class FakeEval:
            frozen = thawed = 0
            def freeze(self, *a): FakeEval.frozen += 1; return True
            def thaw(self): FakeEval.thawed += 1
            def invalidate(self): pass
        monkeypatch.setattr(main, "_evaluator", FakeEval())
        monkeypatch.setattr(main, "device", type("D", (), {"type": "cuda"})())
        monkeypatch.setattr(main, "OVERLAP_TRAIN", "serial")
        main.Agent.model.load_state_dict(w0); main.optimizer.load_state_dict(opt0)
        random.seed(4); torch.manual_seed(4); main.step = 0
        state_before = random.getstate()
        main.train_async(1, 1)
        assert main.step == 0 and FakeEval.frozen == 1 and random.getstate() != state_before
        monkeypatch.undo()                              # (the real device again for the pass itself)
        monkeypatch.setattr(main, "_evaluator", FakeEval())
        got = main.train_join()
        assert main.step == 4 and FakeEval.thawed == 1
        np.testing.assert_array_equal(np.array(got), np.array(ref))
        # a pass that cannot be planned (the reference's ValueError, main.py:263-264) leaves nothing frozen and nothing in flight
        monkeypatch.setattr(main, "device", type("D", (), {"type": "cuda"})())
        monkeypatch.setattr(main, "OVERLAP_TRAIN", "serial")
        main.TRAIN_STEPS = 1000
        with pytest.raises(ValueError):
            main.train_async(1, 1)
        assert FakeEval.frozen == 2 and FakeEval.thawed == 2 and main._train_job is None
        main.TRAIN_STEPS = 4
    finally:
        monkeypatch.undo()
        main._train_job = None
        main.TRAIN_STEPS, main.BATCH_SIZE, main.OVERLAP_TRAIN = None, 32, False
        main.rep_memory.clear(); main.cur_memory.clear()


def test_gpu_numa_mapping_and_pinning_from_a_fake_sysfs(tmp_path, monkeypatch):
    """parallel.pin_to_gpu_numa (round-5 review, first 8-GPU contact): GPU -> PCI address -> numa_node -> cpulist -> sched_setaffinity,
    looked up under /sys/bus/pci/devices or through /sys/class/drm/card*/device; platforms that report no node (-1) and the
    AO_NO_AFFINITY switch leave the process alone. Host logic only: a fake sysfs tree."""
    from alpha_omok_amd import parallel
    assert parallel.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parallel.parse_cpulist("5") == [5] and parallel.parse_cpulist("") == []
    root = tmp_path / "sys"
    mine = sorted(os.sched_getaffinity(0))
    node1 = mine[:max(1, len(mine) // 2)]

    def cpulist(cpus):
        return ",".join(str(c) for c in cpus) + "\n"
    for node, cpus in ((0, [100000, 100001]), (1, node1)):
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpulist(cpus))
    for pci, node in (("0000:05:00.0", 1), ("0000:85:00.0", 0), ("0000:c5:00.0", -1)):
        d = root / "bus" / "pci" / "devices" / pci
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    # a device only reachable through the drm class links
    real = root / "devices" / "pci0000:e0" / "0000:e5:00.0"
    real.mkdir(parents=True)
    (real / "numa_node").write_text("1\n")
    drm = root / "class" / "drm"
    drm.mkdir(parents=True)
    (drm / "card3").mkdir()
    os.symlink(str(real), str(drm / "card3" / "device"))
    (drm / "card3-DP-1").mkdir()
    assert parallel.gpu_numa_cpus("0000:05:00.0", str(root)) == (1, node1)
    assert parallel.gpu_numa_cpus("0000:85:00.0", str(root)) == (0, [100000, 100001])
    assert parallel.gpu_numa_cpus("0000:C5:00.0", str(root)) == (None, [])          # the platform says "no node"
    assert parallel.gpu_numa_cpus("0000:e5:00.0", str(root)) == (1, node1)          # via /sys/class/drm/card3/device
    assert parallel.gpu_numa_cpus("0000:ff:00.0", str(root)) == (None, [])
    try:
        r = parallel.pin_to_gpu_numa(0, str(root), pci_bus_id="0000:05:00.0")
        assert r == dict(numa_node=1, cpus=len(node1), pci="0000:05:00.0") and sorted(os.sched_getaffinity(0)) == node1
        os.sched_setaffinity(0, mine)
        r = parallel.pin_to_gpu_numa(0, str(root), pci_bus_id="0000:85:00.0")    # a node whose CPUs this process may not use
        assert "skipped" in r and sorted(os.sched_getaffinity(0)) == mine
        r = parallel.pin_to_gpu_numa(0, str(root), pci_bus_id="0000:c5:00.0")
        assert "skipped" in r and sorted(os.sched_getaffinity(0)) == mine
        monkeypatch.setenv("AO_NO_AFFINITY", "1")
        assert parallel.pin_to_gpu_numa(0, str(root), pci_bus_id="0000:05:00.0") == dict(skipped="AO_NO_AFFINITY")
        assert sorted(os.sched_getaffinity(0)) == mine
    finally:
        os.sched_setaffinity(0, mine)


_DEAD_PEER_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(repo)r)
import torch
import torch.distributed as dist
from alpha_omok_amd import parallel
rank = int(os.environ["RANK"])
parallel.init_from_env("gloo", timeout_s=60, pin=False)
t = torch.ones(4)
dist.all_reduce(t)                                        # start-up: both ranks are there
assert float(t[0]) == 2.0
assert parallel.set_collective_timeout(4.0)
mode = os.environ["AO_TEST_MODE"]
with parallel.Watchdog(6.0, "test loop") as dog:
    for i in range(1000):
        if rank == 1 and i == 3:
            if mode == "die":
                os._exit(9)
            time.sleep(3600)                              # "hang": no beats, no collectives
        dist.all_reduce(t)
        dog.beat("step %%d" %% i)
        time.sleep(0.05)
'''


@pytest.mark.parametrize("mode", ["die", "hang"])
def test_a_dead_or_hung_rank_fails_every_rank_quickly(tmp_path, mode):
    """First-contact hardening (round-5 review): under torch.distributed a rank that raises or hangs used to leave its peers blocked
    in their next all-reduce for torch's default 30 minutes (10 under RCCL). With parallel.init_from_env's timeout,
    set_collective_timeout after start-up and parallel.Watchdog: rank 1 dies (exit 9) or stops making progress -- rank 0's next
    collective fails within the timeout and it exits non-zero; a hung rank ends itself with exit code 86. Two plain processes, no
    launcher that would clean up for them; gloo on CPU."""
    import socket
    import subprocess
    import time
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_DEAD_PEER_WORKER % dict(repo=REPO))
    procs = []
    t0 = time.time()
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   AO_TEST_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=120) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    wall = time.time() - t0
    rc = [p.returncode for p in procs]
    assert rc[0] != 0, "rank 0 did not notice its peer: %r" % (outs[0][1][-500:],)
    assert rc[1] == (9 if mode == "die" else 86), (rc, outs[1][1][-500:])
    if mode == "hang":
        assert "made no progress" in outs[1][1]
    assert wall < 60.0, wall


def _cpu_samples(rs, n, B=9):
    return [((rs.rand(5, B, B) < 0.3).astype(np.float64), rs.dirichlet(np.ones(B * B)), float(rs.choice([-1, 0, 1]))) for _ in range(n)]


def test_fp16_grid_training_on_cpu_keeps_conv_weights_on_the_grid_and_masters_off_it():
    """configure(fp16_grid_weights=True) (round 6: networks whose conv weights are fp16 numbers run on the two-product kernels):
    host logic only -- torch on CPU. After a pass every 3x3 conv weight of the module is an fp16 number, the fp32 master copies Adam
    moves are not, everything else trains as ever, the state_dict is plain fp32, load_data re-synchronises the masters, and turning
    the switch off drops them."""
    import torch
    from alpha_omok_amd import main
    from alpha_omok_amd.pvnet import PVNet
    torch.manual_seed(11)
    try:
        main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9), fp16_grid_weights=True,
                       overlap_train=False, carry_over=False, device_replay=False)
        model = main.Agent.model
        convs = [p for p in model.parameters() if p.dim() == 4 and p.shape[2] == 3]
        assert len(convs) == 3 and len(main._grid_masters) == 3
        for p in convs:
            assert torch.equal(p.detach().half().float(), p.detach())               # projected at configure()
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        rs = np.random.RandomState(3)
        random.seed(9)
        main.cur_memory.clear(); main.rep_memory.clear()
        main.cur_memory.extend(_cpu_samples(rs, 3))
        main.rep_memory.extend(_cpu_samples(rs, 200))
        main.step = 0
        losses = main.train(1, 1)
        assert len(losses) == 3 and main.step == 3 and all(np.isfinite(l).all() for l in losses)
        for p in convs:
            assert torch.equal(p.detach().half().float(), p.detach())
        assert all(not torch.equal(m.half().float(), m) for _, m in main._grid_masters)    # the masters are ordinary fp32
        assert all(not torch.equal(p.detach(), before[n]) for n, p in model.named_parameters())
        assert all(v.dtype == torch.float32 for v in model.state_dict().values() if v.is_floating_point())
        # the module's conv weights ARE the rounding of the masters
        for p, m in main._grid_masters:
            assert torch.equal(p.detach(), m.half().float())
    finally:
        main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9), fp16_grid_weights=False)
    assert main._grid_masters == [] and main.FP16_GRID is False


def _nonfinite_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from alpha_omok_amd import main, parallel
    from alpha_omok_amd.pvnet import PVNet
    parallel.init_from_env("gloo", pin=False)
    torch.manual_seed(50 + rank)
    main.configure(board_size=9, n_blocks=1, out_planes=32, seed=4, model=PVNet(1, 5, 32, 9), fp16_grid_weights=False)
    rs = np.random.RandomState(100 + rank)
    random.seed(200 + rank)
    main.cur_memory.clear(); main.rep_memory.clear()
    main.cur_memory.extend(_cpu_samples(rs, 2))
    rep = _cpu_samples(rs, 64)
    if rank == 1:                                         # ONE rank's shard is poisoned: every batch it draws has an infinite target
        rep = [(s, p, float("inf")) for s, p, _ in rep]
    main.rep_memory.extend(rep)
    main.step = 0
    main.skipped_steps = 0
    w0 = {k: v.clone() for k, v in main.Agent.model.state_dict().items()}
    losses = main.train(1, 1)
    sd = {k: v.clone() for k, v in main.Agent.model.state_dict().items()}
    torch.save(dict(sd=sd, w0=w0, skipped=main.skipped_steps, step=main.step, n_losses=len(losses)), out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_a_non_finite_gradient_on_one_rank_is_dropped_on_every_rank(tmp_path):
    """main.train_batch's guard (round-5 advisor: it used to synchronise the host per step and to apply with one process only, so
    multi-rank training wrote NaN into the weights as the reference would, main.py:296-305): after the gradient all-reduce every rank
    holds the same -- non-finite -- gradient, zeroes it on the device and counts the mini-batch. Weights finite and bit-identical on
    both ranks, both counted every step of the pass, and with zero gradients from step one Adam leaves the parameters where they were."""
    import torch
    import torch.multiprocessing as mp
    port = 29500 + random.randint(0, 2000)
    out = str(tmp_path / "nf%d.pt")
    mp.spawn(_nonfinite_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0["step"] == r1["step"] == 2 and r0["skipped"] == r1["skipped"] == 2
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
        if r0["sd"][k].is_floating_point():
            assert torch.isfinite(r0["sd"][k]).all(), k
    params = [k for k in r0["sd"] if "running_" not in k and "num_batches" not in k]
    assert all(torch.equal(r0["sd"][k], r0["w0"][k]) for k in params)       # (rank 0's initial weights: configure() broadcast them)
