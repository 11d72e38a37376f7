"""-m gpu: the loop people actually run -- tools/train_omok.py's defaults, device_replay + oversubscribe + carry_over +
overlap_train TOGETHER (the reference's loop: main.py:122-250 self_play, :377-414 the iteration loop) -- and the guard that
makes a search short of its simulations an error (agents.py:105-132: the reference always runs num_mcts of them).

Round-5 review: each switch had a test, the product of them had none; the over-subscribed search could leave its catch-up
loop with games short of their simulations and nobody looked."""
import os
import socket
import sys
from collections import deque

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

B, S, N, CALLS, ROWS_CAP, MEM = 9, 32, 24, 3, 16, 6000


def _trained_sd():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from make_trained_fixture import load
    return load(os.path.join(REPO, "tests", "golden", "trained_2block_9x9.npz"))


def _model():
    from alpha_omok_amd.pvnet import PVNet
    m = PVNet(2, 5, 128, B)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in _trained_sd().items()})
    return m.cuda().eval()


def _loop(prod, train_steps, device_replay=None, n=N):
    """CALLS iterations of main.run()'s body: self_play(n); train (iterations > 0); reset_iter. prod = the production switches,
    otherwise the synchronous schedule (per-move packing, one slot per row, nothing in flight between calls, train() in line).
    Returns per call the samples, the results, the replay memory's content at the end and what the engine did."""
    import random
    import alpha_omok_amd.main as main
    main.MAX_CONCURRENT = ROWS_CAP
    main.MEMORY_SIZE = MEM
    main.rep_memory = deque(maxlen=MEM)
    dr = prod if device_replay is None else device_replay
    main.configure(board_size=B, n_mcts=S, n_blocks=2, in_planes=5, out_planes=128, seed=21, model=_model(), reproducible=True,
                   node_cap=0, strict=True, device_replay=dr,
                   carry_over=bool(prod), oversubscribe=1.25 if prod else 1.0, rows='auto' if prod else 'static',
                   overlap_train='serial' if prod else False)
    main.TRAIN_STEPS, main.BATCH_SIZE = train_steps, 32
    main.result.update(Black=0, White=0, Draw=0)
    main.rep_memory.clear()
    main.cur_memory.clear()
    main.step = 0
    random.seed(5)
    calls, results, rets = [], [], []
    try:
        for c in range(CALLS):
            ret = main.self_play(n)
            rets.append(ret)
            assert ret['moves'] == len(main.cur_memory)
            calls.append([(np.asarray(s, np.float64).copy(), np.asarray(p, np.float64).copy(), float(z)) for s, p, z in main.cur_memory])
            results.append(dict(main.result))
            if c > 0:
                if main.OVERLAP_TRAIN:
                    main.train_async(1, c)
                else:
                    main.train(1, c)
            main.reset_iter(main.result, main.cur_memory)
        main.train_join()
        rep = [(np.asarray(s, np.float64), np.asarray(p, np.float64), float(z)) for s, p, z in main.rep_memory]
        eng = main._engine
        info = dict(G=eng.G, rows=eng.row_stats(), in_flight=int(main._pool.active.sum()) if main._pool is not None else 0,
                    step=main.step, totals=dict(main.search_totals), rep_len=len(main.rep_memory), rep_max=main.rep_memory.maxlen)
    finally:
        main.MAX_CONCURRENT = 4096
        main.MEMORY_SIZE = 30000
        main.TRAIN_STEPS = None
        main.rep_memory = deque(maxlen=30000)
        main.configure(board_size=B, n_mcts=S, n_blocks=2, out_planes=128, seed=0, reproducible=False, strict=False, carry_over=False,
                       oversubscribe=1.0, rows='auto', overlap_train=False, device_replay=False)
        main.release_engine()
    return calls, results, rep, info, rets


def _episodes_of(mem):
    eps = []
    for s, p, z in mem:
        if not s[:4].any():   # no stones in the history planes: first ply of a game
            eps.append([])
        eps[-1].append((s, p, z))
    return eps


def _same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for (s0, p0, z0), (s1, p1, z1) in zip(a, b):
        assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1


def test_production_switches_together_deliver_the_synchronous_runs_episodes():
    """device_replay + oversubscribe 1.25 + carry_over + overlap_train='serial' (20 game slots on 16 rows, 24 episodes per call so
    games end -- and slots are refilled with LATER calls' episodes -- inside every call), strict=True (visit.sum() == inherited + S
    checked after every search), a trained network (terminal leaves in the batches), three calls with a training pass after the
    second and third. With a pass that leaves the weights alone (TRAIN_STEPS = 0) and one kernel family (reproducible=True):
      (i)   every call returns exactly its own n episodes, in episode order: the samples of the synchronous, per-move packed,
            nothing-in-flight run, sample for sample (an episode is the game its seed fixes);
      (ii)  the results (Black / White / Draw) agree call by call;
      (iii) the replay memory ends up with the same entries in the same order as the reference's deque path
            (rep_memory.extend(augment_dataset(...)) with maxlen cutting the oldest), the ring having wrapped."""
    sync_calls, sync_res, sync_rep, sync_info, _ = _loop(False, 0)
    prod_calls, prod_res, prod_rep, prod_info, rets = _loop(True, 0)
    assert sync_info['G'] == 16 and prod_info['G'] == 20
    assert sync_info['rows']['launches'] == 0 and prod_info['rows']['launches'] > 0
    assert prod_info['in_flight'] > 0                       # later calls' games were started: the engine stayed full
    total = 0
    for c in range(CALLS):
        assert rets[c]['episodes'] == N
        assert len(_episodes_of(prod_calls[c])) == N
        _same(sync_calls[c], prod_calls[c])
        total += len(prod_calls[c])
    assert sync_res == prod_res
    assert 8 * total > MEM                                  # (the ring wrapped)
    assert prod_info['rep_len'] == sync_info['rep_len'] == min(8 * total, MEM) and prod_info['rep_max'] == MEM
    _same(sync_rep, prod_rep)


def test_production_switches_with_a_real_training_pass_deliver_every_episode_once():
    """The same loop with passes that DO change the weights (8 mini-batches after the second and third call): episodes started
    under older weights cannot be compared with the synchronous run any more; what must hold is the bookkeeping -- every call
    returns n complete episodes that begin at the empty board and end with a decided z, rep_memory grew by 8 x samples per call
    (until full), the optimiser stepped, and strict mode saw inherited + S visits in every search of every game."""
    calls, results, rep, info, rets = _loop(True, 8)
    assert info['step'] == 8 * (CALLS - 1)
    total = 0
    for c in range(CALLS):
        eps = _episodes_of(calls[c])
        assert rets[c]['episodes'] == N and len(eps) == N
        for ep in eps:
            assert not ep[0][0][:4].any()                   # starts at the empty board
            stones = [int(s[:4].any(axis=0).sum()) for s, _, _ in ep]
            # plies in order: the stones on the board (union of the history planes) never decrease
            assert all(b >= a for a, b in zip(stones, stones[1:]))
            zs = {abs(z) for _, _, z in ep}
            assert zs in ({1.0}, {0.0})
        assert sum(results[c].values()) == N
        total += len(calls[c])
    assert info['rep_len'] == min(8 * total, MEM)


def test_a_search_short_of_its_simulations_is_an_error_not_a_result(monkeypatch):
    """ao_search on an over-subscribed engine (72 games on 48 rows): after the nominal number of launches some games are short of
    their simulations (leaves that waited for a row, launches sat out) and the catch-up loop runs until every game has them.
    AO_CATCHUP_ROUNDS=0 (developer switch) skips that loop: the search must then END IN AN ERROR that names a game and its deficit
    (k_end_move's ERR_SHORT) -- never in a pi built from fewer than S visits -- and the engine must be usable afterwards. Without the
    switch: S visits on top of what was inherited in every game, equal to the search of the same games with one row per game.
    The step-wise protocol has the same guard (ao_end_move before the simulations are through)."""
    from alpha_omok_amd.engine import Engine, EngineError, Net
    G, ROWS, S_ = 72, 48, 48
    net = Net(2, 5, 128, B, 0)
    net.load_state_dict(_trained_sd())
    net.set_mode(6)
    seeds = [900 + g for g in range(G)]
    tau = np.ones(G, np.int8)

    def engine(cap):
        e = Engine(B, S_, 5, games=G, noise=True)
        e.seed_all(seeds)
        e.set_row_cap(cap)
        return e

    ref, over = engine(0), engine(ROWS)
    inherited = np.zeros(G, np.int64)
    alive = np.ones(G, bool)
    for ply in range(8):                                    # (a trained network: terminal leaves inside the batches from ply 6 or so; no game can end before ply 9)
        pi0, vis0, pol0 = ref.search(net, tau=tau)
        pi1, vis1, pol1 = over.search(net, tau=tau)
        assert np.array_equal(vis0[alive], vis1[alive]) and np.array_equal(pi0[alive], pi1[alive]) and np.array_equal(pol0[alive], pol1[alive])
        assert np.array_equal(vis1.sum(axis=1)[alive], (inherited + S_)[alive])
        a0, w0 = ref.play()
        a1, w1 = over.play()
        assert np.array_equal(a0, a1) and np.array_equal(w0, w1)
        inherited = np.maximum(vis1[np.arange(G), np.maximum(a1, 0)].astype(np.int64) - 1, 0)
        alive &= (w1 == 0)
    assert alive.all()
    assert over.search_stats()['terminal'] > 0              # terminal leaves (which take no row) were met
    assert over.row_stats()['waits'] > 0                    # leaves DID wait: the catch-up loop had work to do
    # the branch a stalled round takes -- nobody sits out any more (the controller's window is dropped in the middle of a move) -- forced
    # from the first catch-up round on: the same results
    monkeypatch.setenv("AO_CATCHUP_WINDOW_OFF", "1")
    pi0, vis0, pol0 = ref.search(net, tau=tau)
    pi1, vis1, pol1 = over.search(net, tau=tau)
    monkeypatch.delenv("AO_CATCHUP_WINDOW_OFF")
    assert np.array_equal(vis0[alive], vis1[alive]) and np.array_equal(pi0[alive], pi1[alive]) and np.array_equal(pol0[alive], pol1[alive])
    a0, w0 = ref.play()
    a1, w1 = over.play()
    assert np.array_equal(a0, a1) and np.array_equal(w0, w1)
    alive &= (w1 == 0)
    # the same engine, catch-up loop cut off
    monkeypatch.setenv("AO_CATCHUP_ROUNDS", "0")
    short = 0
    for _ in range(3):
        try:
            over.search(net, tau=tau)
            ref.search(net, tau=tau)
            over.play()
            ref.play()
        except EngineError as ex:
            msg = str(ex)
            assert "simulations" in msg and "game " in msg and "rows per simulation: 48" in msg, msg
            short += 1
            break
    assert short == 1, "no search came up short with the catch-up loop disabled"
    monkeypatch.delenv("AO_CATCHUP_ROUNDS")
    # the engine is usable after the error: fresh games, full searches again
    over.reset()
    over.seed_all(seeds)
    pi, vis, _ = over.search(net, tau=tau)
    assert np.all(vis.sum(axis=1) == S_)
    # step-wise protocol: ending the move before the simulations are through
    e = Engine(B, 8, 5, games=4, noise=False)
    e.seed_all([1, 2, 3, 4])
    planes = torch.zeros((4, 5, B, B), dtype=torch.float32, device="cuda:0")
    e.begin_move()
    for _ in range(3):
        e.collect_leaves(planes.data_ptr())
        e.sync()
        p, v = net(planes)
        torch.cuda.synchronize()
        e.apply_evals(p.data_ptr(), v.data_ptr())
    with pytest.raises(EngineError, match="3 of 9 simulations"):
        e.end_move(np.ones(4, np.int8))


# ---- the same loop under two ranks (gloo rendezvous, both on cuda:0) ----
def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from alpha_omok_amd import parallel
    parallel.init_from_env("gloo")
    calls, results, rep, info, rets = _loop(True, 0)
    torch.save(dict(calls=calls, results=results, rep_len=info['rep_len'], rets=rets, in_flight=info['in_flight']), out % rank)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_production_switches_under_two_ranks_return_each_ranks_shard_of_every_call(tmp_path):
    """Two processes (episodes e % 2 == rank), the production switches on both: call by call each rank returns ITS episodes of that
    call -- the samples the single-process synchronous run produces for them -- and its rank-local replay holds 8 x its samples."""
    sync_calls, _, _, _, _ = _loop(False, 0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "prod%d.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(out % rank, weights_only=False)
        total = 0
        for c in range(CALLS):
            want = _episodes_of(sync_calls[c])
            got = _episodes_of(r["calls"][c])
            shard = list(range(rank, N, 2))
            assert len(want) == N and len(got) == len(shard) == r["rets"][c]["episodes"]
            for e, g in zip(shard, got):
                _same(want[e], g)
            total += len(r["calls"][c])
        assert r["rep_len"] == min(8 * total, MEM)
