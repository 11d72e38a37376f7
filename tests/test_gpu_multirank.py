"""-m gpu: the N > 1 self-play path on ONE GPU -- two processes (torch.distributed,
gloo rendezvous on 127.0.0.1) share cuda:0, main.self_play shards the episodes e % world == rank,
nothing is exchanged. Episode e's samples must not depend on the sharding: the union of the two
ranks' memories equals the single-process run of the same episodes, bit for bit (the native PVNet
with default-initialised weights under torch.manual_seed evaluates the leaves, so the split-fp16 /
per-board kernels are on the path too)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, S, NB, EPISODES = 9, 24, 2, 6


def _run(n_episodes):
    from alpha_omok_amd import main
    main.configure(board_size=B, n_mcts=S, n_blocks=NB, seed=7, gpu=0)
    main.cur_memory.clear()
    main.self_play(n_episodes)
    return [(np.asarray(s, np.float64), np.asarray(p, np.float64), float(z)) for s, p, z in main.cur_memory]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from alpha_omok_amd import parallel
    parallel.init_from_env("gloo")
    mem = _run(EPISODES)
    torch.save(dict(shard=parallel.shard_games(EPISODES, rank, world), mem=mem), out % rank)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _episodes_of(mem):
    """Split a chronological sample list into episodes: an episode starts at the empty board."""
    eps = []
    for s, p, z in mem:
        if not s[:4].any():   # no stones in the history planes: first ply of a game
            eps.append([])
        eps[-1].append((s, p, z))
    return eps


def test_two_ranks_on_one_gpu_reproduce_the_single_process_run(tmp_path):
    single = _episodes_of(_run(EPISODES))
    assert len(single) == EPISODES
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    seen = set()
    for rank in range(2):
        r = torch.load(out % rank, weights_only=False)
        assert r["shard"] == list(range(rank, EPISODES, 2))
        eps = _episodes_of(r["mem"])
        assert len(eps) == len(r["shard"])
        for e, got in zip(r["shard"], eps):
            want = single[e]
            assert len(got) == len(want), "episode %d: %d plies on rank %d, %d alone" % (e, len(got), rank, len(want))
            for (s0, p0, z0), (s1, p1, z1) in zip(got, want):
                assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1
            seen.add(e)
    assert seen == set(range(EPISODES))
