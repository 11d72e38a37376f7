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
    main.configure(board_size=B, n_mcts=S, n_blocks=NB, in_planes=5, out_planes=128, seed=7, gpu=0, node_cap=0, strict=False)   # (every knob: other tests leave theirs in the module)
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


def _run_calls(calls, carry, conc):
    """`calls` consecutive self_play(EPISODES) calls; one sample list per call."""
    from alpha_omok_amd import main
    main.MAX_CONCURRENT = conc
    main.configure(board_size=B, n_mcts=S, n_blocks=NB, in_planes=5, out_planes=128, seed=11, gpu=0, node_cap=0, strict=False,
                   carry_over=carry)
    out = []
    for _ in range(calls):
        main.cur_memory.clear()
        main.self_play(EPISODES)
        out.append([(np.asarray(s, np.float64), np.asarray(p, np.float64), float(z)) for s, p, z in main.cur_memory])
    in_flight = int(main._pool.active.sum()) if main._pool is not None else 0
    main.MAX_CONCURRENT = 4096
    main.configure(board_size=B, n_mcts=S, n_blocks=NB, in_planes=5, out_planes=128, seed=7, gpu=0, carry_over=False)
    return out, in_flight


def _carry_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from alpha_omok_amd import parallel
    parallel.init_from_env("gloo")
    mems, in_flight = _run_calls(3, True, 2)
    torch.save(dict(mems=mems, in_flight=in_flight), out % rank)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_carry_over_return_their_shards_of_every_call(tmp_path):
    """configure(carry_over=True) under two ranks: each rank keeps ITS shard's episodes of the later calls in flight (two
    slots per rank, three episodes per rank and call) and returns, call by call, exactly the episodes e % 2 == rank of that
    call -- the same samples a single process produces for them with the synchronous schedule."""
    single, fl = _run_calls(3, False, 4096)
    assert fl == 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "carry%d.pt")
    mp.spawn(_carry_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(out % rank, weights_only=False)
        assert r["in_flight"] > 0
        for c in range(3):
            want = _episodes_of(single[c])
            got = _episodes_of(r["mems"][c])
            shard = list(range(rank, EPISODES, 2))
            assert len(want) == EPISODES and len(got) == len(shard)
            for e, g in zip(shard, got):
                assert len(g) == len(want[e]), "call %d episode %d: %d plies on rank %d, %d alone" % (c, e, len(g), rank, len(want[e]))
                for (s0, p0, z0), (s1, p1, z1) in zip(g, want[e]):
                    assert np.array_equal(s0, s1) and np.array_equal(p0, p1) and z0 == z1


def _run_worker(rank, world, port, out, tmp, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import random
    from alpha_omok_amd import main, parallel
    parallel.init_from_env("gloo")
    torch.manual_seed(100 + rank)                      # configure() must broadcast rank 0's weights
    main.configure(board_size=B, n_mcts=8, n_blocks=1, out_planes=32, seed=3, gpu=0, overlap_train=overlap)
    random.seed(300 + rank)                            # rank-local replay draws
    main.rep_memory.clear(); main.cur_memory.clear()
    main.step = 0; main.start_iter = 0
    lens = []
    orig_plan = main._train_plan
    def spy():                                         # (train and train_async both plan their pass here, on the calling thread)
        lens.append(len(main.cur_memory))
        return orig_plan()
    main._train_plan = spy
    threads = []
    if overlap:
        orig_async = main.train_async
        def spy_async(n_epochs, n_iter):
            orig_async(n_epochs, n_iter)
            threads.append(main._train_job['thread'] is not None)
        main.train_async = spy_async
    n = main.run(total_iter=4 if overlap else 3, n_selfplay=7, save_every=2, directory=tmp)
    main._train_plan = orig_plan
    if overlap:
        main.train_async = orig_async
        assert threads == [True] * 3 and main._train_job is None, threads
    sd = {k: v.detach().cpu().clone() for k, v in main.Agent.model.state_dict().items()}
    torch.save(dict(n=n, step=main.step, sd=sd, lens=lens, rep=len(main.rep_memory)), out % rank)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_run_the_training_loop_without_deadlock(tmp_path):
    """main.run (the reference's __main__ loop, main.py:377-414) under torch.distributed: iteration 0
    shards 7 games 4 / 3, iterations 1 and 2 play one game per rank (different lengths) and train --
    every rank runs the same number of optimiser steps (ceil(sum of new samples / world)), the
    weights and BatchNorm buffers stay bit-identical across ranks, only rank 0 writes checkpoints."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "run%d.pt")
    ck = tmp_path / "ck"
    mp.spawn(_run_worker, args=(2, port, out, str(ck)), nprocs=2, join=True)
    r0, r1 = (torch.load(out % r, weights_only=False) for r in range(2))
    assert r0["n"] == r1["n"] == 3
    assert len(r0["lens"]) == len(r1["lens"]) == 2
    want_steps = sum(-(-(a + b) // 2) for a, b in zip(r0["lens"], r1["lens"]))
    assert r0["step"] == r1["step"] == want_steps > 0
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    models = [f for f in os.listdir(ck) if f.endswith("_step_model.pickle")]
    assert len(models) == 2                               # n_iter 0 and 2, written once (rank 0)


def test_two_ranks_run_the_overlapped_training_loop(tmp_path):
    """The same loop with configure(overlap_train=True): every pass is planned on the main thread (the agreement collectives), its
    gradient all-reduces are issued by the worker thread while the main thread plays the next iteration's games (no collective
    there), joined before the next plan / checkpoint -- every rank issues the same collectives in the same order: no deadlock,
    the same number of optimiser steps everywhere, bit-identical weights and BatchNorm buffers across ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "run%d.pt")
    ck = tmp_path / "ck"
    mp.spawn(_run_worker, args=(2, port, out, str(ck), True), nprocs=2, join=True)
    r0, r1 = (torch.load(out % r, weights_only=False) for r in range(2))
    assert r0["n"] == r1["n"] == 4
    assert len(r0["lens"]) == len(r1["lens"]) == 3
    want_steps = sum(-(-(a + b) // 2) for a, b in zip(r0["lens"], r1["lens"]))
    assert r0["step"] == r1["step"] == want_steps > 0
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    models = [f for f in os.listdir(ck) if f.endswith("_step_model.pickle")]
    assert len(models) == 2                               # n_iter 0 and 2 (its pass joined first), written once (rank 0)


def _rccl_worker(rank, world, port, out):
    """ONE rank, backend "nccl" (= RCCL on ROCm) on cuda:0: every collective of parallel.py runs on device buffers."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from alpha_omok_amd import parallel
    from alpha_omok_amd.pvnet import PVNet
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and parallel.world() == (0, 1) and parallel._collectives_on()
    torch.manual_seed(5)
    net = PVNet(2, 5, 32, 9).cuda()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    parallel.broadcast_parameters(net)                       # ncclBroadcast of one flat device buffer
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    net.train()
    x = (torch.rand(8, 5, 9, 9, device="cuda") < 0.3).float()
    p, v = net(x)
    (p.log().mean() + v.pow(2).mean()).backward()
    g0 = [q.grad.clone() for q in net.parameters()]
    numel, contributors = parallel.allreduce_gradients(net, contributes=True)   # ncclAllReduce(sum) + contributor count
    assert contributors == 1 and numel == sum(q.numel() for q in net.parameters())
    for q, g in zip(net.parameters(), g0):
        assert torch.equal(q.grad, g), "a one-rank all-reduce must leave the gradient unchanged"
    b0 = [b.clone() for b in net.buffers()]
    parallel.average_buffers(net, contributes=True)
    for b, w in zip(net.buffers(), b0):
        assert torch.equal(b, w)
    assert parallel.agree(17, "sum", torch.device("cuda", 0)) == 17 and parallel.agree(3, "max") == 3
    # a rank that does not contribute adds zeros and the count says so
    _, c0 = parallel.allreduce_gradients(net, contributes=False)
    assert c0 == 0
    maps = open("/proc/self/maps").read()
    torch.save(dict(rccl_mapped="librccl" in maps, backend=dist.get_backend(), numel=numel), out)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_every_collective_on_device_buffers(tmp_path):
    """RCCL itself (librccl.so, the `nccl` backend) on this 1-GPU box: a process group of ONE rank drives
    broadcast_parameters / allreduce_gradients / average_buffers / agree on cuda:0 buffers -- the branches the gloo
    tests never enter. At world 8 the same code runs over xGMI."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(1, port, out), nprocs=1, join=True)
    r = torch.load(out, weights_only=False)
    assert r["backend"] == "nccl" and r["rccl_mapped"], "librccl is not mapped: the nccl backend did not load RCCL"


def test_bench_self_launches_two_ranks_with_training_step(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (torch.distributed.run on
    127.0.0.1) and reports n_gpus = 2 with the configs[3] training step inside the timed region. On this 1-GPU box the
    ranks share cuda:0 and the collective runs over gloo (AO_BENCH_SHARE_GPU / AO_BENCH_BACKEND: RCCL refuses two
    ranks on one device); on an N-GPU node the same command runs one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AO_BENCH_SHARE_GPU="1", AO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--games", "64",
           "--sims", "16", "--blocks", "1", "--prefill-games", "3", "--prefill-sims", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 64 * 2 / (d["ms_per_step"] * 2 * 1e-3), rel=1e-6)   # both ranks' move decisions / max time
    ts = d["train_step"]
    assert ts["allreduce_ranks"] == 2 and ts["allreduce_backend"] == "gloo" and ts["train_steps"] == 2 and ts["inside_timed_region"]
    assert ts["allreduce_elements"] > 1000 and ts["mean_loss"] is not None and np.isfinite(ts["mean_loss"])
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and pr["ms_per_step_max"] == max(pr["ms_per_step"]) and pr["slowest_rank"] in (0, 1)
    assert d["ms_per_step"] >= pr["ms_per_step_max"] * 0.999                # the reported time is the max over the ranks
    assert len(pr["trunk_avg_launch_ms"]) == 2 and all(x > 0 for x in pr["trunk_avg_launch_ms"])


@pytest.mark.parametrize("mode", ["DIE", "HANG"])
def test_bench_fails_as_a_whole_when_one_rank_dies_or_hangs(mode):
    """First 8-GPU contact, failure side: rank 1 dies (exit 9) or stops making progress right after the warm-up
    (AO_BENCH_TEST_DIE_RANK / AO_BENCH_TEST_HANG_RANK). The job must END, non-zero, with no bench line: the peers' next barrier
    runs into the steady-state collective timeout (AO_DIST_TIMEOUT), the hung rank's watchdog ends it (exit code 86), the launcher
    reaps the rest. Without the timeouts rank 0 sat in its barrier for gloo's 30 minutes (RCCL: 10)."""
    import subprocess
    import sys
    import time
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AO_BENCH_SHARE_GPU="1", AO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", AO_DIST_TIMEOUT="8", AO_WATCHDOG_S="10")
    env["AO_BENCH_TEST_%s_RANK" % mode] = "1"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--games", "64",
           "--sims", "16", "--blocks", "1", "--prefill-games", "3", "--prefill-sims", "8"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    wall = time.time() - t0
    assert r.returncode != 0, "the job reported success although rank 1 was gone"
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "a bench line was printed by a job that lost a rank"
    # (HANG: whichever fires first ends the job -- rank 0's barrier timing out, here after 8 s, or rank 1's watchdog after 10 s with
    # "made no progress" on stderr; the watchdog's own exit is pinned by tests/test_host_side.py without a launcher in the way)
    # start-up (two imports of torch, engines, replay prefill) + at most timeout / watchdog + the launcher's clean-up
    assert wall < 120.0, wall


def test_bench_eight_rank_launch_shape_on_the_shared_gpu(tmp_path):
    """The driver's 8-GPU command shape (`bench.py --gpus 8`: eight ranks, the configs[3] training step with its gradient
    all-reduce after every step) at tiny sizes on this 1-GPU box -- the ranks share cuda:0, the collectives run over gloo.
    What it pins before an 8-GPU node ever sees the code: ONE JSON line from rank 0 with n_gpus = 8, all eight ranks'
    move decisions in `value`, no deadlock in agree_sums / allreduce_gradients / the barriers with eight participants,
    bit-identical weights on every rank after the training steps, and the per-rank host-thread budget
    (LOCAL_WORLD_SIZE = 8 as torchrun exports it)."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AO_BENCH_SHARE_GPU="1", AO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--games", "32",
           "--sims", "8", "--blocks", "1", "--prefill-games", "2", "--prefill-sims", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(8 * 32 * 2 / (d["ms_per_step"] * 2 * 1e-3), rel=1e-6)
    ts = d["train_step"]
    assert ts["allreduce_ranks"] == 8 and ts["train_steps"] == 2 and ts["inside_timed_region"]
    assert ts["weights_identical_across_ranks"] is True
    assert ts["mean_loss"] is not None and np.isfinite(ts["mean_loss"])
    aff = d.get("host_affinity") or {}
    # (bound to its GPU's NUMA node a rank's share is that node's CPUs / the ranks on it; unbound: all CPUs / LOCAL_WORLD_SIZE)
    assert d["config"]["host_threads_per_rank"] == (aff["host_threads"] if "host_threads" in aff else max(1, min(32, (os.cpu_count() or 1) // 8)))
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and len(pr["trunk_avg_launch_ms"]) == 8 and len(pr["tree_avg_launch_ms"]) == 8
    assert pr["ms_per_step_min"] <= pr["ms_per_step_max"] and 0 <= pr["slowest_rank"] < 8 and 0 <= pr["fastest_rank"] < 8
    assert d["rccl_ranks"]["world_size"] == 8 and d["rccl_ranks"]["ranks_in_allreduce"] == 8
    assert "host_affinity" in d                                             # (what pin_to_gpu_numa did on rank 0; None only at one rank)
