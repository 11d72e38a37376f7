#!/usr/bin/env python3
"""bench.py -- self-play move-decisions/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the largest single-GPU configuration; configs[3] is the same
per GPU x 8): 9x9 Omok, 4096 concurrent self-play games per GPU, 400 MCTS simulations per move,
one leaf per game per wave (leaf batch 4096), random-init 4-block/128-channel PVNet, fp32 results
(the conv contraction runs on fp16 MFMAs with every fp32 operand split in two halves and fp32
accumulation -- error against an fp64 evaluation equal to the fp32-MFMA path's; `--trunk-mode 2`
selects the fp32-MFMA trunk, whose number is also reported as `fp32_mfma_trunk`).
A "step" = one move decision (400 simulations, 401 on a game's first move) for every game of the
rank, then utils.get_action + env step + re-rooting; finished games are reset and re-seeded.
Games shard across ranks with no data-path collective (weak scaling); value = all ranks' move
decisions / max-over-ranks time.

The JSON line also carries:
  roofline      the dominant kernel (the conv stack): algorithmic FLOPs per launch (zero padding
                counted) / average launch duration from HIP events on the launch
                stream, against the dense MFMA peak of the instruction the kernel issues: 2500
                TFLOP/s (fp16) for k_trunk16h, which spends THREE fp16 MFMA products per fp32
                multiply-add (`executed_tflops` = what the matrix pipe actually does), 157.3
                TFLOP/s for the fp32-MFMA kernels. Taps that fall off the board are skipped
                (625 of 729 per 9x9 board do work), so the fp32 kernel's frac can exceed 1.
  cpu_baseline  the sequential per-game search (oracle/ C restatement of agents.py) with the
                PVNet forward on PyTorch-CPU at batch 1 -- what main.self_play does -- timed on
                this host for a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 MFMA
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: ~2.5 PFLOP/s dense fp16 / bf16 MFMA
SUSTAINED_F16_MFMA_TFLOPS = 2050.0  # measured: pure v_mfma_f32_16x16x32_f16 stream, data-like operands (power limit)


def trunk_conv_flops(board, planes, boards):
    """Algorithmic FLOPs of one 3x3 planes->planes convolution launch (2*MAC, zero padding counted
    as in SURVEY.md 8(d): A*9*planes^2 MACs per board)."""
    return 2.0 * board * board * 9 * planes * planes * boards


def eval_flops(board, inplanes, planes, n_block):
    """F_eval of SURVEY.md 8(d): convs + FCs of one PVNet evaluation."""
    A = board * board
    return 2.0 * (A * 9 * inplanes * planes + n_block * 2 * A * 9 * planes * planes + A * planes * 2 +
                  2 * A * A + A * planes + A * planes + planes)


def host_cpu():
    """(hardware threads, CPU model string) of this host."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline(board, sims, n_block, planes, state_dict, budget_s):
    """Sequential reference-style search on the host CPU: oracle tree + torch-CPU net, batch 1."""
    from alpha_omok_amd.pvnet import PVNet
    from oracle import oracle_py as O
    net = PVNet(n_block, 5, planes, board)
    net.load_state_dict(state_dict)
    net.eval()
    def ev(moves, planes_, sim):
        with torch.no_grad():
            p, v = net(torch.from_numpy(planes_[None].copy()))
        return p[0].numpy(), np.float32(v[0].item())

    # batch-1 convolutions do not scale to a whole server socket: the intra-op thread count is picked by a calibration on the
    # REAL loop -- per candidate two 40-simulation searches (oracle tree + callback + forward), after one untimed search -- and
    # every candidate's rate is reported. (Round 4 calibrated on 20 bare forwards: the pick flipped between 8, 16 and 32 threads
    # from run to run and the baseline swung 2 x with it.)
    calib_sims = min(sims, 40)
    calibration = {}
    best = (1, 0.0)
    for nt in (1, 2, 4, 8, 16, 32):
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        cal = O.Agent(board, calib_sims, 5, noise=True, evaluator=ev)
        cal.seed(0)
        cal.get_pi((0,), 1)
        cal.reset()
        t0 = time.perf_counter()
        n_done = 0
        root = (0,)
        for _ in range(2):
            pi, vis, pol = cal.get_pi(root, 1)
            n_done += calib_sims
            root = root + (int(cal.rng.choice_p(pi)),)
        rate = n_done / (time.perf_counter() - t0)
        calibration[str(nt)] = round(rate, 1)
        if rate > best[1]:
            best = (nt, rate)
    torch.set_num_threads(best[0])
    cores = best[0]

    ag = O.Agent(board, sims, 5, noise=True, evaluator=ev)
    ag.seed(0)
    root = (0,)
    ag.get_pi(root, 1)  # warm-up move (also pages torch in); not timed
    t0 = time.perf_counter()
    moves = 0
    while time.perf_counter() - t0 < budget_s and moves < 80:
        pi, vis, pol = ag.get_pi(root, 1 if len(root) <= 6 else 0)
        a = ag.rng.choice_p(pi)
        root = root + (int(a),)
        moves += 1
        if O.check_win(O.get_board(list(root)[1:], board), 3 if board == 3 else 5) != 0:
            ag.reset()
            root = (0,)
    dt = time.perf_counter() - t0
    nproc, cpu_model = host_cpu()
    return dict(value=moves / dt, unit="move-decisions/s", cores=cores, kind="port", host_nproc=nproc, host_cpu_model=cpu_model,
                calibration={"unit": "simulations/s of two %d-simulation searches per intra-op thread count" % calib_sims, "threads": calibration},
                sample="%d move decisions of one 9x9 game, %d sims each, oracle C tree + PyTorch-CPU "
                       "PVNet at batch 1, %d threads, %.1f s" % (moves, sims, cores, dt))


def cpu_worker(board, sims, n_block, planes, sd_path, budget_s):
    """One single-threaded CPU process of the all-cores baseline: independent games, torch at 1 thread.
    A 400-simulation move takes ~30 s on one thread of a loaded host, so the sample is counted in
    SIMULATIONS (searches of `chunk` simulations from successive positions of a game) and scaled to
    move decisions of `sims` simulations by the parent."""
    torch.set_num_threads(1)
    from alpha_omok_amd.pvnet import PVNet
    from oracle import oracle_py as O
    net = PVNet(n_block, 5, planes, board)
    net.load_state_dict(torch.load(sd_path))
    net.eval()

    def ev(moves, planes_, sim):
        with torch.no_grad():
            p, v = net(torch.from_numpy(planes_[None].copy()))
        return p[0].numpy(), np.float32(v[0].item())

    chunk = min(sims, 40)
    ag = O.Agent(board, chunk, 5, noise=True, evaluator=ev)
    ag.seed(os.getpid() & 0xffff)
    root = (0,)
    warm = O.Agent(board, 4, 5, noise=True, evaluator=ev)
    warm.seed(1)
    warm.get_pi(root, 1)  # pages torch in; not timed
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:
        pi, vis, pol = ag.get_pi(root, 1 if len(root) <= 6 else 0)
        done += int(vis.sum()) if len(root) == 1 else chunk
        root = root + (int(ag.rng.choice_p(pi)),)
        if O.check_win(O.get_board(list(root)[1:], board), 3 if board == 3 else 5) != 0:
            ag.reset()
            root = (0,)
    print(json.dumps({"sims": done, "seconds": time.perf_counter() - t0}))


def cpu_all_cores(board, sims, n_block, planes, state_dict, budget_s):
    """the fair all-cores CPU number -- one single-threaded process per core, independent games."""
    import subprocess
    import tempfile
    procs_n = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as d:
        sd_path = os.path.join(d, "sd.pt")
        torch.save(state_dict, sd_path)
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", sd_path, "--board", str(board), "--sims", str(sims),
               "--blocks", str(n_block), "--planes", str(planes), "--cpu-budget", str(budget_s)]
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for _ in range(procs_n)]
        rate = 0.0
        ok = 0
        for pr in procs:
            out, _ = pr.communicate()
            try:
                r = json.loads(out.decode().strip().splitlines()[-1])
                rate += r["sims"] / float(sims) / r["seconds"]
                ok += 1
            except Exception:
                pass
    return dict(value=rate, unit="move-decisions/s (ESTIMATED: simulations/s / %d, searched in chunks of %d simulations -- shallower "
                                 "trees than a %d-simulation move, so an upper bound for this host)" % (sims, min(sims, 40), sims),
                simulations_per_s=rate * sims, chunk_sims=min(sims, 40), processes=ok, threads_per_process=1,
                sample="%d independent single-threaded processes (one per hardware thread), each searching successive "
                       "positions of its own game in chunks of %d simulations for %.0f s; simulations / %d = move decisions"
                       % (ok, min(sims, 40), budget_s, sims))


def tictactoe_bench(device, sims=200, budget_s=3.0):
    """BASELINE configs[0]: 3x3 tic-tac-toe, pure UCT (1_tictactoe_MCTS/mcts_vs.py:134-202), `sims` simulations per
    move. CPU = the oracle's C restatement of that search on one host core, playing whole games; GPU = k_ttt_search
    (one wavefront per board, a whole search per launch) for 1 board (latency) and 4096 boards (throughput)."""
    from alpha_omok_amd import utils
    from alpha_omok_amd.tictactoe import TttEngine
    from oracle import oracle_py as O
    rng = O.PyRandom()
    rng.seed(0)
    t0 = time.perf_counter()
    moves = games = 0
    while time.perf_counter() - t0 < budget_s:
        board = np.zeros((3, 3), np.int8)
        turn = 0
        while True:
            a, q, n = O.ttt_search(board, turn, sims, rng)
            board[a // 3, a % 3] = 1 if turn == 0 else -1
            turn ^= 1
            moves += 1
            if utils.check_win(board.astype(float), 3) != 0:
                break
        games += 1
    cpu_dt = time.perf_counter() - t0
    out = {"workload": "BASELINE configs[0]: 3x3 tic-tac-toe, pure UCT (mcts_vs.py), %d sims/move" % sims,
           "cpu_port": {"value": moves / cpu_dt, "unit": "move-decisions/s", "cores": 1, "kind": "port",
                        "sample": "%d move decisions of %d whole games, oracle C restatement, one thread, %.1f s" % (moves, games, cpu_dt)}}
    rs = np.random.RandomState(0)
    for G in (1, 4096):
        eng = TttEngine(sims, games=G, device=device)
        for g in range(G):
            eng.seed(g, g)
        boards = np.zeros((G, 3, 3), np.int8)
        # a mix of opening positions: 0-2 stones already on the board
        for g in range(G):
            k = g % 3
            cells = rs.permutation(9)[:k]
            for j, c in enumerate(cells):
                boards[g, c // 3, c % 3] = 1 if j % 2 == 0 else -1
        turns = np.array([(g % 3) % 2 for g in range(G)], np.int32)
        eng.search(boards, turns)
        reps = 20 if G == 1 else 5
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.search(boards, turns)
        dt = (time.perf_counter() - t1) / reps
        out["gpu_%d_board%s" % (G, "" if G == 1 else "s")] = {
            "value": G / dt, "unit": "move-decisions/s", "ms_per_launch": dt * 1e3,
            "kernel": "k_ttt_search (one wavefront per board, all %d simulations in one launch)" % sims}
        eng.close()
    return out


def kernel_key(name):
    """'k_trunk16hb<9, 4> (conv1 + ...)' / 'void ao::k_trunk16hb<9, 4>(ao::TrunkHArgs)' -> ('k_trunk16hb', '9'):
    kernel base name + board width (the first template argument), the part every naming of a launch agrees on."""
    n = name.replace("void ", "").replace("ao::", "").strip()
    base = n.split("<")[0].split("(")[0].strip()
    first = n.split("<")[1].split(",")[0].split(">")[0].strip() if "<" in n else ""
    return base, first


def match_traffic(tj, kname, csrc_sha, workload_is_default):
    """The rule that lets a committed rocprofv3 PMC summary (profiles/r*_traffic.json) speak for this run: collected
    from the same kernel sources (hash of csrc/), on the default workload, for the kernel that runs now. Returns
    (dominant-kernel HBM bytes per launch or None, tree-kernel record or None, reason when rejected)."""
    same_code = tj.get("csrc_sha16") == csrc_sha
    if not same_code:
        return None, None, "collected from other kernel sources (csrc hash %s, now %s)" % (tj.get("csrc_sha16"), csrc_sha)
    if not workload_is_default:
        return None, None, "collected on the default workload (4096 games, 9x9, 4 blocks), this run differs"
    tree = tj.get("tree")
    if kernel_key(tj.get("kernel", "")) != kernel_key(kname):
        return None, tree, "collected for kernel %s, this run's dominant kernel is %s" % (tj.get("kernel"), kname.split(" (")[0])
    return tj.get("hbm_bytes_per_launch"), tree, None


def other_workload_traffic(tj, csrc_sha, board, games, blocks, planes, kname):
    """HBM bytes per launch from the profile's `other_workloads` -- records of other workloads collected from the SAME kernel
    sources (e.g. the configs[4] per-GPU shape) -- or None."""
    if not tj or tj.get("csrc_sha16") != csrc_sha:
        return None
    for w in tj.get("other_workloads", []):
        wl = w.get("workload", {})
        if ((wl.get("board"), wl.get("games"), wl.get("blocks"), wl.get("planes")) == (board, games, blocks, planes)
                and kernel_key(w.get("kernel", "")) == kernel_key(kname)):
            return w.get("hbm_bytes_per_launch")
    return None


def newest_traffic_profile():
    import glob
    import re
    files = glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json"))

    def key(p):  # r<round><letter...>: newest round, then newest letter
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(p))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    return sorted(files, key=key)[-1] if files else None


class TrainStep:
    """BASELINE configs[3]'s training step beside the self-play: one mini-batch of 32 from the rank-local replay
    shard (device-resident ring, filled before the timed region by real self-play games of this engine at a small
    simulation count), the reference's loss and Adam step (alpha_omok_amd.main.train_batch = main.py:283-305), ONE
    all-reduce of the flattened fp32 gradient over the ranks, then the updated weights are re-exported to the
    native forward the search uses."""

    def __init__(self, args, model, local, rank, world, net):
        import random
        from alpha_omok_amd import main as M
        self.M, self.net, self.random, self.world = M, net, random, world
        M.PRINT_SELFPLAY = False
        M.configure(board_size=args.board, n_mcts=args.prefill_sims, n_blocks=args.blocks, out_planes=args.planes,
                    seed=1000 + rank, model=model.to(torch.device("cuda", local)), gpu=local, device_replay=True)
        M.rep_memory.clear()
        M.cur_memory.clear()
        M.self_play(args.prefill_games * world)   # episodes shard e % world == rank
        M.cur_memory.clear()
        M.release_engine()
        random.seed(2000 + rank)
        M.Agent.model.train()
        self.numel = sum(p.numel() for p in M.Agent.model.parameters())
        self.ms, self.export_ms, self.losses = [], [], []
        # the samples all ranks put behind a step, agreed ONCE (the shards do not change during the bench): the per-step
        # gradient all-reduce then reads nothing back from the device (parallel.allreduce_gradients, `total`)
        from alpha_omok_amd import parallel
        # (the batch size is frozen with the divisor: the shards GROW during the timed loop -- self-play appends -- and a rank that
        # started below BATCH_SIZE would otherwise draw larger batches than the agreed total accounts for)
        self.batch_n = min(M.BATCH_SIZE, len(M.rep_memory))
        self.total = parallel.agree_sums([self.batch_n], torch.device("cuda", local))[0] if world > 1 else None

    def step(self, record):
        M = self.M
        t0 = time.perf_counter()
        n = len(M.rep_memory)
        batch = self.random.sample(range(n), min(self.batch_n, n))
        out = M.train_batch(batch, self.total)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.net.load_state_dict(M.Agent.model.state_dict())
        t2 = time.perf_counter()
        if record:
            self.ms.append((t1 - t0) * 1e3)
            self.export_ms.append((t2 - t1) * 1e3)
            if out is not None:
                self.losses.append(float(out[0]))

    def allreduce_probe(self, dist, dev):
        """Isolated cost of the step's collective: the same flattened buffer, 20 repetitions."""
        if dist is None:
            return None
        flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=dev if dist.get_backend() == "nccl" else "cpu")
        for _ in range(3):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3

    def report(self, dist, dev, ms_per_step):
        M = self.M
        r = {"workload": "one batch-%d training step per GPU after every step (refill cycle), rank-local replay shard, "
                         "one all-reduce of the flattened gradient (BASELINE configs[3])" % M.BATCH_SIZE,
             "train_steps": len(self.ms), "ms_per_train_step": float(np.mean(self.ms)) if self.ms else None,
             "ms_weight_export": float(np.mean(self.export_ms)) if self.export_ms else None,
             "allreduce_elements": self.numel + 1, "allreduce_bytes": 4 * (self.numel + 1),
             "allreduce_backend": (dist.get_backend() if dist is not None else None), "allreduce_ranks": self.world,
             "allreduce_ms_isolated": self.allreduce_probe(dist, dev),
             "replay_entries_per_gpu": len(M.rep_memory), "mean_loss": float(np.mean(self.losses)) if self.losses else None,
             "inside_timed_region": True}
        # every rank applied the same all-reduced gradient with the same Adam state: the weights must still be
        # bit-identical across the ranks (two 64-bit checksums of the parameter BITS, gathered over the group)
        if dist is not None:
            bits = torch.cat([p.detach().reshape(-1) for p in M.Agent.model.parameters()]).view(torch.int32).to(torch.int64)
            idx = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
            mine = torch.stack([bits.sum(), (bits * (idx % 8191)).sum()])
            mine = mine if dist.get_backend() == "nccl" else mine.cpu()
            got = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(got, mine)
            r["weights_identical_across_ranks"] = bool(all(torch.equal(g, got[0]) for g in got))
        if self.ms:
            r["share_of_step"] = (float(np.mean(self.ms)) + float(np.mean(self.export_ms))) / ms_per_step
        return r


# HIP-event pairs around EVERY launch cost ~1 % of a step (tools/time_move_phases.py --events): every 8th launch is timed;
# the averages are the same estimate, shares are avg x launches issued
TIMING_STRIDE = 8
TRAINED_CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r4_trained_9x9_4block.pt")
# a checkpoint trained FROM SCRATCH with tools/train_omok.py --fp16-grid-weights (round 6, tools/exp/r6j.sh): its conv weights are fp16 numbers as saved
TRAINED_GRID_CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r6j_trained_fp16grid_9x9_4block.pt")


def trained_net_bench(args, local, path, steps=8, warm_plies=8, oversubscribe=1.25, fp16_grid=False):
    """The headline workload with a TRAINED network (a checkpoint of tools/train_omok.py: this engine's own self-play +
    main.train on the MI355X) instead of random-init weights: sharp priors, so the searches go deep, meet terminal leaves
    and keep most of the tree from move to move -- the regime the tree kernels exist for. Fresh engine, `warm_plies`
    untimed move decisions (the tau switch at ply 6 included), then `steps` timed ones with refills.

    Two runs on the same box, back to back:
      * `static_rows`: G games, the evaluation batch packed per move by the host (rounds 3 - 4): a game whose leaf is terminal
        keeps its stale row in the batch, so 11 - 19 % of the trunk's rows are junk with this network;
      * the result proper: `oversubscribe` x G games on the same G rows per simulation, handed out by the tree kernel per
        simulation (ao_set_row_cap(G): terminal leaves take none; a share of the games sits out each launch in turn) -- the
        trunk launch is the same 256 groups, but every row is a live leaf. A step is still one move decision of every game.

    fp16_grid (the `trained_net_fp16grid` leg): the same checkpoint with its 3x3 conv weights rounded to the fp16 grid -- what
    tools/train_omok.py --fp16-grid-weights maintains during training (a relative change of <= 2^-11 per weight) -- so that the
    library plans the TWO-product split-fp16 kernels (ao_net_products); only the over-subscribed run is made."""
    from alpha_omok_amd.engine import Engine
    from alpha_omok_amd.pvnet import PVNet
    B, S, G = args.board, args.sims, args.games
    model = PVNet(args.blocks, 5, args.planes, B)
    model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    already_on_grid = False
    if fp16_grid:
        with torch.no_grad():
            convs = [p for p in model.parameters() if p.dim() == 4 and p.shape[2] == 3]
            already_on_grid = all(torch.equal(p.to(torch.float16).to(p.dtype), p) for p in convs)
            for p in convs:
                p.copy_(p.to(torch.float16).to(p.dtype))
    model.eval()
    net = model.to_native(local)
    products, weights_fp16 = net.products()
    if fp16_grid and products != 2:
        raise RuntimeError("the fp16-grid weights did not plan the two-product kernels (products %d, weights_fp16 %r)" % (products, weights_fp16))
    A_ = B * B

    def run(games, row_cap):
        # more trees than the default arena rule plans for: half of the HBM for them (ao_config.arena_fraction; at most 16 x (sims + 1) nodes each)
        eng = Engine(B, S, 5, games=games, noise=True, device=local, arena_fraction=0.5 if games > G else 0.0)
        if row_cap:
            eng.set_row_cap(row_cap)
        eng.seed_all(np.arange(7 * G, 7 * G + games, dtype=np.uint32))
        ply = np.zeros(games, np.int64)
        nxt = [9 * G]
        c = dict(levels=0, evaluated=0, terminal=0, games=0)

        def one(count):
            eng.search(net, tau=(ply < 6).astype(np.int8))
            st = eng.search_stats()
            act, win = eng.play()
            ply[:] += 1
            done = win != 0
            if count:
                for k in ("levels", "evaluated", "terminal"):
                    c[k] += st[k]
                c["games"] += int(done.sum())
            if done.any():
                eng.reset(done.astype(np.uint8))
                gs = np.nonzero(done)[0]
                eng.seed_games(gs, np.arange(nxt[0], nxt[0] + gs.size))
                nxt[0] += gs.size
                ply[done] = 0

        for _ in range(warm_plies):
            one(False)
        eng.tree_timing(TIMING_STRIDE)
        net.conv_timing(TIMING_STRIDE)
        rs0 = eng.row_stats()
        eng.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(True)
        eng.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        conv_ms, conv_n = net.conv_timing(False)
        tree_ms, tree_n = eng.tree_timing(False)
        rs1 = eng.row_stats()
        sims_total = max(c["evaluated"] + c["terminal"], 1)
        d_bar = c["levels"] / sims_total
        tree_bytes = games * (A_ * (16.0 * d_bar + 44.0) + 24.0 * (d_bar + 1.0) + 4.0)   # SURVEY.md 8(d), C = 5
        tree_avg = tree_ms / max(tree_n, 1)
        dropped, trimmed = eng.trim_stats()
        ev, evg = eng.fp16_range_events()
        kname, f_launch = net.dominant_kernel(row_cap or games)
        launches = rs1["launches"] - rs0["launches"]
        r = {"games": games, "rows_per_simulation": row_cap or games,
             "rows": "handed out per simulation by the tree kernel (ao_set_row_cap)" if row_cap else "packed per move by the host",
             "value": games * steps / dt, "unit": "move-decisions/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
             "mean_select_depth": d_bar, "terminal_leaf_fraction": c["terminal"] / sims_total, "games_finished": c["games"],
             # rows the trunk really evaluated per simulation of a game: 1 with the per-move packing (terminal leaves keep a stale
             # row), 1 - terminal share when rows are handed out per simulation
             "rows_evaluated_per_simulation": (rs1["rows_live"] - rs0["rows_live"]) / sims_total if row_cap else 1.0,
             "trunk_kernel": kname.split(" (")[0], "trunk_avg_launch_ms": conv_ms / max(conv_n, 1),
             "trunk_time_share": conv_ms * TIMING_STRIDE * 1e-3 / dt,
             # the conv stack against the dense fp16 MFMA peak, computed as the headline's roofline is: algorithmic FLOPs of a launch
             # at the batch's CAPACITY (zero padding and empty rows counted as the headline counts them) / its average duration
             "roofline": {"bound": "mfma", "kernel": kname.split(" (")[0], "products": products, "flop_per_launch": f_launch,
                          "avg_launch_ms": conv_ms / max(conv_n, 1), "launches_timed": conv_n,
                          "achieved": f_launch / (conv_ms / max(conv_n, 1) * 1e-3) / 1e12 if conv_n else 0.0, "peak": PEAK_F16_MFMA_TFLOPS,
                          "unit": "TFLOP/s", "frac": f_launch / (conv_ms / max(conv_n, 1) * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS if conv_n else 0.0,
                          "executed_tflops": products * f_launch / (conv_ms / max(conv_n, 1) * 1e-3) / 1e12 if conv_n else 0.0},
             "roofline_tree": {"kernel": "k_expand_select", "avg_launch_ms": tree_avg, "launches_timed": tree_n,
                               "algorithmic_bytes_per_launch": tree_bytes,
                               "achieved": tree_bytes / (tree_avg * 1e-3) / 1e9 if tree_n else 0.0, "unit": "GB/s", "peak": 8000.0,
                               "frac": tree_bytes / (tree_avg * 1e-3) / 1e9 / 8000.0 if tree_n else 0.0,
                               "time_share": tree_ms * TIMING_STRIDE * 1e-3 / dt},
             "node_cap": eng.node_cap()[0], "arena_trims": {"subtrees_dropped": dropped, "reroots_trimmed": trimmed},
             "fp16_range_events": ev}
        if row_cap:
            r["network_launches_per_move"] = launches / max(steps, 1)
            r["batch_fill"] = (rs1["rows_live"] - rs0["rows_live"]) / max(rs1["rows_launched"] - rs0["rows_launched"], 1)
            r["leaves_that_waited_a_launch"] = rs1["waits"] - rs0["waits"]
        eng.close()
        return r

    static = run(G, 0) if not (fp16_grid and oversubscribe and oversubscribe > 1.0) else None
    over = run(int(round(G * oversubscribe / 256.0)) * 256, G) if oversubscribe and oversubscribe > 1.0 else None
    r = dict(over if over else static)
    r["weights"] = os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
    r["mfma_products"] = products
    if fp16_grid:
        r["weights"] += (" (trained with tools/train_omok.py --fp16-grid-weights: its 3x3 conv weights are fp16 numbers as saved)" if already_on_grid else
                         " with the 3x3 conv weights rounded to the fp16 grid (tools/train_omok.py --fp16-grid-weights keeps them there)")
    r["workload"] = ("the headline's sims / rows per simulation, network = the committed checkpoint trained by this engine "
                     "(tools/train_omok.py); plies %d-%d timed" % (warm_plies, warm_plies + steps - 1))
    if over and static:
        r["static_rows"] = static
        r["vs_static_rows"] = over["value"] / static["value"]
    net.close()
    return r


def launch_ranks(n):
    """Re-run this script as n ranks under torch.distributed.run (rendezvous on 127.0.0.1). Fails loudly -- exit code
    2, nothing printed on stdout -- when fewer than n GPUs are visible, instead of reporting a 1-GPU number under an
    N-GPU label."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not os.environ.get("AO_BENCH_SHARE_GPU"):
        print("bench.py: --gpus %d requested but %d GPU(s) are visible on this node; refusing to report an %d-GPU number "
              "(AO_BENCH_SHARE_GPU=1 AO_BENCH_BACKEND=gloo runs the ranks on shared devices for a functional check)"
              % (n, have, n), file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed move decisions per game slot (20 steps from ply 5 on: ~750 of 4096 games finish and are refilled inside the timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--planes", type=int, default=128)
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU baseline sample (>= 30 move decisions on the GPU box's host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-game", action="store_true")
    ap.add_argument("--no-tictactoe", action="store_true", help="skip BASELINE configs[0] (3x3 UCT, 200 sims): oracle on one core beside k_ttt_search")
    ap.add_argument("--trunk-mode", type=int, default=0,
                    help="0 auto, 1 layer kernels, 2 group-resident fp32-MFMA trunk, 3 per-board, 4 row-chunked, "
                         "5 group-resident split-fp16 trunk")
    ap.add_argument("--no-fp32-compare", action="store_true", help="skip the extra fp32-MFMA-trunk measurement")
    ap.add_argument("--no-ten-block", action="store_true",
                    help="skip the extra measurement with the reference's default net (N_BLOCKS = 10, main.py:33)")
    ap.add_argument("--cpu-all-cores", action="store_true", help="(kept for compatibility: the all-cores sample is on by default)")
    ap.add_argument("--train-step", choices=("auto", "on", "off"), default="auto",
                    help="BASELINE configs[3]: after every step (= refill cycle) one batch-32 training step per GPU "
                         "from the rank-local replay shard with ONE all-reduce of the flattened gradient; "
                         "auto = on when more than one GPU takes part")
    ap.add_argument("--prefill-sims", type=int, default=16, help="simulations per move of the replay-filling games")
    ap.add_argument("--prefill-games", type=int, default=48, help="self-play games per GPU that fill the replay shard (untimed)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the one-process-per-hardware-thread CPU sample")
    ap.add_argument("--cpu-all-cores-budget", type=float, default=6.0, help="seconds per process of the all-cores sample")
    ap.add_argument("--trained-weights", default=TRAINED_CKPT, help="state_dict of a trained 9x9 / 4-block / 128-plane PVNet for the `trained_net` leg")
    ap.add_argument("--oversubscribe", type=float, default=1.25, help="`trained_net` leg: games per row of the evaluation batch (1.25: 5120 games on "
                    "the 4096 rows per simulation the tree kernel hands out; 1: the per-move packing only)")
    ap.add_argument("--no-wide-board", action="store_true", help="skip the `wide_board` leg (BASELINE configs[4]'s per-GPU shape: 15x15, 10 blocks, 800 sims, 1024 games)")
    ap.add_argument("--no-fp16-grid", action="store_true", help="skip the `trained_net_fp16grid` leg (the trained checkpoint with its conv weights on the fp16 grid: two-product kernels)")
    ap.add_argument("--no-trained-net", action="store_true", help="skip the `trained_net` leg (the headline workload with trained weights)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.board, args.sims, args.blocks, args.planes, args.cpu_worker, args.cpu_budget)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL), exactly
        # as the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` would
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s): reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    from alpha_omok_amd import parallel
    affinity = None
    # a rank that dies or hangs must fail the JOB, quickly: the process group's collective timeout takes the ranks that wait for it
    # down (AO_DIST_TIMEOUT, 40 s once the warm-up is over; AO_DIST_INIT_TIMEOUT, 300 s, for rendezvous, RCCL's communicator set-up and
    # a cold box's page-in), a watchdog thread ends a rank that itself stops making progress (AO_WATCHDOG_S, 50 s / 300 s at start-up)
    steady_timeout = float(os.environ.get("AO_DIST_TIMEOUT", "40"))
    steady_stall = float(os.environ.get("AO_WATCHDOG_S", "50"))
    init_timeout = float(os.environ.get("AO_DIST_INIT_TIMEOUT", "300"))
    dog = parallel.Watchdog(max(init_timeout, steady_stall) if world > 1 else 0.0, "start-up")
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # AO_BENCH_BACKEND=gloo + AO_BENCH_SHARE_GPU=1 exercise this path with two ranks on a 1-GPU box
        backend = os.environ.get("AO_BENCH_BACKEND", "nccl")
        if os.environ.get("AO_BENCH_SHARE_GPU"):
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        # this rank's threads (the engine's host pool, torch's) on the NUMA node of its GPU (AO_NO_AFFINITY=1: off)
        affinity = parallel.pin_to_gpu_numa(local)
        timeout = datetime.timedelta(seconds=init_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=timeout)
        else:
            dist.init_process_group(backend, timeout=timeout)
    dog.__enter__()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from alpha_omok_amd.engine import Engine
    from alpha_omok_amd.pvnet import PVNet

    B, S, G = args.board, args.sims, args.games
    torch.manual_seed(0)
    model = PVNet(args.blocks, 5, args.planes, B)  # random init, BN gamma 1 / beta 0 (model.py:86-89)
    model.eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    net = model.to_native(local)
    net.set_mode(args.trunk_mode)
    eng = Engine(B, S, 5, games=G, noise=True, device=local)
    base_seed = rank * G
    eng.seed_all(np.arange(base_seed, base_seed + G, dtype=np.uint32))
    next_seed = [world * G + rank]
    ply = np.zeros(G, np.int64)
    counters = dict(moves=0, games=0, levels=0, evaluated=0, terminal=0)

    def step(count, use_net=None):
        tau = (ply < 6).astype(np.int8)  # main.py:150-153 TAU_THRES
        eng.search(use_net if use_net is not None else net, tau=tau)
        st = eng.search_stats()
        act, win = eng.play()
        ply[:] += 1
        done = win != 0
        if count:
            counters["moves"] += G
            counters["levels"] += st["levels"]
            counters["evaluated"] += st["evaluated"]
            counters["terminal"] += st["terminal"]
            counters["games"] += int(done.sum())
        if done.any():  # refill finished slots with fresh games (Agent.reset(), main.py:248)
            eng.reset(done.astype(np.uint8))
            gs = np.nonzero(done)[0]
            eng.seed_games(gs, next_seed[0] + world * np.arange(gs.size))
            next_seed[0] += world * gs.size
            ply[done] = 0

    dog.beat("engine and network built")
    train_on = args.train_step == "on" or (args.train_step == "auto" and world > 1)
    trainer = TrainStep(args, model, local, rank, world, net) if train_on else None
    dog.beat("replay shard filled")
    ncycle = [0]

    def cycle(count):
        step(count)
        if trainer is not None:
            trainer.step(count)
        ncycle[0] += 1
        dog.beat("after step %d (%s)" % (ncycle[0], "timed" if count else "warm-up"))

    for _ in range(args.warmup):
        cycle(False)
    if dist is not None:
        dist.barrier()                                    # every rank is through its start-up: from here on a silent peer is a dead peer
        parallel.set_collective_timeout(steady_timeout)
        dog.stall_s = steady_stall
        dog.beat("warm-up over, steady-state timeouts in force")
    # developer switches of tests/test_gpu_multirank.py: rank r dies (exit code 9) / stops making progress after the warm-up
    if os.environ.get("AO_BENCH_TEST_DIE_RANK") == str(rank):
        os._exit(9)
    if os.environ.get("AO_BENCH_TEST_HANG_RANK") == str(rank):
        time.sleep(3600)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.tree_timing(TIMING_STRIDE)
    net.conv_timing(TIMING_STRIDE)  # reset + enable HIP-event timing of the trunk conv launches (every TIMING_STRIDE-th)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle(True)
    fence()
    dt = time.perf_counter() - t0
    conv_ms, conv_launches = net.conv_timing(False)
    tree_ms, tree_launches = eng.tree_timing(False)

    t = torch.tensor([dt], dtype=torch.float64, device=dev if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    dog.beat("timed region over")
    # per-rank view of the same run (one small all-gather): the first multi-GPU record should show stragglers and clock spread,
    # not just the maximum
    per_rank = None
    if dist is not None:
        mine = torch.tensor([dt / args.steps * 1e3, conv_ms / max(conv_launches, 1), tree_ms / max(tree_launches, 1)], dtype=torch.float64,
                            device=dev if dist.get_backend() == "nccl" else "cpu")
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        rows = [[float(x) for x in g.cpu().tolist()] for g in got]
        ms = [r_[0] for r_ in rows]
        per_rank = {"ms_per_step": ms, "ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "slowest_rank": int(np.argmax(ms)),
                    "fastest_rank": int(np.argmin(ms)), "trunk_avg_launch_ms": [r_[1] for r_ in rows],
                    "tree_avg_launch_ms": [r_[2] for r_ in rows]}
    # what the collective library itself saw during the timed run: the group's size and backend, and the sum over the group of one
    # 1 per rank (the reduction just above ran on the same group) -- so that an N-GPU line proves RCCL ("nccl") had N ranks
    rccl_ranks = None
    if dist is not None:
        ones = torch.ones(1, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl_ranks = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks_in_allreduce": int(round(float(ones.item())))}
    total_moves = world * G * args.steps
    value = total_moves / dt_max
    train_report = trainer.report(dist, dev, dt_max / args.steps * 1e3) if trainer is not None else None  # (collective)

    if rank == 0:
        kname, f_launch = net.dominant_kernel(G)
        avg_ms = conv_ms / max(conv_launches, 1)
        achieved = f_launch / (avg_ms * 1e-3) / 1e12 if conv_launches else 0.0
        # HBM bytes per launch come from rocprofv3 PMC passes (they cannot be collected inside this process): the
        # newest committed summary is used when it was collected from THIS kernel source (csrc hash) on this workload
        traffic = None
        traffic_src = None
        tree_traffic = None
        tj = None
        traffic_why = "no profile"
        from alpha_omok_amd.build import source_hash
        csrc_sha = source_hash()
        try:
            traffic_src = newest_traffic_profile()
            with open(traffic_src) as f:
                tj = json.load(f)
            same_load = (G == 4096 and B == 9 and args.blocks == 4 and args.planes == 128 and S == 400)
            traffic, tree_traffic, traffic_why = match_traffic(tj, kname, csrc_sha, same_load)
            if traffic is None and not same_load:
                t2 = other_workload_traffic(tj, csrc_sha, B, G, args.blocks, args.planes, kname)
                if t2 is not None:
                    traffic, traffic_why = t2, None
        except Exception as e:
            traffic, traffic_why = None, "profile unreadable: %r" % (e,)
        sims_total = max(counters["evaluated"] + counters["terminal"], 1)
        split16 = kname.startswith(("k_trunk16h", "k_layer16h", "k_row16hk", "k_boardh", "k_conv_cells_h"))
        peak = PEAK_F16_MFMA_TFLOPS if split16 else PEAK_F32_MFMA_TFLOPS
        out = {
            "metric": "self-play move-decisions/sec (%dx%d, %d sims/move)" % (B, B, S),
            "value": value,
            "unit": "move-decisions/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "per_rank": per_rank,
            "host_affinity": affinity,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # the arithmetic type AND, for the resident split-fp16 trunk, the precision its activations are STORED with between
            # layers (kPairBytes: two fp16 halves = ~22 significand bits, or fp16 high half + one low byte = 19)
            "dtype": ("f32 (fp16x2-split MFMA operands, 3 products per multiply-add, f32 accumulate; inter-layer activations stored as "
                      + ("fp16 high half + 8 low bits: 19 significand bits)" if kname.startswith("k_trunk16h") and ", 1> (" in kname
                         else "two fp16 halves: ~22 significand bits)")) if split16 else "f32",
            "data": "synthetic (self-play from the empty board, random-init weights, per-game seeds)",
            "config": {
                "workload": "%s: %dx%d Omok, %d concurrent self-play games per GPU, %d sims/move, "
                            "leaf batch %d, random-init %d-block/%d-ch PVNet" % (
                                "BASELINE configs[3] (per-GPU shape of configs[2] + training step with gradient all-reduce)"
                                if (B, G, S, args.blocks) == (9, 4096, 400, 4) and train_on else
                                "BASELINE configs[2]" if (B, G, S, args.blocks) == (9, 4096, 400, 4) else
                                "BASELINE configs[4] per-GPU shape" if (B, G, S, args.blocks) == (15, 1024, 800, 10) else
                                "custom shape", B, B, G, S, G, args.blocks, args.planes),
                "games_per_gpu": G, "sims": S, "board": B, "n_block": args.blocks, "planes": args.planes,
                "parallelism": "games sharded over %d GPU(s), no data-path collective in self-play%s" % (
                    world, "; one gradient all-reduce per training step" if train_on else ""),
                "host_threads_per_rank": int(__import__("alpha_omok_amd._lib", fromlist=["load"]).load().ao_host_threads()),
                "mean_select_depth": counters["levels"] / sims_total,
                "terminal_leaf_fraction": counters["terminal"] / sims_total,
                "games_finished": counters["games"],
                "flops_per_move_algorithmic": S * eval_flops(B, 5, args.planes, args.blocks),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": kname,
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "mfma_dtype": "f16" if split16 else "f32",
                "executed_tflops": achieved * 3.0 if split16 else achieved,
                # orientation beside `frac` (which prices the ALGORITHMIC fp32 FLOPs against the nominal fp16 peak):
                # the MFMAs really issued (taps that fall off the board are skipped: (3B-2)^2 of (3B)^2) against the
                # rate the pipe sustains on data-like operands under the chip's power limit, measured with
                # tools/mfma_peak.hip (profiles/r1j_mfma_peak_microbench.txt), and the fp32-MFMA peak the same
                # algorithmic work would be priced against if it ran on v_mfma_f32_16x16x4_f32
                "issued_tflops": achieved * (3.0 if split16 else 1.0) * ((3 * B - 2) ** 2) / float((3 * B) ** 2),
                "sustained_peak_measured": SUSTAINED_F16_MFMA_TFLOPS if split16 else 156.0,
                "issued_frac_of_sustained": achieved * (3.0 if split16 else 1.0) * ((3 * B - 2) ** 2) / float((3 * B) ** 2)
                                            / (SUSTAINED_F16_MFMA_TFLOPS if split16 else 156.0),
                "algorithmic_vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "bytes of HBM traffic per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, %s)" % (
                    (os.path.relpath(traffic_src, REPO) + ("" if traffic is not None else ": NOT used, " + str(traffic_why)))
                    if traffic_src else "no profile"),
                "csrc_sha16": csrc_sha,
                "flop_per_launch": f_launch,
                "avg_launch_ms": avg_ms,
                "launches_timed": conv_launches,
                "conv_time_share": (conv_ms * TIMING_STRIDE * 1e-3) / dt if dt > 0 else None,
            },
        }
        # the tree side of the path (north_star: HBM GB/s of the tree kernels): k_expand_select = expansion + backup of
        # simulation i and selection + terminal test + plane encoding of simulation i+1 for every game, one launch
        d_bar = counters["levels"] / sims_total
        A_ = B * B
        tree_bytes = G * (A_ * (16.0 * d_bar + 44.0) + 24.0 * (d_bar + 1.0) + 4.0)   # SURVEY.md 8(d), C = 5
        tree_avg_ms = tree_ms / max(tree_launches, 1)
        tree_gbs = tree_bytes / (tree_avg_ms * 1e-3) / 1e9 if tree_launches else 0.0
        out["roofline_tree"] = {
            "bound": "hbm", "kernel": "k_expand_select (one wavefront per game, %d games per workgroup)" % 4,
            "achieved": tree_gbs, "peak": 8000.0, "unit": "GB/s", "frac": tree_gbs / 8000.0,
            "algorithmic_bytes_per_launch": tree_bytes, "bytes_per_sim_per_game": tree_bytes / G,
            "mean_select_depth": d_bar, "avg_launch_ms": tree_avg_ms, "launches_timed": tree_launches,
            "time_share": (tree_ms * TIMING_STRIDE * 1e-3) / dt if dt > 0 else None,
            "traffic": (tree_traffic or {}).get("hbm_bytes_per_launch"),
            "traffic_over_algorithmic": ((tree_traffic or {}).get("hbm_bytes_per_launch") / tree_bytes
                                         if tree_traffic and tree_traffic.get("hbm_bytes_per_launch") else None),
            "note": "latency-bound, not bandwidth-bound: each game's descent is a chain of dependent row loads "
                    "(one wave per game), so GB/s stays far below the HBM peak by construction",
        }
        if train_report is not None:
            out["train_step"] = train_report
        if world == 1 and split16 and not args.no_fp32_compare:
            # the same workload with the fp32-MFMA trunk (k_trunk16), for comparison
            try:
                net.set_mode(2)
                step(False)
                fence()
                t1 = time.perf_counter()
                for _ in range(2):
                    step(False)
                fence()
                d1 = time.perf_counter() - t1
                out["fp32_mfma_trunk"] = {"kernel": "k_trunk16 (fp32 MFMA 16x16x4)", "value": 2 * G / d1,
                                          "unit": "move-decisions/s", "ms_per_step": d1 / 2 * 1e3}
            except Exception as e:
                out["fp32_mfma_trunk"] = {"value": None, "error": repr(e)}
            net.set_mode(args.trunk_mode)
        if world == 1 and args.blocks != 10 and not args.no_ten_block:
            # the reference's own default network (main.py:33 N_BLOCKS = 10) on the same workload, five steps
            try:
                torch.manual_seed(0)
                m10 = PVNet(10, 5, args.planes, B)
                m10.eval()
                net10 = m10.to_native(local)
                step(False, net10)
                fence()
                t1 = time.perf_counter()
                n10 = 5
                for _ in range(n10):
                    step(False, net10)
                fence()
                d1 = time.perf_counter() - t1
                out["ten_block_net"] = {"workload": "same games, random-init 10-block/%d-ch PVNet (the reference's default, main.py:33)" % args.planes,
                                        "value": n10 * G / d1, "unit": "move-decisions/s", "ms_per_step": d1 / n10 * 1e3, "steps": n10,
                                        "flops_per_move_algorithmic": S * eval_flops(B, 5, args.planes, 10)}
                net10.close()
            except Exception as e:
                out["ten_block_net"] = {"value": None, "error": repr(e)}
        # (the main engine's trees -- 124 GB of arena at 4096 games -- are not needed by any leg below: free them before the
        # legs that build engines of their own)
        eng.close()
        if world == 1 and not args.no_wide_board and (B, G, S) == (9, 4096, 400) and args.planes == 128:
            # BASELINE configs[4]'s per-GPU shape beside the headline: 15x15 (env_regular), 10-block net, 800 sims, 1024 concurrent games
            # (8192 games on 8 GPUs = 1024 per GPU); 1 untimed + 2 timed move decisions from the empty board. Kernel: k_boardh<15, 2>.
            try:
                from alpha_omok_amd.engine import Engine
                Bw, Gw, Sw = 15, 1024, 800
                torch.manual_seed(0)
                mw = PVNet(10, 5, args.planes, Bw)
                mw.eval()
                netw = mw.to_native(local)
                engw = Engine(Bw, Sw, 5, games=Gw, noise=True, device=local)
                engw.seed_all(np.arange(11 * Gw, 12 * Gw, dtype=np.uint32))

                def wstep():
                    engw.search(netw, tau=np.ones(Gw, np.int8))
                    engw.play()
                wstep()
                netw.conv_timing(True)
                engw.sync()
                torch.cuda.synchronize()
                tw0 = time.perf_counter()
                nw = 2
                for _ in range(nw):
                    wstep()
                engw.sync()
                torch.cuda.synchronize()
                dw = time.perf_counter() - tw0
                cw_ms, cw_n = netw.conv_timing(False)
                kw, fw = netw.dominant_kernel(Gw)
                aw = cw_ms / max(cw_n, 1)
                out["wide_board"] = {"workload": "BASELINE configs[4] per-GPU shape: 15x15 Omok, %d concurrent games, %d sims/move, random-init 10-block/%d-ch PVNet" % (Gw, Sw, args.planes),
                                     "value": nw * Gw / dw, "unit": "move-decisions/s", "ms_per_step": dw / nw * 1e3, "steps": nw,
                                     "flops_per_move_algorithmic": Sw * eval_flops(Bw, 5, args.planes, 10),
                                     "roofline": {"bound": "mfma", "kernel": kw.split(" (")[0], "flop_per_launch": fw, "avg_launch_ms": aw, "launches_timed": cw_n,
                                                  "achieved": fw / (aw * 1e-3) / 1e12 if cw_n else 0.0, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                  "frac": fw / (aw * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS if cw_n else 0.0,
                                                  "traffic": other_workload_traffic(tj, csrc_sha, Bw, Gw, 10, args.planes, kw)}}
                if not args.no_fp16_grid:
                    # the same shape on weights rounded to the fp16 grid: k_boardh_w16, two products per multiply-add (fresh games, same seeds)
                    with torch.no_grad():
                        for p_ in mw.parameters():
                            if p_.dim() == 4 and p_.shape[2] == 3:
                                p_.copy_(p_.to(torch.float16).to(p_.dtype))
                    netw = mw.to_native(local, netw)
                    engw.reset()
                    engw.seed_all(np.arange(11 * Gw, 12 * Gw, dtype=np.uint32))
                    wstep()
                    netw.conv_timing(True)
                    engw.sync()
                    torch.cuda.synchronize()
                    tw0 = time.perf_counter()
                    for _ in range(nw):
                        wstep()
                    engw.sync()
                    torch.cuda.synchronize()
                    dw2 = time.perf_counter() - tw0
                    c2_ms, c2_n = netw.conv_timing(False)
                    k2, f2 = netw.dominant_kernel(Gw)
                    a2 = c2_ms / max(c2_n, 1)
                    out["wide_board"]["fp16grid"] = {
                        "weights": "the same random-init network with its 3x3 conv weights rounded to the fp16 grid", "mfma_products": netw.products()[0],
                        "value": nw * Gw / dw2, "unit": "move-decisions/s", "ms_per_step": dw2 / nw * 1e3, "vs_three_products": dw / dw2,
                        "roofline": {"bound": "mfma", "kernel": k2.split(" (")[0], "products": netw.products()[0], "flop_per_launch": f2, "avg_launch_ms": a2,
                                     "launches_timed": c2_n, "achieved": f2 / (a2 * 1e-3) / 1e12 if c2_n else 0.0, "peak": PEAK_F16_MFMA_TFLOPS,
                                     "unit": "TFLOP/s", "frac": f2 / (a2 * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS if c2_n else 0.0}}
                engw.close()
                netw.close()
            except Exception as e:
                out.setdefault("wide_board", {"value": None})["error"] = repr(e)
        if world == 1 and not args.no_trained_net and os.path.exists(args.trained_weights) and (B, args.blocks, args.planes) == (9, 4, 128):
            try:
                out["trained_net"] = trained_net_bench(args, local, args.trained_weights, oversubscribe=args.oversubscribe)
            except Exception as e:
                out["trained_net"] = {"value": None, "error": repr(e)}
            if not args.no_fp16_grid:
                # the same leg on weights that sit on the fp16 grid: two MFMA products per multiply-add instead of three. NOT the
                # headline (which stays the arbitrary-fp32 random-init network on three products); reported beside `trained_net`
                try:
                    leg = trained_net_bench(args, local, args.trained_weights, oversubscribe=args.oversubscribe, fp16_grid=True)
                    if out["trained_net"].get("value"):
                        leg["vs_trained_net"] = leg["value"] / out["trained_net"]["value"]
                    out["trained_net_fp16grid"] = leg
                    if os.path.exists(TRAINED_GRID_CKPT):
                        # ... and a network that was TRAINED on the grid (no rounding at load time: the file's conv weights are fp16 numbers);
                        # another network = other search depths, so this is a second data point, not the A/B of the line above
                        own = trained_net_bench(args, local, TRAINED_GRID_CKPT, steps=6, oversubscribe=args.oversubscribe, fp16_grid=True)
                        leg["trained_on_the_grid"] = {k: own.get(k) for k in ("weights", "value", "unit", "ms_per_step", "mfma_products", "mean_select_depth",
                                                                               "terminal_leaf_fraction", "trunk_kernel", "trunk_avg_launch_ms", "roofline")}
                except Exception as e:
                    out["trained_net_fp16grid"] = {"value": None, "error": repr(e)}
        if world == 1 and G > 1 and not args.no_single_game:
            # BASELINE configs[1] beside the headline: ONE game, 400 sims/move, same network
            # (latency path: per-board conv kernels, no concurrency to hide behind)
            try:
                one = Engine(B, S, 5, games=1, noise=True, device=local)
                one.seed(0, 0)
                tau1 = np.ones(1, np.int8)
                one.search(net, tau=tau1)
                one.play()
                one.sync()
                t1 = time.perf_counter()
                nmv = 6
                for _ in range(nmv):
                    one.search(net, tau=tau1)
                    one.play()
                one.sync()
                d1 = time.perf_counter() - t1
                out["single_game"] = {"workload": "BASELINE configs[1]: 1 game, %d sims/move, same net" % S,
                                      "value": nmv / d1, "unit": "move-decisions/s", "ms_per_move": d1 / nmv * 1e3}
                one.close()
            except Exception as e:
                out["single_game"] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_tictactoe:
            try:
                out["tictactoe"] = tictactoe_bench(local)
            except Exception as e:
                out["tictactoe"] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(B, S, args.blocks, args.planes, sd, args.cpu_budget)
                if not args.no_cpu_all_cores:
                    out["cpu_baseline"]["all_cores"] = cpu_all_cores(B, S, args.blocks, args.planes, sd,
                                                                     args.cpu_all_cores_budget)
            except Exception as e:  # the baseline is a report, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "move-decisions/s", "cores": None,
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    dog.beat("report printed")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    dog.__exit__(None, None, None)


if __name__ == "__main__":
    main()
