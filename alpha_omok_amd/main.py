"""Drop-in for the reference's `main.self_play` / `main.train` (main.py:122-336).

Same module-level names as the reference (constants, `Agent`, `optimizer`, `rep_memory`,
`cur_memory`, `result`, `step`), so a driver written against `main` keeps working:

    import alpha_omok_amd.main as main
    main.configure(board_size=9, n_mcts=400, n_blocks=4)   # replaces editing the constants
    main.self_play(4096)       # 4096 games CONCURRENTLY on the local MI355X
    main.train(1, n_iter)      # reference loss/Adam; one gradient all-reduce per mini-batch

Differences by design:
  * self_play(n) plays its n games side by side (G = n slots of one engine, finished slots are
    refilled) instead of one after another. Episode e gets its own np.random stream seeded
    SEED + e (the reference draws all games from one stream); with n == 1 the process-global
    np.random state is used, which reproduces `np.random.seed(s); main.self_play(1)` exactly, and
    self_play(n, single_stream=True) is the reference's sequential one-stream schedule for any n.
  * under torch.distributed (one process per GPU) episodes are sharded e % world == rank, train()
    agrees on the mini-batch count and all-reduces the flattened gradient once per mini-batch
    (parallel.py), and run() plays one game PER RANK in the iterations after the first.
  * configure(device_replay=True) keeps rep_memory in HBM (replay.DeviceReplay): the eight
    symmetries are made by a HIP kernel and train() gathers its mini-batches on the device. The
    entries, their order and the batches drawn under a given `random` state are the reference's.
"""
import logging
import os
import random
import time
from collections import deque

import numpy as np

from . import agents, parallel, utils
from .engine import Engine
from .evaluator import Evaluator

# Game
BOARD_SIZE = 9
N_MCTS = 400
TAU_THRES = 6
SEED = 0
PRINT_SELFPLAY = False

# Net
N_BLOCKS = 10
IN_PLANES = 5  # history * 2 + 1
OUT_PLANES = 128

# Training
N_SELFPLAY = 100
TOTAL_ITER = 10000000
MEMORY_SIZE = 30000
N_EPOCHS = 1
BATCH_SIZE = 32
LR = 2e-4
L2 = 0
MAX_CONCURRENT = 4096   # rows of the evaluation batch = leaves evaluated per simulation (and, with OVERSUBSCRIBE = 1, games resident on one GPU)
OVERSUBSCRIBE = 1.0     # configure(oversubscribe=): game slots per row. 1.25 = 5120 resident games on 4096 rows: with a trained network
                        # 11 - 19 % of all leaves are terminal and take no row (the tree kernel hands the rows out per simulation), so
                        # 4096 games leave that share of the trunk's batch empty; the extra games fill it (ao_set_row_cap)
ROWS = 'auto'           # configure(rows=): 'static' = batch rows packed per move by the host; 'dynamic' = handed out per simulation by the
                        # tree kernel (terminal leaves take none); 'auto' = dynamic once a search met >= 3 % terminal leaves, or over-subscribed
GAMES_PER_ITER = None   # run(): games of every iteration after the first (None: the reference's ONE game -- per rank)
TRAIN_STEPS = None      # train(): mini-batches per pass (None: the reference's len(cur_memory), main.py:263-264)

rep_memory = deque(maxlen=MEMORY_SIZE)
cur_memory = utils.SampleQueue()   # the reference's deque (main.py:56) that can also hold a call's samples as ONE block (utils.LazySamples)
step = 0
skipped_steps = 0       # mini-batches whose gradient was not finite and that therefore contributed nothing (see train_batch); updated once per pass
start_iter = 0
total_epoch = 0
result = {'Black': 0, 'White': 0, 'Draw': 0}

Agent = None
optimizer = None
device = None
_engine = None
_evaluator = None
_episodes_played = 0
_trim_seen = [0]
_trim_base = [0, 0]          # counters of engines that were closed since configure()
STRICT = False               # configure(strict=True): an arena trim raises TreeTrimmed instead of logging a warning
NODE_CAP = 0                 # ao_config.node_cap of the self-play engine (0: 16*(sims+1) within 40 % of the HBM; -1: grow into the free HBM)
last_trace = []              # AO_SELFPLAY_TRACE=1: (active games, seconds) of every search of the last _play_episodes call
trim_stats = {'subtrees_dropped': 0, 'reroots_trimmed': 0}   # cumulative since configure(); also returned by self_play
# search shape, cumulative since configure(): PUCT levels walked, exact-tie draws, terminal leaves, evaluated leaves over all
# simulations of all searches (levels / (evaluated + terminal) = mean selection depth)
phase_seconds = {'play': 0.0, 'emit': 0.0, 'train_wait': 0.0}   # cumulative since configure(): self_play's wall time inside the searches / building and
                             # appending the samples / waiting for an overlapped training pass (train_join) before the samples are appended
search_totals = {'levels': 0, 'ties': 0, 'terminal': 0, 'evaluated': 0, 'searches': 0}
CARRY_OVER = None            # configure(carry_over=True): slots freed at the end of one self_play call start the NEXT call's episodes
                             # (None = automatic: on inside run() when GAMES_PER_ITER is set, off otherwise)
DEVICE_STATES = True         # with configure(device_replay=True): the samples' state planes are built on the device from the move lists
                             # (ao_replay_extend_moves) and cur_memory's entries rebuild theirs on first access; False = host-built
CARRY_CALLS = 2              # ... of at most this many calls ahead
OVERLAP_TRAIN = False        # configure(overlap_train=True): run() trains on a worker thread + side stream WHILE the next iteration's games are
                             # played (train_async / train_join); 'serial' = the same deferred schedule without the thread (tests)
played_ahead = [0]           # carry-over + overlapped training: searches made for LATER calls' games while a call that was complete waited for the pass
_train_job = None            # the training pass in flight (train_async): dict(thread, plan, n_epochs, losses, error, done)
_train_stream = None
last_train_losses = None     # losses of the pass train_join() finished last (self_play joins silently)
_pool = None                 # games in flight between self_play calls (carry-over mode only)
_terminal_share = 0.0        # terminal leaves / simulations of the last search (drives ROWS = 'auto')
_carry_auto = False          # run() with GAMES_PER_ITER set and CARRY_OVER None
FP16_GRID = False            # configure(fp16_grid_weights=True): the 3x3 conv weights of Agent.model are kept on the fp16 grid (see _grid_sync)
_grid_masters = []           # ... [(parameter, fp32 master copy)]: Adam moves the master, the module holds its projection


def configure(board_size=None, n_mcts=None, n_blocks=None, in_planes=None, out_planes=None, seed=None,
              model=None, gpu=None, noise=True, device_replay=False, node_cap=None, strict=None, reproducible=False,
              carry_over=None, oversubscribe=None, rows=None, overlap_train=None, fp16_grid_weights=None):
    """Build `Agent`, `Agent.model` and `optimizer` (main.py:58-85). Call instead of editing constants.
    node_cap: expanded-node capacity of a game's tree arena (0 = 16*(n_mcts+1) where 40 % of the HBM
    holds that for all games, at least 4*(n_mcts+1); -1 = grow into the free HBM);
    strict=True makes self_play raise TreeTrimmed when re-rooting had to forget subtrees (otherwise a warning is
    logged and `trim_stats` / self_play's return value carry the counters) and checks after every search that each game's
    visit counts sum to inherited + n_mcts (_check_visits). reproducible=True evaluates every batch
    size with ONE kernel family (ao_net_set_mode 6): an episode's samples then depend on its seed only, not on how many
    other episodes share the engine or on MAX_CONCURRENT (slower for very small and very large batches).
    carry_over=True keeps the engine full across self_play calls: a slot whose game ends when no episode of the current
    call is left to start begins an episode of the NEXT call (same n_selfplay assumed, at most CARRY_CALLS calls ahead) instead
    of idling while the call's longest games run down; every call still returns exactly its own episodes' samples, in
    episode order. The price is the reference's strict alternation (main.py:250-262: all of an iteration's games are played
    by that iteration's network): an episode may have been started -- and partly played -- with the weights of an earlier
    iteration. Off by default for direct self_play calls; run() turns it on when GAMES_PER_ITER is set (thousands of games per
    iteration: the engine would otherwise run down to a handful of games at the end of every call) unless carry_over=False.
    oversubscribe: game slots per row of the evaluation batch (MAX_CONCURRENT rows). 1.25 keeps 5120 games resident on 4096
    rows: the tree kernel hands out the rows per simulation, terminal leaves (11 - 19 % with a trained network) take none,
    and the extra games fill what they leave -- every game still runs the reference's strictly sequential search.
    rows: 'static' / 'dynamic' / 'auto' (see ROWS).
    overlap_train=True: run() does not wait for an iteration's training pass -- it runs on a worker thread and a side stream while
    the NEXT iteration's games are played with the weights exported before it started (train_async; the samples of those games are
    appended, and the new weights published, after train_join). One iteration more of the asynchronous-actor trade carry_over
    already makes; one process per GPU as before (the pass's collectives are issued by the worker thread only).
    fp16_grid_weights=True: the 3x3 conv weights of the network (model.py:6-10: conv1 and the ResBlocks' convs) are kept on the fp16
    grid -- after every optimiser step the module's weights are the fp16 rounding of an fp32 master copy that Adam moves (the
    gradient is taken at the rounded weights: straight-through). Such a network runs on the TWO-product split-fp16 kernels
    (ao_net_products: a third fewer MFMAs, the same fp32-equivalent contraction). The state_dict stays the reference's wire format:
    plain fp32 tensors whose conv entries happen to be fp16 numbers."""
    global BOARD_SIZE, N_MCTS, N_BLOCKS, IN_PLANES, OUT_PLANES, SEED, Agent, optimizer, device
    global _engine, _evaluator, _episodes_played, rep_memory, STRICT, NODE_CAP, CARRY_OVER, _pool, OVERSUBSCRIBE, ROWS, _terminal_share
    global OVERLAP_TRAIN, FP16_GRID
    train_join()
    if fp16_grid_weights is not None:
        FP16_GRID = bool(fp16_grid_weights)
    if overlap_train is not None:
        if overlap_train not in (True, False, 'serial'):
            raise ValueError("overlap_train must be True, False or 'serial'")
        OVERLAP_TRAIN = overlap_train
    if carry_over is not None:
        CARRY_OVER = bool(carry_over)
    if oversubscribe is not None:
        if not 1.0 <= float(oversubscribe) <= 2.0:
            raise ValueError("oversubscribe must be in [1, 2]")
        OVERSUBSCRIBE = float(oversubscribe)
    if rows is not None:
        if rows not in ('auto', 'static', 'dynamic'):
            raise ValueError("rows must be 'auto', 'static' or 'dynamic'")
        ROWS = rows
    _terminal_share = 0.0
    _pool = None
    STRICT = STRICT if strict is None else bool(strict)
    NODE_CAP = NODE_CAP if node_cap is None else int(node_cap)
    _trim_base[0] = _trim_base[1] = 0
    trim_stats['subtrees_dropped'] = trim_stats['reroots_trimmed'] = 0
    for k in search_totals:
        search_totals[k] = 0
    phase_seconds['play'] = phase_seconds['emit'] = phase_seconds['train_wait'] = 0.0
    import torch
    from .pvnet import PVNet
    BOARD_SIZE = board_size or BOARD_SIZE
    N_MCTS = n_mcts or N_MCTS
    N_BLOCKS = N_BLOCKS if n_blocks is None else n_blocks
    IN_PLANES = in_planes or IN_PLANES
    OUT_PLANES = out_planes or OUT_PLANES
    SEED = SEED if seed is None else seed
    rank, world = parallel.world()
    if gpu is None:
        gpu = rank % max(torch.cuda.device_count(), 1)
    device = torch.device('cuda', gpu) if torch.cuda.is_available() else torch.device('cpu')
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(SEED)
    agents.PRINT_MCTS = PRINT_SELFPLAY
    Agent = agents.ZeroAgent(BOARD_SIZE, N_MCTS, IN_PLANES, noise=noise, device=gpu)
    Agent.model = model if model is not None else PVNet(N_BLOCKS, IN_PLANES, OUT_PLANES, BOARD_SIZE).to(device)
    if hasattr(Agent.model, "parameters"):
        parallel.broadcast_parameters(Agent.model)
        optimizer = torch.optim.Adam(Agent.model.parameters(), lr=LR, weight_decay=L2, eps=1e-6)
    _grid_sync()
    _engine = None
    _evaluator = Evaluator(gpu)
    _evaluator.net_mode = 6 if reproducible else 0
    _episodes_played = 0
    if device_replay:
        from .replay import DeviceReplay
        rep_memory = DeviceReplay(BOARD_SIZE, IN_PLANES, MEMORY_SIZE, device=gpu)
    elif not isinstance(rep_memory, deque):
        rep_memory = deque(maxlen=MEMORY_SIZE)
    return Agent


def _grid_sync():
    """configure(fp16_grid_weights=True): (re)build the fp32 master copies of the 3x3 conv weights from what the module holds NOW
    (fresh weights, a loaded checkpoint) and put the module on the fp16 grid. No-op when the switch is off or the model is not a
    torch module."""
    del _grid_masters[:]
    if not FP16_GRID or Agent is None or not hasattr(Agent.model, "named_parameters"):
        return
    import torch
    with torch.no_grad():
        for name, p in Agent.model.named_parameters():
            if p.dim() == 4 and p.shape[2] == 3 and p.shape[3] == 3:
                _grid_masters.append((p, p.detach().clone()))
                p.copy_(p.to(torch.float16).to(p.dtype))
    if _evaluator is not None:
        _evaluator.invalidate()


def _optimizer_step():
    """optimizer.step(); with fp16_grid_weights Adam updates the fp32 master copies (swapped in for the step) and the module gets
    their fp16 rounding back -- everything enqueued, no read-back."""
    if not _grid_masters:
        optimizer.step()
        return
    import torch
    with torch.no_grad():
        for p, m in _grid_masters:
            p.copy_(m)
        optimizer.step()
        for p, m in _grid_masters:
            m.copy_(p)
            p.copy_(p.to(torch.float16).to(p.dtype))


def _get_engine(games):
    global _engine
    if _engine is None or _engine.G != games:
        if _engine is not None:
            d, t = _engine.trim_stats()
            _trim_base[0] += d
            _trim_base[1] += t
            _engine.close()
        _trim_seen[0] = 0
        # more trees than the library's default arena rule plans for (40 % of the HBM for MAX_CONCURRENT-like counts): half of the
        # HBM for an over-subscribed engine's trees -- still the DEFAULT rule (ao_config.arena_fraction), so its clamps and its
        # shrink-on-allocation-failure retry apply (round-5 advisor: an explicit node_cap bypassed both)
        frac = 0.5 if (NODE_CAP == 0 and games > MAX_CONCURRENT) else 0.0
        _engine = Engine(BOARD_SIZE, N_MCTS, IN_PLANES, games=games, noise=Agent.noise, device=Agent._device, node_cap=NODE_CAP,
                         arena_fraction=frac)
    return _engine


def _slots(episodes_available):
    """Game slots of the self-play engine: OVERSUBSCRIBE x MAX_CONCURRENT, never more than there are episodes to play."""
    return int(min(episodes_available, max(1, int(round(MAX_CONCURRENT * OVERSUBSCRIBE)))))


def _set_rows(eng):
    """Before every search: how the evaluation batch gets its rows (ao_set_row_cap; see ROWS). Over-subscribed engines always
    hand them out per simulation -- MAX_CONCURRENT of them."""
    if eng.G > MAX_CONCURRENT or ROWS == 'dynamic' or (ROWS == 'auto' and _terminal_share >= 0.03):
        cap = MAX_CONCURRENT
    else:
        cap = 0
    if getattr(eng, "row_cap", 0) != cap:
        eng.set_row_cap(cap)


def release_engine():
    """Free the self-play engine's HBM (trees of MAX_CONCURRENT games); the next self_play() rebuilds it.
    (Carry-over mode: the games in flight are dropped; their episodes are played again from the start.)"""
    global _engine, _pool
    _pool = None
    if _engine is not None:
        d, t = _engine.trim_stats()                       # the cumulative counters outlive the engine
        _trim_base[0] += d
        _trim_base[1] += t
        _trim_seen[0] = 0
        _engine.close()
        _engine = None


class TreeTrimmed(RuntimeError):
    """strict=True: re-rooting had to forget subtrees because a game's kept tree outgrew node_cap (the reference's
    dict never forgets, agents.py:52), so the searches of those games no longer follow the reference."""


def _count_search(eng):
    """Adds the last search's shape counters (ao_search_stats: per move, summed over its games) to search_totals."""
    global _terminal_share
    st = eng.search_stats()
    for k in ('levels', 'ties', 'terminal', 'evaluated'):
        search_totals[k] += st[k]
    search_totals['searches'] += 1
    _terminal_share = st['terminal'] / max(st['terminal'] + st['evaluated'], 1)


def _check_trim(eng):
    dropped, trimmed = eng.trim_stats()
    trim_stats['subtrees_dropped'] = _trim_base[0] + dropped
    trim_stats['reroots_trimmed'] = _trim_base[1] + trimmed
    if trimmed > _trim_seen[0]:
        msg = ('tree arenas full in {} re-rootings so far ({} child subtrees forgotten; node_cap {}): the searches of those '
               'games diverge from the reference, raise node_cap'.format(trimmed, dropped, eng.node_cap()[0]))
        _trim_seen[0] = trimmed
        if STRICT:
            raise TreeTrimmed(msg)
        logging.warning(msg)


def _check_visits(eng, vis, act, on, inherit, fp16_seen):
    """configure(strict=True): every search ran ALL its simulations (agents.py:105-132) -- visit.sum() is N_MCTS for a fresh root
    and inherited + N_MCTS for a root that was a child of the last one (the child was expanded by the first
    of its n visits, so it brings n - 1). The C side already refuses to end a move that is short (ERR_SHORT); this is the same
    statement made from the outside, on what the caller is handed. Updates `inherit` for the next ply; returns the engine's
    fp16-range event count (a move repeated on the fp32-MFMA trunk starts from fresh trees: nothing inherited)."""
    ev = eng.fp16_range_events()[0]
    if fp16_seen is None:
        fp16_seen = ev
    got = vis[on].sum(axis=1)
    want = (inherit[on] if ev == fp16_seen else 0) + N_MCTS
    if not np.array_equal(got, want):
        bad = np.flatnonzero(got != want)
        g = int(on[bad[0]])
        from .engine import EngineError
        raise EngineError("self-play: the search of game slot %d returned %d visits, %d expected (%d inherited + %d simulations); "
                          "%d of %d searches of this ply are off" % (g, int(got[bad[0]]), int(np.broadcast_to(want, got.shape)[bad[0]]),
                                                                     int(inherit[g]) if ev == fp16_seen else 0, N_MCTS, bad.size, on.size))
    inherit[on] = np.maximum(vis[on, act[on]] - 1, 0)
    return ev


def _play_episodes(episodes, use_global, seed_of):
    """Plays the listed episodes on one engine (G = min(len, MAX_CONCURRENT) slots, finished slots refilled).
    Returns (moves [E, A] int32 (-1 padded), lengths [E], wins [E], ep_of [N], ply_of [N], pis [N, A] float64):
    E = len(episodes), rows in the order of `episodes`; the N = sum(lengths) samples sorted by (row, ply), sample i
    being the search at ply ply_of[i] of row ep_of[i]. Per ply the host only touches whole [G]-arrays; the games
    that finished in that ply are the only per-game work."""
    global _pool
    _pool = None                                          # (the engine is reset below: nothing stays in flight)
    E = len(episodes)
    A = BOARD_SIZE * BOARD_SIZE
    G = _slots(E)
    eng = _get_engine(G)
    eng.reset()
    slot_row = np.full(G, -1, np.int64)                   # row (position in `episodes`) played on each slot
    next_row = 0
    lengths = np.zeros(E, np.int64)
    wins = np.zeros(E, np.int64)
    moves = np.full((E, A), -1, np.int32)
    for g in range(G):
        slot_row[g] = next_row
        if use_global:
            st = np.random.get_state()
            eng.set_rng_state(g, st[1], st[2], st[3], st[4])
        else:
            eng.seed(g, seed_of(episodes[next_row]))
        next_row += 1
    active = np.ones(G, np.uint8)
    ply = np.zeros(G, np.int64)
    hist = []                                             # per search: (rows [n], plies [n], pi [n, A]) of the active slots
    inherit = np.zeros(G, np.int64)                       # strict mode: visits the next root of each slot brings along
    fp16_seen = eng.fp16_range_events()[0]
    trace = [] if os.environ.get("AO_SELFPLAY_TRACE") else None   # (active games, seconds) per search, for tools/time_self_play.py
    while active.any():
        tau = (ply < TAU_THRES).astype(np.int8)           # main.py:150-153
        t_search = time.perf_counter()
        _set_rows(eng)
        pi, vis, _ = _evaluator.search(eng, Agent.model, tau, active=active)
        _count_search(eng)
        act, win = eng.play()                             # utils.get_action + env.step
        if trace is not None:
            trace.append((int(active.sum()), time.perf_counter() - t_search))
        on = np.flatnonzero(active)
        if STRICT:
            _check_trim(eng)                              # (a trimmed re-rooting is reported as what it is, not as a visit mismatch)
            fp16_seen = _check_visits(eng, vis, act, on, inherit, fp16_seen)
        rows = slot_row[on]
        hist.append((rows, ply[on].copy(), pi[on]))
        moves[rows, ply[on]] = act[on]
        ply[on] += 1
        done = on[win[on] != 0]
        if done.size:
            r = slot_row[done]
            lengths[r] = ply[done]
            wins[r] = win[done]
            refill = np.zeros(G, np.uint8)
            for g in done:                                # finished games only
                if next_row < E:
                    refill[g] = 1
                    slot_row[g] = next_row
                    next_row += 1
                else:
                    active[g] = 0
                    slot_row[g] = -1
                ply[g] = 0
                inherit[g] = 0
            if refill.any():
                eng.reset(refill)                         # Agent.reset() (main.py:248)
                gs = np.flatnonzero(refill)
                eng.seed_games(gs, [seed_of(episodes[slot_row[g]]) for g in gs])
    _check_trim(eng)
    if trace is not None:
        last_trace[:] = trace
    if use_global:
        mt, pos, hg, gs = eng.get_rng_state(0)
        np.random.set_state(('MT19937', mt, pos, hg, gs))
    # pi rows per episode in ply order: one stable sort over all searches instead of a Python append per game and ply
    rows = np.concatenate([h[0] for h in hist])
    plies = np.concatenate([h[1] for h in hist])
    pis = np.concatenate([h[2] for h in hist])
    order = np.lexsort((plies, rows))
    return moves, lengths, wins, rows[order], plies[order], pis[order]


class _CarryPool:
    """Carry-over self-play (configure(carry_over=True)): the engine and its games outlive a self_play call.
    Episodes are known by their GLOBAL number (episodes played before the call + index in the call), which also fixes
    their seed, so an episode is the same game whichever call started it (as long as the network did not change under it)."""

    def __init__(self, eng, n_call, rank, world):
        A = eng.A
        self.eng, self.n_call, self.rank, self.world = eng, n_call, rank, world
        self.G = eng.G
        self.slot_id = np.full(self.G, -1, np.int64)
        self.ply = np.zeros(self.G, np.int64)
        self.active = np.zeros(self.G, np.uint8)
        self.slot_moves = np.full((self.G, A), -1, np.int32)
        self.inherit = np.zeros(self.G, np.int64)         # strict mode (_check_visits)
        self.fp16_seen = None                             # ... the engine's fp16-range event count, taken at the first search
        self.finished = {}                                # global id -> (moves [A], length, win)
        self.hist = []                                    # (ids [n], plies [n], pi [n, A]) per search
        self.expect = None                                # first global id of the call expected next
        self.next_call = None                             # first global id of the call the next episode is taken from
        self.next_pos = 0                                 # ... and the position in this rank's shard of that call
        self.shard = parallel.shard_games(n_call, rank, world)

    def take_next(self, limit):
        """Global id of the next episode of this rank to start, or -1 (ids >= limit are not started yet)."""
        if self.next_pos >= len(self.shard):
            self.next_call += self.n_call
            self.next_pos = 0
        gid = self.next_call + self.shard[self.next_pos]
        if gid >= limit:
            return -1
        self.next_pos += 1
        return gid


def _play_carry(first_episode, n_call, rank, world):
    """_play_episodes for carry-over mode: returns this rank's episodes of the call [first_episode, first_episode + n_call)
    -- some of them started, or even finished, during earlier calls -- and leaves later calls' games in flight."""
    global _pool
    A = BOARD_SIZE * BOARD_SIZE
    shard = parallel.shard_games(n_call, rank, world)
    G = _slots(len(shard) * (1 + CARRY_CALLS))            # (later calls' episodes may fill what this call's leave free)
    eng = _get_engine(G)
    pool = _pool
    if pool is None or pool.eng is not eng or (pool.n_call, pool.rank, pool.world, pool.expect) != (n_call, rank, world, first_episode):
        if pool is not None:
            logging.warning('carry-over self-play: n_selfplay / world size / episode numbering changed, %d games in flight dropped',
                            int(pool.active.sum()))
        pool = _pool = _CarryPool(eng, n_call, rank, world)
        pool.next_call = first_episode
        eng.reset()
    pool.expect = first_episode + n_call                 # the call that may pick these games up
    limit = first_episode + n_call * (1 + CARRY_CALLS)
    seed_of = lambda gid: (SEED + gid) & 0xFFFFFFFF
    mine = np.asarray([first_episode + e for e in shard], np.int64)
    todo = set(int(i) for i in mine) - set(pool.finished)

    def start(slots):
        mask = np.zeros(G, np.uint8)
        for g in slots:
            gid = pool.take_next(limit)
            if gid < 0:
                break
            mask[g] = 1
            pool.slot_id[g] = gid
        if mask.any():
            eng.reset(mask)                               # Agent.reset() (main.py:248)
            gs = np.flatnonzero(mask)
            eng.seed_games(gs, [seed_of(int(pool.slot_id[g])) for g in gs])
            pool.active[mask != 0] = 1
            pool.ply[mask != 0] = 0
            pool.slot_moves[mask != 0] = -1
            pool.inherit[mask != 0] = 0

    start(np.flatnonzero(pool.active == 0))
    trace = [] if os.environ.get("AO_SELFPLAY_TRACE") else None

    def pass_running():
        # an overlapped training pass (train_async) that is still at work: self_play would only wait for it (train_join, before
        # the samples are appended) -- the games of the calls to come are played on in the meantime, under the same frozen weights
        job = _train_job
        return job is not None and job['thread'] is not None and job['thread'].is_alive()

    while todo or (pass_running() and pool.active.any()):
        if not todo:
            played_ahead[0] += 1
        if not pool.active.any():
            raise RuntimeError("carry-over self-play: episodes %r are neither in flight nor finished" % sorted(todo)[:8])
        tau = (pool.ply < TAU_THRES).astype(np.int8)      # main.py:150-153
        t_search = time.perf_counter()
        _set_rows(eng)
        pi, vis, _ = _evaluator.search(eng, Agent.model, tau, active=pool.active)
        _count_search(eng)
        act, win = eng.play()
        if trace is not None:
            trace.append((int(pool.active.sum()), time.perf_counter() - t_search))
        on = np.flatnonzero(pool.active)
        if STRICT:
            _check_trim(eng)
            pool.fp16_seen = _check_visits(eng, vis, act, on, pool.inherit, pool.fp16_seen)
        pool.hist.append((pool.slot_id[on].copy(), pool.ply[on].copy(), pi[on]))
        pool.slot_moves[on, pool.ply[on]] = act[on]
        pool.ply[on] += 1
        done = on[win[on] != 0]
        if done.size:
            for g in done:                                # finished games only
                gid = int(pool.slot_id[g])
                pool.finished[gid] = (pool.slot_moves[g].copy(), int(pool.ply[g]), int(win[g]))
                todo.discard(gid)
                pool.slot_id[g] = -1
            pool.active[done] = 0
            start(done)
    _check_trim(eng)
    if trace is not None:
        last_trace[:] = trace
    # this call's episodes out of the pool; the rest stays for the calls to come
    ids = np.concatenate([h[0] for h in pool.hist])
    plies = np.concatenate([h[1] for h in pool.hist])
    pis = np.concatenate([h[2] for h in pool.hist])
    sel = (ids >= first_episode) & (ids < first_episode + n_call)
    pool.hist = [(ids[~sel], plies[~sel], pis[~sel])] if (~sel).any() else []
    ids, plies, pis = ids[sel], plies[sel], pis[sel]
    rows = np.searchsorted(mine, ids)
    order = np.lexsort((plies, rows))
    recs = [pool.finished.pop(int(i)) for i in mine]
    moves = np.stack([r[0] for r in recs])
    lengths = np.asarray([r[1] for r in recs], np.int64)
    wins = np.asarray([r[2] for r in recs], np.int64)
    return moves, lengths, wins, rows[order], plies[order], pis[order]


def self_play(n_selfplay, seeds=None, single_stream=False):
    """Plays n_selfplay episodes and appends their samples to cur_memory / rep_memory exactly as the
    reference does: per episode, plies in chronological order, (state [C,B,B] f64, pi [A] f64, z).

    single_stream=True is the reference's own schedule for n_selfplay > 1 (main.py:136-142): the episodes are
    played ONE AFTER ANOTHER and all draw from the process-global np.random stream, so
    `np.random.seed(s); self_play(n, single_stream=True)` reproduces the reference's memory bit for bit (gv9) --
    at one game's speed. The default plays the episodes side by side with per-episode streams.

    Returns a summary dict (the reference returns None): episodes and move decisions of this rank, and the
    cumulative arena-trim counters (`trim_stats`; with configure(strict=True) a trim raises TreeTrimmed)."""
    global _episodes_played, _pool
    if Agent is None:
        configure()
    if hasattr(Agent.model, "eval") and _train_job is None:
        Agent.model.eval()                                # (main.py:124; not while an overlapped pass is training this module: the
                                                          # searches run on the exported copy, and the join below sets the mode)
    rank, world = parallel.world()
    if single_stream and world > 1:
        raise ValueError("single_stream self-play is the reference's sequential schedule: one process only")
    episodes = parallel.shard_games(n_selfplay, rank, world)
    first_episode = _episodes_played
    _episodes_played += n_selfplay                        # identical on every rank, shard or no shard
    if not episodes:
        Agent.reset()
        return dict(episodes=0, moves=0, **trim_stats)

    def seed_of(ep):
        return int(seeds[ep]) if seeds is not None else (SEED + first_episode + ep) & 0xFFFFFFFF

    A = BOARD_SIZE * BOARD_SIZE
    t_play = time.perf_counter()
    try:
        if single_stream or (n_selfplay == 1 and seeds is None and world == 1):
            parts = [_play_episodes([ep], True, seed_of) for ep in episodes]   # sequential, each on the global stream where the last left it
            moves = np.concatenate([p[0] for p in parts])
            lengths = np.concatenate([p[1] for p in parts])
            wins = np.concatenate([p[2] for p in parts])
            ep_of = np.concatenate([np.full(p[3].shape[0], i, np.int64) for i, p in enumerate(parts)])
            ply_of = np.concatenate([p[4] for p in parts])
            pis = np.concatenate([p[5] for p in parts])
        elif CARRY_OVER or (CARRY_OVER is None and _carry_auto):
            if seeds is not None:
                raise ValueError("carry-over self-play starts episodes of later calls: their seeds are SEED + episode number, "
                                 "an explicit seeds= list cannot be honoured")
            moves, lengths, wins, ep_of, ply_of, pis = _play_carry(first_episode, n_selfplay, rank, world)
        else:
            moves, lengths, wins, ep_of, ply_of, pis = _play_episodes(episodes, False, seed_of)
    except BaseException:
        # a call that did not deliver (TreeTrimmed under strict=True, an engine error, Ctrl-C) leaves nothing behind:
        # the episode numbering is where it was -- a retry plays the same episodes with the same seeds -- and the
        # carry-over pool is dropped (its finished-but-undelivered episodes and pi histories would otherwise stay in
        # it forever and the later calls' games would keep occupying slots)
        # -- with ONE process. Under torch.distributed the other ranks have already advanced past these episode numbers (the
        # counter must stay identical on every rank), so a rank that catches the exception and carries on keeps the advanced
        # counter: the failed call's episodes are skipped, not replayed under seeds the others have left behind. (And in
        # single_stream mode the global np.random stream has moved on: a retry is a new draw, not a replay.)
        if world == 1:
            _episodes_played = first_episode
        _pool = None
        raise

    t_emit = time.perf_counter()
    phase_seconds['play'] += t_emit - t_play
    if _train_job is not None:
        # an overlapped training pass (train_async) draws its mini-batches from rep_memory by position: it has to be over before this
        # call's samples move the ring -- and its weights are published here, for the next call's searches
        train_join()
        if hasattr(Agent.model, "eval"):
            Agent.model.eval()
        t_join = time.perf_counter()
        phase_seconds['train_wait'] += t_join - t_emit
        t_emit = t_join
    # results and samples in episode order (main.py:201-227); samples arrive sorted by (episode, ply)
    result['Black'] += int((wins == 1).sum())
    result['White'] += int((wins == 2).sum())
    result['Draw'] += int(((wins != 1) & (wins != 2)).sum())
    reward_black = np.where(wins == 1, 1., np.where(wins == 2, -1., 0.))
    z = np.where(ply_of % 2 == 0, reward_black[ep_of], -reward_black[ep_of])
    z = np.where(z == 0, 0., z)                           # (no -0.0: the reference's draw reward is +0.0 for both colours)
    if DEVICE_STATES and hasattr(rep_memory, "extend_augmented_moves"):
        # device-side sample emission: the replay memory builds the planes of utils.get_state_pt from the move lists in a kernel
        # (ao_replay_extend_moves); cur_memory gets entries that rebuild their state only if somebody looks at it
        n_new = int(ep_of.shape[0])
        cur_memory.extend(utils.LazySamples(moves, ep_of, ply_of, pis, z, BOARD_SIZE, IN_PLANES))   # one block, no per-sample object
        Agent.reset()
        rep_memory.extend_augmented_moves(moves, ep_of, ply_of, pis, z)
        phase_seconds['emit'] += time.perf_counter() - t_emit
        return dict(episodes=len(episodes), moves=n_new, **trim_stats)
    states = utils.states_of_episodes(moves, ep_of, ply_of, BOARD_SIZE, IN_PLANES)
    n_new = states.shape[0]
    zl = z.tolist()
    cur_memory.extend(zip(states, pis, zl))               # rows of the big arrays: (state f64 [C,B,B], pi f64 [A], z float)
    Agent.reset()
    if hasattr(rep_memory, "extend_augmented_arrays"):
        rep_memory.extend_augmented_arrays(states, pis, z)   # symmetries made on the device; only what can survive is uploaded
    else:
        # deque(maxlen) keeps the newest entries only: once this call alone fills it, just the samples whose
        # symmetries can survive are augmented (identical final content, without 8 x n_new Python tuples)
        keep = n_new
        if rep_memory.maxlen is not None and 8 * n_new >= rep_memory.maxlen:
            keep = min(n_new, -(-rep_memory.maxlen // 8))
        tail = [(states[i], pis[i], zl[i]) for i in range(n_new - keep, n_new)]
        rep_memory.extend(utils.augment_dataset(tail, BOARD_SIZE))
    phase_seconds['emit'] += time.perf_counter() - t_emit
    return dict(episodes=len(episodes), moves=int(n_new), **trim_stats)


def train_batch(batch, total=None):
    """One optimiser step of main.py:283-305 on `batch` (entries of rep_memory, or positions in it when
    rep_memory lives on the device; an EMPTY batch means this rank has nothing for this step and only
    takes part in the collective). Local forward/backward, one all-reduce of the flattened gradient over
    the ranks (none with one process), Adam step. `total`: the number of samples ALL ranks put behind this step,
    when the caller knows it (train() does, from one agreement per pass) -- the all-reduce then needs no read-back.
    Returns (loss, v_loss, p_loss) -- device scalars, float() reads them -- or None for an empty batch; `step` advances
    whenever any rank contributed."""
    global step
    import torch
    optimizer.zero_grad()
    out = None
    if len(batch) > 0:
        if hasattr(rep_memory, "batch"):
            s_batch, pi_batch, z_batch = rep_memory.batch(batch)
        else:
            s_batch = torch.tensor(np.stack([b[0] for b in batch])).to(device).float()
            pi_batch = torch.tensor(np.stack([b[1] for b in batch])).to(device).float()
            z_batch = torch.tensor(np.array([b[2] for b in batch])).to(device).float()
        p_batch, v_batch = Agent.model(s_batch)
        v_loss = (v_batch - z_batch).pow(2).mean()
        p_loss = -(pi_batch * p_batch.log()).sum(dim=-1).mean()
        loss = v_loss + p_loss
        loss.backward()
    # each rank's gradient counts for the samples behind it (a shard that ran short contributes a partial batch)
    _, contributors = parallel.allreduce_gradients(Agent.model, contributes=len(batch) > 0, weight=len(batch), total=total)
    if contributors == 0:
        return None
    # -(pi * p.log()) is the reference's loss (main.py:296-299) and it is -inf * pi as soon as the softmax underflows to an exact 0 on
    # a move the search visited; the reference would write NaN into every weight and carry on. Here such a mini-batch contributes
    # NOTHING: when the (all-reduced) gradient is not finite, all of it is replaced by zeros ON THE DEVICE before the optimiser step
    # (Adam then coasts on its moments for one step; no weight can become NaN) -- no host synchronisation per step, and the same on
    # every rank: the gradients are identical after the all-reduce, so the ranks agree without talking. The event is counted on
    # the device and read back once per pass (`skipped_steps`, the pass's log line; _train_execute).
    _zero_nonfinite_gradients()
    _optimizer_step()
    step += 1
    if len(batch) > 0:
        out = _LossRecord((loss.detach(), v_loss.detach(), p_loss.detach()))
    return out


class _LossRecord(tuple):
    """(loss, v_loss, p_loss) of a mini-batch, still on the device: float(x) / np.array(...) read them back, and _train_execute does
    that once per pass instead of once per step."""


_skipped_dev = None          # device counter of the mini-batches whose gradient was not finite (see train_batch)


def _zero_nonfinite_gradients():
    global _skipped_dev
    import torch
    grads = [p.grad for g in optimizer.param_groups for p in g['params'] if p.grad is not None]
    if not grads:
        return
    finite = torch.isfinite(torch.stack(torch._foreach_norm(grads))).all()
    if _skipped_dev is None or _skipped_dev.device != finite.device:
        _skipped_dev = torch.zeros((), dtype=torch.int64, device=finite.device)
    _skipped_dev += (~finite).to(torch.int64)
    for g in grads:
        g.nan_to_num_(nan=0.0, posinf=0.0, neginf=0.0)    # (so that the multiplication below yields zeros, not NaN)
    torch._foreach_mul_(grads, finite.to(grads[0].dtype))


def _collect_skipped():
    """Adds the device counter of skipped mini-batches to `skipped_steps` (one read-back) and returns how many were new."""
    global skipped_steps
    if _skipped_dev is None:
        return 0
    n = int(_skipped_dev.item())
    _skipped_dev.zero_()
    if n:
        skipped_steps += n
        logging.warning('train: %d mini-batch(es) of this pass had a non-finite gradient and contributed nothing (%d so far)', n, skipped_steps)
    return n


def train(n_epochs, n_iter):
    """One pass over 32*len(cur_memory) samples of rep_memory, batch 32, loss = MSE(v, z) +
    CE(pi, p), Adam (main.py:253-336). With one process this is the reference's pass, error
    behaviour included: random.sample raises ValueError when rep_memory holds fewer than
    32*len(cur_memory) entries (main.py:263-264).

    TRAIN_STEPS (module constant, None by default) replaces the reference's len(cur_memory) mini-batches per pass:
    with thousands of concurrent games per iteration one mini-batch per NEW sample is hours of optimiser steps;
    BATCH_SIZE x TRAIN_STEPS then sets the replay draws per pass (tools/train_omok.py).

    Under torch.distributed (one process per GPU, rank-local cur_memory / rep_memory) it is the
    data-parallel form of the same pass. The ranks first agree on the number of mini-batches --
    ceil(sum over ranks of len(cur_memory) / world), so every new sample still buys 32 replay draws in
    total -- then every rank draws that many batches of 32 from its OWN replay shard and each step
    is: local forward/backward, ONE all-reduce of the flattened gradient, identical Adam step
    everywhere. A rank whose shard is too small for a step (or empty) adds zeros and is left out of
    the divisor, so every rank issues exactly the same collectives and the weights stay bit-identical
    across ranks; the BatchNorm running statistics are averaged once at the end of the pass."""
    train_join()
    return _train_execute(_train_plan(), n_epochs)


def _pass_watchdog(what):
    """Under torch.distributed every mini-batch of a pass is a collective: a rank that stops making progress there ends itself
    (parallel.Watchdog, AO_WATCHDOG_S seconds without a finished mini-batch, default 120) and the process group's timeout
    (parallel.init_from_env) takes the waiting peers down -- no rank is left blocked in an all-reduce. One process: nothing armed."""
    stall = float(os.environ.get("AO_WATCHDOG_S", "120")) if parallel.world()[1] > 1 else 0.0
    return parallel.Watchdog(stall, what)


def _train_plan():
    """What a training pass needs from the CALLING thread: the number of mini-batches, the positions random.sample picks (the
    `random` stream is consumed here) and, under torch.distributed, the per-step sample totals all ranks agree on."""
    rank, world = parallel.world()
    on_device = hasattr(rep_memory, "batch")
    with _pass_watchdog("planning a training pass (agreement on the mini-batch count)"):
        return _train_plan_locked(rank, world, on_device)


def _train_plan_locked(rank, world, on_device):
    if world == 1:
        n_steps = len(cur_memory) if TRAIN_STEPS is None else int(TRAIN_STEPS)
        n = BATCH_SIZE * n_steps
    else:
        n_steps = -(-parallel.agree(len(cur_memory), "sum", device) // world) if TRAIN_STEPS is None else int(TRAIN_STEPS)
        n = min(BATCH_SIZE * n_steps, len(rep_memory))
    # random.sample picks POSITIONS: sampling range(len) draws the same entries, with the same
    # consumption of the `random` stream, as sampling the sequence itself
    train_memory = random.sample(range(len(rep_memory)), n) if on_device else random.sample(list(rep_memory), n)
    logging.warning('current memory size: {}'.format(len(cur_memory)))
    logging.warning('replay memory size: {}'.format(len(rep_memory)))
    logging.warning('train memory size: {}'.format(len(train_memory)))
    # the samples all ranks together put behind each mini-batch of the pass: one small all-reduce here instead of a
    # device read-back inside every gradient all-reduce
    totals = [None] * n_steps
    if world > 1:
        totals = parallel.agree_sums([len(train_memory[i * BATCH_SIZE:(i + 1) * BATCH_SIZE]) for i in range(n_steps)], device)
    return n_steps, train_memory, totals


def _train_execute(plan, n_epochs, publish=True):
    """The mini-batches of a planned pass (main.py:266-336). publish=False (train_async's worker): the searches' native copy of the
    weights is NOT invalidated here -- train_join does that on the thread that runs the searches."""
    global total_epoch
    with _pass_watchdog("training pass") as dog:
        return _train_execute_watched(plan, n_epochs, publish, dog)


def _train_execute_watched(plan, n_epochs, publish, dog):
    global total_epoch
    n_steps, train_memory, totals = plan
    Agent.model.train()
    losses = []
    trained = False
    for epoch in range(n_epochs):
        for i in range(n_steps):
            dog.beat("training pass, mini-batch %d of %d" % (i + 1, n_steps))
            batch = train_memory[i * BATCH_SIZE:(i + 1) * BATCH_SIZE]
            out = train_batch(batch, totals[i])
            if out is not None:
                trained = True
                losses.append(out)
                if PRINT_SELFPLAY:
                    print('{:4} Step Loss: {:.4f}   Loss V: {:.4f}   Loss P: {:.4f}'.format(step, *[float(x) for x in out]))
        total_epoch += 1
        if losses:
            import torch
            done = len([x for x in losses if not isinstance(x, _LossRecord)])
            if done < len(losses):                        # this epoch's records: ONE read-back for all of them
                host = torch.stack([torch.stack(list(x)) for x in losses[done:]]).cpu().tolist()
                losses[done:] = [tuple(h) for h in host]
            with np.errstate(all='ignore'):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore', RuntimeWarning)
                    m = np.nanmean(np.array(losses), axis=0)   # (a mini-batch whose loss was not finite is not part of the mean)
            logging.warning('{:2} Epoch Loss: {:.4f}   Loss_V: {:.4f}   Loss_P: {:.4f}'.format(total_epoch, *m))
    _collect_skipped()
    parallel.average_buffers(Agent.model, contributes=trained)
    if publish and _evaluator is not None:
        _evaluator.invalidate()                           # the native copy of the weights is stale now
    return losses


def train_async(n_epochs, n_iter):
    """train() that does not make the GPU's games wait (round-4 review, item 4b; the loop it serves: main.py:377-414). The pass is
    planned here, on the calling thread (sampling, agreement across ranks); its mini-batches then run on a worker thread and a
    side stream while the caller goes on to the next self_play call. Until train_join() the searches keep the weights exported
    BEFORE the pass started (Evaluator.freeze: the native forward reads its own repacked copy in HBM, so the module can be
    updated in place next to it); self_play calls train_join() itself before it appends its samples -- the pass draws from
    rep_memory by position -- and so do train / save_model / save_dataset / load_data / configure. A model without a native
    form (its module would be called by the searches while it is being trained) gets the plain synchronous pass.
    OVERLAP_TRAIN = 'serial' keeps the schedule and drops the thread: the pass runs inside train_join (tests)."""
    global _train_job, _train_stream
    train_join()
    if _evaluator is None or not device.type == 'cuda' or not _evaluator.freeze(Agent.model, BOARD_SIZE, IN_PLANES):
        losses = _train_execute(_train_plan(), n_epochs)
        _train_job = dict(thread=None, losses=losses, error=None, done=True)
        return
    try:
        plan = _train_plan()
    except BaseException:                                 # (e.g. the reference's ValueError for a replay memory that is too small)
        _evaluator.thaw()
        raise
    job = dict(thread=None, plan=plan, n_epochs=n_epochs, losses=None, error=None, done=False)
    if OVERLAP_TRAIN == 'serial':
        _train_job = job
        return
    import threading
    import torch
    torch.cuda.synchronize(device)                        # everything the pass reads (the replay ring's last append) is in memory
    if _train_stream is None or _train_stream.device != device:
        _train_stream = torch.cuda.Stream(device)         # (torch's pool streams do not synchronise with the null stream)

    def work():
        try:
            torch.cuda.set_device(device)
            with torch.cuda.stream(_train_stream):
                job['losses'] = _train_execute(plan, n_epochs, publish=False)
            _train_stream.synchronize()
        except BaseException as e:                        # handed to the thread that joins
            job['error'] = e
        job['done'] = True

    job['thread'] = threading.Thread(target=work, name="alpha_omok_amd.train", daemon=True)
    _train_job = job
    job['thread'].start()


def train_join():
    """Wait for the pass train_async started, publish its weights to the searches, return its losses (None: nothing was in
    flight). Raises what the pass raised."""
    global _train_job, last_train_losses
    job = _train_job
    if job is None:
        return None
    _train_job = None
    try:
        if job['thread'] is not None:
            job['thread'].join()
        elif not job['done']:
            job['losses'] = _train_execute(job['plan'], job['n_epochs'], publish=False)
    finally:
        if _evaluator is not None:
            _evaluator.thaw()
    if job['error'] is not None:
        raise job['error']
    last_train_losses = job['losses']
    return job['losses']


def reset_iter(result_, cur_memory_):
    """main.py:368-374"""
    global total_epoch
    result_['Black'] = 0
    result_['White'] = 0
    result_['Draw'] = 0
    total_epoch = 0
    cur_memory_.clear()


# ---- checkpoint wire format (main.py:339-365): torch.save(state_dict) / pickled rep_memory, with the
# iteration and step encoded in the file name and parsed back on load ----
def save_model(agent, n_iter, step_, directory='data', datetime_now=None):
    train_join()
    import os
    from datetime import datetime
    import torch
    datetime_now = datetime_now or datetime.now().strftime('%y%m%d')
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, '{}_{}_{}_step_model.pickle'.format(datetime_now, n_iter, step_))
    torch.save(agent.model.state_dict(), path)
    return path


def save_dataset(memory, n_iter, step_, directory='data', datetime_now=None):
    train_join()
    import os
    import pickle
    from datetime import datetime
    datetime_now = datetime_now or datetime.now().strftime('%y%m%d')
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, '{}_{}_{}_step_dataset.pickle'.format(datetime_now, n_iter, step_))
    # replay memories are rank-local under torch.distributed: rank 0 writes the reference's file name, every other rank
    # its own shard beside it (<name>.rank<r>of<w>), so a resume loses nobody's samples (load_data)
    rank, world = parallel.world()
    if world > 1 and rank > 0:
        path += '.rank{}of{}'.format(rank, world)
    if not isinstance(memory, deque):                     # DeviceReplay: pickle what the reference pickles
        memory = deque(list(memory), maxlen=memory.maxlen)
    with open(path, 'wb') as f:
        pickle.dump(memory, f, pickle.HIGHEST_PROTOCOL)
    return path


def load_data(model_path, dataset_path):
    """Tolerant load like the reference: state.update(torch.load(path)); iteration / step come from
    the file NAME ({yymmdd}_{iter}_{step}_step_model.pickle)."""
    global rep_memory, step, start_iter
    import os
    import pickle
    import torch
    train_join()
    if model_path:
        state = Agent.model.state_dict()
        state.update(torch.load(model_path, map_location=device))
        Agent.model.load_state_dict(state)
        _grid_sync()
        if _evaluator is not None:
            _evaluator.invalidate()
        name = os.path.basename(model_path)
        step = int(name.split('_')[2])
        start_iter = int(name.split('_')[1]) + 1
    if dataset_path:
        import glob
        rank, world = parallel.world()
        with open(dataset_path, 'rb') as f:
            loaded = pickle.load(f)
        # shard files beside it, by their parsed suffix: {written world size: {rank: path}}
        import re
        found = {}
        for sp in glob.glob(glob.escape(dataset_path) + '.rank*of*'):
            m = re.fullmatch(r'\.rank(\d+)of(\d+)', sp[len(dataset_path):])
            if m:
                found.setdefault(int(m.group(2)), {})[int(m.group(1))] = sp
        # the same decision on every rank, from the file names alone: the shards of a job of THIS shape are there
        # (ranks 1 .. world-1 of `world`, nothing left over from a job of another size)
        same_shape = world > 1 and list(found) == [world] and sorted(found[world]) == list(range(1, world))
        if same_shape:
            if rank > 0:                                  # every rank takes its own shard back
                with open(found[world][rank], 'rb') as f:
                    loaded = pickle.load(f)
        elif world > 1 or found:
            # another shape (or a single-process file next to stale shards): pool rank 0's file with the newest shard set and
            # deal it out. The shards are interleaved round robin (entry i of every shard before entry i + 1 of any), so
            # a memory that has to be cut to MEMORY_SIZE loses the OLDEST entries of every shard alike, not one rank's.
            parts = [list(loaded)]
            if found:
                w = max(found, key=lambda k: max(os.path.getmtime(q) for q in found[k].values()))   # the newest set
                for r in sorted(found[w]):
                    with open(found[w][r], 'rb') as f:
                        parts.append(list(pickle.load(f)))
            longest = max(len(q) for q in parts)
            # right-aligned interleave: the newest entries of all shards end up at the end of the pool
            pool = [q[i - (longest - len(q))] for i in range(longest) for q in parts if i >= longest - len(q)]
            pool = pool[max(0, len(pool) - MEMORY_SIZE * world):]
            loaded = pool[rank::world]
        if hasattr(rep_memory, "extend_augmented"):
            rep_memory.clear()
            rep_memory.extend(loaded)
        else:
            rep_memory = deque(loaded, maxlen=MEMORY_SIZE)


def run(total_iter=None, model_path=None, dataset_path=None, n_selfplay=None, save_every=100, directory='data'):
    """The reference's top-level loop (main.py:377-414): iteration 0 fills the replay memory with
    N_SELFPLAY games; every later iteration plays ONE game (per rank: `world` games under
    torch.distributed, so no GPU idles) and trains N_EPOCHS on it; model and
    dataset are saved when n_iter % save_every == 0 (named n_iter + save_every, as the reference does);
    result / cur_memory are reset after every iteration. Returns the number of iterations run."""
    from datetime import datetime
    if Agent is None:
        configure()
    load_data(model_path, dataset_path)
    total_iter = TOTAL_ITER if total_iter is None else total_iter
    n_first = N_SELFPLAY if n_selfplay is None else n_selfplay
    done = 0
    for n_iter in range(start_iter, total_iter):
        logging.warning(datetime.now().isoformat())
        logging.warning('=' * 58)
        logging.warning(' ' * 20 + '  {:2} Iteration  '.format(n_iter) + ' ' * 20)
        logging.warning('=' * 58)
        if n_iter > 0:
            # the reference's one game -- on every GPU; GAMES_PER_ITER games (over all ranks) when set. With thousands of games
            # per iteration the engine is kept full ACROSS iterations (carry-over, see configure) unless carry_over=False was
            # asked for: the price is the reference's strict alternation -- an episode may have been started under the
            # weights of up to CARRY_CALLS iterations ago.
            global _carry_auto
            _carry_auto = GAMES_PER_ITER is not None
            try:
                self_play(parallel.world()[1] if GAMES_PER_ITER is None else int(GAMES_PER_ITER))
            finally:
                _carry_auto = False
            if OVERLAP_TRAIN:
                train_async(N_EPOCHS, n_iter)             # joined by the next self_play before it appends its samples (or below)
            else:
                train(N_EPOCHS, n_iter)
        else:
            self_play(n_first)
        if n_iter % save_every == 0:
            train_join()                                  # a checkpoint holds the weights AFTER this iteration's pass, as in the reference
            today = parallel.agree(int(datetime.now().strftime('%y%m%d')), "max", device)   # one file-name date for all ranks
            if parallel.world()[0] == 0:
                save_model(Agent, n_iter + save_every, step, directory, '{:06d}'.format(today))
            save_dataset(rep_memory, n_iter + save_every, step, directory, '{:06d}'.format(today))   # every rank: its own shard
        reset_iter(result, cur_memory)
        done += 1
    train_join()
    return done


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description="self-play + training loop of the reference's main.py on the MI355X engine")
    ap.add_argument('--board', type=int, default=BOARD_SIZE)
    ap.add_argument('--sims', type=int, default=N_MCTS)
    ap.add_argument('--blocks', type=int, default=N_BLOCKS)
    ap.add_argument('--iters', type=int, default=TOTAL_ITER)
    ap.add_argument('--selfplay', type=int, default=N_SELFPLAY, help='games of iteration 0')
    ap.add_argument('--model', default=None)
    ap.add_argument('--dataset', default=None)
    ap.add_argument('--device-replay', action='store_true')
    ap.add_argument('--games-per-iter', type=int, default=None, help='games of every iteration after the first (default: the reference\'s one game per rank)')
    ap.add_argument('--train-steps', type=int, default=None, help='mini-batches per training pass (default: the reference\'s one per new sample)')
    ap.add_argument('--oversubscribe', type=float, default=None, help='game slots per row of the evaluation batch (1.25 with a trained network)')
    ap.add_argument('--overlap-train', action='store_true', help='train beside the next iteration\'s games (train_async)')
    ap.add_argument('--fp16-grid-weights', action='store_true', help='keep the 3x3 conv weights on the fp16 grid: two-product split-fp16 kernels (configure)')
    a = ap.parse_args()
    parallel.init_from_env()
    GAMES_PER_ITER, TRAIN_STEPS = a.games_per_iter, a.train_steps
    configure(board_size=a.board, n_mcts=a.sims, n_blocks=a.blocks, device_replay=a.device_replay, oversubscribe=a.oversubscribe,
              overlap_train=a.overlap_train, fp16_grid_weights=a.fp16_grid_weights)
    run(a.iters, a.model, a.dataset, a.selfplay)
