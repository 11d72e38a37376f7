"""Python handles over the C ABI: `Engine` (G concurrent games) and `Net` (PVNet weights).

Thin by design: argument marshalling and error propagation only. Tensors are exchanged as raw
device pointers (torch.Tensor.data_ptr()); no torch types cross the boundary.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AO_ROOT_EXPANDED, AO_ROOT_FRESH, AO_ROOT_UNEXPANDED  # noqa: F401


class EngineError(RuntimeError):
    pass


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


def plan_kernel(n_block, inplanes, planes, board_size, boards, in_kind=2, trunk_mode=0):
    """(kernel name as rocprofv3 prints it + description, algorithmic FLOPs per launch) of the kernel that would
    carry the conv stack for `boards` positions -- planning logic only, needs no GPU (ao_net_plan_kernel).
    in_kind 2: the engine's bit planes (what ao_search feeds), 1: the fp32 plane batch (ao_net_forward)."""
    L = _lib.load()
    buf = C.create_string_buffer(256)
    f = C.c_double(0)
    if L.ao_net_plan_kernel(int(n_block), int(inplanes), int(planes), int(board_size), int(trunk_mode), int(boards),
                            int(in_kind), buf, 256, C.byref(f)):
        raise EngineError("ao_net_plan_kernel: bad network shape")
    return buf.value.decode(), f.value


class Net:
    """model.PVNet(n_block, inplanes, planes, board_size) in eval() mode, on the MI355X."""

    def __init__(self, n_block, inplanes, planes, board_size, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        if self._L.ao_net_create(n_block, inplanes, planes, board_size, device, C.byref(h)):
            raise EngineError("ao_net_create: " + self._L.ao_net_last_error(None).decode())
        self._h = h
        self.n_block, self.inplanes, self.planes, self.board_size = n_block, inplanes, planes, board_size
        self.device = device
        self.A = board_size * board_size

    def _check(self, rc, what):
        if rc:
            raise EngineError("%s: %s" % (what, self._L.ao_net_last_error(self._h).decode()))

    def load_state_dict(self, state_dict):
        """state_dict: name -> array-like (numpy or torch tensor), reference key names. Device tensors come over in
        ONE flattened copy (a per-tensor .cpu() costs ~60 synchronous transfers per export)."""
        items = list(state_dict.items())
        dev = [(k, v) for k, v in items if hasattr(v, "is_cuda") and v.is_cuda]
        host = {}
        if dev:
            import torch
            flat = torch.cat([v.detach().reshape(-1).to(torch.float32) for _, v in dev]).cpu().numpy()
            off = 0
            for k, v in dev:
                n = v.numel()
                host[k] = flat[off:off + n]
                off += n
        for name, val in items:
            if name in host:
                arr = host[name]
            else:
                if hasattr(val, "detach"):
                    val = val.detach().cpu().numpy()
                arr = np.ascontiguousarray(np.asarray(val), dtype=np.float32).reshape(-1)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            self._check(self._L.ao_net_set_param(self._h, name.encode(), _ptr(arr, C.c_float), arr.size),
                        "ao_net_set_param(%s)" % name)
        self._check(self._L.ao_net_finalize(self._h), "ao_net_finalize")
        return self

    def forward_ptr(self, planes_ptr, batch, policy_ptr, value_ptr, stream=None):
        self._check(self._L.ao_net_forward(self._h, planes_ptr, batch, policy_ptr, value_ptr, stream),
                    "ao_net_forward")

    def __call__(self, x):
        """x: torch.cuda float32 [batch, C, B, B] -> (policy [batch, A], value [batch]) tensors."""
        import torch
        x = x.contiguous().float()
        batch = x.shape[0]
        pol = torch.empty((batch, self.A), dtype=torch.float32, device=x.device)
        val = torch.empty((batch,), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self.forward_ptr(x.data_ptr(), batch, pol.data_ptr(), val.data_ptr(), stream)
        return pol, val

    def set_mode(self, mode):
        """0 auto, 1 layer kernels (32-board groups), 2 group-resident trunk, 3 per-board, 4 row-chunked layers,
        5 group-resident trunk on split-fp16 MFMAs (fp32-accurate; 128 planes, board <= 9x9), 6 the per-layer split-fp16
        kernels for every batch size (a position's evaluation then does not depend on what else is in the batch)."""
        self._check(self._L.ao_net_set_mode(self._h, int(mode)), "ao_net_set_mode")

    def get_mode(self):
        """The mode in force: what set_mode asked for, or 2 while the fp16-range fallback holds the network on the
        fp32-MFMA trunk (three repeated moves with the same weights; ends with the next load_state_dict / set_mode)."""
        return int(self._L.ao_net_get_mode(self._h))

    def status(self, clear=True, stream=None):
        """Status word (ao_net_status): bit 0 = an activation left the fp16 range in the split-fp16 trunk since
        the last clear (those forwards were clamped, not fp32-equivalent). Synchronises `stream`."""
        f = C.c_int32(0)
        self._check(self._L.ao_net_status(self._h, stream, C.byref(f), 1 if clear else 0), "ao_net_status")
        return f.value

    def products(self, request=-1):
        """(MFMA products per multiply-add in force: 2 or 3, True when the loaded conv weights are fp16 numbers) -- ao_net_products.
        request 0: two products whenever the weights allow (default), 3: always three, -1: query only."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self._L.ao_net_products(self._h, int(request), C.byref(a), C.byref(b)), "ao_net_products")
        return a.value, bool(b.value)

    def dominant_kernel(self, boards):
        """(name, algorithmic FLOPs per launch) of the kernel conv_timing() measures."""
        buf = C.create_string_buffer(256)
        f = C.c_double(0)
        self._check(self._L.ao_net_dominant_kernel(self._h, int(boards), buf, 256, C.byref(f)),
                    "ao_net_dominant_kernel")
        return buf.value.decode(), f.value

    def conv_timing(self, enable=True):
        """Returns (total ms, launches) of the TIMED trunk conv launches since the last call; enable = n > 1 times every
        n-th forward only (an event pair per launch costs ~1 % of a 1.5 ms step)."""
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._check(self._L.ao_net_conv_timing(self._h, int(enable), C.byref(ms), C.byref(cnt)),
                    "ao_net_conv_timing")
        return ms.value, cnt.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.ao_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """G concurrent games: SoA search trees in HBM + the tree kernels (see include/omok_hip.h)."""

    def __init__(self, board_size, num_mcts, inplanes=5, games=1, noise=True, device=0, node_cap=0,
                 c_puct=0.0, alpha=0.0, win_mark=0, arena_fraction=0.0):
        self._L = _lib.load()
        cfg = _lib.AoConfig(board=board_size, win_mark=win_mark, sims=num_mcts, inplanes=inplanes,
                            games=games, noise=1 if noise else 0, node_cap=node_cap, device=device,
                            c_puct=c_puct, alpha=alpha, arena_fraction=arena_fraction)
        h = C.c_void_p()
        if self._L.ao_create(C.byref(cfg), C.byref(h)):
            raise EngineError("ao_create: " + self._L.ao_last_error(None).decode())
        self._h = h
        self.board_size, self.num_mcts, self.inplanes, self.G = board_size, num_mcts, inplanes, games
        self.A = board_size * board_size
        self.device = device
        self._fp16_seen = self._fp16_games_seen = 0

    def _check(self, rc, what):
        if rc:
            raise EngineError("%s: %s" % (what, self._L.ao_last_error(self._h).decode()))

    # -- rng
    def seed(self, game, seed):
        self._check(self._L.ao_seed(self._h, game, seed & 0xFFFFFFFF), "ao_seed")

    def seed_all(self, seeds):
        s = np.ascontiguousarray(seeds, np.uint32)
        assert s.size == self.G
        self._check(self._L.ao_seed_all(self._h, _ptr(s, C.c_uint32)), "ao_seed_all")

    def seed_games(self, games, seeds):
        """seed() for the listed games with one synchronisation (ao_seed_games)."""
        g = np.ascontiguousarray(games, np.int32)
        s = np.ascontiguousarray(np.asarray(seeds, np.uint64) & 0xFFFFFFFF, np.uint32)
        assert g.size == s.size
        if g.size:
            self._check(self._L.ao_seed_games(self._h, _ptr(g, C.c_int32), _ptr(s, C.c_uint32), int(g.size)), "ao_seed_games")

    def get_rng_state(self, game):
        mt = np.zeros(624, np.uint32)
        pos, hg, gs = C.c_int32(0), C.c_int32(0), C.c_double(0)
        self._check(self._L.ao_get_rng_state(self._h, game, _ptr(mt, C.c_uint32), C.byref(pos), C.byref(hg),
                                             C.byref(gs)), "ao_get_rng_state")
        return mt, pos.value, hg.value, gs.value

    def set_rng_state(self, game, mt, pos, has_gauss=0, gauss=0.0):
        mt = np.ascontiguousarray(mt, np.uint32)
        self._check(self._L.ao_set_rng_state(self._h, game, _ptr(mt, C.c_uint32), int(pos), int(has_gauss),
                                             float(gauss)), "ao_set_rng_state")

    # -- state
    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self._L.ao_reset(self._h, _ptr(m, C.c_uint8)), "ao_reset")

    def set_root(self, game, moves):
        mv = np.ascontiguousarray(list(moves), np.int32)
        st = C.c_int32(0)
        self._check(self._L.ao_set_root(self._h, game, _ptr(mv, C.c_int32), mv.size, C.byref(st)), "ao_set_root")
        return st.value

    def set_roots(self, ids, mask=None):
        """ids: one reference-style root id (0, a1, a2, ...) per game; mask selects the games that move. One
        launch for all of them. Returns the AO_ROOT_* status per game (int32 [G], -2 where unmasked)."""
        mv = np.zeros((self.G, self.A), np.int32)
        n = np.zeros(self.G, np.int32)
        for g, rid in enumerate(ids):
            if mask is not None and not mask[g]:
                continue
            m = list(rid)[1:]
            n[g] = len(m)
            mv[g, :len(m)] = m
        st = np.full(self.G, -2, np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self._L.ao_set_roots(self._h, _ptr(mk, C.c_uint8), _ptr(mv, C.c_int32), self.A, _ptr(n, C.c_int32),
                                         _ptr(st, C.c_int32)), "ao_set_roots")
        return st

    def get_moves(self, game):
        mv = np.zeros(self.A + 1, np.int32)
        n = C.c_int32(0)
        self._check(self._L.ao_get_moves(self._h, game, _ptr(mv, C.c_int32), C.byref(n)), "ao_get_moves")
        return mv[:n.value].tolist()

    # -- stepwise move
    def set_stream(self, stream_ptr):
        self._check(self._L.ao_set_stream(self._h, stream_ptr), "ao_set_stream")

    def sync(self):
        self._check(self._L.ao_sync(self._h), "ao_sync")

    def begin_move(self, active=None):
        a = None if active is None else np.ascontiguousarray(active, np.uint8)
        self._check(self._L.ao_begin_move(self._h, _ptr(a, C.c_uint8)), "ao_begin_move")

    def sims_left(self):
        return self._L.ao_sims_left(self._h)

    def collect_leaves(self, planes_ptr=None):
        self._check(self._L.ao_collect_leaves(self._h, planes_ptr), "ao_collect_leaves")

    def apply_evals(self, policy_ptr, value_ptr):
        self._check(self._L.ao_apply_evals(self._h, policy_ptr, value_ptr), "ao_apply_evals")

    def _outs(self):
        return tuple(np.zeros((self.G, self.A), np.float64) for _ in range(3))

    def end_move(self, tau=None):
        """Returns (pi, visit, policy), float64 [G, A] each."""
        t = None if tau is None else np.ascontiguousarray(np.broadcast_to(tau, (self.G,)), np.int8)
        pi, vis, pol = self._outs()
        self._check(self._L.ao_end_move(self._h, _ptr(t, C.c_int8), _ptr(pi, C.c_double), _ptr(vis, C.c_double),
                                        _ptr(pol, C.c_double)), "ao_end_move")
        return pi, vis, pol

    def play(self):
        """utils.get_action + env.step + re-root. Returns (action[G], win_index[G])."""
        act = np.zeros(self.G, np.int32)
        win = np.zeros(self.G, np.int32)
        self._check(self._L.ao_play(self._h, _ptr(act, C.c_int32), _ptr(win, C.c_int32)), "ao_play")
        return act, win

    # -- fused move with the native network
    def search(self, net, tau=None, active=None):
        t = None if tau is None else np.ascontiguousarray(np.broadcast_to(tau, (self.G,)), np.int8)
        a = None if active is None else np.ascontiguousarray(active, np.uint8)
        pi, vis, pol = self._outs()
        self._check(self._L.ao_search(self._h, net._h, _ptr(a, C.c_uint8), _ptr(t, C.c_int8),
                                      _ptr(pi, C.c_double), _ptr(vis, C.c_double), _ptr(pol, C.c_double)),
                    "ao_search")
        ev, games = self.fp16_range_events()
        if ev != self._fp16_seen:
            import warnings
            warnings.warn("the split-fp16 trunk met an activation beyond the fp16 range (|x| > 65504): this move was searched "
                          "again on the fp32-MFMA trunk with fresh trees for its %d games (%d such move(s) of this engine so far). "
                          "From the third such move of the SAME weights on, the network stays on the fp32-MFMA trunk -- also "
                          "when mode 6 (reproducible=True) had been asked for -- until new weights are loaded or set_mode is "
                          "called; Net.get_mode() tells which kernels run" % (games - self._fp16_games_seen, ev),
                          RuntimeWarning, stacklevel=2)
            self._fp16_seen, self._fp16_games_seen = ev, games
        return pi, vis, pol

    def fp16_range_events(self):
        """(moves ao_search repeated on the fp32-MFMA trunk, games searched again) since the engine was created."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._L.ao_fp16_range_events(self._h, C.byref(a), C.byref(b)), "ao_fp16_range_events")
        return a.value, b.value

    def node_cap(self):
        """(expanded-node capacity of one game's arena, True if it was derived from the free HBM: node_cap=-1)."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self._L.ao_node_cap(self._h, C.byref(a), C.byref(b)), "ao_node_cap")
        return a.value, bool(b.value)

    # -- introspection
    def root_children(self, game):
        A = self.A
        act = np.zeros(A, np.int32)
        n, w, q, p = (np.zeros(A) for _ in range(4))
        cnt = C.c_int32(0)
        self._check(self._L.ao_get_root_children(self._h, game, _ptr(act, C.c_int32), _ptr(n, C.c_double),
                                                 _ptr(w, C.c_double), _ptr(q, C.c_double), _ptr(p, C.c_double),
                                                 C.byref(cnt)), "ao_get_root_children")
        k = cnt.value
        return dict(action=act[:k].copy(), n=n[:k].copy(), w=w[:k].copy(), q=q[:k].copy(), p=p[:k].copy())

    def tree_nodes(self, game):
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._L.ao_tree_nodes(self._h, game, C.byref(a), C.byref(b)), "ao_tree_nodes")
        return a.value, b.value

    def tree_timing(self, enable=True):
        """(total ms, launches) of the TIMED per-simulation tree kernel launches (k_expand_select) since the last call;
        enable = n > 1 times every n-th launch only."""
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._check(self._L.ao_tree_timing(self._h, int(enable), C.byref(ms), C.byref(cnt)), "ao_tree_timing")
        return ms.value, cnt.value

    def trim_stats(self):
        """(child subtrees dropped, re-rootings that dropped any) since the engine was created: non-zero only when
        a game's kept tree outgrew node_cap - sims - 1 nodes (ao_trim_stats)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._L.ao_trim_stats(self._h, C.byref(a), C.byref(b)), "ao_trim_stats")
        return a.value, b.value

    def search_stats(self):
        v = [C.c_int64(0) for _ in range(4)]
        self._check(self._L.ao_search_stats(self._h, *[C.byref(x) for x in v]), "ao_search_stats")
        return dict(levels=v[0].value, ties=v[1].value, terminal=v[2].value, evaluated=v[3].value)

    def set_row_cap(self, rows):
        """rows > 0: `search` hands out the evaluation-batch rows per simulation (terminal leaves take none) and evaluates at
        most `rows` leaves per simulation -- fewer rows than games = over-subscription; 0: the per-move packing (ao_set_row_cap)."""
        self._check(self._L.ao_set_row_cap(self._h, int(rows)), "ao_set_row_cap")
        self.row_cap = int(rows)

    def row_stats(self):
        """dict(launches, rows_live, rows_launched, waits) over the searches that handed out rows per simulation (ao_row_stats)."""
        v = [C.c_int64(0) for _ in range(4)]
        self._check(self._L.ao_row_stats(self._h, *[C.byref(x) for x in v]), "ao_row_stats")
        return dict(launches=v[0].value, rows_live=v[1].value, rows_launched=v[2].value, waits=v[3].value)

    def set_eval_log(self, games, dev_ptr=None, capacity_floats=0):
        """Test hook (ao_set_eval_log): record the (policy, value) the listed games' leaves are evaluated with in `search`."""
        g = np.ascontiguousarray(list(games), np.int32)
        self._check(self._L.ao_set_eval_log(self._h, _ptr(g, C.c_int32) if g.size else None, int(g.size), dev_ptr,
                                            int(capacity_floats)), "ao_set_eval_log")

    def eval_log_count(self):
        return int(self._L.ao_eval_log_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.ao_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
