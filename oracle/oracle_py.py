"""ctypes binding of the CPU oracle (oracle/libomok_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product package (alpha_omok_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libomok_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("omok_oracle.c", "rollout_oracle.c", "omok_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "libomok_oracle.so"])
    return _SO


class _Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("pos", C.c_int), ("has_gauss", C.c_int),
                ("gauss", C.c_double)]


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_float),
                      C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    P = C.POINTER
    L.oo_rng_seed.argtypes = [P(_Rng), C.c_uint32]
    L.oo_rng_next32.argtypes = [P(_Rng)]
    L.oo_rng_next32.restype = C.c_uint32
    L.oo_rng_double.argtypes = [P(_Rng)]
    L.oo_rng_double.restype = C.c_double
    L.oo_rng_below.argtypes = [P(_Rng), C.c_int64]
    L.oo_rng_below.restype = C.c_int64
    L.oo_rng_dirichlet.argtypes = [P(_Rng), C.c_double, C.c_int, P(C.c_double)]
    L.oo_rng_choice_p.argtypes = [P(_Rng), P(C.c_double), C.c_int]
    L.oo_rng_choice_p.restype = C.c_int
    L.oo_legal_actions.argtypes = [P(C.c_int), C.c_int, C.c_int, P(C.c_int)]
    L.oo_legal_actions.restype = C.c_int
    L.oo_check_win.argtypes = [P(C.c_int8), C.c_int, C.c_int]
    L.oo_check_win.restype = C.c_int
    L.oo_get_board.argtypes = [P(C.c_int), C.c_int, C.c_int, P(C.c_int8)]
    L.oo_get_state_pt.argtypes = [P(C.c_int), C.c_int, C.c_int, C.c_int, P(C.c_float)]
    L.oo_pairwise_sum.argtypes = [P(C.c_double), C.c_int]
    L.oo_pairwise_sum.restype = C.c_double
    L.oo_agent_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.oo_agent_create.restype = C.c_void_p
    L.oo_agent_destroy.argtypes = [C.c_void_p]
    L.oo_agent_set_eval.argtypes = [C.c_void_p, EVAL_FN, C.c_void_p]
    L.oo_agent_set_win_mark.argtypes = [C.c_void_p, C.c_int]
    L.oo_agent_use_stub.argtypes = [C.c_void_p, C.c_int]
    L.oo_agent_rng.argtypes = [C.c_void_p]
    L.oo_agent_rng.restype = P(_Rng)
    L.oo_agent_reset.argtypes = [C.c_void_p]
    L.oo_agent_get_pi.argtypes = [C.c_void_p, P(C.c_int), C.c_int, C.c_int, P(C.c_double),
                                  P(C.c_double), P(C.c_double)]
    L.oo_agent_get_pi.restype = C.c_int
    L.oo_agent_children.argtypes = [C.c_void_p, P(C.c_int), C.c_int, P(C.c_double),
                                    P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_int)]
    L.oo_agent_children.restype = C.c_int
    L.oo_agent_tree_size.argtypes = [C.c_void_p]
    L.oo_agent_tree_size.restype = C.c_long
    L.oo_agent_num_evals.argtypes = [C.c_void_p]
    L.oo_agent_num_evals.restype = C.c_long
    L.oo_agent_last_stats.argtypes = [C.c_void_p, P(C.c_long), P(C.c_long), P(C.c_long)]
    L.oo_stub_eval_planes.argtypes = [P(C.c_float), C.c_int, C.c_int, C.c_int, P(C.c_float),
                                      P(C.c_float)]
    L.oo_self_play_game.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, P(C.c_int),
                                    P(C.c_double), P(C.c_double), P(C.c_int)]
    L.oo_self_play_game.restype = C.c_int
    L.oo_rollout_search.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_int), C.c_int, P(_Rng),
                                    P(C.c_double), P(C.c_double), P(C.c_int)]
    L.oo_rollout_search.restype = C.c_int
    L.oo_pyrandom_seed.argtypes = [P(_Rng), C.c_uint32]
    L.oo_pyrandom_below.argtypes = [P(_Rng), C.c_int]
    L.oo_pyrandom_below.restype = C.c_int
    L.oo_ttt_search.argtypes = [C.c_int, C.c_int, C.c_int, P(C.c_int8), C.c_int, P(_Rng), P(C.c_double),
                                P(C.c_double)]
    L.oo_ttt_search.restype = C.c_int
    _lib = L
    return L


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Rng:
    """numpy legacy RandomState restatement (np.random.seed / choice / dirichlet ...)."""

    def __init__(self, seed=0, _ptr=None):
        self._own = _Rng() if _ptr is None else None
        self._p = C.pointer(self._own) if _ptr is None else _ptr
        if _ptr is None:
            self.seed(seed)

    def seed(self, s):
        lib().oo_rng_seed(self._p, s)

    def next32(self):
        return lib().oo_rng_next32(self._p)

    def random_sample(self):
        return lib().oo_rng_double(self._p)

    def choice(self, k):
        return lib().oo_rng_below(self._p, k)

    def dirichlet(self, alpha, k):
        out = np.zeros(max(k, 1), np.float64)
        lib().oo_rng_dirichlet(self._p, alpha, k, _dp(out))
        return out[:k]

    def choice_p(self, p):
        p = np.ascontiguousarray(p, np.float64)
        return lib().oo_rng_choice_p(self._p, _dp(p), len(p))

    @property
    def pos(self):
        return self._p.contents.pos

    def state_words(self):
        return np.ctypeslib.as_array(self._p.contents.mt).copy()

    def set_state(self, words, pos, has_gauss=0, gauss=0.0):
        st = self._p.contents
        for i, w in enumerate(np.asarray(words, np.uint32).tolist()):
            st.mt[i] = w
        st.pos = int(pos)
        st.has_gauss = int(has_gauss)
        st.gauss = float(gauss)


def legal_actions(moves, board):
    m = np.ascontiguousarray(moves, np.int32)
    out = np.zeros(board * board, np.int32)
    n = lib().oo_legal_actions(_ip(m), len(m), board, _ip(out))
    return out[:n].copy()


def check_win(board_arr, win_mark):
    b = np.ascontiguousarray(board_arr, np.int8)
    return lib().oo_check_win(b.ctypes.data_as(C.POINTER(C.c_int8)), b.shape[0], win_mark)


def get_board(moves, board):
    m = np.ascontiguousarray(moves, np.int32)
    out = np.zeros((board, board), np.int8)
    lib().oo_get_board(_ip(m), len(m), board, out.ctypes.data_as(C.POINTER(C.c_int8)))
    return out


def get_state_pt(moves, board, channels):
    m = np.ascontiguousarray(moves, np.int32)
    out = np.zeros((channels, board, board), np.float32)
    lib().oo_get_state_pt(_ip(m), len(m), board, channels, _fp(out))
    return out


def pairwise_sum(a):
    a = np.ascontiguousarray(a, np.float64)
    return lib().oo_pairwise_sum(_dp(a), len(a))


def stub_eval(planes, mode=0):
    """planes float32 [C,B,B] -> (policy float32 [A], value float32)."""
    pl = np.ascontiguousarray(planes, np.float32)
    Cn, B, _ = pl.shape
    pol = np.zeros(B * B, np.float32)
    val = C.c_float(0)
    lib().oo_stub_eval_planes(_fp(pl), B, Cn, mode, _fp(pol), C.byref(val))
    return pol, np.float32(val.value)


def rollout_search(mode, board, num_mcts, root_id, rng, win_mark=0):
    """PUCTAgent.get_pi (mode 0) / UCTAgent.get_pi (mode 1) on root_id with `rng` as np.random.
    Returns (pi one-hot [A], stat [A] = child visits (PUCT) / child q (UCT, -inf elsewhere), action, nodes)."""
    moves = np.ascontiguousarray(list(root_id)[1:], dtype=np.int32)
    A = board * board
    pi = np.zeros(A, np.float64)
    stat = np.zeros(A, np.float64)
    act = C.c_int(0)
    nodes = lib().oo_rollout_search(mode, board, win_mark, num_mcts, _ip(moves) if len(moves) else None,
                                    len(moves), rng._p, _dp(pi), _dp(stat), C.byref(act))
    return pi, stat, act.value, nodes


class PyRandom(Rng):
    """Python's `random` module stream (random.seed(int) / _randbelow) on the same MT19937 core."""

    def seed(self, s):
        lib().oo_pyrandom_seed(self._p, s)

    def randbelow(self, n):
        return lib().oo_pyrandom_below(self._p, n)


def ttt_search(game_board, turn, num_mcts, rng, win_mark=3):
    """The per-move UCT search of 1_tictactoe_MCTS/mcts_vs.py:153-183. Returns (max_action, q [A], n [A])."""
    b = np.ascontiguousarray(np.asarray(game_board), dtype=np.int8)
    B = b.shape[0]
    q = np.zeros(B * B, np.float64)
    n = np.zeros(B * B, np.float64)
    a = lib().oo_ttt_search(B, win_mark, num_mcts, b.ctypes.data_as(C.POINTER(C.c_int8)), int(turn), rng._p, _dp(q), _dp(n))
    return a, q, n


class Agent:
    """Oracle ZeroAgent. evaluator: 'stub0'/'stub1'/'stub2' or a python callable
    f(moves, planes[C,B,B] f32, sim) -> (policy[A] f32, value f32)."""

    def __init__(self, board, num_mcts, inplanes=5, noise=True, evaluator="stub0"):
        self.board, self.A, self.C = board, board * board, inplanes
        self._h = lib().oo_agent_create(board, num_mcts, inplanes, 1 if noise else 0)
        self.rng = Rng(_ptr=lib().oo_agent_rng(self._h))
        self._cb = None
        self.set_evaluator(evaluator)

    def set_evaluator(self, evaluator):
        if isinstance(evaluator, str):
            assert evaluator.startswith("stub")
            lib().oo_agent_use_stub(self._h, int(evaluator[4:] or 0))
            self._cb = None
            return
        A, Cn, B = self.A, self.C, self.board

        def _cb(ctx, moves, nmoves, planes, sim, policy, value):
            mv = [moves[i] for i in range(nmoves)]
            pl = np.ctypeslib.as_array(planes, shape=(Cn, B, B))
            p, v = evaluator(mv, pl, sim)
            np.ctypeslib.as_array(policy, shape=(A,))[:] = np.asarray(p, np.float32)
            value[0] = float(np.float32(v))

        self._cb = EVAL_FN(_cb)
        lib().oo_agent_set_eval(self._h, self._cb, None)

    def seed(self, s):
        self.rng.seed(s)

    def set_win_mark(self, k):
        """ZeroAgent.win_mark (agents.py:41-44): stones in a row that win; above the board size only the full board ends a game."""
        lib().oo_agent_set_win_mark(self._h, int(k))

    def reset(self):
        lib().oo_agent_reset(self._h)

    def get_pi(self, root_id, tau):
        """root_id is the reference's id tuple (0, a1, a2, ...). Returns (pi, visit, policy)."""
        m = np.ascontiguousarray(list(root_id)[1:], np.int32)
        pi, vis, pol = (np.zeros(self.A) for _ in range(3))
        lib().oo_agent_get_pi(self._h, _ip(m), len(m), int(tau), _dp(pi), _dp(vis), _dp(pol))
        return pi, vis, pol

    def children(self, node_id):
        m = np.ascontiguousarray(list(node_id)[1:], np.int32)
        cn, cw, cq, cp = (np.zeros(self.A) for _ in range(4))
        order = np.zeros(self.A, np.int32)
        k = lib().oo_agent_children(self._h, _ip(m), len(m), _dp(cn), _dp(cw), _dp(cq), _dp(cp),
                                    _ip(order))
        if k < 0:
            return None
        return dict(n=cn, w=cw, q=cq, p=cp, order=order[:k].copy())

    def tree_size(self):
        return lib().oo_agent_tree_size(self._h)

    def num_evals(self):
        return lib().oo_agent_num_evals(self._h)

    def last_stats(self):
        a, b, c = C.c_long(0), C.c_long(0), C.c_long(0)
        lib().oo_agent_last_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(levels=a.value, ties=b.value, terminal_leaves=c.value)

    def self_play_game(self, seed, tau_thres=6, max_plies=0):
        """main.self_play's episode loop. Returns (moves, pis[plies,A], visits[plies,A], win)."""
        moves = np.zeros(self.A + 1, np.int32)
        pis = np.zeros((self.A + 1, self.A))
        vis = np.zeros((self.A + 1, self.A))
        win = C.c_int(0)
        n = lib().oo_self_play_game(self._h, seed, tau_thres, max_plies, _ip(moves), _dp(pis),
                                    _dp(vis), C.byref(win))
        return moves[:n].copy(), pis[:n].copy(), vis[:n].copy(), win.value

    def __del__(self):
        try:
            lib().oo_agent_destroy(self._h)
        except Exception:
            pass
