"""The per-move UCT search of the reference's 1_tictactoe_MCTS/mcts_vs.py (BASELINE configs[0]) on the
device: MCTS.selection / expansion / simulation / backup (mcts_vs.py:15-131) and the driver loop with
its q_list / max_action (mcts_vs.py:153-183), one kernel per call, one wavefront per board.

    max_action, q_list = uct_search(game_board, turn, num_mcts)      # uses and advances Python's `random`

replaces, in mcts_vs.py's __main__, the `for i in range(num_mcts)` loop and the q_list arg-max."""
import ctypes as C
import random

import numpy as np

from . import _lib


class TttError(RuntimeError):
    pass


class TttEngine:
    def __init__(self, num_mcts, games=1, board_size=3, win_mark=0, device=0):
        self._L = _lib.load()
        self.B, self.A, self.S, self.G = int(board_size), int(board_size) ** 2, int(num_mcts), int(games)
        cfg = _lib.AoTttConfig(board=self.B, win_mark=int(win_mark), sims=self.S, games=self.G, device=int(device))
        h = C.c_void_p()
        if self._L.ao_ttt_create(C.byref(cfg), C.byref(h)):
            raise TttError(self._L.ao_ttt_last_error(None).decode())
        self._h = h

    def _check(self, rc, what):
        if rc:
            raise TttError("%s: %s" % (what, self._L.ao_ttt_last_error(self._h).decode()))

    def seed(self, game, seed):
        """random.seed(seed) for this game's stream (int < 2**32)."""
        self._check(self._L.ao_ttt_seed(self._h, int(game), int(seed) & 0xFFFFFFFF), "ao_ttt_seed")

    def get_rng_state(self, game):
        mt = np.zeros(624, np.uint32)
        pos = C.c_int32(0)
        self._check(self._L.ao_ttt_get_rng_state(self._h, int(game), mt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 C.byref(pos)), "ao_ttt_get_rng_state")
        return mt, pos.value

    def set_rng_state(self, game, mt, pos):
        mt = np.ascontiguousarray(mt, np.uint32)
        self._check(self._L.ao_ttt_set_rng_state(self._h, int(game), mt.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos)),
                    "ao_ttt_set_rng_state")

    def search(self, boards, turns, active=None):
        """boards [G,B,B] (+1 O, -1 X), turns [G]. Returns (max_action [G], q [G,A] (-inf for non-children), n [G,A])."""
        b = np.ascontiguousarray(np.asarray(boards).reshape(self.G, self.A), dtype=np.int8)
        t = np.ascontiguousarray(turns, dtype=np.int32).reshape(self.G)
        q = np.zeros((self.G, self.A), np.float64)
        n = np.zeros((self.G, self.A), np.float64)
        act = np.zeros(self.G, np.int32)
        ap = None
        if active is not None:
            a8 = np.ascontiguousarray(active, np.uint8)
            ap = a8.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self._L.ao_ttt_search(self._h, b.ctypes.data_as(C.POINTER(C.c_int8)),
                                          t.ctypes.data_as(C.POINTER(C.c_int32)), ap,
                                          q.ctypes.data_as(C.POINTER(C.c_double)), n.ctypes.data_as(C.POINTER(C.c_double)),
                                          act.ctypes.data_as(C.POINTER(C.c_int32))), "ao_ttt_search")
        return act, q, n

    def close(self):
        if getattr(self, "_h", None):
            self._L.ao_ttt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_engines = {}


def uct_search(game_board, turn, num_mcts, win_mark=3, device=0):
    """mcts_vs.py:153-183 for one board, on Python's process-global `random` stream (moved into the
    engine and written back, so `random.seed(s)` reproduces the reference's search and leaves the
    stream where the reference leaves it). Returns (max_action, {(0, a): q})."""
    gb = np.asarray(game_board)
    key = (gb.shape[0], int(num_mcts), int(win_mark), int(device))
    eng = _engines.get(key)
    if eng is None:
        eng = _engines[key] = TttEngine(num_mcts, games=1, board_size=gb.shape[0], win_mark=win_mark, device=device)
    ver, st, gauss = random.getstate()
    eng.set_rng_state(0, np.array(st[:624], np.uint32), st[624])
    act, q, n = eng.search(gb[None], [turn])
    mt, pos = eng.get_rng_state(0)
    random.setstate((ver, tuple(int(x) for x in mt) + (int(pos),), gauss))
    q_list = {(0, a): float(q[0, a]) for a in range(gb.size) if q[0, a] != -np.inf}
    return int(act[0]), q_list
