"""Python binding of the rollout-search part of libomok_hip.so (ao_rollout_*): the reference's
PUCTAgent / UCTAgent searches (agents.py:263-614) for G independent games, one kernel per get_pi."""
import ctypes as C

import numpy as np

from . import _lib

PUCT, UCT = 0, 1


class RolloutError(RuntimeError):
    pass


class RolloutEngine:
    def __init__(self, board_size, num_mcts, mode, games=1, device=0, win_mark=0, c_puct=0.0):
        self._L = _lib.load()
        self.B, self.A, self.S, self.G, self.mode = int(board_size), int(board_size) ** 2, int(num_mcts), int(games), int(mode)
        cfg = _lib.AoRolloutConfig(board=self.B, win_mark=int(win_mark), sims=self.S, games=self.G, mode=self.mode,
                                   device=int(device), c_puct=float(c_puct))
        h = C.c_void_p()
        if self._L.ao_rollout_create(C.byref(cfg), C.byref(h)):
            raise RolloutError(self._L.ao_rollout_last_error(None).decode())
        self._h = h

    def _check(self, rc, what):
        if rc:
            raise RolloutError("%s: %s" % (what, self._L.ao_rollout_last_error(self._h).decode()))

    def seed(self, game, seed):
        self._check(self._L.ao_rollout_seed(self._h, int(game), int(seed) & 0xFFFFFFFF), "ao_rollout_seed")

    def get_rng_state(self, game):
        mt = np.zeros(624, np.uint32)
        pos, hg, gs = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        self._check(self._L.ao_rollout_get_rng_state(self._h, int(game), mt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     C.byref(pos), C.byref(hg), C.byref(gs)), "ao_rollout_get_rng_state")
        return mt, pos.value, hg.value, gs.value

    def set_rng_state(self, game, mt, pos, has_gauss=0, gauss=0.0):
        mt = np.ascontiguousarray(mt, np.uint32)
        self._check(self._L.ao_rollout_set_rng_state(self._h, int(game), mt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     int(pos), int(has_gauss), float(gauss)), "ao_rollout_set_rng_state")

    def search(self, root_ids, active=None):
        """root_ids: one reference node id (0, a1, a2, ...) per game. Returns (pi [G,A] one-hot,
        stat [G,A] = child visits (PUCT) / child q (UCT, -inf elsewhere), action [G])."""
        if len(root_ids) != self.G:
            raise RolloutError("expected %d root ids" % self.G)
        moves = np.zeros((self.G, self.A), np.int32)
        nm = np.zeros(self.G, np.int32)
        for g, rid in enumerate(root_ids):
            mv = list(rid)[1:]
            nm[g] = len(mv)
            if len(mv) > self.A:
                raise RolloutError("root id longer than the board")
            moves[g, :len(mv)] = mv
        pi = np.zeros((self.G, self.A), np.float64)
        stat = np.zeros((self.G, self.A), np.float64)
        act = np.zeros(self.G, np.int32)
        ap = None
        if active is not None:
            a8 = np.ascontiguousarray(active, np.uint8)
            ap = a8.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self._L.ao_rollout_search(self._h, moves.ctypes.data_as(C.POINTER(C.c_int32)),
                                              nm.ctypes.data_as(C.POINTER(C.c_int32)), ap,
                                              pi.ctypes.data_as(C.POINTER(C.c_double)),
                                              stat.ctypes.data_as(C.POINTER(C.c_double)),
                                              act.ctypes.data_as(C.POINTER(C.c_int32))), "ao_rollout_search")
        return pi, stat, act

    def close(self):
        if getattr(self, "_h", None):
            self._L.ao_rollout_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
