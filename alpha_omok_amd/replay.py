"""Device-resident replay memory: the reference's `rep_memory = deque(maxlen=MEMORY_SIZE)`
(main.py:55) kept in HBM, with `utils.augment_dataset` (utils.py:226-239) run on the device and
mini-batches assembled on the device (main.py:262-292).

Behaves like the deque for everything main.py does with it: len(), maxlen, extend(),
iteration / indexing (entries come back as the reference's (state f64 [C,B,B], pi f64 [A], z)
tuples), clear(). `extend_augmented(samples)` == `extend(utils.augment_dataset(samples, B))`."""
import ctypes as C

import numpy as np

from . import _lib


class ReplayError(RuntimeError):
    pass


class DeviceReplay:
    def __init__(self, board_size, inplanes, maxlen, device=0):
        self._L = _lib.load()
        self.B, self.C, self.A = int(board_size), int(inplanes), int(board_size) ** 2
        self.device = int(device)
        h = C.c_void_p()
        if self._L.ao_replay_create(self.B, self.C, int(maxlen), self.device, C.byref(h)):
            raise ReplayError(self._L.ao_replay_last_error(None).decode())
        self._h = h

    def _check(self, rc, what):
        if rc:
            raise ReplayError("%s: %s" % (what, self._L.ao_replay_last_error(self._h).decode()))

    @property
    def maxlen(self):
        return int(self._L.ao_replay_capacity(self._h))

    def __len__(self):
        return int(self._L.ao_replay_size(self._h))

    def clear(self):
        self._check(self._L.ao_replay_clear(self._h), "ao_replay_clear")

    def _push(self, samples, augment):
        samples = list(samples)
        if not samples:
            return
        s = np.ascontiguousarray(np.stack([m[0] for m in samples]), dtype=np.float32)
        pi = np.ascontiguousarray(np.stack([m[1] for m in samples]), dtype=np.float64)
        z = np.ascontiguousarray(np.array([m[2] for m in samples]), dtype=np.float32)
        if s.shape[1:] != (self.C, self.B, self.B) or pi.shape[1:] != (self.A,):
            raise ReplayError("sample shapes %r / %r do not match the memory" % (s.shape[1:], pi.shape[1:]))
        self._check(self._L.ao_replay_extend(self._h, s.ctypes.data_as(C.POINTER(C.c_float)),
                                             pi.ctypes.data_as(C.POINTER(C.c_double)),
                                             z.ctypes.data_as(C.POINTER(C.c_float)), len(samples),
                                             1 if augment else 0, None), "ao_replay_extend")

    def extend(self, samples):
        """deque.extend: the samples as they are."""
        self._push(samples, False)

    def extend_augmented(self, samples):
        """rep_memory.extend(utils.augment_dataset(samples, board_size)) with the symmetries made on the device."""
        self._push(samples, True)

    def extend_augmented_arrays(self, states, pi, z):
        """extend_augmented for samples that already are arrays (states [n,C,B,B], pi [n,A], z [n], any float type):
        no per-sample Python objects, and only the samples whose symmetries can survive in the memory are converted
        and uploaded (the rest of the call just moves the ring position, ao_replay_extend_skip)."""
        n = int(states.shape[0])
        if n == 0:
            return
        if tuple(states.shape[1:]) != (self.C, self.B, self.B) or tuple(pi.shape[1:]) != (self.A,):
            raise ReplayError("sample shapes %r / %r do not match the memory" % (states.shape[1:], pi.shape[1:]))
        cap = self.maxlen
        first = 0
        if 8 * n > cap:
            first = max(0, n - (-(-cap // 8) + 1))
            if 8 * (n - first) < cap:
                first = 0
        s = np.ascontiguousarray(states[first:], dtype=np.float32)
        p = np.ascontiguousarray(pi[first:], dtype=np.float64)
        zz = np.ascontiguousarray(z[first:], dtype=np.float32)
        self._check(self._L.ao_replay_extend_skip(self._h, s.ctypes.data_as(C.POINTER(C.c_float)),
                                                  p.ctypes.data_as(C.POINTER(C.c_double)),
                                                  zz.ctypes.data_as(C.POINTER(C.c_float)), n - first, 1, first, None),
                    "ao_replay_extend_skip")

    def extend_augmented_moves(self, moves, ep_of, ply_of, pi, z):
        """extend_augmented_arrays without the states: sample i is the position of episode ep_of[i] after ply_of[i] of its
        moves (moves [E, L] action indices, -1 padded) and the planes of utils.get_state_pt (utils.py:139-168) are built by a
        kernel (ao_replay_extend_moves) -- nothing of size n x C x B x B exists on the host."""
        n = int(np.shape(ep_of)[0])
        if n == 0:
            return
        moves = np.asarray(moves)
        if moves.ndim != 2 or tuple(np.shape(pi)[1:]) != (self.A,) or np.shape(ply_of)[0] != n or np.shape(pi)[0] != n or np.shape(z)[0] != n:
            raise ReplayError("moves [E, L], ep_of / ply_of / z [n] and pi [n, %d] expected" % self.A)
        cap = self.maxlen
        first = 0
        if 8 * n > cap:
            first = max(0, n - (-(-cap // 8) + 1))
            if 8 * (n - first) < cap:
                first = 0
        mv = np.ascontiguousarray(moves[:, :self.A], dtype=np.int16)
        e = np.ascontiguousarray(ep_of[first:], dtype=np.int32)
        t = np.ascontiguousarray(ply_of[first:], dtype=np.int32)
        p = np.ascontiguousarray(pi[first:], dtype=np.float64)
        zz = np.ascontiguousarray(z[first:], dtype=np.float32)
        self._check(self._L.ao_replay_extend_moves(self._h, mv.ctypes.data_as(C.POINTER(C.c_int16)), mv.shape[0], mv.shape[1],
                                                   e.ctypes.data_as(C.POINTER(C.c_int32)), t.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   p.ctypes.data_as(C.POINTER(C.c_double)), zz.ctypes.data_as(C.POINTER(C.c_float)),
                                                   n - first, 1, first, None), "ao_replay_extend_moves")

    def read(self, first, n):
        s = np.empty((n, self.C, self.B, self.B), np.float64)
        pi = np.empty((n, self.A), np.float64)
        z = np.empty(n, np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._L.ao_replay_read(self._h, int(first), int(n), s.ctypes.data_as(dp), pi.ctypes.data_as(dp),
                                           z.ctypes.data_as(dp)), "ao_replay_read")
        return s, pi, z

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            idx = range(*i.indices(n))
            return [self[j] for j in idx]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("replay index out of range")
        s, pi, z = self.read(i, 1)
        return s[0], pi[0], float(z[0])

    def __iter__(self):
        n = len(self)
        step = 4096
        for f in range(0, n, step):
            s, pi, z = self.read(f, min(step, n - f))
            for k in range(s.shape[0]):
                yield s[k], pi[k], float(z[k])

    def batch(self, indices):
        """float32 cuda tensors (s [m,C,B,B], pi [m,A], z [m]) for deque indices -- what main.train builds with
        torch.tensor(np.stack(...)).to(device).float() (main.py:283-290)."""
        import torch
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int64))
        m = int(idx.shape[0])
        dev = torch.device("cuda", self.device)
        s = torch.empty((m, self.C, self.B, self.B), dtype=torch.float32, device=dev)
        pi = torch.empty((m, self.A), dtype=torch.float32, device=dev)
        z = torch.empty((m,), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._check(self._L.ao_replay_gather(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), m, s.data_ptr(),
                                             pi.data_ptr(), z.data_ptr(), stream), "ao_replay_gather")
        return s, pi, z

    def close(self):
        if getattr(self, "_h", None):
            self._L.ao_replay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
