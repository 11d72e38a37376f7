"""Host-side helpers with the reference's `utils` names and results (utils.py:22-239).

Inside a search these are HIP device code (tree_kernels.hip); the functions here serve the callers
on either side of the hot path: building training samples, sampling the played move from the
process-global numpy stream, data augmentation. Implementations are vectorised numpy.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view


def legal_actions(node_id, board_size):
    """Empty cells in the reference's order (utils.py:22-27): iteration order of a CPython set
    difference, ascending except in deep endgames."""
    return list(set(range(board_size * board_size)) - set(node_id[1:]))


def get_board(node_id, board_size):
    """+1 black / -1 white, black moves first (utils.py:171-179). float64 [B, B]."""
    flat = np.zeros(board_size * board_size)
    mv = np.asarray(node_id[1:], dtype=np.int64)
    flat[mv[0::2]] = 1.0
    flat[mv[1::2]] = -1.0
    return flat.reshape(board_size, board_size)


def get_turn(node_id):
    """0 = black to move, 1 = white to move (utils.py:182-186)."""
    return 0 if len(node_id) % 2 == 1 else 1


def check_win(board, win_mark):
    """0 playing / 1 black / 2 white / 3 draw (utils.py:30-59). Windows are visited row-major and
    black is tested before white inside a window, like the reference's scan."""
    b = np.asarray(board)
    n = b.shape[0]
    k = win_mark
    if n >= k:
        win = sliding_window_view(b, (k, k))                      # [n-k+1, n-k+1, k, k]
        rows = win.sum(axis=3)                                    # horizontal lines
        cols = win.sum(axis=2)                                    # vertical lines
        d1 = np.trace(win, axis1=2, axis2=3)
        d2 = np.trace(win[:, :, ::-1, :], axis1=2, axis2=3)
        black = (rows == k).any(axis=2) | (cols == k).any(axis=2) | (d1 == k) | (d2 == k)
        white = (rows == -k).any(axis=2) | (cols == -k).any(axis=2) | (d1 == -k) | (d2 == -k)
        hit = np.flatnonzero((black | white).ravel())
        if hit.size:
            return 1 if black.ravel()[hit[0]] else 2
    if np.count_nonzero(b) == n * n:
        return 3
    return 0


def get_state_pt(node_id, board_size, channel_size):
    """Network input planes, float64 [C, B, B] (utils.py:139-168): the stones of the mover of each
    of the last C-1 plies as they stood after that ply, oldest first, then the colour plane."""
    A = board_size * board_size
    mv = np.asarray(node_id[1:], dtype=np.int64)
    k = mv.size
    planes = np.zeros((channel_size, A))
    for j in range(channel_size - 1):
        ply = k - j                      # X_{k-j}
        if ply < 1:
            continue
        own = mv[(ply - 1) % 2:ply:2]    # moves of that ply's mover up to and including it
        planes[channel_size - 2 - j, own] = 1.0
    planes[channel_size - 1, :] = 1.0 if k % 2 == 0 else 0.0
    return planes.reshape(channel_size, board_size, board_size)


def states_of_episodes(moves, ep_of, ply_of, board_size, channel_size, dtype=np.float64, chunk=256):
    """get_state_pt for MANY positions at once: sample i is the root (0, m_1, ..., m_t) of episode ep_of[i] after
    t = ply_of[i] of its moves (moves [E, L] int, -1 padded). Returns [N, C, B, B] of `dtype`, equal entry for entry
    to np.stack([get_state_pt(...)]) (utils.py:139-168) -- cumulative stone sets per colour, then one gather per
    history plane instead of a Python call per sample. Episodes are processed `chunk` at a time, so the temporaries
    (one-hot and cumulative stone sets, [chunk, L, A] bytes each) stay at a few MB whatever E is."""
    moves = np.asarray(moves, dtype=np.int64)
    ep_of = np.asarray(ep_of, dtype=np.int64)
    ply_of = np.asarray(ply_of, dtype=np.int64)
    E, L = moves.shape
    A = board_size * board_size
    N = ep_of.shape[0]
    out = np.zeros((N, channel_size, A), dtype)
    out[:, channel_size - 1, :] = (ply_of % 2 == 0).astype(dtype)[:, None]
    if N == 0:
        return out.reshape(N, channel_size, board_size, board_size)
    by_ep = np.argsort(ep_of, kind="stable")              # (already sorted when the samples come from main.self_play)
    bounds = np.searchsorted(ep_of[by_ep], np.arange(0, E + chunk, chunk))
    par = (np.arange(L) % 2)[None, :, None]
    for c, e0 in enumerate(range(0, E, chunk)):
        sel = by_ep[bounds[c]:bounds[c + 1]]
        if sel.size == 0:
            continue
        mv = moves[e0:e0 + chunk]
        # cum[col, e, j] = stones of colour col (0 black: even move index) after the first j+1 moves of episode e0 + e
        onehot = np.zeros((mv.shape[0], L, A), np.uint8)
        ee, jj = np.nonzero(mv >= 0)
        onehot[ee, jj, mv[ee, jj]] = 1
        cum = np.empty((2,) + onehot.shape, np.uint8)
        np.cumsum(onehot * (par == 0), axis=1, dtype=np.uint8, out=cum[0])
        np.cumsum(onehot * (par == 1), axis=1, dtype=np.uint8, out=cum[1])
        e_loc, t = ep_of[sel] - e0, ply_of[sel]
        for j in range(channel_size - 1):
            p = t - j                         # X_p: the mover of (1-based) ply p after that ply; black moves the odd plies
            ok = np.flatnonzero(p >= 1)
            if ok.size:
                pp = p[ok]
                out[sel[ok], channel_size - 2 - j] = cum[(pp - 1) % 2, e_loc[ok], pp - 1]
    return out.reshape(N, channel_size, board_size, board_size)


class LazySamples:
    """The samples of one main.self_play call -- (state [C, B, B] f64, pi [A] f64, z float) as main.py:159-166 appends them --
    without the states: they are rebuilt (states_of_episodes, all of the block at once) the first time anyone looks at one.
    With rep_memory on the device the states are made there (ao_replay_extend_moves) and cur_memory is only ever counted,
    so nothing of size n x C x B x B is built on the host. Iterating yields one tuple-like entry per sample."""

    def __init__(self, moves, ep_of, ply_of, pis, z, board_size, channel_size):
        self.moves, self.ep_of, self.ply_of, self.pis = moves, ep_of, ply_of, pis
        self.z = np.asarray(z, dtype=np.float64).tolist()                     # Python floats, as the reference stores them
        self.board_size, self.channel_size = board_size, channel_size
        self._states = None

    def __len__(self):
        return len(self.z)

    def states(self):
        if self._states is None:
            self._states = states_of_episodes(self.moves, self.ep_of, self.ply_of, self.board_size, self.channel_size)
        return self._states

    def __iter__(self):
        return (LazySample(self, i) for i in range(len(self)))


class SampleQueue:
    """main.cur_memory: the reference's deque of (state, pi, z) tuples (main.py:56) -- len / iteration / indexing / append /
    extend / pop / popleft / clear -- that can also take a whole LazySamples block by reference: extend(block) is O(1), no
    per-sample object exists until somebody iterates or indexes (main.train only ever counts cur_memory)."""
    maxlen = None

    def __init__(self, iterable=()):
        self._seg = []          # [block, lo, hi] (LazySamples, live range) or a plain list of entries
        self.extend(iterable)

    def __len__(self):
        return sum(g[2] - g[1] if _is_block(g) else len(g) for g in self._seg)

    def __bool__(self):
        return len(self) > 0

    def _tail_list(self):
        if not self._seg or _is_block(self._seg[-1]):
            self._seg.append([])
        return self._seg[-1]

    def append(self, x):
        self._tail_list().append(x)

    def extend(self, iterable):
        if isinstance(iterable, LazySamples):
            if len(iterable):
                self._seg.append(_Block((iterable, 0, len(iterable))))
        else:
            self._tail_list().extend(iterable)

    def clear(self):
        self._seg = []

    def _drop_empty(self):
        self._seg = [g for g in self._seg if (g[2] - g[1] if _is_block(g) else len(g)) > 0]

    def pop(self):
        self._drop_empty()
        if not self._seg:
            raise IndexError("pop from an empty SampleQueue")
        g = self._seg[-1]
        if _is_block(g):
            g[2] -= 1
            return LazySample(g[0], g[2])
        return g.pop()

    def popleft(self):
        self._drop_empty()
        if not self._seg:
            raise IndexError("pop from an empty SampleQueue")
        g = self._seg[0]
        if _is_block(g):
            g[1] += 1
            return LazySample(g[0], g[1] - 1)
        return g.pop(0)

    def __iter__(self):
        for g in list(self._seg):
            if _is_block(g):
                for i in range(g[1], g[2]):
                    yield LazySample(g[0], i)
            else:
                yield from list(g)

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("SampleQueue index out of range")
        for g in self._seg:
            m = g[2] - g[1] if _is_block(g) else len(g)
            if i < m:
                return LazySample(g[0], g[1] + i) if _is_block(g) else g[i]
            i -= m
        raise IndexError("SampleQueue index out of range")


class _Block(list):
    """[LazySamples, lo, hi]: a block segment of a SampleQueue (a list subclass so that it is told apart from a list of entries)"""


def _is_block(g):
    return type(g) is _Block


class LazySample:
    """One entry of LazySamples: unpacks, indexes and compares like the tuple (state, pi, z)."""
    __slots__ = ("_blk", "_i")

    def __init__(self, blk, i):
        self._blk, self._i = blk, i

    def _tuple(self):
        b, i = self._blk, self._i
        return (b.states()[i], b.pis[i], b.z[i])

    def __iter__(self):
        return iter(self._tuple())

    def __getitem__(self, k):
        return self._tuple()[k]

    def __len__(self):
        return 3


def get_action(pi):
    """Sample the played move from np.random (utils.py:189-195). Returns (one-hot, index)."""
    n = len(pi)
    idx = np.random.choice(n, p=pi)
    onehot = np.zeros(n)
    onehot[idx] = 1
    return onehot, idx


def argmax_onehot(pi):
    """Uniform choice among the maxima of pi (utils.py:198-205). Returns (one-hot, index)."""
    best = np.flatnonzero(pi == pi.max())
    idx = best[np.random.choice(len(best))]
    onehot = np.zeros(len(pi))
    onehot[idx] = 1
    return onehot, idx


def augment_dataset(memory, board_size):
    """8 symmetries per sample in the reference's order r0, r0f, r1, r1f, ... (utils.py:226-239)."""
    out = []
    for s, pi, z in memory:
        grid = pi.reshape(board_size, board_size)
        for r in range(4):
            s_r = np.rot90(s, r, axes=(1, 2))
            p_r = np.rot90(grid, r)
            out.append((s_r.copy(), p_r.reshape(-1).copy(), z))
            out.append((s_r[:, :, ::-1].copy(), p_r[:, ::-1].reshape(-1).copy(), z))
    return out
