// rollout.hip -- the reference's net-free rollout agents as one HIP kernel.
//
// Reference (paths relative to /root/reference/2_AlphaOmok/):
//   PUCTAgent  agents.py:263-441  PUCT with uniform priors 1/len(actions) + random playout
//   UCTAgent   agents.py:443-614  UCB1 (q + sqrt(2 ln N / n), unvisited = +inf) + random playout
//   utils.valid_actions (utils.py:8-19, ascending cells), utils.get_reward (utils.py:208-223)
//
// One wavefront owns one game and runs the WHOLE search of a get_pi call -- num_mcts + 1
// simulations of selection, expansion, playout and backup -- inside a single launch: there is no
// network to wait for, so nothing ever returns to the host. The tree is a structure of arrays in
// HBM (per expanded node one row of edge counters N, W and child links CH); the statistics the
// reference keeps on child nodes live on the parent's edges, w is an exact integer (rewards are
// +1 / -1 / 0) and q = w / n is formed in float64 where it is used, which is bit-identical to
// the stored Python float. Draws come from the game's numpy-legacy MT19937 stream in the
// reference's order: tie-breaks of the descent, playout moves, final arg-max tie. ln(total_n) comes
// from a host table (glibc log; see oracle/rollout_oracle.c on numpy's AVX512 log).
// Every get_pi of the reference is a fresh search (_init_mcts overwrites the root node), so there
// is no tree re-use to carry between calls.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/omok_hip.h"
#include "host_rng.hpp"
#include "tree_device.hpp"

namespace ao {

struct RollParams {
    int B, A, Ap, win_mark, G, cap, sims, mode;  // mode 0 = PUCT, 1 = UCT
    double c_puct;
    int32_t* N; int32_t* W; int32_t* CH;          // [G][cap][Ap]
    int16_t* NK;                                  // [G][cap] children of an expanded node
    Pos* rootpos;                                 // [G]
    uint32_t* mt; int32_t* mtpos;                 // [G][624], [G]
    const double* log_lut;                        // ln(n), n = 0 .. sims + 1 (host libm)
    double* out_pi; double* out_stat;             // [G][A]
    int32_t* action; int32_t* err;                // [G]
    const uint8_t* active;                        // [G] or null
};

// the r-th (0-based) empty cell of s in ascending order (wave-uniform; r < number of empties)
template <int NCH>
__device__ __forceinline__ int nth_empty(const Pos& s, int A, int r) {
    const int lane = lane_id();
    int cell = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        const uint64_t m = __ballot(i < A && !pos_occupied(s, i));
        const int cnt = __popcll(m);
        if (cell < 0) {
            if (r < cnt) cell = 64 * c + nth_set_bit(m, r);
            else r -= cnt;
        }
    }
    return cell;
}

template <int NCH>
__global__ __launch_bounds__(64) void k_rollout_search(RollParams p) {
    __shared__ uint32_t s_mt[624];
    __shared__ int32_t s_pnode[kMaxCells + 2];
    __shared__ int16_t s_pedge[kMaxCells + 2];
    const int g = blockIdx.x;
    const int lane = lane_id();
    if (p.active && !p.active[g]) return;
    MtDev mt;
    mt.open(p.mt + static_cast<size_t>(g) * 624, p.mtpos + g, s_mt);
    const size_t gbase = static_cast<size_t>(g) * p.cap;
    const Pos root = p.rootpos[g];
    int used = 0;

    auto new_node = [&](int nk) -> int {  // rows of a freshly expanded node: n = w = 0, children unvisited
        const int nd = used++;
        const size_t eb = (gbase + nd) * p.Ap;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e = lane + 64 * c;
            if (e < nk) { p.N[eb + e] = 0; p.W[eb + e] = 0; p.CH[eb + e] = CH_UNVISITED; }
        }
        if (lane == 0) p.NK[gbase + nd] = static_cast<int16_t>(nk);
        return nd;
    };

    for (int s = 0; s <= p.sims; ++s) {
        if (s == 0) {  // first simulation: the root is the leaf, it is expanded, reward 0, no playout
            new_node(p.A - root.ply);
            wsync();
            continue;
        }
        if (used >= p.cap - 1) {
            if (lane == 0) atomicOr(&p.err[g], ERR_NODE_CAP);
            break;
        }
        Pos cur = root;
        int node = 0, depth = 0, reward = 0;
        for (;;) {
            // one memory round trip per level: child count, statistics and child links requested together, unconditionally
            // (lanes past Ap re-read edge Ap-1, everything below masks by e < L), and pinned -- as `e < L ? row[e] : 0` the rows
            // waited for L, and the chosen edge's CH entry was a third trip after the pick (see select_game, tree_device.hpp)
            const size_t eb = (gbase + node) * p.Ap;
            int L = p.NK[gbase + node];
            int n[NCH], w[NCH], chv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                const int ec = e < p.Ap ? e : p.Ap - 1;
                n[c] = p.N[eb + ec];
                w[c] = p.W[eb + ec];
                chv[c] = p.CH[eb + ec];
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(n[c]), "+v"(w[c]), "+v"(chv[c]));
            asm volatile("" : "+v"(L));
            int tot = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                if (e >= L) { n[c] = 0; w[c] = 0; }
                tot += n[c];
            }
            tot = wave_sum_i(tot);
            // PUCT: c_puct * p * np.sqrt(total_n) / (n + 1), p = 1 / len(actions)   (agents.py:361-365)
            // UCT:  np.inf if n == 0 else np.sqrt(2 * np.log(total_n) / n)          (agents.py:537-541)
            double common;
            if (p.mode == 0) {
                const double pr = __ddiv_rn(1.0, static_cast<double>(L));
                common = __dmul_rn(__dmul_rn(p.c_puct, pr), __dsqrt_rn(static_cast<double>(tot)));
            } else {
                common = __dmul_rn(2.0, p.log_lut[tot]);
            }
            double sc[NCH];
            double mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                sc[c] = -INFINITY;
                if (e < L) {
                    const double q = (n[c] > 0) ? __ddiv_rn(static_cast<double>(w[c]), static_cast<double>(n[c])) : 0.0;
                    double u;
                    if (p.mode == 0) u = __ddiv_rn(common, static_cast<double>(n[c] + 1));
                    else u = (n[c] == 0) ? INFINITY : __dsqrt_rn(__ddiv_rn(common, static_cast<double>(n[c])));
                    sc[c] = __dadd_rn(q, u);
                }
                mx = sc[c] > mx ? sc[c] : mx;
            }
            mx = wave_max_d(mx);
            uint64_t tm[NCH];
            int k = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                tm[c] = __ballot(e < L && sc[c] == mx);
                k += __popcll(tm[c]);
            }
            int r = mt.below(k);  // ids[np.random.choice(len(ids))]; a single maximum draws nothing
            int esel = -1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cnt = __popcll(tm[c]);
                if (esel < 0) {
                    if (r < cnt) esel = 64 * c + nth_set_bit(tm[c], r);
                    else r -= cnt;
                }
            }
            if (esel < 0) {  // NaN scores cannot occur (integer statistics); defensive
                if (lane == 0) atomicOr(&p.err[g], ERR_PATH);
                mt.close();
                return;
            }
            if (lane == 0) { s_pnode[depth] = node; s_pedge[depth] = static_cast<int16_t>(esel); }
            ++depth;
            const int cell = nth_empty<NCH>(cur, p.A, esel);  // children are the empty cells, ascending
            pos_place(cur, cell);
            int ch = CH_UNVISITED;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cv = read_lane(chv[c], esel & 63);
                if ((esel >> 6) == c) ch = cv;
            }
            if (ch >= 0) { node = ch; continue; }
            if (ch == CH_TERMINAL) { reward = 1; break; }
            const int win = win_after_move(cur, cell, p.B, p.win_mark);
            if (win != 0) {  // terminal leaf (draws included): reward 1, never expanded
                if (lane == 0) p.CH[eb + esel] = CH_TERMINAL;
                reward = 1;
                break;
            }
            const int leaf = new_node(p.A - cur.ply);
            if (lane == 0) p.CH[eb + esel] = leaf;
            // random playout from the leaf (agents.py:392-413)
            Pos sim = cur;
            int ws;
            for (;;) {
                const int ls = p.A - sim.ply;
                const int c2 = nth_empty<NCH>(sim, p.A, mt.below(ls));
                pos_place(sim, c2);
                ws = win_after_move(sim, c2, p.B, p.win_mark);
                if (ws != 0) break;
            }
            // utils.get_reward(win, leaf_id): turn = get_turn(leaf_id), 1 = white to move at the leaf
            const int turn = cur.ply & 1;
            if (ws == 1) reward = (turn == 1) ? 1 : -1;
            else if (ws == 2) reward = (turn == 1) ? -1 : 1;
            else reward = 0;
            break;
        }
        wsync();
        // backup: the edge into the leaf gets +reward, signs alternate towards the root
        for (int d = lane; d < depth; d += 64) {
            const int cnt = depth - 1 - d;
            const size_t idx = (gbase + s_pnode[d]) * p.Ap + s_pedge[d];
            p.N[idx] += 1;
            p.W[idx] += (cnt & 1) ? -reward : reward;
        }
        wsync();
    }

    // tail of get_pi: one-hot on the most visited child (PUCT) / the child with the largest q (UCT)
    const int L0 = p.NK[gbase];
    double st[NCH];
    double mx = -INFINITY;
    uint64_t em[NCH];  // empty cells of the root, per 64-cell chunk (ballots in uniform control flow)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        em[c] = __ballot(cell < p.A && !pos_occupied(root, cell));
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        st[c] = (p.mode == 0) ? 0.0 : -INFINITY;
        if (cell < p.A && !pos_occupied(root, cell)) {
            // edge index of this cell = number of empty cells below it
            int e = __popcll(em[c] & lanes_below());
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2)
                if (c2 < c) e += __popcll(em[c2]);
            if (e < L0) {
                const int nn = p.N[gbase * p.Ap + e];
                const int ww = p.W[gbase * p.Ap + e];
                if (p.mode == 0) st[c] = static_cast<double>(nn);
                else st[c] = (nn > 0) ? __ddiv_rn(static_cast<double>(ww), static_cast<double>(nn)) : 0.0;
            }
        }
        mx = st[c] > mx ? st[c] : mx;
    }
    mx = wave_max_d(mx);
    uint64_t tm[NCH];
    int k = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        tm[c] = __ballot(cell < p.A && st[c] == mx);
        k += __popcll(tm[c]);
    }
    int r = mt.below(k);
    int pick = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cnt = __popcll(tm[c]);
        if (pick < 0) {
            if (r < cnt) pick = 64 * c + nth_set_bit(tm[c], r);
            else r -= cnt;
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        if (cell < p.A) {
            p.out_stat[static_cast<size_t>(g) * p.A + cell] = st[c];
            p.out_pi[static_cast<size_t>(g) * p.A + cell] = (cell == pick) ? 1.0 : 0.0;
        }
    }
    if (lane == 0) p.action[g] = pick;
    mt.close();
}


// ----------------------------------------------------------------------------------------------
// 1_tictactoe_MCTS/mcts_vs.py (BASELINE configs[0]): the UCT search its __main__ runs per move
// (selection :15-46, expansion :48-93, simulation :95-111, backup :113-131, driver :153-183).
// Differences from the Omok agents above, all reproduced: only the root or a node with n > 10 is
// expanded, and then ONE of the new children is picked with random.sample; an unvisited child is
// scored with n = 0.0001; the first maximum wins (strict '>', no random tie-break); the playout
// result is scored from the ROOT player's side (draw 0.8, win +1, loss -1) and the same value is
// added at every node of the path; w is a float64 sum; the stream is Python's `random`
// (getrandbits(k) = word >> (32 - k), rejection on k = n.bit_length() bits). Nodes are a
// structure of arrays per game: parent / action / player / first child / child count / n / w.
// ----------------------------------------------------------------------------------------------
struct TttParams {
    int B, A, win_mark, G, cap, sims;
    int32_t* PAR; int32_t* KID; int32_t* N; double* W;   // [G][cap]
    int16_t* NK; uint8_t* ACT; uint8_t* PL;              // [G][cap]
    const int8_t* boards;                                // [G][A] +1 first player (O), -1 second (X)
    const int32_t* turns;                                // [G]
    uint32_t* mt; int32_t* mtpos;
    const double* log_lut;                               // ln(n), n = 0 .. sims + 1
    double* out_q; double* out_n; int32_t* action; int32_t* err;
    const uint8_t* active;
};

__device__ __forceinline__ int py_randbelow(MtDev& mt, int n) {  // random._randbelow_with_getrandbits
    const int k = 32 - __clz(n);                                // n.bit_length(), n >= 1
    uint32_t v;
    do { v = mt.next32() >> (32 - k); } while (static_cast<int>(v) >= n);
    return static_cast<int>(v);
}

template <int NCH>
__global__ __launch_bounds__(64) void k_ttt_search(TttParams p) {
    __shared__ uint32_t s_mt[624];
    __shared__ int32_t s_path[kMaxCells + 2];
    const int g = blockIdx.x;
    const int lane = lane_id();
    if (p.active && !p.active[g]) return;
    MtDev mt;
    mt.open(p.mt + static_cast<size_t>(g) * 624, p.mtpos + g, s_mt);
    const size_t nb = static_cast<size_t>(g) * p.cap;
    // root position: bb[0] = +1 stones, bb[1] = -1 stones
    Pos root;
    pos_clear(root);
    int stones0 = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        const int v = (cell < p.A) ? p.boards[static_cast<size_t>(g) * p.A + cell] : 0;
        const uint64_t mo = __ballot(v > 0), mx = __ballot(v < 0);
        root.bb[0][c] = mo;
        root.bb[1][c] = mx;
        stones0 += __popcll(mo) + __popcll(mx);
    }
    const int root_player = p.turns[g];
    if (lane == 0) {
        p.PAR[nb] = -1; p.KID[nb] = -1; p.NK[nb] = 0; p.N[nb] = 0; p.W[nb] = 0.0; p.ACT[nb] = 0xFF;
        p.PL[nb] = static_cast<uint8_t>(root_player);
    }
    wsync();
    int used = 1;
    auto place = [&](Pos& s, int cell, int player) { s.bb[player][cell >> 6] |= 1ull << (cell & 63); };

    for (int it = 0; it < p.sims; ++it) {
        // ---- selection: first maximum of q + u among the children, down to a childless node ----
        Pos cur = root;
        int node = 0, depth = 0, stones = stones0, win = 0;
        for (;;) {
            const int nk = p.NK[nb + node];
            if (nk == 0) break;
            const int first = p.KID[nb + node];
            const double x2 = __dmul_rn(2.0, p.log_lut[p.N[nb + node]]);  // 2 * np.log(total_n)
            double sc[NCH];
            double mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int i = lane + 64 * c;
                sc[c] = -INFINITY;
                if (i < nk) {
                    const int n = p.N[nb + first + i];
                    const double w = p.W[nb + first + i];
                    const double dn = (n == 0) ? 0.0001 : static_cast<double>(n);
                    const double q = __ddiv_rn(w, dn);
                    const double u = __dmul_rn(5.0, __dsqrt_rn(__ddiv_rn(x2, dn)));
                    sc[c] = __dadd_rn(q, u);
                }
                mx = sc[c] > mx ? sc[c] : mx;
            }
            mx = wave_max_d(mx);
            int pick = -1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int i = lane + 64 * c;
                const uint64_t m = __ballot(i < nk && sc[c] == mx && sc[c] > -100.0);
                if (pick < 0 && m) pick = 64 * c + __ffsll(static_cast<long long>(m)) - 1;
            }
            if (pick < 0) {  // the reference would spin forever here (needs q + u <= -100 or NaN)
                if (lane == 0) atomicOr(&p.err[g], ERR_PATH);
                mt.close();
                return;
            }
            const int kid = first + pick;
            const int cell = p.ACT[nb + kid];
            const int pl = p.PL[nb + node];
            place(cur, cell, pl);
            ++stones;
            win = win_after_move_by(cur, cell, p.B, p.win_mark, pl, stones);
            if (lane == 0) s_path[depth] = kid;
            ++depth;
            node = kid;
        }
        // ---- expansion: the root, or a node visited more than 10 times, unless terminal ----
        int child = node;
        int player = p.PL[nb + node];
        if (win == 0 && (node == 0 || p.N[nb + node] > 10)) {
            const int na = p.A - stones;
            if (used + na > p.cap) {
                if (lane == 0) atomicOr(&p.err[g], ERR_NODE_CAP);
                break;
            }
            const int first = used;
            used += na;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cell = lane + 64 * c;
                const bool empty = cell < p.A && !pos_occupied(cur, cell);
                const uint64_t m = __ballot(empty);
                int base = 0;
#pragma unroll
                for (int c2 = 0; c2 < NCH; ++c2) {
                    const int i2 = lane + 64 * c2;
                    const uint64_t m2 = __ballot(i2 < p.A && !pos_occupied(cur, i2));
                    if (c2 < c) base += __popcll(m2);
                }
                if (empty) {
                    const size_t k = nb + first + base + __popcll(m & lanes_below());
                    p.PAR[k] = node; p.KID[k] = -1; p.NK[k] = 0; p.N[k] = 0; p.W[k] = 0.0;
                    p.ACT[k] = static_cast<uint8_t>(cell);
                    p.PL[k] = static_cast<uint8_t>(player ^ 1);
                }
            }
            if (lane == 0) { p.KID[nb + node] = first; p.NK[nb + node] = static_cast<int16_t>(na); }
            const int j = py_randbelow(mt, na);                  // random.sample(childs, 1)[0]
            child = first + j;
            const int cell = nth_empty<NCH>(cur, p.A, j);
            place(cur, cell, player);
            ++stones;
            win = win_after_move_by(cur, cell, p.B, p.win_mark, player, stones);
            if (lane == 0) s_path[depth] = child;
            ++depth;
            player ^= 1;
        }
        // ---- simulation: random.choice(valid_actions) until the game ends ----
        while (win == 0) {
            const int cell = nth_empty<NCH>(cur, p.A, py_randbelow(mt, p.A - stones));
            place(cur, cell, player);
            ++stones;
            win = win_after_move_by(cur, cell, p.B, p.win_mark, player, stones);
            player ^= 1;
        }
        // ---- backup: one value, from the root player's side, at every node of the path ----
        const double value = (win == 3) ? 0.8 : ((win - 1 == root_player) ? 1.0 : -1.0);
        wsync();
        for (int d = lane; d < depth; d += 64) {
            const size_t k = nb + s_path[d];
            p.N[k] += 1;
            p.W[k] = __dadd_rn(p.W[k], value);
        }
        if (lane == 0) p.N[nb] += 1;
        wsync();
    }

    // ---- q_list / max_action (first maximum in child order) ----
    const int nk = p.NK[nb];
    const int first = p.KID[nb];
    double qv[NCH];
    double mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        if (cell < p.A) {
            p.out_q[static_cast<size_t>(g) * p.A + cell] = -INFINITY;
            p.out_n[static_cast<size_t>(g) * p.A + cell] = 0.0;
        }
    }
    wsync();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        qv[c] = -INFINITY;
        if (i < nk) {
            const int n = p.N[nb + first + i];
            const double w = p.W[nb + first + i];
            qv[c] = (n > 0) ? __ddiv_rn(w, static_cast<double>(n)) : 0.0;   // 'q': 0 until the first backup
            const int cell = p.ACT[nb + first + i];
            p.out_q[static_cast<size_t>(g) * p.A + cell] = qv[c];
            p.out_n[static_cast<size_t>(g) * p.A + cell] = static_cast<double>(n);
        }
        mx = qv[c] > mx ? qv[c] : mx;
    }
    mx = wave_max_d(mx);
    int pick = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        const uint64_t m = __ballot(i < nk && qv[c] == mx);
        if (pick < 0 && m) pick = 64 * c + __ffsll(static_cast<long long>(m)) - 1;
    }
    if (lane == 0) p.action[g] = (pick >= 0) ? p.ACT[nb + first + pick] : -1;
    mt.close();
}

}  // namespace ao

// ==============================================================================================
// host side
// ==============================================================================================
struct ao_rollout {
    ao_rollout_config cfg{};
    ao::RollParams p{};
    int G = 0, A = 0;
    std::vector<void*> allocs;
    uint8_t* d_active = nullptr;
    std::vector<int32_t> has_gauss;
    std::vector<double> gauss;
    hipStream_t stream = nullptr;
    std::string err;
    int fail(const std::string& m) { err = m; return 1; }
};

static thread_local std::string g_rollout_create_error;

#define RO_HIP(r, call)                                                                        \
    do {                                                                                       \
        hipError_t st_ = (call);                                                               \
        if (st_ != hipSuccess) return (r)->fail(std::string(#call) + ": " + hipGetErrorString(st_)); \
    } while (0)

template <typename T>
static int ro_alloc(ao_rollout* r, T** out, size_t count) {
    void* q = nullptr;
    hipError_t st = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16));
    if (st != hipSuccess) return r->fail(std::string("hipMalloc: ") + hipGetErrorString(st));
    r->allocs.push_back(q);
    *out = static_cast<T*>(q);
    return 0;
}

extern "C" {

int ao_rollout_create(const ao_rollout_config* cfg, ao_rollout** out) {
    *out = nullptr;
    ao_rollout_config c = *cfg;
    if (c.board < 3 || c.board > ao::kMaxBoard || c.sims < 1 || c.games < 1 || (c.mode != 0 && c.mode != 1)) {
        g_rollout_create_error = "ao_rollout_create: board 3..15, sims >= 1, games >= 1, mode 0 (PUCT) or 1 (UCT)";
        return 1;
    }
    if (c.win_mark <= 0) c.win_mark = (c.board == 3) ? 3 : 5;  // agents.py:270
    if (c.c_puct <= 0) c.c_puct = 5.0;                         // agents.py:271
    ao_rollout* r = new ao_rollout;
    r->cfg = c;
    r->G = c.games;
    r->A = c.board * c.board;
    ao::RollParams& p = r->p;
    p.B = c.board; p.A = r->A; p.Ap = (r->A + 15) & ~15; p.win_mark = c.win_mark; p.G = c.games;
    p.cap = c.sims + 2; p.sims = c.sims; p.mode = c.mode; p.c_puct = c.c_puct;
    const size_t edges = static_cast<size_t>(p.G) * p.cap * p.Ap;
    double* lut = nullptr;
    bool bad = hipSetDevice(c.device) != hipSuccess || hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess;
    bad = bad || ro_alloc(r, &p.N, edges) || ro_alloc(r, &p.W, edges) || ro_alloc(r, &p.CH, edges) ||
          ro_alloc(r, &p.NK, static_cast<size_t>(p.G) * p.cap) || ro_alloc(r, &p.rootpos, p.G) ||
          ro_alloc(r, &p.mt, static_cast<size_t>(p.G) * 624) || ro_alloc(r, &p.mtpos, p.G) ||
          ro_alloc(r, &lut, c.sims + 3) || ro_alloc(r, &p.out_pi, static_cast<size_t>(p.G) * r->A) ||
          ro_alloc(r, &p.out_stat, static_cast<size_t>(p.G) * r->A) || ro_alloc(r, &p.action, p.G) ||
          ro_alloc(r, &p.err, p.G) || ro_alloc(r, &r->d_active, p.G);
    if (!bad) {
        std::vector<double> h(static_cast<size_t>(c.sims) + 3);
        for (size_t i = 0; i < h.size(); ++i) h[i] = std::log(static_cast<double>(i));  // np.log(total_n); ln 0 = -inf
        bad = hipMemcpy(lut, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice) != hipSuccess;
        p.log_lut = lut;
    }
    if (bad) {
        g_rollout_create_error = "ao_rollout_create: " + (r->err.empty() ? std::string("HIP initialisation failed") : r->err);
        for (void* q : r->allocs) hipFree(q);
        if (r->stream) hipStreamDestroy(r->stream);
        delete r;
        return 1;
    }
    r->has_gauss.assign(p.G, 0);
    r->gauss.assign(p.G, 0.0);
    std::vector<uint32_t> mt(624);
    for (int g = 0; g < p.G; ++g) {  // np.random.seed(g) until the caller says otherwise
        ao::HostMT::seed(mt.data(), static_cast<uint32_t>(g));
        hipMemcpy(p.mt + static_cast<size_t>(g) * 624, mt.data(), sizeof(uint32_t) * 624, hipMemcpyHostToDevice);
        const int32_t pos = 624;
        hipMemcpy(p.mtpos + g, &pos, sizeof(int32_t), hipMemcpyHostToDevice);
    }
    *out = r;
    return 0;
}

void ao_rollout_destroy(ao_rollout* r) {
    if (!r) return;
    hipSetDevice(r->cfg.device);
    if (r->stream) { hipStreamSynchronize(r->stream); hipStreamDestroy(r->stream); }
    for (void* q : r->allocs) hipFree(q);
    delete r;
}

const char* ao_rollout_last_error(const ao_rollout* r) { return r ? r->err.c_str() : g_rollout_create_error.c_str(); }

int ao_rollout_set_rng_state(ao_rollout* r, int g, const uint32_t* mt, int32_t pos, int32_t has_gauss, double gauss) {
    if (g < 0 || g >= r->G) return r->fail("game index out of range");
    RO_HIP(r, hipSetDevice(r->cfg.device));
    RO_HIP(r, hipMemcpyAsync(r->p.mt + static_cast<size_t>(g) * 624, mt, sizeof(uint32_t) * 624, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemcpyAsync(r->p.mtpos + g, &pos, sizeof(int32_t), hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipStreamSynchronize(r->stream));
    r->has_gauss[g] = has_gauss;
    r->gauss[g] = gauss;
    return 0;
}

int ao_rollout_get_rng_state(ao_rollout* r, int g, uint32_t* mt, int32_t* pos, int32_t* has_gauss, double* gauss) {
    if (g < 0 || g >= r->G) return r->fail("game index out of range");
    RO_HIP(r, hipSetDevice(r->cfg.device));
    RO_HIP(r, hipMemcpyAsync(mt, r->p.mt + static_cast<size_t>(g) * 624, sizeof(uint32_t) * 624, hipMemcpyDeviceToHost, r->stream));
    RO_HIP(r, hipMemcpyAsync(pos, r->p.mtpos + g, sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
    RO_HIP(r, hipStreamSynchronize(r->stream));
    if (has_gauss) *has_gauss = r->has_gauss[g];
    if (gauss) *gauss = r->gauss[g];
    return 0;
}

int ao_rollout_seed(ao_rollout* r, int g, uint32_t seed) {
    std::vector<uint32_t> mt(624);
    ao::HostMT::seed(mt.data(), seed);
    return ao_rollout_set_rng_state(r, g, mt.data(), 624, 0, 0.0);
}

int ao_rollout_search(ao_rollout* r, const int32_t* moves, const int32_t* nmoves, const uint8_t* active, double* pi,
                      double* stat, int32_t* action) {
    RO_HIP(r, hipSetDevice(r->cfg.device));
    const int G = r->G, A = r->A, B = r->cfg.board;
    std::vector<ao::Pos> pos(static_cast<size_t>(G));
    std::vector<uint8_t> act(static_cast<size_t>(G), 1);
    for (int g = 0; g < G; ++g) {
        ao::Pos& s = pos[static_cast<size_t>(g)];
        std::memset(&s, 0, sizeof(s));
        for (int i = 0; i < ao::kLastMoves; ++i) s.last[i] = 0xFF;
        if (active && !active[g]) { act[static_cast<size_t>(g)] = 0; continue; }
        const int n = nmoves[g];
        if (n < 0 || n >= A) return r->fail("game " + std::to_string(g) + ": root id must leave at least one empty cell");
        int last_win = 0;
        for (int i = 0; i < n; ++i) {
            const int a = moves[static_cast<size_t>(g) * A + i];
            if (a < 0 || a >= A) return r->fail("game " + std::to_string(g) + ": action index out of range");
            const int colour = i & 1;  // black first (utils.get_board, utils.py:171-179)
            uint64_t& word = s.bb[colour][a >> 6];
            if (((s.bb[0][a >> 6] | s.bb[1][a >> 6]) >> (a & 63)) & 1ull)
                return r->fail("game " + std::to_string(g) + ": move onto an occupied cell");
            word |= 1ull << (a & 63);
            // five (or win_mark) in a row through the new stone ends the game: such a root cannot be searched
            const int row = a / B, col = a % B;
            const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 0, 1, -1};
            for (int d = 0; d < 4; ++d) {
                int run = 1;
                for (int sgn = -1; sgn <= 1; sgn += 2)
                    for (int k = 1; k < r->cfg.win_mark; ++k) {
                        const int rr = row + sgn * k * dr[d], cc = col + sgn * k * dc[d];
                        if (rr < 0 || rr >= B || cc < 0 || cc >= B) break;
                        const int cell = rr * B + cc;
                        if (!((s.bb[colour][cell >> 6] >> (cell & 63)) & 1ull)) break;
                        ++run;
                    }
                if (run >= r->cfg.win_mark) last_win = 1;
            }
        }
        if (last_win) return r->fail("game " + std::to_string(g) + ": the root position is already won");
        s.ply = static_cast<int16_t>(n);
    }
    RO_HIP(r, hipMemcpyAsync(r->p.rootpos, pos.data(), sizeof(ao::Pos) * G, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemcpyAsync(r->d_active, act.data(), G, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemsetAsync(r->p.err, 0, sizeof(int32_t) * G, r->stream));
    ao::RollParams p = r->p;
    p.active = r->d_active;
    const int nch = (A + 63) / 64;
    switch (nch) {
        case 1: hipLaunchKernelGGL(ao::k_rollout_search<1>, dim3(G), dim3(64), 0, r->stream, p); break;
        case 2: hipLaunchKernelGGL(ao::k_rollout_search<2>, dim3(G), dim3(64), 0, r->stream, p); break;
        case 3: hipLaunchKernelGGL(ao::k_rollout_search<3>, dim3(G), dim3(64), 0, r->stream, p); break;
        default: hipLaunchKernelGGL(ao::k_rollout_search<4>, dim3(G), dim3(64), 0, r->stream, p); break;
    }
    RO_HIP(r, hipGetLastError());
    std::vector<int32_t> herr(static_cast<size_t>(G));
    RO_HIP(r, hipMemcpyAsync(herr.data(), r->p.err, sizeof(int32_t) * G, hipMemcpyDeviceToHost, r->stream));
    if (pi) RO_HIP(r, hipMemcpyAsync(pi, r->p.out_pi, sizeof(double) * G * A, hipMemcpyDeviceToHost, r->stream));
    if (stat) RO_HIP(r, hipMemcpyAsync(stat, r->p.out_stat, sizeof(double) * G * A, hipMemcpyDeviceToHost, r->stream));
    if (action) RO_HIP(r, hipMemcpyAsync(action, r->p.action, sizeof(int32_t) * G, hipMemcpyDeviceToHost, r->stream));
    RO_HIP(r, hipStreamSynchronize(r->stream));
    for (int g = 0; g < G; ++g)
        if (herr[static_cast<size_t>(g)]) return r->fail("game " + std::to_string(g) + ": rollout search failed (error bits " + std::to_string(herr[static_cast<size_t>(g)]) + ")");
    return 0;
}


}  // extern "C"

// ---- 1_tictactoe_MCTS/mcts_vs.py ------------------------------------------------------------------
struct ao_ttt {
    ao_ttt_config cfg{};
    ao::TttParams p{};
    int G = 0, A = 0;
    std::vector<void*> allocs;
    int8_t* d_boards = nullptr; int32_t* d_turns = nullptr; uint8_t* d_active = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    int fail(const std::string& m) { err = m; return 1; }
};

static thread_local std::string g_ttt_create_error;

template <typename T>
static int tt_alloc(ao_ttt* r, T** out, size_t count) {
    void* q = nullptr;
    hipError_t st = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16));
    if (st != hipSuccess) return r->fail(std::string("hipMalloc: ") + hipGetErrorString(st));
    r->allocs.push_back(q);
    *out = static_cast<T*>(q);
    return 0;
}

// random.seed(int < 2**32): MT19937 init_by_array([seed]) (CPython Modules/_randommodule.c)
static void py_random_seed(uint32_t* mt, uint32_t seed) {
    mt[0] = 19650218u;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + static_cast<uint32_t>(i);
    int i = 1;
    for (int k = 624; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + seed;  // + key[0] + j, j is always 0
        if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    for (int k = 623; k; --k) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - static_cast<uint32_t>(i);
        if (++i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

extern "C" {

int ao_ttt_create(const ao_ttt_config* cfg, ao_ttt** out) {
    *out = nullptr;
    ao_ttt_config c = *cfg;
    if (c.board < 3 || c.board > ao::kMaxBoard || c.sims < 1 || c.games < 1) {
        g_ttt_create_error = "ao_ttt_create: board 3..15, sims >= 1, games >= 1";
        return 1;
    }
    if (c.win_mark <= 0) c.win_mark = (c.board == 3) ? 3 : 5;
    ao_ttt* r = new ao_ttt;
    r->cfg = c; r->G = c.games; r->A = c.board * c.board;
    ao::TttParams& p = r->p;
    p.B = c.board; p.A = r->A; p.win_mark = c.win_mark; p.G = c.games; p.sims = c.sims;
    p.cap = 1 + c.sims * r->A;   // one expansion of at most A children per iteration
    const size_t nodes = static_cast<size_t>(p.G) * p.cap;
    double* lut = nullptr;
    bool bad = hipSetDevice(c.device) != hipSuccess || hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess;
    bad = bad || tt_alloc(r, &p.PAR, nodes) || tt_alloc(r, &p.KID, nodes) || tt_alloc(r, &p.N, nodes) ||
          tt_alloc(r, &p.W, nodes) || tt_alloc(r, &p.NK, nodes) || tt_alloc(r, &p.ACT, nodes) || tt_alloc(r, &p.PL, nodes) ||
          tt_alloc(r, &p.mt, static_cast<size_t>(p.G) * 624) || tt_alloc(r, &p.mtpos, p.G) || tt_alloc(r, &lut, c.sims + 3) ||
          tt_alloc(r, &p.out_q, static_cast<size_t>(p.G) * r->A) || tt_alloc(r, &p.out_n, static_cast<size_t>(p.G) * r->A) ||
          tt_alloc(r, &p.action, p.G) || tt_alloc(r, &p.err, p.G) || tt_alloc(r, &r->d_boards, static_cast<size_t>(p.G) * r->A) ||
          tt_alloc(r, &r->d_turns, p.G) || tt_alloc(r, &r->d_active, p.G);
    if (!bad) {
        std::vector<double> h(static_cast<size_t>(c.sims) + 3);
        for (size_t i = 0; i < h.size(); ++i) h[i] = std::log(static_cast<double>(i));
        bad = hipMemcpy(lut, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice) != hipSuccess;
        p.log_lut = lut;
    }
    if (bad) {
        g_ttt_create_error = "ao_ttt_create: " + (r->err.empty() ? std::string("HIP initialisation failed") : r->err);
        for (void* q : r->allocs) hipFree(q);
        if (r->stream) hipStreamDestroy(r->stream);
        delete r;
        return 1;
    }
    p.boards = r->d_boards;
    p.turns = r->d_turns;
    *out = r;
    for (int g = 0; g < p.G; ++g) ao_ttt_seed(r, g, static_cast<uint32_t>(g));
    return 0;
}

void ao_ttt_destroy(ao_ttt* r) {
    if (!r) return;
    hipSetDevice(r->cfg.device);
    if (r->stream) { hipStreamSynchronize(r->stream); hipStreamDestroy(r->stream); }
    for (void* q : r->allocs) hipFree(q);
    delete r;
}

const char* ao_ttt_last_error(const ao_ttt* r) { return r ? r->err.c_str() : g_ttt_create_error.c_str(); }

int ao_ttt_set_rng_state(ao_ttt* r, int g, const uint32_t* mt, int32_t pos) {
    if (g < 0 || g >= r->G) return r->fail("game index out of range");
    RO_HIP(r, hipSetDevice(r->cfg.device));
    RO_HIP(r, hipMemcpyAsync(r->p.mt + static_cast<size_t>(g) * 624, mt, sizeof(uint32_t) * 624, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemcpyAsync(r->p.mtpos + g, &pos, sizeof(int32_t), hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipStreamSynchronize(r->stream));
    return 0;
}

int ao_ttt_get_rng_state(ao_ttt* r, int g, uint32_t* mt, int32_t* pos) {
    if (g < 0 || g >= r->G) return r->fail("game index out of range");
    RO_HIP(r, hipSetDevice(r->cfg.device));
    RO_HIP(r, hipMemcpyAsync(mt, r->p.mt + static_cast<size_t>(g) * 624, sizeof(uint32_t) * 624, hipMemcpyDeviceToHost, r->stream));
    RO_HIP(r, hipMemcpyAsync(pos, r->p.mtpos + g, sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
    RO_HIP(r, hipStreamSynchronize(r->stream));
    return 0;
}

int ao_ttt_seed(ao_ttt* r, int g, uint32_t seed) {
    std::vector<uint32_t> mt(624);
    py_random_seed(mt.data(), seed);
    return ao_ttt_set_rng_state(r, g, mt.data(), 624);
}

int ao_ttt_search(ao_ttt* r, const int8_t* boards, const int32_t* turns, const uint8_t* active, double* q, double* n,
                  int32_t* action) {
    RO_HIP(r, hipSetDevice(r->cfg.device));
    const int G = r->G, A = r->A;
    std::vector<uint8_t> act(static_cast<size_t>(G), 1);
    for (int g = 0; g < G; ++g) {
        if (active && !active[g]) { act[static_cast<size_t>(g)] = 0; continue; }
        if (turns[g] != 0 && turns[g] != 1) return r->fail("game " + std::to_string(g) + ": turn must be 0 or 1");
        int empty = 0;
        for (int c = 0; c < A; ++c) {
            const int v = boards[static_cast<size_t>(g) * A + c];
            if (v < -1 || v > 1) return r->fail("game " + std::to_string(g) + ": board cells must be -1, 0 or +1");
            empty += (v == 0);
        }
        if (empty == 0) return r->fail("game " + std::to_string(g) + ": the board is full");
    }
    RO_HIP(r, hipMemcpyAsync(r->d_boards, boards, static_cast<size_t>(G) * A, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemcpyAsync(r->d_turns, turns, sizeof(int32_t) * G, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemcpyAsync(r->d_active, act.data(), G, hipMemcpyHostToDevice, r->stream));
    RO_HIP(r, hipMemsetAsync(r->p.err, 0, sizeof(int32_t) * G, r->stream));
    ao::TttParams p = r->p;
    p.active = r->d_active;
    switch ((A + 63) / 64) {
        case 1: hipLaunchKernelGGL(ao::k_ttt_search<1>, dim3(G), dim3(64), 0, r->stream, p); break;
        case 2: hipLaunchKernelGGL(ao::k_ttt_search<2>, dim3(G), dim3(64), 0, r->stream, p); break;
        case 3: hipLaunchKernelGGL(ao::k_ttt_search<3>, dim3(G), dim3(64), 0, r->stream, p); break;
        default: hipLaunchKernelGGL(ao::k_ttt_search<4>, dim3(G), dim3(64), 0, r->stream, p); break;
    }
    RO_HIP(r, hipGetLastError());
    std::vector<int32_t> herr(static_cast<size_t>(G));
    RO_HIP(r, hipMemcpyAsync(herr.data(), r->p.err, sizeof(int32_t) * G, hipMemcpyDeviceToHost, r->stream));
    if (q) RO_HIP(r, hipMemcpyAsync(q, r->p.out_q, sizeof(double) * G * A, hipMemcpyDeviceToHost, r->stream));
    if (n) RO_HIP(r, hipMemcpyAsync(n, r->p.out_n, sizeof(double) * G * A, hipMemcpyDeviceToHost, r->stream));
    if (action) RO_HIP(r, hipMemcpyAsync(action, r->p.action, sizeof(int32_t) * G, hipMemcpyDeviceToHost, r->stream));
    RO_HIP(r, hipStreamSynchronize(r->stream));
    for (int g = 0; g < G; ++g)
        if (herr[static_cast<size_t>(g)]) return r->fail("game " + std::to_string(g) + ": search failed (error bits " + std::to_string(herr[static_cast<size_t>(g)]) + ")");
    return 0;
}

}  // extern "C"
