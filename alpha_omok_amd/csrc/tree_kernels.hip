// tree_kernels.hip -- the per-simulation and per-move kernels of the batched MCTS.
//
// One wavefront (64 lanes) owns one game: grid = G blocks of 64 threads. Every game runs the
// reference's strictly sequential search (one leaf per game per wave, no virtual loss), so the
// visit counts are bit-identical to agents.py given the same evaluator outputs and RNG stream.
//
// Reference map (paths relative to /root/reference/2_AlphaOmok/):
//   k_select          agents.py:134-168 (_selection) + utils.py:30-59 (check_win, here an
//                     incremental five-in-a-row test through the new stone, decided by one
//                     wave ballot) + utils.py:139-168 (get_state_pt -> evaluation batch)
//   k_expand_backup   agents.py:170-239 (_expansion_evaluation, _backup) + utils.py:22-27
//                     (legal_actions incl. CPython set order) + numpy pairwise fp64 sum
//   k_begin_move      agents.py:93-103 (re-noise of an inherited root)
//   k_end_move        agents.py:64-80 + utils.py:198-205 (argmax_onehot)
//   k_play            utils.py:189-195 (get_action) + env/env_small.py:154-176,196 (step)
//                     + re-rooting (tree reuse across moves, main.py:171)
//   k_walk            the `root_id in self.tree` lookup of agents.py:84 for an arbitrary id
//
// Arithmetic contract: fp64 PUCT evaluated left to right with no
// FMA contraction (explicit __dmul_rn/__ddiv_rn/__dadd_rn), sqrt(total_n) from a host-built
// table of correctly rounded values, fp32 w/q with correctly rounded add/divide.
#include "tree_device.hpp"

namespace ao {

template <int NCH>
__global__ __launch_bounds__(64) void k_select(TreeParams p) {
    __shared__ uint32_t s_mt[624];
    unsigned sit_n, sit_off;
    sit_window(p, sit_n, sit_off);
    select_game<NCH>(p, blockIdx.x, s_mt, nullptr, nullptr, TakeRowWave{p.live, p.row_cap}, true, sit_n, sit_off);
}

template <int NCH>
__global__ __launch_bounds__(64) void k_expand_backup(TreeParams p) {
    __shared__ uint8_t s_ord[256];
    __shared__ double s_prior[256];
    __shared__ int16_t s_tab[256];
    expand_backup_game<NCH>(p, blockIdx.x, s_ord, s_prior, s_tab);
}

// expansion + backup of simulation i, then selection + input planes of simulation i+1, in one
// launch: the fused search loop (ao_search) is select | net | expand_select | net | ... -- one
// launch fewer per simulation, which matters when a handful of games make every kernel a few
// microseconds long. The trailing selection idles by itself once a game has all its simulations.
// kGamesPerWG games per workgroup (one wave each, nothing shared between them): a quarter of the workgroups to
// dispatch at 4096 games.
constexpr int kGamesPerWG = 4;
template <int NCH>
__global__ __launch_bounds__(64 * kGamesPerWG) void k_expand_select(TreeParams p) {
    __shared__ uint32_t s_mt[kGamesPerWG][624];
    __shared__ uint8_t s_ord[kGamesPerWG][256];
    __shared__ double s_prior[kGamesPerWG][256];
    __shared__ int16_t s_tab[kGamesPerWG][256];
    __shared__ unsigned s_need[kGamesPerWG + 1];
    const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int slot = blockIdx.x * kGamesPerWG + w;
    const bool exists = slot < p.G;
    if (!exists && !p.live) return;
    // (over-subscribed searches: more games than the chip holds waves, so the last workgroups start when the first ones retire --
    // the games whose descents were deepest in the previous move are dispatched first, k_order)
    const int g = (exists && p.order) ? __builtin_amdgcn_readfirstlane(p.order[slot]) : slot;
    unsigned sit_n, sit_off;
    sit_window(p, sit_n, sit_off);
    AO_TT(0);
    GameHdr hdr;
    if (exists) expand_backup_game<NCH>(p, g, s_ord[w], s_prior[w], s_tab[w], &hdr);
    wsync();
    AO_TT(1);
    // (rows handed out per simulation: the waves of the workgroup meet once inside select_game, see TakeRowWG)
    select_game<NCH>(p, g, s_mt[w], nullptr, &hdr, TakeRowWG<kGamesPerWG>{p.live, p.row_cap, s_need}, exists, sit_n, sit_off);
    AO_TT(2);
}

// ----------------------------------------------------------------------------------------------
// k_begin_move: re-noise the children of an inherited, expanded root (agents.py:93-103)
// ----------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(64) void k_begin_move(TreeParams p) {
    const int g = blockIdx.x;
    const int lane = lane_id();
    if (p.active && !p.active[g]) return;
    if (lane == 0) { p.sims_done[g] = 0; p.leaf_status[g] = LS_IDLE; }   // (no leaf of an aborted move is left waiting for a row)
    const int node = p.root_node[g];
    if (!(p.gflags[g] & 1) || node < 0) return;
    const size_t slot = node_slot(p, p.cur[g], g, node);
    const int L = nodePos(p, slot)->nchild;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        if (i < L) {
            const double t1 = __dmul_rn(0.75, rowP(p, slot)[i]);
            const double t2 = __dmul_rn(0.25, p.noise_buf[static_cast<size_t>(g) * p.Ap + i]);
            rowP(p, slot)[i] = __dadd_rn(t1, t2);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// k_end_move: visit / policy / pi of get_pi (agents.py:64-80), argmax_onehot for tau == 0
// ----------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(64) void k_end_move(TreeParams p) {
    __shared__ uint32_t s_mt[624];
    __shared__ int s_vis[256];
    __shared__ double s_pol[256];
    const int g = blockIdx.x;
    const int lane = lane_id();
    if (p.active && !p.active[g]) return;
    // The reference's search always runs its num_mcts simulations (agents.py:105-132): a game that is short of them -- a caller that
    // ends the move early, an over-subscribed search that left its catch-up loop -- gets an error word instead of a pi, and its
    // stream is not touched.
    if (p.sims_done[g] != p.sims_target[g]) {
        if (lane == 0) atomicOr(&p.err[g], ERR_SHORT);
        return;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int a = lane + 64 * c;
        if (a < p.A) { s_vis[a] = 0; s_pol[a] = 0.0; }
    }
    __syncthreads();
    const int node = p.root_node[g];
    int tot = 0;
    if (node >= 0) {
        const size_t slot = node_slot(p, p.cur[g], g, node);
        const int L = nodePos(p, slot)->nchild;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = lane + 64 * c;
            if (i < L) {
                const int a = rowACT(p, slot)[i];
                const int n = rowN(p, slot)[i];
                s_vis[a] = n;
                s_pol[a] = rowP(p, slot)[i];
                tot += n;
            }
        }
    }
    tot = wave_sum_i(tot);
    __syncthreads();
    const double total = static_cast<double>(tot);
    double pi[NCH];
    int vmax = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int a = lane + 64 * c;
        pi[c] = 0.0;
        if (a < p.A) {
            pi[c] = __ddiv_rn(static_cast<double>(s_vis[a]), total);  // visit / visit.sum()
            vmax = s_vis[a] > vmax ? s_vis[a] : vmax;
            if (p.out_visit) p.out_visit[static_cast<size_t>(g) * p.A + a] = static_cast<double>(s_vis[a]);
            if (p.out_policy) p.out_policy[static_cast<size_t>(g) * p.A + a] = s_pol[a];
        }
    }
    const int tau = p.tau ? p.tau[g] : 1;
    if (tau == 0) {
        // utils.argmax_onehot: np.argwhere(pi == pi.max()) in ascending action order.
        // Equal pi <=> equal visit count (same divisor), so compare the exact integers.
        vmax = wave_max_i(vmax);
        MtDev mt;
        mt.open(p.mt + static_cast<size_t>(g) * 624, p.mtpos + g, s_mt);
        uint64_t tm[NCH];
        int k = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int a = lane + 64 * c;
            tm[c] = __ballot(a < p.A && s_vis[a] == vmax);
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
        mt.close();
#pragma unroll
        for (int c = 0; c < NCH; ++c) pi[c] = (lane + 64 * c == pick) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int a = lane + 64 * c;
        if (a < p.A) p.out_pi[static_cast<size_t>(g) * p.A + a] = pi[c];
    }
}

// ----------------------------------------------------------------------------------------------
// re-rooting: copy the subtree of the node k_play / k_walk chose (p.pending_root) into the other arena, breadth first, so the
// arena holds only what later searches can reach. Reads the old arena only; the BFS queue lives in LDS.
// One workgroup of kRerootWaves waves per game. With a network that has learned something the played child keeps most of the
// root's visits -- subtrees of one to two thousand nodes -- and a copy that takes one node per memory round trip (rounds 1 - 5:
// one wave per game inside k_play) lasted 12.5 ms per move of 4096 games (tools/time_move_phases.py --weights ...). Here wave w
// takes queue entry head + w: the record reads of kRerootWaves nodes are in flight together, the children's new indices come from
// a prefix over the waves' child counts, in queue order -- the SAME numbering as the one-node-at-a-time copy, entry for entry.
// ----------------------------------------------------------------------------------------------
constexpr int kRerootWaves = 8;

template <int NCH>
__global__ __launch_bounds__(64 * kRerootWaves) void k_reroot(TreeParams p, const int32_t* games) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    int32_t* s_cnt = reinterpret_cast<int32_t*>(s_dyn);        // [kRerootWaves] children each wave's node brings, [kRerootWaves] dropped
    int32_t* s_old = s_cnt + 2 * kRerootWaves;                 // [cap] old index of the node that becomes new index i
    const int g = games ? games[blockIdx.x] : blockIdx.x;
    const int pend = p.pending_root[g];
    if (pend == 0) return;                                     // (uniform over the workgroup)
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int oa = p.cur[g];
    const int na = oa ^ 1;
    if (threadIdx.x == 0) s_old[0] = pend - 1;
    if (lane == 0) s_cnt[kRerootWaves + w] = 0;
    __syncthreads();
    int tail = 1;
    int dropped = 0;
    for (int head = 0; head < tail;) {
        const int h = head + w;
        const bool have = h < tail;                            // (wave-uniform) this round takes queue entries [head, min(head + waves, tail))
        const int next_head = head + kRerootWaves < tail ? head + kRerootWaves : tail;
        size_t so = 0;
        PosR m;
        int ch[NCH], nn[NCH];
        uint8_t act[NCH];
        float ww[NCH], qq[NCH];
        double pp[NCH];
        bool valid[NCH];
        int cnt = 0;
        if (have) {
            so = node_slot(p, oa, g, s_old[h]);
            m = pos_load(nodePos(p, so));
            const int L = m.nchild;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                valid[c] = e < L;
                ch[c] = valid[c] ? rowCH(p, so)[e] : CH_UNVISITED;
                nn[c] = valid[c] ? rowN(p, so)[e] : 0;
                ww[c] = valid[c] ? rowW(p, so)[e] : 0.f;
                qq[c] = valid[c] ? rowQ(p, so)[e] : 0.f;
                pp[c] = valid[c] ? rowP(p, so)[e] : 0.0;
                act[c] = valid[c] ? rowACT(p, so)[e] : 0;
                cnt += __popcll(__ballot(valid[c] && ch[c] >= 0));
            }
        }
        if (lane == 0) s_cnt[w] = cnt;
        __syncthreads();
        int base = tail, total = 0;
#pragma unroll
        for (int k = 0; k < kRerootWaves; ++k) {
            const int ck = s_cnt[k];
            if (k < w) base += ck;
            total += ck;
        }
        if (have) {
            const size_t sn = node_slot(p, na, g, h);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                const bool ex = valid[c] && ch[c] >= 0;
                const uint64_t mk = __ballot(ex);
                const int idx = base + __popcll(mk & lanes_below());
                // A full arena (a long game whose visits keep following the played line): the breadth-first copy stops at
                // keep_max nodes, so the next move's expansions always fit. A child subtree that is not kept becomes an
                // unvisited child again (its statistics are forgotten); the event is counted in p.trimmed (ao_trim_stats).
                const bool keep = ex && idx < p.keep_max;
                const bool drop = ex && !keep;
                if (keep) s_old[idx] = ch[c];
                if (valid[c]) {
                    rowCH(p, sn)[e] = keep ? idx : (drop ? CH_UNVISITED : ch[c]);
                    rowN(p, sn)[e] = drop ? 0 : nn[c];
                    rowW(p, sn)[e] = drop ? 0.f : ww[c];
                    rowQ(p, sn)[e] = drop ? 0.f : qq[c];
                    rowP(p, sn)[e] = pp[c];
                    rowACT(p, sn)[e] = act[c];
                }
                dropped += __popcll(__ballot(drop));
                base += __popcll(mk);
            }
            if (lane == 0) pos_store(nodePos(p, sn), m);
        }
        // (indices handed out in queue order: everything below keep_max is kept, so the queue grows by what fits)
        const int room = p.keep_max - tail;
        tail += total < room ? total : (room > 0 ? room : 0);
        head = next_head;
        __syncthreads();
    }
    if (lane == 0 && dropped > 0) s_cnt[kRerootWaves + w] = dropped;
    __syncthreads();
    if (threadIdx.x == 0) {
        int d = 0;
        for (int k = 0; k < kRerootWaves; ++k) d += s_cnt[kRerootWaves + k];
        p.nodes_used[g] = tail;
        p.cur[g] = na;
        p.root_node[g] = 0;
        p.pending_root[g] = 0;
        if (d > 0) {
            p.trimmed[2 * g] += d;
            p.trimmed[2 * g + 1] += 1;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// k_play: utils.get_action on the game's stream, env step, re-root on the chosen child
// ----------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(64) void k_play(TreeParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t* s_mt = reinterpret_cast<uint32_t*>(s_dyn);             // 624 * 4 = 2496 B
    double* s_cdf = reinterpret_cast<double*>(s_dyn + 2496);         // 256 * 8 = 2048 B
    const int g = blockIdx.x;
    const int lane = lane_id();
    if (p.active && !p.active[g]) return;
    MtDev mt;
    mt.open(p.mt + static_cast<size_t>(g) * 624, p.mtpos + g, s_mt);
    // cdf = pi.cumsum() (sequential fp64); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
    if (lane == 0) {
        double s = 0.0;
        for (int a = 0; a < p.A; ++a) {
            const double x = p.out_pi[static_cast<size_t>(g) * p.A + a];
            s = (a == 0) ? x : __dadd_rn(s, x);
            s_cdf[a] = s;
        }
    }
    __syncthreads();
    const double last = s_cdf[p.A - 1];
    const double u = mt.next_double();
    mt.close();
    int action = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int a = lane + 64 * c;
        const bool gt = (a < p.A) && (__ddiv_rn(s_cdf[a], last) > u);
        const uint64_t mk = __ballot(gt);
        if (action < 0 && mk) action = 64 * c + __ffsll(static_cast<long long>(mk)) - 1;
    }
    if (action < 0) action = p.A - 1;  // pi == NaN (no visits): numpy would return A here
    // env.step: place the stone, flip the turn, check_win (env_small.py:154-176,196)
    PosR rp = pos_load(p.rootpos + g);
    const bool occupied = pos_occupied(rp, action);
    pos_place(rp, action);
    const int w = win_after_move(rp, action, p.B, p.win_mark);
    // tree reuse: the chosen child becomes the root (main.py:171 -> agents.py:84)
    const int node = p.root_node[g];
    int ch = CH_UNVISITED;
    if (node >= 0) {
        const size_t slot = node_slot(p, p.cur[g], g, node);
        const int L = nodePos(p, slot)->nchild;
        int found = -1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = lane + 64 * c;
            const uint64_t mk = __ballot(i < L && rowACT(p, slot)[i] == action);
            if (found < 0 && mk) found = 64 * c + __ffsll(static_cast<long long>(mk)) - 1;
        }
        if (found >= 0) ch = rowCH(p, slot)[found];
    }
    if (ch >= 0 && w == 0) {
        // The arena is compacted (k_reroot, launched right behind this kernel: the subtree copied into the other arena) only when
        // the next search might not fit behind what is in it: nodes_used <= keep_max = cap - sims - 1 leaves room for every
        // expansion of the next move, so the root just moves to the child and the unreachable nodes stay where they are until a
        // later move needs the room. Same trees, same trims (a subtree above keep_max nodes implies nodes_used above it), a
        // copy every (cap - subtree) / sims moves instead of every move.
        if (lane == 0) {
            if (p.nodes_used[g] <= p.keep_max && !p.compact_always) p.root_node[g] = ch;
            else p.pending_root[g] = ch + 1;
        }
    } else if (lane == 0) {
        p.nodes_used[g] = 0;
        p.root_node[g] = -1;
    }
    if (lane == 0) {
        rp.nchild = 0;
        pos_store(p.rootpos + g, rp);
        p.action[g] = action;
        p.win[g] = w;
        // the new root is a record of the reference's dict iff its parent was expanded
        p.rstatus[g] = (ch >= 0 && w == 0) ? 2 : ((node >= 0) ? 1 : 0);
        if (occupied) atomicOr(&p.err[g], ERR_BAD_MOVE);
    }
}

// ----------------------------------------------------------------------------------------------
// k_walk: move the roots of the listed games along their `extra` moves (ZeroAgent.get_pi with an
// arbitrary id that extends the previous root id). One workgroup per listed game: entry b walks game
// games[b] along extra[b * stride .. + m[b]). status: 0 fresh, 1 known but unexpanded, 2 expanded,
// -1 illegal move in the id.
// ----------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(64) void k_walk(TreeParams p, const int32_t* games, const int32_t* extra_all, int stride,
                                             const int32_t* m_all, const int32_t* prev_known_all, int32_t* status_out) {
    const int lane = lane_id();
    const int b = blockIdx.x;
    const int g = games[b];
    const int32_t* extra = extra_all + static_cast<size_t>(b) * stride;
    const int m = m_all[b];
    const int prev_known = prev_known_all[b];
    int node = p.root_node[g];
    PosR rp = pos_load(p.rootpos + g);
    // `known`: the current id is a key of the reference's dict
    bool known = (node >= 0) || prev_known;
    bool bad = false;
    for (int i = 0; i < m; ++i) {
        const int a = extra[i];
        if (a < 0 || a >= p.A || pos_occupied(rp, a)) bad = true;
        int next = -1;
        bool next_known = false;
        if (node >= 0) {
            const size_t slot = node_slot(p, p.cur[g], g, node);
            const int L = nodePos(p, slot)->nchild;
            int found = -1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                const uint64_t mk = __ballot(e < L && rowACT(p, slot)[e] == a);
                if (found < 0 && mk) found = 64 * c + __ffsll(static_cast<long long>(mk)) - 1;
            }
            if (found >= 0) {
                next_known = true;  // children of an expanded node are dict entries
                const int ch = rowCH(p, slot)[found];
                if (ch >= 0) next = ch;
            }
        }
        node = next;
        known = next_known;
        if (!bad) pos_place(rp, a);
    }
    int status = 0;
    if (node >= 0) {
        status = 2;
        if (m > 0 && lane == 0) p.pending_root[g] = node + 1;   // k_reroot (launch_walk) copies the subtree
    } else {
        status = known ? 1 : 0;
        if (lane == 0) { p.nodes_used[g] = 0; p.root_node[g] = -1; }
    }
    if (lane == 0) {
        rp.nchild = 0;
        pos_store(p.rootpos + g, rp);
        status_out[b] = bad ? -1 : status;
    }
}

// k_eval_log (ao_set_eval_log, a test / debugging hook): the policy row and the value the listed games' leaves were evaluated
// with in this simulation, copied out of the evaluation batch -- what lets the parity tests replay ao_search through the oracle
__global__ void k_eval_log(const int32_t* games, const int32_t* row_of_game, const float* policy, const float* value, int A, float* out,
                           const int32_t* sims_done, const int32_t* leaf_status, int what) {
    const int k = blockIdx.x;
    const int g = games[k];
    const int row = row_of_game[g];
    float* o = out + static_cast<size_t>(k) * (A + 3);
    if (what & 1) {
        for (int i = threadIdx.x; i < A; i += blockDim.x) o[i] = policy[static_cast<size_t>(row) * A + i];
        if (threadIdx.x == 0) o[A] = value[row];
    }
    if ((what & 2) && threadIdx.x == 0) {
        o[A + 1] = static_cast<float>(sims_done[g]);      // simulations the game has completed: this record belongs to the next one ...
        o[A + 2] = static_cast<float>(leaf_status[g]);    // ... if its leaf is waiting for this evaluation (LS_EXPAND / _ROOT) or is terminal
    }
}

// k_reset: ZeroAgent.reset() for the masked games
__global__ void k_reset(TreeParams p, const uint8_t* mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.G) return;
    if (mask && !mask[g]) return;
    PosR e;
    pos_clear(e);
    pos_store(p.rootpos + g, e);
    p.root_node[g] = -1;
    p.nodes_used[g] = 0;
    p.cur[g] = 0;
    p.gflags[g] = 0;
    p.sims_done[g] = 0;
    p.sims_target[g] = 0;
    p.err[g] = 0;
    p.leaf_status[g] = LS_IDLE;
    p.pending_root[g] = 0;
}

// ----------------------------------------------------------------------------------------------
// launchers (called from engine.hip)
// ----------------------------------------------------------------------------------------------
#define AO_DISPATCH_NCH(nch, ...)                    \
    switch (nch) {                                   \
        case 1: { constexpr int NCH = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int NCH = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NCH = 3; __VA_ARGS__; } break; \
        default: { constexpr int NCH = 4; __VA_ARGS__; } break; \
    }

static inline int nch_of(const TreeParams& p) { return (p.A + 63) / 64; }

void launch_select(const TreeParams& p, hipStream_t s) {
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_select<NCH>, dim3(p.G), dim3(64), 0, s, p));
}
void launch_expand_backup(const TreeParams& p, hipStream_t s) {
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_expand_backup<NCH>, dim3(p.G), dim3(64), 0, s, p));
}
void launch_expand_select(const TreeParams& p, hipStream_t s) {
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_expand_select<NCH>, dim3((p.G + kGamesPerWG - 1) / kGamesPerWG), dim3(64 * kGamesPerWG), 0, s, p));
#ifdef AO_PROF
    if (getenv("AO_PROF_TREE")) {
        static int count = 0;
        if (++count % 97 == 0) {
            unsigned long long h[16];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(ao_prof_tree), sizeof(h));
            fprintf(stderr, "AO_PROF k_expand_select (game 0) ticks: expansion+backup %llu, selection+planes %llu | last level: header %llu rows %llu puct %llu "
                    "tie+pick %llu child %llu | after loop: pos_store %llu planes %llu status+counters %llu mt.close %llu\n", h[1] - h[0], h[2] - h[1], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6],
                    h[8] - h[7], h[9] - h[8], h[10] - h[9], h[11] - h[10], h[2] - h[11]);
        }
    }
#endif
}
// ----------------------------------------------------------------------------------------------
// k_order: launch order of the next move's k_expand_select slots. An over-subscribed engine holds more games (5120) than the chip holds
// tree waves (4096 at 4 per SIMD): the last fifth of the workgroups starts when earlier ones retire, and a launch lasts as long as
// its deepest descent -- a deep game that starts late sets the launch's length (134 against 115 us for 4096 games).
// A game's depth is a property of its position (a forced line stays 40 - 55 levels deep for hundreds of simulations), so the levels it
// walked in the move that just ended (p.stats, about to be cleared) rank the next move's descents: 256 classes by share of the
// maximum, deepest first; the order inside a class is whatever the atomics give. Nothing a game computes depends on its slot.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_order(TreeParams p, int32_t* order) {
    constexpr int kClasses = 256;
    __shared__ unsigned s_max, s_cnt[kClasses], s_cur[kClasses];
    const int t = threadIdx.x;
    if (t == 0) s_max = 0;
    if (t < kClasses) s_cnt[t] = 0;
    __syncthreads();
    unsigned m = 0;
    for (int g = t; g < p.G; g += 1024) { const unsigned l = p.stats[static_cast<size_t>(g) * 4]; m = l > m ? l : m; }
    atomicMax(&s_max, m);
    __syncthreads();
    const unsigned long long mx = static_cast<unsigned long long>(s_max) + 1ull;
    auto cls = [&](int g) { return kClasses - 1 - static_cast<int>(p.stats[static_cast<size_t>(g) * 4] * static_cast<unsigned long long>(kClasses) / mx); };
    for (int g = t; g < p.G; g += 1024) atomicAdd(&s_cnt[cls(g)], 1u);
    __syncthreads();
    if (t == 0) { unsigned a = 0; for (int b = 0; b < kClasses; ++b) { s_cur[b] = a; a += s_cnt[b]; } }
    __syncthreads();
    for (int g = t; g < p.G; g += 1024) order[atomicAdd(&s_cur[cls(g)], 1u)] = g;
}
void launch_order(const TreeParams& p, int32_t* order, hipStream_t s) { hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, s, p, order); }

void launch_begin_move(const TreeParams& p, hipStream_t s) {
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_begin_move<NCH>, dim3(p.G), dim3(64), 0, s, p));
}
void launch_end_move(const TreeParams& p, hipStream_t s) {
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_end_move<NCH>, dim3(p.G), dim3(64), 0, s, p));
}
static void launch_reroot(const TreeParams& p, int count, const int32_t* games, hipStream_t s) {
    const size_t lds = (2 * kRerootWaves + static_cast<size_t>(p.cap)) * 4;
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_reroot<NCH>, dim3(count), dim3(64 * kRerootWaves), lds, s, p, games));
}
void launch_play(const TreeParams& p, hipStream_t s) {
    const size_t lds = 2496 + 2048;
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_play<NCH>, dim3(p.G), dim3(64), lds, s, p));
    launch_reroot(p, p.G, nullptr, s);
}
void launch_walk(const TreeParams& p, int count, const int32_t* games, const int32_t* extra, int stride, const int32_t* m,
                 const int32_t* prev_known, int32_t* status_out, hipStream_t s) {
    AO_DISPATCH_NCH(nch_of(p), hipLaunchKernelGGL(k_walk<NCH>, dim3(count), dim3(64), 0, s, p, games, extra, stride, m,
                                                   prev_known, status_out));
    launch_reroot(p, count, games, s);
}
void launch_eval_log(const int32_t* games, int n, const int32_t* row_of_game, const float* policy, const float* value, int A,
                     float* out, const int32_t* sims_done, const int32_t* leaf_status, int what, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_log, dim3(n), dim3(128), 0, s, games, row_of_game, policy, value, A, out, sims_done, leaf_status, what);
}
void launch_reset(const TreeParams& p, const uint8_t* mask, hipStream_t s) {
    hipLaunchKernelGGL(k_reset, dim3((p.G + 255) / 256), dim3(256), 0, s, p, mask);
}

}  // namespace ao
