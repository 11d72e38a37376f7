// engine_types.hpp -- POD types shared by the tree kernels, the network kernels and the host
// driver of libomok_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ao {

constexpr int kBBWords = 4;        // 4 x u64 = 256 bits >= 15*15 cells
constexpr int kMaxBoard = 15;
constexpr int kMaxCells = kMaxBoard * kMaxBoard;
constexpr int kLastMoves = 8;      // history kept for the input planes (C <= 9)
constexpr int kMaxPlanes = 9;      // inplanes = 2 * history + 1 <= 9 (main.py:34)
constexpr int kGroup = 32;         // boards per MFMA column group (32x32x2 f32 MFMA, N = boards)

// A board position. bb[0] = black stones, bb[1] = white stones, bit index = row*B + col
// (the reference's action index, utils.py:153-154). last[i] = move (ply - i), 0xFF if none.
struct Pos {
    uint64_t bb[2][kBBWords];
    int16_t ply;
    int16_t nchild;   // only meaningful inside a node record
    uint8_t last[kLastMoves];
    uint32_t pad_;
};
static_assert(sizeof(Pos) == 80, "Pos layout");

// child-link codes in the CH edge array
constexpr int32_t CH_UNVISITED = -1;  // reference: child record exists with n == 0
constexpr int32_t CH_TERMINAL = -2;   // visited child whose position ends the game

// leaf status written by the select kernel for the expand/backup kernel
// LS_WAIT / LS_WAIT_ROOT (round 5, rows handed out per simulation): the descent of this simulation is done -- path and leaf
// position are stored -- but every row of the evaluation batch was taken (TreeParams::row_cap); the game asks again in the
// next launch, before anyone else, and is expanded one launch later. Nothing observable changes: a game's search is strictly
// sequential either way.
// LS_DESCEND: the descent of this simulation used up the launch's level budget (TreeParams::max_levels) and goes on in the next launch
// from the node it reached (path so far stored; the node itself in the path slot behind it).
enum : int32_t { LS_IDLE = 0, LS_EXPAND = 1, LS_EXPAND_ROOT = 2, LS_TERMINAL = 3, LS_WAIT = 4, LS_WAIT_ROOT = 5, LS_DESCEND = 6 };

// per-game error bits
enum : int32_t { ERR_NODE_CAP = 1, ERR_PATH = 2, ERR_BAD_MOVE = 4, ERR_SHORT = 8 };   // ERR_SHORT: ao_end_move on a game short of its simulations

// Everything a tree kernel needs, passed by value.
//
// Tree arena in HBM: two arenas per game (ping-pong across re-rooting), each `cap` expanded nodes; a node owns Ap edge
// slots (A padded to 16), edge = position in the node's stored child order (the order of utils.legal_actions,
// agents.py:182,212). Node (arena, g, node) is record number node_slot() = (arena*G + g)*cap + node.
//
// ONE RECORD PER NODE (round 4; rounds 1-3 kept six separate arrays). A level of the PUCT descent reads a node's P, N, Q, CH,
// ACT rows and its position -- 2.4 KB -- and the next level depends on it; out of six multi-GB arrays that was seven pages
// and seven DRAM rows per level, and the descent of a trained network's tree (14 levels, the launch lasting as long as the
// deepest of 4096 descents) spent 4 - 5 k cycles per level waiting. tools/tree_layout_latency.hip replays the access
// pattern: 4.07 us per level on the six arrays, 2.11 us on interleaved records of the same footprint. Layout of a record
// (`rec` bytes, a multiple of 128; the rows the selection reads are contiguous at its start):
//   P   f64[Ap]  prior         (reference 'p', np.float64)                                   offset 0
//   N   i32[Ap]  visit count   (reference 'n', an integer-valued python float)               8 Ap
//   Q   f32[Ap]  mean value    (reference 'q', np.float32)                                   12 Ap
//   CH  i32[Ap]  expanded-child node index, CH_UNVISITED or CH_TERMINAL                      16 Ap
//   ACT u8[Ap]   action index of the edge                                                    20 Ap
//   W   f32[Ap]  total value   (reference 'w', np.float32 under numpy 2.x; backup only)      21 Ap
//   Pos          the node's position (80 B)                                                  25 Ap
struct TreeParams {
    int B, A, Ap, C, win_mark, G, cap, maxd, noise;
    int keep_max;  // most nodes a re-rooting keeps: cap - sims - 1, so the next move's expansions always fit
    int nchq;  // channel quads of the interleaved input batch: ceil(C/4) rounded up to even
    int nchq_live;  // quads the plane encoder writes: nchq, or ceil(C/4) when the padding quads are known to be zero
    double c_puct;
    // arena: one record of `rec` bytes per node (see above; row accessors below)
    unsigned char* arena; unsigned rec;
    // per game
    int32_t* cur;         // which arena is live
    int32_t* root_node;   // node index of the search root, -1 if the root is not expanded
    int32_t* nodes_used;
    Pos* rootpos;         // position of the root (= real game position)
    uint32_t* mt;         // [G][624] MT19937 state
    int32_t* mtpos;       // [G]
    double* noise_buf;    // [G][Ap] Dirichlet draw of this move (child order)
    int32_t* sims_target; int32_t* sims_done;
    int32_t* gflags;      // bit0: apply re-noise in begin_move (host-owned)
    int32_t* rstatus;     // root status after k_play: 0 fresh, 1 known/unexpanded, 2 expanded
    const int32_t* order; // [G] k_expand_select: slot -> game, the deepest descents of the previous move first (k_order; over-subscribed searches only) or nullptr
    int compact_always;   // developer switch (AO_COMPACT_ALWAYS): re-root by copying after every move, as rounds 1 - 5 did
    int32_t* pending_root;  // [G] 1 + the node (current arena) k_play / k_walk chose as the next root; k_reroot copies its subtree and clears it (0: nothing to do)
    // per simulation scratch
    int32_t* leaf_status; int32_t* path_len; int32_t* path_node; int16_t* path_edge; Pos* leaf_pos;
    int32_t* err;
    int32_t* trimmed;     // [G][2] cumulative: child subtrees dropped at re-rooting because the arena was full, re-rootings that dropped any
    unsigned* stats;      // [G][4] per game: levels, ties, terminal leaves, evaluated leaves
    const double* sqrt_lut; int sqrt_lut_n;
    // evaluation batch
    float* batch_il;      // interleaved [grp][cell][cq][il_group][4] (native network input) or null
    int il_group;         // boards per group of batch_il: 32 (layer kernels) or 16 (group-resident trunk)
    float* batch_nchw;    // [G][C][B][B] or null
    uint8_t* batch_u8;    // bit planes [G][u8_row] (byte per cell, bit q = plane q) or null: the split-fp16 network's input
    int u8_row;           // 128 (A <= 128) or 256
    int32_t* row_of_game; // row of game g in the native network's batch, or null = g (batch_nchw / external p, v). Two regimes:
                          // `live` null -- the host packed the ACTIVE games to the front once per move (the fused per-board step, the
                          // step-wise API); `live` set -- the selection hands out rows PER SIMULATION and writes this array:
    unsigned* live;       // rows taken so far by THIS launch's selections (zero before the launch), or null. A game whose new leaf
                          // needs the network takes the next row with one returning atomic; terminal leaves take none (the
                          // reference evaluates them and throws the result away, agents.py:171-178,216-221 -- SURVEY Q9), so the
                          // trunk sees live rows only: 11 - 19 % fewer with a trained network. The trunk kernels read the same word.
    // Over-subscription (more games than rows per simulation): a window of the game indices sits out every launch -- the games g
    // with (g + sit_off) mod G < sit_n start no descent -- and moves on by its own length, so every game sits out equally often.
    // The window's size follows the DEMAND: a controller on the device (tree_device.hpp, sit_window) reads the rows the previous
    // launch was asked for and steers towards row_target (a little below row_cap). ctl: [2][4] words, slot (launch & 1) holds
    // this launch's {sit_n, sit_off, sit as float bits}; null = nobody sits out.
    unsigned* ctl; int ctl_cur; const unsigned* live_prev; unsigned row_target;
    // A launch of the tree kernel lasts as long as its DEEPEST descent (trained network: ~55 levels against a mean of 14.5). With
    // max_levels > 0 (rows handed out per simulation only) a descent pauses after that many levels of one launch and resumes in
    // the next (LS_DESCEND); the game takes no row meanwhile. 0 = no budget.
    int max_levels;
    unsigned row_cap;     // rows one simulation may hand out (the evaluation batch the trunk is launched for); a game that finds
                          // them taken waits for the next launch (LS_WAIT): more games than rows = over-subscription
    const float* policy;  // [G][A]
    const float* value;   // [G]
    // move results
    double* out_pi; double* out_visit; double* out_policy;  // [G][A]
    const int8_t* tau;    // [G]
    int32_t* action; int32_t* win;  // [G]
    const uint8_t* active; // [G]
};

__host__ __device__ inline size_t node_slot(const TreeParams& p, int arena, int g, int node) {
    return (static_cast<size_t>(arena) * p.G + g) * p.cap + node;
}
__host__ __device__ inline unsigned node_rec_bytes(int Ap) { return (25u * Ap + 80u + 127u) & ~127u; }
__host__ __device__ inline unsigned char* node_rec(const TreeParams& p, size_t slot) { return p.arena + slot * p.rec; }
__host__ __device__ inline double* rowP(const TreeParams& p, size_t slot) { return reinterpret_cast<double*>(node_rec(p, slot)); }
__host__ __device__ inline int32_t* rowN(const TreeParams& p, size_t slot) { return reinterpret_cast<int32_t*>(node_rec(p, slot) + 8u * p.Ap); }
__host__ __device__ inline float* rowQ(const TreeParams& p, size_t slot) { return reinterpret_cast<float*>(node_rec(p, slot) + 12u * p.Ap); }
__host__ __device__ inline int32_t* rowCH(const TreeParams& p, size_t slot) { return reinterpret_cast<int32_t*>(node_rec(p, slot) + 16u * p.Ap); }
__host__ __device__ inline uint8_t* rowACT(const TreeParams& p, size_t slot) { return node_rec(p, slot) + 20u * p.Ap; }
__host__ __device__ inline float* rowW(const TreeParams& p, size_t slot) { return reinterpret_cast<float*>(node_rec(p, slot) + 21u * p.Ap); }
__host__ __device__ inline Pos* nodePos(const TreeParams& p, size_t slot) { return reinterpret_cast<Pos*>(node_rec(p, slot) + 25u * p.Ap); }

// parameters of both heads (model.py:34-73); BatchNorm folded into sc3/sh3
struct HeadParams {
    const float* w3;    // [3][planes]   1x1 convs: 2 policy channels + 1 value channel
    const float* sc3;   // [3]
    const float* sh3;   // [3]
    const float* wp_t;  // [2A][A]       policy_fc weight, transposed (input-major)
    const float* bp;    // [A]
    const float* w1_t;  // [A][planes]   value_fc1 weight, transposed
    const float* b1;    // [planes]
    const float* w2;    // [planes]      value_fc2
    const float* b2;    // [1]
};

// What the fused per-game step (k_step_board, step_kernels.hip: heads of the last leaf -> expansion + backup + selection ->
// conv1 of the new leaf, one workgroup per game) needs from the network of the per-board path.
struct StepNet {
    HeadParams heads;
    float* act;            // [rows][A][planes] fp32: the trunk's output (read by the heads) AND conv1's output (written)
    float* policy;         // [rows][A]
    float* value;          // [rows]
    const uint4* w1h;      // conv1 weights, split fp16, K = tap * 8 + plane padded to 96: [k step 3][cout tile][lane 64] x 8 halves
    const uint4* w1l;
    const float* sc1;      // [planes] BatchNorm scale / weight pre-scale
    const float* sh1;      // [planes]
    int planes;
    int C;
};

}  // namespace ao
