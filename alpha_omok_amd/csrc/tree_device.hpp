// tree_device.hpp -- device code of the per-simulation tree work: selection (agents.py:134-168),
// expansion + backup (agents.py:170-239), the per-game numpy-legacy MT19937 stream, the input
// plane encoder (utils.py:139-168), legal-move order (utils.py:22-27, CPython set order) and
// numpy's pairwise fp64 sum. One wavefront owns one game. Included by tree_kernels.hip (one wave
// per game, up to four games per workgroup) and by rollout.hip; see tree_kernels.hip for the
// reference map and the arithmetic contract.
#pragma once
#include "engine_types.hpp"

#ifdef AO_PROF
static __device__ unsigned long long ao_prof_tree[16];   // game 0 of k_expand_select: phase ends (shader-clock ticks)
#define AO_TT(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) ao_prof_tree[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AO_TT(k)
#endif
namespace ao {

// ----------------------------------------------------------------------------------------------
// wave helpers (wavefront = 64 lanes)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

// Ordering point between the lanes of ONE wavefront that exchange data through LDS. The per-game
// code is single-wave (LDS requests of a wave complete in order), so only the compiler has to be
// kept from reordering; a workgroup barrier here would deadlock when the waves of a workgroup run
// different games (k_expand_select: four games per workgroup, each with its own control flow).
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave reductions on the DPP path (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, row_bcast:15 and row_bcast:31 across them:
// six VALU instructions, the result in lane 63, handed back through v_readlane as a wave-uniform value). As xor butterflies over
// ds_bpermute they were 6 (int) / 12 (double) LDS-crossbar round trips with a wait each: ~1 k cycles per level of a descent
// (AO_PROF "puct"). Integer sums and maxima of non-NaN values do not depend on the order, so the results are the same bits.
// All 64 lanes must be active (they are: the per-game code is wave-uniform).
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_mov(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWS, 0xf, false); }
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppBcast15 = 0x142, kDppBcast31 = 0x143;

__device__ __forceinline__ int wave_sum_i(int v) {
    v += dpp_mov<kDppShr1, 0xf>(0, v);
    v += dpp_mov<kDppShr2, 0xf>(0, v);
    v += dpp_mov<kDppShr4, 0xf>(0, v);
    v += dpp_mov<kDppShr8, 0xf>(0, v);
    v += dpp_mov<kDppBcast15, 0xa>(0, v);
    v += dpp_mov<kDppBcast31, 0xc>(0, v);
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int wave_max_i(int v) {
    auto mx = [](int a, int b) { return b > a ? b : a; };
    v = mx(v, dpp_mov<kDppShr1, 0xf>(v, v));
    v = mx(v, dpp_mov<kDppShr2, 0xf>(v, v));
    v = mx(v, dpp_mov<kDppShr4, 0xf>(v, v));
    v = mx(v, dpp_mov<kDppShr8, 0xf>(v, v));
    v = mx(v, dpp_mov<kDppBcast15, 0xa>(v, v));
    v = mx(v, dpp_mov<kDppBcast31, 0xc>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_max_d(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double t = __hiloint2double(dpp_mov<CTRL, ROWS>(hi, hi), dpp_mov<CTRL, ROWS>(lo, lo));
    return t > v ? t : v;
}
__device__ __forceinline__ double wave_max_d(double v) {
    v = dpp_max_d<kDppShr1, 0xf>(v);
    v = dpp_max_d<kDppShr2, 0xf>(v);
    v = dpp_max_d<kDppShr4, 0xf>(v);
    v = dpp_max_d<kDppShr8, 0xf>(v);
    v = dpp_max_d<kDppBcast15, 0xa>(v);
    v = dpp_max_d<kDppBcast31, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// value of lane `l` (wave-uniform l) as a wave-uniform value: v_readlane, not an LDS-crossbar shuffle
__device__ __forceinline__ int read_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ uint64_t lanes_below() { return (1ull << lane_id()) - 1ull; }

// index of the r-th (0-based) set bit of m, m has more than r bits set
__device__ __forceinline__ int nth_set_bit(uint64_t m, int r) {
    const bool mine = ((m >> lane_id()) & 1ull) && (__popcll(m & lanes_below()) == r);
    const uint64_t hit = __ballot(mine);
    return __ffsll(static_cast<long long>(hit)) - 1;
}

// ----------------------------------------------------------------------------------------------
// positions
// ----------------------------------------------------------------------------------------------
// Word `w` of a bitboard WITHOUT a dynamically indexed access: `cell >> 6` differs from lane to lane, and a per-lane
// index into Pos::bb makes the compiler keep the whole 80-byte position in scratch (private memory: 20 dwords x 64
// lanes = 5 KB of global traffic per copy -- rocprof showed 16 KB written per game and simulation, five times the
// tree's own rows). The four words are passed BY VALUE and picked by selects: handed a pointer, instcombine folds
// "select of loads" back into a load through a selected address and the position is pinned in scratch again.
__device__ __forceinline__ uint64_t pick4(uint64_t a, uint64_t b, uint64_t c, uint64_t d, int w) {
    uint64_t v = a;
    v = (w == 1) ? b : v;
    v = (w == 2) ? c : v;
    v = (w == 3) ? d : v;
    return v;
}
template <int C, class PT>
__device__ __forceinline__ uint64_t pos_word(const PT& s, int w) {
    return pick4(s.bb[C][0], s.bb[C][1], s.bb[C][2], s.bb[C][3], w);
}
template <int C, class PT>
__device__ __forceinline__ bool pos_test(const PT& s, int cell) {   // colour C has a stone on `cell`
    return (pos_word<C>(s, cell >> 6) >> (cell & 63)) & 1ull;
}

__device__ __forceinline__ bool bb_test(const uint64_t* bb, int cell) {
    return (bb[cell >> 6] >> (cell & 63)) & 1ull;
}

// A position as the tree kernels hold it in REGISTERS: the same content as Pos (engine_types.hpp, the 80-byte record in
// HBM), with the move history as one 64-bit word and every member reachable by compile-time indices only, so nothing
// forces it into scratch. pos_load / pos_store move it with five 16-byte accesses.
struct PosR {
    uint64_t bb[2][kBBWords];
    uint64_t last64;   // byte i = move (ply - i), 0xFF if none
    int ply;
    int nchild;
    uint32_t pad_;
};

__device__ __forceinline__ PosR pos_load(const Pos* q) {
    const uint4* q4 = reinterpret_cast<const uint4*>(q);
    const uint4 a0 = q4[0], a1 = q4[1], a2 = q4[2], a3 = q4[3], a4 = q4[4];
    PosR r;
    r.bb[0][0] = a0.x | (static_cast<uint64_t>(a0.y) << 32); r.bb[0][1] = a0.z | (static_cast<uint64_t>(a0.w) << 32);
    r.bb[0][2] = a1.x | (static_cast<uint64_t>(a1.y) << 32); r.bb[0][3] = a1.z | (static_cast<uint64_t>(a1.w) << 32);
    r.bb[1][0] = a2.x | (static_cast<uint64_t>(a2.y) << 32); r.bb[1][1] = a2.z | (static_cast<uint64_t>(a2.w) << 32);
    r.bb[1][2] = a3.x | (static_cast<uint64_t>(a3.y) << 32); r.bb[1][3] = a3.z | (static_cast<uint64_t>(a3.w) << 32);
    r.ply = static_cast<int>(static_cast<int16_t>(a4.x & 0xFFFFu));
    r.nchild = static_cast<int>(static_cast<int16_t>(a4.x >> 16));
    r.last64 = a4.y | (static_cast<uint64_t>(a4.z) << 32);
    r.pad_ = a4.w;
    return r;
}

__device__ __forceinline__ void pos_store(Pos* q, const PosR& r) {
    uint4* q4 = reinterpret_cast<uint4*>(q);
    auto lo = [](uint64_t v) { return static_cast<unsigned>(v); };
    auto hi = [](uint64_t v) { return static_cast<unsigned>(v >> 32); };
    q4[0] = make_uint4(lo(r.bb[0][0]), hi(r.bb[0][0]), lo(r.bb[0][1]), hi(r.bb[0][1]));
    q4[1] = make_uint4(lo(r.bb[0][2]), hi(r.bb[0][2]), lo(r.bb[0][3]), hi(r.bb[0][3]));
    q4[2] = make_uint4(lo(r.bb[1][0]), hi(r.bb[1][0]), lo(r.bb[1][1]), hi(r.bb[1][1]));
    q4[3] = make_uint4(lo(r.bb[1][2]), hi(r.bb[1][2]), lo(r.bb[1][3]), hi(r.bb[1][3]));
    q4[4] = make_uint4((static_cast<unsigned>(r.ply) & 0xFFFFu) | (static_cast<unsigned>(r.nchild) << 16), lo(r.last64), hi(r.last64), r.pad_);
}

template <class PT>
__device__ __forceinline__ bool pos_occupied(const PT& s, int cell) {
    return ((pos_word<0>(s, cell >> 6) | pos_word<1>(s, cell >> 6)) >> (cell & 63)) & 1ull;
}

// env step / get_board: stone colour alternates, black first (utils.py:171-179)
__device__ __forceinline__ void pos_place(Pos& s, int cell) {
    const int colour = s.ply & 1;
    const int w = cell >> 6;
    const uint64_t bit = 1ull << (cell & 63);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < kBBWords; ++i) s.bb[c][i] |= (c == colour && i == w) ? bit : 0ull;
#pragma unroll
    for (int i = kLastMoves - 1; i > 0; --i) s.last[i] = s.last[i - 1];
    s.last[0] = static_cast<uint8_t>(cell);
    s.ply = static_cast<int16_t>(s.ply + 1);
}

__device__ __forceinline__ void pos_place(PosR& s, int cell) {
    const int colour = s.ply & 1;
    const int w = cell >> 6;
    const uint64_t bit = 1ull << (cell & 63);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < kBBWords; ++i) s.bb[c][i] |= (c == colour && i == w) ? bit : 0ull;
    s.last64 = (s.last64 << 8) | static_cast<uint64_t>(cell & 0xFF);
    s.ply += 1;
}

__device__ __forceinline__ void pos_clear(PosR& s) {
#pragma unroll
    for (int i = 0; i < kBBWords; ++i) { s.bb[0][i] = 0; s.bb[1][i] = 0; }
    s.ply = 0;
    s.nchild = 0;
    s.last64 = ~0ull;
    s.pad_ = 0;
}

__device__ __forceinline__ void pos_clear(Pos& s) {
#pragma unroll
    for (int i = 0; i < kBBWords; ++i) { s.bb[0][i] = 0; s.bb[1][i] = 0; }
    s.ply = 0;
    s.nchild = 0;
#pragma unroll
    for (int i = 0; i < kLastMoves; ++i) s.last[i] = 0xFF;
    s.pad_ = 0;
}

// Terminal test of the position reached by the stone just placed on `cell`
// (wave-uniform; all 64 lanes must call). Returns the reference's win_index:
// 0 playing, 1 black, 2 white, 3 draw (utils.py:30-59). Equivalent to the reference's window
// scan because the parent position was not terminal: only a line through the new stone can be
// new, and any run >= win_mark contains a window of exactly win_mark (overlines count).
// `mover` = colour index (0 / 1) of the stone just placed, `stones` = stones on the board now.
template <class PT>
__device__ __forceinline__ int win_after_move_by(const PT& s, int cell, int B, int win_mark, int mover, int stones) {
    const int lane = lane_id();
    // lane l < 32: direction l>>3, offset index l&7 -> offsets -4..-1, +1..+4
    const int dir = (lane >> 3) & 3;
    const int j = lane & 7;
    const int off = j < 4 ? j - 4 : j - 3;
    const int dr = (dir == 0) ? 0 : 1;
    const int dc = (dir == 0) ? 1 : (dir == 1) ? 0 : (dir == 2) ? 1 : -1;
    const int r = cell / B + off * dr;
    const int c = cell % B + off * dc;
    bool bit = false;
    if (lane < 32 && r >= 0 && r < B && c >= 0 && c < B) {
        const int cl = r * B + c;
        // (a mask blend, not `mover ? .. : ..`: the compiler turns that into a select of two POINTERS into the position,
        // which pins it in scratch)
        const uint64_t mk = 0ull - static_cast<uint64_t>(mover & 1);
        const uint64_t wd = (pos_word<0>(s, cl >> 6) & ~mk) | (pos_word<1>(s, cl >> 6) & mk);
        bit = (wd >> (cl & 63)) & 1ull;
    }
    const uint64_t m = __ballot(bit);
    bool won = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned m8 = static_cast<unsigned>((m >> (8 * d)) & 0xFFu);
        int run = 1;
        // offsets -1,-2,-3,-4 are bits 3,2,1,0 ; +1..+4 are bits 4..7
        for (int b = 3; b >= 0 && ((m8 >> b) & 1u); --b) ++run;
        for (int b = 4; b < 8 && ((m8 >> b) & 1u); ++b) ++run;
        won = won || (run >= win_mark);
    }
    if (won) return mover + 1;
    if (stones == B * B) return 3;
    return 0;
}

template <class PT>
__device__ __forceinline__ int win_after_move(const PT& s, int cell, int B, int win_mark) {
    return win_after_move_by(s, cell, B, win_mark, (s.ply - 1) & 1, s.ply);
}

// ----------------------------------------------------------------------------------------------
// per-game MT19937 stream on the device (numpy legacy RandomState)
// ----------------------------------------------------------------------------------------------
struct MtDev {
    uint32_t* g_mt;
    int32_t* g_pos;
    uint32_t* lds;  // [624]
    int pos;
    bool in_lds;
    // The next 64 words of the state, one per lane, requested when the stream is opened: a tie-break draw then is a v_readlane,
    // not a dependent memory round trip (every descent that ends at a freshly expanded node draws: its children all score 0,
    // agents.py:161-163 -- and the masked rejection takes 1.3 - 2 words per draw).
    uint32_t win;
    int win0;

    __device__ void load_window() {
        win0 = pos;
        const int i = pos + lane_id();
        win = g_mt[i < 624 ? i : 623];
    }
    __device__ void open(uint32_t* mt_row, int32_t* pos_ptr, uint32_t* lds_buf) {
        g_mt = mt_row;
        g_pos = pos_ptr;
        lds = lds_buf;
        pos = *pos_ptr;
        in_lds = false;
        load_window();
    }
    __device__ void open_at(uint32_t* mt_row, int32_t* pos_ptr, uint32_t* lds_buf, int known_pos) {   // the position is in a register already
        g_mt = mt_row;
        g_pos = pos_ptr;
        lds = lds_buf;
        pos = known_pos;
        in_lds = false;
        load_window();
    }

    // regenerate all 624 words (wave-parallel through LDS), write the state back to HBM
    __device__ void twist() {
        const int lane = lane_id();
        if (!in_lds) {
            for (int i = lane; i < 624; i += 64) lds[i] = g_mt[i];
            wsync();
        }
        constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
        for (int k0 = 0; k0 < 227; k0 += 64) {
            const int k = k0 + lane;
            uint32_t v = 0;
            if (k < 227) {
                const uint32_t y = (lds[k] & UP) | (lds[k + 1] & LO);
                v = lds[k + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
            }
            wsync();
            if (k < 227) lds[k] = v;
            wsync();
        }
        for (int k0 = 227; k0 < 623; k0 += 64) {
            const int k = k0 + lane;
            uint32_t v = 0;
            if (k < 623) {
                const uint32_t y = (lds[k] & UP) | (lds[k + 1] & LO);
                v = lds[k - 227] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
            }
            wsync();
            if (k < 623) lds[k] = v;
            wsync();
        }
        if (lane == 0) {
            const uint32_t y = (lds[623] & UP) | (lds[0] & LO);
            lds[623] = lds[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        wsync();
        for (int i = lane; i < 624; i += 64) g_mt[i] = lds[i];
        in_lds = true;
        pos = 0;
    }

    __device__ uint32_t next32() {  // wave-uniform
        if (pos >= 624) twist();
        uint32_t y;
        if (in_lds) y = lds[pos];
        else if (static_cast<unsigned>(pos - win0) < 64u) y = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), pos - win0));
        else y = g_mt[pos];
        ++pos;
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    // np.random.choice(k) == randint(0, k): masked rejection on 32-bit words; k == 1 draws nothing
    __device__ int below(int k) {
        if (k <= 1) return 0;
        uint32_t rng = static_cast<uint32_t>(k - 1);
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = next32() & mask; } while (v > rng);
        return static_cast<int>(v);
    }

    // np.random.random_sample(): 53-bit double from two words
    __device__ double next_double() {
        const uint32_t a = next32() >> 5;
        const uint32_t b = next32() >> 6;
        return __ddiv_rn(__dadd_rn(__dmul_rn(static_cast<double>(a), 67108864.0), static_cast<double>(b)),
                         9007199254740992.0);
    }

    __device__ void close() {
        if (lane_id() == 0) *g_pos = pos;
    }
};

// ----------------------------------------------------------------------------------------------
// get_state_pt (utils.py:139-168): planes [X_{k-C+2} .. X_k, colour], X_j = stones of the player
// who made move j as they stood after move j. Written into the evaluation batch.
// ----------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ void encode_planes(const TreeParams& p, int g, const PosR& s, int row = -1, uint8_t* lds_bits = nullptr) {
    const int lane = lane_id();
    // batch row of the native network (active games packed); select_game requests it with the game header -- read here it was a
    // dependent memory round trip at the very end of the kernel (AO_PROF: 1.7 k of the 5 k cycles after the descent)
    if (row < 0) row = p.row_of_game ? p.row_of_game[g] : g;
    const int k = s.ply;
    const int stm = k & 1;           // 0: black to move
    const int C = p.C;
    // the last moves as one 64-bit word: byte i = move (ply - i); extracted with shifts (an indexed
    // byte array would live in scratch memory)
    const uint64_t last64 = s.last64;
    // Plane q < C-1 is X_{k-j}, j = C-2-q: the stones of the player who made move k-j as they stood after it = that
    // player's stones now minus his later moves. Per cell that is: "holds a stone of that colour, and the stone is not one of
    // the last j moves" -- the cell's colour bits and its AGE (index of the cell in the move history, 255 if it is not
    // among the last eight moves) are found once, every plane is then two compares. (Built plane by plane as bitboards with
    // the later moves cleared this was ~1000 vector instructions: 4.7 k of the 23 k cycles of a single game's tree step,
    // AO_PROF; one wave alone issues an instruction every 4-5 cycles.)
    unsigned bits[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        const unsigned b0 = static_cast<unsigned>((s.bb[0][c] >> lane) & 1ull), b1 = static_cast<unsigned>((s.bb[1][c] >> lane) & 1ull);
        int age = 255;
#pragma unroll
        for (int i = kLastMoves - 1; i >= 0; --i)
            if (static_cast<int>((last64 >> (8 * i)) & 0xFF) == cell) age = i;
        unsigned bc = 0u;
#pragma unroll
        for (int q = 0; q < kMaxPlanes - 1; ++q) {
            const int j = C - 2 - q;
            if (q >= C - 1 || k - j < 1) continue;   // (uniform)
            const int col = (j & 1) ? stm : (stm ^ 1);   // mover of move k-j: the opponent of the side to move when j is even
            bc |= ((col ? b1 : b0) & static_cast<unsigned>(age >= j)) << q;
        }
        bits[c] = bc;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        if (cell >= p.A) continue;
        const unsigned bq = bits[c] | ((stm == 0) ? (1u << (C - 1)) : 0u);   // + the colour plane
        if (lds_bits) {   // the fused per-game step (k_step_board): the planes stay in the workgroup, as bits
            lds_bits[cell] = static_cast<uint8_t>(bq & ((1u << C) - 1u));
            continue;
        }
        auto plane = [&](int q) -> float { return (q < C && ((bq >> q) & 1u)) ? 1.f : 0.f; };
        if (p.batch_u8) p.batch_u8[static_cast<size_t>(row) * p.u8_row + cell] = static_cast<uint8_t>(bq & ((1u << C) - 1u));
        if (p.batch_nchw) {
            for (int q = 0; q < C; ++q) p.batch_nchw[(static_cast<size_t>(g) * C + q) * p.A + cell] = plane(q);
        }
        if (p.batch_il) {
            const size_t grp = static_cast<size_t>(row / p.il_group);
            const int b = row % p.il_group;
            for (int cq = 0; cq < p.nchq_live; ++cq) {
                const float4 v4 = make_float4(plane(4 * cq), plane(4 * cq + 1), plane(4 * cq + 2), plane(4 * cq + 3));
                float4* dst = reinterpret_cast<float4*>(p.batch_il) +
                              (((grp * p.A + cell) * p.nchq + cq) * p.il_group + b);
                *dst = v4;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// k_select
// ----------------------------------------------------------------------------------------------
// What select_game needs to know about its game before the first level. When expansion + backup of the previous simulation
// ran in the same wave just before (k_expand_select, k_step_board), that code has loaded or produced every one of these
// values and hands them over in registers: re-reading them was one more dependent memory round trip at the head of every
// descent (AO_PROF, one game: 5.6 k of the tree step's 23 k cycles).
struct GameHdr {
    int valid = 0;
    int is_active, done, target, arena, root_node, batch_row, mtpos;
    int prev_status;   // leaf status the expansion found: LS_WAIT / LS_WAIT_ROOT = the leaf of an earlier launch still needs its row
};

// Rows of the evaluation batch handed out per simulation (TreeParams::live). take(need) is called exactly ONCE by every wave of
// the kernel, at one place of select_game, and returns this wave's row, or -1 (none needed / the batch of this simulation is
// full). Returning atomics on ONE word serialise at ~11 ns each on the MI355X (tools/row_alloc_atomics.hip,
// profiles/r5a_row_alloc_atomics.txt: 3620 of them stretch a 3 us kernel to 44 us), so the waves of a workgroup meet at a barrier
// and ONE of them adds the workgroup's count (TakeRowWG: 15 us for 905 workgroups; 1.3 us on top of a 50 us kernel whose waves
// arrive spread out). TakeRowWave: one game per workgroup (k_select: once per move).
struct TakeRowWave {
    unsigned* live; unsigned cap;
    __device__ __forceinline__ int operator()(bool need) const {
        unsigned r = 0xffffffffu;
        if (need && lane_id() == 0) r = atomicAdd(live, 1u);
        r = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(r)));
        return r < cap ? static_cast<int>(r) : -1;
    }
};
template <int WAVES>
struct TakeRowWG {
    unsigned* live; unsigned cap;
    unsigned* s_need;   // [WAVES + 1] LDS: each wave's request, then the workgroup's first row
    __device__ __forceinline__ int operator()(bool need) const {
        const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
        if (lane_id() == 0) s_need[w] = need ? 1u : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) tot += s_need[i];
            s_need[WAVES] = tot ? atomicAdd(live, tot) : 0u;
        }
        __syncthreads();
        unsigned r = s_need[WAVES];
#pragma unroll
        for (int i = 0; i < WAVES; ++i) r += (i < w) ? s_need[i] : 0u;
        r = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(r)));
        return (need && r < cap) ? static_cast<int>(r) : -1;
    }
};
// The sit-out window of this launch (TreeParams::ctl) -- and, by ONE thread of the grid, the next launch's: an integrator on
// (rows the previous launch was asked for - row_target). The demand of launch i-1 answers the window of launch i-1 and sets the
// window of launch i+1: z^2 - z + gain * a = 0 with a = rows asked per game that does not sit out (~0.9) -- gain 0.3 settles in a
// handful of launches without ringing. When the games run out at the end of a move the demand falls and the window closes by itself.
__device__ __forceinline__ void sit_window(const TreeParams& p, unsigned& sit_n, unsigned& sit_off) {
    sit_n = sit_off = 0u;
    if (!p.ctl) return;
    const unsigned* cur = p.ctl + 4 * p.ctl_cur;
    sit_n = cur[0];
    sit_off = cur[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = __uint_as_float(cur[2]);
        if (p.live_prev) s += 0.3f * (static_cast<float>(*p.live_prev) - static_cast<float>(p.row_target));
        s = s < 0.f ? 0.f : s;
        s = s > static_cast<float>(p.G - 1) ? static_cast<float>(p.G - 1) : s;
        unsigned* nxt = p.ctl + 4 * (p.ctl_cur ^ 1);
        unsigned off = sit_off + sit_n;
        off = off >= static_cast<unsigned>(p.G) ? off - static_cast<unsigned>(p.G) : off;
        nxt[0] = static_cast<unsigned>(s + 0.5f);
        nxt[1] = off;
        nxt[2] = __float_as_uint(s);
    }
}

struct TakeRowNone {   // rows are not handed out by the kernel (TreeParams::live == nullptr)
    __device__ __forceinline__ int operator()(bool) const { return -1; }
};

// `exists` false: a wave of the workgroup beyond the last game -- it only keeps the row hand-out's barrier company.
template <int NCH, class TakeRow = TakeRowNone>
__device__ __forceinline__ void select_game(const TreeParams& p, const int g, uint32_t* s_mt /*[624]*/, uint8_t* lds_bits = nullptr,
                                            const GameHdr* hdr = nullptr, const TakeRow& take = TakeRow(), const bool exists = true,
                                            const unsigned sit_n = 0u, const unsigned sit_off = 0u) {
    const int lane = lane_id();
    AO_TT(3);
    if (!exists) { if (p.live) (void)take(false); return; }
    // The descent is a chain of dependent memory round trips (one wave per game has nothing else to
    // overlap them with), so every step requests all it can in ONE trip: first the game header ...
    const bool have = hdr != nullptr && hdr->valid;
    const int is_active = have ? hdr->is_active : (p.active ? p.active[g] : 1);
    const int done = have ? hdr->done : p.sims_done[g], target = have ? hdr->target : p.sims_target[g];
    const int arena = have ? hdr->arena : p.cur[g];
    int node = have ? hdr->root_node : p.root_node[g];
    int batch_row = have ? hdr->batch_row : (p.row_of_game ? p.row_of_game[g] : g);
    const int prev = !p.live ? LS_IDLE : have ? hdr->prev_status : p.leaf_status[g];
    // what this launch does for the game: 0 nothing, 1 its waiting leaf gets a row, 2 a descent
    int mode = 2;
    if (!is_active || done >= target) mode = 0;
    else if (prev == LS_WAIT || prev == LS_WAIT_ROOT) mode = 1;
    else if (prev == LS_DESCEND) mode = 4;   // a descent paused by the level budget goes on (it does not sit out: the simulation is under way)
    else if (sit_n > 0) {
        // over-subscribed (more games than rows per simulation): this launch's share of the games sits out -- a window of
        // game indices that moves on by its own length with every launch, so every game sits out equally often and all of them
        // reach their simulation count within a launch or two of each other. (Who finds the batch full is NOT left to the order
        // of arrival: the deepest descents arrive last, every time.)
        unsigned k = static_cast<unsigned>(g) + sit_off;
        k = k >= static_cast<unsigned>(p.G) ? k - static_cast<unsigned>(p.G) : k;
        if (k < sit_n) mode = 3;
    }
    if (mode == 0 || mode == 3) {
        if (lane == 0) p.leaf_status[g] = LS_IDLE;
        if (p.live) (void)take(false);
        return;
    }
    if (mode == 1) {
        // A leaf that found the batch full in an earlier launch: no second descent, no draw from the stream. It takes its row
        // with a wave's own atomic, at once -- before the games that expand and descend first get anywhere near theirs.
        const int r = TakeRowWave{p.live, p.row_cap}(true);
        if (r >= 0) {
            const PosR wl = pos_load(p.leaf_pos + g);
            encode_planes<NCH>(p, g, wl, r, lds_bits);
            if (lane == 0) {
                p.row_of_game[g] = r;
                p.leaf_status[g] = prev == LS_WAIT ? LS_EXPAND : LS_EXPAND_ROOT;
                atomicAdd(p.stats + static_cast<size_t>(g) * 4 + 3, 1u);
            }
        }
        (void)take(false);   // (mode 1 exists with p.live only)
        return;
    }
    MtDev mt;
    if (have) mt.open_at(p.mt + static_cast<size_t>(g) * 624, p.mtpos + g, s_mt, hdr->mtpos);
    else mt.open(p.mt + static_cast<size_t>(g) * 624, p.mtpos + g, s_mt);

    int depth = 0;
    int status = LS_EXPAND_ROOT;
    bool failed = false;
    unsigned levels = 0, ties = 0;
    int path_n[NCH], path_e[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) path_n[c] = path_e[c] = 0;
    PosR lp;
    bool paused = false;
    if (mode == 4) {   // resume: the path so far back into the registers, the node reached from the slot behind it
        depth = p.path_len[g];
        const size_t pb = static_cast<size_t>(g) * p.maxd;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = lane + 64 * c;
            const int dc = d < p.maxd ? d : p.maxd - 1;
            path_n[c] = p.path_node[pb + dc];
            path_e[c] = p.path_edge[pb + dc];
        }
        node = p.path_node[pb + depth];
    }
    if (node < 0) {
        lp = pos_load(p.rootpos + g);
    } else {
        for (;;) {
            AO_TT(4);
            const size_t slot = node_slot(p, arena, g, node);
            const double* rP = rowP(p, slot);
            const int32_t* rN = rowN(p, slot);
            const float* rQ = rowQ(p, slot);
            int32_t* rCH = rowCH(p, slot);
            const uint8_t* rACT = rowACT(p, slot);
            // ... then, per level, the node record together with all five edge rows. The rows are Ap
            // wide, so the addresses do not depend on the child count; lanes past it are masked after
            // the loads (CH and ACT of the chosen edge then come from a lane shuffle, not from memory).
            // The loads are UNCONDITIONAL (lanes past Ap re-read edge Ap-1; everything below masks by e < L <= Ap): written
            // as `in ? row[e] : 0` every load sat in its own predicated block, and the compiler put a wait behind the position
            // and behind each byte load -- three to four dependent memory round trips per level instead of one (ISA;
            // tools/tree_level_latency.hip: a lone wave fetches these rows in 1.1 us, the kernel's levels took ~3).
            int n[NCH], chv[NCH];
            unsigned acv[NCH];
            float qv[NCH];
            double pv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                const int ec = e < p.Ap ? e : p.Ap - 1;
                pv[c] = rP[ec];
                n[c] = rN[ec];
                qv[c] = rQ[ec];
                chv[c] = rCH[ec];
                acv[c] = rACT[ec];
            }
            PosR m = pos_load(nodePos(p, slot));
            // (pinned: the compiler otherwise sinks the loads whose first use sits under `e < L` into that block -- a second trip --
            // and leaves the position's words, used only when the descent ends, to be waited for at the top of the next level)
#pragma unroll
            for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(pv[c]), "+v"(n[c]), "+v"(qv[c]), "+v"(chv[c]), "+v"(acv[c]));
#pragma unroll
            for (int w = 0; w < kBBWords; ++w) asm volatile("" : "+v"(m.bb[0][w]), "+v"(m.bb[1][w]));
            asm volatile("" : "+v"(mt.win), "+v"(m.last64), "+v"(m.ply), "+v"(m.nchild));
            const int L = m.nchild;
            int tot = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                n[c] = (e < L) ? n[c] : 0;
                tot += n[c];
            }
            tot = wave_sum_i(tot);
            AO_TT(5);
            // np.sqrt(total_n) (total_n is an exact integer): the device's correctly rounded double sqrt equals the host's for
            // every integer below 2^24 (tools/sqrt_exact.hip, checked exhaustively on the MI355X) -- computed, not looked up: the
            // table read depended on total_n and was one more memory round trip per level of the descent
            const double sq = __dsqrt_rn(static_cast<double>(tot));
            double sc[NCH];
            double mx = -1.0e300;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                sc[c] = -1.0e300;
                if (e < L) {
                    const double q = static_cast<double>(qv[c]);
                    // u = c_puct * p * sqrt(total_n) / (n + 1)   (agents.py:158, left to right)
                    double t = __dmul_rn(p.c_puct, pv[c]);
                    t = __dmul_rn(t, sq);
                    const double u = __ddiv_rn(t, static_cast<double>(n[c] + 1));
                    sc[c] = __dadd_rn(q, u);
                }
                mx = sc[c] > mx ? sc[c] : mx;
            }
            mx = wave_max_d(mx);
            AO_TT(6);
            // every exact-equal maximum, in child order; uniform pick (agents.py:161-163)
            uint64_t tm[NCH];
            int k = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int e = lane + 64 * c;
                tm[c] = __ballot(e < L && sc[c] == mx);
                k += __popcll(tm[c]);
            }
            int r = 0;
            if (k > 1) { r = mt.below(k); ++ties; }
            int esel = -1;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cnt = __popcll(tm[c]);
                if (esel < 0) {
                    if (r < cnt) esel = 64 * c + nth_set_bit(tm[c], r);
                    else r -= cnt;
                }
            }
            if (esel < 0 || depth >= p.maxd || depth >= 64 * NCH) {  // NaN priors (policy summed to 0) or an inconsistent tree (the path registers hold 64 * NCH entries)
                failed = true;
                break;
            }
            // the path stays in registers (entry d in lane d & 63 of chunk d >> 6; a consistent tree is at most A <= 64 * NCH levels deep,
            // anything deeper was refused above) and is written once
            // after the descent: two stores per level kept the next level's loads behind their acknowledgement
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (lane + 64 * c == depth) { path_n[c] = node; path_e[c] = esel; }
            }
            ++depth;
            ++levels;
            AO_TT(7);
            int ch = CH_UNVISITED, a = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cv = read_lane(chv[c], esel & 63), av = read_lane(static_cast<int>(acv[c]), esel & 63);
                if ((esel >> 6) == c) { ch = cv; a = av; }
            }
            if (ch >= 0) {
                node = ch;
                if (p.max_levels > 0 && static_cast<int>(levels) >= p.max_levels) { paused = true; break; }   // (level budget of this launch)
                continue;
            }
            if (ch == CH_TERMINAL) { status = LS_TERMINAL; break; }
            // first visit of this child: build its position, test for the end of the game
            lp = m;
            pos_place(lp, a);
            const int w = win_after_move(lp, a, p.B, p.win_mark);
            if (w != 0) {
                status = LS_TERMINAL;
                if (lane == 0) rCH[esel] = CH_TERMINAL;
            } else {
                status = LS_EXPAND;
            }
            break;
        }
    }
    AO_TT(8);
    const bool to_expand = !failed && !paused && (status == LS_EXPAND || status == LS_EXPAND_ROOT);
    bool wait = false;
    if (p.live) {   // the leaf's row in this simulation's batch (every wave of the workgroup gets here or to one of the take(false) above)
        batch_row = take(to_expand);
        wait = to_expand && batch_row < 0;
    }
    if (failed) {
        if (lane == 0) { atomicOr(&p.err[g], ERR_PATH); p.leaf_status[g] = LS_IDLE; }
        mt.close();
        return;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int d = lane + 64 * c;
        if (d < depth) {
            p.path_node[static_cast<size_t>(g) * p.maxd + d] = path_n[c];
            p.path_edge[static_cast<size_t>(g) * p.maxd + d] = static_cast<int16_t>(path_e[c]);
        }
    }
    if (paused) {
        if (lane == 0) {
            p.path_node[static_cast<size_t>(g) * p.maxd + depth] = node;
            p.leaf_status[g] = LS_DESCEND;
            p.path_len[g] = depth;
            unsigned* st = p.stats + static_cast<size_t>(g) * 4;
            atomicAdd(st + 0, levels);
            atomicAdd(st + 1, ties);
        }
        mt.close();
        return;
    }
    if (to_expand) {
        lp.nchild = 0;
        if (lane == 0) pos_store(p.leaf_pos + g, lp);
        AO_TT(9);
        if (!wait) encode_planes<NCH>(p, g, lp, batch_row, lds_bits);
    }
    AO_TT(10);
    if (lane == 0) {
        p.leaf_status[g] = wait ? (status == LS_EXPAND ? LS_WAIT : LS_WAIT_ROOT) : status;
        p.path_len[g] = depth;
        if (p.live && to_expand && !wait) p.row_of_game[g] = batch_row;
        // per-game counters (a shared word would serialise 4 x G atomics per simulation); no-return atomics: as plain
        // read-modify-writes they were one more dependent round trip before the kernel could end
        unsigned* st = p.stats + static_cast<size_t>(g) * 4;
        atomicAdd(st + 0, levels);
        atomicAdd(st + 1, ties);
        if (!wait) atomicAdd(st + ((status == LS_TERMINAL) ? 2 : 3), 1u);
    }
    AO_TT(11);
    mt.close();
}

// ----------------------------------------------------------------------------------------------
// utils.legal_actions order (utils.py:22-27) = iteration order of a CPython 3.10 set difference.
// Ascending unless the result set's final hash table is smaller than its largest key
// (SURVEY.md Q5); that case is emulated on lane 0 (set_add_entry / set_table_resize).
// Writes the ordered actions to s_ord, returns their count (wave-uniform).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int set_probe(const int16_t* tab, int mask, int key) {
    unsigned perturb = static_cast<unsigned>(key);
    unsigned i = static_cast<unsigned>(key) & static_cast<unsigned>(mask);
    for (;;) {
        int probes = (i + 9u <= static_cast<unsigned>(mask)) ? 9 : 0;  // LINEAR_PROBES
        unsigned e = i;
        do {
            if (tab[e] < 0) return static_cast<int>(e);
            ++e;
        } while (probes--);
        perturb >>= 5;  // PERTURB_SHIFT
        i = (i * 5u + 1u + perturb) & static_cast<unsigned>(mask);
    }
}

template <int NCH>
__device__ int legal_order(const PosR& s, int A, uint8_t* s_ord, int16_t* s_tab /*[2][128]*/) {
    const int lane = lane_id();
    int L = 0;
    int maxkey = -1;
    bool legal[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        legal[c] = (cell < A) && !pos_occupied(s, cell);
        const uint64_t m = __ballot(legal[c]);
        if (legal[c]) s_ord[L + __popcll(m & lanes_below())] = static_cast<uint8_t>(cell);
        if (m) maxkey = 64 * c + 63 - __clzll(static_cast<long long>(m));
        L += __popcll(m);
    }
    wsync();
    const int nstones = s.ply;
    if ((A >> 2) > nstones) return L;  // set_copy_and_difference: ascending
    const int tsize = (L <= 4) ? 8 : (L <= 18) ? 32 : (L <= 76) ? 128 : 512;
    if (maxkey < tsize) return L;      // every key sits in its own slot: ascending
    if (lane == 0) {
        int16_t* tab = s_tab;
        int16_t* alt = s_tab + 128;
        int mask = 7, fill = 0;
        for (int i = 0; i < 8; ++i) tab[i] = -1;
        for (int idx = 0; idx < L; ++idx) {  // `so` iterates ascending; s_ord holds that order
            const int key = s_ord[idx];
            tab[set_probe(tab, mask, key)] = static_cast<int16_t>(key);
            ++fill;
            if (fill * 5 >= mask * 3) {
                const int minused = fill * 4;
                int ns = 8;
                while (ns <= minused) ns <<= 1;
                for (int i = 0; i < ns; ++i) alt[i] = -1;
                for (int i = 0; i <= mask; ++i)
                    if (tab[i] >= 0) alt[set_probe(alt, ns - 1, tab[i])] = tab[i];
                int16_t* t = tab; tab = alt; alt = t;
                mask = ns - 1;
            }
        }
        int cnt = 0;
        for (int i = 0; i <= mask; ++i)
            if (tab[i] >= 0) s_ord[cnt++] = static_cast<uint8_t>(tab[i]);
    }
    wsync();
    return L;
}

// numpy pairwise fp64 sum of a length-n vector (8 <= n <= 256), DOUBLE_pairwise_sum.
// All 64 lanes call; lanes 0-7 reduce the left block, 8-15 the right block (n > 128).
__device__ __forceinline__ double pairwise_block(const double* a, int n, int l8) {
    double r = a[l8];
    const int lim = n - (n % 8);
    for (int i = 8; i < lim; i += 8) r = __dadd_rn(r, a[i + l8]);
    r = __dadd_rn(r, __shfl_xor(r, 1));
    r = __dadd_rn(r, __shfl_xor(r, 2));
    r = __dadd_rn(r, __shfl_xor(r, 4));
    for (int i = lim; i < n; ++i) r = __dadd_rn(r, a[i]);
    return r;
}

__device__ __forceinline__ double pairwise_sum_dev(const double* a, int n) {
    const int lane = lane_id();
    const int l8 = lane & 7;
    const int grp = lane >> 3;
    if (n <= 128) {
        const double r = pairwise_block(a, n, l8);
        return __shfl(r, 0);
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const double* base = (grp == 1) ? a + n2 : a;
    const int len = (grp == 1) ? n - n2 : n2;
    const double r = pairwise_block(base, len, l8);
    const double left = __shfl(r, 0);
    const double right = __shfl(r, 8);
    return __dadd_rn(left, right);
}

// ----------------------------------------------------------------------------------------------
// k_expand_backup
// ----------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ void expand_backup_game(const TreeParams& p, const int g, uint8_t* s_ord /*[256]*/,
                                                   double* s_prior /*[256]*/, int16_t* s_tab /*[256]*/, GameHdr* hdr = nullptr) {
    const int lane = lane_id();
    // TWO memory round trips for everything the step needs to know: (1) what depends on the game only -- leaf record, status,
    // counters, the first 64 * NCH path entries, the next selection's header (GameHdr), the game's row of the evaluation batch --
    // and (2) that row's policy and value. Every vector load is unconditional (clamped index) and the values are pinned by ONE
    // empty asm after the last request: written as `cond ? load : 0` each load sat in its own predicated block with a
    // `s_waitcnt vmcnt(0)` behind it -- six dependent round trips where the comments said one (ISA; AO_PROF: 14 - 20 k cycles
    // for expansion + backup, most of it here).
    const size_t pbase = static_cast<size_t>(g) * p.maxd;
    const int status = p.leaf_status[g];
    const int arena = p.cur[g];
    const int depth = p.path_len[g];
    const int newn = p.nodes_used[g];
    const int done = p.sims_done[g];
    const int row = p.row_of_game ? p.row_of_game[g] : g;
    const int target = p.sims_target[g];
    const int root_node = p.root_node[g];
    const int mtpos = p.mtpos[g];
    int active_v = 1;
    if (p.active) active_v = p.active[g];
    PosR lp = pos_load(p.leaf_pos + g);
    int pn0[NCH], pe0[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int d = lane + 64 * c;
        const int dc = d < p.maxd ? d : p.maxd - 1;
        pn0[c] = p.path_node[pbase + dc];
        pe0[c] = p.path_edge[pbase + dc];
    }
    float v_eval = p.value[row];
    float pol[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cell = lane + 64 * c;
        pol[c] = p.policy[static_cast<size_t>(row) * p.A + (cell < p.A ? cell : p.A - 1)];
    }
    // the path entries' edge statistics (the backup's read-modify-write below) are requested here, before the expansion's
    // arithmetic, not after it: the rows of the new node and the parent's CH entry are the only things the expansion writes
    int bk_n[NCH];
    float bk_w[NCH];
    {
        asm volatile("" : "+v"(pn0[0]), "+v"(pe0[0]));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = lane + 64 * c;
            const bool in = d < depth && status != LS_IDLE;
            const size_t bs = node_slot(p, arena, g, in ? pn0[c] : 0);
            const int e = in ? pe0[c] : 0;
            bk_n[c] = rowN(p, bs)[e];
            bk_w[c] = rowW(p, bs)[e];
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(pn0[c]), "+v"(pe0[c]), "+v"(pol[c]), "+v"(bk_n[c]), "+v"(bk_w[c]));
    asm volatile("" : "+v"(active_v), "+v"(v_eval), "+v"(lp.ply), "+v"(lp.nchild), "+v"(lp.last64), "+v"(lp.bb[0][0]), "+v"(lp.bb[1][0]));
    if (hdr) {   // the next selection's header (see GameHdr); `done` / `root_node` are updated below
        hdr->is_active = active_v;
        hdr->target = target;
        hdr->root_node = root_node;
        hdr->mtpos = mtpos;
        hdr->arena = arena;
        hdr->batch_row = row;
        hdr->done = done;
        hdr->prev_status = status;
        hdr->valid = 1;
    }
    if (status == LS_IDLE || status == LS_WAIT || status == LS_WAIT_ROOT || status == LS_DESCEND) return;   // (a waiting leaf / a paused descent: see select_game)
    float v = 0.f;
    if (status == LS_EXPAND || status == LS_EXPAND_ROOT) {
        if (newn >= p.cap) {
            if (lane == 0) { atomicOr(&p.err[g], ERR_NODE_CAP); p.sims_done[g] = target; }
            if (hdr) hdr->done = target;
            return;
        }
        const int L = legal_order<NCH>(lp, p.A, s_ord, s_tab);
        // prior_prob = zeros(A); prior_prob[a] = policy[a] for legal a (agents.py:183-187)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int cell = lane + 64 * c;
            if (cell < p.A) s_prior[cell] = pos_occupied(lp, cell) ? 0.0 : static_cast<double>(pol[c]);
        }
        wsync();
        const double sum = pairwise_sum_dev(s_prior, p.A);  // prior_prob.sum() (agents.py:189)
        const bool add_noise = p.noise && status == LS_EXPAND_ROOT;  // agents.py:191-204
        const size_t slot = node_slot(p, arena, g, newn);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = lane + 64 * c;
            if (i < L) {
                const int a = s_ord[i];
                double pr = __ddiv_rn(s_prior[a], sum);
                if (add_noise) {
                    const double t1 = __dmul_rn(0.75, pr);
                    const double t2 = __dmul_rn(0.25, p.noise_buf[static_cast<size_t>(g) * p.Ap + i]);
                    pr = __dadd_rn(t1, t2);
                }
                rowN(p, slot)[i] = 0;
                rowW(p, slot)[i] = 0.f;
                rowQ(p, slot)[i] = 0.f;
                rowP(p, slot)[i] = pr;
                rowCH(p, slot)[i] = CH_UNVISITED;
                rowACT(p, slot)[i] = static_cast<uint8_t>(a);
            }
        }
        // link from the parent edge (the last path entry)
        int pn = 0, pe = 0;
        if (status == LS_EXPAND) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int t_n = read_lane(pn0[c], (depth - 1) & 63), t_e = read_lane(pe0[c], (depth - 1) & 63);
                if (((depth - 1) >> 6) == c) { pn = t_n; pe = t_e; }
            }
        }
        if (lane == 0) {
            lp.nchild = L;
            pos_store(nodePos(p, slot), lp);
            p.nodes_used[g] = newn + 1;
            if (status == LS_EXPAND) rowCH(p, node_slot(p, arena, g, pn))[pe] = newn;
            else p.root_node[g] = newn;
        }
        if (hdr && status == LS_EXPAND_ROOT) hdr->root_node = newn;
        v = v_eval;
    }
    // backup (agents.py:223-239): the edge into the leaf gets -v (or +1 for a terminal leaf,
    // draws included), the sign alternates towards the root. The root's own record is not kept.
    const bool terminal = (status == LS_TERMINAL);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int d = lane + 64 * c;
        if (d >= depth) continue;
        const int cnt = depth - 1 - d;
        float s;
        if (terminal) s = (cnt & 1) ? -1.f : 1.f;
        else s = (cnt & 1) ? v : -v;
        const size_t bs = node_slot(p, arena, g, pn0[c]);
        const int e = pe0[c];
        const int n = bk_n[c] + 1;
        const float w = __fadd_rn(bk_w[c], s);
        rowN(p, bs)[e] = n;
        rowW(p, bs)[e] = w;
        rowQ(p, bs)[e] = __fdiv_rn(w, static_cast<float>(n));
    }
    if (lane == 0) p.sims_done[g] = done + 1;
    if (hdr) hdr->done = done + 1;
}


}  // namespace ao
