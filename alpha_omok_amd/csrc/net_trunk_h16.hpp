// net_trunk_h16.hpp -- the conv stack with the fp32 contraction carried by fp16 MFMAs on split operands:
// k_trunk16h (group-resident) and k_layer16h (per layer). Included by net.hip after net_trunk_f32.hpp
// (trunk_heads is shared).
#pragma once

namespace ao {

// ----------------------------------------------------------------------------------------------
struct TrunkHLayer {
    const uint4* wh;   // [tap 9][c32][tile][lane 64] 8 x fp16: high halves, lane = oct*16 + cout
    const uint4* wl;   // low halves
    const float4* sc;  // BatchNorm scale x 2^-s (s = the layer's weight pre-scale)
    const float4* sh;
    int* ovf;          // status word of the network: bit 0 is set when an activation left the fp16 range (ao_net_status)
};

struct TrunkHArgs {
    const float4* in0;  // conv1 input: fp32 plane batch [grp][cell][quad 8][board 16], or the bit planes [grp][board 16][row] (IN = 2)
    uint4* bufA;  // conv1 output / ResBlock input-output (split-fp16 layout)
    uint4* bufB;
    int nlayers;  // 1 + 2 * n_block, conv1 included
    int CQ, COUT;
    const float *w3, *sc3, *sh3, *wp_t, *bp, *w1_t, *b1, *w2, *b2;
    float* policy;
    float* value;
    const unsigned* live;   // live rows of this simulation's batch (net_common.hpp, live_groups16) or null
    unsigned row_cap;
    TrunkHLayer layers[kMaxTrunkLayers];
};

// All global traffic of k_trunk16h goes through buffer descriptors: address = descriptor base + one
// 32-bit per-lane offset (VGPR) + a uniform offset (SGPR). With plain pointers the compiler keeps a 64-bit
// address VGPR pair per access site, hoists them out of the loops and spills them (270 registers in
// the first version of this kernel).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 buf_ld_h8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// Cache-policy bits of the three activation streams (aux operand of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1),
// compile-time knobs for the A/B runs in profiles/r2h_trunk16h_cache_policy_ab.txt
#ifndef AO_AUX_RES
#define AO_AUX_RES 0     // residual loads of the conv2 epilogue
#endif
#ifndef AO_AUX_ST
#define AO_AUX_ST 0      // activation stores of every epilogue
#endif
#ifndef AO_AUX_STAGE
#define AO_AUX_STAGE 0   // LDS-direct loads that stage the input rows
#endif
__device__ __forceinline__ half4 buf_ld_h4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AO_AUX_RES));
}
__device__ __forceinline__ void buf_st_h4(half4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, AO_AUX_ST);
}

#ifdef AO_PROF
// phase timing of k_trunk16h (build with AO_EXTRA_FLAGS=-DAO_PROF; tools/time_net.py prints it): shader-clock
// cycles per wave of one group, summed over the trunk layers: [0] row-0 staging, [1] slab loops, [2] row
// epilogues, [3] row barriers, [4] last epilogue + layer boundary, [5] heads, [6] conv1 total
static __device__ unsigned long long ao_prof[8 * 12];   // (static: the header is compiled into two translation units)
#define AO_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define AO_ACC(k, t0, t1) prof[k] += (t1) - (t0)
#else
#define AO_T(x)
#define AO_ACC(k, t0, t1)
#endif

// One conv layer of one 16-board group. NCI = 32-channel blocks of the INPUT (NC32 for a trunk layer).
// FIRST = conv1: the input is the fp32 plane batch ([cell][quad 8][board 16][float4], 32 channels, 5 real); it
// is split into its two halves while it is staged (the engine's planes are 0/1 and have a zero low half, but
// ao_net_forward accepts any float planes).
// Knock-out switches for timing experiments (-DAO_KO=n together with -DAO_PROF; RESULTS ARE WRONG for n != 0, the
// default build has AO_KO = 0 and every condition below folds away): 1 weights loaded for the first slabs only,
// 2 LDS operand fragments read once per slab, 3 no staging of input rows, 4 no row epilogues (residual loads +
// stores), 5 / 6 activations of all groups aliased to an 85 / 170 MB footprint, 8 no heads, 10 the low halves move half
// their bytes (staged by lanes 0-31 only, stored / re-read as 4 bytes per lane: the traffic of a 3-byte activation format
// without its conversion work; BUT the unstaged half of every low fragment then keeps conv1's zeros in LDS, and an MFMA
// on zeros draws less power), 12 the same traffic with the staged 512 bytes loaded twice so that every operand stays
// data-like. Round 4 re-takes 3 and 4 with DATA-LIKE operands (the round-2 builds froze or zeroed what the MFMAs read and measured
// MFMA power, not traffic): 13 = 3 with the LDS row buffers pre-filled with pseudo-random activations (half of them zero, like
// post-ReLU data) that are then never refreshed; 14 = 4 with the activation buffers in HBM pre-filled the same way (net.hip
// ensure_workspace), so the rows staged are data-like although no epilogue ever writes them; 15 = both. Same switches in the
// per-layer kernel (trunk_h_layer_tile). Measured: profiles/r1j_trunk16h_phase_timing.txt, r3a_trunk16h_bytes_ko.txt,
// r4f_ko_datalike.txt
#ifndef AO_KO
#define AO_KO 0
#endif
#define AO_KO_NOSTAGE (AO_KO == 3 || AO_KO == 13 || AO_KO == 15)
#define AO_KO_NOEPI (AO_KO == 4 || AO_KO == 14 || AO_KO == 15)
// pseudo-random post-ReLU-like activation pair for the data-like knock-outs: half zeros, else a high half in [2^-6, 2) with a
// full 10-bit mantissa; the low half a 2^-11-scaled copy (as the residue of a split would be)
__device__ __forceinline__ unsigned ao_ko_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint4 ao_ko_fragment(unsigned seed, bool low) {
    unsigned wd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned out = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned r = ao_ko_hash(seed * 8u + k * 2u + h);
            const unsigned e = (low ? 4u : 15u) - ((r >> 10) & 7u);          // exponent field: 8 .. 15 (low: 11 binades below)
            const unsigned v = (r & 0x80000000u) ? 0u : ((e << 10) | (r & 0x3ffu));
            out |= v << (16 * h);
        }
        wd[k] = out;
    }
    return make_uint4(wd[0], wd[1], wd[2], wd[3]);
}
#if AO_KO != 0 && !defined(AO_PROF) && !defined(AO_WRONG_RESULTS_OK)
#error "AO_KO builds compute wrong results on purpose: timing only, build them with -DAO_PROF"
#endif
// Split row barrier of the resident kernel (-DAO_SPLIT_BARRIER=1; measured, NOT the default). The eight waves of a group
// share the two LDS row buffers, so between input rows they must agree that (a) every wave's share of the next row has
// landed and (b) every wave has finished reading the row whose buffer is refilled next. With the hardware s_barrier after
// the row epilogue the two waves of a SIMD run their epilogues (no MFMA) at the same time; in the split form a wave
// ARRIVES (one LDS atomic add on a monotonic counter) as soon as its slabs are done and its staging share has landed, runs
// its epilogue, and only then WAITS for the other seven, so one wave's epilogue can run beside its partner's MFMAs.
// Result (profiles/r2c_trunk16h_barrier_ab.txt, same box, 4096 boards): hard barrier 1.603-1.609 ms, split 1.600-1.601,
// and with NO row synchronisation at all (results wrong) 1.594 ms -- the row barrier costs < 1 %. The launch time is the
// sum of the MFMA time and the activation traffic's time (each 170 MB pass over the activations costs ~27 us whether
// or not waves wait for each other): the chip is power-limited, energy adds up, idle cycles are given back as clock.
#ifndef AO_SPLIT_BARRIER
#define AO_SPLIT_BARRIER 0
#endif
#ifndef AO_CONV1_SPLIT
#define AO_CONV1_SPLIT 0   // 1: conv1 on bit planes requests the next row's bytes one slab before they are written to LDS -- measured neutral (r3j: 1.3964 / 1.3953 ms), off
#endif
__device__ __forceinline__ void row_arrive(unsigned* cnt, int lane) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void row_wait(unsigned* cnt, unsigned target) {
    for (;;) {
        const unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane(v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// KIND: 0 = a trunk layer (split-fp16 activations in), 1 = conv1 on the fp32 plane batch, 2 = conv1 on the engine's
// BIT planes: one byte per (board, cell), bit q = plane q ([board 16][kPlaneRow(BW)] bytes per group) -- the planes are
// 0/1, so 81 bytes per leaf carry what the fp32 batch spends 2.6 KB on (tree_device.hpp encode_planes).
__host__ __device__ constexpr int kPlaneRow(int bw) { return bw * bw <= 128 ? 128 : 256; }
// FMT: how the trunk activations of the RESIDENT kernel live in HBM between its layers (private to one launch):
//   0  two fp16 halves, x = xh + xl: 2 KB per (cell, 32-channel block) of a 16-board group, ~22 significand bits
//   1  fp16 high half + ONE low byte: 1.5 KB per (cell, block), 19 significand bits. Activations are post-ReLU (x >= 0):
//      with t = RNE_19bit(x) as the 24-bit word [E5 | M18] (fp32 bits re-biased to fp16's exponent), the high half is
//      t >> 8 -- a valid fp16, the TRUNCATION of the rounded value -- and the low byte is t & 255, the next 8 mantissa
//      bits, i.e. xl = byte * 2^(E5 - 33) >= 0. The high halves are still staged by LDS-direct loads; the low bytes are
//      staged into the first half of their LDS slot and expanded in place to the fp16 fragment the MFMAs read (one
//      conversion per element and layer, by the wave that staged it). 25 % fewer activation bytes per pass:
//      profiles/r3a_trunk16h_bytes_ko.txt (-15 % launch time as a traffic knock-out), numerics gate
//      profiles/r3a_lo8_numerics_emulation.txt (tools/emulate_lo8_storage.py).
__host__ __device__ constexpr unsigned kPairBytes(int fmt) { return fmt == 1 ? 1536u : 2048u; }

// The epilogue of format 1 works on activations SCALED BY 2^-112 (folded into the BatchNorm scale / shift): an fp32 number
// v' = x * 2^-112 carries fp16's exponent field directly (E8 = E5 for 2^-14 <= x < 65536), so the 24-bit word is its bits
// >> 5 after rounding to nearest even at bit 5, no re-biasing; x < 2^-14 (v' below the smallest normal fp32) is flushed to
// zero. Decoding is the reverse: as_float(word << 5) IS x * 2^-112, ready to be added to the scaled BatchNorm output.
constexpr float kLo8Scale = 0x1p-112f;
__device__ __forceinline__ float lo8_scaled(unsigned h, unsigned l) { return __uint_as_float(((h << 8) | l) << 5); }
__device__ __forceinline__ unsigned lo8_word(float vs) {   // vs: scaled, clamped to [0, 65504 * 2^-112], 0 below 2^-126
    const unsigned u = __float_as_uint(vs);
    return (u + 15u + ((u >> 5) & 1u)) >> 5;
}

// W16: every conv weight of the network (after its layer's power-of-two pre-scale) IS an fp16 number, so the low weight halves are
// all zero and the xh * wl product adds exact zeros: it is left out -- TWO products per multiply-add, (xh + xl) * wh, the same
// bits as the three-product form on such weights (an MFMA on a zero operand leaves its accumulator as it was), a third fewer
// MFMAs on a power-limited kernel and no low-half weight loads. ao_net_finalize finds out (net.hip, Net::w16); checkpoints get
// there by keeping their conv weights on the fp16 grid (tools/train_omok.py --fp16-grid-weights).
template <int BW, int NC32, int NCI, int KIND, int FMT = 0, bool W16 = false>
struct TrunkHLayerFn {
static __device__ __forceinline__ void run(const void* src, uint4* dst, const TrunkHLayer& L, const bool RES, uint4* s_x,
                                           int tile, int lane, unsigned long long* prof, const bool flip,
                                           unsigned* s_cnt, const unsigned rows_before) {
    constexpr bool FIRST = KIND != 0;
    constexpr bool BITS = KIND == 2;
    // flip: this layer walks the board from the LAST row to the first (logical row y = physical row BW-1-y, tap rows
    // mirrored). Layers alternate direction, so a layer starts with the rows the previous one wrote last -- still in
    // L2 / the Infinity Cache -- instead of the ones that left the caches 340 MB of traffic ago.
    auto py = [&](int y) { return flip ? BW - 1 - y : y; };
    constexpr int A = BW * BW;
    constexpr int NT = NC32 * 2;             // 16-channel output tiles = waves (two per SIMD at 128 channels)
    constexpr int NSP = 2;                   // halves of an input fragment
    constexpr int NFR = BW * NCI * NSP;      // input fragments per board row
    constexpr int NB = NCI * 3;              // (32-channel block, tap row) slabs per input row
    constexpr int NPR = 3;                   // MFMA products per multiply-add
    constexpr int PAIR = static_cast<int>(kPairBytes(FMT));   // bytes of a (cell, block) fragment pair in HBM
    constexpr int NPAIR = BW * NCI;          // fragment pairs per board row
    const int kq = lane >> 4, b = lane & 15;
    const int lane16 = lane * 16;
    // per-lane byte offset of this lane's 4 output channels inside a (cell, 32-channel block) fragment pair
    const int out_voff = (((tile & 1) * 2 + (kq >> 1)) * 16 + b) * 16 + (kq & 1) * 8;
    float4 sc = L.sc[tile * 4 + kq], sh = L.sh[tile * 4 + kq];
    if (FMT == 1) {   // the epilogue runs on x * 2^-112 (see lo8_word)
        sc.x *= kLo8Scale; sc.y *= kLo8Scale; sc.z *= kLo8Scale; sc.w *= kLo8Scale;
        sh.x *= kLo8Scale; sh.y *= kLo8Scale; sh.z *= kLo8Scale; sh.w *= kLo8Scale;
    }
    const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(L.wh, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(L.wl, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_src =
        make_rsrc(src, BITS ? 16u * kPlaneRow(BW) : FIRST ? static_cast<unsigned>(A) * 8u * 16u * 16u : static_cast<unsigned>(A) * NCI * kPairBytes(FMT));
    const __amdgpu_buffer_rsrc_t rs_dst = make_rsrc(dst, static_cast<unsigned>(A) * NC32 * kPairBytes(FMT));
    // weights of slab (c, dy): 3 taps x {high, low}, streamed from L2 one slab ahead (the other wave of the
    // SIMD computes meanwhile)
    // (conv1 has a single 32-channel block: its 9 x 2 fragments are simply loaded once)
    half8 wA[2][3], wB[2][3], wres[2][FIRST ? 9 : 1];
    auto load_w = [&](int slab, half8 (&W)[2][3]) {
        if (FIRST) return;
        if (AO_KO == 1 && slab > 2) return;
        const int c = (slab / 3) % NCI, dy = slab % 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ub = ((((flip ? 2 - dy : dy) * 3 + dx) * NCI + c) * NT + tile) * 1024;
            W[0][dx] = buf_ld_h8(rs_wh, lane16, ub);
            if (!W16) W[1][dx] = buf_ld_h8(rs_wl, lane16, ub);
        }
    };
    if (FIRST) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wres[0][t] = buf_ld_h8(rs_wh, lane16, (t * NT + tile) * 1024);
            if (!W16) wres[1][t] = buf_ld_h8(rs_wl, lane16, (t * NT + tile) * 1024);
        }
    }
    // conv1: one fragment = channels 8*kq .. 8*kq+7 of (cell, board b) = two float4 quads of the fp32 batch
    auto load_planes = [&](int cell, int split) -> half8 {   // split 0: high halves, 1: low halves (0 for 0/1 planes)
        if (BITS) {
            // channels 0..7 of (cell, board b) = the bits of one byte; only the kq == 0 lanes carry channels < 8
            const unsigned bits = __builtin_amdgcn_raw_buffer_load_b8(rs_src, b * kPlaneRow(BW) + cell, 0, 0);
            half8 h;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                h[k] = (kq == 0 && split == 0 && ((bits >> k) & 1u)) ? static_cast<_Float16>(1.0f) : static_cast<_Float16>(0.0f);
            return h;
        }
        const int o = ((cell * 8 + 2 * kq) * 16 + b) * 16;
        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_src, o, 0, 0);
        const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_src, o + 256, 0, 0);
        half8 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = __uint_as_float(q0[k]), v1 = __uint_as_float(q1[k]);
            const _Float16 h0 = static_cast<_Float16>(v0), h1 = static_cast<_Float16>(v1);
            h[k] = split ? static_cast<_Float16>(v0 - static_cast<float>(h0)) : h0;
            h[4 + k] = split ? static_cast<_Float16>(v1 - static_cast<float>(h1)) : h1;
        }
        return h;
    };
    float peak = 0.f;  // largest pre-clamp activation this lane wrote
    unsigned pbits[BITS ? (NFR + NT - 1) / NT : 1] = {};   // conv1 on bit planes: plane bytes of the next row in flight
    f32x4 acc[3][BW];  // output rows yi-1, yi, yi+1
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < BW; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Output row in two batches of cells: all residual loads of a batch are issued before the first is used
    // (written cell by cell the compiler produced load, wait, store, load, wait-for-everything ...: nine serial
    // memory round trips per row, 3.8 us; the registers of the X fragments are free here)
    auto epilogue = [&](int yo) {
        constexpr int HB = (BW + 1) / 2;
#pragma unroll
        for (int i0 = 0; i0 < BW; i0 += HB) {
            half4 rh[HB], rl[HB];
            unsigned rb[HB];   // FMT 1: the four low bytes of the residual
            if (RES) {
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const int i = i0 + k < BW ? i0 + k : BW - 1;
                    const int ob = ((py(yo) * BW + i) * NC32 + (tile >> 1)) * PAIR;
                    rh[k] = buf_ld_h4(rs_dst, out_voff, ob);
                    if (FMT == 1) {
                        rb[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_dst, out_voff >> 1, ob + 1024, AO_AUX_RES);
                    } else if (AO_KO == 10 || AO_KO == 12) {
                        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(rs_dst, out_voff >> 1, ob + 1024, AO_AUX_RES);
                        rl[k] = __builtin_bit_cast(half4, u32x2{u, u});
                    } else {
                        rl[k] = buf_ld_h4(rs_dst, out_voff, ob + 1024);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int i = i0 + k;
                if (i >= BW) continue;
                const f32x4 c = acc[0][i];
                float f[4] = {fmaf(c[0], sc.x, sh.x), fmaf(c[1], sc.y, sh.y), fmaf(c[2], sc.z, sh.z), fmaf(c[3], sc.w, sh.w)};
                const int ob = ((py(yo) * BW + i) * NC32 + (tile >> 1)) * PAIR;
                if (RES) {
                    if (FMT == 1) {
                        const u32x2 hb = __builtin_bit_cast(u32x2, rh[k]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            f[r] += lo8_scaled((hb[r >> 1] >> (16 * (r & 1))) & 0xffffu, (rb[k] >> (8 * r)) & 0xffu);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] += static_cast<float>(rh[k][r]) + static_cast<float>(rl[k][r]);
                    }
                }
                peak = fmaxf(fmaxf(peak, fmaxf(f[0], f[1])), fmaxf(f[2], f[3]));
                if (FMT == 1) {
                    unsigned t[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)   // ReLU + flush below 2^-14 (the smallest normal fp32 in the scaled domain), upper clamp
                        t[r] = lo8_word(f[r] >= 0x1p-126f ? fminf(f[r], 65504.f * kLo8Scale) : 0.f);
                    const u32x2 hw = {(t[0] >> 8) | ((t[1] >> 8) << 16), (t[2] >> 8) | ((t[3] >> 8) << 16)};
                    const unsigned lw = (t[0] & 0xffu) | ((t[1] & 0xffu) << 8) | ((t[2] & 0xffu) << 16) | (t[3] << 24);
                    __builtin_amdgcn_raw_buffer_store_b64(hw, rs_dst, out_voff, ob, AO_AUX_ST);
                    __builtin_amdgcn_raw_buffer_store_b32(lw, rs_dst, out_voff >> 1, ob + 1024, AO_AUX_ST);
                    continue;
                }
                half4 hh, hl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // ReLU; the upper clamp keeps an activation beyond the fp16 range (65504 -- far outside what a
                    // BatchNorm-ed residual tower produces) finite instead of turning the board into inf / NaN; the
                    // event is reported through L.ovf (ao_net_status), the result of such a forward is not fp32-equivalent
                    const float v = fminf(fmaxf(f[r], 0.f), 65504.f);
                    hh[r] = static_cast<_Float16>(v);
                    hl[r] = static_cast<_Float16>(v - static_cast<float>(hh[r]));
                }
                buf_st_h4(hh, rs_dst, out_voff, ob);
                if (AO_KO == 10 || AO_KO == 12) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32x2, hl)[0], rs_dst, out_voff >> 1, ob + 1024, AO_AUX_ST);
                else
                buf_st_h4(hl, rs_dst, out_voff, ob + 1024);
            }
        }
    };
    // FMT 1 staging of input row y into row buffer xb: wave w owns fragment PAIRS w, w + NT, ... -- the high half by an
    // LDS-direct load of all 64 lanes, the 512 low bytes by lanes 0-31 into the first half of the pair's low slot
    auto stage_pairs = [&](int y, uint4* xb, int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            const int p = tile + NT * k;
            if (p < NPAIR) {
                const int so = (py(y) * NPAIR + p) * PAIR;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xb + (2 * p) * 64), 16, lane16,
                                                         so, 0, AO_AUX_STAGE);
                if (lane < 32)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xb + (2 * p + 1) * 64), 16,
                                                             lane16, so + 1024, 0, AO_AUX_STAGE);
            }
        }
    };
    // ... and, once those loads have landed (s_waitcnt vmcnt(0) of THIS wave: it staged both halves of its pairs), the
    // expansion of the low bytes to the fp16 low-half fragment, in place: xl = byte * 2^(E5 - 33), exact in fp16 (down to
    // its subnormal quantum, which the MFMA honours: tools/mfma_denorm.hip). DS operations of a wave execute in order, so
    // every lane's 8-byte read of the slot precedes the 16-byte writes.
    auto expand_pairs = [&](uint4* xb, int k0, int k1) {
        // per dword (two channels): y = 1024 + byte as fp16 (0x6400 | byte), c = y * 2^-18 - 2^-8 = byte * 2^-18 (exact, one
        // rounding), xl = c * 2^(E5-15), the second factor being the high half with its mantissa cleared
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        const half2v k18 = {static_cast<_Float16>(0x1p-18f), static_cast<_Float16>(0x1p-18f)};
        const half2v m8 = {static_cast<_Float16>(-0x1p-8f), static_cast<_Float16>(-0x1p-8f)};
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            const int p = tile + NT * k;
            if (p < NPAIR) {
                const uint4 hv = xb[(2 * p) * 64 + lane];
                const u32x2 lv = reinterpret_cast<const u32x2*>(xb + (2 * p + 1) * 64)[lane];
                const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
                unsigned ow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // bytes (2j, 2j+1) of the 8 low bytes -> 0x64 b1 0x64 b0
                    const unsigned y = __builtin_amdgcn_perm(0x64646464u, lv[j >> 1], (j & 1) ? 0x04030402u : 0x04010400u);
                    const half2v c = __builtin_elementwise_fma(__builtin_bit_cast(half2v, y), k18, m8);
                    const half2v sc2 = __builtin_bit_cast(half2v, hw[j] & 0x7c007c00u);
                    ow[j] = __builtin_bit_cast(unsigned, c * sc2);
                }
                xb[(2 * p + 1) * 64 + lane] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        }
    };

    // stage input row 0 (wave w copies fragments w, w + NT, ...)
    AO_T(t_a);
    __syncthreads();  // the previous layer is done with both row buffers
    AO_T(t_a1);
    // (all loads of the wave in flight at once: written as a loop over f the compiler emits load, wait, LDS write
    // per fragment -- nine serial HBM round trips, 10 us per layer)
    if (FIRST) {
#pragma unroll
        for (int k = 0; k < (NFR + NT - 1) / NT; ++k) {
            const int f = tile + NT * k;
            if (f < NFR) s_x[f * 64 + lane] = __builtin_bit_cast(uint4, load_planes(f >> 1, f & 1));
        }
    } else if (FMT == 1) {
        stage_pairs(0, s_x, 0, (NPAIR + NT - 1) / NT);
    } else if (!(AO_KO == 13 || AO_KO == 15)) {
#pragma unroll
        for (int k = 0; k < (NFR + NT - 1) / NT; ++k) {
            const int f = tile + NT * k;
            if (f < NFR && !(AO_KO == 10 && (f & 1) && lane >= 32))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(s_x + f * 64), 16,
                                                         (AO_KO == 12 && (f & 1) && lane >= 32) ? lane16 - 512 : lane16,
                                                         (py(0) * NFR + f) * 1024, 0, AO_AUX_STAGE);
        }
    }
    AO_T(t_a2);
    load_w(0, wA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (FMT == 1 && !FIRST) {
        expand_pairs(s_x, 0, (NPAIR + NT - 1) / NT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    AO_T(t_a3);
    __syncthreads();
    AO_T(t_b);
    AO_ACC(0, t_a, t_b);
    AO_ACC(8, t_a, t_a1);
    AO_ACC(9, t_a1, t_a2);
    AO_ACC(10, t_a2, t_a3);
    AO_ACC(11, t_a3, t_b);

    for (int yi = 0; yi < BW; ++yi) {
        AO_T(t_r0);
        const uint4* xs = s_x + static_cast<size_t>(yi & 1) * NFR * 64;         // this row
        uint4* xn = s_x + static_cast<size_t>((yi + 1) & 1) * NFR * 64;         // next row's buffer
        const int yn = yi + 1 < BW ? yi + 1 : yi;                               // next input row (clamped)
#pragma unroll
        for (int slab = 0; slab < NB; ++slab) {
            const int c = slab / 3, dy = slab % 3;
            // (NB is even for the trunk layers: the buffer parity carries over from one row to the next)
            half8 (&w)[2][3] = (slab & 1) ? wB : wA;
            half8 (&wn)[2][3] = (slab & 1) ? wA : wB;
            load_w(slab + 1, wn);
            if (dy == 1 && !AO_KO_NOSTAGE && FMT == 1 && !FIRST) {
                // next input row into LDS, a share of this wave's fragment pairs per block
                // (the share staged one block earlier has landed long ago: it is expanded first, so that the wait the
                // compiler puts in front of its LDS reads does not cover the loads issued below)
                constexpr int K2 = (NPAIR / NT + NCI - 1) / (NCI > 1 ? NCI - 1 : 1);   // shares in blocks 0 .. NCI-2, the last block only expands
                if (c > 0 && AO_KO != 11) expand_pairs(xn, (c - 1) * K2, c * K2);   // (AO_KO 11: no expansion, timing only)
                if (c + 1 < NCI) stage_pairs(yn, xn, c * K2, (c + 1) * K2);
            } else if (BITS && !AO_KO_NOSTAGE && AO_CONV1_SPLIT) {
                // conv1 on bit planes: the next row's plane bytes are requested in the first slab of the row and turned
                // into fragments in the last one, so the byte loads' round trip is covered by a slab of MFMAs (a conv1
                // row has only three slabs; load and LDS write in the same slab stalled every row)
                constexpr int KF = (NFR + NT - 1) / NT;
#pragma unroll
                for (int k = 0; k < KF; ++k) {
                    const int f = tile + NT * k;
                    if (f < NFR) {
                        if (dy == 0 && !(f & 1))
                            pbits[k] = __builtin_amdgcn_raw_buffer_load_b8(rs_src, b * kPlaneRow(BW) + yn * BW + (f >> 1), 0, 0);
                        if (dy == 2) {
                            half8 h;
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                h[q] = (kq == 0 && !(f & 1) && ((pbits[k] >> q) & 1u)) ? static_cast<_Float16>(1.0f) : static_cast<_Float16>(0.0f);
                            xn[f * 64 + lane] = __builtin_bit_cast(uint4, h);
                        }
                    }
                }
            } else if (dy == 1 && (!AO_KO_NOSTAGE || FIRST)) {
                // next input row into LDS, a share per block (always-executed slab)
#pragma unroll
                for (int k = 0; k < (NFR / NT + NCI) / NCI; ++k) {
                    const int f = tile + NT * (c * ((NFR / NT + NCI) / NCI) + k);
                    if (f < NFR) {
                        if (FIRST) xn[f * 64 + lane] = __builtin_bit_cast(uint4, load_planes(yn * BW + (f >> 1), f & 1));
                        else if (!(AO_KO == 10 && (f & 1) && lane >= 32))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xn + f * 64),
                                                                     16, (AO_KO == 12 && (f & 1) && lane >= 32) ? lane16 - 512 : lane16,
                                                                     (py(yn) * NFR + f) * 1024, 0, AO_AUX_STAGE);
                    }
                }
            }
            const int yo = yi + 1 - dy;
            if (yo >= 0 && yo < BW) {   // (uniform)
                half8 xh = __builtin_bit_cast(half8, xs[((0 * NCI + c) * NSP + 0) * 64 + lane]);
                half8 xl = __builtin_bit_cast(half8, xs[((0 * NCI + c) * NSP + 1) * 64 + lane]);
#pragma unroll
                for (int xi = 0; xi < BW; ++xi) {
                    half8 nh = xh, nl = xl;
                    if (xi + 1 < BW && AO_KO != 2) {
                        nh = __builtin_bit_cast(half8, xs[(((xi + 1) * NCI + c) * NSP + 0) * 64 + lane]);
                        nl = __builtin_bit_cast(half8, xs[(((xi + 1) * NCI + c) * NSP + 1) * 64 + lane]);
                    }
                    // input cell (yi, xi) feeds output row yo at cells xi-dx+1: xh*wh, xh*wl, xl*wh, ordered so that
                    // consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr) {
                        if (BITS && pr == 2) continue;   // bit planes are 0 / 1: their low halves are zero, no xl * wh product
                        if (W16 && pr == 1) continue;    // fp16 weights: their low halves are zero, no xh * wl product
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int i = xi - dx + 1;
                            if (i < 0 || i >= BW) continue;
                            const half8 wv = FIRST ? wres[pr == 1 ? 1 : 0][FIRST ? dy * 3 + dx : 0] : w[pr == 1 ? 1 : 0][dx];
                            acc[2 - dy][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, pr == 2 ? xl : xh, acc[2 - dy][i], 0, 0, 0);
                        }
                    }
                    xh = nh;
                    xl = nl;
                    // keeps the scheduler from hoisting every LDS read of the slab to its top (72 registers)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        AO_T(t_r1);
#if AO_SPLIT_BARRIER
        // arrive: this wave's share of the next row has landed (LDS-direct loads count in vmcnt) and its reads of this
        // row are done
        row_arrive(s_cnt, lane);
#endif
        if (yi >= 1 && (!AO_KO_NOEPI || (AO_KO == 4 && yi == 1))) epilogue(yi - 1);
        else if (yi >= 1 && (AO_KO == 14 || AO_KO == 15)) {
            // (no epilogue, but the row's accumulators stay "used": without a consumer the compiler deletes the MFMAs that
            // feed them -- the first data-like no-epilogue build ran in 0.27 ms, and round 2's AO_KO=4 had lost part of its
            // MFMAs the same way)
#pragma unroll
            for (int i = 0; i < BW; ++i) asm volatile("" ::"v"(acc[0][i]));
        }
#pragma unroll
        for (int i = 0; i < BW; ++i) {
            acc[0][i] = acc[1][i];
            acc[1][i] = acc[2][i];
            acc[2][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        AO_T(t_r2);
#if AO_SPLIT_BARRIER
        // wait: all NT waves arrived for this row (the layer boundary below is a full barrier: no wait after the last row)
        if (yi + 1 < BW && AO_KO != 9) row_wait(s_cnt, static_cast<unsigned>(NT) * (rows_before + static_cast<unsigned>(yi) + 1u));
#else
        // next row staged by all waves (LDS-direct loads count in vmcnt), this row's buffer free
        if (AO_KO == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        AO_T(t_r3);
        AO_ACC(1, t_r0, t_r1);
        AO_ACC(2, t_r1, t_r2);
        AO_ACC(3, t_r2, t_r3);
    }
    AO_T(t_c);
    if (!AO_KO_NOEPI || FIRST) epilogue(BW - 1);
    if (peak > (FMT == 1 ? 65504.f * kLo8Scale : 65504.f)) atomicOr(L.ovf, 1);
    // layer boundary inside the workgroup (see k_trunk16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    AO_T(t_d);
    AO_ACC(4, t_c, t_d);
}
};

// The same layer for ONE (row chunk [yb, ye), column tile x0 .. x0+XT-1) of a group: the per-layer form for
// batches too small to give every CU a whole group (k_layer16h: one launch per conv, workgroup = group x row
// chunk x column tile) and for boards whose rows do not fit LDS (15 x 15: XT = 5, a staged row is the tile
// plus one halo column on each side = 7 cells = 56 KB, two of them 112 KB). Halo columns that fall off the
// board are staged as zeros, so the MFMA stream needs no per-column conditions; halo rows are handled by the
// slab conditions (uniform per row) exactly as in the fp32 row-chunk kernel.
template <int BW, int XT, int NC32, int NCI, int KIND, bool W16 = false>
__device__ __forceinline__ void trunk_h_layer_tile(const void* src, uint4* dst, const TrunkHLayer& L, const bool RES,
                                                   uint4* s_x, int tile, int lane, int x0, int yb, int ye) {
    constexpr bool FIRST = KIND != 0;
    constexpr bool BITS = KIND == 2;
    constexpr int A = BW * BW;
    constexpr bool HALO = XT < BW;
    constexpr int NX = HALO ? XT + 2 : XT;   // staged input cells per row; staged cell j = board column x0 - 1 + j (HALO) or j
    constexpr int NT = NC32 * 2;
    constexpr int NSP = 2;
    constexpr int NFR = NX * NCI * NSP;
    constexpr int NB = NCI * 3;
    constexpr int NPR = 3;
    const int kq = lane >> 4, b = lane & 15;
    const int lane16 = lane * 16;
    const int out_voff = (((tile & 1) * 2 + (kq >> 1)) * 16 + b) * 16 + (kq & 1) * 8;
    const float4 sc = L.sc[tile * 4 + kq], sh = L.sh[tile * 4 + kq];
    const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(L.wh, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(L.wl, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_src =
        make_rsrc(src, BITS ? 16u * kPlaneRow(BW) : FIRST ? static_cast<unsigned>(A) * 8u * 16u * 16u : static_cast<unsigned>(A) * NCI * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rs_dst = make_rsrc(dst, static_cast<unsigned>(A) * NC32 * 2u * 1024u);
    half8 wA[2][3], wB[2][3], wres[2][FIRST ? 9 : 1];
    auto load_w = [&](int slab, half8 (&W)[2][3]) {
        if (FIRST) return;
        const int c = (slab / 3) % NCI, dy = slab % 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ub = (((dy * 3 + dx) * NCI + c) * NT + tile) * 1024;
            W[0][dx] = buf_ld_h8(rs_wh, lane16, ub);
            if (!W16) W[1][dx] = buf_ld_h8(rs_wl, lane16, ub);
        }
    };
    if (FIRST) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wres[0][t] = buf_ld_h8(rs_wh, lane16, (t * NT + tile) * 1024);
            if (!W16) wres[1][t] = buf_ld_h8(rs_wl, lane16, (t * NT + tile) * 1024);
        }
    }
    auto load_planes = [&](int cell, int split) -> half8 {   // split 0: high halves, 1: low halves (0 for 0/1 planes)
        if (BITS) {
            // channels 0..7 of (cell, board b) = the bits of one byte; only the kq == 0 lanes carry channels < 8
            const unsigned bits = __builtin_amdgcn_raw_buffer_load_b8(rs_src, b * kPlaneRow(BW) + cell, 0, 0);
            half8 h;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                h[k] = (kq == 0 && split == 0 && ((bits >> k) & 1u)) ? static_cast<_Float16>(1.0f) : static_cast<_Float16>(0.0f);
            return h;
        }
        const int o = ((cell * 8 + 2 * kq) * 16 + b) * 16;
        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_src, o, 0, 0);
        const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_src, o + 256, 0, 0);
        half8 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = __uint_as_float(q0[k]), v1 = __uint_as_float(q1[k]);
            const _Float16 h0 = static_cast<_Float16>(v0), h1 = static_cast<_Float16>(v1);
            h[k] = split ? static_cast<_Float16>(v0 - static_cast<float>(h0)) : h0;
            h[4 + k] = split ? static_cast<_Float16>(v1 - static_cast<float>(h1)) : h1;
        }
        return h;
    };
    // stage this wave's share of input row y into row buffer `xb`
    auto stage = [&](int y, uint4* xb) {
#pragma unroll
        for (int k = 0; k < (NFR + NT - 1) / NT; ++k) {
            const int f = tile + NT * k;   // staged fragment: (cell j, block c, half)
            if (f < NFR) {
                const int j = f / (NCI * NSP), rest = f - j * (NCI * NSP);
                const int xin = HALO ? x0 - 1 + j : j;
                if (xin < 0 || xin >= BW) {
                    xb[f * 64 + lane] = make_uint4(0, 0, 0, 0);   // halo column outside the board
                } else if (FIRST) {
                    xb[f * 64 + lane] = __builtin_bit_cast(uint4, load_planes(y * BW + xin, rest));
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xb + f * 64), 16,
                                                             lane16, ((y * BW + xin) * NCI * 2 + rest) * 1024, 0, AO_AUX_STAGE);
                }
            }
        }
    };
    float peak = 0.f;
    f32x4 acc[3][XT];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < XT; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto epilogue = [&](int yo) {
        // all residual loads of the row first (see trunk_h_layer)
        half4 rh[XT], rl[XT];
        if (RES) {
#pragma unroll
            for (int i = 0; i < XT; ++i) {
                const int xo = x0 + i < BW ? x0 + i : BW - 1;
                const int ob = (((yo * BW + xo) * NC32 + (tile >> 1)) * 2) * 1024;
                rh[i] = buf_ld_h4(rs_dst, out_voff, ob);
                rl[i] = buf_ld_h4(rs_dst, out_voff, ob + 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            if (x0 + i >= BW) continue;   // (only when XT does not divide BW; uniform)
            const f32x4 c = acc[0][i];
            float f[4] = {fmaf(c[0], sc.x, sh.x), fmaf(c[1], sc.y, sh.y), fmaf(c[2], sc.z, sh.z), fmaf(c[3], sc.w, sh.w)};
            const int ob = (((yo * BW + x0 + i) * NC32 + (tile >> 1)) * 2) * 1024;
            if (RES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] += static_cast<float>(rh[i][r]) + static_cast<float>(rl[i][r]);
            }
            half4 hh, hl;
            peak = fmaxf(fmaxf(peak, fmaxf(f[0], f[1])), fmaxf(f[2], f[3]));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fminf(fmaxf(f[r], 0.f), 65504.f);
                hh[r] = static_cast<_Float16>(v);
                hl[r] = static_cast<_Float16>(v - static_cast<float>(hh[r]));
            }
            buf_st_h4(hh, rs_dst, out_voff, ob);
            buf_st_h4(hl, rs_dst, out_voff, ob + 1024);
        }
    };

    const int y0 = yb > 0 ? yb - 1 : 0;          // input rows that feed output rows [yb, ye)
    const int y1 = ye < BW ? ye : BW - 1;
    if ((AO_KO == 13 || AO_KO == 15) && !FIRST) {   // data-like LDS rows that are never refreshed
        for (int f = tile; f < 2 * NFR; f += NT) s_x[f * 64 + lane] = ao_ko_fragment((blockIdx.x * 2u * NFR + f) * 64u + lane, f & 1);
    } else
    stage(y0, s_x);
    load_w(0, wA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int yi = y0; yi <= y1; ++yi) {
        const int par = (yi - y0) & 1;
        const uint4* xs = s_x + static_cast<size_t>(par) * NFR * 64;
        uint4* xn = s_x + static_cast<size_t>(par ^ 1) * NFR * 64;
#pragma unroll
        for (int slab = 0; slab < NB; ++slab) {
            const int c = slab / 3, dy = slab % 3;
            half8 (&w)[2][3] = (slab & 1) ? wB : wA;
            half8 (&wn)[2][3] = (slab & 1) ? wA : wB;
            load_w(slab + 1, wn);
            if (slab == 1 && yi < y1 && (!AO_KO_NOSTAGE || FIRST)) stage(yi + 1, xn);
            const int yo = yi + 1 - dy;
            if (yo >= yb && yo < ye) {   // (uniform)
                half8 xh = __builtin_bit_cast(half8, xs[((0 * NCI + c) * NSP + 0) * 64 + lane]);
                half8 xl = __builtin_bit_cast(half8, xs[((0 * NCI + c) * NSP + 1) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    half8 nh = xh, nl = xl;
                    if (j + 1 < NX) {
                        nh = __builtin_bit_cast(half8, xs[(((j + 1) * NCI + c) * NSP + 0) * 64 + lane]);
                        nl = __builtin_bit_cast(half8, xs[(((j + 1) * NCI + c) * NSP + 1) * 64 + lane]);
                    }
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr) {
                        if (BITS && pr == 2) continue;   // (see trunk_h_layer)
                        if (W16 && pr == 1) continue;
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int i = HALO ? j - dx : j - dx + 1;   // output cell of the tile fed through tap column dx
                            if (i < 0 || i >= XT) continue;
                            const half8 wv = FIRST ? wres[pr == 1 ? 1 : 0][FIRST ? dy * 3 + dx : 0] : w[pr == 1 ? 1 : 0][dx];
                            acc[2 - dy][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, pr == 2 ? xl : xh, acc[2 - dy][i], 0, 0, 0);
                        }
                    }
                    xh = nh;
                    xl = nl;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (yi - 1 >= yb && (!AO_KO_NOEPI || FIRST)) epilogue(yi - 1);
        else if (yi - 1 >= yb && (AO_KO == 14 || AO_KO == 15)) {
#pragma unroll
            for (int i = 0; i < XT; ++i) asm volatile("" ::"v"(acc[0][i]));   // (see trunk_h_layer)
        }
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            acc[0][i] = acc[1][i];
            acc[1][i] = acc[2][i];
            acc[2][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (ye == BW && (!AO_KO_NOEPI || FIRST)) epilogue(BW - 1);
    if (peak > 65504.f) atomicOr(L.ovf, 1);
}

struct LayerHArgs {
    const void* src;   // fp32 plane batch (conv1) or split-fp16 activations
    uint4* dst;
    TrunkHLayer layer;
    int res, nch;
    const unsigned* live;   // live rows of this simulation's batch (net_common.hpp, live_groups16) or null
    unsigned row_cap;
};

template <int BW, int XT, int NC32, int KIND, bool W16>
__device__ __forceinline__ void layer16h_body(const LayerHArgs& a) {
    constexpr int A = BW * BW;
    constexpr int NXT = (BW + XT - 1) / XT;
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];
    const int xt = blockIdx.x % NXT;
    const int rest = blockIdx.x / NXT;
    const int grp = rest / a.nch, ch = rest - grp * a.nch;
    if (grp >= live_groups16(a.live, a.row_cap, grp + 1)) return;   // no live row in this group (rows handed out per simulation)
    const int base = BW / a.nch, extra = BW % a.nch;
    const int yb = ch * base + (ch < extra ? ch : extra);
    const int ye = yb + base + (ch < extra ? 1 : 0);
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    uint4* dst = a.dst + static_cast<size_t>(grp) * A * NC32 * 2 * 64;
    if (KIND == 2) {
        trunk_h_layer_tile<BW, XT, NC32, 1, 2, W16>(static_cast<const uint8_t*>(a.src) + static_cast<size_t>(grp) * 16 * kPlaneRow(BW), dst,
                                               a.layer, false, s_x, tile, lane, xt * XT, yb, ye);
    } else if (KIND == 1) {
        trunk_h_layer_tile<BW, XT, NC32, 1, 1, W16>(static_cast<const float4*>(a.src) + static_cast<size_t>(grp) * A * 8 * 16, dst, a.layer,
                                               false, s_x, tile, lane, xt * XT, yb, ye);
    } else {
        trunk_h_layer_tile<BW, XT, NC32, NC32, 0, W16>(static_cast<const uint4*>(a.src) + static_cast<size_t>(grp) * A * NC32 * 2 * 64, dst,
                                                      a.layer, a.res != 0, s_x, tile, lane, xt * XT, yb, ye);
    }
}

template <int BW, int XT, int NC32, int KIND>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_layer16h(LayerHArgs a) {
    layer16h_body<BW, XT, NC32, KIND, false>(a);
}
// the two-product form (see TrunkHLayerFn, W16): launched from net_w16.hip
template <int BW, int XT, int NC32, int KIND>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_layer16h_w16(LayerHArgs a) {
    layer16h_body<BW, XT, NC32, KIND, true>(a);
}

template <int BW, int NC32, int INK, int FMT, bool W16 = false>   // INK: 1 fp32 plane batch, 2 bit planes (see trunk_h_layer); FMT: see kPairBytes
__device__ __forceinline__ void trunk16h_body(const TrunkHArgs& a) {
    static_assert(FMT == 0 || AO_SPLIT_BARRIER == 0, "the split row barrier is implemented for the 4-byte format only");
    constexpr int A = BW * BW;
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];  // [2][row fragments][64] uint4
    const int grp = blockIdx.x;
    if (grp >= live_groups16(a.live, a.row_cap, grp + 1)) return;   // no live row in this group (rows handed out per simulation)
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);  // this wave's output tile
    // first activation byte of this group, in uint4 units (AO_KO 5 / 6: groups share buffers, timing experiment only)
    const size_t gq = static_cast<size_t>(AO_KO == 5 ? grp % 64 : AO_KO == 6 ? grp % 128 : grp) * A * NC32 * (kPairBytes(FMT) / 16);
    uint4* bufA = a.bufA + gq;
    uint4* bufB = a.bufB + gq;
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long* pp = prof;
    // arrival counter of the split row barrier: behind the two row buffers (kTrunkHCntOffset), monotonic over the launch
    unsigned* s_cnt = reinterpret_cast<unsigned*>(s_x + static_cast<size_t>(2) * BW * NC32 * 2 * 64);
    if (threadIdx.x == 0) *s_cnt = 0u;   // (published by the first barrier of conv1's prologue)
    const bool ko_fill = AO_KO == 13 || AO_KO == 15;
    // conv1: fp32 planes -> x
    AO_T(t0);
    if (INK == 2)
        TrunkHLayerFn<BW, NC32, 1, 2, FMT, W16>::run(reinterpret_cast<const uint8_t*>(a.in0) + static_cast<size_t>(grp) * 16 * kPlaneRow(BW), bufA,
                                      a.layers[0], false, s_x, tile, lane, pp, false, s_cnt, 0u);
    else
        TrunkHLayerFn<BW, NC32, 1, 1, FMT, W16>::run(a.in0 + static_cast<size_t>(grp) * A * 8 * 16, bufA, a.layers[0], false, s_x, tile, lane, pp, false,
                                      s_cnt, 0u);
    AO_T(t1);
    if (ko_fill) {   // data-like LDS rows for the trunk layers, never refreshed (see AO_KO)
        __syncthreads();
        for (int f = tile; f < 2 * BW * NC32 * 2; f += NC32 * 2) s_x[f * 64 + lane] = ao_ko_fragment((grp * 4u * BW * NC32 + f) * 64u + lane, f & 1);
        __syncthreads();
    }
#ifdef AO_PROF
    for (int k = 0; k < 12; ++k) prof[k] = 0;
#endif
    for (int l = 1; l < a.nlayers; ++l) {
        // l odd: first conv of a ResBlock (x -> t); l even: second conv (t -> x, + x in place)
        const bool second = (l & 1) == 0;
        TrunkHLayerFn<BW, NC32, NC32, 0, FMT, W16>::run(second ? bufB : bufA, second ? bufA : bufB, a.layers[l], second, s_x, tile, lane, pp,
                                             (l & 1) != 0, s_cnt, static_cast<unsigned>(l) * BW);
    }
    AO_T(t2);
    if (AO_KO != 8) trunk_heads<BW, FMT == 1 ? 2 : 1>(a, reinterpret_cast<const float4*>(a.bufA), static_cast<size_t>(grp) * A, grp);
#ifdef AO_PROF
    AO_T(t3);
    prof[5] = t3 - t2;
    prof[6] = t1 - t0;
    prof[7] = t3 - t0;
    if (grp == 5 && lane == 0)
        for (int k = 0; k < 12; ++k) ao_prof[tile * 12 + k] = prof[k];
#endif
}

// conv1 on the fp32 plane batch (ao_net_forward: any float planes) / on the engine's bit planes (ao_search)
template <int BW, int NC32, int FMT>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_trunk16h(TrunkHArgs a) {
    trunk16h_body<BW, NC32, 1, FMT>(a);
}
template <int BW, int NC32, int FMT>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_trunk16hb(TrunkHArgs a) {
    trunk16h_body<BW, NC32, 2, FMT>(a);
}
// the two-product forms (see TrunkHLayerFn, W16): launched from net_w16.hip
template <int BW, int NC32, int FMT>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_trunk16h_w16(TrunkHArgs a) {
    trunk16h_body<BW, NC32, 1, FMT, true>(a);
}
template <int BW, int NC32, int FMT>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_trunk16hb_w16(TrunkHArgs a) {
    trunk16h_body<BW, NC32, 2, FMT, true>(a);
}

}  // namespace ao
