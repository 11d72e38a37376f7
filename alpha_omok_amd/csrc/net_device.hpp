// net_device.hpp -- device code of the small-batch ("latency") network path, shared by the
// per-layer kernels of net.hip (k_conv_cells, k_heads_board). (The persistent single-game search
// kernel these were first shared with was measured slower and is not in the tree: DESIGN.md section 4.)
//
// Layout: per-board NHWC, act[board][cell][channel] as float4 channel quads. The CELLS of one
// board are the MFMA N dimension:   D[cout 16][cell 16] += Wt[cout 16][k 4] * X[k 4][cell 16]
// (model.py:6-31: bias-free 3x3 conv, BatchNorm folded to scale/shift, residual, ReLU).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "engine_types.hpp"

namespace ao {

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ float block_reduce(float v, float* s_red, bool is_max) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    __syncthreads();
    if (lane == 0) s_red[wid] = v;
    __syncthreads();
    float r = s_red[0];
    for (int i = 1; i < static_cast<int>(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, s_red[i]) : r + s_red[i];
    return r;
}

#ifdef AO_PROF
static __device__ unsigned long long ao_prof_conv[8];    // k_conv_cells, block 0: wave 0 start / operands landed / MFMAs done / reduced / stored
#define AO_CT(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) ao_prof_conv[k] = __builtin_amdgcn_s_memtime(); } while (0)
static __device__ unsigned long long ao_prof_heads[8];   // phase ends of k_heads_board (thread 0 of board 0), shader-clock ticks
#define AO_HT(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) ao_prof_heads[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AO_HT(k)
#define AO_CT(k)
#endif
// One (16 cells x 16 output channels) tile of one board's 3x3 conv, computed by NW waves:
// each takes 9/NW taps (that share of the K loop), the partial tiles are summed through LDS.
// A single wave per tile is bound by its own in-order chain of 144 loads; nine waves cut that
// chain to 16 (best for one board), three to 48 (best for a few dozen).
// Out-of-board taps are zero-filled per lane (cells of a tile differ in position).
// All NW waves of the workgroup must call; s_red holds (NW-1) x 64 x 4 floats.
// board rows a 16-cell tile's 3x3 neighbourhood can span, and the float4 slots of the LDS copy of those rows (one quad of
// padding per cell: consecutive cells then start 16 bytes apart in the 256-byte bank window, so the 16 cells x 4 quads a
// wave reads at once are conflict-free)
__host__ __device__ constexpr int conv_cells_rows(int bw) {
    return ((16 + bw - 2) / bw + 1 + 2) < bw ? ((16 + bw - 2) / bw + 1 + 2) : bw;
}
__host__ __device__ constexpr int conv_cells_lds_quads(int bw, int ncqg) { return conv_cells_rows(bw) * bw * (ncqg * 4 + 1); }

// LDSX: the tile's board rows are copied to LDS once and the taps read their shifted views there (else every tap loads
// its view from global memory: better when many small workgroups share a CU and LDS would limit them).
typedef _Float16 cc_half8 __attribute__((ext_vector_type(8)));
template <int BW, int NCQG, int NW, bool LDSX = true>
__device__ __forceinline__ void conv_cells_tile(const float4* __restrict__ in, const float4* __restrict__ wt,
                                                const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                const float4* res, float4* out, int CQI, int COUT, int relu_res,
                                                int ct, int ctile, int board, float* s_red, float4* s_x) {
    constexpr int A = BW * BW;
    constexpr int TP = 9 / NW;   // whole taps per wave
    constexpr int QS = NCQG * 4 + 1;   // padded quads per cell in s_x
    const int lane = threadIdx.x & 63;
    const int w3 = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int kq = lane >> 4, ci = lane & 15;
    const int cell = ctile * 16 + ci;
    const int cy = cell / BW, cx = cell - cy * BW;
    const float4* xb = in + static_cast<size_t>(board) * A * CQI;
    // four independent accumulator chains (one chain would pay the 40-cycle dependent-MFMA latency
    // on every instruction)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    // what the epilogue needs besides the accumulators depends on indices only: requested now, behind the operand loads,
    // instead of as a dependent L2 round trip after the reduction (measured: 5.16 -> 5.10 us per launch, r3h)
    const int cqo = ct * 4 + kq;
    const size_t o_idx = (static_cast<size_t>(board) * A + (cell < A ? cell : 0)) * (COUT >> 2) + cqo;
    float4 e_sc = make_float4(0.f, 0.f, 0.f, 0.f), e_sh = e_sc, e_res = e_sc;
    if (w3 == 0) {
        e_sc = scale[cqo];
        e_sh = shift[cqo];
        if (relu_res && cell < A) e_res = res[o_idx];
    }
    float4 rx[TP][NCQG], rw[TP][NCQG];
    // The nine taps of a tile read shifted views of the SAME few board rows. Loaded per tap from global memory that is
    // 8 KB per wave and 72 KB per workgroup, as much as the weights, and the workgroup's 144 KB of operands through one
    // CU's ~64 B/clk was most of the launch (AO_PROF: operands 3.1 k cycles for the first wave, the ninth another 2 k
    // later, of 9 k in the kernel). The rows are now copied to LDS once (9x9, 128 planes: <= 45 cells = 23 KB) and the
    // taps read their views there; only the weights still stream from L2.
    const int c_lo = ctile * 16, c_hi = (c_lo + 15 < A ? c_lo + 15 : A - 1);
    const int r0 = c_lo / BW > 0 ? c_lo / BW - 1 : 0;
    const int r1 = c_hi / BW + 1 < BW ? c_hi / BW + 1 : BW - 1;
    auto load_w = [&](int tap, float4 (&W)[NCQG]) {
        const float4* wp = wt + (static_cast<size_t>(tap) * CQI + kq) * COUT + ct * 16 + ci;
#pragma unroll
        for (int cqg = 0; cqg < NCQG; ++cqg) W[cqg] = wp[static_cast<size_t>(cqg) * 4 * COUT];
    };
    auto load_x = [&](int tap, float4 (&X)[NCQG]) {
        const int yy = cy + tap / 3 - 1, xx = cx + tap % 3 - 1;
        const bool ok = cell < A && yy >= 0 && yy < BW && xx >= 0 && xx < BW;
        const float4* xp = LDSX ? s_x + (ok ? (yy - r0) * BW + xx : 0) * QS + kq
                                : xb + static_cast<size_t>(ok ? yy * BW + xx : 0) * CQI + kq;
#pragma unroll
        for (int cqg = 0; cqg < NCQG; ++cqg) {
            float4 x = xp[cqg * 4];
            if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
            X[cqg] = x;
        }
    };
    auto compute_tap = [&](const float4 (&X)[NCQG], const float4 (&W)[NCQG]) {
#pragma unroll
        for (int cqg = 0; cqg < NCQG; ++cqg) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[cqg].x, X[cqg].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[cqg].y, X[cqg].y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[cqg].z, X[cqg].z, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[cqg].w, X[cqg].w, acc3, 0, 0, 0);
        }
    };
    AO_CT(0);
    // weights first (they are not touched until the MFMAs), then this thread's share of the rows
#pragma unroll
    for (int j = 0; j < TP; ++j) load_w(TP * w3 + j, rw[j]);
    if (LDSX) {
        const int nq = (r1 - r0 + 1) * BW * (NCQG * 4);
        const float4* src = xb + static_cast<size_t>(r0) * BW * CQI;
        for (int i = threadIdx.x; i < nq; i += NW * 64) {
            const int c = i / (NCQG * 4), q = i - c * (NCQG * 4);
            s_x[c * QS + q] = src[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TP; ++j) load_x(TP * w3 + j, rx[j]);
#ifdef AO_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    AO_CT(1);
#pragma unroll
    for (int j = 0; j < TP; ++j) compute_tap(rx[j], rw[j]);
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
    if (w3 > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[((w3 - 1) * 64 + lane) * 4 + r] = acc[r];
    }
    AO_CT(2);
    __syncthreads();
    AO_CT(3);
    if (w3 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = acc[r];
#pragma unroll
            for (int k = 0; k < NW - 1; ++k) t += s_red[(k * 64 + lane) * 4 + r];
            acc[r] = t;
        }
        // D row = cout 4*kq + reg, col = cell ci
        if (cell < A) {
            const float4 sc = e_sc, sh = e_sh;
            const size_t o = o_idx;
            float4 v;
            v.x = fmaf(acc[0], sc.x, sh.x); v.y = fmaf(acc[1], sc.y, sh.y);
            v.z = fmaf(acc[2], sc.z, sh.z); v.w = fmaf(acc[3], sc.w, sh.w);
            if (relu_res) {
                const float4 rr = e_res;
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            out[o] = v;
        }
    }
#ifdef AO_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    AO_CT(4);
}

// The tile on v_mfma_f32_16x16x32_f16 with split operands (x*w = xh*wh + xh*wl + xl*wh, fp32 accumulate, as k_trunk16h):
// weights come pre-split (wh / wl: [tap][32-channel block][cout tile][lane] x 8 fp16, pre-scaled, `scale` = the matching
// BatchNorm scale), the activations are split while they are read from LDS. A work unit = (tap, 32-channel block): 9 x NC32
// of them, NUH per wave (128 planes: 36 units on 12 waves, three MFMA triples each and three waves per SIMD -- with one tap
// per wave the SIMD that got three of the nine waves finished 1.5 k cycles after the others).
// It serves SEVERAL boards in turn (k_conv_cells_h): a workgroup owns one (16 cells x 16 channels) tile position
// and walks `nb` boards with it. The wave's weights -- three (tap, 32-channel block) units, high and low halves, 24
// registers -- are loaded ONCE; per board only the tile's board rows (23 KB at 9x9) come through the CU. One workgroup per
// board and tile re-pulled the 72 KB of weights for every board: fine for a handful of games, L2-bound for a hundred.
// s_red holds two buffers of (NW - 1) x 64 x 4 floats (alternating per board, so a wave may write the next board's partial
// sums while wave 0 still adds up this board's).
// W16: the conv weights are fp16 numbers (zero low halves, net_trunk_h16.hpp): no xh*wl product and -- what matters on this latency path -- half
// the weight bytes through the CU.
template <int BW, int NCQG, int NW, bool W16 = false>
__device__ __forceinline__ void conv_cells_tile_h(const float4* __restrict__ in, const uint4* __restrict__ wh, const uint4* __restrict__ wl,
                                                  const float4* __restrict__ scale, const float4* __restrict__ shift, const float4* res,
                                                  float4* out, int CQI, int COUT, int relu_res, int ct, int ctile, int board0, int nb,
                                                  float* s_red, float4* s_x, int* ovf) {
    constexpr int A = BW * BW;
    constexpr int QS = NCQG * 4 + 1;
    constexpr int NC32 = NCQG / 2;
    constexpr int NUH = (9 * NC32) / NW;
    static_assert(NUH * NW == 9 * NC32 && NCQG % 2 == 0, "the waves must divide the (tap, 32-channel block) units evenly");
    const int lane = threadIdx.x & 63;
    const int w3 = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int kq = lane >> 4, ci = lane & 15;
    const int cell = ctile * 16 + ci;
    const int cy = cell / BW, cx = cell - cy * BW;
    const int c_lo = ctile * 16, c_hi = (c_lo + 15 < A ? c_lo + 15 : A - 1);
    const int r0 = c_lo / BW > 0 ? c_lo / BW - 1 : 0;
    const int r1 = c_hi / BW + 1 < BW ? c_hi / BW + 1 : BW - 1;
    const int cqo = ct * 4 + kq;
    AO_CT(0);
    // once per workgroup: this wave's weights and the epilogue's scale / shift
    cc_half8 whr[NUH], wlr[NUH];
    {
        const int nt = COUT >> 4;
#pragma unroll
        for (int j = 0; j < NUH; ++j) {
            const size_t idx = (static_cast<size_t>(NUH * w3 + j) * nt + ct) * 64 + lane;   // unit = tap * NC32 + block
            whr[j] = __builtin_bit_cast(cc_half8, wh[idx]);
            if (!W16) wlr[j] = __builtin_bit_cast(cc_half8, wl[idx]);
        }
    }
    float4 e_sc = make_float4(0.f, 0.f, 0.f, 0.f), e_sh = e_sc;
    if (w3 == 0) { e_sc = scale[cqo]; e_sh = shift[cqo]; }
    // per unit: where this lane's B operand sits in the LDS copy of the rows (the same for every board)
    int xoff[NUH];
    bool xok[NUH];
#pragma unroll
    for (int j = 0; j < NUH; ++j) {
        const int u = NUH * w3 + j, tap = u / NC32, c = u - tap * NC32;
        const int yy = cy + tap / 3 - 1, xx = cx + tap % 3 - 1;
        xok[j] = cell < A && yy >= 0 && yy < BW && xx >= 0 && xx < BW;
        xoff[j] = (xok[j] ? (yy - r0) * BW + xx : 0) * QS + c * 8 + kq * 2;
    }
    const int nq = (r1 - r0 + 1) * BW * (NCQG * 2);   // staging items: (cell, 8-channel group)
    uint4* s_xh = reinterpret_cast<uint4*>(s_x);
    float peak = 0.f;
    for (int bi = 0; bi < nb; ++bi) {
        const int board = board0 + bi;
        const size_t o_idx = (static_cast<size_t>(board) * A + (cell < A ? cell : 0)) * (COUT >> 2) + cqo;
        float4 e_res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w3 == 0 && relu_res && cell < A) e_res = res[o_idx];
        // the tile's board rows of this board into LDS (every wave has finished reading the previous board's: they all
        // passed the reduction barrier below after their LDS reads) -- split into fp16 high / low halves HERE, once per
        // element: 8 consecutive channels per thread, stored as the 16-byte operand groups the MFMAs read. (Split by the
        // consuming waves, every element was converted up to nine times -- once per tap -- and the ~150 vector instructions
        // per lane sat between the operands' arrival and the first MFMA.)
        {
            const float4* src = in + (static_cast<size_t>(board) * A + static_cast<size_t>(r0) * BW) * CQI;
            for (int i = threadIdx.x; i < nq; i += NW * 64) {
                const int c = i / (NCQG * 2), o = i - c * (NCQG * 2);   // cell of the staged rows, group of 8 channels
                const float4 q0 = src[c * (NCQG * 4) + 2 * o], q1 = src[c * (NCQG * 4) + 2 * o + 1];
                const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                cc_half8 xh, xl;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    peak = fmaxf(peak, v[k]);
                    const float t = fminf(v[k], 65504.f);   // (inputs are post-ReLU; beyond the fp16 range: clamped and reported)
                    const _Float16 hh = static_cast<_Float16>(t);
                    xh[k] = hh;
                    xl[k] = static_cast<_Float16>(t - static_cast<float>(hh));
                }
                uint4* dst = s_xh + c * QS + o * 2;   // [cell][32-channel block][8-channel group][high | low]
                dst[0] = __builtin_bit_cast(uint4, xh);
                dst[1] = __builtin_bit_cast(uint4, xl);
            }
        }
        AO_CT(1);
        __syncthreads();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma unroll
        for (int j = 0; j < NUH; ++j) {
            uint4 uh = s_xh[xoff[j]], ul = s_xh[xoff[j] + 1];
            if (!xok[j]) { uh = make_uint4(0u, 0u, 0u, 0u); ul = uh; }
            const cc_half8 xh = __builtin_bit_cast(cc_half8, uh), xl = __builtin_bit_cast(cc_half8, ul);
            // one accumulator chain per unit (up to four); hh, hl, lh of a unit go to the same chain
            f32x4& a = (j & 3) == 0 ? acc0 : (j & 3) == 1 ? acc1 : (j & 3) == 2 ? acc2 : acc3;
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(whr[j], xh, a, 0, 0, 0);
            if (!W16) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlr[j], xh, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(whr[j], xl, a, 0, 0, 0);
        }
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
        AO_CT(2);
        float* red = s_red + (bi & 1) * (NW - 1) * 64 * 4;
        if (w3 > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((w3 - 1) * 64 + lane) * 4 + r] = acc[r];
        }
        __syncthreads();
        AO_CT(3);
        if (w3 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[r];
#pragma unroll
                for (int k = 0; k < NW - 1; ++k) t += red[(k * 64 + lane) * 4 + r];
                acc[r] = t;
            }
            if (cell < A) {   // D row = cout 4*kq + reg, col = cell ci
                float4 v;
                v.x = fmaf(acc[0], e_sc.x, e_sh.x); v.y = fmaf(acc[1], e_sc.y, e_sh.y);
                v.z = fmaf(acc[2], e_sc.z, e_sh.z); v.w = fmaf(acc[3], e_sc.w, e_sh.w);
                if (relu_res) { v.x += e_res.x; v.y += e_res.y; v.z += e_res.z; v.w += e_res.w; }
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                out[o_idx] = v;
            }
        }
    }
    if (peak > 65504.f && ovf) atomicOr(ovf, 1);
    AO_CT(4);
}

// LDS floats heads_board_dev needs
__host__ __device__ inline size_t heads_lds_floats(int A, int planes) {
    return static_cast<size_t>(3) * planes + 3 * A + 36 * A + 6 * A + 4 * planes + planes + 16;
}

// Policy and value heads of ONE board (model.py:34-73) by one workgroup of any size >= 64:
// 1x1 convs + BN + ReLU -> flatten in NCHW order (c*A + cell, the reference's .view) ->
// policy_fc + softmax, value_fc1 + ReLU + value_fc2 + tanh. Every dot product is cut into
// slices so that all threads carry a short, independent chain of loads (the weight matrices come
// from L2: 52 KB + 41 KB at 9x9), partial sums meet in LDS.
// act = this board's activations [A][planes/4]; policy -> [A], value -> [1]. All threads must call.

__device__ __forceinline__ void heads_board_dev(const HeadParams& h, const float4* __restrict__ act,
                                                float* __restrict__ policy, float* __restrict__ value, int A,
                                                int planes, float* s_mem) {
    constexpr int KS = 12;  // K slices of the 1x1 convs: a work item = (slice, cell) and forms all three head channels from
                            // its activation quads (9x9: 972 items for 1024 threads, 3 quads each -- one round)
    constexpr int NPS = 6;  // slices of the 2A-long policy_fc dot products (9x9: 486 + 512 FC items <= 1024 threads: one round of
                            // 27 / 21 weights per thread; a second round of a few items keeps the whole block waiting)
    constexpr int NVS = 4;  // slices of the A-long value_fc1 dot products
    float* s_w3 = s_mem;                 // [3*planes]
    float* s_h = s_w3 + 3 * planes;      // [3A]
    float* s_hp = s_h + 3 * A;           // [KS][3A]
    float* s_part = s_hp + KS * 3 * A;   // [NPS][A]
    float* s_vpart = s_part + NPS * A;   // [NVS][planes]
    float* s_hid = s_vpart + NVS * planes;  // [planes]
    (void)(s_hid + planes);              // [16] spare
    const int tid = threadIdx.x, nt = blockDim.x;
    const int CQ = planes >> 2;
    AO_HT(0);
    for (int i = tid; i < 3 * planes; i += nt) s_w3[i] = h.w3[i];
    __syncthreads();
    AO_HT(1);
    // the per-output constants of the later phases are requested now: each was a dependent L2 round trip in its phase
    const float bp_r = tid < A ? h.bp[tid] : 0.f;
    const float b2_r = h.b2[0];
    // 1x1 convs: (K slice, cell) per thread, the three output channels together: a quad is loaded once and its twelve
    // weights come as three 16-byte LDS reads (was: one item per channel, 64 four-byte LDS reads per item)
    const int cqs = (CQ + KS - 1) / KS;
    for (int i = tid; i < KS * A; i += nt) {
        const int ks = i / A, cell = i - ks * A;
        const int q0 = ks * cqs, q1 = min(CQ, q0 + cqs);
        const float4* xp = act + static_cast<size_t>(cell) * CQ;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        constexpr int XB = 3;
        for (int qb = q0; qb < q1; qb += XB) {
            float4 xr[XB];
#pragma unroll
            for (int k = 0; k < XB; ++k) xr[k] = (qb + k < q1) ? xp[qb + k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < XB; ++k) {
                if (qb + k < q1) {
                    const float4 x = xr[k];
                    const float4 w0 = *reinterpret_cast<const float4*>(s_w3 + 4 * (qb + k));
                    const float4 w1 = *reinterpret_cast<const float4*>(s_w3 + planes + 4 * (qb + k));
                    const float4 w2 = *reinterpret_cast<const float4*>(s_w3 + 2 * planes + 4 * (qb + k));
                    a0 = fmaf(x.x, w0.x, a0); a0 = fmaf(x.y, w0.y, a0); a0 = fmaf(x.z, w0.z, a0); a0 = fmaf(x.w, w0.w, a0);
                    a1 = fmaf(x.x, w1.x, a1); a1 = fmaf(x.y, w1.y, a1); a1 = fmaf(x.z, w1.z, a1); a1 = fmaf(x.w, w1.w, a1);
                    a2 = fmaf(x.x, w2.x, a2); a2 = fmaf(x.y, w2.y, a2); a2 = fmaf(x.z, w2.z, a2); a2 = fmaf(x.w, w2.w, a2);
                }
            }
        }
        s_hp[ks * 3 * A + cell] = a0;
        s_hp[ks * 3 * A + A + cell] = a1;
        s_hp[ks * 3 * A + 2 * A + cell] = a2;
    }
    __syncthreads();
    AO_HT(2);
    for (int r = tid; r < 3 * A; r += nt) {
        const int c = r / A;
        float t = s_hp[r];
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) t += s_hp[ks * 3 * A + r];
        s_h[r] = fmaxf(fmaf(t, h.sc3[c], h.sh3[c]), 0.f);
    }
    __syncthreads();
    AO_HT(3);
    // policy_fc (NPS slices per output) and value_fc1 (NVS slices per hidden unit) in one sweep
    const int np_items = NPS * A, nv_items = NVS * planes;
    const int pslice = (2 * A + NPS - 1) / NPS, vslice = (A + NVS - 1) / NVS;
    for (int i = tid; i < np_items + nv_items; i += nt) {
        float acc = 0.f;
        if (i < np_items) {
            const int part = i / A, a = i - part * A;
            const int j0 = part * pslice, j1 = min(2 * A, j0 + pslice);
            const float* wcol = h.wp_t + a;
#pragma unroll 8
            for (int j = j0; j < j1; ++j) acc = fmaf(wcol[static_cast<size_t>(j) * A], s_h[j], acc);
            s_part[i] = acc;
        } else {
            const int k = i - np_items;
            const int part = k / planes, o = k - part * planes;
            const int j0 = part * vslice, j1 = min(A, j0 + vslice);
            const float* wcol = h.w1_t + o;
#pragma unroll 8
            for (int j = j0; j < j1; ++j) acc = fmaf(wcol[static_cast<size_t>(j) * planes], s_h[2 * A + j], acc);
            s_vpart[k] = acc;
        }
    }
    __syncthreads();
    AO_HT(4);
    // softmax and the value head's tail by ONE wave each, on wave shuffles only: as block-wide reductions these were three
    // rounds of two barriers with six idle waves (6 k of the kernel's 25 k cycles)
    const int wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {
        constexpr int NL = 4;                     // logits per lane: A <= 256
        float lg[NL];
        float lmax = -3.0e38f;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int a = lane + 64 * k;
            lg[k] = -3.0e38f;
            if (a < A) {
                float l = (k == 0) ? bp_r : h.bp[a];
#pragma unroll
                for (int q = 0; q < NPS; ++q) l += s_part[q * A + a];
                lg[k] = l;
                lmax = fmaxf(lmax, l);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        float lsum = 0.f;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int a = lane + 64 * k;
            lg[k] = a < A ? expf(lg[k] - lmax) : 0.f;
            lsum += lg[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int a = lane + 64 * k;
            if (a < A) policy[a] = lg[k] / lsum;
        }
    }
    if (wave == (nt > 64 ? 1 : 0)) {
        float part = 0.f;
        for (int o = lane; o < planes; o += 64) {
            float t = h.b1[o];
#pragma unroll
            for (int q = 0; q < NVS; ++q) t += s_vpart[q * planes + o];
            part = fmaf(h.w2[o], fmaxf(t, 0.f), part);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) value[0] = tanhf(part + b2_r);
    }
    AO_HT(5);
    AO_HT(6);
}

}  // namespace ao
