// net.hip -- policy/value ResNet forward (model.py:76-104 PVNet, eval mode) for gfx950.
//
// Layout. Activations live in HBM as   act[grp][cell][cq][b][4]   (float32), where a group is
// 32 boards (b), cq = channel/4. One (cell, cq) slab is 32 boards x 16 B = 512 contiguous
// bytes, which is exactly one half-wave's B-operand fragment of v_mfma_f32_32x32x2_f32 with the
// BOARDS as the MFMA N dimension:
//     D[cout 32][board 32] += Wt[cout 32][k 2] * X[k 2][board 32]
// A lane loads 16 B = 4 consecutive input channels of its board (lanes 0-31: quad cq0, lanes
// 32-63: quad cq0+1) and issues 4 MFMAs, MFMA t consuming the k-pair {4*cq0+t, 4*cq0+4+t}.
// Weights are repacked to wt[tap][cq][cout][4] so the A fragment is the same 16-B-per-lane,
// 512-B-contiguous load. Every fragment load and every output store is a full-line coalesced
// dwordx4 access; no LDS and no im2col buffer are needed (the "im2col" is the tap loop).
//
// Because the 32 rows of an MFMA tile are 32 different boards at the SAME cell, a tap that
// falls outside the board is outside for the whole tile and is skipped: 625 of the 729
// (cell, tap) pairs of a 9x9 board do work, the zero padding costs nothing.
//
// One workgroup = one board row of one group (BW output cells), all output channels:
// wave w owns output-channel tile w (32 couts) and keeps BW accumulator tiles (16 VGPR each).
// The epilogue fuses BatchNorm (running stats folded to scale/shift), the residual add and ReLU.
//
// The 3x3 stack is >99.9 % of the FLOPs; the heads (1x1 convs, FCs, softmax, tanh) are small
// VALU kernels on the same layout.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/omok_hip.h"
#include "engine_types.hpp"
#include "net_device.hpp"

namespace ao {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct Frag {
    float v[4];
};

__device__ __forceinline__ Frag ld_frag(const float4* p) {
    const float4 t = *p;
    Frag f;
    f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
    return f;
}

// Buffer-descriptor loads/stores: address = descriptor base (SGPRs) + per-lane 32-bit voffset +
// wave-uniform soffset (an SGPR). A step's dozens of fragment loads then share ONE address VGPR.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}

__device__ __forceinline__ Frag buf_ld_frag(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    Frag f;
    f.v[0] = __uint_as_float(t.x); f.v[1] = __uint_as_float(t.y);
    f.v[2] = __uint_as_float(t.z); f.v[3] = __uint_as_float(t.w);
    return f;
}

// XCD-aware block id remap: consecutive virtual ids (rows of one group, neighbouring groups)
// run on one XCD and share its L2 (blocks are dispatched round-robin over the 8 XCDs).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int NX>
struct StepRegs {
    Frag x[NX];
    Frag w[3];
};

// 3x3 convolution, padding 1, no bias (model.py:6-10) + folded BatchNorm + optional residual
// + ReLU. in: [grp][A][CQI][32] float4, wt: [9][CQI][COUT] float4, out/res: [grp][A][COUT/4][32].
// A workgroup computes XT consecutive cells of one board row (XT == BW: the whole row).
template <int BW, int XT, bool RES>
__global__ __launch_bounds__(256) void k_conv3x3(const float4* __restrict__ in,
                                                 const float4* __restrict__ wt,
                                                 const float4* __restrict__ scale,
                                                 const float4* __restrict__ shift,
                                                 const float4* res, float4* out, int CQI, int COUT,
                                                 int nblk) {
    constexpr int A = BW * BW;
    constexpr int NXT = (BW + XT - 1) / XT;
    constexpr int NX = XT + 2;
    const int vid = xcd_remap(blockIdx.x, nblk);
    const int grp = vid / (BW * NXT);
    const int rem = vid - grp * (BW * NXT);
    const int y = rem / NXT;
    const int x0 = (NXT == 1) ? 0 : (rem - y * NXT) * XT;
    const int lane = threadIdx.x & 63;
    const int ct = threadIdx.x >> 6;  // output-channel tile of this wave
    const int half = lane >> 5;
    const int b = lane & 31;
    const int rlo = (y == 0) ? 1 : 0;
    const int rhi = (y == BW - 1) ? 1 : 2;
    const int nrows = rhi - rlo + 1;
    const int nsteps = (CQI >> 1) * nrows;

    f32x16 acc[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

    const size_t in_grp = static_cast<size_t>(grp) * A;

    auto load = [&](int s, StepRegs<NX>& R) {
        const int cqp = s / nrows;
        const int r = rlo + (s - cqp * nrows);
        const int yy = y - 1 + r;
        const int cq = cqp * 2 + half;
        const float4* xp = in + ((in_grp + static_cast<size_t>(yy) * BW) * CQI + cq) * kGroup + b;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int xi = x0 - 1 + j;
            if (xi >= 0 && xi < BW) R.x[j] = ld_frag(xp + static_cast<size_t>(xi) * CQI * kGroup);
        }
        const float4* wp = wt + (static_cast<size_t>(r * 3) * CQI + cq) * COUT + ct * 32 + b;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) R.w[dx] = ld_frag(wp + static_cast<size_t>(dx) * CQI * COUT);
    };
    auto compute = [&](const StepRegs<NX>& R) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
            for (int i = 0; i < XT; ++i) {
                const int xo = x0 + i;
                const int xi = xo + dx - 1;
                if (xo >= BW || xi < 0 || xi >= BW) continue;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.w[dx].v[t], R.x[i + dx].v[t], acc[i], 0, 0, 0);
            }
        }
    };

    StepRegs<NX> Ra, Rb;
    load(0, Ra);
    for (int s = 0; s < nsteps; s += 2) {
        if (s + 1 < nsteps) load(s + 1, Rb);
        compute(Ra);
        if (s + 2 < nsteps) load(s + 2, Ra);
        if (s + 1 < nsteps) compute(Rb);
    }

    // epilogue: D row = cout (reg&3) + 8*(reg>>2) + 4*half, col = board b
    const int CQO = COUT >> 2;
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        const int xo = x0 + i;
        if (xo >= BW) continue;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int cqo = ct * 8 + 2 * rg + half;
            const float4 sc = scale[cqo];
            const float4 sh = shift[cqo];
            const size_t o = ((in_grp + static_cast<size_t>(y) * BW + xo) * CQO + cqo) * kGroup + b;
            float4 v;
            v.x = fmaf(acc[i][4 * rg + 0], sc.x, sh.x);
            v.y = fmaf(acc[i][4 * rg + 1], sc.y, sh.y);
            v.z = fmaf(acc[i][4 * rg + 2], sc.z, sh.z);
            v.w = fmaf(acc[i][4 * rg + 3], sc.w, sh.w);
            if (RES) {
                const float4 rr = res[o];
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            out[o] = v;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Group-resident trunk: ONE workgroup carries a group of 16 boards through conv1 and all
// residual blocks. G = 4096 games = 256 groups = one workgroup per CU of an MI355X, every CU does
// identical work (no tail between layers, one launch instead of 1+2*n_block), and consecutive
// layers need no device-wide synchronisation because a layer of a group only depends on the
// previous layer of the same group: a workgroup barrier + an L1 invalidate is enough.
//
// MFMA shape: v_mfma_f32_16x16x4_f32, D[cout 16][board 16] += Wt[cout 16][k 4] * X[k 4][board 16].
// Lane l loads 16 B = 4 input channels of channel quad cq0 + (l>>4) for board (B operand) or
// output channel (A operand) l&15: a wave-wide fragment load is 1 KiB contiguous in
// act[grp][cell][cq][16][4]. Wave w owns output-channel tile w (16 couts) and walks the board row
// by row with BW accumulators (4 VGPR each); 8 waves = 128 output channels, 2 waves per SIMD.
// ----------------------------------------------------------------------------------------------

struct TrunkLayer {
    const float4* w;   // [9][cqi][COUT] float4
    const float4* sc;  // [COUT/4]
    const float4* sh;
};

constexpr int kMaxTrunkLayers = 44;

struct TrunkArgs {
    const float4* in0;  // [grp][A][cq0][16]
    float4* bufA;       // [grp][A][CQ][16]
    float4* bufB;
    int nlayers, cq0, CQ, COUT;
    int cq0_real;       // channel quads of the input that are not padding
    // heads (model.py:34-73), run by the same workgroup once its trunk is done
    const float *w3, *sc3, *sh3, *wp_t, *bp, *w1_t, *b1, *w2, *b2;
    float* policy;      // [boards][A]
    float* value;       // [boards]
    TrunkLayer layers[kMaxTrunkLayers];
};

// One conv layer of one group, "sliding window" form. A wave owns TPW output-channel tiles and
// keeps the accumulators of THREE output rows (3 x XT cells x TPW tiles, AGPRs). A step = one
// input row yi x 16 input channels: its XT(+2) activation fragments and the 9 x TPW weight
// fragments feed every (dy, dx) tap at once -- up to 75 x 4 x TPW MFMAs -- so each activation is
// loaded exactly once per layer and wave (not once per output row) and a step of loads is covered
// by ~10-20k cycles of matrix work. Activations are double-buffered one step ahead; the weight
// fragments of tap row dy are re-loaded for the next step right after their last MFMA.
// When input row yi is done, output row yi-1 is complete: its epilogue (BN scale/shift, residual,
// ReLU, store) runs and the window slides (accumulator registers move down one row).
template <int BW, int XT, int TPW>
__device__ __forceinline__ void trunk_layer(const float4* __restrict__ src, float4* dst,
                                            const float4* __restrict__ wt, const float4* __restrict__ scp,
                                            const float4* __restrict__ shp, const bool RES, int cqi, int cq_real,
                                            int COUT, size_t gbase, int ct0, int kq, int b, int yb, int ye) {
    // computes the output rows [yb, ye) of the layer (the whole board inside the resident kernel, a
    // row chunk when one launch per layer spreads a group over several workgroups)
    constexpr int NXT = (BW + XT - 1) / XT;
    constexpr int NX = XT + 2;
    constexpr int GB = 16;
    const int CQO = COUT >> 2;
    const int ncqg = cqi >> 2;  // even (asserted on the host)

    float4 sc[TPW], sh[TPW];
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        sc[tl] = scp[(ct0 + tl) * 4 + kq];
        sh[tl] = shp[(ct0 + tl) * 4 + kq];
    }

    for (int xt = 0; xt < NXT; ++xt) {
        const int x0 = (NXT == 1) ? 0 : xt * XT;
        f32x4 acc0[XT][TPW], acc1[XT][TPW], acc2[XT][TPW];  // output rows yi-1, yi, yi+1
#pragma unroll
        for (int i = 0; i < XT; ++i)
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl) {
                acc0[i][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc1[i][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc2[i][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        Frag xa[NX], xb[NX], w[3][3][TPW];

        // Addresses are "buffer descriptor + one per-lane 32-bit offset + uniform SGPR offset" so
        // the dozens of fragment loads of a step cost scalar, not vector, address arithmetic.
        // Every load is unconditional (cells outside the board are clamped, their MFMAs skipped).
        const int lane_x = (kq * GB + b) * 16;    // bytes inside one (cell, 4-quad) slab
        const int lane_w = (kq * COUT + b) * 16;  // bytes inside one (tap, 4-quad) weight slab
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(src + gbase * cqi * GB, BW * BW * cqi * GB * 16u);
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wt, 9u * cqi * COUT * 16u);
        const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(dst + gbase * CQO * GB, BW * BW * CQO * GB * 16u);
        auto load_x = [&](int yi, int cqg, Frag (&X)[NX]) {
            const int row = (yi * BW * cqi + cqg * 4) * GB * 16;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                if (NXT == 1 && (j == 0 || j == NX - 1)) continue;  // statically outside the board
                int xi = x0 - 1 + j;
                xi = xi < 0 ? 0 : (xi >= BW ? BW - 1 : xi);
                X[j] = buf_ld_frag(rs_x, lane_x, row + xi * cqi * GB * 16);
            }
        };
        auto load_w = [&](int cqg, int dy) {
            const int row = ((dy * 3 * cqi + cqg * 4) * COUT + ct0 * 16) * 16;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl)
                    w[dy][dx][tl] = buf_ld_frag(rs_w, lane_w, row + (dx * cqi * COUT + tl * 16) * 16);
        };
        auto taps = [&](const Frag (&X)[NX], int dy, f32x4 (&acc)[XT][TPW]) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {  // consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl) {
#pragma unroll
                        for (int i = 0; i < XT; ++i) {
                            const int xo = x0 + i;
                            const int xi = xo + dx - 1;
                            if (xo >= BW || xi < 0 || xi >= BW) continue;
                            acc[i][tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[dy][dx][tl].v[t], X[i + dx].v[t],
                                                                              acc[i][tl], 0, 0, 0);
                        }
                    }
                }
            }
        };
        // one step: input row yi, channel group of X; (nyi, ncq) is the step after it
        // `live` is false for a k-step whose 16 input channels are all padding (conv1: 5 planes in
        // a 32-channel slab): its loads are issued to keep the stream uniform, its MFMAs are not.
        auto step = [&](const Frag (&X)[NX], Frag (&Xn)[NX], int yi, int nyi, int ncq, bool live) {
            // sched_barrier keeps each weight re-load BELOW the last MFMA that reads the registers it
            // overwrites; hoisted above, it would need a second copy of the weight fragments
            load_x(nyi, ncq, Xn);
            if (live && yi + 1 < ye) taps(X, 0, acc2);            // dy = 0 -> output row yi + 1
            __builtin_amdgcn_sched_barrier(0);
            load_w(ncq, 0);
            if (live && yi >= yb && yi < ye) taps(X, 1, acc1);    // dy = 1 -> output row yi
            __builtin_amdgcn_sched_barrier(0);
            load_w(ncq, 1);
            if (live && yi - 1 >= yb) taps(X, 2, acc0);           // dy = 2 -> output row yi - 1
            __builtin_amdgcn_sched_barrier(0);
            load_w(ncq, 2);
        };
        // D row = cout 4*kq + reg, col = board b -> one float4 of 4 couts per lane
        auto epilogue = [&](int yo) {
            Frag rr[XT][TPW];
            const int orow = (yo * BW * CQO + ct0 * 4) * GB * 16;
            if (RES) {
#pragma unroll
                for (int i = 0; i < XT; ++i) {
                    int xo = x0 + i;
                    xo = xo >= BW ? BW - 1 : xo;
#pragma unroll
                    for (int tl = 0; tl < TPW; ++tl)
                        rr[i][tl] = buf_ld_frag(rs_o, lane_x, orow + (xo * CQO + tl * 4) * GB * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < XT; ++i) {
                const int xo = x0 + i;
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) {
                    const f32x4 c = acc0[i][tl];
                    float vx = fmaf(c[0], sc[tl].x, sh[tl].x);
                    float vy = fmaf(c[1], sc[tl].y, sh[tl].y);
                    float vz = fmaf(c[2], sc[tl].z, sh[tl].z);
                    float vw = fmaf(c[3], sc[tl].w, sh[tl].w);
                    if (RES) { vx += rr[i][tl].v[0]; vy += rr[i][tl].v[1]; vz += rr[i][tl].v[2]; vw += rr[i][tl].v[3]; }
                    u32x4 o;
                    o.x = __float_as_uint(fmaxf(vx, 0.f)); o.y = __float_as_uint(fmaxf(vy, 0.f));
                    o.z = __float_as_uint(fmaxf(vz, 0.f)); o.w = __float_as_uint(fmaxf(vw, 0.f));
                    // The whole address goes into the per-lane offset, soffset stays the constant 0:
                    // a 128-bit MUBUF store reads its data registers for a few cycles after issue,
                    // and the compiler only inserts the wait states that protects them from the
                    // next VALU write when soffset is NOT an SGPR. With an SGPR soffset the rows
                    // of boards 12-15 (the last data beat) were overwritten on gfx950.
                    if (xo < BW)
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, lane_x + orow + (xo * CQO + tl * 4) * GB * 16, 0, 0);
                }
            }
        };
        auto slide = [&]() {
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int tl = 0; tl < TPW; ++tl) {
                    acc0[i][tl] = acc1[i][tl];
                    acc1[i][tl] = acc2[i][tl];
                    acc2[i][tl] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        };

        // input rows that feed the output rows [yb, ye): one halo row above and below
        const int y0 = yb > 0 ? yb - 1 : 0;
        const int y1 = ye < BW ? ye : BW - 1;
        load_x(y0, 0, xa);
        load_w(0, 0);
        load_w(0, 1);
        load_w(0, 2);
        for (int yi = y0; yi <= y1; ++yi) {
            for (int cqg = 0; cqg < ncqg; cqg += 2) {
                step(xa, xb, yi, yi, cqg + 1, cqg * 4 < cq_real);
                const bool same = cqg + 2 < ncqg;
                const bool last = !same && (yi + 1 > y1);  // end of the chunk: harmless re-load
                step(xb, xa, yi, same || last ? yi : yi + 1, same ? cqg + 2 : (last ? cqg + 1 : 0),
                     (cqg + 1) * 4 < cq_real);
            }
            if (yi - 1 >= yb) epilogue(yi - 1);
            slide();
        }
        if (ye == BW) epilogue(BW - 1);  // after the last slide the bottom row sits in acc0
    }
}

// Policy and value heads of one 16-board group inside the resident kernel (model.py:34-73):
// 1x1 convs + BN + ReLU into LDS (flatten order c*A + cell, as the reference's .view), then one
// wave per board: policy_fc + softmax, value_fc1 + ReLU + value_fc2 + tanh.
// H16: the activations are in the split-fp16 layout of k_trunk16h (x = high half + low half)
template <int BW, bool H16 = false, typename Args = TrunkArgs>
__device__ __forceinline__ void trunk_heads(const Args& a, const float4* act, size_t gbase, int grp) {
    constexpr int A = BW * BW;
    constexpr int GB = 16;
    constexpr int NA = (A + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float s_heads[];
    const int planes = a.COUT, CQ = a.CQ;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    float* s_w3 = s_heads;              // [3][planes]
    float* s_h = s_w3 + 3 * planes;     // [16 boards][3][A]
    for (int i = tid; i < 3 * planes; i += nthreads) s_w3[i] = a.w3[i];
    __syncthreads();
    {
        const int b = tid & 15;
        for (int cell = tid >> 4; cell < A; cell += nthreads >> 4) {
            const float4* xp = act + ((gbase + cell) * CQ) * GB + b;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int cq = 0; cq < CQ; ++cq) {
                float4 x;
                if (H16) {
                    // [cell][c32][split][kq 4][board 16][8 halfs]: quad cq = halfs (cq&1)*4.. of oct (cq&7)>>1 of block cq>>3
                    const char* base = reinterpret_cast<const char*>(act) +
                                       ((((gbase + cell) * (CQ >> 3) + (cq >> 3)) * 2) * 64 + ((cq & 7) >> 1) * 16 + b) * 16 +
                                       (cq & 1) * 8;
                    const half4 hh = *reinterpret_cast<const half4*>(base);
                    const half4 hl = *reinterpret_cast<const half4*>(base + 1024);
                    x = make_float4(static_cast<float>(hh[0]) + static_cast<float>(hl[0]),
                                    static_cast<float>(hh[1]) + static_cast<float>(hl[1]),
                                    static_cast<float>(hh[2]) + static_cast<float>(hl[2]),
                                    static_cast<float>(hh[3]) + static_cast<float>(hl[3]));
                } else {
                    x = xp[static_cast<size_t>(cq) * GB];
                }
                const float* w0 = s_w3 + 4 * cq;
                const float* w1 = s_w3 + planes + 4 * cq;
                const float* w2 = s_w3 + 2 * planes + 4 * cq;
                a0 = fmaf(x.x, w0[0], a0); a0 = fmaf(x.y, w0[1], a0); a0 = fmaf(x.z, w0[2], a0); a0 = fmaf(x.w, w0[3], a0);
                a1 = fmaf(x.x, w1[0], a1); a1 = fmaf(x.y, w1[1], a1); a1 = fmaf(x.z, w1[2], a1); a1 = fmaf(x.w, w1[3], a1);
                a2 = fmaf(x.x, w2[0], a2); a2 = fmaf(x.y, w2[1], a2); a2 = fmaf(x.z, w2[2], a2); a2 = fmaf(x.w, w2[3], a2);
            }
            float* h = s_h + b * 3 * A + cell;
            h[0] = fmaxf(fmaf(a0, a.sc3[0], a.sh3[0]), 0.f);
            h[A] = fmaxf(fmaf(a1, a.sc3[1], a.sh3[1]), 0.f);
            h[2 * A] = fmaxf(fmaf(a2, a.sc3[2], a.sh3[2]), 0.f);
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
    if (H16) {
        // Both FC layers for the 16 boards AT ONCE: a weight is loaded once and used for all 16 boards (one
        // wave per board re-read the 93 KB of FC weights 16 times: ~60 us per group). Waves split the input
        // index j, lanes the outputs, every lane keeps 16 board accumulators per output; partial sums meet in LDS.
        float* s_pp = s_h + GB * 3 * A;        // [nw][GB][A]      policy_fc partials
        float* s_vp = s_pp + nw * GB * A;      // [nw][GB][planes] value_fc1 partials
        const int NP = (planes + 63) / 64;
        {
            const int js = (2 * A + nw - 1) / nw, j0 = wave * js, j1 = min(2 * A, j0 + js);
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int o = lane + 64 * c;
                float acc[GB];
#pragma unroll
                for (int bb = 0; bb < GB; ++bb) acc[bb] = 0.f;
                if (o < A) {
                    for (int j = j0; j < j1; ++j) {
                        const float w = a.wp_t[static_cast<size_t>(j) * A + o];
#pragma unroll
                        for (int bb = 0; bb < GB; ++bb) acc[bb] = fmaf(w, s_h[bb * 3 * A + j], acc[bb]);
                    }
#pragma unroll
                    for (int bb = 0; bb < GB; ++bb) s_pp[(wave * GB + bb) * A + o] = acc[bb];
                }
            }
        }
        {
            const int js = (A + nw - 1) / nw, j0 = wave * js, j1 = min(A, j0 + js);
            for (int c = 0; c < NP; ++c) {
                const int o = lane + 64 * c;
                float acc[GB];
#pragma unroll
                for (int bb = 0; bb < GB; ++bb) acc[bb] = 0.f;
                if (o < planes) {
                    for (int j = j0; j < j1; ++j) {
                        const float w = a.w1_t[static_cast<size_t>(j) * planes + o];
#pragma unroll
                        for (int bb = 0; bb < GB; ++bb) acc[bb] = fmaf(w, s_h[bb * 3 * A + 2 * A + j], acc[bb]);
                    }
#pragma unroll
                    for (int bb = 0; bb < GB; ++bb) s_vp[(wave * GB + bb) * planes + o] = acc[bb];
                }
            }
        }
        __syncthreads();
        for (int bb = wave; bb < GB; bb += nw) {   // softmax / tanh: one wave per board
            const size_t board = static_cast<size_t>(grp) * GB + bb;
            float lg[NA];
            float mx = -3.0e38f;
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int o = lane + 64 * c;
                lg[c] = -3.0e38f;
                if (o < A) {
                    float t = a.bp[o];
                    for (int q = 0; q < nw; ++q) t += s_pp[(q * GB + bb) * A + o];
                    lg[c] = t;
                    mx = fmaxf(mx, t);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int o = lane + 64 * c;
                lg[c] = (o < A) ? expf(lg[c] - mx) : 0.f;
                sum += lg[c];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const int o = lane + 64 * c;
                if (o < A) a.policy[board * A + o] = lg[c] / sum;
            }
            float part = 0.f;
            for (int o = lane; o < planes; o += 64) {
                float t = a.b1[o];
                for (int q = 0; q < nw; ++q) t += s_vp[(q * GB + bb) * planes + o];
                part = fmaf(a.w2[o], fmaxf(t, 0.f), part);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            if (lane == 0) a.value[board] = tanhf(part + a.b2[0]);
        }
        return;
    }
    for (int bb = wave; bb < GB; bb += nw) {
        const float* h = s_h + bb * 3 * A;
        const size_t board = static_cast<size_t>(grp) * GB + bb;
        float lg[NA];
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const int o = lane + 64 * c;
            lg[c] = -3.0e38f;
            if (o < A) {
                float acc = a.bp[o];
                for (int j = 0; j < 2 * A; ++j) acc = fmaf(a.wp_t[static_cast<size_t>(j) * A + o], h[j], acc);
                lg[c] = acc;
                mx = fmaxf(mx, acc);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const int o = lane + 64 * c;
            lg[c] = (o < A) ? expf(lg[c] - mx) : 0.f;
            sum += lg[c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const int o = lane + 64 * c;
            if (o < A) a.policy[board * A + o] = lg[c] / sum;
        }
        float part = 0.f;
        for (int o = lane; o < planes; o += 64) {
            float acc = a.b1[o];
            for (int j = 0; j < A; ++j) acc = fmaf(a.w1_t[static_cast<size_t>(j) * planes + o], h[2 * A + j], acc);
            part = fmaf(a.w2[o], fmaxf(acc, 0.f), part);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) a.value[board] = tanhf(part + a.b2[0]);
    }
}

// TPW = output-channel tiles per wave (1 is what runs: one wave per tile, two waves per SIMD).
template <int BW, int XT, int TPW>
__global__ __launch_bounds__(256 * (3 - TPW), 1) void k_trunk16(TrunkArgs a) {
    constexpr int A = BW * BW;
    const int grp = blockIdx.x;
    const int lane = threadIdx.x & 63;
    // readfirstlane makes the wave index provably uniform: it feeds buffer-load SGPR offsets, and a
    // "divergent" offset would wrap every such load in a waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int ct0 = wave * TPW;       // first output-channel tile (16 couts) of this wave
    const int kq = lane >> 4;         // which of the 4 channel quads of a k-step this lane loads
    const int b = lane & 15;
    const size_t gbase = static_cast<size_t>(grp) * A;

    for (int l = 0; l < a.nlayers; ++l) {
        const float4* src = (l == 0) ? a.in0 : ((l & 1) ? a.bufA : a.bufB);
        float4* dst = (l == 0) ? a.bufA : ((l & 1) ? a.bufB : a.bufA);
        // even l > 0: second conv of a ResBlock, + x (held in bufA = dst)
        trunk_layer<BW, XT, TPW>(src, dst, a.layers[l].w, a.layers[l].sc, a.layers[l].sh, l > 0 && (l & 1) == 0,
                                 l == 0 ? a.cq0 : a.CQ, l == 0 ? a.cq0_real : a.CQ, a.COUT, gbase, ct0, kq, b, 0, BW);
        // layer boundary inside the workgroup: all stores of this layer acknowledged, then a WORKGROUP-scope
        // acquire. The group's activations are private to this workgroup, whose waves share one CU and one L1
        // (write-through, coherent for the CU's own stores), so nothing has to be invalidated; the agent-scope
        // acquire used at first (buffer_inv sc1) made every CU re-fetch its working set after each layer -- 15 k
        // cycles per layer of the split-fp16 kernel (AO_PROF phase timing).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // the trunk output of this group (bufA: nlayers is odd) is still in this XCD's L2: run both heads
    trunk_heads<BW>(a, a.bufA, gbase, grp);
}

// ----------------------------------------------------------------------------------------------
// k_trunk16h -- the group-resident trunk with the fp32 contraction carried by fp16 MFMAs.
//
// Every fp32 operand is split into two halves, x = xh + xl (xh = fp16(x), xl = fp16(x - xh)),
// and x*w is formed as xh*wh + xh*wl + xl*wh with v_mfma_f32_16x16x32_f16: each fp16 x fp16
// product is exact in fp32 and the accumulation is fp32, so the only departure from an fp32
// contraction is the dropped xl*wl term, <= 2^-22 of the product (fp32's own rounding is 2^-24).
// Weights are pre-scaled by a power of two per layer (undone exactly in the BatchNorm scale) so
// that their low halves stay normal numbers. Three fp16 MFMAs do the work of eight fp32 MFMAs at
// half the cycles each: 5.3x fewer matrix-pipe cycles than k_trunk16.
//
// That only pays if the operands keep up (2 KB per 16-cycle MFMA):
//   * activations are shared by the eight waves of the workgroup through LDS: one input row (9 cells x
//     128 channels x 16 boards x {high, low} = 72 KB) is staged with LDS-direct loads while the previous
//     one is consumed (144 KB of the CU's 160 KB);
//   * a wave owns one 16-channel output tile, two waves per SIMD (as in k_trunk16: the other wave's
//     MFMAs cover this wave's loads -- a one-wave-per-SIMD variant with the weights resident in 512
//     registers ran at 37 % MFMA utilisation because every load issue was exposed);
//   * weights stream from L2, one (32-channel block, tap row) slab = 3 taps x {high, low} ahead:
//     18.5 B/cycle/CU, 2.5x the fp32 kernel's operand traffic;
//   * same sliding window of three output rows as the fp32 kernel (108 accumulator registers).
// Layout of a group's activations: [cell][32-channel block][half: high, low][k-oct 4][board 16][8 x fp16]
// (a fragment = 1 KB = one B operand of the MFMA: lane = oct*16 + board holds 8 consecutive channels).
// ----------------------------------------------------------------------------------------------
struct TrunkHLayer {
    const uint4* wh;   // [tap 9][c32][tile][lane 64] 8 x fp16: high halves, lane = oct*16 + cout
    const uint4* wl;   // low halves
    const float4* sc;  // BatchNorm scale x 2^-s (s = the layer's weight pre-scale)
    const float4* sh;
};

struct TrunkHArgs {
    const float4* in0;  // fp32 plane batch [grp][cell][quad 8][board 16] (conv1 input)
    uint4* bufA;  // conv1 output / ResBlock input-output (split-fp16 layout)
    uint4* bufB;
    int nlayers;  // 1 + 2 * n_block, conv1 included
    int CQ, COUT;
    const float *w3, *sc3, *sh3, *wp_t, *bp, *w1_t, *b1, *w2, *b2;
    float* policy;
    float* value;
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
__device__ __forceinline__ half4 buf_ld_h4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st_h4(half4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
}

#ifdef AO_PROF
// phase timing of k_trunk16h (build with AO_EXTRA_FLAGS=-DAO_PROF; tools/time_net.py prints it): shader-clock
// cycles per wave of one group, summed over the trunk layers: [0] row-0 staging, [1] slab loops, [2] row
// epilogues, [3] row barriers, [4] last epilogue + layer boundary, [5] heads, [6] conv1 total
__device__ unsigned long long ao_prof[8 * 12];
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
// stores), 5 / 6 activations of all groups aliased to an 85 / 170 MB footprint. Measured: profiles/r1j_trunk16h_phase_timing.txt
#ifndef AO_KO
#define AO_KO 0
#endif
#if AO_KO != 0 && !defined(AO_PROF)
#error "AO_KO builds compute wrong results on purpose: timing only, build them with -DAO_PROF"
#endif
template <int BW, int NC32, int NCI, bool FIRST>
__device__ __forceinline__ void trunk_h_layer(const void* src, uint4* dst, const TrunkHLayer& L, const bool RES, uint4* s_x,
                                              int tile, int lane, unsigned long long* prof) {
    constexpr int A = BW * BW;
    constexpr int NT = NC32 * 2;             // 16-channel output tiles = waves (two per SIMD at 128 channels)
    constexpr int NSP = 2;                   // halves of an input fragment
    constexpr int NFR = BW * NCI * NSP;      // input fragments per board row
    constexpr int NB = NCI * 3;              // (32-channel block, tap row) slabs per input row
    constexpr int NPR = 3;                   // MFMA products per multiply-add
    const int kq = lane >> 4, b = lane & 15;
    const int lane16 = lane * 16;
    // per-lane byte offset of this lane's 4 output channels inside a (cell, 32-channel block) fragment pair
    const int out_voff = (((tile & 1) * 2 + (kq >> 1)) * 16 + b) * 16 + (kq & 1) * 8;
    const float4 sc = L.sc[tile * 4 + kq], sh = L.sh[tile * 4 + kq];
    const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(L.wh, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(L.wl, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_src =
        make_rsrc(src, FIRST ? static_cast<unsigned>(A) * 8u * 16u * 16u : static_cast<unsigned>(A) * NCI * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rs_dst = make_rsrc(dst, static_cast<unsigned>(A) * NC32 * 2u * 1024u);
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
            const int ub = (((dy * 3 + dx) * NCI + c) * NT + tile) * 1024;
            W[0][dx] = buf_ld_h8(rs_wh, lane16, ub);
            W[1][dx] = buf_ld_h8(rs_wl, lane16, ub);
        }
    };
    if (FIRST) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wres[0][t] = buf_ld_h8(rs_wh, lane16, (t * NT + tile) * 1024);
            wres[1][t] = buf_ld_h8(rs_wl, lane16, (t * NT + tile) * 1024);
        }
    }
    // conv1: one fragment = channels 8*kq .. 8*kq+7 of (cell, board b) = two float4 quads of the fp32 batch
    auto load_planes = [&](int cell, int split) -> half8 {   // split 0: high halves, 1: low halves (0 for 0/1 planes)
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
            if (RES) {
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const int i = i0 + k < BW ? i0 + k : BW - 1;
                    const int ob = (((yo * BW + i) * NC32 + (tile >> 1)) * 2) * 1024;
                    rh[k] = buf_ld_h4(rs_dst, out_voff, ob);
                    rl[k] = buf_ld_h4(rs_dst, out_voff, ob + 1024);
                }
            }
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int i = i0 + k;
                if (i >= BW) continue;
                const f32x4 c = acc[0][i];
                float f[4] = {fmaf(c[0], sc.x, sh.x), fmaf(c[1], sc.y, sh.y), fmaf(c[2], sc.z, sh.z), fmaf(c[3], sc.w, sh.w)};
                const int ob = (((yo * BW + i) * NC32 + (tile >> 1)) * 2) * 1024;
                if (RES) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] += static_cast<float>(rh[k][r]) + static_cast<float>(rl[k][r]);
                }
                half4 hh, hl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // ReLU; the upper clamp keeps an activation beyond the fp16 range (65504 -- far outside what a
                    // BatchNorm-ed residual tower produces) finite instead of turning the board into inf / NaN
                    const float v = fminf(fmaxf(f[r], 0.f), 65504.f);
                    hh[r] = static_cast<_Float16>(v);
                    hl[r] = static_cast<_Float16>(v - static_cast<float>(hh[r]));
                }
                buf_st_h4(hh, rs_dst, out_voff, ob);
                buf_st_h4(hl, rs_dst, out_voff, ob + 1024);
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
    } else {
#pragma unroll
        for (int k = 0; k < (NFR + NT - 1) / NT; ++k) {
            const int f = tile + NT * k;
            if (f < NFR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(s_x + f * 64), 16, lane16,
                                                         f * 1024, 0, 0);
        }
    }
    AO_T(t_a2);
    load_w(0, wA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
            if (dy == 1 && AO_KO != 3) {
                // next input row into LDS, a share per block (always-executed slab)
#pragma unroll
                for (int k = 0; k < (NFR / NT + NCI) / NCI; ++k) {
                    const int f = tile + NT * (c * ((NFR / NT + NCI) / NCI) + k);
                    if (f < NFR) {
                        if (FIRST) xn[f * 64 + lane] = __builtin_bit_cast(uint4, load_planes(yn * BW + (f >> 1), f & 1));
                        else
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xn + f * 64),
                                                                     16, lane16, (yn * NFR + f) * 1024, 0, 0);
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
        if (yi >= 1 && (AO_KO != 4 || yi == 1)) epilogue(yi - 1);
#pragma unroll
        for (int i = 0; i < BW; ++i) {
            acc[0][i] = acc[1][i];
            acc[1][i] = acc[2][i];
            acc[2][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        AO_T(t_r2);
        // next row staged by all waves (LDS-direct loads count in vmcnt), this row's buffer free
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        AO_T(t_r3);
        AO_ACC(1, t_r0, t_r1);
        AO_ACC(2, t_r1, t_r2);
        AO_ACC(3, t_r2, t_r3);
    }
    AO_T(t_c);
    epilogue(BW - 1);
    // layer boundary inside the workgroup (see k_trunk16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    AO_T(t_d);
    AO_ACC(4, t_c, t_d);
}

// The same layer for ONE (row chunk [yb, ye), column tile x0 .. x0+XT-1) of a group: the per-layer form for
// batches too small to give every CU a whole group (k_layer16h: one launch per conv, workgroup = group x row
// chunk x column tile) and for boards whose rows do not fit LDS (15 x 15: XT = 5, a staged row is the tile
// plus one halo column on each side = 7 cells = 56 KB, two of them 112 KB). Halo columns that fall off the
// board are staged as zeros, so the MFMA stream needs no per-column conditions; halo rows are handled by the
// slab conditions (uniform per row) exactly as in the fp32 row-chunk kernel.
template <int BW, int XT, int NC32, int NCI, bool FIRST>
__device__ __forceinline__ void trunk_h_layer_tile(const void* src, uint4* dst, const TrunkHLayer& L, const bool RES,
                                                   uint4* s_x, int tile, int lane, int x0, int yb, int ye) {
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
        make_rsrc(src, FIRST ? static_cast<unsigned>(A) * 8u * 16u * 16u : static_cast<unsigned>(A) * NCI * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rs_dst = make_rsrc(dst, static_cast<unsigned>(A) * NC32 * 2u * 1024u);
    half8 wA[2][3], wB[2][3], wres[2][FIRST ? 9 : 1];
    auto load_w = [&](int slab, half8 (&W)[2][3]) {
        if (FIRST) return;
        const int c = (slab / 3) % NCI, dy = slab % 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ub = (((dy * 3 + dx) * NCI + c) * NT + tile) * 1024;
            W[0][dx] = buf_ld_h8(rs_wh, lane16, ub);
            W[1][dx] = buf_ld_h8(rs_wl, lane16, ub);
        }
    };
    if (FIRST) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wres[0][t] = buf_ld_h8(rs_wh, lane16, (t * NT + tile) * 1024);
            wres[1][t] = buf_ld_h8(rs_wl, lane16, (t * NT + tile) * 1024);
        }
    }
    auto load_planes = [&](int cell, int split) -> half8 {   // split 0: high halves, 1: low halves (0 for 0/1 planes)
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
                                                             lane16, ((y * BW + xin) * NCI * 2 + rest) * 1024, 0, 0);
                }
            }
        }
    };
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
            if (slab == 1 && yi < y1) stage(yi + 1, xn);
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
        if (yi - 1 >= yb) epilogue(yi - 1);
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            acc[0][i] = acc[1][i];
            acc[1][i] = acc[2][i];
            acc[2][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (ye == BW) epilogue(BW - 1);
}

struct LayerHArgs {
    const void* src;   // fp32 plane batch (conv1) or split-fp16 activations
    uint4* dst;
    TrunkHLayer layer;
    int res, nch;
};

template <int BW, int XT, int NC32, bool FIRST>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_layer16h(LayerHArgs a) {
    constexpr int A = BW * BW;
    constexpr int NXT = (BW + XT - 1) / XT;
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];
    const int xt = blockIdx.x % NXT;
    const int rest = blockIdx.x / NXT;
    const int grp = rest / a.nch, ch = rest - grp * a.nch;
    const int base = BW / a.nch, extra = BW % a.nch;
    const int yb = ch * base + (ch < extra ? ch : extra);
    const int ye = yb + base + (ch < extra ? 1 : 0);
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    uint4* dst = a.dst + static_cast<size_t>(grp) * A * NC32 * 2 * 64;
    if (FIRST) {
        trunk_h_layer_tile<BW, XT, NC32, 1, true>(static_cast<const float4*>(a.src) + static_cast<size_t>(grp) * A * 8 * 16, dst, a.layer,
                                                  false, s_x, tile, lane, xt * XT, yb, ye);
    } else {
        trunk_h_layer_tile<BW, XT, NC32, NC32, false>(static_cast<const uint4*>(a.src) + static_cast<size_t>(grp) * A * NC32 * 2 * 64, dst,
                                                      a.layer, a.res != 0, s_x, tile, lane, xt * XT, yb, ye);
    }
}

template <int BW, int NC32>
__global__ __launch_bounds__(NC32 * 2 * 64, 1) void k_trunk16h(TrunkHArgs a) {
    constexpr int A = BW * BW;
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];  // [2][row fragments][64] uint4
    const int grp = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);  // this wave's output tile
    // first activation fragment of this group (AO_KO 5 / 6: groups share buffers, timing experiment only)
    const size_t gfrag = static_cast<size_t>(AO_KO == 5 ? grp % 64 : AO_KO == 6 ? grp % 128 : grp) * A * NC32 * 2;
    uint4* bufA = a.bufA + gfrag * 64;
    uint4* bufB = a.bufB + gfrag * 64;
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long* pp = prof;
    // conv1: fp32 planes -> x
    AO_T(t0);
    trunk_h_layer<BW, NC32, 1, true>(a.in0 + static_cast<size_t>(grp) * A * 8 * 16, bufA, a.layers[0], false, s_x, tile, lane, pp);
    AO_T(t1);
#ifdef AO_PROF
    for (int k = 0; k < 12; ++k) prof[k] = 0;
#endif
    for (int l = 1; l < a.nlayers; ++l) {
        // l odd: first conv of a ResBlock (x -> t); l even: second conv (t -> x, + x in place)
        const bool second = (l & 1) == 0;
        trunk_h_layer<BW, NC32, NC32, false>(second ? bufB : bufA, second ? bufA : bufB, a.layers[l], second, s_x, tile, lane, pp);
    }
    AO_T(t2);
    trunk_heads<BW, true>(a, reinterpret_cast<const float4*>(a.bufA), static_cast<size_t>(grp) * A, grp);
#ifdef AO_PROF
    AO_T(t3);
    prof[5] = t3 - t2;
    prof[6] = t1 - t0;
    prof[7] = t3 - t0;
    if (grp == 5 && lane == 0)
        for (int k = 0; k < 12; ++k) ao_prof[tile * 12 + k] = prof[k];
#endif
}

// ----------------------------------------------------------------------------------------------
// Small batches (a drop-in ZeroAgent has ONE game): boards cannot fill the MFMA N dimension, so
// the CELLS of one board do:   D[cout 16][cell 16] += Wt[cout 16][k 4] * X[k 4][cell 16]
// on the plain per-board NHWC layout act[board][cell][channel]. One wave per (16 cells, 16 output
// channels, board): a 9x9x128 layer is 48 independent waves of 288 MFMAs (~4 us) instead of nine
// workgroups of ~130 us, which is what matters when 400 evaluations run back to back.
// Out-of-board taps are zero-filled per lane (cells of a tile differ in position).
// ----------------------------------------------------------------------------------------------
// NCQG = 16-channel k-steps per tap; NW = waves per tile (9: one tap each, 3: one tap row each).
// The tile code is conv_cells_tile (net_device.hpp), shared with the persistent single-game kernel.
template <int BW, int NCQG, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_conv_cells(const float4* __restrict__ in, const float4* __restrict__ wt,
                                                   const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                   const float4* res, float4* out, int CQI, int COUT, int relu_res) {
    __shared__ float s_red[(NW - 1) * 64 * 4];
    // 1-D grid with the output-channel tile fastest: workgroups are dispatched round-robin over
    // the 8 XCDs, so (for 8 tiles) XCD x only ever reads the weights of tile x -- 1/8 of the
    // network per L2, which then stays resident from one evaluation to the next (the whole net
    // is 5 MB, an XCD's L2 4 MB).
    const int ntile = COUT >> 4;
    const int ct = blockIdx.x % ntile;
    const int rest = blockIdx.x / ntile;
    constexpr int NCT = (BW * BW + 15) / 16;
    conv_cells_tile<BW, NCQG, NW>(in, wt, scale, shift, res, out, CQI, COUT, relu_res, ct, rest % NCT, rest / NCT, s_red);
}

// One conv layer per launch for medium batches: a 16-board group is split into `nch` row chunks,
// one workgroup each, so 64 groups x 4 chunks still give every CU one workgroup. The chunk runs
// the same sliding-window code over its rows (plus one halo input row on each side); the launch
// boundary is the synchronisation between layers, nothing is exchanged inside a launch.
struct LayerArgs {
    const float4* src;
    float4* dst;
    TrunkLayer layer;
    int res, cqi, cq_real, COUT, nch;
};

template <int BW, int XT>
__global__ __launch_bounds__(512, 1) void k_layer16(LayerArgs a) {
    constexpr int A = BW * BW;
    const int grp = blockIdx.x / a.nch;
    const int c = blockIdx.x - grp * a.nch;
    const int base = BW / a.nch, extra = BW % a.nch;
    const int yb = c * base + (c < extra ? c : extra);
    const int ye = yb + base + (c < extra ? 1 : 0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    trunk_layer<BW, XT, 1>(a.src, a.dst, a.layer.w, a.layer.sc, a.layer.sh, a.res != 0, a.cqi, a.cq_real, a.COUT,
                           static_cast<size_t>(grp) * A, wave, lane >> 4, lane & 15, yb, ye);
}

// 1x1 convs of both heads (model.py:37,56) + their BatchNorm + ReLU.
// hbuf[board][3][A]: channel 0,1 = policy head, 2 = value head.
// H16: `in` is in the split-fp16 layout of the k_trunk16h / k_layer16h kernels (GB = 16)
template <bool H16>
__global__ __launch_bounds__(256) void k_head_conv(const float4* __restrict__ in, const float* __restrict__ w3,
                                                   const float* __restrict__ sc3, const float* __restrict__ sh3,
                                                   float* __restrict__ hbuf, int A, int CQ, int GB) {
    extern __shared__ float s_w[];  // [3][planes]
    const int planes = CQ * 4;
    for (int i = threadIdx.x; i < 3 * planes; i += blockDim.x) s_w[i] = w3[i];
    __syncthreads();
    const int ppb = 256 / GB;  // cells per block
    const int nchunk = (A + ppb - 1) / ppb;
    const int grp = blockIdx.x / nchunk;
    const int pos = (blockIdx.x - grp * nchunk) * ppb + static_cast<int>(threadIdx.x) / GB;
    const int b = static_cast<int>(threadIdx.x) % GB;
    if (pos >= A) return;
    const float4* xp = in + ((static_cast<size_t>(grp) * A + pos) * CQ) * GB + b;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int cq = 0; cq < CQ; ++cq) {
        float4 x;
        if (H16) {
            const char* base = reinterpret_cast<const char*>(in) +
                               ((((static_cast<size_t>(grp) * A + pos) * (CQ >> 3) + (cq >> 3)) * 2) * 64 + ((cq & 7) >> 1) * 16 + b) * 16 +
                               (cq & 1) * 8;
            const half4 hh = *reinterpret_cast<const half4*>(base);
            const half4 hl = *reinterpret_cast<const half4*>(base + 1024);
            x = make_float4(static_cast<float>(hh[0]) + static_cast<float>(hl[0]), static_cast<float>(hh[1]) + static_cast<float>(hl[1]),
                            static_cast<float>(hh[2]) + static_cast<float>(hl[2]), static_cast<float>(hh[3]) + static_cast<float>(hl[3]));
        } else {
            x = xp[static_cast<size_t>(cq) * GB];
        }
        const float* w0 = s_w + 4 * cq;
        const float* w1 = s_w + planes + 4 * cq;
        const float* w2 = s_w + 2 * planes + 4 * cq;
        a0 = fmaf(x.x, w0[0], a0); a0 = fmaf(x.y, w0[1], a0); a0 = fmaf(x.z, w0[2], a0); a0 = fmaf(x.w, w0[3], a0);
        a1 = fmaf(x.x, w1[0], a1); a1 = fmaf(x.y, w1[1], a1); a1 = fmaf(x.z, w1[2], a1); a1 = fmaf(x.w, w1[3], a1);
        a2 = fmaf(x.x, w2[0], a2); a2 = fmaf(x.y, w2[1], a2); a2 = fmaf(x.z, w2[2], a2); a2 = fmaf(x.w, w2[3], a2);
    }
    const size_t board = static_cast<size_t>(grp) * GB + b;
    float* h = hbuf + board * 3 * A + pos;
    h[0] = fmaxf(fmaf(a0, sc3[0], sh3[0]), 0.f);
    h[A] = fmaxf(fmaf(a1, sc3[1], sh3[1]), 0.f);
    h[2 * A] = fmaxf(fmaf(a2, sc3[2], sh3[2]), 0.f);
}

// policy_fc + softmax (model.py:40-50), value_fc1 + ReLU + value_fc2 + tanh (model.py:59-73).
// One block per board. The flatten order before the FCs is NCHW (c*A + cell), which is hbuf's.
__global__ __launch_bounds__(256) void k_head_fc(const float* __restrict__ hbuf, const float* __restrict__ wp_t,
                                                 const float* __restrict__ bp, const float* __restrict__ w1_t,
                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, float* __restrict__ policy,
                                                 float* __restrict__ value, int A, int planes) {
    extern __shared__ float s_h[];  // [3A] inputs, [A] logits, [planes] hidden, [8] reduce
    float* s_logit = s_h + 3 * A;
    float* s_hid = s_logit + A;
    float* s_red = s_hid + planes;
    const size_t board = blockIdx.x;
    for (int i = threadIdx.x; i < 3 * A; i += blockDim.x) s_h[i] = hbuf[board * 3 * A + i];
    __syncthreads();
    float lmax = -3.0e38f;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        float acc = bp[a];
        for (int j = 0; j < 2 * A; ++j) acc = fmaf(wp_t[static_cast<size_t>(j) * A + a], s_h[j], acc);
        s_logit[a] = acc;
        lmax = fmaxf(lmax, acc);
    }
    lmax = block_reduce(lmax, s_red, true);
    float lsum = 0.f;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
        const float ex = expf(s_logit[a] - lmax);
        s_logit[a] = ex;
        lsum += ex;
    }
    lsum = block_reduce(lsum, s_red, false);
    for (int a = threadIdx.x; a < A; a += blockDim.x) policy[board * A + a] = s_logit[a] / lsum;
    // value head
    for (int o = threadIdx.x; o < planes; o += blockDim.x) {
        float acc = b1[o];
        for (int j = 0; j < A; ++j) acc = fmaf(w1_t[static_cast<size_t>(j) * planes + o], s_h[2 * A + j], acc);
        s_hid[o] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    float part = 0.f;
    for (int o = threadIdx.x; o < planes; o += blockDim.x) part = fmaf(w2[o], s_hid[o], part);
    part = block_reduce(part, s_red, false);
    if (threadIdx.x == 0) value[board] = tanhf(part + b2[0]);
}

// Both heads of ONE board in one block (per-board NHWC input), for the small-batch path
// (heads_board_dev, net_device.hpp).
__global__ __launch_bounds__(512) void k_heads_board(HeadParams h, const float4* __restrict__ act,
                                                     float* __restrict__ policy, float* __restrict__ value, int A,
                                                     int planes) {
    extern __shared__ float s_hb[];
    const size_t board = blockIdx.x;
    heads_board_dev(h, act + board * A * (planes >> 2), policy + board * A, value + board, A, planes, s_hb);
}

// [batch][C][A] float32 (Agent.model's input layout, agents.py:175) -> interleaved batch
__global__ void k_nchw_to_il(const float* __restrict__ x, float4* __restrict__ il, int batch, int C, int A,
                             int nchq, int boards_padded, int GB) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t total = static_cast<size_t>(boards_padded) * A;
    if (i >= total) return;
    const int board = static_cast<int>(i / A), cell = static_cast<int>(i - static_cast<size_t>(board) * A);
    const size_t grp = board / GB;
    const int b = board % GB;
    for (int cq = 0; cq < nchq; ++cq) {
        float v[4];
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * cq + k;
            v[k] = (board < batch && c < C) ? x[(static_cast<size_t>(board) * C + c) * A + cell] : 0.f;
        }
        il[((grp * A + cell) * nchq + cq) * GB + b] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace ao

// ==============================================================================================
// host side
// ==============================================================================================
struct ao_net {
    int nb = 0, C = 0, planes = 0, B = 0, A = 0, device = 0;
    int nchq32 = 0;  // input channel quads of the layer-kernel path (groups of 32 boards)
    int nchq16 = 0;  // ... of the group-resident path (groups of 16 boards): multiple of 8
    int CQ = 0;
    int mode = 0;    // 0 auto, 1 layer kernels (32), 2 group-resident trunk, 3 per-board, 4 row-chunked layers (16)
    int num_cu = 256;
    bool finalized = false;
    std::string err;
    std::map<std::string, std::vector<float>> params;
    std::vector<void*> allocs;
    // device parameters
    std::vector<float*> conv_w, conv_sc, conv_sh;  // [1 + 2*nb]; conv_w[0] packed for nchq32
    // split-fp16 trunk (mode 5): per trunk conv after conv1 the high / low weight halves (pre-scaled by a
    // power of two) and the BatchNorm scale with that power of two folded back
    std::vector<uint4*> convh_wh, convh_wl;
    std::vector<float*> convh_sc;
    float* conv0_w16 = nullptr;                    // conv1 weights packed for nchq16
    float* conv0_w1 = nullptr;                     // conv1 weights packed for nchq1 (per-board NHWC path)
    int nchq1 = 0;                                 // input channel quads of the per-board path: multiple of 4
    float *head_w3 = nullptr, *head_sc3 = nullptr, *head_sh3 = nullptr;
    float *wp_t = nullptr, *bp = nullptr, *w1_t = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    // workspace (sized in boards, padded to 32)
    int ws_boards = 0;
    float *act_x = nullptr, *act_t = nullptr, *hbuf = nullptr, *il_in = nullptr;
    float *tmp_p = nullptr, *tmp_v = nullptr;
    // timing of the dominant kernel (trunk conv launches)
    bool timing = false;
    static constexpr int kRing = 512;
    std::vector<hipEvent_t> ev0, ev1;
    int ring_head = 0, ring_count = 0;
    double ms_total = 0.0;
    int64_t launches = 0;

    int fail(const std::string& m) { err = m; return 1; }
};

#define NET_HIP(n, call)                                                                      \
    do {                                                                                      \
        hipError_t st_ = (call);                                                              \
        if (st_ != hipSuccess)                                                                \
            return (n)->fail(std::string(#call) + ": " + hipGetErrorString(st_));             \
    } while (0)

static thread_local std::string g_net_create_error;

template <typename T>
static int net_alloc(ao_net* n, T** out, size_t count) {
    void* p = nullptr;
    hipError_t st = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16));
    if (st != hipSuccess) return n->fail(std::string("hipMalloc: ") + hipGetErrorString(st));
    n->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return 0;
}

static int upload(ao_net* n, float** dst, const std::vector<float>& src) {
    if (net_alloc(n, dst, src.size())) return 1;
    NET_HIP(n, hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

static void harvest(ao_net* n, int count) {
    for (int i = 0; i < count; ++i) {
        const int idx = (n->ring_head - n->ring_count + ao_net::kRing * 2) % ao_net::kRing;
        hipEventSynchronize(n->ev1[idx]);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, n->ev0[idx], n->ev1[idx]) == hipSuccess) {
            n->ms_total += ms;
            n->launches += 1;
        }
        --n->ring_count;
    }
}

static int timer_begin(ao_net* n, hipStream_t s) {
    if (n->ring_count == ao_net::kRing) harvest(n, ao_net::kRing / 2);
    const int idx = n->ring_head;
    hipEventRecord(n->ev0[idx], s);
    return idx;
}

static void timer_end(ao_net* n, int idx, hipStream_t s) {
    hipEventRecord(n->ev1[idx], s);
    n->ring_head = (n->ring_head + 1) % ao_net::kRing;
    ++n->ring_count;
}

// mode 5 (split-fp16 MFMA trunk: k_trunk16h resident for boards up to 9x9 with >= 192 groups, k_layer16h per
// layer otherwise) is built for 128 planes and at least one ResBlock
static bool h16_supported(const ao_net* n) {
    return n->planes == 128 && n->nb >= 1 && 1 + 2 * n->nb <= ao::kMaxTrunkLayers && n->nchq16 == 8;
}

namespace ao {

int net_check(const ao_net* n, int board, int inplanes, int device, std::string* why) {
    if (!n->finalized) { *why = "network not finalized"; return 1; }
    if (n->B != board || n->C != inplanes) { *why = "network board/inplanes differ from the engine's"; return 1; }
    if (n->device != device) { *why = "network lives on another device"; return 1; }
    return 0;
}

// Execution plan for a batch of `boards` positions: which trunk runs and which interleaved input
// layout (boards per group, channel quads) it expects.
//   3  per-board NHWC, cells as MFMA N      -- up to ~160 9x9 boards (13k cells; latency path)
//   2  group-resident trunk, 16 boards/WG   -- >= 192 groups: one workgroup per CU for the whole net
//   4  one launch per layer over (16-board group x row chunk) -- everything in between
//   1  one launch per layer over 32-board groups x board rows (first-generation kernel, explicit only)
int pick_mode_public(const ao_net* n, int boards);
static int pick_mode(const ao_net* n, int boards, int* nch_out) {
    int mode = n->mode;
    const int g16 = (boards + 15) / 16;
    // measured cross-overs (us per simulation, 9x9, 4 blocks): per-board path vs fp32 row-chunked layers 529 / 676
    // at 128 boards and 979 / 676 at 256; per-board path vs split-fp16 layers 174 / 246 at 32 boards and 307 / 258
    // at 64 (15x15, 10 blocks: 518 / 501 at 16 boards)
    if (mode == 0) {
        const long cells = static_cast<long>(boards) * n->A;
        if (h16_supported(n)) mode = cells <= 3800 ? 3 : 5;
        else mode = cells <= 13000 ? 3 : (g16 >= 192 ? 2 : 4);
    }
    if (mode == 2 && (1 + 2 * n->nb > kMaxTrunkLayers)) mode = 4;
    int nch = 1;
    if (mode == 4) {
        // row chunks per group: minimise (rounds of workgroups over the CUs) x (rows per chunk)
        long best = -1;
        for (int c = 1; c <= n->B; ++c) {
            const long rounds = (static_cast<long>(g16) * c + n->num_cu - 1) / n->num_cu;
            const long cost = rounds * ((n->B + c - 1) / c);
            if (best < 0 || cost < best) { best = cost; nch = c; }
        }
    }
    if (nch_out) *nch_out = nch;
    return mode;
}

int pick_mode_public(const ao_net* n, int boards) { return pick_mode(n, boards, nullptr); }

void net_plan(const ao_net* n, int boards, int* group, int* nchq) {
    const int mode = pick_mode(n, boards, nullptr);
    if (mode == 2 || mode == 4 || mode == 5) { *group = 16; *nchq = n->nchq16; }
    else if (mode == 3) { *group = 1; *nchq = n->nchq1; }
    else { *group = 32; *nchq = n->nchq32; }
}

template <int BW>
static void launch_conv(ao_net* n, int layer, const float* in, int cqi, const float* res, float* out,
                        int groups, hipStream_t s) {
    constexpr int XT = (BW <= 9) ? BW : 8;  // cells per workgroup (accumulator tiles per wave)
    constexpr int NXT = (BW + XT - 1) / XT;
    const int nblk = groups * BW * NXT;
    const dim3 grid(nblk), block(64 * (n->planes / 32));
    const bool timed = n->timing && layer > 0;
    const int idx = timed ? timer_begin(n, s) : 0;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    const float4* w4 = reinterpret_cast<const float4*>(n->conv_w[layer]);
    const float4* sc4 = reinterpret_cast<const float4*>(n->conv_sc[layer]);
    const float4* sh4 = reinterpret_cast<const float4*>(n->conv_sh[layer]);
    float4* out4 = reinterpret_cast<float4*>(out);
    if (res)
        hipLaunchKernelGGL((k_conv3x3<BW, XT, true>), grid, block, 0, s, in4, w4, sc4, sh4,
                           reinterpret_cast<const float4*>(res), out4, cqi, n->planes, nblk);
    else
        hipLaunchKernelGGL((k_conv3x3<BW, XT, false>), grid, block, 0, s, in4, w4, sc4, sh4,
                           static_cast<const float4*>(nullptr), out4, cqi, n->planes, nblk);
    if (timed) timer_end(n, idx, s);
}

template <int BW>
static void launch_trunk16(ao_net* n, const float* in_il, int groups, float* policy, float* value,
                           hipStream_t s) {
    constexpr int XT = (BW <= 9) ? BW : 5;  // cells per window row: 3 rows x XT x 2 tiles of accumulators
    TrunkArgs a;
    a.in0 = reinterpret_cast<const float4*>(in_il);
    a.bufA = reinterpret_cast<float4*>(n->act_x);
    a.bufB = reinterpret_cast<float4*>(n->act_t);
    a.nlayers = 1 + 2 * n->nb;
    a.cq0 = n->nchq16;
    a.cq0_real = (n->C + 3) / 4;
    a.CQ = n->CQ;
    a.COUT = n->planes;
    a.w3 = n->head_w3; a.sc3 = n->head_sc3; a.sh3 = n->head_sh3;
    a.wp_t = n->wp_t; a.bp = n->bp; a.w1_t = n->w1_t; a.b1 = n->b1; a.w2 = n->w2; a.b2 = n->b2;
    a.policy = policy;
    a.value = value;
    for (int l = 0; l < a.nlayers; ++l) {
        a.layers[l].w = reinterpret_cast<const float4*>(l == 0 ? n->conv0_w16 : n->conv_w[l]);
        a.layers[l].sc = reinterpret_cast<const float4*>(n->conv_sc[l]);
        a.layers[l].sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
    }
    const int idx = n->timing ? timer_begin(n, s) : 0;
    // 96 KiB of (unused) dynamic LDS pins one workgroup per CU: with 256 groups every CU of the
    // chip gets exactly one group instead of some CUs receiving two
    // one wave per output-channel tile: 8 waves (2 per SIMD) at 128 channels. (A 2-tiles-per-wave
    // variant with the window in AGPRs was slower and is not instantiated.)
    hipLaunchKernelGGL((k_trunk16<BW, XT, 1>), dim3(groups), dim3(64 * (n->planes / 16)), 96 * 1024, s, a);
    if (n->timing) timer_end(n, idx, s);
}

static int ensure_workspace(ao_net* n, int boards) {
    boards = (boards + 31) / 32 * 32;
    if (boards <= n->ws_boards) return 0;
    // grow-only; old buffers stay in n->allocs until destroy (forward sizes rarely change)
    const size_t act = static_cast<size_t>(boards) * n->A * n->planes;
    if (net_alloc(n, &n->act_x, act) || net_alloc(n, &n->act_t, act) ||
        net_alloc(n, &n->hbuf, static_cast<size_t>(boards) * 3 * n->A) ||
        net_alloc(n, &n->il_in, static_cast<size_t>(boards) * n->A * std::max(std::max(n->nchq16, n->nchq32), n->nchq1) * 4) ||
        net_alloc(n, &n->tmp_p, static_cast<size_t>(boards) * n->A) || net_alloc(n, &n->tmp_v, boards))
        return 1;
    n->ws_boards = boards;
    return 0;
}

static HeadParams head_params(const ao_net* n) {
    HeadParams h;
    h.w3 = n->head_w3; h.sc3 = n->head_sc3; h.sh3 = n->head_sh3;
    h.wp_t = n->wp_t; h.bp = n->bp; h.w1_t = n->w1_t; h.b1 = n->b1; h.w2 = n->w2; h.b2 = n->b2;
    return h;
}

// in_il: interleaved batch in the layout net_plan(n, boards) announced. policy/value must have
// room for `boards` rounded up to the plan's group size.
int net_forward_il(ao_net* n, const float* in_il, int boards, float* policy, float* value, hipStream_t s) {
    if (!n->finalized) return n->fail("ao_net_finalize has not been called");
    NET_HIP(n, hipSetDevice(n->device));
    if (ensure_workspace(n, boards)) return 1;
    int group = 32, nchq = 0, nch = 1;
    bool heads_h16 = false;   // the separate head kernels read the split-fp16 layout
    net_plan(n, boards, &group, &nchq);
    const int mode = pick_mode(n, boards, &nch);
    const int groups = (boards + group - 1) / group;
    if (group == 1) {
        // per-board NHWC path: one wave per (16 cells, 16 couts, board)
        auto conv = [&](int layer, const float* in, int cqi, const float* res, float* out) {
            const int nw = (boards <= 4) ? 9 : 3;
            const dim3 grid(((n->A + 15) / 16) * (n->planes / 16) * boards), block(64 * nw);
            const float4* w4 = reinterpret_cast<const float4*>(layer == 0 ? n->conv0_w1 : n->conv_w[layer]);
            const bool timed = n->timing && layer > 0;
            const int idx = timed ? timer_begin(n, s) : 0;
            switch (n->B) {
#define AO_CELLS_LAUNCH2(W, Q, NWV)                                                                           \
    hipLaunchKernelGGL((k_conv_cells<W, Q, NWV>), grid, block, 0, s, reinterpret_cast<const float4*>(in), w4, \
                       reinterpret_cast<const float4*>(n->conv_sc[layer]),                                   \
                       reinterpret_cast<const float4*>(n->conv_sh[layer]), reinterpret_cast<const float4*>(res), \
                       reinterpret_cast<float4*>(out), cqi, n->planes, res ? 1 : 0)
#define AO_CELLS_LAUNCH(W, Q)                                                                                 \
    do {                                                                                                      \
        if (nw == 9) AO_CELLS_LAUNCH2(W, Q, 9);                                                               \
        else AO_CELLS_LAUNCH2(W, Q, 3);                                                                       \
    } while (0)
#define AO_BW_CASE(W)                                                                                         \
    case W:                                                                                                   \
        switch (cqi >> 2) {                                                                                   \
            case 1: AO_CELLS_LAUNCH(W, 1); break;                                                             \
            case 2: AO_CELLS_LAUNCH(W, 2); break;                                                             \
            case 4: AO_CELLS_LAUNCH(W, 4); break;                                                             \
            case 6: AO_CELLS_LAUNCH(W, 6); break;                                                             \
            default: AO_CELLS_LAUNCH(W, 8); break;                                                            \
        }                                                                                                     \
        break;
                AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
                AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
#undef AO_CELLS_LAUNCH
#undef AO_CELLS_LAUNCH2
            }
            if (timed) timer_end(n, idx, s);
        };
        conv(0, in_il, n->nchq1, nullptr, n->act_x);
        for (int i = 0; i < n->nb; ++i) {
            conv(1 + 2 * i, n->act_x, n->CQ, nullptr, n->act_t);
            conv(2 + 2 * i, n->act_t, n->CQ, n->act_x, n->act_x);
        }
        const size_t lds1 = heads_lds_floats(n->A, n->planes) * sizeof(float);
        hipLaunchKernelGGL(k_heads_board, dim3(boards), dim3(512), lds1, s, head_params(n),
                           reinterpret_cast<const float4*>(n->act_x), policy, value, n->A, n->planes);
        NET_HIP(n, hipGetLastError());
        return 0;
#ifdef AO_PROF
    } else if (group == 16 && mode == 5 && !(n->B <= 9 && (groups >= 192 || getenv("AO_FORCE_RESIDENT")))) {
#else
    } else if (group == 16 && mode == 5 && !(n->B <= 9 && groups >= 192)) {
#endif
        // split-fp16 trunk, one launch per conv: workgroup = (16-board group, row chunk, column tile). For batches
        // that cannot give every CU a whole group, and for boards wider than 9 (a staged row must fit LDS twice)
        const int nxt = n->B <= 9 ? 1 : (n->B + 4) / 5;
        int nchh = 1;
        {
            long best = -1;
            for (int c = 1; c <= n->B; ++c) {
                const long rounds = (static_cast<long>(groups) * c * nxt + n->num_cu - 1) / n->num_cu;
                const long cost = rounds * ((n->B + c - 1) / c + 1);   // rows of a chunk + its halo rows' staging
                if (best < 0 || cost < best) { best = cost; nchh = c; }
            }
        }
        static bool attr_l[16][2] = {};
        auto layer = [&](int l) -> int {
            LayerHArgs a;
            a.src = l == 0 ? static_cast<const void*>(in_il) : static_cast<const void*>((l & 1) ? n->act_x : n->act_t);
            a.dst = reinterpret_cast<uint4*>((l == 0 || !(l & 1)) ? n->act_x : n->act_t);
            a.layer.wh = n->convh_wh[l];
            a.layer.wl = n->convh_wl[l];
            a.layer.sc = reinterpret_cast<const float4*>(n->convh_sc[l]);
            a.layer.sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
            a.res = (l > 0 && !(l & 1)) ? 1 : 0;
            a.nch = nchh;
            const dim3 grid(groups * nchh * nxt), block(512);
            const bool timed = n->timing && l > 0;
            const int idx = timed ? timer_begin(n, s) : 0;
            switch (n->B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr int XT_ = (W <= 9) ? W : 5;                                                                          \
        constexpr int NX_ = (XT_ < W) ? XT_ + 2 : XT_;                                                                 \
        constexpr size_t lds_ = static_cast<size_t>(2) * NX_ * 4 * 2 * 1024;                                           \
        if (!attr_l[W][0]) {                                                                                           \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16h<W, XT_, 4, false>),               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16h<W, XT_, 4, true>),                \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            attr_l[W][0] = true;                                                                                       \
        }                                                                                                              \
        if (l == 0) hipLaunchKernelGGL((k_layer16h<W, XT_, 4, true>), grid, block, lds_, s, a);                       \
        else hipLaunchKernelGGL((k_layer16h<W, XT_, 4, false>), grid, block, lds_, s, a);                             \
    } break;
                AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
                AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
            }
            if (timed) timer_end(n, idx, s);
            return 0;
        };
        for (int l = 0; l <= 2 * n->nb; ++l)
            if (layer(l)) return 1;
        NET_HIP(n, hipGetLastError());
        heads_h16 = true;   // k_head_conv<true> / k_head_fc below
    } else if (group == 16 && mode == 5) {
        // split-fp16 resident trunk: one launch carries every 16-board group through conv1 (fp32 planes converted
        // while they are staged), the ResBlocks and the heads
        TrunkHArgs a;
        a.in0 = reinterpret_cast<const float4*>(in_il);
        a.bufA = reinterpret_cast<uint4*>(n->act_x);
        a.bufB = reinterpret_cast<uint4*>(n->act_t);
        a.nlayers = 1 + 2 * n->nb;
        a.CQ = n->CQ;
        a.COUT = n->planes;
        a.w3 = n->head_w3; a.sc3 = n->head_sc3; a.sh3 = n->head_sh3;
        a.wp_t = n->wp_t; a.bp = n->bp; a.w1_t = n->w1_t; a.b1 = n->b1; a.w2 = n->w2; a.b2 = n->b2;
        a.policy = policy;
        a.value = value;
        for (int l = 0; l < a.nlayers; ++l) {
            a.layers[l].wh = n->convh_wh[l];
            a.layers[l].wl = n->convh_wl[l];
            a.layers[l].sc = reinterpret_cast<const float4*>(n->convh_sc[l]);
            a.layers[l].sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
        }
        static bool attr_done[16] = {};
        const int idx = n->timing ? timer_begin(n, s) : 0;
        switch (n->B) {
#define AO_BW_CASE(W)                                                                                        \
    case W: {                                                                                                \
        constexpr size_t heads_ = (static_cast<size_t>(3) * 128 + 16 * 3 * W * W + 8 * 16 * W * W + 8 * 16 * 128) * 4;  \
        constexpr size_t lds_ = (static_cast<size_t>(2) * W * 4 * 2 * 1024 > heads_) ? static_cast<size_t>(2) * W * 4 * 2 * 1024 : heads_; \
        if (!attr_done[W]) {                                                                                 \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16h<W, 4>),                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_))); \
            attr_done[W] = true;                                                                             \
        }                                                                                                    \
        hipLaunchKernelGGL((k_trunk16h<W, 4>), dim3(groups), dim3(512), lds_, s, a);                         \
    } break;
            AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
#undef AO_BW_CASE
            default: return n->fail("split-fp16 trunk: board larger than 9x9");
        }
        if (n->timing) timer_end(n, idx, s);
        NET_HIP(n, hipGetLastError());
#ifdef AO_PROF
        if (getenv("AO_PROF_PRINT")) {
            unsigned long long h[96];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(ao_prof), sizeof(h));
            static const char* nm[12] = {"stage0", "slabs", "epilogue", "barrier", "last_epi+boundary", "heads", "conv1", "total",
                                         "s0:sync1", "s0:stage-issue", "s0:loads-land", "s0:sync2"};
            for (int k = 0; k < 12; ++k) {
                fprintf(stderr, "AO_PROF %-18s", nm[k]);
                for (int t = 0; t < 8; ++t) fprintf(stderr, " %9llu", h[t * 12 + k]);
                fprintf(stderr, "\n");
            }
        }
#endif
        return 0;  // the heads ran inside the resident kernel
    } else if (group == 16 && mode == 4) {
        auto layer = [&](int l, const float* in, int cqi, int cq_real, bool res, float* out) {
            LayerArgs a;
            a.src = reinterpret_cast<const float4*>(in);
            a.dst = reinterpret_cast<float4*>(out);
            a.layer.w = reinterpret_cast<const float4*>(l == 0 ? n->conv0_w16 : n->conv_w[l]);
            a.layer.sc = reinterpret_cast<const float4*>(n->conv_sc[l]);
            a.layer.sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
            a.res = res ? 1 : 0; a.cqi = cqi; a.cq_real = cq_real; a.COUT = n->planes; a.nch = nch;
            const dim3 grid(groups * nch), block(64 * (n->planes / 16));
            const bool timed = n->timing && l > 0;
            const int idx = timed ? timer_begin(n, s) : 0;
            switch (n->B) {
#define AO_BW_CASE(W)                                                                       \
    case W: {                                                                               \
        constexpr int XT_ = (W <= 9) ? W : 5;                                               \
        hipLaunchKernelGGL((k_layer16<W, XT_>), grid, block, 0, s, a);                       \
    } break;
                AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
                AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
            }
            if (timed) timer_end(n, idx, s);
        };
        layer(0, in_il, n->nchq16, (n->C + 3) / 4, false, n->act_x);
        for (int i = 0; i < n->nb; ++i) {
            layer(1 + 2 * i, n->act_x, n->CQ, n->CQ, false, n->act_t);
            layer(2 + 2 * i, n->act_t, n->CQ, n->CQ, true, n->act_x);   // + x, in place
        }
        // heads below (k_head_conv / k_head_fc on the 16-board layout)
    } else if (group == 16) {
        static bool lds_attr_done[16] = {};
        switch (n->B) {
#define AO_BW_CASE(W)                                                                                        \
    case W: {                                                                                                \
        constexpr int XT_ = (W <= 9) ? W : 5;                                                                \
        if (!lds_attr_done[W]) {                                                                             \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16<W, XT_, 1>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));          \
        }                                                                                                    \
        lds_attr_done[W] = true;                                                                             \
        launch_trunk16<W>(n, in_il, groups, policy, value, s);                                               \
    } break;
            AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
            AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
        }
        NET_HIP(n, hipGetLastError());
        return 0;  // the heads ran inside the resident kernel
    } else {
        auto conv = [&](int layer, const float* in, int cqi, const float* res, float* out) {
            switch (n->B) {
#define AO_BW_CASE(W) case W: launch_conv<W>(n, layer, in, cqi, res, out, groups, s); break;
                AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
                AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
            }
        };
        conv(0, in_il, n->nchq32, nullptr, n->act_x);                      // conv1 + bn1 + relu
        for (int i = 0; i < n->nb; ++i) {                                   // ResBlock (model.py:22-31)
            conv(1 + 2 * i, n->act_x, n->CQ, nullptr, n->act_t);
            conv(2 + 2 * i, n->act_t, n->CQ, n->act_x, n->act_x);
        }
    }
    const int ppb = 256 / group;
    const int nchunk = (n->A + ppb - 1) / ppb;
    if (heads_h16)
        hipLaunchKernelGGL(k_head_conv<true>, dim3(groups * nchunk), dim3(256), 3 * n->planes * sizeof(float), s,
                           reinterpret_cast<const float4*>(n->act_x), n->head_w3, n->head_sc3, n->head_sh3, n->hbuf,
                           n->A, n->CQ, group);
    else
        hipLaunchKernelGGL(k_head_conv<false>, dim3(groups * nchunk), dim3(256), 3 * n->planes * sizeof(float), s,
                           reinterpret_cast<const float4*>(n->act_x), n->head_w3, n->head_sc3, n->head_sh3, n->hbuf,
                           n->A, n->CQ, group);
    const size_t lds = (static_cast<size_t>(4) * n->A + n->planes + 8) * sizeof(float);
    hipLaunchKernelGGL(k_head_fc, dim3(groups * group), dim3(256), lds, s, n->hbuf, n->wp_t, n->bp, n->w1_t,
                       n->b1, n->w2, n->b2, policy, value, n->A, n->planes);
    NET_HIP(n, hipGetLastError());
    return 0;
}

}  // namespace ao

extern "C" {

const char* ao_net_last_error(const ao_net* n) { return n ? n->err.c_str() : g_net_create_error.c_str(); }

int ao_net_create(int n_block, int inplanes, int planes, int board, int device, ao_net** out) {
    if (!out) return 1;
    *out = nullptr;
    auto bad = [&](const char* m) { g_net_create_error = m; return 1; };
    if (n_block < 0 || n_block > 64) return bad("n_block out of range");
    if (inplanes < 1 || inplanes > 12) return bad("inplanes must be in 1..12");
    if (planes < 32 || planes > 128 || planes % 32) return bad("planes must be 32, 64, 96 or 128");
    if (board < 3 || board > ao::kMaxBoard) return bad("board must be in 3..15");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad("no HIP device available");
    if (device < 0 || device >= ndev) return bad("device ordinal out of range");
    ao_net* n = new ao_net();
    n->nb = n_block; n->C = inplanes; n->planes = planes; n->B = board; n->A = board * board;
    n->device = device;
    n->nchq32 = (((inplanes + 3) / 4) + 1) & ~1;  // consumed in pairs (32x32x2 MFMA, two quads per step)
    n->nchq16 = (((inplanes + 3) / 4) + 7) & ~7;  // 16 channels per k-step, steps taken in pairs
    n->nchq1 = (((inplanes + 3) / 4) + 3) & ~3;   // 16 channels per k-step
    n->CQ = planes / 4;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            n->num_cu = prop.multiProcessorCount;
    }
    *out = n;
    return 0;
}

void ao_net_destroy(ao_net* n) {
    if (!n) return;
    hipSetDevice(n->device);
    hipDeviceSynchronize();
    for (void* p : n->allocs) hipFree(p);
    for (auto e : n->ev0) hipEventDestroy(e);
    for (auto e : n->ev1) hipEventDestroy(e);
    delete n;
}

int ao_net_set_mode(ao_net* n, int mode) {
    if (mode < 0 || mode > 5)
        return n->fail("mode must be 0 (auto), 1 (layer kernels), 2 (group-resident trunk), 3 (per-board), 4 (row-chunked) or 5 (split-fp16 trunk)");
    if (mode == 5 && !h16_supported(n))
        return n->fail("mode 5 (split-fp16 MFMA trunk) needs 128 planes and at least one ResBlock");
    n->mode = mode;
    return 0;
}

int ao_net_set_param(ao_net* n, const char* name, const float* data, int64_t numel) {
    if (!name || (!data && numel > 0) || numel < 0) return n->fail("bad argument");
    n->params[name] = std::vector<float>(data, data + numel);
    n->finalized = false;
    return 0;
}

static int get_param(ao_net* n, const std::string& name, size_t numel, const std::vector<float>** out) {
    auto it = n->params.find(name);
    if (it == n->params.end()) return n->fail("missing parameter " + name);
    if (it->second.size() != numel)
        return n->fail("parameter " + name + " has " + std::to_string(it->second.size()) + " elements, expected " +
                       std::to_string(numel));
    *out = &it->second;
    return 0;
}

// BatchNorm2d in eval mode (eps 1e-5): y = x*scale + shift
static int fold_bn(ao_net* n, const std::string& prefix, int c, std::vector<float>* sc, std::vector<float>* sh) {
    const std::vector<float>*w, *b, *m, *v;
    if (get_param(n, prefix + ".weight", c, &w) || get_param(n, prefix + ".bias", c, &b) ||
        get_param(n, prefix + ".running_mean", c, &m) || get_param(n, prefix + ".running_var", c, &v))
        return 1;
    sc->resize(c); sh->resize(c);
    for (int i = 0; i < c; ++i) {
        const double s = static_cast<double>((*w)[i]) / std::sqrt(static_cast<double>((*v)[i]) + 1e-5);
        (*sc)[i] = static_cast<float>(s);
        (*sh)[i] = static_cast<float>(static_cast<double>((*b)[i]) - static_cast<double>((*m)[i]) * s);
    }
    return 0;
}

// OIHW -> [tap][cq][cout][4], input channels zero-padded to 4*cqi
static std::vector<float> pack_conv(const std::vector<float>& w, int cout, int cin, int cqi) {
    std::vector<float> packed(static_cast<size_t>(9) * cqi * cout * 4, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t)
                packed[((static_cast<size_t>(t) * cqi + (ci >> 2)) * cout + co) * 4 + (ci & 3)] =
                    w[(static_cast<size_t>(co) * cin + ci) * 9 + t];
    return packed;
}

// OIHW fp32 -> two fp16 planes [tap][c32][tile][oct 4][cout 16][8]: w * 2^s = high + low
static void pack_conv_h(const std::vector<float>& w, int cout, int cin, int s, std::vector<uint16_t>* hi,
                        std::vector<uint16_t>* lo) {
    const int nc32 = cin / 32, nt = cout / 16;
    hi->assign(static_cast<size_t>(9) * nc32 * nt * 64 * 8, 0);
    lo->assign(hi->size(), 0);
    const float scale = std::ldexp(1.0f, s);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const float v = w[(static_cast<size_t>(co) * cin + ci) * 9 + t] * scale;
                const _Float16 h = static_cast<_Float16>(v);
                const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
                const size_t idx = ((((static_cast<size_t>(t) * nc32 + ci / 32) * nt + co / 16) * 4 + (ci % 32) / 8) * 16 +
                                    co % 16) * 8 + ci % 8;
                std::memcpy(&(*hi)[idx], &h, 2);
                std::memcpy(&(*lo)[idx], &l, 2);
            }
}


int ao_net_finalize(ao_net* n) {
    NET_HIP(n, hipSetDevice(n->device));
    NET_HIP(n, hipDeviceSynchronize());
    for (void* p : n->allocs) hipFree(p);
    n->allocs.clear();
    n->conv_w.clear(); n->conv_sc.clear(); n->conv_sh.clear();
    n->convh_wh.clear(); n->convh_wl.clear(); n->convh_sc.clear();
    n->ws_boards = 0;
    const int P = n->planes, A = n->A;
    auto add_conv = [&](const std::string& wname, const std::string& bnname, int cin, int cqi) -> int {
        const std::vector<float>* w;
        if (get_param(n, wname, static_cast<size_t>(P) * cin * 9, &w)) return 1;
        std::vector<float> sc, sh;
        if (fold_bn(n, bnname, P, &sc, &sh)) return 1;
        float *dw, *dsc, *dsh;
        if (upload(n, &dw, pack_conv(*w, P, cin, cqi)) || upload(n, &dsc, sc) || upload(n, &dsh, sh)) return 1;
        n->conv_w.push_back(dw); n->conv_sc.push_back(dsc); n->conv_sh.push_back(dsh);
        return 0;
    };
    if (add_conv("conv1.weight", "bn1", n->C, n->nchq32)) return 1;
    {
        const std::vector<float>* w;
        if (get_param(n, "conv1.weight", static_cast<size_t>(P) * n->C * 9, &w)) return 1;
        if (upload(n, &n->conv0_w16, pack_conv(*w, P, n->C, n->nchq16))) return 1;
        if (upload(n, &n->conv0_w1, pack_conv(*w, P, n->C, n->nchq1))) return 1;
    }
    for (int i = 0; i < n->nb; ++i) {
        const std::string pre = "layers." + std::to_string(i);
        if (add_conv(pre + ".conv1.weight", pre + ".bn1", P, n->CQ)) return 1;
        if (add_conv(pre + ".conv2.weight", pre + ".bn2", P, n->CQ)) return 1;
    }
    if (h16_supported(n)) {
        for (int l = 0; l <= 2 * n->nb; ++l) {
            const std::string pre = "layers." + std::to_string(l ? (l - 1) / 2 : 0);
            const std::string cname = l == 0 ? std::string("conv1.weight") : pre + ((l & 1) ? ".conv1.weight" : ".conv2.weight");
            const std::string bname = l == 0 ? std::string("bn1") : pre + ((l & 1) ? ".bn1" : ".bn2");
            const int cin = l == 0 ? n->C : P;
            const std::vector<float>* w0;
            if (get_param(n, cname, static_cast<size_t>(P) * cin * 9, &w0)) return 1;
            // conv1: input channels zero-padded to one 32-channel block
            std::vector<float> wpad;
            const std::vector<float>* w = w0;
            const int cinp = l == 0 ? 32 : P;
            if (l == 0) {
                wpad.assign(static_cast<size_t>(P) * 32 * 9, 0.f);
                for (int co = 0; co < P; ++co)
                    for (int ci = 0; ci < n->C; ++ci)
                        for (int t = 0; t < 9; ++t) wpad[(static_cast<size_t>(co) * 32 + ci) * 9 + t] = (*w0)[(static_cast<size_t>(co) * n->C + ci) * 9 + t];
                w = &wpad;
            }
            float mx = 0.f;
            for (float v : *w) mx = std::max(mx, std::fabs(v));
            // power-of-two pre-scale: largest |w| lands in [4, 8), so the low halves are normal fp16 numbers
            const int sft = (mx > 0.f && std::isfinite(mx)) ? 2 - static_cast<int>(std::floor(std::log2(mx))) : 0;
            std::vector<uint16_t> hi, lo;
            pack_conv_h(*w, P, cinp, sft, &hi, &lo);
            std::vector<float> sc, sh;
            if (fold_bn(n, bname, P, &sc, &sh)) return 1;
            for (float& v : sc) v = std::ldexp(v, -sft);
            void *dh = nullptr, *dl = nullptr;
            float* dsc = nullptr;
            NET_HIP(n, hipMalloc(&dh, hi.size() * 2));
            n->allocs.push_back(dh);
            NET_HIP(n, hipMalloc(&dl, lo.size() * 2));
            n->allocs.push_back(dl);
            NET_HIP(n, hipMemcpy(dh, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
            NET_HIP(n, hipMemcpy(dl, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
            if (upload(n, &dsc, sc)) return 1;
            n->convh_wh.push_back(static_cast<uint4*>(dh));
            n->convh_wl.push_back(static_cast<uint4*>(dl));
            n->convh_sc.push_back(dsc);
        }
    }
    // heads
    const std::vector<float>*pw, *vw, *fcw, *fcb, *f1w, *f1b, *f2w, *f2b;
    if (get_param(n, "policy_head.policy_head.weight", 2 * P, &pw) ||
        get_param(n, "value_head.value_head.weight", P, &vw) ||
        get_param(n, "policy_head.policy_fc.weight", static_cast<size_t>(A) * 2 * A, &fcw) ||
        get_param(n, "policy_head.policy_fc.bias", A, &fcb) ||
        get_param(n, "value_head.value_fc1.weight", static_cast<size_t>(P) * A, &f1w) ||
        get_param(n, "value_head.value_fc1.bias", P, &f1b) ||
        get_param(n, "value_head.value_fc2.weight", P, &f2w) || get_param(n, "value_head.value_fc2.bias", 1, &f2b))
        return 1;
    std::vector<float> w3(static_cast<size_t>(3) * P);
    std::copy(pw->begin(), pw->end(), w3.begin());
    std::copy(vw->begin(), vw->end(), w3.begin() + 2 * P);
    std::vector<float> psc, psh, vsc, vsh;
    if (fold_bn(n, "policy_head.policy_bn", 2, &psc, &psh) || fold_bn(n, "value_head.value_bn", 1, &vsc, &vsh))
        return 1;
    std::vector<float> sc3 = {psc[0], psc[1], vsc[0]}, sh3 = {psh[0], psh[1], vsh[0]};
    std::vector<float> wp_t(static_cast<size_t>(2) * A * A), w1_t(static_cast<size_t>(A) * P);
    for (int a = 0; a < A; ++a)
        for (int j = 0; j < 2 * A; ++j) wp_t[static_cast<size_t>(j) * A + a] = (*fcw)[static_cast<size_t>(a) * 2 * A + j];
    for (int o = 0; o < P; ++o)
        for (int j = 0; j < A; ++j) w1_t[static_cast<size_t>(j) * P + o] = (*f1w)[static_cast<size_t>(o) * A + j];
    if (upload(n, &n->head_w3, w3) || upload(n, &n->head_sc3, sc3) || upload(n, &n->head_sh3, sh3) ||
        upload(n, &n->wp_t, wp_t) || upload(n, &n->bp, *fcb) || upload(n, &n->w1_t, w1_t) ||
        upload(n, &n->b1, *f1b) || upload(n, &n->w2, *f2w) || upload(n, &n->b2, *f2b))
        return 1;
    NET_HIP(n, hipDeviceSynchronize());
    n->finalized = true;
    return 0;
}

int ao_net_forward(ao_net* n, const float* dev_planes_nchw, int batch, float* dev_policy, float* dev_value,
                   void* stream) {
    if (!n->finalized) return n->fail("ao_net_finalize has not been called");
    if (batch < 1) return n->fail("batch must be >= 1");
    NET_HIP(n, hipSetDevice(n->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ao::ensure_workspace(n, batch)) return 1;
    int group = 32, nchq = 0;
    ao::net_plan(n, batch, &group, &nchq);
    const int boards = (batch + group - 1) / group * group;
    // the heads write rows for the padding boards too: run into scratch unless the batch is whole
    float* pol = (boards == batch) ? dev_policy : n->tmp_p;
    float* val = (boards == batch) ? dev_value : n->tmp_v;
    const size_t total = static_cast<size_t>(boards) * n->A;
    hipLaunchKernelGGL(ao::k_nchw_to_il, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s,
                       dev_planes_nchw, reinterpret_cast<float4*>(n->il_in), batch, n->C, n->A, nchq, boards, group);
    if (ao::net_forward_il(n, n->il_in, batch, pol, val, s)) return 1;
    if (boards != batch) {
        NET_HIP(n, hipMemcpyAsync(dev_policy, n->tmp_p, sizeof(float) * batch * n->A, hipMemcpyDeviceToDevice, s));
        NET_HIP(n, hipMemcpyAsync(dev_value, n->tmp_v, sizeof(float) * batch, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int ao_net_conv_timing(ao_net* n, int enable, double* ms_total, int64_t* launches) {
    NET_HIP(n, hipSetDevice(n->device));
    if (n->ev0.empty() && enable) {
        n->ev0.resize(ao_net::kRing);
        n->ev1.resize(ao_net::kRing);
        for (int i = 0; i < ao_net::kRing; ++i) {
            NET_HIP(n, hipEventCreate(&n->ev0[i]));
            NET_HIP(n, hipEventCreate(&n->ev1[i]));
        }
    }
    harvest(n, n->ring_count);
    if (ms_total) *ms_total = n->ms_total;
    if (launches) *launches = n->launches;
    n->ms_total = 0.0;
    n->launches = 0;
    n->timing = enable != 0;
    return 0;
}

int ao_net_dominant_kernel(ao_net* n, int boards, char* name, int name_cap, double* flop_per_launch) {
    int group = 32, nchq = 0;
    ao::net_plan(n, boards, &group, &nchq);
    const int padded = (boards + group - 1) / group * group;
    const double conv = 2.0 * n->A * 9.0 * n->planes * n->planes * padded;   // one planes->planes 3x3 conv
    const double conv1 = 2.0 * n->A * 9.0 * n->C * n->planes * padded;
    std::string nm;
    double f;
    if (group == 1) {
        nm = "k_conv_cells<" + std::to_string(n->B) + "> (one 3x3 conv, per-board NHWC, fp32 MFMA 16x16x4)";
        f = conv;
 } else if (group == 16 && ao::pick_mode_public(n, boards) == 4) {
        nm = "k_layer16<" + std::to_string(n->B) + "> (one 3x3 conv per launch, 16-board groups x row chunks, fp32 MFMA 16x16x4)";
        f = conv;
    } else if (group == 16 && ao::pick_mode_public(n, boards) == 5 && !(n->B <= 9 && (boards + 15) / 16 >= 192)) {
        nm = "k_layer16h<" + std::to_string(n->B) + "> (one 3x3 conv per launch as split-fp16 MFMA 16x16x32 (3 products, fp32 accumulate), "
             "16-board groups x row chunks x column tiles)";
        f = conv;
    } else if (group == 16 && ao::pick_mode_public(n, boards) == 5) {
        nm = "k_trunk16h<" + std::to_string(n->B) + "> (conv1 + " + std::to_string(2 * n->nb) +
             " 3x3 convs as split-fp16 MFMA 16x16x32 (3 products, fp32 accumulate), one resident launch)";
        f = conv1 + 2.0 * n->nb * conv;
    } else if (group == 16) {
        nm = "k_trunk16<" + std::to_string(n->B) + "> (conv1 + " + std::to_string(2 * n->nb) +
             " 3x3 convs, one launch, fp32 MFMA 16x16x4)";
        f = conv1 + 2.0 * n->nb * conv;
    } else {
        nm = "k_conv3x3<" + std::to_string(n->B) + "> (one 3x3 " + std::to_string(n->planes) + "->" +
             std::to_string(n->planes) + " conv, fp32 MFMA 32x32x2)";
        f = conv;
    }
    if (name && name_cap > 0) {
        std::strncpy(name, nm.c_str(), static_cast<size_t>(name_cap) - 1);
        name[name_cap - 1] = 0;
    }
    if (flop_per_launch) *flop_per_launch = f;
    return 0;
}

}  // extern "C"
