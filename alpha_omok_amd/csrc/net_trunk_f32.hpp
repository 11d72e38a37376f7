// net_trunk_f32.hpp -- the conv stack on fp32 MFMAs: k_conv3x3 (32-board groups, one launch per conv), the
// group-resident k_trunk16 (16-board groups, all layers + heads in one launch) and its per-layer form k_layer16.
// Included by net.hip after net_common.hpp.
//
// k_conv3x3:
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
#pragma once

namespace ao {

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
// H16: 1 = the activations are in the split-fp16 layout of k_trunk16h (x = high half + low half), 2 = its 3-byte form
// (fp16 high half + one low byte, kPairBytes(1) in net_trunk_h16.hpp), 0 = fp32
template <int BW, int H16 = 0, typename Args = TrunkArgs>
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
    if (H16 != 0) {
        // split-fp16 layouts: a lane's 8 channels of one (cell, 32-channel block, oct) are ONE 16-byte slot of the high
        // fragment (+ 16 / 8 bytes of the low one). All slots of two blocks are requested before the first is used: written
        // quad by quad this loop was a chain of 32 dependent round trips per cell to data that has mostly left L2 by
        // now (0.10 M of the launch's 2.75 M cycles, profiles/r3a_trunk16h_bytes_ko.txt).
        const int b = tid & 15;
        const int nblk = CQ >> 3;
        for (int cell = tid >> 4; cell < A; cell += nthreads >> 4) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            constexpr int CB = 4;   // blocks in flight (the split-fp16 kernels are built for 128 planes = 4 blocks)
            for (int c0 = 0; c0 < nblk; c0 += CB) {
                uint4 hv[CB][4], lv[CB][4];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) {
                        const int c = c0 + cb < nblk ? c0 + cb : nblk - 1;
                        const char* base = reinterpret_cast<const char*>(act) +
                                           ((gbase + cell) * nblk + c) * (H16 == 2 ? 1536 : 2048) + (oc * 16 + b) * 16;
                        hv[cb][oc] = *reinterpret_cast<const uint4*>(base);
                        if (H16 == 2) {
                            const uint2 t = *reinterpret_cast<const uint2*>(base + 1024 - (oc * 16 + b) * 8);
                            lv[cb][oc] = make_uint4(t.x, t.y, 0u, 0u);
                        } else {
                            lv[cb][oc] = *reinterpret_cast<const uint4*>(base + 1024);
                        }
                    }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    if (c0 + cb >= nblk) continue;
#pragma unroll
                    for (int oc = 0; oc < 4; ++oc) {
                        const unsigned hw[4] = {hv[cb][oc].x, hv[cb][oc].y, hv[cb][oc].z, hv[cb][oc].w};
                        const unsigned lw[4] = {lv[cb][oc].x, lv[cb][oc].y, lv[cb][oc].z, lv[cb][oc].w};
                        float x[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const unsigned hb = (hw[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                            if (H16 == 2) {
                                const unsigned t = (hb << 8) | ((lw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                                x[k] = __uint_as_float(t ? (t << 5) + 0x38000000u : 0u);
                            } else {
                                const unsigned lb = (lw[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                                x[k] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(hb))) +
                                       static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(lb)));
                            }
                        }
                        const int ch = ((c0 + cb) * 4 + oc) * 8;   // first channel of this slot
                        const float* w0 = s_w3 + ch;
                        const float* w1 = s_w3 + planes + ch;
                        const float* w2 = s_w3 + 2 * planes + ch;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            a0 = fmaf(x[k], w0[k], a0);
                            a1 = fmaf(x[k], w1[k], a1);
                            a2 = fmaf(x[k], w2[k], a2);
                        }
                    }
                }
            }
            // s_h is [3A][16 boards] here: the FC sweeps below read the 16 boards of an input index as four 16-byte
            // broadcasts instead of sixteen 4-byte ones (the sweeps were LDS-instruction bound: ~60 k of the heads' 88 k cycles)
            s_h[(cell) * GB + b] = fmaxf(fmaf(a0, a.sc3[0], a.sh3[0]), 0.f);
            s_h[(A + cell) * GB + b] = fmaxf(fmaf(a1, a.sc3[1], a.sh3[1]), 0.f);
            s_h[(2 * A + cell) * GB + b] = fmaxf(fmaf(a2, a.sc3[2], a.sh3[2]), 0.f);
        }
    } else {
        const int b = tid & 15;
        for (int cell = tid >> 4; cell < A; cell += nthreads >> 4) {
            const float4* xp = act + ((gbase + cell) * CQ) * GB + b;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int cq = 0; cq < CQ; ++cq) {
                const float4 x = xp[static_cast<size_t>(cq) * GB];
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
#pragma unroll 4
                    for (int j = j0; j < j1; ++j) {
                        const float w = a.wp_t[static_cast<size_t>(j) * A + o];
                        const float4* hq = reinterpret_cast<const float4*>(s_h + j * GB);
#pragma unroll
                        for (int q = 0; q < GB / 4; ++q) {
                            const float4 h4 = hq[q];
                            acc[4 * q] = fmaf(w, h4.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w, h4.y, acc[4 * q + 1]);
                            acc[4 * q + 2] = fmaf(w, h4.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w, h4.w, acc[4 * q + 3]);
                        }
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
#pragma unroll 4
                    for (int j = j0; j < j1; ++j) {
                        const float w = a.w1_t[static_cast<size_t>(j) * planes + o];
                        const float4* hq = reinterpret_cast<const float4*>(s_h + (2 * A + j) * GB);
#pragma unroll
                        for (int q = 0; q < GB / 4; ++q) {
                            const float4 h4 = hq[q];
                            acc[4 * q] = fmaf(w, h4.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w, h4.y, acc[4 * q + 1]);
                            acc[4 * q + 2] = fmaf(w, h4.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w, h4.w, acc[4 * q + 3]);
                        }
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

// One conv layer per launch for medium batches: a 16-board group is split into `nch` row chunks,
// one workgroup each, so 64 groups x 4 chunks still give every CU one workgroup. The chunk runs
// the same sliding-window code over its rows (plus one halo input row on each side); the launch
// boundary is the synchronisation between layers, nothing is exchanged inside a launch.
struct LayerArgs {
    const float4* src;
    float4* dst;
    TrunkLayer layer;
    int res, cqi, cq_real, COUT, nch;
    int nsplit;   // workgroups per (group, row chunk): eight 16-channel output tiles each (networks wider than 128 planes)
};

template <int BW, int XT>
__global__ __launch_bounds__(512, 1) void k_layer16(LayerArgs a) {
    constexpr int A = BW * BW;
    const int half = blockIdx.x % a.nsplit;
    const int rest = blockIdx.x / a.nsplit;
    const int grp = rest / a.nch;
    const int c = rest - grp * a.nch;
    const int base = BW / a.nch, extra = BW % a.nch;
    const int yb = c * base + (c < extra ? c : extra);
    const int ye = yb + base + (c < extra ? 1 : 0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int ct0 = half * 8 + wave;                  // this wave's 16-channel output tile
    if (ct0 * 16 >= a.COUT) return;                   // (160 / 192 / 224 planes: the second workgroup has fewer tiles; no barriers below)
    trunk_layer<BW, XT, 1>(a.src, a.dst, a.layer.w, a.layer.sc, a.layer.sh, a.res != 0, a.cqi, a.cq_real, a.COUT,
                           static_cast<size_t>(grp) * A, ct0, lane >> 4, lane & 15, yb, ye);
}

}  // namespace ao
