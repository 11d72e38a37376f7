// net_small.hpp -- small-batch (per-board) conv kernel, the stand-alone head kernels and the NCHW -> interleaved
// plane repack. Included by net.hip after net_trunk_h16.hpp.
#pragma once

namespace ao {

// ----------------------------------------------------------------------------------------------
// Small batches (a drop-in ZeroAgent has ONE game): boards cannot fill the MFMA N dimension, so
// the CELLS of one board do:   D[cout 16][cell 16] += Wt[cout 16][k 4] * X[k 4][cell 16]
// on the plain per-board NHWC layout act[board][cell][channel]. One wave per (16 cells, 16 output
// channels, board): a 9x9x128 layer is 48 independent waves of 288 MFMAs (~4 us) instead of nine
// workgroups of ~130 us, which is what matters when 400 evaluations run back to back.
// Out-of-board taps are zero-filled per lane (cells of a tile differ in position).
// ----------------------------------------------------------------------------------------------
// NCQG = 16-channel k-steps per tap; NW = waves per tile (9: one tap each, 3: one tap row each).
// The tile code is conv_cells_tile (net_device.hpp).
template <int BW, int NCQG, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_conv_cells(const float4* __restrict__ in, const float4* __restrict__ wt,
                                                   const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                   const float4* res, float4* out, int CQI, int COUT, int relu_res) {
    // the tile's board rows in LDS for the one-tap-per-wave form (a handful of boards: one workgroup per CU anyway);
    // the three-taps-per-wave form of larger batches keeps loading from global memory -- 24 KB of LDS per workgroup would
    // cut the workgroups per CU from ten to six (16 games: 105 -> 142 us per simulation, measured)
    constexpr bool LDSX = NW == 9;
    __shared__ float s_red[(NW - 1) * 64 * 4];
    __shared__ __attribute__((aligned(16))) float4 s_x[LDSX ? conv_cells_lds_quads(BW, NCQG) : 1];
    // 1-D grid with the output-channel tile fastest: workgroups are dispatched round-robin over
    // the 8 XCDs, so (for 8 tiles) XCD x only ever reads the weights of tile x -- 1/8 of the
    // network per L2, which then stays resident from one evaluation to the next (the whole net
    // is 5 MB, an XCD's L2 4 MB).
    const int ntile = COUT >> 4;
    const int ct = blockIdx.x % ntile;
    const int rest = blockIdx.x / ntile;
    constexpr int NCT = (BW * BW + 15) / 16;
    conv_cells_tile<BW, NCQG, NW, LDSX>(in, wt, scale, shift, res, out, CQI, COUT, relu_res, ct, rest % NCT, rest / NCT, s_red, s_x);
}

// The same tile on split-fp16 MFMAs (conv_cells_tile_h): 128-plane trunk layers, `bpw` boards per workgroup in turn. With
// the activations coming from LDS the fp32 MFMAs of the nine waves (9 x 32 x 32 cycles on one CU's four matrix pipes)
// were what was left of the launch; three 16-cycle fp16 MFMAs per 32-channel block replace eight 32-cycle fp32 ones.
template <int BW, int NCQG, bool W16>
__device__ __forceinline__ void conv_cells_h_body(const float4* __restrict__ in, const uint4* __restrict__ wh,
                                                             const uint4* __restrict__ wl, const float4* __restrict__ scale,
                                                             const float4* __restrict__ shift, const float4* res, float4* out,
                                                             int CQI, int COUT, int relu_res, int* ovf, int boards, int bpw, float* s_red, float4* s_x) {
    // grid: output-channel tile fastest (workgroups go round robin over the 8 XCDs: XCD x only ever reads the weights of
    // tile x), then the cell tile, then the chunk of `bpw` boards this workgroup walks
    const int ntile = COUT >> 4;
    const int ct = blockIdx.x % ntile;
    const int rest = blockIdx.x / ntile;
    constexpr int NCT = (BW * BW + 15) / 16;
    const int b0 = (rest / NCT) * bpw;
    const int nb = boards - b0 < bpw ? boards - b0 : bpw;
    conv_cells_tile_h<BW, NCQG, 12, W16>(in, wh, wl, scale, shift, res, out, CQI, COUT, relu_res, ct, rest % NCT, b0, nb, s_red, s_x, ovf);
}

template <int BW, int NCQG>
__global__ __launch_bounds__(64 * 12, 1) void k_conv_cells_h(const float4* __restrict__ in, const uint4* __restrict__ wh,
                                                             const uint4* __restrict__ wl, const float4* __restrict__ scale,
                                                             const float4* __restrict__ shift, const float4* res, float4* out,
                                                             int CQI, int COUT, int relu_res, int* ovf, int boards, int bpw) {
    __shared__ float s_red[2 * 11 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float4 s_x[conv_cells_lds_quads(BW, NCQG)];
    conv_cells_h_body<BW, NCQG, false>(in, wh, wl, scale, shift, res, out, CQI, COUT, relu_res, ovf, boards, bpw, s_red, s_x);
}
// the two-product form (fp16 weights): no low weight halves are read
template <int BW, int NCQG>
__global__ __launch_bounds__(64 * 12, 1) void k_conv_cells_h_w16(const float4* __restrict__ in, const uint4* __restrict__ wh,
                                                                 const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                                 const float4* res, float4* out, int CQI, int COUT, int relu_res, int* ovf,
                                                                 int boards, int bpw) {
    __shared__ float s_red[2 * 11 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float4 s_x[conv_cells_lds_quads(BW, NCQG)];
    conv_cells_h_body<BW, NCQG, true>(in, wh, nullptr, scale, shift, res, out, CQI, COUT, relu_res, ovf, boards, bpw, s_red, s_x);
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
        // (unrolled: 16 coalesced weight loads in flight per thread instead of one -- 70 -> 32 us for 1024 15x15 boards, r3r)
#pragma unroll 16
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
#pragma unroll 16
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
__global__ __launch_bounds__(1024) void k_heads_board(HeadParams h, const float4* __restrict__ act,
                                                     float* __restrict__ policy, float* __restrict__ value, int A,
                                                     int planes) {
    extern __shared__ __attribute__((aligned(16))) float s_hb[];
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
