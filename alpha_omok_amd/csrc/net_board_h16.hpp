// net_board_h16.hpp -- k_boardh<BW>: the trunk of ONE board resident in LDS, cells as the MFMA N dimension (round 5).
//
// Boards wider than 9 (15x15: BASELINE configs[4]) had no multi-layer kernel: a 16-board group's input row does not fit LDS twice,
// so k_layer16h runs one launch per conv, every workgroup pays halo columns, a staging round trip and an epilogue per row step,
// and the activations make a round trip through HBM per layer (model.py:13-31,97-104: 21 convs for 10 blocks). Here the
// decomposition is turned around:
//   * a workgroup carries ONE board through all trunk convs; the board's activations live in LDS as the two fp16 halves of the
//     split-fp16 scheme (x = xh + xl, net_trunk_h16.hpp), rows padded to 16 cells: fragment (row, 32-channel block, half) =
//     1 KB = one B operand of v_mfma_f32_16x16x32_f16 with the row's CELLS as N (lane = k-octet * 16 + cell holds 8 consecutive
//     channels). 15 x 4 x 2 KB = 120 KB; the 16th cell of a row is a zero pad;
//   * a tap's column shift dx = +-1 is a DPP row shift of the fragment's registers (a "row" of the DPP network is exactly the 16
//     lanes of one k-octet; the pad cell / bound_ctrl supply the zeros off the board), the row shift dy = +-1 is a whole-tile
//     shift: input row r feeds output rows r-1, r, r+1, and rows off the board are skipped tile-uniformly -- no halo, no
//     MFMA on padding except the 16th cell (1.07 x the algorithmic MFMAs; k_layer16h<15> issues 15 x 18 / (15 x 15) = 1.2 x on its
//     column tiles + halo);
//   * wave w owns cout tile w (8 waves, two per SIMD) and keeps the accumulators of ALL 15 output rows (60 registers); the
//     contraction runs block-outermost: for each 32-channel input block the wave has its 9 taps x {high, low} weight fragments in
//     registers (72; from L2: every workgroup of the chip reads the same 590 KB per layer) and sweeps the board's rows -- 2 LDS
//     reads, 4 DPP shifts and 27 MFMAs per (row, block), no barrier inside a layer: the waves drift. (First form: weights streamed
//     per (block, tap row) slab, double-buffered, 9 MFMAs per read + shift: 3 x the LDS reads and DPP moves per MFMA, 9 % slower --
//     the chip is power-limited on this kernel, 2.1 of 2.4 GHz, and what the MFMAs do not need costs clock: profiles/r5o_*.)
//   * layer boundary = barrier (everyone has read the input) -> epilogue (BatchNorm, residual, ReLU, split, 8-byte LDS writes in
//     place) -> barrier. The ResBlock input of a wave's own cout tile stays in its registers (60) across the block's two convs;
//   * conv1 runs in the same launch on the engine's bit planes (ao_search; 90 MFMAs per wave, K = tap * 8 + plane) -- or, for
//     ao_net_forward's arbitrary float planes, as k_layer16h<BW, ..., KIND 1> before it, its output gathered from the group layout
//     (a group's 16 boards go to 16 workgroups of one XCD: they share every line). The heads' 1x1 convs (128 -> 3 channels) are
//     taken from the last layer's registers (round 6: the trunk's output -- 118 MB of fp32 NHWC per 1024 boards -- is not written
//     any more); the FC layers stay a batched launch (k_head_fc: 405 KB of policy_fc weights per board want the whole chip, not one
//     workgroup -- run inside this kernel by heads_board_dev the heads cost 70 us per board): 23 launches become 2.
// Same arithmetic family as the other split-fp16 kernels (3 products per multiply-add, fp32 accumulate, weights pre-scaled by a
// power of two per layer); the summation order over taps / blocks differs, so results agree to fp32 rounding, not bit for bit.
#pragma once

// Timing knock-outs (-DAO_BKO=n, WRONG RESULTS, experiment builds only: AO_BUILD_TAG): 1 no DPP shifts (every tap column reads the
// unshifted fragment), 3 no epilogue (no LDS writes), 4 the weights of a layer's block 0 only, 5 no board load, 6 no residual (the ResBlock
// input is not kept: prices the 60 registers it occupies and the scratch traffic of what spills; round 6). (1 - 5 measured on the first,
// slab-streamed form of the kernel: profiles/r5k_boardh_knockouts.txt.)
#ifndef AO_BKO
#define AO_BKO 0
#endif
#if AO_BKO != 0 && !defined(AO_WRONG_RESULTS_OK)
#error "AO_BKO builds compute wrong results on purpose: timing only, build them with -DAO_WRONG_RESULTS_OK"
#endif

namespace ao {

struct BoardHArgs {
    const uint4* act;    // IN 1: conv1's output in the group layout [group of 16][cell][block 4][half 2][oct 4][board 16] x 16 B (k_layer16h KIND 1)
    const uint8_t* planes;   // IN 2: the engine's bit planes, [board][kPlaneRow(BW)] bytes, bit q = plane q (tree_device.hpp encode_planes)
    float* hbuf;         // [board][3][A]: the heads' 1x1 convs + BatchNorm + ReLU of the trunk's output (model.py:37-39,56-58), what k_head_fc reads
    const float *w3, *sc3, *sh3;   // head conv weights [3][128] (policy 0, 1; value 2) and their folded BatchNorm
    int nlayers;         // 1 + 2 * n_block, conv1 included (layers[0]: its BatchNorm scale / shift for IN 2)
    int nboards;
    const unsigned* live;   // live rows of this simulation's batch (net_common.hpp) or null
    unsigned row_cap;
    TrunkHLayer layers[kMaxTrunkLayers];
};

constexpr int kDppRowShr1 = 0x111, kDppRowShl1 = 0x101;
template <int CTRL>
__device__ __forceinline__ half8 dpp_shift_h8(const half8 v) {
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = static_cast<unsigned>(__builtin_amdgcn_mov_dpp(static_cast<int>(u[k]), CTRL, 0xf, 0xf, true));   // bound_ctrl: lanes without a source read 0
    return __builtin_bit_cast(half8, r);
}

// IN: 2 = conv1 runs here, on the engine's bit planes (ao_search); 1 = conv1 ran as k_layer16h on the fp32 plane batch (ao_net_forward
// takes any float planes) and its output is gathered from the group layout
// W16: the conv weights are fp16 numbers (zero low halves): two products per multiply-add, no low weight fragments (36 registers
// less) -- see TrunkHLayerFn
template <int BW, int IN, bool W16>
__device__ __forceinline__ void boardh_body(const BoardHArgs& a) {
    static_assert(BW >= 10 && BW <= 15, "rows are padded to 16 cells and need at least one zero pad");
    __shared__ uint8_t s_pl[256];                  // IN 2: the board's plane bytes
    __shared__ __attribute__((aligned(16))) float s_w3[384];
    constexpr int A = BW * BW;
    constexpr int NCI = 4, NT = 8;                 // 128 channels: four 32-channel blocks, eight 16-channel cout tiles
    constexpr int NFR = BW * NCI * 2;              // 1 KB fragments of the board
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];   // [row][block][half][64] : NFR KB
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int kq = lane >> 4, n = lane & 15;       // B operand: k-octet, cell of the row; D operand: cout quad, cell
    const int lane16 = lane * 16;
    unsigned nlive = static_cast<unsigned>(a.nboards);
    if (a.live) {
        const unsigned lv = *a.live < a.row_cap ? *a.live : a.row_cap;
        nlive = lv < nlive ? lv : nlive;
    }
    // the lane's 8-byte slot inside output fragment (row, tile >> 1, half): cout quad q = kq -> k-octet (tile & 1) * 2 + (kq >> 1)
    const int out_off = (((tile & 1) * 2 + (kq >> 1)) * 16 + n) * 16 + (kq & 1) * 8;
    float peak = 0.f;
    // the head convs' weights [3][128] in LDS: read in the last epilogue only (held in registers from here they cost 12 live VGPRs
    // through every layer of a kernel that has none to spare)
    if (threadIdx.x < 384) s_w3[threadIdx.x] = a.w3[threadIdx.x];
    // Which board a workgroup carries: the 16 boards of a GROUP share every 128-byte line of the group layout (a board's share of a
    // line is 16 bytes), so they go to 16 workgroups of ONE XCD at the same time -- workgroups are dealt round robin over the 8 XCDs:
    // virtual index v = round * gridDim + blockIdx -> XCD x = v % 8, position j = v / 8: group (j / 16) * 8 + x, slot j % 16. A line
    // then crosses the fabric once and is hit 15 times in that XCD's L2 (dealt board by board it was fetched by all 8 XCDs).
    const unsigned nbv = (static_cast<unsigned>(a.nboards) + 127u) & ~127u;   // (whole rounds of 8 groups; boards beyond the batch are skipped)
    for (unsigned v = blockIdx.x; v < nbv; v += gridDim.x) {
        const unsigned x8 = v & 7u, j = v >> 3;
        const unsigned grp = (j >> 4) * 8u + x8, bslot = j & 15u;
        const unsigned board = grp * 16u + bslot;
        if (board >= nlive) continue;   // (uniform for the workgroup)
        const char* gact = reinterpret_cast<const char*>(a.act) + static_cast<size_t>(grp) * A * NCI * 2048u;
        // the ResBlock input of THIS wave's cout tile stays in registers across the block's two convs (60 registers; parked in a
        // global scratch it cost a round trip per epilogue batch and 2.4 GB of L2 traffic per launch)
        f32x4 xres[BW];
        __syncthreads();   // (the previous board's last layer has read the buffer)
        if (IN == 2) {
            // ---- conv1 (model.py:86-89) on the board's bit planes: a lane's B operand for a tap is the plane byte of cell (row + tap row - 1,
            // cell + tap column - 1) expanded to eight halves 0 / 1 (exact: no low half, two products).
            if (threadIdx.x < 64) reinterpret_cast<uint32_t*>(s_pl)[threadIdx.x] =
                reinterpret_cast<const uint32_t*>(a.planes + static_cast<size_t>(board) * kPlaneRow(BW))[threadIdx.x < kPlaneRow(BW) / 4 ? threadIdx.x : 0];
            // Same operands in the same order as conv1 of the per-layer path (k_layer16h KIND 2 / KIND 1 on 0 / 1 planes: per output cell
            // tap rows ascending, tap columns ascending, wh then wl into ONE accumulator), so a position's evaluation has the same bits
            // whether its planes came as bits (ao_search) or as floats (ao_net_forward, the step-wise protocol): 18 MFMAs per output row,
            // three rows interleaved.
            half8 ah[9], al[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                ah[t] = __builtin_bit_cast(half8, a.layers[0].wh[(t * NT + tile) * 64 + lane]);
                if (!W16) al[t] = __builtin_bit_cast(half8, a.layers[0].wl[(t * NT + tile) * 64 + lane]);
            }
            const float4 sc1 = a.layers[0].sc[tile * 4 + kq], sh1 = a.layers[0].sh[tile * 4 + kq];
            __syncthreads();
#pragma unroll
            for (int y0 = 0; y0 < BW; y0 += 3) {
                f32x4 c[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    half8 xb[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int r = y0 + j + t / 3 - 1, x = n + t % 3 - 1;
                        unsigned bq = 0;
                        if (kq == 0 && y0 + j < BW && r >= 0 && r < BW && x >= 0 && x < BW) bq = s_pl[r * BW + x];   // channels 0..7 = the k-octet 0 lanes
                        uint4 v;
                        v.x = ((bq & 1u) ? 0x3C00u : 0u) | ((bq & 2u) ? 0x3C000000u : 0u);
                        v.y = ((bq & 4u) ? 0x3C00u : 0u) | ((bq & 8u) ? 0x3C000000u : 0u);
                        v.z = ((bq & 16u) ? 0x3C00u : 0u) | ((bq & 32u) ? 0x3C000000u : 0u);
                        v.w = ((bq & 64u) ? 0x3C00u : 0u) | ((bq & 128u) ? 0x3C000000u : 0u);
                        xb[j] = __builtin_bit_cast(half8, v);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], xb[j], c[j], 0, 0, 0);
                    if (!W16) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], xb[j], c[j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int y = y0 + j;
                    if (y >= BW) continue;
                    const float f[4] = {fmaf(c[j][0], sc1.x, sh1.x), fmaf(c[j][1], sc1.y, sh1.y), fmaf(c[j][2], sc1.z, sh1.z), fmaf(c[j][3], sc1.w, sh1.w)};
                    half4 hh, hl;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        peak = fmaxf(peak, n < BW ? f[k] : 0.f);
                        const float v = n < BW ? fminf(fmaxf(f[k], 0.f), 65504.f) : 0.f;
                        hh[k] = static_cast<_Float16>(v);
                        hl[k] = static_cast<_Float16>(v - static_cast<float>(hh[k]));
                        if (AO_BKO != 6) xres[y][k] = static_cast<float>(hh[k]) + static_cast<float>(hl[k]);   // (what the gather of IN 1 reconstructs)
                    }
                    char* frag = reinterpret_cast<char*>(s_x + ((y * NCI + (tile >> 1)) * 2) * 64);
                    *reinterpret_cast<half4*>(frag + out_off) = hh;
                    *reinterpret_cast<half4*>(frag + 1024 + out_off) = hl;
                }
            }
        } else {
            // ---- the board's activations (conv1's output) out of the group layout into LDS
#pragma unroll 1
            for (int f = tile; f < (AO_BKO == 5 ? 0 : BW * NCI); f += NT) {            // (row, block): both halves
                const int row = f / NCI, kb = f % NCI;
                uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
                if (n < BW) {
                    const char* p = gact + (static_cast<size_t>(row * BW + n) * NCI + kb) * 2048u + (kq * 16 + bslot) * 16;
                    vh = *reinterpret_cast<const uint4*>(p);
                    vl = *reinterpret_cast<const uint4*>(p + 1024);
                }
                s_x[(f * 2 + 0) * 64 + lane] = vh;
                s_x[(f * 2 + 1) * 64 + lane] = vl;
            }
#pragma unroll
            for (int y = 0; y < BW; ++y) {   // x = xh + xl of conv1's output, this wave's cout tile in D-operand order
                xres[y] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < BW && AO_BKO != 5 && AO_BKO != 6) {
                    const char* p = gact + (static_cast<size_t>(y * BW + n) * NCI + (tile >> 1)) * 2048u + (((tile & 1) * 2 + (kq >> 1)) * 16 + bslot) * 16 + (kq & 1) * 8;
                    const half4 hh = *reinterpret_cast<const half4*>(p), hl = *reinterpret_cast<const half4*>(p + 1024);
#pragma unroll
                    for (int c = 0; c < 4; ++c) xres[y][c] = static_cast<float>(hh[c]) + static_cast<float>(hl[c]);
                }
            }
        }
        __syncthreads();
        // slab = (32-channel block kb, tap row ky): 3 taps x {high, low} weight fragments, streamed from L2 one slab ahead into the
        // other register set while this slab's sweep multiplies -- across the layer boundary too: the last slab of a layer requests
        // slab 0 of the NEXT layer, which then lands during the epilogue
        half8 w9[2][9];
        auto load_w9 = [&](const TrunkHLayer& LL, int kb) {
            const __amdgpu_buffer_rsrc_t r_h = make_rsrc(LL.wh, 9u * NCI * NT * 1024u);
            const __amdgpu_buffer_rsrc_t r_l = make_rsrc(LL.wl, 9u * NCI * NT * 1024u);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ub = ((t * NCI + kb) * NT + tile) * 1024;
                w9[0][t] = buf_ld_h8(r_h, lane16, ub);
                if (!W16) w9[1][t] = buf_ld_h8(r_l, lane16, ub);
            }
        };
        load_w9(a.layers[1], 0);
#pragma unroll 1
        for (int l = 1; l < a.nlayers; ++l) {
            const TrunkHLayer& L = a.layers[l];
            const TrunkHLayer& Lnext = a.layers[l + 1 < a.nlayers ? l + 1 : l];
            const bool second = (l & 1) == 0;          // second conv of a ResBlock: + x
            const bool last = l + 1 == a.nlayers;
            const float4 sc = L.sc[tile * 4 + kq], sh = L.sh[tile * 4 + kq];
            f32x4 acc[BW];
#pragma unroll
            for (int y = 0; y < BW; ++y) acc[y] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Block-outermost, all NINE taps of the block in registers (72, requested when the previous block's -- or, across the layer
            // boundary, the previous layer's -- last row was done): an input row is read from LDS and shifted ONCE per block and feeds
            // 27 MFMAs into the accumulators of output rows r + 1, r, r - 1 (consecutive MFMAs on different registers).
#pragma unroll 1
            for (int kb = 0; kb < NCI; ++kb) {
#pragma unroll
                for (int r = 0; r < BW; ++r) {
                    half8 xh[3], xl[3];
                    xh[1] = __builtin_bit_cast(half8, s_x[((r * NCI + kb) * 2 + 0) * 64 + lane]);
                    xl[1] = __builtin_bit_cast(half8, s_x[((r * NCI + kb) * 2 + 1) * 64 + lane]);
                    // tap column kx reads input cell (output cell + kx - 1): kx = 0 from the lane below, kx = 2 from the lane above
                    xh[0] = AO_BKO == 1 ? xh[1] : dpp_shift_h8<kDppRowShr1>(xh[1]);
                    xl[0] = AO_BKO == 1 ? xl[1] : dpp_shift_h8<kDppRowShr1>(xl[1]);
                    xh[2] = AO_BKO == 1 ? xh[1] : dpp_shift_h8<kDppRowShl1>(xh[1]);
                    xl[2] = AO_BKO == 1 ? xl[1] : dpp_shift_h8<kDppRowShl1>(xl[1]);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {          // products xh*wh, xh*wl, xl*wh
                        if (W16 && pr == 1) continue;         // (wl == 0)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) {
                                const int y = r + 1 - ky;     // input row r = output row y + ky - 1; rows off the board are skipped (uniform)
                                if (y < 0 || y >= BW) continue;
                                acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w9[pr == 1 ? 1 : 0][ky * 3 + kx], pr == 2 ? xl[kx] : xh[kx], acc[y], 0, 0, 0);
                            }
                        }
                    }
                }
                if (AO_BKO != 4) {
                    if (kb + 1 < NCI) load_w9(L, kb + 1);
                    else load_w9(Lnext, 0);                   // (after the last layer: its block 0 again, never used)
                }
            }
            // ---- layer boundary: every wave has read the input; the output replaces it
            __syncthreads();
            if (AO_BKO == 3) {
#pragma unroll
                for (int y = 0; y < BW; ++y) asm volatile("" ::"v"(acc[y]));
                __syncthreads();
                continue;
            }
#pragma unroll
            for (int y = 0; y < BW; ++y) {
                float f[4] = {fmaf(acc[y][0], sc.x, sh.x), fmaf(acc[y][1], sc.y, sh.y), fmaf(acc[y][2], sc.z, sh.z), fmaf(acc[y][3], sc.w, sh.w)};
                if (second && AO_BKO != 6) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) f[c] += xres[y][c];
                }
                half4 hh, hl;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    peak = fmaxf(peak, n < BW ? f[c] : 0.f);
                    v[c] = n < BW ? fminf(fmaxf(f[c], 0.f), 65504.f) : 0.f;   // ReLU, fp16-range clamp (reported), zero pad cell
                    hh[c] = static_cast<_Float16>(v[c]);
                    hl[c] = static_cast<_Float16>(v[c] - static_cast<float>(hh[c]));
                }
                if (last) {
                    // The trunk's output never leaves the chip (round 6; it was 118 MB of fp32 NHWC per launch of 1024 boards for a
                    // separate k_head_conv launch): the heads' 1x1 convs (128 -> 2 + 1 channels, model.py:37,56) are taken here, from
                    // the registers -- this lane's four couts, then the four k-quads of the wave (lanes n, n + 16, n + 32, n + 48),
                    // then, through the LDS the activations no longer need, the eight waves in a fixed order.
                    float hs[3];
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const float4 w4 = *reinterpret_cast<const float4*>(s_w3 + h * 128 + tile * 16 + kq * 4);
                        float t = fmaf(v[0], w4.x, 0.f);
                        t = fmaf(v[1], w4.y, t);
                        t = fmaf(v[2], w4.z, t);
                        t = fmaf(v[3], w4.w, t);
                        t += __shfl_xor(t, 16);
                        t += __shfl_xor(t, 32);
                        hs[h] = t;
                    }
                    if (kq == 0 && n < BW) {
                        float* part = reinterpret_cast<float*>(s_x);
#pragma unroll
                        for (int h = 0; h < 3; ++h) part[(tile * 3 + h) * A + y * BW + n] = hs[h];
                    }
                    continue;
                }
                char* frag = reinterpret_cast<char*>(s_x + ((y * NCI + (tile >> 1)) * 2) * 64);
                *reinterpret_cast<half4*>(frag + out_off) = hh;
                *reinterpret_cast<half4*>(frag + 1024 + out_off) = hl;
                if (second && AO_BKO != 6) xres[y] = f32x4{v[0], v[1], v[2], v[3]};   // the next block's input
            }
            __syncthreads();
            if (last) {
                // the eight waves' partial head sums -> BatchNorm + ReLU -> hbuf (k_head_fc: the FC layers want the whole chip)
                const float* part = reinterpret_cast<const float*>(s_x);
                for (int i = threadIdx.x; i < 3 * A; i += 512) {
                    const int h = i / A, cell = i - h * A;
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < NT; ++w) t += part[(w * 3 + h) * A + cell];
                    a.hbuf[static_cast<size_t>(board) * 3 * A + i] = fmaxf(fmaf(t, a.sc3[h], a.sh3[h]), 0.f);
                }
            }
        }
    }
    if (peak > 65504.f) atomicOr(a.layers[1].ovf, 1);
}

template <int BW, int IN>
__global__ __launch_bounds__(512, 1) void k_boardh(BoardHArgs a) {
    boardh_body<BW, IN, false>(a);
}
template <int BW, int IN>
__global__ __launch_bounds__(512, 1) void k_boardh_w16(BoardHArgs a) {   // launched from net_w16.hip
    boardh_body<BW, IN, true>(a);
}

}  // namespace ao
