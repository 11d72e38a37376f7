// net_layer_ksplit.hpp -- k_layer16hk: one 3x3 trunk conv per launch for MEDIUM batches (a few dozen 16-board groups),
// where neither split-fp16 form of net_trunk_h16.hpp fills the chip: the resident kernel gives every group ONE workgroup
// (64 groups = a quarter of the CUs), and k_layer16h cuts a group into row chunks -- 2-3 output rows per workgroup, each
// paying a halo row above and below and a full row step (barrier, staging, epilogue) for a third to two thirds of a row
// step's MFMAs: 84 us per conv at 1024 boards against 31 us of MFMA issue (profiles/r3n_*).
//
// Here a group is split by OUTPUT CHANNELS instead: workgroup (group, j) computes cout tiles 2j, 2j+1 for ALL cells of
// the group (KS = 4 workgroups per group, 1024 boards -> 256 workgroups of identical work, no halo rows, nine full row
// steps each; KS = 2 for 65 .. 128 groups: cout QUADS, two workgroups per group). Its eight waves split the contraction:
// wave (t, kp) owns cout tile t of the pair and the kp-th 32-channel INPUT block (KS = 2: two blocks). Consequences (KS = 4):
//   * the wave's weights -- 9 taps x {high, low} of ONE (tile, block) = 18 fragments = 72 VGPRs -- are loaded ONCE per
//     layer and stay in registers (the other kernels re-stream a tile's 72 KB per board row);
//   * an input fragment pair read from LDS feeds 27 MFMAs of its wave (all nine taps x three products);
//   * every wave holds PARTIAL sums over its block, so an output row is finished by a 4-way exchange through LDS. The
//     exchange needs no LDS beyond the two row buffers: when row step s ends, the buffer of input row s is free, the
//     waves park the partial tiles of output row s-1 in it (block (writer wave, cell), 1 KB each: 72 blocks = the
//     buffer's 72 fragments), and during step s+1 the OWNER of a cell (wave (t, cell % 4)) adds its three partners'
//     tiles to its own in the fixed order kp = 0, 1, 2, 3 (deterministic, whoever owns the cell), runs the epilogue
//     (BatchNorm, residual, ReLU, fp16 split, store) and then stages the next input row into exactly the blocks it has
//     just read -- reader and stager of a block are the same wave, so the refill needs no barrier. Two barriers per row
//     step: end of step (existing) and "partials visible".
//   * a row step of this kernel is SHORT (225 MFMAs per wave against 900 in the resident kernel), so what the two waves of
//     a SIMD do besides MFMAs must not coincide: the waves 0-3 run their exchange + epilogue right after the "partials
//     visible" barrier, their SIMD partners 4-7 half a step later -- one multiplies while the other exchanges; and the
//     three-row accumulator window rotates by NAME (the step loop is unrolled by three) instead of being shifted through
//     108 register moves per step. (First version, without both: 13 k cycles per step for 7.8 k of MFMA issue --
//     profiles/r4c_ksplit_medium_batches.txt.)
// KS = 2: a wave owns two input blocks; their weights do not fit the registers beside the window, so they stream from L2
// one (block, tap row) slab ahead exactly as in the resident kernel (six slabs per row step).
// fp32-equivalent like the other split-fp16 kernels (three fp16 x fp16 products, fp32 accumulate); the summation order
// over the four input blocks differs from theirs (there: one accumulator over all blocks), i.e. by fp32 rounding.
// ao_net_set_mode(6) (one arithmetic for every batch size) therefore never plans this kernel.
#pragma once

namespace ao {

// W16: the conv weights are fp16 numbers (zero low halves): two products per multiply-add, no low weight fragments -- see
// TrunkHLayerFn in net_trunk_h16.hpp. The resident weights of a wave are then 36 registers instead of 72.
template <int BW, int KS, bool W16>
__device__ __forceinline__ void layer16hk_body(const LayerHArgs& a) {
    static_assert(KS == 2 || KS == 4, "workgroups per group");
    constexpr int TW = 8 / KS;               // cout tiles of a workgroup
    constexpr int CB = 4 / KS;               // 32-channel input blocks of a wave
    constexpr int NC32 = 4, NCI = 4, NT = 8; // 128 channels: four 32-channel blocks, eight 16-channel cout tiles
    constexpr int A = BW * BW;
    constexpr int NFR = BW * NCI * 2;        // 1 KB fragments of a staged input row = exchange blocks (8 waves x BW cells)
    static_assert(NFR == 8 * BW, "row fragments and exchange blocks share one index space");
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];   // [2][NFR][64]

    // XCD-aware placement: the KS workgroups of a group share its input rows -- same XCD, same L2 (blocks are dealt
    // round robin over the 8 XCDs)
    const int groups = live_groups16(a.live, a.row_cap, a.nch);   // (groups without a live row: nothing to do)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, rr = bid >> 3;
    const int j = rr % KS, grp = (rr / KS) * 8 + xcd;
    if (grp >= groups) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int t_loc = w % TW, kp = w / TW;   // cout tile within the workgroup; partner index in the exchange = input block(s)
    // the SIMD partner of wave w - 4 exchanges half a step later. (Resident weights only: with streamed weights the MFMA stream
    // carries counted s_waitcnt vmcnt(N) for its slabs, and an exchange -- staging loads, stores -- under a wave-dependent
    // branch makes the compiler fall back to vmcnt(0): every slab then waits for the row being staged.)
    const bool late = CB == 1 && w >= 4;
    const int tile = j * TW + t_loc;         // cout tile of the layer
    const int kq = lane >> 4, b = lane & 15;
    const int lane16 = lane * 16;
    const int out_voff = (((tile & 1) * 2 + (kq >> 1)) * 16 + b) * 16 + (kq & 1) * 8;
    const TrunkHLayer& L = a.layer;
    const bool RES = a.res != 0;
    const float4 sc = L.sc[tile * 4 + kq], sh = L.sh[tile * 4 + kq];
    const uint4* src = static_cast<const uint4*>(a.src) + static_cast<size_t>(grp) * A * NC32 * 2 * 64;
    uint4* dst = a.dst + static_cast<size_t>(grp) * A * NC32 * 2 * 64;
    const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(L.wh, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(L.wl, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(src, static_cast<unsigned>(A) * NCI * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rs_dst = make_rsrc(dst, static_cast<unsigned>(A) * NC32 * 2u * 1024u);

    // weights. CB == 1: W[half][tap] of the wave's (tile, block), resident for the whole layer. CB == 2: slab (block, tap
    // row) = 3 taps x {high, low}, double-buffered one slab ahead (a row step has 6 slabs: the parity carries over)
    half8 W[2][CB == 1 ? 9 : 1];
    half8 wA[2][CB == 1 ? 1 : 3], wB[2][CB == 1 ? 1 : 3];
    if constexpr (CB == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ub = ((t * NCI + kp) * NT + tile) * 1024;
            W[0][t] = buf_ld_h8(rs_wh, lane16, ub);
            if (!W16) W[1][t] = buf_ld_h8(rs_wl, lane16, ub);
        }
    }
    auto load_slab = [&](int sl, half8 (&Wd)[2][CB == 1 ? 1 : 3]) {
        if constexpr (CB > 1) {
            const int c = kp * CB + (sl / 3) % CB, dy = sl % 3;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ub = (((dy * 3 + dx) * NCI + c) * NT + tile) * 1024;
                Wd[0][dx] = buf_ld_h8(rs_wh, lane16, ub);
                if (!W16) Wd[1][dx] = buf_ld_h8(rs_wl, lane16, ub);
            }
        }
    };

    // Blocks this wave READS in the exchange and STAGES afterwards: (writer wave w2, cell i) with the same cout tile
    // (w2 % TW == t_loc) and i % KS == kp. Block index = fragment index = w2 * BW + i.
    constexpr int NOWN = (BW + KS - 1) / KS;         // cells a wave can own: kp, kp + KS, ...
    auto stage_row = [&](int y, uint4* xb) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                const int i = kp + KS * o;
                if (i < BW) {
                    const int f = (t_loc + TW * q) * BW + i;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xb + f * 64), 16, lane16,
                                                             (y * NFR + f) * 1024, 0, AO_AUX_STAGE);
                }
            }
        }
    };

    float peak = 0.f;
    f32x4 acc[3][BW];                        // three output rows; which is which rotates with the step (see step())
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < BW; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // park this wave's partial tiles of one output row in row buffer xb (the cells it owns too: their blocks are read by
    // nobody else, and the window's registers are free for the next row at once)
    auto park = [&](const f32x4 (&row)[BW], uint4* xb) {
#pragma unroll
        for (int i = 0; i < BW; ++i) xb[(w * BW + i) * 64 + lane] = __builtin_bit_cast(uint4, row[i]);
    };
    // owner side of the exchange for output row yo (rh / rl = the residual of the owned cells)
    auto finish = [&](int yo, const uint4* xb, const half4 (&rh)[NOWN], const half4 (&rl)[NOWN]) {
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int i = kp + KS * o;
            if (i >= BW) continue;
            f32x4 p[KS];
#pragma unroll
            for (int q = 0; q < KS; ++q) p[q] = __builtin_bit_cast(f32x4, xb[((t_loc + TW * q) * BW + i) * 64 + lane]);
            f32x4 c = p[0] + p[1];                             // fixed order over the partners = input blocks
#pragma unroll
            for (int q = 2; q < KS; ++q) c = c + p[q];
            float f[4] = {fmaf(c[0], sc.x, sh.x), fmaf(c[1], sc.y, sh.y), fmaf(c[2], sc.z, sh.z), fmaf(c[3], sc.w, sh.w)};
            if (RES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] += static_cast<float>(rh[o][r]) + static_cast<float>(rl[o][r]);
            }
            peak = fmaxf(fmaxf(peak, fmaxf(f[0], f[1])), fmaxf(f[2], f[3]));
            half4 hh, hl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fminf(fmaxf(f[r], 0.f), 65504.f);   // ReLU; clamp + report beyond the fp16 range (see trunk_h_layer)
                hh[r] = static_cast<_Float16>(v);
                hl[r] = static_cast<_Float16>(v - static_cast<float>(hh[r]));
            }
            const int ob = (((yo * BW + i) * NC32 + (tile >> 1)) * 2) * 1024;
            buf_st_h4(hh, rs_dst, out_voff, ob);
            buf_st_h4(hl, rs_dst, out_voff, ob + 1024);
        }
    };
    auto load_res = [&](int yo, half4 (&rh)[NOWN], half4 (&rl)[NOWN]) {
        if (!RES) return;
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int i = kp + KS * o < BW ? kp + KS * o : BW - 1;
            const int ob = (((yo * BW + i) * NC32 + (tile >> 1)) * 2) * 1024;
            rh[o] = buf_ld_h4(rs_dst, out_voff, ob);
            rl[o] = buf_ld_h4(rs_dst, out_voff, ob + 1024);
        }
    };
    auto ldx = [&](const uint4* xs, int xi, int c, int half) -> half8 {
        return __builtin_bit_cast(half8, xs[((xi * NCI + c) * 2 + half) * 64 + lane]);
    };
    // prologue: input row 0, the first weight slab
    stage_row(0, s_x);
    load_slab(0, wA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (CB == 1) {
        // The resident weights HAVE arrived (the wait above), but the compiler's wait-count pass does not read inline
        // assembly: without a use it can see, it guards their first use inside the step loop with s_waitcnt vmcnt(N) -- and
        // in the steady state that N covers the staging loads of the row in flight, so every row step stalled until its
        // successor had landed (the first build of this kernel: 2 x its MFMA time). A no-op "use" of every fragment here
        // makes the pass place its wait now.
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            asm volatile("" ::"v"(W[0][t]));
            if (!W16) asm volatile("" ::"v"(W[1][t]));
        }
    }

    // One row step. PH = s % 3 names the accumulator rows: output row s-1 (tap row dy = 2) is acc[PH], row s is
    // acc[(PH+1) % 3], row s+1 is acc[(PH+2) % 3]; after the step acc[PH] is complete, is parked, and becomes row s+2.
    auto step = [&](const int s, auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        const uint4* xs = s_x + static_cast<size_t>(s & 1) * NFR * 64;          // input row s
        uint4* xn = s_x + static_cast<size_t>((s + 1) & 1) * NFR * 64;          // partials of output row s-2, then input row s+1
        if (s < 2 && s + 1 < BW) stage_row(s + 1, xn);                           // nothing to exchange yet: stage at once
        auto exchange = [&]() {
            // (the residual's round trip is covered by the SIMD partner, which multiplies while this wave exchanges)
            half4 rh[NOWN], rl[NOWN];
            load_res(s - 2, rh, rl);
            finish(s - 2, xn, rh, rl);
            if (s + 1 < BW) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // my reads of the blocks are done: refill them
                stage_row(s + 1, xn);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // after the first unit: "partials visible" (every wave parked its tiles at the end of the previous step); the waves
        // 0-3 exchange now, their SIMD partners after the unit `late_at`
        auto sync_point = [&](int unit, int late_at) {
            if (s < 2) return;
            if (unit == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (!late) exchange();
            }
            if (unit == late_at && late) exchange();
        };
        if constexpr (CB == 1) {
            half8 xh = ldx(xs, 0, kp, 0), xl = ldx(xs, 0, kp, 1);
#pragma unroll
            for (int xi = 0; xi < BW; ++xi) {
                half8 nh = xh, nl = xl;
                if (xi + 1 < BW) {
                    nh = ldx(xs, xi + 1, kp, 0);
                    nl = ldx(xs, xi + 1, kp, 1);
                }
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    if (W16 && pr == 1) continue;   // (wl == 0)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int yo = s + 1 - dy;
                        if (yo < 0 || yo >= BW) continue;   // (uniform)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int i = xi - dx + 1;
                            if (i < 0 || i >= BW) continue;
                            acc[(PH + 2 - dy) % 3][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                W[pr == 1 ? 1 : 0][dy * 3 + dx], pr == 2 ? xl : xh, acc[(PH + 2 - dy) % 3][i], 0, 0, 0);
                        }
                    }
                }
                xh = nh;
                xl = nl;
                __builtin_amdgcn_sched_barrier(0);
                sync_point(xi, BW / 2);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < CB * 3; ++sl) {
                const int cb = sl / 3, dy = sl % 3;
                const int c = kp * CB + cb;
                half8 (&wc)[2][3] = (sl & 1) ? wB : wA;
                half8 (&wn)[2][3] = (sl & 1) ? wA : wB;
                load_slab(sl + 1, wn);
                const int yo = s + 1 - dy;
                if (yo >= 0 && yo < BW) {   // (uniform)
                    half8 xh = ldx(xs, 0, c, 0), xl = ldx(xs, 0, c, 1);
#pragma unroll
                    for (int xi = 0; xi < BW; ++xi) {
                        half8 nh = xh, nl = xl;
                        if (xi + 1 < BW) {
                            nh = ldx(xs, xi + 1, c, 0);
                            nl = ldx(xs, xi + 1, c, 1);
                        }
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr) {
                    if (W16 && pr == 1) continue;   // (wl == 0)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const int i = xi - dx + 1;
                                if (i < 0 || i >= BW) continue;
                                acc[(PH + 2 - dy) % 3][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    wc[pr == 1 ? 1 : 0][dx], pr == 2 ? xl : xh, acc[(PH + 2 - dy) % 3][i], 0, 0, 0);
                            }
                        }
                        xh = nh;
                        xl = nl;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                sync_point(sl, CB * 3 / 2);
            }
        }
        // end of the row step: my share of row s+1 has landed, everybody is done reading row s
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s >= 1) {
            // output row s-1 is complete in acc[PH]: park the partials in the buffer of row s (free now)
            park(acc[PH], const_cast<uint4*>(xs));
        }
#pragma unroll
        for (int i = 0; i < BW; ++i) acc[PH][i] = f32x4{0.f, 0.f, 0.f, 0.f};   // ... and it becomes output row s+2
    };
    for (int s0 = 0; s0 < BW; s0 += 3) {
        step(s0, std::integral_constant<int, 0>{});
        if (s0 + 1 < BW) step(s0 + 1, std::integral_constant<int, 1>{});
        if (s0 + 2 < BW) step(s0 + 2, std::integral_constant<int, 2>{});
    }
    // the last two output rows: BW-2 is parked in the buffer of row BW-1; BW-1 sits in the window and goes to the other one
    {
        constexpr int PL = ((BW - 1) % 3 + 1) % 3;    // after the last step (phase (BW-1) % 3) row BW-1 is acc[phase + 1]
        uint4* x7 = s_x + static_cast<size_t>((BW - 1) & 1) * NFR * 64;
        uint4* x8 = s_x + static_cast<size_t>(BW & 1) * NFR * 64;
        park(acc[PL], x8);
        half4 rh[NOWN], rl[NOWN], rh8[NOWN], rl8[NOWN];
        load_res(BW - 2, rh, rl);
        load_res(BW - 1, rh8, rl8);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        finish(BW - 2, x7, rh, rl);
        finish(BW - 1, x8, rh8, rl8);
    }
    if (peak > 65504.f) atomicOr(L.ovf, 1);
}

template <int BW, int KS>
__global__ __launch_bounds__(512, 1) void k_layer16hk(LayerHArgs a) {
    layer16hk_body<BW, KS, false>(a);
}
template <int BW, int KS>
__global__ __launch_bounds__(512, 1) void k_layer16hk_w16(LayerHArgs a) {   // launched from net_w16.hip
    layer16hk_body<BW, KS, true>(a);
}


// k_row16hk: the same decomposition for SMALL and medium-small batches, where even four workgroups per group leave most of
// the chip idle: workgroup (group, output row y, cout pair j) -- 36 workgroups per 9x9 group, all of a group on one XCD. Its
// eight waves are k_layer16hk's (cout tile of the pair x 32-channel input block). ONE row buffer (72 KB on 9x9, so TWO
// workgroups share a CU and one multiplies while the other waits for its row): for each of the three input rows y-1, y, y+1 it
// stages the row and the wave's six weight fragments of that tap row (3 taps x {high, low}), waits, multiplies; the finished
// partial tiles go through the same buffer in the 4-way exchange of k_layer16hk. No window, no pipeline inside the workgroup:
// what it buys is parallelism (16 groups: 576 workgroups, two per CU, instead of k_layer16hk's 64 or k_layer16h's 144). Per
// accumulator the order of the MFMAs (input rows y-1, y, y+1; inside a row: cell, product, tap column) and the order of the
// exchange (input blocks 0 .. 3) are k_layer16hk's: the results are the same bits.
template <int BW, bool W16>
__device__ __forceinline__ void row16hk_body(const LayerHArgs& a) {
    constexpr int KS = 4, TW = 2;
    constexpr int NC32 = 4, NCI = 4, NT = 8;
    constexpr int A = BW * BW;
    constexpr int NFR = BW * NCI * 2;
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];   // [NFR][64]
    const int groups = live_groups16(a.live, a.row_cap, a.nch);   // (groups without a live row: nothing to do)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, rr = bid >> 3;
    const int j = rr % KS, y = (rr / KS) % BW, grp = (rr / (KS * BW)) * 8 + xcd;
    if (grp >= groups) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int t_loc = w % TW, kp = w / TW;
    const int tile = j * TW + t_loc;
    const int kq = lane >> 4, b = lane & 15;
    const int lane16 = lane * 16;
    const int out_voff = (((tile & 1) * 2 + (kq >> 1)) * 16 + b) * 16 + (kq & 1) * 8;
    const TrunkHLayer& L = a.layer;
    const bool RES = a.res != 0;
    const float4 sc = L.sc[tile * 4 + kq], sh = L.sh[tile * 4 + kq];
    const uint4* src = static_cast<const uint4*>(a.src) + static_cast<size_t>(grp) * A * NC32 * 2 * 64;
    uint4* dst = a.dst + static_cast<size_t>(grp) * A * NC32 * 2 * 64;
    const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(L.wh, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(L.wl, 9u * NCI * NT * 1024u);
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(src, static_cast<unsigned>(A) * NCI * 2u * 1024u);
    const __amdgpu_buffer_rsrc_t rs_dst = make_rsrc(dst, static_cast<unsigned>(A) * NC32 * 2u * 1024u);
    constexpr int NOWN = (BW + KS - 1) / KS;
    auto stage_row = [&](int yy) {   // (the wave's share of the row's 8 * BW fragments: see k_layer16hk)
#pragma unroll
        for (int q = 0; q < KS; ++q) {
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                const int i = kp + KS * o;
                if (i < BW) {
                    const int f = (t_loc + TW * q) * BW + i;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(s_x + f * 64), 16, lane16,
                                                             (yy * NFR + f) * 1024, 0, AO_AUX_STAGE);
                }
            }
        }
    };
    f32x4 acc[BW];
#pragma unroll
    for (int i = 0; i < BW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto ldx = [&](int xi, int half) -> half8 { return __builtin_bit_cast(half8, s_x[((xi * NCI + kp) * 2 + half) * 64 + lane]); };
    // one tap row: stage input row y - 1 + DY, fetch the wave's six weight fragments of that tap row, wait, multiply
    auto phase = [&](auto dy_tag, bool first) {
        constexpr int DY = decltype(dy_tag)::value;
        const int yy = y - 1 + DY;
        if (yy < 0 || yy >= BW) return;   // (uniform)
        if (!first) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everybody is done reading the previous row
        stage_row(yy);
        half8 W[2][3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ub = (((DY * 3 + dx) * NCI + kp) * NT + tile) * 1024;
            W[0][dx] = buf_ld_h8(rs_wh, lane16, ub);
            if (!W16) W[1][dx] = buf_ld_h8(rs_wl, lane16, ub);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {   // (a use the wait-count pass can see: k_layer16hk)
            asm volatile("" ::"v"(W[0][dx]));
            if (!W16) asm volatile("" ::"v"(W[1][dx]));
        }
        half8 xh = ldx(0, 0), xl = ldx(0, 1);
#pragma unroll
        for (int xi = 0; xi < BW; ++xi) {
            half8 nh = xh, nl = xl;
            if (xi + 1 < BW) {
                nh = ldx(xi + 1, 0);
                nl = ldx(xi + 1, 1);
            }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr) {
                    if (W16 && pr == 1) continue;   // (wl == 0)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int i = xi - dx + 1;
                    if (i < 0 || i >= BW) continue;
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[pr == 1 ? 1 : 0][dx], pr == 2 ? xl : xh, acc[i], 0, 0, 0);
                }
            }
            xh = nh;
            xl = nl;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    phase(std::integral_constant<int, 0>{}, true);
    phase(std::integral_constant<int, 1>{}, y == 0);
    phase(std::integral_constant<int, 2>{}, false);
    // the residual of the owned cells: requested before the exchange's barriers
    half4 rh[NOWN], rl[NOWN];
    if (RES) {
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int i = kp + KS * o < BW ? kp + KS * o : BW - 1;
            const int ob = (((y * BW + i) * NC32 + (tile >> 1)) * 2) * 1024;
            rh[o] = buf_ld_h4(rs_dst, out_voff, ob);
            rl[o] = buf_ld_h4(rs_dst, out_voff, ob + 1024);
        }
    }
    // exchange through the row buffer: every wave parks its nine partial tiles, the owner of a cell adds the four partners' in
    // the order of the input blocks (= k_layer16hk's finish)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the last row is read
#pragma unroll
    for (int i = 0; i < BW; ++i) s_x[(w * BW + i) * 64 + lane] = __builtin_bit_cast(uint4, acc[i]);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (RES) {
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            asm volatile("" ::"v"(rh[o]));
            asm volatile("" ::"v"(rl[o]));
        }
    }
    float peak = 0.f;
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int i = kp + KS * o;
        if (i >= BW) continue;
        f32x4 p[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) p[q] = __builtin_bit_cast(f32x4, s_x[((t_loc + TW * q) * BW + i) * 64 + lane]);
        f32x4 c = p[0] + p[1];
#pragma unroll
        for (int q = 2; q < KS; ++q) c = c + p[q];
        float f[4] = {fmaf(c[0], sc.x, sh.x), fmaf(c[1], sc.y, sh.y), fmaf(c[2], sc.z, sh.z), fmaf(c[3], sc.w, sh.w)};
        if (RES) {
#pragma unroll
            for (int r = 0; r < 4; ++r) f[r] += static_cast<float>(rh[o][r]) + static_cast<float>(rl[o][r]);
        }
        peak = fmaxf(fmaxf(peak, fmaxf(f[0], f[1])), fmaxf(f[2], f[3]));
        half4 hh, hl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = fminf(fmaxf(f[r], 0.f), 65504.f);
            hh[r] = static_cast<_Float16>(v);
            hl[r] = static_cast<_Float16>(v - static_cast<float>(hh[r]));
        }
        const int ob = (((y * BW + i) * NC32 + (tile >> 1)) * 2) * 1024;
        buf_st_h4(hh, rs_dst, out_voff, ob);
        buf_st_h4(hl, rs_dst, out_voff, ob + 1024);
    }
    if (peak > 65504.f) atomicOr(L.ovf, 1);
}

template <int BW>
__global__ __launch_bounds__(512, 2) void k_row16hk(LayerHArgs a) {
    row16hk_body<BW, false>(a);
}
template <int BW>
__global__ __launch_bounds__(512, 2) void k_row16hk_w16(LayerHArgs a) {   // launched from net_w16.hip
    row16hk_body<BW, true>(a);
}

}  // namespace ao
