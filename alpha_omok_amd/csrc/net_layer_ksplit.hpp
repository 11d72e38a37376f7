// net_layer_ksplit.hpp -- k_layer16hk: one 3x3 trunk conv per launch for MEDIUM batches (a few dozen 16-board groups),
// where neither split-fp16 form of net_trunk_h16.hpp fills the chip: the resident kernel gives every group ONE workgroup
// (64 groups = a quarter of the CUs), and k_layer16h cuts a group into row chunks -- 2-3 output rows per workgroup, each
// paying a halo row above and below and a full row step (barrier, staging, epilogue) for a third to two thirds of a row
// step's MFMAs: 84 us per conv at 1024 boards against 31 us of MFMA issue (profiles/r3n_*).
//
// Here a group is split by OUTPUT CHANNELS instead: workgroup (group, j) computes cout tiles 2j, 2j+1 for ALL cells of
// the group (KS = 4 workgroups per group, 1024 boards -> 256 workgroups of identical work, no halo rows, nine full row
// steps each). Its eight waves split the contraction: wave (t, kp) owns cout tile t of the pair and the kp-th 32-channel
// INPUT block. Consequences:
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
// fp32-equivalent like the other split-fp16 kernels (three fp16 x fp16 products, fp32 accumulate); the summation order
// over the four input blocks differs from theirs (there: one accumulator over all blocks), i.e. by fp32 rounding.
// ao_net_set_mode(6) (one arithmetic for every batch size) therefore never plans this kernel.
#pragma once

namespace ao {

template <int BW>
__global__ __launch_bounds__(512, 1) void k_layer16hk(LayerHArgs a) {
    constexpr int KS = 4;                    // workgroups per group = waves per cout tile
    constexpr int NC32 = 4, NCI = 4, NT = 8; // 128 channels: four 32-channel blocks, eight 16-channel cout tiles
    constexpr int A = BW * BW;
    constexpr int NFR = BW * NCI * 2;        // 1 KB fragments of a staged input row = exchange blocks (8 waves x BW cells)
    static_assert(NFR == 8 * BW, "row fragments and exchange blocks share one index space");
    extern __shared__ __attribute__((aligned(16))) uint4 s_x[];   // [2][NFR][64]

    // XCD-aware placement: the four workgroups of a group share its input rows -- same XCD, same L2 (blocks are dealt
    // round robin over the 8 XCDs)
    const int groups = a.nch;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, rr = bid >> 3;
    const int j = rr % KS, grp = (rr / KS) * 8 + xcd;
    if (grp >= groups) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
    const int t_loc = w & 1, kp = w >> 1;    // cout tile of the pair, input block (= partner index in the exchange)
    const int tile = j * 2 + t_loc;          // cout tile of the layer
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

    // this wave's weights, resident for the whole layer: W[half][tap]
    half8 W[2][9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ub = ((t * NCI + kp) * NT + tile) * 1024;
        W[0][t] = buf_ld_h8(rs_wh, lane16, ub);
        W[1][t] = buf_ld_h8(rs_wl, lane16, ub);
    }

    // Blocks this wave READS in the exchange and STAGES afterwards: (writer wave w2, cell i) with the same cout tile
    // (w2 & 1 == t_loc) and i % 4 == kp. Block index = fragment index = w2 * BW + i.
    constexpr int NOWN = (BW + KS - 1) / KS;         // cells a wave can own: kp, kp + 4, kp + 8
    auto stage_row = [&](int y, uint4* xb) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                const int i = kp + KS * o;
                if (i < BW) {
                    const int f = (t_loc + 2 * q) * BW + i;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void*)(xb + f * 64), 16, lane16,
                                                             (y * NFR + f) * 1024, 0, AO_AUX_STAGE);
                }
            }
        }
    };

    float peak = 0.f;
    f32x4 acc[3][BW];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < BW; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // park this wave's partial tiles of one output row (all cells it does not own) in row buffer xb
    auto park = [&](const f32x4 (&row)[BW], uint4* xb) {
#pragma unroll
        for (int i = 0; i < BW; ++i)
            if ((i & (KS - 1)) != kp) xb[(w * BW + i) * 64 + lane] = __builtin_bit_cast(uint4, row[i]);
    };
    // owner side of the exchange for output row yo: own[o] = this wave's partial of cell kp + 4 o (copied out of the
    // accumulator window before it moved on); rh / rl = the residual, requested by the caller ahead of the barrier
    auto finish = [&](int yo, const f32x4 (&own)[NOWN], const uint4* xb, const half4 (&rh)[NOWN], const half4 (&rl)[NOWN]) {
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int i = kp + KS * o;
            if (i >= BW) continue;
            f32x4 p[KS];
#pragma unroll
            for (int q = 0; q < KS; ++q)
                p[q] = (q == kp) ? own[o] : __builtin_bit_cast(f32x4, xb[((t_loc + 2 * q) * BW + i) * 64 + lane]);
            const f32x4 c = ((p[0] + p[1]) + p[2]) + p[3];     // fixed order over the input blocks
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
    // the MFMAs of input cell (s, xi): 3 tap rows x 3 tap columns x 3 products of this wave's input block
    auto ldx = [&](const uint4* xs, int xi, int half) -> half8 {
        return __builtin_bit_cast(half8, xs[((xi * NCI + kp) * 2 + half) * 64 + lane]);
    };
    auto cell = [&](int s, int xi, const half8 xh, const half8 xl) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yo = s + 1 - dy;
                if (yo < 0 || yo >= BW) continue;   // (uniform)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int i = xi - dx + 1;
                    if (i < 0 || i >= BW) continue;
                    acc[2 - dy][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[pr == 1 ? 1 : 0][dy * 3 + dx], pr == 2 ? xl : xh,
                                                                           acc[2 - dy][i], 0, 0, 0);
                }
            }
        }
    };
    // this wave's own cell kp + 4 o of an accumulator row, picked with value selects (a register array must not be
    // indexed by the wave-uniform but dynamic kp)
    auto pick = [&](const f32x4 (&row)[BW], int o) -> f32x4 {
        f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < BW; ++i)
            if (i == kp + KS * o) r = row[i];
        return r;
    };

    // prologue: input row 0
    stage_row(0, s_x);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    f32x4 own[NOWN];                         // this wave's partials of the cells it owns, of the output row in exchange
#pragma unroll
    for (int o = 0; o < NOWN; ++o) own[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < BW; ++s) {
        const uint4* xs = s_x + static_cast<size_t>(s & 1) * NFR * 64;          // input row s
        uint4* xn = s_x + static_cast<size_t>((s + 1) & 1) * NFR * 64;          // partials of output row s-2, then input row s+1
        half4 rh[NOWN], rl[NOWN];
        if (s >= 2) load_res(s - 2, rh, rl);                                     // in flight under the first cell's MFMAs
        else if (s + 1 < BW) stage_row(s + 1, xn);                               // nothing to exchange yet: stage at once
        half8 xh = ldx(xs, 0, 0), xl = ldx(xs, 0, 1);
#pragma unroll
        for (int xi = 0; xi < BW; ++xi) {
            half8 nh = xh, nl = xl;
            if (xi + 1 < BW) {
                nh = ldx(xs, xi + 1, 0);
                nl = ldx(xs, xi + 1, 1);
            }
            cell(s, xi, xh, xl);
            xh = nh;
            xl = nl;
            __builtin_amdgcn_sched_barrier(0);
            if (xi == 0 && s >= 2) {
                // "partials visible": every wave parked its tiles (end of the previous step) before this barrier
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                finish(s - 2, own, xn, rh, rl);
                if (s + 1 < BW) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // my reads of the blocks are done: refill them
                    stage_row(s + 1, xn);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // end of the row step: my share of row s+1 has landed, everybody is done reading row s
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s >= 1) {
            // output row s-1 is complete in acc[0]: park the partials in the buffer of row s (free now), keep my own cells
            park(acc[0], const_cast<uint4*>(xs));
#pragma unroll
            for (int o = 0; o < NOWN; ++o) own[o] = pick(acc[0], o);
        }
#pragma unroll
        for (int i = 0; i < BW; ++i) {
            acc[0][i] = acc[1][i];
            acc[1][i] = acc[2][i];
            acc[2][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the last two output rows: BW-2 is parked in the buffer of row BW-1, BW-1 sits in acc[0] and goes to the other one
    {
        uint4* x7 = s_x + static_cast<size_t>((BW - 1) & 1) * NFR * 64;
        uint4* x8 = s_x + static_cast<size_t>(BW & 1) * NFR * 64;
        f32x4 own8[NOWN];
#pragma unroll
        for (int o = 0; o < NOWN; ++o) own8[o] = pick(acc[0], o);
        park(acc[0], x8);
        half4 rh[NOWN], rl[NOWN], rh8[NOWN], rl8[NOWN];
        if (BW >= 2) load_res(BW - 2, rh, rl);
        load_res(BW - 1, rh8, rl8);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (BW >= 2) finish(BW - 2, own, x7, rh, rl);
        finish(BW - 1, own8, x8, rh8, rl8);
    }
    if (peak > 65504.f) atomicOr(L.ovf, 1);
}

}  // namespace ao
