// step_kernels.hip -- the fused per-game search step of the small-batch ("latency") path.
//
// A handful of concurrent games (BASELINE configs[1]: ONE game, 400 simulations per move) makes every kernel of a simulation a
// few microseconds long. k_step_board does what three launches did -- k_heads_board, k_expand_select and conv1's
// k_conv_cells -- in one workgroup per game. Measured (end): for ONE game it is a wash (54.7 vs 54.4 us per
// simulation: the launches already followed each other without idle time and the three phases cost inside one kernel what
// they cost alone); from a few games on it wins, because a game's step is one 16-wave workgroup instead of three tiny grids
// (8 / 24 / 48 games: +1 / +2.7 / +4 %). AO_FUSED_STEP=0 falls back to the three launches.
//
//   all 16 waves   policy + value head of the game's last leaf from the trunk's output   (heads_board_dev; model.py:34-73)
//   wave 0         expansion + backup of that leaf, selection of the next one, its input planes as bits in LDS
//                  (expand_backup_game, select_game: agents.py:134-239, utils.py:139-168 -- the same device code as
//                  k_expand_select, so the search stays bit-identical given the same evaluations)
//   all 16 waves   conv1 + BatchNorm + ReLU of the new leaf (model.py:97-99) from those bits: K = 9 taps x 8 planes (three
//                  32-deep steps of v_mfma_f32_16x16x32_f16; a lane's 8 operand values are the 8 planes of ONE neighbour cell,
//                  one 16-byte LDS read), weights split into fp16 high + low halves (the planes are 0 / 1: exact in fp16, so
//                  two products give the fp32-equivalent result), straight into the trunk's fp32 buffer.
//
// A simulation is then 8 trunk convs (k_conv_cells_h) + this kernel: 9 launches instead of 11.
#include <cstdio>
#include <cstdlib>

#include "tree_device.hpp"
#include "net_device.hpp"

namespace ao {

typedef _Float16 st_half8 __attribute__((ext_vector_type(8)));

#ifdef AO_PROF
__device__ unsigned long long ao_prof_step[8];   // game 0, thread 0: start / heads done / barrier / tree done / barrier / conv1 done
#define AO_ST(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) ao_prof_step[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AO_ST(k) do { } while (0)
#endif

template <int NCH>
__global__ __launch_bounds__(1024) void k_step_board(TreeParams p, StepNet f, const int32_t* __restrict__ game_of_row) {
    extern __shared__ __attribute__((aligned(16))) float s_hb[];   // heads_lds_floats(A, planes)
    __shared__ uint32_t s_mt[624];
    __shared__ uint8_t s_ord[256];
    __shared__ double s_prior[256];
    __shared__ int16_t s_tab[256];
    __shared__ uint8_t s_lin[256];                                  // bit planes of the new leaf, byte per cell
    // ... as fp16 0 / 1, 8 planes per cell, with a border of empty cells (entry 0 of the border doubles as "no tap")
    __shared__ __attribute__((aligned(16))) uint4 s_x[(kMaxBoard + 2) * (kMaxBoard + 2)];
    // One workgroup per ROW of the evaluation batch (= per active game; the others were set idle by the move's k_select):
    // the heads need the row only and start at once, the game number arrives while they run.
    const int row = blockIdx.x;
    const int g = game_of_row[row];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int A = p.A, BW = p.B, P = f.planes;
    AO_ST(0);
    if (tid < 256) s_lin[tid] = 0;
    for (int i = tid; i < (BW + 2) * (BW + 2); i += 1024) s_x[i] = make_uint4(0u, 0u, 0u, 0u);
    float4* act = reinterpret_cast<float4*>(f.act) + static_cast<size_t>(row) * A * (P >> 2);
    heads_board_dev(f.heads, act, f.policy + static_cast<size_t>(row) * A, f.value + row, A, P, s_hb);
    AO_ST(1);
    __syncthreads();   // (policy / value of this row are in memory: written and read by this workgroup only)
    AO_ST(2);
    const int kq = lane >> 4, ci = lane & 15;
    const int nt = P >> 4, nct = (A + 15) >> 4;
    const float4* sc4 = reinterpret_cast<const float4*>(f.sc1);
    const float4* sh4 = reinterpret_cast<const float4*>(f.sh1);
    // conv1's weights of the wave's first channel tile: requested by waves 1 .. 15 while wave 0 walks the tree (first touch:
    // an L2 miss), by wave 0 afterwards (then an L1 hit: wave 8 asked for the same lines)
    st_half8 ah[3], al[3];
    float4 sc, sf;
    auto load_a = [&](int ct) {
        sc = sc4[ct * 4 + kq];
        sf = sh4[ct * 4 + kq];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            ah[ks] = __builtin_bit_cast(st_half8, f.w1h[(static_cast<size_t>(ks) * nt + ct) * 64 + lane]);
            al[ks] = __builtin_bit_cast(st_half8, f.w1l[(static_cast<size_t>(ks) * nt + ct) * 64 + lane]);
        }
    };
    if (w == 0) {
        GameHdr hdr;
        expand_backup_game<NCH>(p, g, s_ord, s_prior, s_tab, &hdr);
        wsync();
        select_game<NCH>(p, g, s_mt, s_lin, &hdr);
    } else {
        load_a(w % nt);
    }
    AO_ST(3);
    __syncthreads();
    AO_ST(4);
    if (w == 0) load_a(0);
    if (tid < A) {   // the leaf's bits -> 8 halves per cell
        const unsigned bq = s_lin[tid];
        uint4 v;
        v.x = ((bq & 1u) ? 0x3C00u : 0u) | ((bq & 2u) ? 0x3C000000u : 0u);
        v.y = ((bq & 4u) ? 0x3C00u : 0u) | ((bq & 8u) ? 0x3C000000u : 0u);
        v.z = ((bq & 16u) ? 0x3C00u : 0u) | ((bq & 32u) ? 0x3C000000u : 0u);
        v.w = ((bq & 64u) ? 0x3C00u : 0u) | ((bq & 128u) ? 0x3C000000u : 0u);
        s_x[(tid / BW + 1) * (BW + 2) + tid % BW + 1] = v;
    }
    __syncthreads();
    int ct_loaded = w % nt;
    // k step ks, quarter kq = tap 4 * ks + kq (taps 9 .. 11 do not exist: their weights are zero, they read cell 0)
    int toff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int tap = 4 * ks + kq;
        toff[ks] = tap < 9 ? (tap / 3 - 1) * (BW + 2) + (tap % 3 - 1) : -(1 << 20);
    }
    for (int tile = w; tile < nt * nct; tile += 16) {
        const int ct = tile % nt, ctile = tile / nt;
        if (ct != ct_loaded) {   // (128 planes: 8 channel tiles, 16 waves -- a wave keeps its tile)
            load_a(ct);
            ct_loaded = ct;
        }
        const int cell = ctile * 16 + ci;
        const bool in = cell < A;
        const int base = in ? (cell / BW + 1) * (BW + 2) + cell % BW + 1 : (BW + 2) + 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int at = base + toff[ks];
            const st_half8 xb = __builtin_bit_cast(st_half8, s_x[(in && at >= 0) ? at : 0]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xb, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], xb, acc2, 0, 0, 0);
        }
        const int cq = ct * 4 + kq;   // D row = cout 4 * kq + reg, col = cell ci
        if (in) {
            float4 v;
            v.x = fmaxf(fmaf(acc[0] + acc2[0], sc.x, sf.x), 0.f);
            v.y = fmaxf(fmaf(acc[1] + acc2[1], sc.y, sf.y), 0.f);
            v.z = fmaxf(fmaf(acc[2] + acc2[2], sc.z, sf.z), 0.f);
            v.w = fmaxf(fmaf(acc[3] + acc2[3], sc.w, sf.w), 0.f);
            act[static_cast<size_t>(cell) * (P >> 2) + cq] = v;
        }
    }
#ifdef AO_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    AO_ST(5);
}

void launch_step_board(const TreeParams& p, const StepNet& f, int rows, const int32_t* game_of_row, hipStream_t s) {
    const size_t lds = heads_lds_floats(p.A, f.planes) * sizeof(float);
    switch ((p.A + 63) / 64) {
        case 1: hipLaunchKernelGGL(k_step_board<1>, dim3(rows), dim3(1024), lds, s, p, f, game_of_row); break;
        case 2: hipLaunchKernelGGL(k_step_board<2>, dim3(rows), dim3(1024), lds, s, p, f, game_of_row); break;
        case 3: hipLaunchKernelGGL(k_step_board<3>, dim3(rows), dim3(1024), lds, s, p, f, game_of_row); break;
        default: hipLaunchKernelGGL(k_step_board<4>, dim3(rows), dim3(1024), lds, s, p, f, game_of_row); break;
    }
#ifdef AO_PROF
    if (getenv("AO_PROF_TREE")) {
        static int count = 0;
        if (++count % 97 == 0) {
            unsigned long long h[8];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(ao_prof_step), sizeof(h));
            fprintf(stderr, "AO_PROF k_step_board (game 0) ticks: heads %llu, barrier %llu, tree %llu, barrier %llu, conv1 %llu, total %llu\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[5] - h[0]);
            unsigned long long hh[8];
            (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(ao_prof_heads), sizeof(hh));
            fprintf(stderr, "AO_PROF   heads inside k_step_board: entry %llu, w3 %llu, 1x1 conv %llu, reduce %llu, FC %llu, softmax+value %llu, tanh/store %llu\n",
                    hh[0] - h[0], hh[1] - hh[0], hh[2] - hh[1], hh[3] - hh[2], hh[4] - hh[3], hh[5] - hh[4], hh[6] - hh[5]);
        }
    }
#endif
}

}  // namespace ao
