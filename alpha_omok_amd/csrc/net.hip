// net.hip -- policy/value ResNet forward (model.py:76-104 PVNet, eval mode) for gfx950: host side (weights,
// workspace, kernel selection, launches) and the ao_net_* C ABI. The kernels live in the headers included below:
//   net_trunk_h16.hpp  k_trunk16h / k_layer16h   fp32 results on split-fp16 MFMAs (default for 128 planes)
//   net_trunk_f32.hpp  k_trunk16 / k_layer16 / k_conv3x3   fp32 MFMAs
//   net_small.hpp      k_conv_cells, k_head_conv, k_head_fc, k_heads_board, k_nchw_to_il
//   net_device.hpp     device functions of the per-board path (tile conv, heads of one board)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/omok_hip.h"
#include "engine_types.hpp"
#include "net_device.hpp"
#include "net_common.hpp"
#include "net_trunk_f32.hpp"
#include "net_trunk_h16.hpp"
#include "net_layer_ksplit.hpp"
#include "net_board_h16.hpp"
#include "net_small.hpp"
#include "net_w16.hpp"


// ==============================================================================================
// host side
// ==============================================================================================
struct ao_net {
    int nb = 0, C = 0, planes = 0, B = 0, A = 0, device = 0;
    int nchq32 = 0;  // input channel quads of the layer-kernel path (groups of 32 boards)
    int nchq16 = 0;  // ... of the group-resident path (groups of 16 boards): multiple of 8
    int CQ = 0;
    int mode = 0;    // 0 auto, 1 layer kernels (32), 2 group-resident trunk, 3 per-board, 4 row-chunked layers (16)
    int requested_mode = 0;  // what ao_net_set_mode asked for; `mode` differs only while the fp16-range fallback holds the net on 2
    int fp16_fallbacks = 0;  // moves the engine repeated on the fp32-MFMA trunk since these weights were loaded (ao_net_finalize)
    bool forced_fp32 = false;
    int num_cu = 256;
    bool finalized = false;
    std::string err;
    std::map<std::string, std::vector<float>> params;
    std::vector<void*> allocs;                      // workspace (grow-only)
    // parameter buffers: ao_net_finalize runs again after every training step (weights re-exported), always with
    // the same sequence of sizes, so the buffers of the previous export are reused in order instead of ~60
    // hipFree + hipMalloc per export
    std::vector<std::pair<void*, size_t>> pallocs;
    size_t pcursor = 0;
    // device parameters
    std::vector<float*> conv_w, conv_sc, conv_sh;  // [1 + 2*nb]; conv_w[0] packed for nchq32
    // split-fp16 trunk (mode 5): per trunk conv after conv1 the high / low weight halves (pre-scaled by a
    // power of two) and the BatchNorm scale with that power of two folded back
    std::vector<uint4*> convh_wh, convh_wl;
    uint4 *step_w1h = nullptr, *step_w1l = nullptr;   // conv1 for the fused per-game step (StepNet): K = tap * 8 + plane, padded to 96
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
    int* d_status = nullptr;                       // bit 0: an activation left the fp16 range in the split-fp16 trunk
    bool attr_l[16][2] = {}, attr_done[16] = {}, lds_attr_done[16] = {}, attr_k[16] = {};
    // groups for which the trunk convs of the per-layer path run as k_layer16hk: [ksplit_min, ksplit_max] with four workgroups per
    // group (KS = 4: 768 .. 1024 boards, -2 .. -17 % against k_layer16h), (ksplit_max, ksplit_max2] with two (KS = 2: built and
    // correct, but no faster than k_layer16h at 1280 .. 2048 boards -- profiles/r4e_ksplit_medium_batches.txt -- so not planned
    // by default). AO_KSPLIT=lo,hi4,hi2 overrides (0,0,0 = off; 1,64,128 also plans KS = 2)
    int ksplit_min = 48, ksplit_max = 64, ksplit_max2 = 0;
    // ... and below them, down to the per-board path, as k_row16hk (one workgroup per group x output row x cout pair, two per CU):
    // search rate (tools/time_single_game.py --games n) +29 % at 48 games, +52 % at 64, +89 % at 96, +99 % at 128, +11 .. 19 % at
    // 192 .. 512 against the per-board path / k_layer16h (profiles/r4w_row_kernel_small_batches.txt). AO_ROWK="lo,hi".
    int rowk_min = 1, rowk_max = 47;
    bool attr_r[16] = {};
    // boards wider than 9: from this many boards on the trunk convs run as ONE launch of k_boardh (a workgroup = a board resident in
    // LDS through all layers, cells as the MFMA N dimension: net_board_h16.hpp); below it k_layer16h spreads a group over more
    // workgroups than there are boards (15x15, 10 blocks, forward: 0.64 / 0.65 / 0.66 / 0.68 ms at 32 / 64 / 96 / 128 boards -- one board's
    // time, the chip is not full -- against 0.60 / 0.67 / 0.82 / 0.88 for k_layer16h). AO_BOARDK=n overrides (0 = never).
    int boardk_min = 64;
    bool attr_b[16] = {};
    // boards x cells up to which the per-board path is planned (boards of 4x4 .. 9x9, with k_row16hk behind it: 32 boards of 9x9 --
    // it was 96 before that kernel; AO_PERBOARD_CELLS)
    long perboard_cells = 2592;
    int force_xt = 0, force_nch = 0;               // AO_XT / AO_NCH: tiling overrides for timing experiments (read at create)  // dynamic-LDS attribute set for this net's device
    // timing of the dominant kernel (trunk conv launches)
    bool timing = false;                           // THIS forward's launches are timed (see net_forward_il)
    bool timing_on = false;                        // ao_net_conv_timing(enable): every `timing_stride`-th forward is timed
    int timing_stride = 1;
    unsigned timing_tick = 0;
    static constexpr int kRing = 512;
    std::vector<hipEvent_t> ev0, ev1;
    int ring_head = 0, ring_count = 0;
    double ms_total = 0.0;
    int64_t launches = 0;
    int last_in_kind = 1;                          // input of the most recent forward: 1 fp32 plane batch, 2 the engine's bit planes
    // Two products instead of three (net_trunk_h16.hpp, W16): every conv weight x its layer's power of two is an fp16 number --
    // found out by ao_net_finalize, per export. products_req: 0 = use it when the weights allow, 3 = always three (ao_net_products)
    bool w16 = false;
    int products_req = 0;
    int trunk_fmt = -1;                            // activation format inside the resident split-fp16 trunk (kPairBytes): 0 = two fp16 halves (default), 1 = fp16 high half + one low byte (AO_TRUNK_FMT=1)

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

template <typename T>
static int param_alloc(ao_net* n, T** out, size_t count) {
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    if (n->pcursor < n->pallocs.size() && n->pallocs[n->pcursor].second == bytes) {
        *out = static_cast<T*>(n->pallocs[n->pcursor++].first);
        return 0;
    }
    void* p = nullptr;
    hipError_t st = hipMalloc(&p, bytes);
    if (st != hipSuccess) return n->fail(std::string("hipMalloc: ") + hipGetErrorString(st));
    if (n->pcursor < n->pallocs.size()) {
        (void)hipFree(n->pallocs[n->pcursor].first);
        n->pallocs[n->pcursor] = {p, bytes};
    } else {
        n->pallocs.push_back({p, bytes});
    }
    ++n->pcursor;
    *out = static_cast<T*>(p);
    return 0;
}

// uploads are queued on the null stream from a staging copy that lives until the end of ao_net_finalize
static int upload(ao_net* n, float** dst, const std::vector<float>& src) {
    if (param_alloc(n, dst, src.size())) return 1;
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
// mode 6: the per-layer split-fp16 kernels for EVERY batch size (never the resident trunk, never the per-board path), so
// that what evaluates a position -- and therefore a game's whole trajectory -- does not depend on how many other games
// share the batch. k_layer16h's arithmetic per output element is the same for any chunking and any neighbour boards.
static bool layers_only(const ao_net* n) { return n->mode == 6; }

static bool h16_supported(const ao_net* n);
static bool board_resident(const ao_net* n, int boards) {
    return n->mode != 6 && n->B >= 10 && n->nb >= 1 && n->boardk_min > 0 && boards >= n->boardk_min && h16_supported(n);
}

static bool h16_supported(const ao_net* n) {
    return n->planes == 128 && n->nb >= 1 && 1 + 2 * n->nb <= ao::kMaxTrunkLayers && n->nchq16 == 8;
}
// the two-product kernels run (the resident trunk in its default activation format, the trunk layers of k_layer16h / k_layer16hk<B, 4> /
// k_row16hk, the per-board path's k_conv_cells_h, k_boardh; what is left -- conv1 of the per-layer and per-board paths, k_layer16hk<B, 2> --
// keeps its three products: on such weights the same bits)
static bool two_products(const ao_net* n) { return n->w16 && n->products_req != 3; }

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
    if (mode == 6) mode = 5;   // mode 6 = mode 5 restricted to the per-layer kernels (layers_only()): one arithmetic for every batch size
    if (n->planes > 128) mode = 4;   // wide networks: the row-chunked fp32 layer kernels are the one path built for them
    if (mode == 0) {
        const long cells = static_cast<long>(boards) * n->A;
        // (end of round 3, with the activations split once at staging and the fused per-game step: 9x9 per-board vs per-layer
        // 841 vs 653 move-decisions/s at 64 games, 983 vs 960 at 96, 1076 vs 1281 at 128; 15x15 equal at 24 games, 581 vs 816 at 36)
        if (h16_supported(n)) mode = cells <= (n->B > 9 ? 5400 : n->B >= 4 ? n->perboard_cells : 7776) ? 3 : 5;
        else mode = cells <= 13000 ? 3 : (g16 >= 192 ? 2 : 4);
    }
    if (mode == 2 && (1 + 2 * n->nb > kMaxTrunkLayers)) mode = 4;
    int nch = 1;
    if (mode == 4) {
        // row chunks per group: minimise (rounds of workgroups over the CUs) x (rows per chunk)
        long best = -1;
        for (int c = 1; c <= n->B; ++c) {
            const long rounds = (static_cast<long>(g16) * c * ((n->planes + 127) / 128) + n->num_cu - 1) / n->num_cu;
            const long cost = rounds * ((n->B + c - 1) / c);
            if (best < 0 || cost < best) { best = cost; nch = c; }
        }
    }
    if (nch_out) *nch_out = nch;
    return mode;
}

int pick_mode_public(const ao_net* n, int boards) { return pick_mode(n, boards, nullptr); }

// kind (may be null): the input the planned kernels take -- 1 the interleaved fp32 plane batch, 2 the engine's bit
// planes ([boards][kPlaneRow] bytes, bit q = plane q; split-fp16 kernels, C <= 8): what ao_search's encoder writes
void net_plan(const ao_net* n, int boards, int* group, int* nchq, int* kind) {
    const int mode = pick_mode(n, boards, nullptr);
    if (kind) *kind = (mode == 5 && n->C <= 8) ? 2 : 1;
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

#if AO_KO == 14 || AO_KO == 15
// data-like knock-outs (net_trunk_h16.hpp, AO_KO): the activation buffers hold pseudo-random split-fp16 pairs from the start
__global__ void k_ko_fill(uint4* buf, size_t nquads) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < nquads; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        buf[i] = ao::ao_ko_fragment(static_cast<unsigned>(i), (i >> 6) & 1);
}
#endif

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
#if AO_KO == 14 || AO_KO == 15
    hipLaunchKernelGGL(k_ko_fill, dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4*>(n->act_x), act / 4);
    hipLaunchKernelGGL(k_ko_fill, dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4*>(n->act_t), act / 4);
    NET_HIP(n, hipDeviceSynchronize());
#endif
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
// The fused per-game step (step_kernels.hip) is available when this batch takes the per-board path with the split-fp16 tiles:
// fills `out` (heads, conv1 in its K = tap * C + plane form, the trunk buffer) and returns 1, else 0. AO_FUSED_STEP=0 turns it off.
int net_step_params(ao_net* n, int boards, float* policy, float* value, StepNet* out) {
    if (!n->finalized || !n->step_w1h || !h16_supported(n) || getenv("AO_CELLS_F32")) return 0;
    if (const char* v = getenv("AO_FUSED_STEP")) if (atoi(v) == 0) return 0;
    int group = 0, nchq = 0;
    net_plan(n, boards, &group, &nchq, nullptr);
    if (group != 1) return 0;
    if (hipSetDevice(n->device) != hipSuccess || ensure_workspace(n, boards)) return 0;
    out->heads = head_params(n);
    out->act = n->act_x;
    out->policy = policy;
    out->value = value;
    out->w1h = n->step_w1h;
    out->w1l = n->step_w1l;
    out->sc1 = n->convh_sc[0];
    out->sh1 = n->conv_sh[0];
    out->planes = n->planes;
    out->C = n->C;
    return 1;
}

// parts (per-board path only): 1 = conv1, 2 = the residual blocks, 4 = the heads -- the fused per-game step (k_step_board)
// computes conv1 and the heads itself and asks for the blocks alone.
// live / row_cap: rows handed out per simulation (engine_types.hpp TreeParams::live) -- `boards` is then the CAPACITY the kernels are
// launched for and the split-fp16 kernels skip the 16-board groups beyond the live count; null = all `boards` rows are live.
int net_forward_il(ao_net* n, const float* in_il, int boards, float* policy, float* value, hipStream_t s, int in_kind, int parts,
                   const unsigned* live, unsigned row_cap) {
    if (!n->finalized) return n->fail("ao_net_finalize has not been called");
    NET_HIP(n, hipSetDevice(n->device));
    if (ensure_workspace(n, boards)) return 1;
    // HIP-event timing of the conv launches: a pair of event records costs ~1 % of a 1.5 ms step when every launch carries one
    // (tools/time_move_phases.py --events); with a stride only every n-th forward is timed -- the mean is the same estimate
    n->timing = n->timing_on && (n->timing_tick++ % static_cast<unsigned>(n->timing_stride) == 0u);
    int group = 32, nchq = 0, nch = 1;
    bool heads_h16 = false;   // the separate head kernels read the split-fp16 layout
    net_plan(n, boards, &group, &nchq, nullptr);
    const int mode = pick_mode(n, boards, &nch);
    if (in_kind == 2 && !(mode == 5 && n->C <= 8)) return n->fail("bit planes handed to a kernel that takes the fp32 batch");
    n->last_in_kind = in_kind == 2 ? 2 : 1;
    const int groups = (boards + group - 1) / group;
    if (group != 1 && parts != 7) return n->fail("a partial forward exists on the per-board path only");
    if (group == 1) {
        // per-board NHWC path: one wave per (16 cells, 16 couts, board)
        auto conv = [&](int layer, const float* in, int cqi, const float* res, float* out) {
            // one tap per wave (with the board rows in LDS) for up to four boards; 128-plane networks take that form -- as
            // k_conv_cells_h, split-fp16 MFMAs -- for every batch of this path (6 ... 32 games: +10 ... +25 % over the
            // three-taps-per-wave fp32 form, profiles/r3h_single_game_attempt.txt)
            const int nw = (boards <= 4 || (h16_supported(n) && !getenv("AO_CELLS_F32"))) ? 9 : 3;
            const dim3 grid(((n->A + 15) / 16) * (n->planes / 16) * boards), block(64 * nw);
            const float4* w4 = reinterpret_cast<const float4*>(layer == 0 ? n->conv0_w1 : n->conv_w[layer]);
            const bool timed = n->timing && layer > 0;
            const int idx = timed ? timer_begin(n, s) : 0;
            const bool use_h = nw == 9 && layer > 0 && h16_supported(n) && !getenv("AO_CELLS_F32");
            if (use_h) {
                // boards per workgroup: the weights are loaded once per workgroup, so as many boards as still leave every CU
                // about three workgroups
                const int ntl = ((n->A + 15) / 16) * (n->planes / 16);
                int bpw = static_cast<int>((static_cast<long>(boards) * ntl) / (static_cast<long>(n->num_cu) * 3));
                if (const char* v = getenv("AO_BPW")) bpw = atoi(v);
                bpw = std::max(1, std::min(bpw, boards));
                const dim3 grid_h(ntl * ((boards + bpw - 1) / bpw));
                switch (n->B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W:                                                                                                            \
        if (two_products(n))                                                                                           \
            hipLaunchKernelGGL((k_conv_cells_h_w16<W, 8>), grid_h, dim3(64 * 12), 0, s, reinterpret_cast<const float4*>(in),   \
                               n->convh_wh[layer], reinterpret_cast<const float4*>(n->convh_sc[layer]),                \
                               reinterpret_cast<const float4*>(n->conv_sh[layer]), reinterpret_cast<const float4*>(res), \
                               reinterpret_cast<float4*>(out), cqi, n->planes, res ? 1 : 0, n->d_status, boards, bpw); \
        else                                                                                                           \
        hipLaunchKernelGGL((k_conv_cells_h<W, 8>), grid_h, dim3(64 * 12), 0, s, reinterpret_cast<const float4*>(in), n->convh_wh[layer],   \
                           n->convh_wl[layer], reinterpret_cast<const float4*>(n->convh_sc[layer]),                    \
                           reinterpret_cast<const float4*>(n->conv_sh[layer]), reinterpret_cast<const float4*>(res),   \
                           reinterpret_cast<float4*>(out), cqi, n->planes, res ? 1 : 0, n->d_status, boards, bpw);     \
        break;
                    AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
                    AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
                }
                if (timed) timer_end(n, idx, s);
                return;
            }
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
        if (parts & 1) conv(0, in_il, n->nchq1, nullptr, n->act_x);
        for (int i = 0; i < n->nb && (parts & 2); ++i) {
            conv(1 + 2 * i, n->act_x, n->CQ, nullptr, n->act_t);
            conv(2 + 2 * i, n->act_t, n->CQ, n->act_x, n->act_x);
        }
        const size_t lds1 = heads_lds_floats(n->A, n->planes) * sizeof(float);
        if (parts & 4)
            hipLaunchKernelGGL(k_heads_board, dim3(boards), dim3(1024), lds1, s, head_params(n),
                               reinterpret_cast<const float4*>(n->act_x), policy, value, n->A, n->planes);
        NET_HIP(n, hipGetLastError());
#ifdef AO_PROF
        if (getenv("AO_PROF_PRINT")) {
            unsigned long long h[8];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(ao_prof_heads), sizeof(h));
            unsigned long long cv[8];
            hipMemcpyFromSymbol(cv, HIP_SYMBOL(ao_prof_conv), sizeof(cv));
            fprintf(stderr, "AO_PROF k_conv_cells_h (last layer, block 0, wave 0) ticks: rows loaded + split + stored %llu, barrier + LDS reads + MFMAs %llu, reduction barrier %llu, reduce+epilogue+store %llu, total %llu\n",
                    cv[1] - cv[0], cv[2] - cv[1], cv[3] - cv[2], cv[4] - cv[3], cv[4] - cv[0]);
            fprintf(stderr, "AO_PROF k_heads_board ticks: w3 %llu, 1x1 conv %llu, reduce %llu, FC %llu, softmax+value %llu, tanh/store %llu, total %llu\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
        }
#endif
        return 0;
#ifdef AO_PROF
    } else if (group == 16 && mode == 5 && (layers_only(n) || !(n->B <= 9 && (groups >= 192 || getenv("AO_FORCE_RESIDENT"))))) {
#else
    } else if (group == 16 && mode == 5 && (layers_only(n) || !(n->B <= 9 && groups >= 192))) {
#endif
        // split-fp16 trunk, one launch per conv: workgroup = (16-board group, row chunk, column tile). For batches
        // that cannot give every CU a whole group, and for boards wider than 9 (a staged row must fit LDS twice)
        // Tiling of a wide board: column tiles of XT = 5 or 4 cells (+ a halo column each side) x row chunks. Every
        // workgroup fills a CU (8 waves x ~250 registers), so the plan is judged by rounds of workgroups over the CUs
        // x the work of one workgroup: its MFMA cells (rows x XT) plus, weighted, the cells it stages (halo rows and
        // columns included). 15 x 15 at 64 groups: XT = 4 gives 256 workgroups that each carry all 15 rows of their
        // tile (one round, no row halos, weights fetched once per group and tile); XT = 5 needs 768 workgroups of 4
        // rows (three rounds, two extra staged rows each).
        int xt = n->B, nchh = 1;
        {
            double best = -1.0;
            const int xts[2] = {5, 4};
            for (int k = 0; k < (n->B <= 9 ? 1 : 2); ++k) {
                const int x = n->B <= 9 ? n->B : xts[k];
                const int ntile = (n->B + x - 1) / x;
                const int staged_cols = n->B <= 9 ? x : x + 2;
                for (int c = 1; c <= n->B; ++c) {
                    const long rounds = (static_cast<long>(groups) * c * ntile + n->num_cu - 1) / n->num_cu;
                    const int rows = (n->B + c - 1) / c;
                    const int staged_rows = rows + (c > 1 ? 2 : 0);
                    const double cost = rounds * (static_cast<double>(rows) * x + 0.25 * staged_rows * staged_cols + 2.0);
                    if (best < 0 || cost < best) { best = cost; nchh = c; xt = x; }
                }
            }
            if (n->force_xt > 0 && n->B > 9) xt = n->force_xt;
            if (n->force_nch > 0) nchh = n->force_nch;
        }
        const int nxt = (n->B + xt - 1) / xt;
        // medium batches of boards up to 9x9: the trunk convs split a group by cout pairs over four workgroups whose
        // waves split the contraction (net_layer_ksplit.hpp); conv1 stays with k_layer16h. Never in mode 6 (one
        // arithmetic for every batch size).
        const int ks = (!layers_only(n) && n->B >= 4 && n->B <= 9 && groups >= n->ksplit_min) ? (groups <= n->ksplit_max ? 4 : groups <= n->ksplit_max2 ? 2 : 0) : 0;
        const bool rowk = !layers_only(n) && n->B >= 4 && n->B <= 9 && groups >= n->rowk_min && groups <= n->rowk_max;
        const bool ksplit = ks != 0 || rowk;
        auto layer_k = [&](int l) -> int {
            LayerHArgs a;
            a.src = static_cast<const void*>((l & 1) ? n->act_x : n->act_t);
            a.dst = reinterpret_cast<uint4*>(!(l & 1) ? n->act_x : n->act_t);
            a.layer.wh = n->convh_wh[l];
            a.layer.wl = n->convh_wl[l];
            a.layer.sc = reinterpret_cast<const float4*>(n->convh_sc[l]);
            a.layer.sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
            a.layer.ovf = n->d_status;
            a.res = !(l & 1) ? 1 : 0;
            a.nch = groups;
            a.live = live; a.row_cap = row_cap;
            const dim3 grid(rowk ? (groups + 7) / 8 * 8 * 4 * n->B : (groups + 7) / 8 * 8 * ks), block(512);
            const int idx = n->timing ? timer_begin(n, s) : 0;
            if (two_products(n) && (rowk || ks == 4)) {
                if (rowk) NET_HIP(n, ao::launch_row16hk_w16(n->device, n->B, grid, s, a));
                else NET_HIP(n, ao::launch_layer16hk_w16(n->device, n->B, grid, s, a));
                if (n->timing) timer_end(n, idx, s);
                return 0;
            }
            if (rowk) {
                switch (n->B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t lds_ = static_cast<size_t>(W) * 8 * 1024;                                                     \
        if (!n->attr_r[W]) {                                                                                           \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_row16hk<W>),                               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            n->attr_r[W] = true;                                                                                       \
        }                                                                                                              \
        hipLaunchKernelGGL((k_row16hk<W>), grid, block, lds_, s, a);                                                   \
    } break;
                    AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
#undef AO_BW_CASE
                    default: return n->fail("k_row16hk: board outside 4 .. 9");
                }
                if (n->timing) timer_end(n, idx, s);
                return 0;
            }
            switch (n->B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t lds_ = static_cast<size_t>(2) * W * 8 * 1024;                                                 \
        if (!n->attr_k[W]) {                                                                                           \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16hk<W, 4>),                          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16hk<W, 2>),                          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            n->attr_k[W] = true;                                                                                       \
        }                                                                                                              \
        if (ks == 4) hipLaunchKernelGGL((k_layer16hk<W, 4>), grid, block, lds_, s, a);                                 \
        else hipLaunchKernelGGL((k_layer16hk<W, 2>), grid, block, lds_, s, a);                                         \
    } break;
                AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
#undef AO_BW_CASE
                default: return n->fail("k_layer16hk: board outside 4 .. 9");
            }
            if (n->timing) timer_end(n, idx, s);
            return 0;
        };
        auto layer = [&](int l) -> int {
            if (ksplit && l > 0) return layer_k(l);
            LayerHArgs a;
            a.src = l == 0 ? static_cast<const void*>(in_il) : static_cast<const void*>((l & 1) ? n->act_x : n->act_t);
            a.dst = reinterpret_cast<uint4*>((l == 0 || !(l & 1)) ? n->act_x : n->act_t);
            a.layer.wh = n->convh_wh[l];
            a.layer.wl = n->convh_wl[l];
            a.layer.sc = reinterpret_cast<const float4*>(n->convh_sc[l]);
            a.layer.sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
            a.layer.ovf = n->d_status;
            a.res = (l > 0 && !(l & 1)) ? 1 : 0;
            a.nch = nchh;
            a.live = live; a.row_cap = row_cap;
            const dim3 grid(groups * nchh * nxt), block(512);
            const bool timed = n->timing && l > 0;
            const int idx = timed ? timer_begin(n, s) : 0;
            if (l > 0 && two_products(n)) {
                NET_HIP(n, ao::launch_layer16h_w16(n->device, n->B, xt, grid, s, a));
                if (timed) timer_end(n, idx, s);
                return 0;
            }
#define AO_LAYERH_LAUNCH(W, XT_)                                                                                       \
    do {                                                                                                               \
        constexpr int NX_ = (XT_ < W) ? XT_ + 2 : XT_;                                                                 \
        constexpr size_t lds_ = static_cast<size_t>(2) * NX_ * 4 * 2 * 1024;                                           \
        if (!n->attr_l[W][XT_ == 4]) {                                                                                 \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16h<W, XT_, 4, 0>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16h<W, XT_, 4, 1>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer16h<W, XT_, 4, 2>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            n->attr_l[W][XT_ == 4] = true;                                                                             \
        }                                                                                                              \
        if (l == 0 && in_kind == 2) hipLaunchKernelGGL((k_layer16h<W, XT_, 4, 2>), grid, block, lds_, s, a);          \
        else if (l == 0) hipLaunchKernelGGL((k_layer16h<W, XT_, 4, 1>), grid, block, lds_, s, a);                     \
        else hipLaunchKernelGGL((k_layer16h<W, XT_, 4, 0>), grid, block, lds_, s, a);                                 \
    } while (0)
            switch (n->B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        if (W <= 9) AO_LAYERH_LAUNCH(W, (W <= 9 ? W : 5));                                                             \
        else if (xt == 4) AO_LAYERH_LAUNCH(W, (W <= 9 ? W : 4));                                                       \
        else AO_LAYERH_LAUNCH(W, (W <= 9 ? W : 5));                                                                    \
    } break;
                AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
                AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
#undef AO_LAYERH_LAUNCH
            }
            if (timed) timer_end(n, idx, s);
            return 0;
        };
        if (board_resident(n, boards)) {
            // ONE launch: a workgroup per board carries it from the bit planes through conv1, all trunk convs and both heads
            // (net_board_h16.hpp). Float planes (ao_net_forward): conv1 as k_layer16h first, its output gathered by the kernel.
            const bool bits = in_kind == 2;
            if (!bits && layer(0)) return 1;
            BoardHArgs a;
            a.act = reinterpret_cast<const uint4*>(n->act_x);
            a.planes = reinterpret_cast<const uint8_t*>(in_il);
            a.hbuf = n->hbuf;
            a.w3 = n->head_w3; a.sc3 = n->head_sc3; a.sh3 = n->head_sh3;
            a.nlayers = 1 + 2 * n->nb;
            a.nboards = boards;
            a.live = live; a.row_cap = row_cap;
            for (int l = 0; l < a.nlayers; ++l) {
                a.layers[l].wh = n->convh_wh[l];
                a.layers[l].wl = n->convh_wl[l];
                a.layers[l].sc = reinterpret_cast<const float4*>(n->convh_sc[l]);
                a.layers[l].sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
                a.layers[l].ovf = n->d_status;
            }
            // (the kernel deals boards to workgroups in rounds of 8 groups -- a group's 16 boards on one XCD --, so the grid covers whole rounds)
            const dim3 grid(std::min((boards + 127) / 128 * 128, n->num_cu)), block(512);
            const int idx = n->timing ? timer_begin(n, s) : 0;
            if (two_products(n)) {
                NET_HIP(n, ao::launch_boardh_w16(n->device, n->B, bits, grid, s, a));
            } else
            switch (n->B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t lds_ = static_cast<size_t>(W) * 8 * 1024;                                                     \
        if (!n->attr_b[W]) {                                                                                           \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_boardh<W, 1>),                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_boardh<W, 2>),                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_)));       \
            n->attr_b[W] = true;                                                                                       \
        }                                                                                                              \
        if (bits) hipLaunchKernelGGL((k_boardh<W, 2>), grid, block, lds_, s, a);                                       \
        else hipLaunchKernelGGL((k_boardh<W, 1>), grid, block, lds_, s, a);                                            \
    } break;
                AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
                default: return n->fail("k_boardh: board outside 10 .. 15");
            }
            if (n->timing) timer_end(n, idx, s);
            NET_HIP(n, hipGetLastError());
            // the FC layers of the heads on the kernel's hbuf (the 1x1 head convs ran in its last epilogue)
            const size_t lds1 = (static_cast<size_t>(4) * n->A + n->planes + 8) * sizeof(float);
            hipLaunchKernelGGL(k_head_fc, dim3(boards), dim3(256), lds1, s, n->hbuf, n->wp_t, n->bp, n->w1_t, n->b1, n->w2, n->b2, policy, value,
                               n->A, n->planes);
            NET_HIP(n, hipGetLastError());
            return 0;
        } else {
            for (int l = 0; l <= 2 * n->nb; ++l)
                if (layer(l)) return 1;
        }
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
        a.live = live; a.row_cap = row_cap;
        for (int l = 0; l < a.nlayers; ++l) {
            a.layers[l].wh = n->convh_wh[l];
            a.layers[l].wl = n->convh_wl[l];
            a.layers[l].sc = reinterpret_cast<const float4*>(n->convh_sc[l]);
            a.layers[l].sh = reinterpret_cast<const float4*>(n->conv_sh[l]);
            a.layers[l].ovf = n->d_status;
        }
        const int idx = n->timing ? timer_begin(n, s) : 0;
        if (two_products(n) && n->trunk_fmt != 1) {
            if (n->B > 9) return n->fail("split-fp16 trunk: board larger than 9x9");
            NET_HIP(n, ao::launch_trunk16h_w16(n->device, n->B, in_kind, groups, s, a));
        } else
        switch (n->B) {
#define AO_BW_CASE(W)                                                                                        \
    case W: {                                                                                                \
        constexpr size_t heads_ = (static_cast<size_t>(3) * 128 + 16 * 3 * W * W + 8 * 16 * W * W + 8 * 16 * 128) * 4;  \
        constexpr size_t rows_ = static_cast<size_t>(2) * W * 4 * 2 * 1024 + 64;   /* two row buffers + the split-barrier counter */ \
        constexpr size_t lds_ = (rows_ > heads_) ? rows_ : heads_;                                            \
        if (!n->attr_done[W]) {                                                                                 \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16h<W, 4, 0>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_))); \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16hb<W, 4, 0>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_))); \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16h<W, 4, 1>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_))); \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16hb<W, 4, 1>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_))); \
            n->attr_done[W] = true;                                                                           \
        }                                                                                                    \
        if (n->trunk_fmt == 1) {                                                                             \
            if (in_kind == 2) hipLaunchKernelGGL((k_trunk16hb<W, 4, 1>), dim3(groups), dim3(512), lds_, s, a);    \
            else hipLaunchKernelGGL((k_trunk16h<W, 4, 1>), dim3(groups), dim3(512), lds_, s, a);                 \
        } else {                                                                                             \
            if (in_kind == 2) hipLaunchKernelGGL((k_trunk16hb<W, 4, 0>), dim3(groups), dim3(512), lds_, s, a);    \
            else hipLaunchKernelGGL((k_trunk16h<W, 4, 0>), dim3(groups), dim3(512), lds_, s, a);                 \
        }                                                                                                    \
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
            a.nsplit = (n->planes + 127) / 128;           // output channels beyond 128: a second workgroup per (group, chunk)
            const dim3 grid(groups * nch * a.nsplit), block(64 * std::min(8, n->planes / 16));
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
        switch (n->B) {
#define AO_BW_CASE(W)                                                                                        \
    case W: {                                                                                                \
        constexpr int XT_ = (W <= 9) ? W : 5;                                                                \
        if (!n->lds_attr_done[W]) {                                                                             \
            NET_HIP(n, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunk16<W, XT_, 1>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));          \
        }                                                                                                    \
        n->lds_attr_done[W] = true;                                                                             \
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

namespace ao {
// The engine's fp16-range recovery (ao_search): the move is repeated on the fp32-MFMA trunk (mode 2). The fallback
// belongs to the WEIGHTS that overflowed: it is counted per network object since its last ao_net_finalize; from the
// third repeated move on the network stays on the fp32-MFMA trunk -- until new weights are loaded (ao_net_finalize
// returns it to the mode ao_net_set_mode asked for) or ao_net_set_mode is called again.
void net_fp16_fallback_begin(ao_net* n) {
    ++n->fp16_fallbacks;
    n->mode = 2;
}
int net_fp16_fallback_end(ao_net* n) {
    n->forced_fp32 = n->fp16_fallbacks >= 3;
    if (!n->forced_fp32) n->mode = n->requested_mode;
    return n->forced_fp32 ? 1 : 0;
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
    // 32 .. 128 planes: every path (128: the split-fp16 MFMA kernels). 160 .. 512: the row-chunked fp32-MFMA layer kernels
    // (mode 4) for every batch size, a group's output channels split over two .. four workgroups (k_layer16) -- model.py:76-85
    // takes any `planes`: other widths reach this call zero-padded to the next multiple of 32 (pvnet.pad_state_dict); beyond 512
    // the caller's torch module evaluates (alpha_omok_amd/evaluator.py: a RuntimeWarning, or an error with strict_native)
    if (planes < 32 || planes > 512 || planes % 32) return bad("planes must be a multiple of 32 in 32 .. 512");
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
    if (const char* v = getenv("AO_KSPLIT")) {   // "lo,hi4,hi2": groups that take k_layer16hk (timing experiments; "0,0,0" turns it off)
        int lo = 0, hi = 0, hi2 = 0;
        if (sscanf(v, "%d,%d,%d", &lo, &hi, &hi2) == 3) { n->ksplit_min = lo > 0 ? lo : 1 << 30; n->ksplit_max = hi; n->ksplit_max2 = hi2; }
    }
    if (const char* v = getenv("AO_PERBOARD_CELLS")) n->perboard_cells = atol(v);
    if (const char* v = getenv("AO_BOARDK")) n->boardk_min = atoi(v);   // boards from which k_boardh carries the trunk of boards wider than 9 (0: never)
    if (const char* v = getenv("AO_ROWK")) {   // "lo,hi": groups that take k_row16hk ("0,-1" turns it off)
        int lo = 0, hi = -1;
        if (sscanf(v, "%d,%d", &lo, &hi) == 2) { n->rowk_min = lo; n->rowk_max = hi; }
    }
    if (const char* v = getenv("AO_TRUNK_FMT")) n->trunk_fmt = atoi(v) == 1 ? 1 : 0;
    // Default: two fp16 halves (4 bytes, ~22 significand bits). The 3-byte format (19 bits) is 2 % faster and costs ~4-8 x the
    // rounding error: on a TRAINED 4-block network (profiles/r4_trained_net_check.json, 4096 real self-play positions) 2.8e-5 on
    // the policy against torch fp32 where the 4-byte format has 7.8e-6 -- inside the 1e-4 bar, but 3.5 x instead of 13 x away
    // from it, for a gain below the box-to-box spread. It stays available as AO_TRUNK_FMT=1.
    if (n->trunk_fmt < 0) n->trunk_fmt = 0;
    if (const char* v = getenv("AO_XT")) n->force_xt = atoi(v) == 4 ? 4 : (atoi(v) == 5 ? 5 : 0);
    if (const char* v = getenv("AO_NCH")) n->force_nch = atoi(v) > 0 && atoi(v) <= board ? atoi(v) : 0;
    *out = n;
    return 0;
}

void ao_net_destroy(ao_net* n) {
    if (!n) return;
    hipSetDevice(n->device);
    hipDeviceSynchronize();
    for (void* p : n->allocs) hipFree(p);
    for (auto& pa : n->pallocs) hipFree(pa.first);
    for (auto e : n->ev0) hipEventDestroy(e);
    for (auto e : n->ev1) hipEventDestroy(e);
    delete n;
}

int ao_net_set_mode(ao_net* n, int mode) {
    if (mode < 0 || mode > 6)
        return n->fail("mode must be 0 (auto), 1 (layer kernels), 2 (group-resident trunk), 3 (per-board), 4 (row-chunked), 5 (split-fp16 trunk) "
                       "or 6 (split-fp16 per-layer kernels for every batch size)");
    if ((mode == 5 || mode == 6) && !h16_supported(n))
        return n->fail("modes 5 / 6 (split-fp16 MFMA trunk) need 128 planes and at least one ResBlock");
    n->mode = n->requested_mode = mode;
    n->forced_fp32 = false;
    return 0;
}


int ao_net_get_mode(const ao_net* n) { return n->mode; }

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
    static const int wl_drop_bits = [] { const char* v = getenv("AO_WL_BITS"); const int b = v ? atoi(v) : 11; return b >= 1 && b < 11 ? 11 - b : 0; }();
    const int nc32 = cin / 32, nt = cout / 16;
    hi->assign(static_cast<size_t>(9) * nc32 * nt * 64 * 8, 0);
    lo->assign(hi->size(), 0);
    const float scale = std::ldexp(1.0f, s);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const float v = w[(static_cast<size_t>(co) * cin + ci) * 9 + t] * scale;
                const _Float16 h = static_cast<_Float16>(v);
                _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
                if (wl_drop_bits > 0) {   // experiment (AO_WL_BITS): round the low half to fewer significant bits (less MFMA operand activity)
                    uint16_t b;
                    std::memcpy(&b, &l, 2);
                    const uint16_t half = static_cast<uint16_t>(1u << (wl_drop_bits - 1));
                    b = static_cast<uint16_t>((b + (half - 1) + ((b >> wl_drop_bits) & 1u)) & ~((1u << wl_drop_bits) - 1u));
                    std::memcpy(&l, &b, 2);
                }
                const size_t idx = ((((static_cast<size_t>(t) * nc32 + ci / 32) * nt + co / 16) * 4 + (ci % 32) / 8) * 16 +
                                    co % 16) * 8 + ci % 8;
                std::memcpy(&(*hi)[idx], &h, 2);
                std::memcpy(&(*lo)[idx], &l, 2);
            }
}


int ao_net_finalize(ao_net* n) {
    NET_HIP(n, hipSetDevice(n->device));
    NET_HIP(n, hipDeviceSynchronize());   // nothing may still read the buffers that are overwritten below
    n->pcursor = 0;
    n->fp16_fallbacks = 0;                // new weights: what the old ones did to the fp16 range says nothing about these
    n->forced_fp32 = false;
    n->mode = n->requested_mode;
    n->conv_w.clear(); n->conv_sc.clear(); n->conv_sh.clear();
    n->convh_wh.clear(); n->convh_wl.clear(); n->convh_sc.clear();
    if (param_alloc(n, &n->d_status, 4)) return 1;
    NET_HIP(n, hipMemset(n->d_status, 0, 16));
    const int P = n->planes, A = n->A;
    // The repacking of the conv weights (strided scatters over 147 k weights per layer and layout) is the bulk of a
    // re-export: it runs on one host thread per layer, the uploads follow in a fixed order (param_alloc's sequence).
    const int nl = 1 + 2 * n->nb;
    const bool h16 = h16_supported(n);
    struct LayerPack {
        const std::vector<float>* w = nullptr;
        std::string bn;
        int cin = 0;
        std::vector<float> w32, w16, w1;   // fp32 layouts: [tap][cq][cout][4] for nchq32 / CQ (and conv1: nchq16, nchq1)
        std::vector<uint16_t> hi, lo;      // split-fp16 layout
        int sft = 0;
    };
    std::vector<LayerPack> pk(nl);
    for (int l = 0; l < nl; ++l) {
        const std::string pre = "layers." + std::to_string(l ? (l - 1) / 2 : 0);
        const std::string cname = l == 0 ? std::string("conv1.weight") : pre + ((l & 1) ? ".conv1.weight" : ".conv2.weight");
        pk[l].bn = l == 0 ? std::string("bn1") : pre + ((l & 1) ? ".bn1" : ".bn2");
        pk[l].cin = l == 0 ? n->C : P;
        if (get_param(n, cname, static_cast<size_t>(P) * pk[l].cin * 9, &pk[l].w)) return 1;
    }
    {
        std::vector<std::thread> th;
        for (int l = 0; l < nl; ++l)
            th.emplace_back([&, l]() {
                LayerPack& k = pk[l];
                const std::vector<float>& w0 = *k.w;
                k.w32 = pack_conv(w0, P, k.cin, l == 0 ? n->nchq32 : n->CQ);
                if (l == 0) {
                    k.w16 = pack_conv(w0, P, n->C, n->nchq16);
                    k.w1 = pack_conv(w0, P, n->C, n->nchq1);
                }
                if (!h16) return;
                // conv1: input channels zero-padded to one 32-channel block
                std::vector<float> wpad;
                const std::vector<float>* w = &w0;
                const int cinp = l == 0 ? 32 : P;
                if (l == 0) {
                    wpad.assign(static_cast<size_t>(P) * 32 * 9, 0.f);
                    for (int co = 0; co < P; ++co)
                        for (int ci = 0; ci < n->C; ++ci)
                            for (int t = 0; t < 9; ++t)
                                wpad[(static_cast<size_t>(co) * 32 + ci) * 9 + t] = w0[(static_cast<size_t>(co) * n->C + ci) * 9 + t];
                    w = &wpad;
                }
                float mx = 0.f;
                for (float v : *w) mx = std::max(mx, std::fabs(v));
                // power-of-two pre-scale: largest |w| lands in [4, 8), so the low halves are normal fp16 numbers
                k.sft = (mx > 0.f && std::isfinite(mx)) ? 2 - static_cast<int>(std::floor(std::log2(mx))) : 0;
                pack_conv_h(*w, P, cinp, k.sft, &k.hi, &k.lo);
            });
        for (auto& t : th) t.join();
    }
    for (int l = 0; l < nl; ++l) {
        std::vector<float> sc, sh;
        if (fold_bn(n, pk[l].bn, P, &sc, &sh)) return 1;
        float *dw, *dsc, *dsh;
        if (upload(n, &dw, pk[l].w32) || upload(n, &dsc, sc) || upload(n, &dsh, sh)) return 1;
        n->conv_w.push_back(dw); n->conv_sc.push_back(dsc); n->conv_sh.push_back(dsh);
        if (l == 0 && (upload(n, &n->conv0_w16, pk[0].w16) || upload(n, &n->conv0_w1, pk[0].w1))) return 1;
    }
    n->w16 = false;
    if (h16) {
        // all low halves zero <=> w * 2^sft is an fp16 number for every conv weight of the network (conv1 included)
        bool all_zero = true;
        for (int l = 0; l < nl && all_zero; ++l)
            for (uint16_t v : pk[l].lo)
                if (v & 0x7fffu) { all_zero = false; break; }
        n->w16 = all_zero;
        for (int l = 0; l < nl; ++l) {
            std::vector<float> sc, sh;
            if (fold_bn(n, pk[l].bn, P, &sc, &sh)) return 1;
            for (float& v : sc) v = std::ldexp(v, -pk[l].sft);
            uint16_t *dh = nullptr, *dl = nullptr;
            float* dsc = nullptr;
            if (param_alloc(n, &dh, pk[l].hi.size()) || param_alloc(n, &dl, pk[l].lo.size())) return 1;
            NET_HIP(n, hipMemcpy(dh, pk[l].hi.data(), pk[l].hi.size() * 2, hipMemcpyHostToDevice));
            NET_HIP(n, hipMemcpy(dl, pk[l].lo.data(), pk[l].lo.size() * 2, hipMemcpyHostToDevice));
            if (upload(n, &dsc, sc)) return 1;
            n->convh_wh.push_back(reinterpret_cast<uint4*>(dh));
            n->convh_wl.push_back(reinterpret_cast<uint4*>(dl));
            n->convh_sc.push_back(dsc);
        }
    }
    n->step_w1h = n->step_w1l = nullptr;
    if (h16 && n->C <= 8) {
        // conv1 once more for the fused per-game step: the contraction index k = tap * 8 + plane (three 32-deep MFMA steps
        // instead of one per tap; a lane's 8 values are the planes of one neighbour cell), same pre-scale as convh_wh[0]
        const int nt = P / 16;
        std::vector<uint16_t> hi(static_cast<size_t>(3) * nt * 64 * 8, 0), lo(hi.size(), 0);
        const float scale = std::ldexp(1.0f, pk[0].sft);
        for (int co = 0; co < P; ++co)
            for (int c = 0; c < n->C; ++c)
                for (int t = 0; t < 9; ++t) {
                    const float v = (*pk[0].w)[(static_cast<size_t>(co) * n->C + c) * 9 + t] * scale;
                    const _Float16 h = static_cast<_Float16>(v);
                    const _Float16 l = static_cast<_Float16>(v - static_cast<float>(h));
                    const int kk = t * 8 + c;
                    const size_t idx = ((((static_cast<size_t>(kk / 32) * nt + co / 16) * 4 + (kk % 32) / 8) * 16 + co % 16) * 8) + kk % 8;
                    std::memcpy(&hi[idx], &h, 2);
                    std::memcpy(&lo[idx], &l, 2);
                }
        uint16_t *dh = nullptr, *dl = nullptr;
        if (param_alloc(n, &dh, hi.size()) || param_alloc(n, &dl, lo.size())) return 1;
        NET_HIP(n, hipMemcpy(dh, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
        NET_HIP(n, hipMemcpy(dl, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
        n->step_w1h = reinterpret_cast<uint4*>(dh);
        n->step_w1l = reinterpret_cast<uint4*>(dl);
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
    ao::net_plan(n, batch, &group, &nchq, nullptr);
    const int boards = (batch + group - 1) / group * group;
    // the heads write rows for the padding boards too: run into scratch unless the batch is whole
    float* pol = (boards == batch) ? dev_policy : n->tmp_p;
    float* val = (boards == batch) ? dev_value : n->tmp_v;
    const size_t total = static_cast<size_t>(boards) * n->A;
    hipLaunchKernelGGL(ao::k_nchw_to_il, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s,
                       dev_planes_nchw, reinterpret_cast<float4*>(n->il_in), batch, n->C, n->A, nchq, boards, group);
    if (ao::net_forward_il(n, n->il_in, batch, pol, val, s, 1, 7, nullptr, 0u)) return 1;
    if (boards != batch) {
        NET_HIP(n, hipMemcpyAsync(dev_policy, n->tmp_p, sizeof(float) * batch * n->A, hipMemcpyDeviceToDevice, s));
        NET_HIP(n, hipMemcpyAsync(dev_value, n->tmp_v, sizeof(float) * batch, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int ao_net_status(ao_net* n, void* stream, int32_t* flags, int clear) {
    if (!n->finalized) return n->fail("ao_net_finalize has not been called");
    NET_HIP(n, hipSetDevice(n->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int32_t f = 0;
    NET_HIP(n, hipMemcpyAsync(&f, n->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    NET_HIP(n, hipStreamSynchronize(s));
    if (clear && f) {
        NET_HIP(n, hipMemsetAsync(n->d_status, 0, sizeof(int32_t), s));
        NET_HIP(n, hipStreamSynchronize(s));
    }
    if (flags) *flags = f;
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
    n->timing_on = enable != 0;
    n->timing_stride = enable > 1 ? enable : 1;
    n->timing_tick = 0;
    n->timing = false;
    return 0;
}

// Name (as rocprofv3 prints it, template arguments included) and algorithmic FLOPs per launch of the kernel that
// carries the conv stack for a batch of `boards` positions; in_kind 2 = the engine's bit planes (ao_search), 1 = the
// fp32 plane batch (ao_net_forward). Pure host logic: no device is touched.
static void dominant_name(const ao_net* n, int boards, int in_kind, std::string* nm_out, double* f_out) {
    int group = 32, nchq = 0;
    ao::net_plan(n, boards, &group, &nchq, nullptr);
    const int padded = (boards + group - 1) / group * group;
    const double conv = 2.0 * n->A * 9.0 * n->planes * n->planes * padded;   // one planes->planes 3x3 conv
    const double conv1 = 2.0 * n->A * 9.0 * n->C * n->planes * padded;
    const int mode = ao::pick_mode_public(n, boards);
    const std::string bw = std::to_string(n->B);
    std::string nm;
    double f;
    if (group == 1) {
        nm = h16_supported(n) ? std::string(two_products(n) ? "k_conv_cells_h_w16<" : "k_conv_cells_h<") + bw + ", 8> (one 3x3 conv, per-board NHWC, board rows in LDS, split-fp16 MFMA 16x16x32, " +
                                    (two_products(n) ? "2 products: the conv weights are fp16 numbers)" : "3 products)")
                              : "k_conv_cells<" + bw + "> (one 3x3 conv, per-board NHWC, fp32 MFMA 16x16x4)";
        f = conv;
    } else if (group == 16 && mode == 4) {
        nm = "k_layer16<" + bw + "> (one 3x3 conv per launch, 16-board groups x row chunks, fp32 MFMA 16x16x4)";
        f = conv;
    } else if (group == 16 && mode == 5 && !layers_only(n) && n->B >= 4 && n->B <= 9 && (boards + 15) / 16 >= n->rowk_min &&
               (boards + 15) / 16 <= n->rowk_max) {
        nm = std::string(two_products(n) ? "k_row16hk_w16<" : "k_row16hk<") + bw + "> (one 3x3 conv per launch as split-fp16 MFMA 16x16x32 (" +
             (two_products(n) ? "2 products: the conv weights are fp16 numbers" : "3 products") + ", fp32 accumulate): one workgroup per "
             "16-board group x output row x cout pair, waves split the contraction by input block, partial tiles exchanged through LDS)";
        f = conv;
    } else if (group == 16 && mode == 5 && !layers_only(n) && n->B >= 4 && n->B <= 9 && (boards + 15) / 16 >= n->ksplit_min &&
               (boards + 15) / 16 <= std::max(n->ksplit_max, n->ksplit_max2)) {
        const bool four = (boards + 15) / 16 <= n->ksplit_max;
        const bool two = two_products(n) && four;
        nm = std::string(two ? "k_layer16hk_w16<" : "k_layer16hk<") + bw + (four ? ", 4" : ", 2") + "> (one 3x3 conv per launch as split-fp16 MFMA 16x16x32 (" +
             (two ? "2 products: the conv weights are fp16 numbers" : "3 products") + ", fp32 accumulate): "
             "a 16-board group split over " + (four ? "four workgroups by cout pairs" : "two workgroups by cout quads") +
             ", waves split the contraction by input block, partial tiles exchanged through LDS)";
        f = conv;
    } else if (group == 16 && mode == 5 && board_resident(n, boards)) {
        nm = std::string(two_products(n) ? "k_boardh_w16<" : "k_boardh<") + bw + (in_kind == 2 ? ", 2" : ", 1") + "> (" + (in_kind == 2 ? "conv1 + " : "") + std::to_string(2 * n->nb) +
             " 3x3 convs in one launch as split-fp16 MFMA 16x16x32 (" + (two_products(n) ? "2 products: the conv weights are fp16 numbers" : "3 products") + ", fp32 accumulate): "
             "a workgroup = one board resident in LDS through all layers, the row's cells as the MFMA N dimension, column shifts as DPP row shifts)";
        f = 2.0 * n->nb * 2.0 * n->A * 9.0 * n->planes * n->planes * boards + (in_kind == 2 ? 2.0 * n->A * 9.0 * n->C * n->planes * boards : 0.0);
    } else if (group == 16 && mode == 5 && (layers_only(n) || !(n->B <= 9 && (boards + 15) / 16 >= 192))) {
        nm = std::string(two_products(n) ? "k_layer16h_w16<" : "k_layer16h<") + bw + "> (one 3x3 conv per launch as split-fp16 MFMA 16x16x32 (" +
             (two_products(n) ? "2 products: the conv weights are fp16 numbers" : "3 products") + ", fp32 accumulate), "
             "16-board groups x row chunks x column tiles)";
        f = conv;
    } else if (group == 16 && mode == 5) {
        const bool two = two_products(n) && n->trunk_fmt != 1;
        nm = std::string(in_kind == 2 ? "k_trunk16hb" : "k_trunk16h") + (two ? "_w16<" : "<") + bw + ", 4, " + std::to_string(n->trunk_fmt) + "> (conv1 + " +
             std::to_string(2 * n->nb) + " 3x3 convs as split-fp16 MFMA 16x16x32 (" + (two ? "2 products: the conv weights are fp16 numbers" : "3 products") + ", fp32 accumulate) + heads, one resident "
             "launch; activations between layers as " + (n->trunk_fmt == 1 ? "fp16 high half + one low byte" : "two fp16 halves") + ")";
        f = conv1 + 2.0 * n->nb * conv;
    } else if (group == 16) {
        nm = "k_trunk16<" + bw + "> (conv1 + " + std::to_string(2 * n->nb) + " 3x3 convs, one launch, fp32 MFMA 16x16x4)";
        f = conv1 + 2.0 * n->nb * conv;
    } else {
        nm = "k_conv3x3<" + bw + "> (one 3x3 " + std::to_string(n->planes) + "->" + std::to_string(n->planes) +
             " conv, fp32 MFMA 32x32x2)";
        f = conv;
    }
    *nm_out = nm;
    *f_out = f;
}

static void copy_name(const std::string& nm, char* name, int name_cap) {
    if (name && name_cap > 0) {
        std::strncpy(name, nm.c_str(), static_cast<size_t>(name_cap) - 1);
        name[name_cap - 1] = 0;
    }
}

int ao_net_products(ao_net* n, int32_t request, int32_t* in_force, int32_t* weights_fp16) {
    if (request != -1 && request != 0 && request != 3) return n->fail("ao_net_products: request must be -1 (query), 0 (automatic) or 3");
    if (request != -1) n->products_req = request;
    if (in_force) *in_force = two_products(n) ? 2 : 3;
    if (weights_fp16) *weights_fp16 = n->w16 ? 1 : 0;
    return 0;
}

int ao_net_dominant_kernel(ao_net* n, int boards, char* name, int name_cap, double* flop_per_launch) {
    std::string nm;
    double f = 0.0;
    dominant_name(n, boards, n->last_in_kind, &nm, &f);
    copy_name(nm, name, name_cap);
    if (flop_per_launch) *flop_per_launch = f;
    return 0;
}

int ao_net_plan_kernel(int n_block, int inplanes, int planes, int board, int trunk_mode, int boards, int in_kind,
                       char* name, int name_cap, double* flop_per_launch) {
    if (n_block < 0 || inplanes < 1 || planes < 32 || planes % 32 || board < 3 || board > ao::kMaxBoard || boards < 1 ||
        trunk_mode < 0 || trunk_mode > 6)
        return 1;
    ao_net n;   // never finalized, owns nothing: the planning fields of ao_net_create without a device
    n.nb = n_block; n.C = inplanes; n.planes = planes; n.B = board; n.A = board * board;
    n.nchq32 = (((inplanes + 3) / 4) + 1) & ~1;
    n.nchq16 = (((inplanes + 3) / 4) + 7) & ~7;
    n.nchq1 = (((inplanes + 3) / 4) + 3) & ~3;
    n.CQ = planes / 4;
    n.mode = trunk_mode;
    if (const char* v = getenv("AO_TRUNK_FMT")) n.trunk_fmt = atoi(v) == 1 ? 1 : 0;
    if (n.trunk_fmt < 0) n.trunk_fmt = 0;   // as ao_net_create
    std::string nm;
    double f = 0.0;
    dominant_name(&n, boards, in_kind, &nm, &f);
    copy_name(nm, name, name_cap);
    if (flop_per_launch) *flop_per_launch = f;
    return 0;
}

}  // extern "C"
