// engine.hip -- host driver of the batched self-play engine and the C ABI of include/omok_hip.h.
//
// Per move decision (ZeroAgent.get_pi, agents.py:60-132) the host does exactly three things:
//   1. draws each game's Dirichlet noise from the game's own MT19937 stream (host_rng.hpp;
//      the state is read back from HBM, 2.5 KB per game per move, and written back),
//   2. queues the kernels: k_begin_move, then per simulation k_select -> evaluator ->
//      k_expand_backup, then k_end_move (and k_play for self-play),
//   3. copies the [G][A] result vectors back.
// Everything else lives in HBM and in the kernels of tree_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/omok_hip.h"
#include "engine_types.hpp"
#include "host_rng.hpp"

namespace ao {
void launch_select(const TreeParams& p, hipStream_t s);
void launch_expand_backup(const TreeParams& p, hipStream_t s);
void launch_expand_select(const TreeParams& p, hipStream_t s);
void launch_begin_move(const TreeParams& p, hipStream_t s);
void launch_order(const TreeParams& p, int32_t* order, hipStream_t s);
void launch_end_move(const TreeParams& p, hipStream_t s);
void launch_play(const TreeParams& p, hipStream_t s);
void launch_walk(const TreeParams& p, int count, const int32_t* games, const int32_t* extra, int stride, const int32_t* m,
                 const int32_t* prev_known, int32_t* status_out, hipStream_t s);
void launch_reset(const TreeParams& p, const uint8_t* mask, hipStream_t s);
// net.hip
int net_forward_il(ao_net* n, const float* in_il, int boards, float* policy, float* value,
                   hipStream_t s, int in_kind, int parts, const unsigned* live, unsigned row_cap);
void launch_eval_log(const int32_t* games, int n, const int32_t* row_of_game, const float* policy, const float* value, int A,
                     float* out, const int32_t* sims_done, const int32_t* leaf_status, int what, hipStream_t s);
int net_step_params(ao_net* n, int boards, float* policy, float* value, StepNet* out);
void net_fp16_fallback_begin(ao_net* n);
int net_fp16_fallback_end(ao_net* n);
// step_kernels.hip
void launch_step_board(const TreeParams& p, const StepNet& f, int rows, const int32_t* game_of_row, hipStream_t s);
void net_plan(const ao_net* n, int boards, int* group, int* nchq, int* kind);
int net_check(const ao_net* n, int board, int inplanes, int device, std::string* why);
}  // namespace ao

static thread_local std::string g_create_error;

struct ao_engine {
    ao_config cfg{};
    ao::TreeParams tp{};
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string err;
    int A = 0, Ap = 0, G = 0, Gp = 0, S = 0;
    std::vector<void*> allocs;
    // device scratch
    uint8_t* d_active = nullptr; int8_t* d_tau = nullptr; int32_t* d_extra = nullptr;
    uint8_t* d_mask = nullptr;
    std::vector<int32_t> h_walk;     // staging of ao_set_root(s): [G][A] moves + games, counts, prev_known, status
    float* d_policy = nullptr; float* d_value = nullptr;  // native-net outputs [Gp][A], [Gp]
    uint8_t* d_planes_u8 = nullptr;                       // bit planes [Gp][u8_row] (input of the split-fp16 kernels)
    int32_t* d_row = nullptr;                             // [2G] batch row of each game in ao_search (active games packed, or handed out per simulation by the tree kernel), then the game of each row
    std::vector<int32_t> h_row;
    // rows handed out per simulation (engine_types.hpp TreeParams::live): one counter word per launch of a move, zeroed at the start
    // of the move (and when the ring wraps); row_cap = rows one simulation may take (0: as many as there are active games)
    static constexpr int kLive = 2048;
    unsigned* d_live = nullptr;
    int32_t* d_order = nullptr;      // [G] launch order of k_expand_select's slots for over-subscribed searches (k_order, per move); AO_TREE_ORDER=0: off
    bool order_on = true;
    unsigned* h_live = nullptr;                           // pinned [kLive]
    unsigned* d_ctl = nullptr;                            // [2][4] the sit-out window of the current / the next launch (tree_device.hpp, sit_window)
    int row_cap = 0;
    double ask_frac = 1.0;                                // rows asked for per simulation of a game in the last over-subscribed move (1 - terminal share)
    int64_t rs_launches = 0, rs_rows_live = 0, rs_rows_launched = 0, rs_waits = 0;   // ao_row_stats
    // ao_set_eval_log: what the network returned to the listed games, per simulation of the current ao_search
    int32_t* d_log_games = nullptr; int log_n = 0; float* log_dev = nullptr; int64_t log_cap = 0; int64_t log_sim = 0;
    size_t il_bytes = 0; int il_group_zeroed = -1, il_nchq_zeroed = -1;  // layout for which batch_il's padding is zero
    // host mirrors
    std::vector<std::vector<int32_t>> moves;
    std::vector<int32_t> status;     // AO_ROOT_*
    std::vector<int32_t> over;       // win index once the game ended
    std::vector<uint8_t> active;
    std::vector<int32_t> has_gauss; std::vector<double> gauss;
    uint32_t* h_mt = nullptr; int32_t* h_pos = nullptr; double* h_noise = nullptr;  // pinned
    double* h_out = nullptr;         // pinned [3][G][A]
    int32_t* h_i32 = nullptr;        // pinned [4][G]
    int sims_left = 0;
    bool in_move = false, ended = false;
    // fp16-range recovery of ao_search: the games' MT19937 states as they were before the move (device copy), the host
    // half of the stream (legacy gauss cache), the number of recovered moves and of games searched again
    uint32_t* d_mt_backup = nullptr; int32_t* d_pos_backup = nullptr;
    std::vector<int32_t> has_gauss_backup; std::vector<double> gauss_backup;
    int64_t fp16_events = 0, fp16_games_redone = 0;
    int node_cap_auto = 0;           // 1: node_cap was derived from the free HBM (ao_config.node_cap == -1)
    // HIP-event timing of the per-simulation tree kernel (k_expand_select) on the launch stream
    bool timing = false;
    int timing_stride = 1;      // ao_tree_timing(enable = n > 1): every n-th launch is timed
    unsigned timing_tick = 0;
    static constexpr int kRing = 256;
    std::vector<hipEvent_t> ev0, ev1;
    int ring_head = 0, ring_count = 0;
    double ms_total = 0.0;
    int64_t launches = 0;

    int fail(const std::string& m) { err = m; return 1; }
};

#define AO_HIP(e, call)                                                                       \
    do {                                                                                      \
        hipError_t st_ = (call);                                                              \
        if (st_ != hipSuccess)                                                                \
            return (e)->fail(std::string(#call) + ": " + hipGetErrorString(st_));             \
    } while (0)

template <typename T>
static int dev_alloc(ao_engine* e, T** out, size_t count) {
    void* p = nullptr;
    hipError_t st = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16));
    if (st != hipSuccess)
        return e->fail(std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + " B): " +
                       hipGetErrorString(st));
    e->allocs.push_back(p);
    *out = static_cast<T*>(p);
    return 0;
}

static void tree_harvest(ao_engine* e, int count) {
    for (int i = 0; i < count; ++i) {
        const int idx = (e->ring_head - e->ring_count + ao_engine::kRing * 2) % ao_engine::kRing;
        (void)hipEventSynchronize(e->ev1[idx]);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->ev0[idx], e->ev1[idx]) == hipSuccess) {
            e->ms_total += ms;
            e->launches += 1;
        }
        --e->ring_count;
    }
}

extern "C" {

const char* ao_version(void) { return "alpha_omok_amd 0.1 (gfx950)"; }
int ao_abi_version(void) { return AO_ABI_VERSION; }

const char* ao_last_error(const ao_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }
void* ao_stream(ao_engine* e) { return e ? e->stream : nullptr; }

int ao_sync(ao_engine* e) {
    AO_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int ao_set_stream(ao_engine* e, void* stream) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    e->stream = stream ? static_cast<hipStream_t>(stream) : e->own_stream;
    return 0;
}

void ao_destroy(ao_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->stream) hipStreamSynchronize(e->stream);
    for (void* p : e->allocs) hipFree(p);
    if (e->h_mt) hipHostFree(e->h_mt);
    if (e->h_pos) hipHostFree(e->h_pos);
    if (e->h_noise) hipHostFree(e->h_noise);
    if (e->h_out) hipHostFree(e->h_out);
    if (e->h_i32) hipHostFree(e->h_i32);
    if (e->h_live) hipHostFree(e->h_live);
    for (auto ev : e->ev0) (void)hipEventDestroy(ev);
    for (auto ev : e->ev1) (void)hipEventDestroy(ev);
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    delete e;
}

static int create_impl(ao_engine* e, const ao_config* cfg) {
    e->cfg = *cfg;
    ao_config& c = e->cfg;
    if (c.board < 3 || c.board > ao::kMaxBoard) return e->fail("board must be in 3..15");
    if (c.win_mark <= 0) c.win_mark = (c.board == 3) ? 3 : 5;
    // (the five-in-a-row test looks four cells each way from the new stone: marks up to 5. A mark ABOVE the board size is also
    // accepted -- no line can win, the full board is the only end: what ZeroAgent.win_mark = 10 does in the reference, and how the
    // parity suite gets descents that run the whole board)
    if (c.win_mark > 5 && c.win_mark <= c.board) return e->fail("win_mark must be <= 5 (or above the board size: no line wins)");
    if (c.sims < 1) return e->fail("sims must be >= 1");
    if (c.inplanes < 3 || c.inplanes > 9 || (c.inplanes % 2) == 0) return e->fail("inplanes must be 3, 5, 7 or 9");
    if (c.games < 1) return e->fail("games must be >= 1");
    if (c.c_puct == 0.0) c.c_puct = 5.0;
    if (c.alpha == 0.0) c.alpha = 10.0 / static_cast<double>(c.board * c.board);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return e->fail("no HIP device available");
    if (c.device < 0 || c.device >= ndev) return e->fail("device ordinal out of range");
    AO_HIP(e, hipSetDevice(c.device));
    if (c.arena_fraction < 0.0 || c.arena_fraction > 0.90) return e->fail("arena_fraction must be in (0, 0.90] (0 = the default 0.40)");
    const double arena_fraction = c.arena_fraction > 0.0 ? c.arena_fraction : 0.40;
    if (c.node_cap == 0) {
        // Default arena: 16 x (sims + 1) expanded nodes per game where the part is large enough, never less than
        // 4 x (sims + 1). Re-rooting keeps the chosen child's subtree, so with a share f of the root's visits in that
        // child the kept tree settles at f / (1 - f) x sims nodes: 4 x holds f <= 0.75 (a random-init network stays far
        // below), 16 x holds f <= 0.93 -- what a trained, sharp policy needs, and the reference's dict never forgets
        // (agents.py:52). Measured with a network trained by this engine (profiles/r4_trained_*): 10.5 M move decisions at
        // 16 x without a single trim; one self_play(4096) at 11.4 x: 38 of 259 k re-rootings trimmed. The number is
        // DETERMINISTIC for a given device model: bounded by 40 % of the device's TOTAL memory for the two arenas (4096
        // games x 400 sims on a 288 GB MI355X: ~5900 nodes = 14.7 x sims, 124 GB -- trees are what the HBM is for, and two
        // such engines still fit side by side), not by what
        // happens to be free when the engine is created -- whether re-rooting has to forget subtrees (ao_trim_stats) must
        // not depend on the GPU's other tenants.
        const int Ap_ = (c.board * c.board + 15) / 16 * 16;
        const double node_bytes = ao::node_rec_bytes(Ap_);   // the node record: P 8 B + N, Q, CH, W 4 B + ACT 1 B per edge slot + the position, padded to 128
        size_t free_b = 0, total_b = 0;
        long cap = 16L * (c.sims + 1);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            cap = std::min<long>(cap, static_cast<long>(arena_fraction * static_cast<double>(total_b) / (2.0 * c.games * node_bytes)));
        cap = std::max<long>(cap, 4L * (c.sims + 1));
        c.node_cap = static_cast<int32_t>(std::min<long>(cap, 15000));
    } else if (c.node_cap < 0) {
        // node_cap = -1, opt-in: grow into the HBM that is free right now -- up to a quarter of it, at most
        // 16 x (sims + 1) -- for sharp (trained) policies that keep more of the tree from move to move. 4096 games x
        // 400 sims on a 288 GB part: ~3400 nodes per arena instead of 1604. The chosen value is reported by ao_node_cap.
        const int Ap_ = (c.board * c.board + 15) / 16 * 16;
        const double node_bytes = ao::node_rec_bytes(Ap_);   // the node record: P 8 B + N, Q, CH, W 4 B + ACT 1 B per edge slot + the position, padded to 128
        size_t free_b = 0, total_b = 0;
        long cap = 4L * (c.sims + 1);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double room = 0.25 * static_cast<double>(free_b) / (2.0 * c.games * node_bytes);
            cap = std::max<long>(cap, std::min<long>(static_cast<long>(room), 16L * (c.sims + 1)));
        }
        c.node_cap = static_cast<int32_t>(std::min<long>(cap, 15000));
        e->node_cap_auto = 1;
    }
    if (c.node_cap < c.sims + 2) return e->fail("node_cap must be at least sims + 2");
    if (c.node_cap > 15000) return e->fail("node_cap must be <= 15000");
    AO_HIP(e, hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;

    const int A = c.board * c.board, Ap = (A + 15) / 16 * 16, G = c.games;
    const int Gp = (G + ao::kGroup - 1) / ao::kGroup * ao::kGroup;
    e->A = A; e->Ap = Ap; e->G = G; e->Gp = Gp; e->S = c.sims;
    ao::TreeParams& p = e->tp;
    p.B = c.board; p.A = A; p.Ap = Ap; p.C = c.inplanes; p.win_mark = c.win_mark; p.G = G;
    p.cap = c.node_cap; p.maxd = A + 2; p.noise = c.noise ? 1 : 0;
    p.keep_max = c.node_cap - c.sims - 1;
    p.compact_always = getenv("AO_COMPACT_ALWAYS") != nullptr;   // developer switch: re-root by copying after every move (rounds 1 - 5)
    p.nchq = (((c.inplanes + 3) / 4) + 7) & ~7;  // worst case of the network's input layouts (net_plan)
    p.nchq_live = p.nchq;
    p.il_group = ao::kGroup;
    p.c_puct = c.c_puct;

    p.rec = ao::node_rec_bytes(Ap);                        // one interleaved record per node (engine_types.hpp)
    // The default arena (ao_config.node_cap == 0) is sized from the device's TOTAL memory: other tenants -- a second engine, torch's
    // training state, a device replay -- may leave less. It then shrinks (x 3/4 per attempt, never below 4 x (sims + 1)) until the
    // allocation succeeds; ao_node_cap reports what was chosen. An explicit node_cap fails as it always did.
    for (;;) {
        const size_t slots = static_cast<size_t>(2) * G * p.cap;
        void* arena = nullptr;
        const hipError_t st = hipMalloc(&arena, std::max<size_t>(slots * p.rec, 16));
        if (st == hipSuccess) {
            e->allocs.push_back(arena);
            p.arena = static_cast<unsigned char*>(arena);
            break;
        }
        (void)hipGetLastError();
        const int floor_cap = 4 * (c.sims + 1);
        if (cfg->node_cap != 0 || p.cap <= floor_cap)
            return e->fail(std::string("hipMalloc(") + std::to_string(slots * p.rec) + " B for the tree arenas): " + hipGetErrorString(st));
        p.cap = std::max(floor_cap, p.cap / 4 * 3);
        c.node_cap = p.cap;
        p.keep_max = c.node_cap - c.sims - 1;
    }
    if (dev_alloc(e, &p.cur, G) || dev_alloc(e, &p.root_node, G) || dev_alloc(e, &p.nodes_used, G) ||
        dev_alloc(e, &p.rootpos, G) || dev_alloc(e, &p.mt, static_cast<size_t>(G) * 624) ||
        dev_alloc(e, &p.mtpos, G) || dev_alloc(e, &p.noise_buf, static_cast<size_t>(G) * Ap) ||
        dev_alloc(e, &p.sims_target, G) || dev_alloc(e, &p.sims_done, G) || dev_alloc(e, &p.gflags, G) ||
        dev_alloc(e, &p.rstatus, G) || dev_alloc(e, &p.pending_root, G) || dev_alloc(e, &p.leaf_status, G) || dev_alloc(e, &p.path_len, G) ||
        dev_alloc(e, &p.path_node, static_cast<size_t>(G) * p.maxd) ||
        dev_alloc(e, &p.path_edge, static_cast<size_t>(G) * p.maxd) || dev_alloc(e, &p.leaf_pos, G) ||
        dev_alloc(e, &p.err, G) || dev_alloc(e, &p.trimmed, static_cast<size_t>(G) * 2) || dev_alloc(e, &p.stats, static_cast<size_t>(G) * 4) ||
        dev_alloc(e, &p.out_pi, static_cast<size_t>(G) * A) || dev_alloc(e, &p.out_visit, static_cast<size_t>(G) * A) ||
        dev_alloc(e, &p.out_policy, static_cast<size_t>(G) * A) || dev_alloc(e, &p.action, G) ||
        dev_alloc(e, &p.win, G) || dev_alloc(e, &e->d_active, G) || dev_alloc(e, &e->d_tau, G) ||
        dev_alloc(e, &e->d_extra, static_cast<size_t>(G) * A + 4 * static_cast<size_t>(G)) || dev_alloc(e, &e->d_mask, G))
        return 1;
    // evaluation batches of the native network: interleaved input, policy/value rows for Gp boards
    float* il = nullptr;
    if (dev_alloc(e, &il, static_cast<size_t>(Gp) * A * p.nchq * 4) ||
        dev_alloc(e, &e->d_policy, static_cast<size_t>(Gp) * A) || dev_alloc(e, &e->d_value, Gp))
        return 1;
    p.u8_row = A <= 128 ? 128 : 256;
    if (dev_alloc(e, &e->d_planes_u8, static_cast<size_t>(Gp) * p.u8_row) || dev_alloc(e, &e->d_row, 2 * static_cast<size_t>(G))) return 1;
    if (dev_alloc(e, &e->d_mt_backup, static_cast<size_t>(G) * 624) || dev_alloc(e, &e->d_pos_backup, G)) return 1;
    if (dev_alloc(e, &e->d_live, ao_engine::kLive) || dev_alloc(e, &e->d_log_games, G) || dev_alloc(e, &e->d_ctl, 8) || dev_alloc(e, &e->d_order, G)) return 1;
    e->order_on = !(getenv("AO_TREE_ORDER") && atoi(getenv("AO_TREE_ORDER")) == 0);
    p.order = nullptr;
    AO_HIP(e, hipMemsetAsync(e->d_row, 0, sizeof(int32_t) * 2 * G, e->stream));
    AO_HIP(e, hipMemsetAsync(e->d_live, 0, sizeof(unsigned) * ao_engine::kLive, e->stream));
    p.row_of_game = nullptr;
    p.live = nullptr;
    p.row_cap = 0;
    p.ctl = nullptr; p.ctl_cur = 0; p.live_prev = nullptr; p.row_target = 0; p.max_levels = 0;
    AO_HIP(e, hipMemsetAsync(e->d_planes_u8, 0, static_cast<size_t>(Gp) * p.u8_row, e->stream));
    p.batch_u8 = nullptr;
    e->il_bytes = static_cast<size_t>(Gp) * A * p.nchq * 4 * sizeof(float);
    AO_HIP(e, hipMemsetAsync(il, 0, e->il_bytes, e->stream));
    p.batch_il = il;
    p.tau = e->d_tau;
    p.active = e->d_active;

    // np.sqrt(total_n) for every reachable integer total (correctly rounded by IEEE sqrt)
    const int lut_n = c.sims * (A + 2) + 8;
    std::vector<double> lut(lut_n);
    for (int i = 0; i < lut_n; ++i) lut[i] = std::sqrt(static_cast<double>(i));
    double* d_lut = nullptr;
    if (dev_alloc(e, &d_lut, lut_n)) return 1;
    AO_HIP(e, hipMemcpy(d_lut, lut.data(), sizeof(double) * lut_n, hipMemcpyHostToDevice));
    p.sqrt_lut = d_lut; p.sqrt_lut_n = lut_n;

    AO_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_mt), sizeof(uint32_t) * 624 * G, hipHostMallocDefault));
    AO_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_pos), sizeof(int32_t) * G, hipHostMallocDefault));
    AO_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_noise), sizeof(double) * G * Ap, hipHostMallocDefault));
    AO_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_out), sizeof(double) * 3 * G * A, hipHostMallocDefault));
    AO_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_i32), sizeof(int32_t) * 4 * G, hipHostMallocDefault));
    AO_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_live), sizeof(unsigned) * (ao_engine::kLive + 8), hipHostMallocDefault));
    std::memset(e->h_noise, 0, sizeof(double) * G * Ap);

    e->moves.assign(G, {});
    e->status.assign(G, AO_ROOT_FRESH);
    e->over.assign(G, 0);
    e->active.assign(G, 1);
    e->has_gauss.assign(G, 0);
    e->gauss.assign(G, 0.0);
    for (int g = 0; g < G; ++g) {
        ao::HostMT::seed(e->h_mt + static_cast<size_t>(g) * 624, static_cast<uint32_t>(g));
        e->h_pos[g] = 624;
    }
    AO_HIP(e, hipMemcpyAsync(p.mt, e->h_mt, sizeof(uint32_t) * 624 * G, hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipMemcpyAsync(p.mtpos, e->h_pos, sizeof(int32_t) * G, hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipMemsetAsync(p.stats, 0, sizeof(unsigned) * 4 * G, e->stream));
    AO_HIP(e, hipMemsetAsync(p.trimmed, 0, sizeof(int32_t) * 2 * G, e->stream));
    AO_HIP(e, hipMemsetAsync(p.noise_buf, 0, sizeof(double) * G * Ap, e->stream));
    ao::launch_reset(p, nullptr, e->stream);
    AO_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int ao_create(const ao_config* cfg, ao_engine** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return 1; }
    ao_engine* e = new ao_engine();
    if (create_impl(e, cfg)) {
        g_create_error = e->err;
        ao_destroy(e);
        *out = nullptr;
        return 1;
    }
    *out = e;
    return 0;
}

// ---- RNG -------------------------------------------------------------------------------------
int ao_set_rng_state(ao_engine* e, int g, const uint32_t* mt, int32_t pos, int32_t has_gauss, double gauss) {
    if (g < 0 || g >= e->G) return e->fail("game index out of range");
    if (e->in_move) return e->fail("ao_set_rng_state / ao_seed inside a move (the move's Dirichlet draw has already consumed the old stream)");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    AO_HIP(e, hipMemcpyAsync(e->tp.mt + static_cast<size_t>(g) * 624, mt, sizeof(uint32_t) * 624,
                             hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipMemcpyAsync(e->tp.mtpos + g, &pos, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    e->has_gauss[g] = has_gauss;
    e->gauss[g] = gauss;
    return 0;
}

int ao_get_rng_state(ao_engine* e, int g, uint32_t* mt, int32_t* pos, int32_t* has_gauss, double* gauss) {
    if (g < 0 || g >= e->G) return e->fail("game index out of range");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    AO_HIP(e, hipMemcpyAsync(mt, e->tp.mt + static_cast<size_t>(g) * 624, sizeof(uint32_t) * 624,
                             hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipMemcpyAsync(pos, e->tp.mtpos + g, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    if (has_gauss) *has_gauss = e->has_gauss[g];
    if (gauss) *gauss = e->gauss[g];
    return 0;
}

int ao_seed(ao_engine* e, int g, uint32_t seed) {
    std::vector<uint32_t> mt(624);
    ao::HostMT::seed(mt.data(), seed);
    return ao_set_rng_state(e, g, mt.data(), 624, 0, 0.0);
}

int ao_seed_games(ao_engine* e, const int32_t* games, const uint32_t* seeds, int32_t n) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    if (n <= 0) return 0;
    if (e->in_move) return e->fail("ao_seed_games inside a move");
    for (int k = 0; k < n; ++k)
        if (games[k] < 0 || games[k] >= e->G) return e->fail("ao_seed_games: game index out of range");
    AO_HIP(e, hipStreamSynchronize(e->stream));   // (the pinned staging rows are free)
    for (int k = 0; k < n; ++k) {
        const int g = games[k];
        ao::HostMT::seed(e->h_mt + static_cast<size_t>(g) * 624, seeds[k]);
        e->h_pos[g] = 624;
        e->has_gauss[g] = 0;
        e->gauss[g] = 0.0;
        AO_HIP(e, hipMemcpyAsync(e->tp.mt + static_cast<size_t>(g) * 624, e->h_mt + static_cast<size_t>(g) * 624, sizeof(uint32_t) * 624,
                                 hipMemcpyHostToDevice, e->stream));
        AO_HIP(e, hipMemcpyAsync(e->tp.mtpos + g, e->h_pos + g, sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    }
    AO_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

int ao_seed_all(ao_engine* e, const uint32_t* seeds) {
    if (e->in_move) return e->fail("ao_seed_all inside a move");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    for (int g = 0; g < e->G; ++g) {
        ao::HostMT::seed(e->h_mt + static_cast<size_t>(g) * 624, seeds[g]);
        e->h_pos[g] = 624;
        e->has_gauss[g] = 0;
        e->gauss[g] = 0.0;
    }
    AO_HIP(e, hipMemcpyAsync(e->tp.mt, e->h_mt, sizeof(uint32_t) * 624 * e->G, hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipMemcpyAsync(e->tp.mtpos, e->h_pos, sizeof(int32_t) * e->G, hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    return 0;
}

// ---- game / tree state -----------------------------------------------------------------------
int ao_reset(ao_engine* e, const uint8_t* mask) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    const uint8_t* dmask = nullptr;
    if (mask) {
        AO_HIP(e, hipMemcpyAsync(e->d_mask, mask, e->G, hipMemcpyHostToDevice, e->stream));
        dmask = e->d_mask;
    }
    ao::launch_reset(e->tp, dmask, e->stream);
    AO_HIP(e, hipStreamSynchronize(e->stream));
    for (int g = 0; g < e->G; ++g) {
        if (mask && !mask[g]) continue;
        e->moves[g].clear();
        e->status[g] = AO_ROOT_FRESH;
        e->over[g] = 0;
    }
    e->in_move = false;
    return 0;
}

// Walks the listed games along their new ids in ONE launch (k_walk: a workgroup per listed game).
// ids[k] / ns[k]: full move list of games[k]. Games whose new id does not extend the kept one are reset first.
static int set_roots_impl(ao_engine* e, int count, const int32_t* games, const int32_t* const* ids, const int32_t* ns,
                          int32_t* status) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    if (count <= 0) return 0;
    const int G = e->G, A = e->A;
    std::vector<uint8_t> seen(G, 0);
    for (int k = 0; k < count; ++k) {
        if (games[k] < 0 || games[k] >= G) return e->fail("game index out of range");
        if (seen[games[k]]) return e->fail("ao_set_roots: a game is listed twice");
        seen[games[k]] = 1;
        if (ns[k] < 0 || ns[k] > A) return e->fail("move list too long");
    }
    // staging: [count][A] new moves, then games / counts / prev_known (device reads), then status (device writes)
    e->h_walk.assign(static_cast<size_t>(count) * A + 4 * static_cast<size_t>(count), 0);
    int32_t* h_extra = e->h_walk.data();
    int32_t* h_games = h_extra + static_cast<size_t>(count) * A;
    int32_t* h_m = h_games + count;
    int32_t* h_pk = h_m + count;
    std::vector<uint8_t> rmask;
    for (int k = 0; k < count; ++k) {
        const int g = games[k];
        const std::vector<int32_t>& cur = e->moves[g];
        const int n = ns[k];
        const bool extends = static_cast<size_t>(n) >= cur.size() && std::equal(cur.begin(), cur.end(), ids[k]);
        int first = 0;
        int prev_known = (e->status[g] != AO_ROOT_FRESH) ? 1 : 0;
        if (!extends) {
            if (rmask.empty()) rmask.assign(G, 0);
            rmask[g] = 1;
            prev_known = 0;
        } else {
            first = static_cast<int>(cur.size());
        }
        h_games[k] = g;
        h_m[k] = n - first;
        h_pk[k] = prev_known;
        std::copy(ids[k] + first, ids[k] + n, h_extra + static_cast<size_t>(k) * A);
    }
    if (!rmask.empty() && ao_reset(e, rmask.data())) return 1;
    int32_t* d_games = e->d_extra + static_cast<size_t>(count) * A;
    AO_HIP(e, hipMemcpyAsync(e->d_extra, h_extra, sizeof(int32_t) * (static_cast<size_t>(count) * A + 3 * static_cast<size_t>(count)),
                             hipMemcpyHostToDevice, e->stream));
    ao::launch_walk(e->tp, count, d_games, e->d_extra, A, d_games + count, d_games + 2 * count, d_games + 3 * count, e->stream);
    AO_HIP(e, hipGetLastError());
    int32_t* h_st = h_pk + count;
    AO_HIP(e, hipMemcpyAsync(h_st, d_games + 3 * count, sizeof(int32_t) * count, hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    bool bad = false;
    std::vector<uint8_t> bmask;
    for (int k = 0; k < count; ++k) {
        const int g = games[k];
        if (h_st[k] < 0) {
            if (bmask.empty()) bmask.assign(G, 0);
            bmask[g] = 1;
            bad = true;
            if (status) status[k] = -1;
            continue;
        }
        e->moves[g].assign(ids[k], ids[k] + ns[k]);
        e->status[g] = h_st[k];
        e->over[g] = 0;
        if (status) status[k] = h_st[k];
    }
    if (bad) {
        ao_reset(e, bmask.data());
        return e->fail("ao_set_root: illegal move in the id (occupied cell or out of range)");
    }
    return 0;
}

int ao_set_root(ao_engine* e, int g, const int32_t* mv, int32_t n, int32_t* status) {
    return set_roots_impl(e, 1, &g, &mv, &n, status);
}

int ao_set_roots(ao_engine* e, const uint8_t* mask, const int32_t* moves, int32_t stride, const int32_t* n, int32_t* status) {
    if (!moves || !n) return e->fail("ao_set_roots: null argument");
    std::vector<int32_t> games, ns;
    std::vector<const int32_t*> ids;
    for (int g = 0; g < e->G; ++g) {
        if (mask && !mask[g]) continue;
        games.push_back(g);
        ns.push_back(n[g]);
        ids.push_back(moves + static_cast<size_t>(g) * stride);
        if (n[g] > stride) return e->fail("ao_set_roots: n[g] exceeds the row stride");
    }
    std::vector<int32_t> st(games.size(), 0);
    const int rc = set_roots_impl(e, static_cast<int>(games.size()), games.data(), ids.data(), ns.data(), st.data());
    if (status)
        for (size_t k = 0; k < games.size(); ++k) status[games[k]] = st[k];
    return rc;
}

int ao_get_moves(ao_engine* e, int g, int32_t* out, int32_t* n) {
    if (g < 0 || g >= e->G) return e->fail("game index out of range");
    std::copy(e->moves[g].begin(), e->moves[g].end(), out);
    *n = static_cast<int32_t>(e->moves[g].size());
    return 0;
}

// ---- one move decision -----------------------------------------------------------------------
int ao_begin_move(ao_engine* e, const uint8_t* active) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    const int G = e->G, A = e->A, Ap = e->Ap;
    ao::TreeParams& p = e->tp;
    int32_t* target = e->h_i32;
    int32_t* flags = e->h_i32 + G;
    int maxt = 0;
    for (int g = 0; g < G; ++g) {
        e->active[g] = (active ? active[g] : 1) && !e->over[g];
        target[g] = 0;
        flags[g] = 0;
        if (!e->active[g]) continue;
        target[g] = (e->status[g] == AO_ROOT_FRESH) ? e->S + 1 : e->S;  // agents.py:107-111
        flags[g] = (p.noise && e->status[g] == AO_ROOT_EXPANDED) ? 1 : 0;
        maxt = std::max(maxt, target[g]);
    }
    if (p.noise) {
        // The Dirichlet draw is the first consumer of the stream in a move in every case:
        // re-noise of an inherited root (agents.py:97) or the root expansion of sim 0 (:194).
        AO_HIP(e, hipMemcpyAsync(e->h_mt, p.mt, sizeof(uint32_t) * 624 * G, hipMemcpyDeviceToHost, e->stream));
        AO_HIP(e, hipMemcpyAsync(e->h_pos, p.mtpos, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
        AO_HIP(e, hipStreamSynchronize(e->stream));
        const double alpha = e->cfg.alpha;
        auto work = [&](int g0, int g1) {
            for (int g = g0; g < g1; ++g) {
                if (!e->active[g]) continue;
                ao::HostMT r{e->h_mt + static_cast<size_t>(g) * 624, e->h_pos + g, &e->has_gauss[g], &e->gauss[g]};
                const int k = A - static_cast<int>(e->moves[g].size());
                r.dirichlet(alpha, k, e->h_noise + static_cast<size_t>(g) * Ap);
            }
        };
        // the process's persistent host pool (host_rng.hpp: hardware threads / LOCAL_WORLD_SIZE, at most 32); games in
        // chunks of 16 so that the ranks of a node do not oversubscribe the host at the start of every move
        if (G < 64) work(0, G);
        else ao::HostPool::get().run(G, 16, work);
        AO_HIP(e, hipMemcpyAsync(p.mt, e->h_mt, sizeof(uint32_t) * 624 * G, hipMemcpyHostToDevice, e->stream));
        AO_HIP(e, hipMemcpyAsync(p.mtpos, e->h_pos, sizeof(int32_t) * G, hipMemcpyHostToDevice, e->stream));
        AO_HIP(e, hipMemcpyAsync(p.noise_buf, e->h_noise, sizeof(double) * G * Ap, hipMemcpyHostToDevice, e->stream));
    }
    AO_HIP(e, hipMemcpyAsync(p.sims_target, target, sizeof(int32_t) * G, hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipMemcpyAsync(p.gflags, flags, sizeof(int32_t) * G, hipMemcpyHostToDevice, e->stream));
    AO_HIP(e, hipMemcpyAsync(e->d_active, e->active.data(), G, hipMemcpyHostToDevice, e->stream));
    if (e->order_on && e->row_cap > 0 && e->row_cap < G) ao::launch_order(p, e->d_order, e->stream);   // (reads the move's stats that the next line clears)
    AO_HIP(e, hipMemsetAsync(p.stats, 0, sizeof(unsigned) * 4 * G, e->stream));
    ao::launch_begin_move(p, e->stream);
    // the pinned staging buffers are reused by the next call: make sure the copies are done
    AO_HIP(e, hipStreamSynchronize(e->stream));
    e->sims_left = maxt;
    e->in_move = true;
    e->ended = false;
    return 0;
}

int ao_sims_left(ao_engine* e) { return e->sims_left; }

int ao_collect_leaves(ao_engine* e, float* dev_planes_nchw) {
    if (!e->in_move) return e->fail("ao_collect_leaves outside ao_begin_move/ao_end_move");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    ao::TreeParams p = e->tp;
    p.batch_nchw = dev_planes_nchw;
    if (dev_planes_nchw) p.batch_il = nullptr;   // an external evaluator only needs the NCHW planes
    else e->il_group_zeroed = -1;                // full-width write in whatever layout was planned last
    ao::launch_select(p, e->stream);
    AO_HIP(e, hipGetLastError());
    return 0;
}

int ao_apply_evals(ao_engine* e, const float* dev_policy, const float* dev_value) {
    if (!e->in_move) return e->fail("ao_apply_evals outside ao_begin_move/ao_end_move");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    ao::TreeParams p = e->tp;
    p.policy = dev_policy;
    p.value = dev_value;
    ao::launch_expand_backup(p, e->stream);
    AO_HIP(e, hipGetLastError());
    if (e->sims_left > 0) --e->sims_left;
    return 0;
}

static int check_game_errors(ao_engine* e) {
    int32_t* herr = e->h_i32 + 2 * e->G;
    AO_HIP(e, hipMemcpyAsync(herr, e->tp.err, sizeof(int32_t) * e->G, hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    for (int g = 0; g < e->G; ++g) {
        if (!herr[g]) continue;
        std::string m = "game " + std::to_string(g) + ":";
        if (herr[g] & ao::ERR_NODE_CAP) m += " tree arena full (raise ao_config.node_cap)";
        if (herr[g] & ao::ERR_PATH) m += " no selectable child (priors are NaN: the policy summed to 0 over the legal moves -- the reference's prior /= prior.sum(), agents.py:189 -- or the tree is inconsistent)";
        if (herr[g] & ao::ERR_BAD_MOVE) m += " move onto an occupied cell";
        if (herr[g] & ao::ERR_SHORT) {
            // the reference always runs num_mcts simulations (agents.py:105-132): a pi from fewer visits is never handed out
            int32_t dt[2] = {0, 0};
            (void)hipMemcpy(&dt[0], e->tp.sims_done + g, sizeof(int32_t), hipMemcpyDeviceToHost);
            (void)hipMemcpy(&dt[1], e->tp.sims_target + g, sizeof(int32_t), hipMemcpyDeviceToHost);
            int nshort = 0, nact = 0;
            for (int k = 0; k < e->G; ++k) { nshort += (herr[k] & ao::ERR_SHORT) ? 1 : 0; nact += e->active[k] ? 1 : 0; }
            m += " search ended after " + std::to_string(dt[0]) + " of " + std::to_string(dt[1]) + " simulations (" + std::to_string(nshort) + " of " +
                 std::to_string(nact) + " active games are short; rows per simulation: " +
                 (e->row_cap > 0 ? std::to_string(e->row_cap) : std::string("one per game")) + ")";
        }
        // leave the engine usable: the error words are cleared and no move is in flight (the caller resets the game)
        (void)hipMemsetAsync(e->tp.err, 0, sizeof(int32_t) * e->G, e->stream);
        (void)hipStreamSynchronize(e->stream);
        e->in_move = false;
        e->ended = false;
        return e->fail(m);
    }
    return 0;
}

int ao_end_move(ao_engine* e, const int8_t* tau, double* pi, double* visit, double* policy) {
    if (!e->in_move) return e->fail("ao_end_move without ao_begin_move");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    const int G = e->G, A = e->A;
    ao::TreeParams p = e->tp;
    if (tau) {
        AO_HIP(e, hipMemcpyAsync(e->d_tau, tau, G, hipMemcpyHostToDevice, e->stream));
        p.tau = e->d_tau;
    } else {
        p.tau = nullptr;
    }
    ao::launch_end_move(p, e->stream);
    const size_t n = static_cast<size_t>(G) * A;
    if (pi) AO_HIP(e, hipMemcpyAsync(e->h_out, p.out_pi, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
    if (visit) AO_HIP(e, hipMemcpyAsync(e->h_out + n, p.out_visit, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
    if (policy) AO_HIP(e, hipMemcpyAsync(e->h_out + 2 * n, p.out_policy, sizeof(double) * n, hipMemcpyDeviceToHost, e->stream));
    if (check_game_errors(e)) return 1;  // synchronises the stream
    if (pi) std::memcpy(pi, e->h_out, sizeof(double) * n);
    if (visit) std::memcpy(visit, e->h_out + n, sizeof(double) * n);
    if (policy) std::memcpy(policy, e->h_out + 2 * n, sizeof(double) * n);
    // after a search every searched root is expanded (it was expanded by sim 0 at the latest)
    for (int g = 0; g < G; ++g)
        if (e->active[g]) e->status[g] = AO_ROOT_EXPANDED;
    e->in_move = false;
    e->ended = true;
    return 0;
}

int ao_play(ao_engine* e, int32_t* action, int32_t* win) {
    if (!e->ended) return e->fail("ao_play must follow ao_end_move");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    const int G = e->G;
    ao::launch_play(e->tp, e->stream);
    int32_t* ha = e->h_i32;
    int32_t* hw = e->h_i32 + G;
    int32_t* hs = e->h_i32 + 3 * G;
    AO_HIP(e, hipMemcpyAsync(ha, e->tp.action, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipMemcpyAsync(hw, e->tp.win, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipMemcpyAsync(hs, e->tp.rstatus, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
    if (check_game_errors(e)) return 1;
    for (int g = 0; g < G; ++g) {
        if (!e->active[g]) {
            if (action) action[g] = -1;
            if (win) win[g] = e->over[g];
            continue;
        }
        e->moves[g].push_back(ha[g]);
        e->status[g] = hs[g];
        e->over[g] = hw[g];
        if (action) action[g] = ha[g];
        if (win) win[g] = hw[g];
    }
    e->ended = false;
    return 0;
}

static int search_impl(ao_engine* e, ao_net* net, const uint8_t* active, const int8_t* tau, double* pi,
                       double* visit, double* policy, bool may_recover) {
    if (!net) return e->fail("ao_search: null network");
    std::string why;
    if (ao::net_check(net, e->cfg.board, e->cfg.inplanes, e->cfg.device, &why)) return e->fail("ao_search: " + why);
    if (may_recover) {
        // what the fp16-range recovery below needs to run this move again: the streams as they are NOW
        AO_HIP(e, hipSetDevice(e->cfg.device));
        AO_HIP(e, hipMemcpyAsync(e->d_mt_backup, e->tp.mt, sizeof(uint32_t) * 624 * e->G, hipMemcpyDeviceToDevice, e->stream));
        AO_HIP(e, hipMemcpyAsync(e->d_pos_backup, e->tp.mtpos, sizeof(int32_t) * e->G, hipMemcpyDeviceToDevice, e->stream));
        e->has_gauss_backup = e->has_gauss;
        e->gauss_backup = e->gauss;
    }
    // the network announces the interleaved input layout it wants for a batch of G boards
    if (ao_begin_move(e, active)) return 1;
    // The network runs on the ACTIVE games only. Two regimes (engine_types.hpp, TreeParams::live):
    //  * a handful of games on the per-board path with the fused per-game step: the host packs the active games to the front of
    //    the evaluation batch once per move (row_of_game / game_of_row);
    //  * everything else (round 5): the tree kernel hands out the rows PER SIMULATION -- a game whose new leaf needs the network takes
    //    the next row, a terminal leaf takes none (agents.py:171-178,216-221: the reference evaluates it and throws the result
    //    away) -- and counts them in one device word per launch that the trunk kernels read: groups without a live row exit at
    //    once, no host read-back in the loop. With a trained network 11 - 19 % of all leaves are terminal. ao_set_row_cap bounds
    //    the rows of one simulation BELOW the number of games (over-subscription: 5120 games on the 4096 rows = 256 groups the
    //    resident trunk fills the chip with); a leaf that finds the batch full waits for the next launch (LS_WAIT), the loop below
    //    runs until every game has its simulations. A game's search is strictly sequential in every regime: same bits.
    int rows = 0;
    e->h_row.assign(2 * static_cast<size_t>(e->G), 0);
    for (int g = 0; g < e->G; ++g)
        if (e->active[g]) {
            e->h_row[e->G + rows] = g;
            e->h_row[g] = rows++;
        }
    if (rows == 0) return ao_end_move(e, tau, pi, visit, policy);
    static const bool static_rows = getenv("AO_STATIC_ROWS") != nullptr;   // developer switch: the per-move packing of rounds 3 - 4 everywhere
    const int cap_rows = (!static_rows && e->row_cap > 0 && e->row_cap < rows) ? e->row_cap : rows;
    int in_kind = 1;
    ao::net_plan(net, cap_rows, &e->tp.il_group, &e->tp.nchq, &in_kind);
    // The padding channels of the fp32 input batch (planes 5..31 of a 32-channel slab) are zero and stay zero: the
    // encoder only writes the quads that hold planes (16 B per lane at a 2 KB stride are partial-line writes --
    // rocprof showed 152 MB of HBM writes per launch for a 42 MB batch). A change of layout re-zeroes the buffer.
    // The split-fp16 kernels take the planes as BITS instead (in_kind 2): one byte per cell, 81 contiguous bytes per
    // leaf, and the fp32 batch is not written at all.
    if (in_kind != 2 && (e->il_group_zeroed != e->tp.il_group || e->il_nchq_zeroed != e->tp.nchq)) {
        AO_HIP(e, hipMemsetAsync(e->tp.batch_il, 0, e->il_bytes, e->stream));
        e->il_group_zeroed = e->tp.il_group;
        e->il_nchq_zeroed = e->tp.nchq;
    }
    // select | net | expand+select | net | ... | expand(+idle select): one launch fewer per simulation
    // than the step-wise protocol, same device code (tree_device.hpp).
    ao::TreeParams p = e->tp;
    p.batch_nchw = nullptr;
    p.policy = e->d_policy;
    p.value = e->d_value;
    p.row_of_game = e->d_row;
    p.nchq_live = (e->cfg.inplanes + 3) / 4;
    const float* net_in = p.batch_il;
    if (in_kind == 2) {
        net_in = reinterpret_cast<const float*>(e->d_planes_u8);
        p.batch_u8 = e->d_planes_u8;
        p.batch_il = nullptr;
    }
    // A few games (the per-board network path): heads, tree step and the next leaf's conv1 are ONE launch per game
    // (k_step_board), the network is asked for the residual blocks alone -- 9 launches per simulation instead of 11.
    ao::StepNet step{};
    const bool fused = cap_rows == rows && ao::net_step_params(net, rows, e->d_policy, e->d_value, &step) != 0;
    static const bool force_dynamic = getenv("AO_DYNAMIC_ROWS") != nullptr;   // developer switch: hand out rows per simulation without ao_set_row_cap
    // (opt-in: the hand-out costs one atomic per workgroup on one word, ~11 ns each -- 10 us on top of a 20 us tree kernel when 4096
    // descents of equal depth arrive together, profiles/r5a_row_alloc_atomics.txt; it pays when leaves are terminal)
    const bool dynamic = !fused && !static_rows && (e->row_cap > 0 || force_dynamic);
    if (dynamic) {
        AO_HIP(e, hipMemsetAsync(e->d_live, 0, sizeof(unsigned) * ao_engine::kLive, e->stream));
        p.row_cap = static_cast<unsigned>(cap_rows);
    } else {
        AO_HIP(e, hipMemcpyAsync(e->d_row, e->h_row.data(), sizeof(int32_t) * 2 * e->G, hipMemcpyHostToDevice, e->stream));
    }
    int launch = 0, harvested = 0;   // selection launches of this move so far (launch i counts its rows in slot i % kLive); slots already summed up
    auto slot = [&](int i) -> unsigned* { return dynamic ? e->d_live + (i % ao_engine::kLive) : nullptr; };
    // Over-subscription (more active games than rows per simulation). Every launch a window of the game indices sits out, sized so
    // that the games that do descend ask for about cap_rows rows -- a share `ask_frac` of them does (the rest ends at terminal
    // leaves), measured over the previous move of this engine, less three standard deviations of that binomial; the window moves on
    // by its own length per launch (see select_game). `unfinished` = games that still need simulations.
    const bool oversub = dynamic && cap_rows < rows;
    p.order = (oversub && e->order_on && e->row_cap > 0 && e->row_cap < e->G) ? e->d_order : nullptr;
    static const int level_budget = getenv("AO_DESCENT_BUDGET") ? atoi(getenv("AO_DESCENT_BUDGET")) : 0;   // experiment: levels of a descent per launch
    p.max_levels = dynamic ? level_budget : 0;
    const bool catch_up = oversub || p.max_levels > 0;   // games may be short of their simulations after the nominal number of launches
    const int64_t rows_live_before = e->rs_rows_live;
    int64_t sims_wanted = 0;
    for (int g = 0; g < e->G; ++g)
        if (e->active[g]) sims_wanted += (e->status[g] == AO_ROOT_FRESH) ? e->S + 1 : e->S;
    unsigned sit_idx = 0;       // the window the move starts with (the host's estimate); the device takes it from there (sit_window)
    unsigned row_target = static_cast<unsigned>(cap_rows);
    if (oversub) {
        const double f = std::min(1.0, std::max(0.5, e->ask_frac));
        const double slack = 2.5 * std::sqrt(static_cast<double>(cap_rows) * (1.0 - f)) + 4.0;
        row_target = static_cast<unsigned>(std::max(1.0, cap_rows - slack));
        const double askers = std::min<double>(rows, std::floor(row_target / f));
        sit_idx = static_cast<unsigned>(std::lround((rows - askers) * e->G / static_cast<double>(rows)));
        if (sit_idx >= static_cast<unsigned>(e->G)) sit_idx = static_cast<unsigned>(e->G) - 1u;
        // the move STARTS with the window that is safe for any demand -- as many games descend as there are rows -- and the
        // controller opens it within a handful of launches: the first descents of a move ask for more rows than its average
        // (measured: 4741 - 4957 asked of 4096 at the average's window, ~2000 leaves per move sent waiting), and a leaf that
        // waits costs its game a launch at the END of the move, when the batch is nearly empty
        const double askers0 = std::min<double>(rows, row_target);
        unsigned sit0 = static_cast<unsigned>(std::lround((rows - askers0) * e->G / static_cast<double>(rows)));
        if (sit0 >= static_cast<unsigned>(e->G)) sit0 = static_cast<unsigned>(e->G) - 1u;
        const float sf = static_cast<float>(sit0);
        unsigned* init = e->h_live + ao_engine::kLive;   // (pinned; every move ends with a stream synchronisation)
        std::memset(init, 0, 8 * sizeof(unsigned));
        init[0] = sit0;
        std::memcpy(&init[2], &sf, 4);
        AO_HIP(e, hipMemcpyAsync(e->d_ctl, init, 8 * sizeof(unsigned), hipMemcpyHostToDevice, e->stream));
    }
    bool sit_off_forced = false;   // the catch-up loop below: nobody sits out any more
    auto set_sit = [&](int i) {
        p.ctl = (oversub && !sit_off_forced) ? e->d_ctl : nullptr;
        p.ctl_cur = i & 1;
        p.live_prev = (oversub && i > 0 && i % ao_engine::kLive != 0) ? slot(i - 1) : nullptr;   // (the ring was zeroed at a wrap: no demand reading for that launch)
        p.row_target = row_target;
    };
    // sums up the row counters of the selection launches [harvested, upto) -- each of them was followed by a network launch
    auto harvest_rows = [&](int upto) -> int {
        if (!dynamic || upto <= harvested) return 0;
        AO_HIP(e, hipMemcpyAsync(e->h_live, e->d_live, sizeof(unsigned) * ao_engine::kLive, hipMemcpyDeviceToHost, e->stream));
        AO_HIP(e, hipStreamSynchronize(e->stream));
        static const bool row_trace = getenv("AO_ROW_TRACE") != nullptr;   // developer switch: rows asked for, launch by launch
        if (row_trace) {
            fprintf(stderr, "AO_ROW_TRACE %d active games, %d rows, launches %d..%d, rows asked for:", rows, cap_rows, harvested, upto - 1);
            for (int i = harvested; i < upto; ++i) fprintf(stderr, " %u", e->h_live[i % ao_engine::kLive]);
            fprintf(stderr, "\n");
        }
        for (int i = harvested; i < upto; ++i) {
            const int64_t want = e->h_live[i % ao_engine::kLive];
            e->rs_rows_live += std::min<int64_t>(want, cap_rows);
            e->rs_waits += std::max<int64_t>(want - cap_rows, 0);
            e->rs_rows_launched += cap_rows;
            ++e->rs_launches;
        }
        harvested = upto;
        return 0;
    };
    // what: 1 = policy + value, 2 = simulation count + leaf status (the fused step computes the heads AND moves on to the next leaf
    // in one kernel: the two halves of a record are taken on either side of it)
    auto log_evals = [&](int what) {
        if (e->log_n > 0 && (e->log_sim + 1) * e->log_n * (e->A + 3) <= e->log_cap) {
            ao::launch_eval_log(e->d_log_games, e->log_n, e->d_row, e->d_policy, e->d_value, e->A,
                                e->log_dev + e->log_sim * e->log_n * (e->A + 3), e->tp.sims_done, e->tp.leaf_status, what, e->stream);
            if (what & 1) ++e->log_sim;
        }
    };
    bool first_sim = true;
    auto one_sim = [&]() -> int {
        if (ao::net_forward_il(net, net_in, cap_rows, e->d_policy, e->d_value, e->stream, in_kind, fused ? (first_sim ? 3 : 2) : 7,
                               slot(launch), static_cast<unsigned>(cap_rows)))
            return e->fail(std::string("network forward failed: ") + ao_net_last_error(net));
        first_sim = false;
        log_evals(fused ? 2 : 3);   // (before the tree kernel moves on / hands out the next simulation's rows)
        const bool timed = e->timing && (e->timing_tick++ % static_cast<unsigned>(e->timing_stride) == 0u);   // (see ao_tree_timing)
        if (timed) {
            if (e->ring_count == ao_engine::kRing) tree_harvest(e, ao_engine::kRing / 2);
            (void)hipEventRecord(e->ev0[e->ring_head], e->stream);
        }
        ++launch;
        if (dynamic && launch % ao_engine::kLive == 0) {   // the counter ring wraps (more than kLive launches in one move): sum it up, zero it
            if (harvest_rows(launch)) return 1;
            AO_HIP(e, hipMemsetAsync(e->d_live, 0, sizeof(unsigned) * ao_engine::kLive, e->stream));
        }
        if (fused) {
            ao::launch_step_board(p, step, rows, e->d_row + e->G, e->stream);
            log_evals(1);              // (the heads of this simulation ran inside the step kernel)
        } else {
            p.live = slot(launch);
            set_sit(launch);
            ao::launch_expand_select(p, e->stream);
        }
        if (timed) {
            (void)hipEventRecord(e->ev1[e->ring_head], e->stream);
            e->ring_head = (e->ring_head + 1) % ao_engine::kRing;
            ++e->ring_count;
        }
        AO_HIP(e, hipGetLastError());
        return 0;
    };
    if (e->sims_left > 0) {
        p.live = slot(0);
        set_sit(0);
        ao::launch_select(p, e->stream);
        AO_HIP(e, hipGetLastError());
    }
    // (Replaying the per-simulation launch sequence as a HIP graph was measured and is slower than
    // eager launches here: 71.7 vs 65.7 us per simulation for one game)
    int rc = 0;
    static const bool launch_timing = getenv("AO_LAUNCH_TIMING") != nullptr;   // developer switch: is the simulation loop host-bound?
    const auto lt0 = std::chrono::steady_clock::now();
    const int lt_sims = e->sims_left;
    while (e->sims_left > 0 && rc == 0) {
        rc = one_sim();
        --e->sims_left;
    }
    if (rc) return rc;
    if (oversub && sit_idx > 0) {
        // a game sits out sit_idx / G of the launches: that many more launches before anyone can be done
        const int planned = static_cast<int>(std::ceil(static_cast<double>(lt_sims) * e->G / (e->G - sit_idx))) - lt_sims;
        for (int k = 0; k < planned - 8 && rc == 0; ++k) rc = one_sim();   // (a few short: the deficit rounds below find out exactly)
        if (rc) return rc;
    }
    if (catch_up) {
        // over-subscribed: leaves that found their simulation's batch full were expanded one launch later, so some games are
        // short of their simulations. The counters say by how much; max(largest deficit, all deficits / rows per launch)
        // is a lower bound of the launches still needed -- run them, look again. The reference ALWAYS runs num_mcts simulations
        // (agents.py:105-132), so the loop ends in exactly two ways: every active game has its simulations, or an error. A round
        // without progress is not a stall yet -- a leaf that waited takes its row in one launch and is expanded by the next, and a
        // game may have sat out the only launch of the round -- so the sit-out window is switched OFF first (nobody sits out, and
        // every round runs two launches at least); two more rounds without progress after that are a real stall. Whatever is still
        // short when the loop ends is reported by k_end_move (ERR_SHORT) -- no result is built from fewer visits than asked for.
        const int G = e->G;
        int32_t* h_done = e->h_i32;
        int32_t* h_target = e->h_i32 + G;
        int32_t* h_err = e->h_i32 + 2 * G;
        int64_t last_sum = -1;
        const char* dev_env = getenv("AO_CATCHUP_ROUNDS");   // developer switch, read per search: cut the loop short (tests: 0 = a short search must be an error)
        const int dev_rounds = dev_env ? atoi(dev_env) : -1;
        const int max_rounds = dev_rounds >= 0 ? dev_rounds : 4 * (e->S + 2);
        bool window_off = getenv("AO_CATCHUP_WINDOW_OFF") != nullptr;   // developer switch (tests): the catch-up rounds run without a sit-out window from the start
        if (window_off) sit_off_forced = true;
        int stalled = 0;
        for (int round = 0; round < max_rounds; ++round) {
            AO_HIP(e, hipMemcpyAsync(h_done, e->tp.sims_done, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
            AO_HIP(e, hipMemcpyAsync(h_target, e->tp.sims_target, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
            AO_HIP(e, hipMemcpyAsync(h_err, e->tp.err, sizeof(int32_t) * G, hipMemcpyDeviceToHost, e->stream));
            AO_HIP(e, hipStreamSynchronize(e->stream));
            int64_t sum = 0;
            int mx = 0;
            bool bad = false;
            for (int g = 0; g < G; ++g) {
                if (!e->active[g]) continue;
                bad = bad || h_err[g] != 0;
                const int d = h_target[g] - h_done[g];
                if (d > 0) { sum += d; mx = std::max(mx, d); }
            }
            if (mx == 0 || bad) break;                      // done / a per-game error (reported by ao_end_move)
            if (sum == last_sum && p.max_levels == 0) {     // (with a level budget a round may pass without a finished simulation)
                if (!window_off) window_off = true;
                else if (++stalled >= 2) break;             // a real stall: ao_end_move names the games
            } else {
                stalled = 0;
            }
            last_sum = sum;
            if (window_off) sit_off_forced = true;
            int extra = static_cast<int>(std::max<int64_t>(mx, (sum + cap_rows - 1) / cap_rows));
            if (window_off) extra = std::max(extra, 2);
            for (int k = 0; k < extra && rc == 0; ++k) rc = one_sim();
            if (rc) return rc;
        }
    }
    if (harvest_rows(launch)) return 1;
    if (oversub && sims_wanted > 0)
        e->ask_frac = static_cast<double>(e->rs_rows_live - rows_live_before) / static_cast<double>(sims_wanted);
    if (launch_timing) {
        const auto lt1 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(e->stream);
        const auto lt2 = std::chrono::steady_clock::now();
        fprintf(stderr, "AO_LAUNCH_TIMING %d simulations: launches enqueued in %.1f us each, stream drained %.1f us after the last enqueue\n", lt_sims,
                std::chrono::duration<double, std::micro>(lt1 - lt0).count() / (lt_sims > 0 ? lt_sims : 1),
                std::chrono::duration<double, std::micro>(lt2 - lt1).count());
    }
    // The split-fp16 trunk clamps activations beyond the fp16 range and reports it: the evaluations of such a move are
    // not the fp32-equivalent ones the engine promises (checked before the per-game errors: clamped evaluations are what
    // makes priors degenerate). The move is then searched AGAIN, transparently, on the fp32-MFMA trunk: the games of this
    // move get their pre-move MT19937 streams back and FRESH trees at their current positions (ao_set_roots semantics:
    // what the search inherited from earlier moves is forgotten -- the one deviation from the reference, which would
    // have searched on top of its inherited counts), the caller gets a result instead of an exception and the event is
    // counted (ao_fp16_range_events; the Python layer turns it into a warning). The network returns to its requested
    // mode afterwards; from the third event of the SAME weights on (counted per network object since its last
    // ao_net_finalize) it stays on the fp32-MFMA trunk -- a checkpoint that keeps leaving the range would otherwise search
    // every move twice -- until new weights are loaded or ao_net_set_mode is called.
    int32_t nflags = 0;
    if (ao_net_status(net, e->stream, &nflags, 1)) return e->fail(std::string("ao_net_status: ") + ao_net_last_error(net));
    if (nflags & AO_NET_FP16_RANGE) {
        (void)hipMemsetAsync(e->tp.err, 0, sizeof(int32_t) * e->G, e->stream);
        (void)hipStreamSynchronize(e->stream);
        e->in_move = false;
        e->ended = false;
        if (!may_recover)
            return e->fail("ao_search: an activation left the fp16 range (|x| > 65504) in the split-fp16 trunk during the repeated move");
        const int G = e->G;
        std::vector<uint8_t> mask(G, 0);
        std::vector<int32_t> games, ns;
        std::vector<std::vector<int32_t>> saved;
        for (int g = 0; g < G; ++g) {
            if (!e->active[g]) continue;
            mask[g] = 1;
            games.push_back(g);
            saved.push_back(e->moves[g]);
            ns.push_back(static_cast<int32_t>(e->moves[g].size()));
        }
        std::vector<const int32_t*> ids;
        for (auto& m : saved) ids.push_back(m.data());
        AO_HIP(e, hipMemcpyAsync(e->tp.mt, e->d_mt_backup, sizeof(uint32_t) * 624 * G, hipMemcpyDeviceToDevice, e->stream));
        AO_HIP(e, hipMemcpyAsync(e->tp.mtpos, e->d_pos_backup, sizeof(int32_t) * G, hipMemcpyDeviceToDevice, e->stream));
        e->has_gauss = e->has_gauss_backup;
        e->gauss = e->gauss_backup;
        if (ao_reset(e, mask.data())) return 1;
        if (set_roots_impl(e, static_cast<int>(games.size()), games.data(), ids.data(), ns.data(), nullptr)) return 1;
        ao::net_fp16_fallback_begin(net);                // mode 2 for the repeated move
        ++e->fp16_events;
        e->fp16_games_redone += static_cast<int64_t>(games.size());
        const int rc2 = search_impl(e, net, active, tau, pi, visit, policy, false);
        ao::net_fp16_fallback_end(net);                  // back to the requested mode, unless these weights did it three times
        return rc2;
    }
    return ao_end_move(e, tau, pi, visit, policy);
}

int ao_search(ao_engine* e, ao_net* net, const uint8_t* active, const int8_t* tau, double* pi,
              double* visit, double* policy) {
    return search_impl(e, net, active, tau, pi, visit, policy, true);
}

int ao_set_row_cap(ao_engine* e, int32_t rows) {
    if (rows < 0) return e->fail("ao_set_row_cap: negative");
    if (e->in_move) return e->fail("ao_set_row_cap inside a move");
    e->row_cap = rows;
    return 0;
}

int ao_row_stats(ao_engine* e, int64_t* launches, int64_t* rows_live, int64_t* rows_launched, int64_t* waits) {
    if (launches) *launches = e->rs_launches;
    if (rows_live) *rows_live = e->rs_rows_live;
    if (rows_launched) *rows_launched = e->rs_rows_launched;
    if (waits) *waits = e->rs_waits;
    return 0;
}

int ao_set_eval_log(ao_engine* e, const int32_t* games, int32_t n, float* dev_log, int64_t capacity_floats) {
    if (n < 0 || n > e->G) return e->fail("ao_set_eval_log: bad game count");
    if (e->in_move) return e->fail("ao_set_eval_log inside a move");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    for (int k = 0; k < n; ++k)
        if (games[k] < 0 || games[k] >= e->G) return e->fail("ao_set_eval_log: game index out of range");
    if (n > 0) {
        if (!dev_log) return e->fail("ao_set_eval_log: null log buffer");
        AO_HIP(e, hipMemcpyAsync(e->d_log_games, games, sizeof(int32_t) * n, hipMemcpyHostToDevice, e->stream));
        AO_HIP(e, hipStreamSynchronize(e->stream));
    }
    e->log_n = n;
    e->log_dev = dev_log;
    e->log_cap = n > 0 ? capacity_floats : 0;
    e->log_sim = 0;
    return 0;
}

int ao_eval_log_count(ao_engine* e) { return static_cast<int>(e->log_sim); }

int ao_fp16_range_events(ao_engine* e, int64_t* moves_repeated, int64_t* games_redone) {
    if (moves_repeated) *moves_repeated = e->fp16_events;
    if (games_redone) *games_redone = e->fp16_games_redone;
    return 0;
}

int ao_host_threads(void) { return static_cast<int>(ao::HostPool::budget()); }

int ao_node_cap(ao_engine* e, int32_t* node_cap, int32_t* from_free_memory) {
    if (node_cap) *node_cap = e->cfg.node_cap;
    if (from_free_memory) *from_free_memory = e->node_cap_auto;
    return 0;
}

// ---- introspection ---------------------------------------------------------------------------
int ao_get_root_children(ao_engine* e, int g, int32_t* act, double* n, double* w, double* q, double* pr,
                         int32_t* count) {
    if (g < 0 || g >= e->G) return e->fail("game index out of range");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    int32_t cur = 0, root = -1;
    AO_HIP(e, hipMemcpy(&cur, e->tp.cur + g, 4, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(&root, e->tp.root_node + g, 4, hipMemcpyDeviceToHost));
    *count = 0;
    if (root < 0) return 0;
    const size_t slot = ao::node_slot(e->tp, cur, g, root);
    ao::Pos m;
    AO_HIP(e, hipMemcpy(&m, ao::nodePos(e->tp, slot), sizeof(m), hipMemcpyDeviceToHost));
    const int L = m.nchild;
    std::vector<int32_t> hn(L); std::vector<float> hw(L), hq(L); std::vector<double> hp(L); std::vector<uint8_t> ha(L);
    AO_HIP(e, hipMemcpy(hn.data(), ao::rowN(e->tp, slot), 4 * L, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(hw.data(), ao::rowW(e->tp, slot), 4 * L, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(hq.data(), ao::rowQ(e->tp, slot), 4 * L, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(hp.data(), ao::rowP(e->tp, slot), 8 * L, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(ha.data(), ao::rowACT(e->tp, slot), L, hipMemcpyDeviceToHost));
    for (int i = 0; i < L; ++i) {
        if (act) act[i] = ha[i];
        if (n) n[i] = hn[i];
        if (w) w[i] = hw[i];
        if (q) q[i] = hq[i];
        if (pr) pr[i] = hp[i];
    }
    *count = L;
    return 0;
}

int ao_tree_nodes(ao_engine* e, int g, int64_t* expanded, int64_t* dict_entries) {
    if (g < 0 || g >= e->G) return e->fail("game index out of range");
    AO_HIP(e, hipSetDevice(e->cfg.device));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    int32_t cur = 0, used = 0, root = -1;
    AO_HIP(e, hipMemcpy(&cur, e->tp.cur + g, 4, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(&used, e->tp.nodes_used + g, 4, hipMemcpyDeviceToHost));
    AO_HIP(e, hipMemcpy(&root, e->tp.root_node + g, 4, hipMemcpyDeviceToHost));
    // What the root reaches (what del_parents would leave): the arena may also hold nodes of earlier roots that no search can reach
    // any more -- it is compacted only when the next search needs their room (k_play) -- so the records in use are walked from the root.
    int64_t nodes = 0, entries = 0;
    if (used > 0 && root >= 0) {
        std::vector<unsigned char> recs(static_cast<size_t>(used) * e->tp.rec);
        AO_HIP(e, hipMemcpy(recs.data(), ao::node_rec(e->tp, ao::node_slot(e->tp, cur, g, 0)), recs.size(), hipMemcpyDeviceToHost));
        std::vector<int32_t> queue{root};
        for (size_t head = 0; head < queue.size(); ++head) {
            const unsigned char* rec = recs.data() + static_cast<size_t>(queue[head]) * e->tp.rec;
            const ao::Pos* m = reinterpret_cast<const ao::Pos*>(rec + 25u * e->tp.Ap);
            const int32_t* ch = reinterpret_cast<const int32_t*>(rec + 16u * e->tp.Ap);
            entries += m->nchild;
            for (int i = 0; i < m->nchild; ++i)
                if (ch[i] >= 0 && ch[i] < used && queue.size() < static_cast<size_t>(used)) queue.push_back(ch[i]);
        }
        nodes = static_cast<int64_t>(queue.size());
        entries += 1;
    }
    if (expanded) *expanded = nodes;
    if (dict_entries) *dict_entries = entries;
    return 0;
}

int ao_tree_timing(ao_engine* e, int enable, double* ms_total, int64_t* launches) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    if (e->ev0.empty() && enable) {
        e->ev0.resize(ao_engine::kRing);
        e->ev1.resize(ao_engine::kRing);
        for (int i = 0; i < ao_engine::kRing; ++i) {
            AO_HIP(e, hipEventCreate(&e->ev0[i]));
            AO_HIP(e, hipEventCreate(&e->ev1[i]));
        }
    }
    tree_harvest(e, e->ring_count);
    if (ms_total) *ms_total = e->ms_total;
    if (launches) *launches = e->launches;
    e->ms_total = 0.0;
    e->launches = 0;
    e->timing = enable != 0;
    e->timing_stride = enable > 1 ? enable : 1;   // every n-th launch carries the event pair (their cost: see net_forward_il)
    e->timing_tick = 0;
    return 0;
}

int ao_trim_stats(ao_engine* e, int64_t* subtrees_dropped, int64_t* reroots_trimmed) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    std::vector<int32_t> per(static_cast<size_t>(e->G) * 2);
    AO_HIP(e, hipMemcpyAsync(per.data(), e->tp.trimmed, sizeof(int32_t) * per.size(), hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    int64_t a = 0, b = 0;
    for (int g = 0; g < e->G; ++g) { a += per[2 * static_cast<size_t>(g)]; b += per[2 * static_cast<size_t>(g) + 1]; }
    if (subtrees_dropped) *subtrees_dropped = a;
    if (reroots_trimmed) *reroots_trimmed = b;
    return 0;
}

int ao_search_stats(ao_engine* e, int64_t* levels, int64_t* ties, int64_t* terminal, int64_t* evaluated) {
    AO_HIP(e, hipSetDevice(e->cfg.device));
    std::vector<unsigned> per(static_cast<size_t>(e->G) * 4);
    AO_HIP(e, hipMemcpyAsync(per.data(), e->tp.stats, sizeof(unsigned) * per.size(), hipMemcpyDeviceToHost, e->stream));
    AO_HIP(e, hipStreamSynchronize(e->stream));
    unsigned long long h[4] = {0, 0, 0, 0};
    for (int g = 0; g < e->G; ++g)
        for (int k = 0; k < 4; ++k) h[k] += per[static_cast<size_t>(g) * 4 + k];
    if (levels) *levels = static_cast<int64_t>(h[0]);
    if (ties) *ties = static_cast<int64_t>(h[1]);
    if (terminal) *terminal = static_cast<int64_t>(h[2]);
    if (evaluated) *evaluated = static_cast<int64_t>(h[3]);
    return 0;
}

}  // extern "C"
