// replay.hip -- device-resident replay memory with on-device augmentation.
//
// Reference (paths relative to /root/reference/2_AlphaOmok/):
//   rep_memory = deque(maxlen=MEMORY_SIZE)                      main.py:55
//   rep_memory.extend(utils.augment_dataset(cur_memory, B))     main.py:229-231, utils.py:226-239
//   train_memory = random.sample(rep_memory, n); batches of 32  main.py:262-292
//
// The ring holds exactly what the deque holds -- (state [C,B,B], pi [A], z) tuples in deque order,
// oldest dropped first -- but in HBM: states as float32 (the planes are 0/1: exact), pi as float64 (the
// reference keeps float64 and casts at batch time; keeping it makes the pickled dataset of
// main.save_dataset bit-identical), z as float32. `extend` uploads the un-augmented samples once
// and a kernel writes the eight symmetries of each in the reference's order
// [r0, r0 flipped, r1, r1 flipped, r2, r2 flipped, r3, r3 flipped] (np.rot90 counter-clockwise,
// flip = reverse of the last axis). `gather` builds a float32 mini-batch for given deque indices
// (the indices come from the caller's `random.sample`, so the batch is the reference's batch).
// Both kernels are pure byte movement: HBM-bound, one thread per output element, coalesced writes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/omok_hip.h"

namespace ao {

// source cell of output cell (i, j) under symmetry sym = 2*r + f (rot90 by r, then flip if f)
__device__ __forceinline__ int sym_source(int sym, int i, int j, int B) {
    const int r = sym >> 1;
    const int jj = (sym & 1) ? B - 1 - j : j;
    int si, sj;
    switch (r) {
        case 0: si = i; sj = jj; break;
        case 1: si = jj; sj = B - 1 - i; break;
        case 2: si = B - 1 - i; sj = B - 1 - jj; break;
        default: si = B - 1 - jj; sj = i; break;
    }
    return si * B + sj;
}

// staged samples [n][C][A] f32 / [n][A] f64 / [n] f32  ->  ring slots (start + k) % cap,
// k = 8 * sample + sym (nsym = 8) or k = sample (nsym = 1)
__global__ void k_replay_write(const float* __restrict__ s_in, const double* __restrict__ pi_in,
                               const float* __restrict__ z_in, long n, int nsym, float* __restrict__ s_ring,
                               double* __restrict__ pi_ring, float* __restrict__ z_ring, long start, long cap, int C,
                               int B) {
    const int A = B * B;
    const long per = static_cast<long>(C + 1) * A;  // C state planes + one "plane" of pi per entry
    const long total = n * nsym * per;
    for (long t = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x; t < total;
         t += static_cast<long>(gridDim.x) * blockDim.x) {
        const long k = t / per;
        const int rem = static_cast<int>(t - k * per);
        const int plane = rem / A, cell = rem - plane * A;
        const long smp = k / nsym;
        const int sym = static_cast<int>(k - smp * nsym);
        const int src = (nsym == 1) ? cell : sym_source(sym, cell / B, cell % B, B);
        const long slot = (start + k) % cap;
        if (plane < C) {
            s_ring[(slot * C + plane) * A + cell] = s_in[(smp * C + plane) * A + src];
        } else {
            pi_ring[slot * A + cell] = pi_in[smp * A + src];
            if (cell == 0) z_ring[slot] = z_in[smp];
        }
    }
}

// Device-side sample emission (reference main.py:159-166, 219-227 builds every sample's state with utils.get_state_pt,
// utils.py:139-168, one Python call per ply): sample i is the position of episode ep_of[i] after t = ply_of[i] of its moves.
// One workgroup per sample: `when[cell]` = 1-based ply that put a stone on the cell (0: empty at that time), then plane
// C-2-j holds X_{t-j}, the stones of the mover of ply t-j after that ply (same colour as that ply, placed no later), and plane
// C-1 the colour to move (1.0 when black is to move). Writes the staged float32 states k_replay_write then spreads.
__global__ __launch_bounds__(128) void k_states_from_moves(const short* __restrict__ moves, int L, const int* __restrict__ ep_of,
                                                           const int* __restrict__ ply_of, long n, float* __restrict__ s_out,
                                                           int C, int A) {
    __shared__ short when[256];
    for (long smp = blockIdx.x; smp < n; smp += gridDim.x) {
        const int t = ply_of[smp];
        const short* mv = moves + static_cast<long>(ep_of[smp]) * L;
        for (int c = threadIdx.x; c < A; c += blockDim.x) when[c] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < t; i += blockDim.x) {
            const int m = mv[i];
            if (m >= 0 && m < A) when[m] = static_cast<short>(i + 1);
        }
        __syncthreads();
        float* out = s_out + smp * C * A;
        for (int idx = threadIdx.x; idx < C * A; idx += blockDim.x) {
            const int plane = idx / A, cell = idx - plane * A;
            bool one;
            if (plane == C - 1) {
                one = (t & 1) == 0;
            } else {
                const int p = t - (C - 2 - plane), w = when[cell];
                one = p >= 1 && w >= 1 && w <= p && ((w ^ p) & 1) == 0;
            }
            out[idx] = one ? 1.f : 0.f;
        }
        __syncthreads();
    }
}

// mini-batch: out_s [m][C][A] f32, out_pi [m][A] f32 (the reference's .float()), out_z [m] f32
__global__ void k_replay_gather(const float* __restrict__ s_ring, const double* __restrict__ pi_ring,
                                const float* __restrict__ z_ring, const long* __restrict__ slots, long m,
                                float* __restrict__ out_s, float* __restrict__ out_pi, float* __restrict__ out_z,
                                int C, int A) {
    const long per = static_cast<long>(C + 1) * A;
    const long total = m * per;
    for (long t = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x; t < total;
         t += static_cast<long>(gridDim.x) * blockDim.x) {
        const long k = t / per;
        const int rem = static_cast<int>(t - k * per);
        const long slot = slots[k];
        if (rem < C * A) {
            out_s[k * C * A + rem] = s_ring[slot * C * A + rem];
        } else {
            const int cell = rem - C * A;
            out_pi[k * A + cell] = static_cast<float>(pi_ring[slot * A + cell]);
            if (cell == 0) out_z[k] = z_ring[slot];
        }
    }
}

}  // namespace ao

struct ao_replay {
    int B = 0, C = 0, A = 0, device = 0;
    int64_t cap = 0, head = 0, count = 0;  // deque index i lives in slot (head + i) % cap
    float* s_ring = nullptr;
    double* pi_ring = nullptr;
    float* z_ring = nullptr;
    // staging (grow-only)
    float* st_s = nullptr; double* st_pi = nullptr; float* st_z = nullptr; long* st_idx = nullptr;
    int64_t st_n = 0, st_m = 0;
    short* st_mv = nullptr; int* st_ep = nullptr; int* st_ply = nullptr;   // moves-based extend: episodes' moves, (episode, ply) per sample
    int64_t st_mv_n = 0, st_ep_n = 0;
    std::string err;
    int fail(const std::string& m) { err = m; return 1; }
};

static thread_local std::string g_replay_create_error;

#define RP_HIP(r, call)                                                                        \
    do {                                                                                       \
        hipError_t st_ = (call);                                                               \
        if (st_ != hipSuccess) return (r)->fail(std::string(#call) + ": " + hipGetErrorString(st_)); \
    } while (0)

extern "C" {

int ao_replay_create(int board, int inplanes, int64_t capacity, int device, ao_replay** out) {
    *out = nullptr;
    if (board < 3 || board > 15 || inplanes < 1 || inplanes > 32 || capacity < 1) {
        g_replay_create_error = "ao_replay_create: board 3..15, inplanes 1..32, capacity >= 1";
        return 1;
    }
    ao_replay* r = new ao_replay;
    r->B = board; r->C = inplanes; r->A = board * board; r->device = device; r->cap = capacity;
    hipError_t st = hipSetDevice(device);
    if (st == hipSuccess) st = hipMalloc(&r->s_ring, static_cast<size_t>(capacity) * r->C * r->A * sizeof(float));
    if (st == hipSuccess) st = hipMalloc(&r->pi_ring, static_cast<size_t>(capacity) * r->A * sizeof(double));
    if (st == hipSuccess) st = hipMalloc(&r->z_ring, static_cast<size_t>(capacity) * sizeof(float));
    if (st != hipSuccess) {
        g_replay_create_error = std::string("ao_replay_create: ") + hipGetErrorString(st);
        if (r->s_ring) hipFree(r->s_ring);
        if (r->pi_ring) hipFree(r->pi_ring);
        if (r->z_ring) hipFree(r->z_ring);
        delete r;
        return 1;
    }
    *out = r;
    return 0;
}

void ao_replay_destroy(ao_replay* r) {
    if (!r) return;
    hipSetDevice(r->device);
    for (void* p : {static_cast<void*>(r->s_ring), static_cast<void*>(r->pi_ring), static_cast<void*>(r->z_ring),
                    static_cast<void*>(r->st_s), static_cast<void*>(r->st_pi), static_cast<void*>(r->st_z),
                    static_cast<void*>(r->st_idx), static_cast<void*>(r->st_mv), static_cast<void*>(r->st_ep),
                    static_cast<void*>(r->st_ply)})
        if (p) hipFree(p);
    delete r;
}

const char* ao_replay_last_error(const ao_replay* r) { return r ? r->err.c_str() : g_replay_create_error.c_str(); }

int64_t ao_replay_size(const ao_replay* r) { return r->count; }
int64_t ao_replay_capacity(const ao_replay* r) { return r->cap; }

int ao_replay_clear(ao_replay* r) {
    r->head = 0;
    r->count = 0;
    return 0;
}

// `skipped` samples logically precede the n given ones in this call but are not supplied: the caller knows that every
// entry of theirs would be overwritten by the given ones (n * nsym >= capacity), so only ring positions move for them.
struct MoveSource {   // states == nullptr: the states are built on the device from the episodes' moves
    const int16_t* moves; int64_t n_ep; int64_t L; const int32_t* ep_of; const int32_t* ply_of;
};

static int extend_impl(ao_replay* r, const float* states, const MoveSource* mv, const double* pi, const float* z, int64_t n,
                       int augment, int64_t skipped, void* stream) {
    if (n < 0 || skipped < 0) return r->fail("ao_replay_extend: negative sample count");
    if (n == 0 && skipped == 0) return 0;
    const int nsym = augment ? 8 : 1;
    if (r->cap < nsym) return r->fail("ao_replay_extend: capacity below one augmented sample (8 entries)");
    if (skipped > 0 && n * nsym < r->cap)
        return r->fail("ao_replay_extend_skip: the given samples do not fill the memory, so skipped ones would survive");
    RP_HIP(r, hipSetDevice(r->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t CA = static_cast<int64_t>(r->C) * r->A;
    // Only the newest `cap` entries of this call can survive (deque(maxlen) semantics); the slots
    // are written as if every entry had been appended in turn.
    const int64_t total = n * nsym;
    // entries of this call that are already overwritten by later entries of the same call are skipped:
    // whole samples from the first one that survives at least partially are staged and written
    const int64_t skip = std::max<int64_t>(0, total - r->cap);
    const int64_t first_smp = skip / nsym;
    const int64_t n_st = n - first_smp;
    if (n_st > r->st_n) {
        for (void* p : {static_cast<void*>(r->st_s), static_cast<void*>(r->st_pi), static_cast<void*>(r->st_z)})
            if (p) hipFree(p);
        r->st_s = nullptr; r->st_pi = nullptr; r->st_z = nullptr; r->st_n = 0;
        RP_HIP(r, hipMalloc(&r->st_s, static_cast<size_t>(n_st) * CA * sizeof(float)));
        RP_HIP(r, hipMalloc(&r->st_pi, static_cast<size_t>(n_st) * r->A * sizeof(double)));
        RP_HIP(r, hipMalloc(&r->st_z, static_cast<size_t>(n_st) * sizeof(float)));
        r->st_n = n_st;
    }
    if (mv) {
        if (mv->n_ep < 1 || mv->L < 1 || mv->L > r->A) return r->fail("ao_replay_extend_moves: n_episodes >= 1 and 1 <= max_len <= board * board");
        for (int64_t i = first_smp; i < n; ++i)
            if (mv->ep_of[i] < 0 || mv->ep_of[i] >= mv->n_ep || mv->ply_of[i] < 0 || mv->ply_of[i] > mv->L)
                return r->fail("ao_replay_extend_moves: sample " + std::to_string(i) + " names an episode or a ply outside the moves given");
        const int64_t nmv = mv->n_ep * mv->L;
        if (nmv > r->st_mv_n) {
            if (r->st_mv) hipFree(r->st_mv);
            r->st_mv = nullptr; r->st_mv_n = 0;
            RP_HIP(r, hipMalloc(&r->st_mv, static_cast<size_t>(nmv) * sizeof(short)));
            r->st_mv_n = nmv;
        }
        if (n_st > r->st_ep_n) {
            if (r->st_ep) hipFree(r->st_ep);
            if (r->st_ply) hipFree(r->st_ply);
            r->st_ep = nullptr; r->st_ply = nullptr; r->st_ep_n = 0;
            RP_HIP(r, hipMalloc(&r->st_ep, static_cast<size_t>(n_st) * sizeof(int)));
            RP_HIP(r, hipMalloc(&r->st_ply, static_cast<size_t>(n_st) * sizeof(int)));
            r->st_ep_n = n_st;
        }
        RP_HIP(r, hipMemcpyAsync(r->st_mv, mv->moves, static_cast<size_t>(nmv) * sizeof(short), hipMemcpyHostToDevice, s));
        RP_HIP(r, hipMemcpyAsync(r->st_ep, mv->ep_of + first_smp, static_cast<size_t>(n_st) * sizeof(int), hipMemcpyHostToDevice, s));
        RP_HIP(r, hipMemcpyAsync(r->st_ply, mv->ply_of + first_smp, static_cast<size_t>(n_st) * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ao::k_states_from_moves, dim3(static_cast<unsigned>(std::min<int64_t>(n_st, 65535L * 16))), dim3(128), 0, s,
                           r->st_mv, static_cast<int>(mv->L), r->st_ep, r->st_ply, static_cast<long>(n_st), r->st_s, r->C, r->A);
    } else {
        RP_HIP(r, hipMemcpyAsync(r->st_s, states + first_smp * CA, static_cast<size_t>(n_st) * CA * sizeof(float), hipMemcpyHostToDevice, s));
    }
    RP_HIP(r, hipMemcpyAsync(r->st_pi, pi + first_smp * r->A, static_cast<size_t>(n_st) * r->A * sizeof(double), hipMemcpyHostToDevice, s));
    RP_HIP(r, hipMemcpyAsync(r->st_z, z + first_smp, static_cast<size_t>(n_st) * sizeof(float), hipMemcpyHostToDevice, s));
    const int64_t tail = (r->head + r->count + (skipped % r->cap) * nsym) % r->cap;  // slot of the first given entry
    const long work = static_cast<long>(std::min<int64_t>(total, r->cap + nsym)) * (r->C + 1) * r->A;
    const int block = 256;
    const int grid = static_cast<int>(std::min<long>((work + block - 1) / block, 65535L * 8));
    if (skip == 0) {
        hipLaunchKernelGGL(ao::k_replay_write, dim3(grid), dim3(block), 0, s, r->st_s, r->st_pi, r->st_z,
                           static_cast<long>(n), nsym, r->s_ring, r->pi_ring, r->z_ring, static_cast<long>(tail),
                           static_cast<long>(r->cap), r->C, r->B);
    } else {
        // more new entries than slots: one launch per wrap so that no two threads of a launch share a slot
        int64_t k0 = first_smp * nsym;
        while (k0 < total) {
            const int64_t smp0 = k0 / nsym;
            const int64_t nsmp = std::min<int64_t>(n - smp0, std::max<int64_t>(1, r->cap / nsym));
            hipLaunchKernelGGL(ao::k_replay_write, dim3(grid), dim3(block), 0, s, r->st_s + (smp0 - first_smp) * CA,
                               r->st_pi + (smp0 - first_smp) * r->A, r->st_z + (smp0 - first_smp), static_cast<long>(nsmp), nsym,
                               r->s_ring, r->pi_ring, r->z_ring, static_cast<long>((tail + k0) % r->cap),
                               static_cast<long>(r->cap), r->C, r->B);
            k0 += nsmp * nsym;
        }
    }
    RP_HIP(r, hipGetLastError());
    RP_HIP(r, hipStreamSynchronize(s));  // the caller's buffers may go away
    const int64_t logical = (skipped + n) * nsym;
    const int64_t newcount = std::min<int64_t>(r->cap, r->count + logical);
    const int64_t dropped = r->count + logical - newcount;
    r->head = (r->head + dropped % r->cap) % r->cap;
    r->count = newcount;
    return 0;
}

int ao_replay_extend(ao_replay* r, const float* states, const double* pi, const float* z, int64_t n, int augment,
                     void* stream) {
    return extend_impl(r, states, nullptr, pi, z, n, augment, 0, stream);
}

int ao_replay_extend_skip(ao_replay* r, const float* states, const double* pi, const float* z, int64_t n, int augment,
                          int64_t skipped, void* stream) {
    return extend_impl(r, states, nullptr, pi, z, n, augment, skipped, stream);
}

int ao_replay_extend_moves(ao_replay* r, const int16_t* moves, int64_t n_episodes, int64_t max_len, const int32_t* ep_of,
                           const int32_t* ply_of, const double* pi, const float* z, int64_t n, int augment, int64_t skipped,
                           void* stream) {
    const MoveSource mv{moves, n_episodes, max_len, ep_of, ply_of};
    return extend_impl(r, nullptr, &mv, pi, z, n, augment, skipped, stream);
}

int ao_replay_gather(ao_replay* r, const int64_t* idx, int64_t m, float* dev_states, float* dev_pi, float* dev_z,
                     void* stream) {
    if (m <= 0) return m == 0 ? 0 : r->fail("ao_replay_gather: negative batch size");
    RP_HIP(r, hipSetDevice(r->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<long> slots(static_cast<size_t>(m));
    for (int64_t i = 0; i < m; ++i) {
        if (idx[i] < 0 || idx[i] >= r->count) return r->fail("ao_replay_gather: index out of range");
        slots[static_cast<size_t>(i)] = static_cast<long>((r->head + idx[i]) % r->cap);
    }
    if (m > r->st_m) {
        if (r->st_idx) hipFree(r->st_idx);
        r->st_idx = nullptr; r->st_m = 0;
        RP_HIP(r, hipMalloc(&r->st_idx, static_cast<size_t>(m) * sizeof(long)));
        r->st_m = m;
    }
    RP_HIP(r, hipMemcpyAsync(r->st_idx, slots.data(), static_cast<size_t>(m) * sizeof(long), hipMemcpyHostToDevice, s));
    const long work = static_cast<long>(m) * (r->C + 1) * r->A;
    const int block = 256;
    const int grid = static_cast<int>(std::min<long>((work + block - 1) / block, 65535L * 8));
    hipLaunchKernelGGL(ao::k_replay_gather, dim3(grid), dim3(block), 0, s, r->s_ring, r->pi_ring, r->z_ring, r->st_idx,
                       static_cast<long>(m), dev_states, dev_pi, dev_z, r->C, r->A);
    RP_HIP(r, hipGetLastError());
    RP_HIP(r, hipStreamSynchronize(s));  // `slots` goes away; the batch is ready for any stream
    return 0;
}

int ao_replay_read(ao_replay* r, int64_t first, int64_t n, double* states, double* pi, double* z) {
    if (first < 0 || n < 0 || first + n > r->count) return r->fail("ao_replay_read: range outside the memory");
    RP_HIP(r, hipSetDevice(r->device));
    RP_HIP(r, hipDeviceSynchronize());
    const int64_t CA = static_cast<int64_t>(r->C) * r->A;
    // at most two contiguous slot ranges (the ring wraps once)
    int64_t done = 0;
    while (done < n) {
        const int64_t slot = (r->head + first + done) % r->cap;
        const int64_t len = std::min<int64_t>(n - done, r->cap - slot);
        if (states) {
            std::vector<float> sb(static_cast<size_t>(len * CA));
            RP_HIP(r, hipMemcpy(sb.data(), r->s_ring + slot * CA, sb.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < sb.size(); ++k) states[done * CA + static_cast<int64_t>(k)] = sb[k];
        }
        if (pi)
            RP_HIP(r, hipMemcpy(pi + done * r->A, r->pi_ring + slot * r->A, sizeof(double) * r->A * len,
                                hipMemcpyDeviceToHost));
        if (z) {
            std::vector<float> zb(static_cast<size_t>(len));
            RP_HIP(r, hipMemcpy(zb.data(), r->z_ring + slot, sizeof(float) * len, hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < len; ++k) z[done + k] = zb[static_cast<size_t>(k)];
        }
        done += len;
    }
    return 0;
}

}  // extern "C"
