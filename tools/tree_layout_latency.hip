// tools/tree_layout_latency.hip -- is the per-level latency of the PUCT descent a property of the tree's LAYOUT?
//
// k_expand_select with a trained network walks 14 levels per simulation and game; -DAO_PROF shows 4 - 5.4 k cycles per level
// waiting for the level's loads (profiles/r4d_tree_deep_phases.txt). A level reads one node's rows out of SIX separate arrays
// (N, Q, P, CH, ACT + the 80-byte node record), each many GB large: seven pages, seven DRAM rows per level. This microbenchmark
// replays that access pattern -- 4096 wavefronts, each a chain of dependent levels, the next node derived from the loaded data --
// on (a) the engine's structure of arrays and (b) one interleaved record per node (2.5 KB contiguous), same total footprint.
//
//   hipcc --offload-arch=gfx950 -O3 tools/tree_layout_latency.hip -o /tmp/tll && /tmp/tll
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kAp = 96;                 // edge slots of a 9x9 node
constexpr int kRec = 2560;              // interleaved record: P 768 | N 384 | Q 384 | W 384 | CH 384 | ACT 96 | pad | node record 80

struct Soa {
    int32_t* N; float* W; float* Q; double* P; int32_t* CH; uint8_t* ACT; uint4* meta;   // meta: 5 x uint4 per node
};

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <bool AOS>
__global__ __launch_bounds__(256) void k_walk(Soa s, const unsigned char* aos, int games, int cap, int used, int levels, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= games) return;
    unsigned node = mix(g * 2654435761u) % used;
    unsigned acc = 0;
    for (int l = 0; l < levels; ++l) {
        const size_t slot = static_cast<size_t>(g) * cap + node;
        int n[2], ch[2], ac[2];
        float q[2];
        double pv[2];
        uint4 m;
        if (AOS) {
            const unsigned char* r = aos + slot * kRec;
            m = reinterpret_cast<const uint4*>(r + 2432)[lane & 3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = lane + 64 * c;
                const bool in = e < kAp;
                pv[c] = in ? reinterpret_cast<const double*>(r)[e] : 0.0;
                n[c] = in ? reinterpret_cast<const int32_t*>(r + 768)[e] : 0;
                q[c] = in ? reinterpret_cast<const float*>(r + 1152)[e] : 0.f;
                ch[c] = in ? reinterpret_cast<const int32_t*>(r + 1920)[e] : 0;
                ac[c] = in ? r[2304 + e] : 0;
            }
        } else {
            const size_t eb = slot * kAp;
            m = s.meta[slot * 5 + (lane & 3)];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = lane + 64 * c;
                const bool in = e < kAp;
                pv[c] = in ? s.P[eb + e] : 0.0;
                n[c] = in ? s.N[eb + e] : 0;
                q[c] = in ? s.Q[eb + e] : 0.f;
                ch[c] = in ? s.CH[eb + e] : 0;
                ac[c] = in ? s.ACT[eb + e] : 0;
            }
        }
        // something of every loaded value decides the next node (a dependent chain, as the PUCT arg-max is)
        unsigned h = m.x ^ m.y;
#pragma unroll
        for (int c = 0; c < 2; ++c) h += static_cast<unsigned>(n[c]) + __float_as_uint(q[c]) + static_cast<unsigned>(__double2hiint(pv[c])) + ch[c] + ac[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
        acc += h;
        node = mix(h + l) % used;
    }
    if (lane == 0) out[g] = acc;
}

int main() {
    const int games = 4096, cap = 6416, used = 3000, levels = 15;
    const size_t slots = static_cast<size_t>(games) * cap;      // ONE arena per game here (the engine has two; one is live)
    Soa s;
    unsigned char* aos;
    unsigned* out;
    hipMalloc(&s.N, slots * kAp * 4); hipMalloc(&s.W, slots * kAp * 4); hipMalloc(&s.Q, slots * kAp * 4);
    hipMalloc(&s.P, slots * kAp * 8); hipMalloc(&s.CH, slots * kAp * 4); hipMalloc(&s.ACT, slots * kAp);
    hipMalloc(&s.meta, slots * 80);
    hipMalloc(&out, games * 4);
    // data: the low bytes of the addresses are as good as random for the chain; memset patterns differ per array
    hipMemset(s.N, 0x11, slots * kAp * 4); hipMemset(s.Q, 0x22, slots * kAp * 4); hipMemset(s.P, 0x33, slots * kAp * 8);
    hipMemset(s.CH, 0x44, slots * kAp * 4); hipMemset(s.ACT, 0x55, slots * kAp); hipMemset(s.meta, 0x66, slots * 80);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](bool aos_mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (aos_mode) hipLaunchKernelGGL(k_walk<true>, dim3(games / 4), dim3(256), 0, 0, s, aos, games, cap, used + rep, levels, out);
            else hipLaunchKernelGGL(k_walk<false>, dim3(games / 4), dim3(256), 0, 0, s, aos, games, cap, used + rep, levels, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        return best;
    };
    const float t_soa = timeit(false);
    printf("structure of arrays (7 allocations, %.0f GB): %d levels x %d games: %.1f us = %.2f us per level\n",
           slots * (kAp * 21.0 + 80.0) / 1e9, levels, games, t_soa * 1e3, t_soa * 1e3 / levels);
    hipFree(s.W);   // (room for the interleaved copy)
    if (hipMalloc(&aos, slots * kRec) != hipSuccess) { printf("no room for the interleaved arena\n"); return 1; }
    hipMemset(aos, 0x5a, slots * kRec);
    const float t_aos = timeit(true);
    printf("one %d-byte record per node (%.0f GB):          %d levels x %d games: %.1f us = %.2f us per level\n", kRec, slots * (double)kRec / 1e9,
           levels, games, t_aos * 1e3, t_aos * 1e3 / levels);
    return 0;
}
