// tools/row_alloc_atomics.hip -- what does it cost to hand every game its evaluation-batch row with an atomic?
//
// Round 5: k_expand_select gives each game whose new leaf needs the network a row of the evaluation batch PER SIMULATION (terminal
// leaves take none), so the trunk evaluates live rows only. The allocation is `atomicAdd(&live, 1)` with the old value returned --
// thousands of waves on ONE address, device scope (eight XCDs, the word lives behind the fabric). This measures it in the shape of
// the tree kernel: G waves (4 per workgroup), each walks `levels` dependent 2 KB node reads first (so the arrivals spread as in
// the real kernel), then
//   mode 0  nothing                      (baseline)
//   mode 1  one returning atomic per wave, lane 0
//   mode 2  one returning atomic per WORKGROUP: the four waves meet at a barrier, wave 0 adds the group's count
//   mode 3  one non-returning atomic per wave (what the per-game counters of ao_search_stats already do)
// and writes 81 bytes at the row it got.
//
//   hipcc --offload-arch=gfx950 -O3 tools/row_alloc_atomics.hip -o /tmp/rowalloc && /tmp/rowalloc
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kRec = 2560;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_alloc(const unsigned char* arena, int games, int cap, int levels, int spread, unsigned* live,
                                               unsigned char* planes, unsigned* row_of_game) {
    __shared__ unsigned s_cnt, s_base;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + w;
    if (MODE == 2 && threadIdx.x == 0) s_cnt = 0;
    if (MODE == 2) __syncthreads();
    unsigned node = mix(g * 2654435761u) % cap;
    unsigned acc = 0;
    const int my_levels = levels + (spread ? static_cast<int>(mix(g + 77u) % (spread + 1)) : 0);
    for (int l = 0; l < my_levels; ++l) {
        const unsigned char* r = arena + (static_cast<size_t>(g) * cap + node) * kRec;
        unsigned h = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) h ^= reinterpret_cast<const unsigned*>(r)[lane + 64 * c];
        for (int o = 32; o; o >>= 1) h ^= __shfl_xor(h, o);
        acc ^= h;
        node = mix(node + h + l) % cap;
    }
    const bool need = (mix(g * 31u + acc) % 100u) < 89u;   // 11 % terminal leaves take no row
    unsigned row = 0xffffffffu;
    if (MODE == 1) {
        if (need) {
            if (lane == 0) row = atomicAdd(live, 1u);
            row = __shfl(row, 0);
        }
    } else if (MODE == 2) {
        unsigned mine = 0;
        if (need && lane == 0) mine = atomicAdd(&s_cnt, 1u);
        __syncthreads();
        if (threadIdx.x == 0) s_base = atomicAdd(live, s_cnt);
        __syncthreads();
        if (need) row = s_base + __shfl(mine, 0);
    } else if (MODE == 3) {
        if (need && lane == 0) atomicAdd(live, 1u);
        row = need ? g : 0xffffffffu;
    } else {
        row = need ? g : 0xffffffffu;
    }
    if (row != 0xffffffffu) {
        planes[static_cast<size_t>(row) * 128 + lane] = static_cast<unsigned char>(acc);
        if (lane < 17) planes[static_cast<size_t>(row) * 128 + 64 + lane] = static_cast<unsigned char>(acc >> 8);
        if (lane == 0) row_of_game[g] = row;
    }
}

int main() {
    const int G = 4096 + 1024, cap = 512;
    unsigned char* arena; unsigned* live; unsigned char* planes; unsigned* rog;
    hipMalloc(&arena, static_cast<size_t>(G) * cap * kRec);
    hipMemset(arena, 1, static_cast<size_t>(G) * cap * kRec);
    hipMalloc(&live, 4096 * 4);
    hipMalloc(&planes, static_cast<size_t>(G) * 128);
    hipMalloc(&rog, G * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int games : {4096, 5120}) {
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int levels = cfg == 0 ? 2 : cfg == 1 ? 14 : 4, spread = cfg == 0 ? 0 : cfg == 1 ? 40 : 0;
            printf("games %d, %d levels (+ up to %d):", games, levels, spread);
            for (int mode = 0; mode < 4; ++mode) {
                float best = 1e9f, sum = 0.f;
                for (int trial = 0; trial < 3; ++trial) {
                    hipMemset(live, 0, 4096 * 4);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    for (int r = 0; r < reps; ++r) {
                        unsigned* lv = live + (r % 1024);
                        switch (mode) {
                            case 0: hipLaunchKernelGGL(k_alloc<0>, dim3(games / 4), dim3(256), 0, 0, arena, games, cap, levels, spread, lv, planes, rog); break;
                            case 1: hipLaunchKernelGGL(k_alloc<1>, dim3(games / 4), dim3(256), 0, 0, arena, games, cap, levels, spread, lv, planes, rog); break;
                            case 2: hipLaunchKernelGGL(k_alloc<2>, dim3(games / 4), dim3(256), 0, 0, arena, games, cap, levels, spread, lv, planes, rog); break;
                            default: hipLaunchKernelGGL(k_alloc<3>, dim3(games / 4), dim3(256), 0, 0, arena, games, cap, levels, spread, lv, planes, rog); break;
                        }
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms = 0.f;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                    sum += ms;
                }
                unsigned h = 0;
                hipMemcpy(&h, live, 4, hipMemcpyDeviceToHost);
                printf("  mode %d %.2f us/launch (live[0] %u)", mode, best * 1000.f / reps, h);
            }
            printf("\n");
        }
    }
    return 0;
}
