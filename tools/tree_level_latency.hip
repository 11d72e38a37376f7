// tools/tree_level_latency.hip -- what ONE level of the PUCT descent costs, taken apart.
//
// With a trained network k_expand_select lasts as long as the deepest of its 4096 descents (profiles/r4d_tree_deep_phases.txt: ~8 k
// cycles per level in the -DAO_PROF build, 3.5 - 5 k of them between requesting the node record and having it). A plain HBM miss is
// ~900 cycles (MI355X guide). This replays the descent's dependent chain on the engine's record-per-node arena
// (tools/tree_layout_latency.hip showed the layout effect) and varies one thing at a time:
//   waves     4096 (every game descending) / 256 / 8 (the tail of a launch: a few stragglers alone on the chip)
//   footprint `used` nodes per game out of `cap` (address spread: translation and cache reach)
//   bytes     the rows the selection reads (P f64, N, Q, CH i32, ACT u8 + 80-byte position: ~2.0 KB of the 2560-byte record, ten
//             vector loads) / a "lite" record (N, Q, P as f32, CH i16, ACT u8 = 15 B per edge + position in 1536 B) / one 16-byte load
//   arithmetic none / the selection's fp64 chain (mul, mul, div, add per edge, two wave reductions)
//
//   hipcc --offload-arch=gfx950 -O3 tools/tree_level_latency.hip -o /tmp/tlv && /tmp/tlv
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int kAp = 96;
constexpr int kRec = 2560;

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: the engine's rows; 1: lite record; 2: one 16-byte load per level.  ARITH: the fp64 PUCT chain on the loaded values.
template <int MODE, bool ARITH>
__global__ __launch_bounds__(64) void k_chain(const unsigned char* arena, int games, int cap, int used, int levels, unsigned* out, long long* ticks) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x;
    if (g >= games) return;
    unsigned node = mix(g * 2654435761u) % used;
    unsigned acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < levels; ++l) {
        const unsigned char* r = arena + (static_cast<size_t>(g) * cap + node) * kRec;
        unsigned h = 0;
        double best = -1e300;
        if (MODE == 0) {
            const uint4 m = reinterpret_cast<const uint4*>(r + 2432)[lane & 3];
            h = m.x ^ m.y;
            int n[2], ch[2], ac[2];
            float q[2];
            double pv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = lane + 64 * c;
                const bool in = e < kAp;
                pv[c] = in ? reinterpret_cast<const double*>(r)[e] : 0.0;
                n[c] = in ? reinterpret_cast<const int32_t*>(r + 768)[e] : 0;
                q[c] = in ? reinterpret_cast<const float*>(r + 1152)[e] : 0.f;
                ch[c] = in ? reinterpret_cast<const int32_t*>(r + 1536)[e] : 0;
                ac[c] = in ? r[1920 + e] : 0;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h += static_cast<unsigned>(n[c]) + __float_as_uint(q[c]) + static_cast<unsigned>(__double2hiint(pv[c])) + ch[c] + ac[c];
                if (ARITH) {
                    double t = __dmul_rn(5.0, fabs(pv[c]) + 1e-3);
                    t = __dmul_rn(t, __dsqrt_rn(static_cast<double>((n[c] & 1023) + 7)));
                    const double u = __ddiv_rn(t, static_cast<double>((n[c] & 255) + 1));
                    const double s = __dadd_rn(static_cast<double>(q[c]), u);
                    best = s > best ? s : best;
                }
            }
        } else if (MODE == 1) {
            const uint4 m = reinterpret_cast<const uint4*>(r + 1456)[lane & 3];
            h = m.x ^ m.y;
            int n[2], ch[2], ac[2];
            float q[2], pf[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = lane + 64 * c;
                const bool in = e < kAp;
                pf[c] = in ? reinterpret_cast<const float*>(r)[e] : 0.f;
                n[c] = in ? reinterpret_cast<const int32_t*>(r + 384)[e] : 0;
                q[c] = in ? reinterpret_cast<const float*>(r + 768)[e] : 0.f;
                ch[c] = in ? reinterpret_cast<const int16_t*>(r + 1152)[e] : 0;
                ac[c] = in ? r[1344 + e] : 0;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                h += static_cast<unsigned>(n[c]) + __float_as_uint(q[c]) + __float_as_uint(pf[c]) + ch[c] + ac[c];
                if (ARITH) {
                    double t = __dmul_rn(5.0, fabs(static_cast<double>(pf[c])) + 1e-3);
                    t = __dmul_rn(t, __dsqrt_rn(static_cast<double>((n[c] & 1023) + 7)));
                    const double u = __ddiv_rn(t, static_cast<double>((n[c] & 255) + 1));
                    const double s = __dadd_rn(static_cast<double>(q[c]), u);
                    best = s > best ? s : best;
                }
            }
        } else {
            const uint4 m = reinterpret_cast<const uint4*>(r)[lane & 3];
            h = m.x ^ m.y ^ m.z;
        }
        if (ARITH) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double other = __shfl_xor(best, o);
                best = other > best ? other : best;
            }
            h += static_cast<unsigned>(__double2loint(best));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
        acc += h;
        node = mix(h + l) % used;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[g] = acc; ticks[g] = t1 - t0; }
}

template <int MODE, bool ARITH>
static void run(const char* what, const unsigned char* arena, int games, int cap, int used, int levels, unsigned* out, long long* ticks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_chain<MODE, ARITH>), dim3(games), dim3(64), 0, 0, arena, games, cap, used + rep, levels, out, ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    long long t[8];
    hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    printf("%-34s waves %4d  used %5d of %5d  : %8.1f us = %6.3f us per level   (wave 0: %lld memtime ticks per level)\n", what, games, used, cap, best * 1e3,
           best * 1e3 / levels, t[0] / levels);
}

int main() {
    const int cap = 6416, levels = 40;
    const int maxg = 4096;
    const size_t bytes = static_cast<size_t>(maxg) * cap * kRec;
    unsigned char* arena;
    unsigned* out;
    long long* ticks;
    if (hipMalloc(&arena, bytes) != hipSuccess) { printf("no room\n"); return 1; }
    hipMalloc(&out, maxg * 4);
    hipMalloc(&ticks, maxg * 8);
    hipMemset(arena, 0x5a, bytes);
    printf("arena %.0f GB, %d levels per chain; s_memtime runs at a constant 100 MHz\n", bytes / 1e9, levels);
    for (int games : {4096, 256, 8}) {
        for (int used : {3000, 300, 30}) {
            run<0, false>("engine rows (2.0 KB, 11 loads)", arena, games, cap, used, levels, out, ticks);
            run<0, true>("engine rows + fp64 PUCT chain", arena, games, cap, used, levels, out, ticks);
            run<1, false>("lite record (1.5 KB, 11 loads)", arena, games, cap, used, levels, out, ticks);
            run<1, true>("lite record + fp64 PUCT chain", arena, games, cap, used, levels, out, ticks);
            run<2, false>("one 16-byte load", arena, games, cap, used, levels, out, ticks);
        }
    }
    // the same chains with every game's nodes packed (cap = used): does the address spread between games matter?
    for (int games : {4096, 8}) {
        run<0, false>("engine rows, arenas packed", arena, games, 3000, 3000, levels, out, ticks);
        run<2, false>("one 16-byte load, arenas packed", arena, games, 3000, 3000, levels, out, ticks);
    }
    return 0;
}
