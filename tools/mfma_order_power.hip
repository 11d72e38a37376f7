// Does the ORDER of the MFMAs change what the power cap lets the matrix pipe sustain?  Same instruction
// (v_mfma_f32_16x16x32_f16), same data-like operands, 9 accumulators x 3 products x 4 B fragments per iteration; what
// varies is which operand stays the same between consecutive instructions:
//   0  trunk-like: per accumulator the three products back to back (A: wh, wl, wh; B: xh, xh, xl): an operand changes every time
//   1  A-stationary: one A fragment against all accumulators' B fragments in a row, then the next A fragment
//   2  B-stationary: one B fragment with all A fragments in a row
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_order_power.hip -o /tmp/mop && /tmp/mop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ half8 mk(unsigned& h, unsigned zf) {
    u32x4 q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h = h * 1664525u + 1013904223u;
        unsigned v = (h & 0x83ff83ffu) | 0x38003800u | ((h >> 3) & 0x04000400u);
        h = h * 1664525u + 1013904223u;
        if (((h >> 8) & 255u) < zf) v &= 0xffff0000u;
        if (((h >> 16) & 255u) < zf) v &= 0x0000ffffu;
        q[k] = v;
    }
    return __builtin_bit_cast(half8, q);
}

template <int ORDER>
__global__ void k(float* out, int iters, unsigned zf) {
    unsigned h = 1u + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    half8 a[6], b[6];   // a: 3 x (wh, wl); b: 3 x (xh, xl)
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = mk(h, 0); b[i] = mk(h, zf); }
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // accumulator i = (A index i / 3, B index i % 3): 27 MFMAs per round either way, each accumulator gets hh, hl, lh
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ORDER == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * (i / 3)], b[2 * (i % 3)], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * (i / 3) + 1], b[2 * (i % 3)], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * (i / 3)], b[2 * (i % 3) + 1], acc[i], 0, 0, 0);
                }
            } else if (ORDER == 1) {
#pragma unroll
                for (int ai = 0; ai < 3; ++ai) {
#pragma unroll
                    for (int bi = 0; bi < 3; ++bi) acc[ai * 3 + bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * ai], b[2 * bi], acc[ai * 3 + bi], 0, 0, 0);
#pragma unroll
                    for (int bi = 0; bi < 3; ++bi) acc[ai * 3 + bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * ai], b[2 * bi + 1], acc[ai * 3 + bi], 0, 0, 0);
#pragma unroll
                    for (int bi = 0; bi < 3; ++bi) acc[ai * 3 + bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * ai + 1], b[2 * bi], acc[ai * 3 + bi], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int bi = 0; bi < 3; ++bi) {
#pragma unroll
                    for (int ai = 0; ai < 3; ++ai) acc[ai * 3 + bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * ai], b[2 * bi], acc[ai * 3 + bi], 0, 0, 0);
#pragma unroll
                    for (int ai = 0; ai < 3; ++ai) acc[ai * 3 + bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * ai + 1], b[2 * bi], acc[ai * 3 + bi], 0, 0, 0);
#pragma unroll
                    for (int ai = 0; ai < 3; ++ai) acc[ai * 3 + bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * ai], b[2 * bi + 1], acc[ai * 3 + bi], 0, 0, 0);
                }
            }
            // new operands every round (as the trunk's slabs bring new fragments), cheaply: rotate
            const half8 t = a[0];
#pragma unroll
            for (int i = 0; i < 5; ++i) a[i] = a[i + 1];
            a[5] = t;
            const half8 u = b[0];
#pragma unroll
            for (int i = 0; i < 5; ++i) b[i] = b[i + 1];
            b[5] = u;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ORDER>
static void run(const char* tag, unsigned zf, int iters) {
    static float* d = nullptr;
    if (!d) hipMalloc(&d, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(512), 0, 0, d, 200, zf);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(512), 0, 0, d, iters, zf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 8 * iters * 4.0 * 27;
    printf("%-64s %8.1f ms  %7.1f TFLOP/s\n", tag, ms, mfma * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
    for (int rep = 0; rep < 3; ++rep) {
        run<0>("order 0 (per accumulator hh, hl, lh), activations 50 % zeros", 128, 40000);
        run<1>("order 1 (A-stationary runs), activations 50 % zeros", 128, 40000);
        run<2>("order 2 (B-stationary runs), activations 50 % zeros", 128, 40000);
    }
    return 0;
}
