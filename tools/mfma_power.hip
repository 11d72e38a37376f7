// Microbenchmark for the power-limited regime of the split-fp16 trunk: how much faster does a v_mfma_f32_16x16x32_f16 stream
// sustain when the operands of the two CORRECTION products (xh*wl, xl*wh: two of every three MFMAs) carry fewer significant
// bits, or are partly zero?  Stream shape: 8 waves per CU, 9 accumulators, products in the trunk's order (hh, hl, lh).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// lmask: AND-mask applied to both halves of every dword of the LOW operands (0xffff = full mantissa, 0xfff8 = low 3
// mantissa bits zero, ...); zero_frac256: that share (x/256) of the low operands' elements is exactly 0
__global__ void k(float* out, int iters, unsigned seed, unsigned lmask, unsigned zero_frac256, unsigned hi_zero_frac256) {
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 ah[3], al[3], bh[4], bl[4];
    unsigned h = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    auto mk = [&](unsigned mask, unsigned zf) {
        u32x4 q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            unsigned v = ((h & 0x83ff83ffu) | 0x38003800u | ((h >> 3) & 0x04000400u)) & (mask | (mask << 16));
            h = h * 1664525u + 1013904223u;
            if (((h >> 8) & 255u) < zf) v &= 0xffff0000u;
            if (((h >> 16) & 255u) < zf) v &= 0x0000ffffu;
            q[k] = v;
        }
        return __builtin_bit_cast(half8, q);
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) { ah[i] = mk(0xffffu, 0); al[i] = mk(lmask, 0); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { bh[i] = mk(0xffffu, hi_zero_frac256); bl[i] = mk(lmask, zero_frac256 > hi_zero_frac256 ? zero_frac256 : hi_zero_frac256); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + r) % 3], bh[(i * 4 + r) % 4], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[(i + r) % 3], bh[(i * 4 + r) % 4], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + r) % 3], bl[(i * 4 + r) % 4], acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double run(const char* tag, unsigned lmask, unsigned zf, unsigned hzf, int iters) {
    static float* d = nullptr;
    if (!d) hipMalloc(&d, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 200, 1u, lmask, zf, hzf);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, 1u, lmask, zf, hzf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 8 * iters * 4.0 * 9 * 3;
    const double tf = mfma * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-78s %8.1f ms  %7.1f TFLOP/s\n", tag, ms, tf);
    return tf;
}

int main() {
    const int it = 6000;   // ~0.25 s per line
    run("warm-up", 0xffffu, 0, 0, it);
    run("all operands full 11-bit mantissas, dense", 0xffffu, 0, 0, it);
    run("low operands (wl, xl): 8 significant bits (low 3 mantissa bits zero)", 0xfff8u, 0, 0, it);
    run("low operands: 6 significant bits", 0xffe0u, 0, 0, it);
    run("low operands: 4 significant bits", 0xff80u, 0, 0, it);
    run("low operands: 1 significant bit (powers of two)", 0xfc00u, 0, 0, it);
    run("xl 50 % zeros (xh dense)", 0xffffu, 128, 0, it);
    run("xh and xl 50 % zeros (post-ReLU activations)", 0xffffu, 128, 128, it);
    run("xh and xl 50 % zeros, low operands 8 significant bits", 0xfff8u, 128, 128, it);
    run("xh and xl 50 % zeros, low operands 6 significant bits", 0xffe0u, 128, 128, it);
    run("xh 50 % zeros, xl 75 % zeros", 0xffffu, 192, 128, it);
    run("xh 50 % zeros, xl all zeros", 0xffffu, 256, 128, it);
    run("all operands full 11-bit mantissas, dense (again)", 0xffffu, 0, 0, it);
    return 0;
}
