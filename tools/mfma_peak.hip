// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x32_f16 rates with W waves per SIMD and 9 independent
// accumulators per wave (the shape of k_trunk16's inner loop, without any loads).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The split-fp16 kernels' instruction: v_mfma_f32_16x16x32_f16 (16384 FLOP). NOISY = operands are
// pseudo-random fp16 bit patterns that differ per lane and per MFMA (dense toggling, what real data
// does to the power budget); otherwise small constants.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool NOISY>
__global__ void kh(float* out, int iters, unsigned seed) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 a[3], b[4];
    unsigned h = seed + threadIdx.x * 2654435761u;
    auto mk = [&]() {
        u32x4 q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            // fp16 pairs with exponents near 1.0 (no inf / NaN): sign + exponent 01110..01111 + random mantissa
            q[k] = NOISY ? ((h & 0x83ff83ffu) | 0x38003800u | ((h >> 3) & 0x04000400u)) : 0x3c003c00u;
        }
        return __builtin_bit_cast(half8, q);
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = mk();
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = mk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) % 3], b[(i * 4 + r) % 4], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_mfma_f32_32x32x16_f16 (32768 FLOP, half the operand bytes per FLOP of the 16x16x32 form), same operand recipe
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool NOISY>
__global__ void kh32(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    half8 a[3], b[4];
    unsigned h = seed + threadIdx.x * 2654435761u;
    auto mk = [&]() {
        u32x4 q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            q[k] = NOISY ? ((h & 0x83ff83ffu) | 0x38003800u | ((h >> 3) & 0x04000400u)) : 0x3c003c00u;
        }
        return __builtin_bit_cast(half8, q);
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = mk();
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = mk();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) % 3], b[(i * 4 + r) % 4], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool NOISY>
void run_h32(int waves_per_cu, const char* tag, int iters = 20000) {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kh32<NACC, NOISY>), dim3(256), dim3(64 * waves_per_cu), 0, 0, d, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kh32<NACC, NOISY>), dim3(256), dim3(64 * waves_per_cu), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * waves_per_cu * iters * 4.0 * NACC;
    const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
    const double cyc = (ms * 1e-3 * 2.4e9) / (mfma / 1024.0);
    printf("f16 32x32x16 %s %s: %d waves/CU, %d acc: %.3f ms, %.1f TFLOP/s, %.2f cycles/MFMA/SIMD at 2.4 GHz\n", tag,
           NOISY ? "random operands" : "constant operands", waves_per_cu, NACC, ms, tf, cyc);
    hipFree(d);
}

template <int NACC, bool NOISY>
void run_h(int waves_per_cu, const char* tag, int iters = 40000) {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kh<NACC, NOISY>), dim3(256), dim3(64 * waves_per_cu), 0, 0, d, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kh<NACC, NOISY>), dim3(256), dim3(64 * waves_per_cu), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * waves_per_cu * iters * 4.0 * NACC;
    const double tf = mfma * 16384.0 / (ms * 1e-3) / 1e12;
    const double cyc = (ms * 1e-3 * 2.4e9) / (mfma / 1024.0);
    printf("f16 16x16x32 %s %s: %d waves/CU, %d acc: %.3f ms, %.1f TFLOP/s, %.2f cycles/MFMA/SIMD at 2.4 GHz\n", tag,
           NOISY ? "random operands" : "constant operands", waves_per_cu, NACC, ms, tf, cyc);
    hipFree(d);
}

template <int NACC>
void run(int waves_per_cu, const char* tag) {
    float* d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(64 * waves_per_cu), 96 * 1024, 0, d, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(64 * waves_per_cu), 96 * 1024, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * waves_per_cu * iters * 4.0 * NACC;
    const double tf = mfma * 2048.0 / (ms * 1e-3) / 1e12;
    const double cyc_per_mfma_simd = (ms * 1e-3 * 2.4e9) / (mfma / 1024.0);
    printf("%s: %d waves/CU, %d acc: %.3f ms, %.1f TFLOP/s, %.2f cycles/MFMA/SIMD at 2.4 GHz\n", tag, waves_per_cu, NACC, ms,
           tf, cyc_per_mfma_simd);
    hipFree(d);
}

int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    run<9>(4, "1 wave/SIMD");
    run<9>(8, "2 waves/SIMD");
    run<4>(4, "1 wave/SIMD");
    run<4>(8, "2 waves/SIMD");
    run_h<9, false>(4, "1 wave/SIMD");
    run_h<9, false>(8, "2 waves/SIMD");
    run_h<9, true>(4, "1 wave/SIMD");
    run_h<9, true>(8, "2 waves/SIMD");
    run_h<9, true>(8, "2 waves/SIMD (again, warm)");
    run_h<9, true>(8, "2 waves/SIMD, 0.5 s sustained", 800000);
    run_h<9, true>(8, "2 waves/SIMD, after that", 40000);
    run_h32<4, false>(8, "2 waves/SIMD");
    run_h32<4, true>(8, "2 waves/SIMD");
    run_h32<4, true>(8, "2 waves/SIMD (again, warm)");
    run_h32<4, true>(8, "2 waves/SIMD, 0.5 s sustained", 400000);
    run_h<9, true>(8, "16x16x32 again");
    return 0;
}
