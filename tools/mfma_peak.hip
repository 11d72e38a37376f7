// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate with W waves per SIMD and 9 independent
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
    return 0;
}
