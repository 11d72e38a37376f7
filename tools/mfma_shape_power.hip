// Does the MFMA SHAPE change what the 1400 W cap lets the matrix pipe sustain?  v_mfma_f32_16x16x32_f16 (the trunk's
// instruction: 8192 multiply-adds per 1024 operand elements) against v_mfma_f32_32x32x16_f16 (16384 per 1024: half the
// register-file operand traffic per FLOP), same data-like operands, the trunk's product order (hh, hl, lh), 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_power.hip -o /tmp/mfma_shape_power && /tmp/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
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

template <int SHAPE>
__global__ void k(float* out, int iters, unsigned zf) {
    unsigned h = 1u + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    half8 ah[3], al[3], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) { ah[i] = mk(h, 0); al[i] = mk(h, 0); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { bh[i] = mk(h, zf); bl[i] = mk(h, zf); }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + r) % 3], bh[(i * 4 + r) % 4], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[(i + r) % 3], bh[(i * 4 + r) % 4], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + r) % 3], bl[(i * 4 + r) % 4], acc[i], 0, 0, 0);
                }
#pragma unroll
        for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(i + r) % 3], bh[(i + r) % 4], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[(i + r) % 3], bh[(i + r) % 4], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(i + r) % 3], bl[(i + r) % 4], acc[i], 0, 0, 0);
                }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
static void run(const char* tag, unsigned zf, int iters) {
    static float* d = nullptr;
    if (!d) hipMalloc(&d, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, 200, zf);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, iters, zf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 8 * iters * 4.0 * (SHAPE == 16 ? 9 : 4) * 3;
    const double tf = mfma * (SHAPE == 16 ? 16384.0 : 32768.0) / (ms * 1e-3) / 1e12;
    printf("%-70s %8.1f ms  %7.1f TFLOP/s\n", tag, ms, tf);
}

int main() {
    for (int rep = 0; rep < 3; ++rep) {
        run<16>("16x16x32 f16, dense data-like operands", 0, 24000);
        run<32>("32x32x16 f16, dense data-like operands", 0, 27000);
        run<16>("16x16x32 f16, activations 50 % zeros", 128, 24000);
        run<32>("32x32x16 f16, activations 50 % zeros", 128, 27000);
    }
    return 0;
}
