// Does v_mfma_f32_16x16x32_f16 honour SUBNORMAL fp16 inputs on gfx950? (the low halves of small activations are
// subnormal: x < 2^-4 => xl < 2^-15).  hipcc --offload-arch=gfx950 tools/mfma_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float a, float b) {
    half8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = static_cast<_Float16>(a); B[i] = static_cast<_Float16>(b); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float cases[][2] = {{1.f, 1.f}, {9.5367431640625e-07f /*2^-20*/, 1.f}, {1.f, 9.5367431640625e-07f}, {5.9604644775390625e-08f /*2^-24*/, 1.f}, {9.5367431640625e-07f, 1024.f}};
    for (auto& c : cases) {
        k<<<1, 64>>>(d, c[0], c[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g: mfma sum over K=32 -> %g (exact %g)\n", c[0], c[1], h, 32.0 * c[0] * c[1]);
    }
    return 0;
}
