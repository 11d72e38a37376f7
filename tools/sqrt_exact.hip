// Is the device's double sqrt correctly rounded for the integers the PUCT formula takes it of (np.sqrt(total_n), agents.py:158)?
//   hipcc --offload-arch=gfx950 -O3 tools/sqrt_exact.hip -o /tmp/sqrt_exact && /tmp/sqrt_exact
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __dsqrt_rn(static_cast<double>(i));
}
int main() {
    const int n = 1 << 24;
    double* d; hipMalloc(&d, sizeof(double) * n);
    k<<<(n + 255) / 256, 256>>>(d, n);
    std::vector<double> h(n);
    hipMemcpy(h.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) if (h[i] != std::sqrt(static_cast<double>(i))) { if (bad < 5) printf("mismatch at %d: %.17g vs %.17g\n", i, h[i], std::sqrt((double)i)); ++bad; }
    printf("__dsqrt_rn(i) == host sqrt(i) (IEEE correctly rounded) for i in [0, 2^24): %ld mismatches\n", bad);
    return 0;
}
