// calib_traffic.hip -- known-byte-count kernels in the tree kernels' access patterns, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE for NARROW accesses on gfx950 (the MI355X guide calibrates only wide streaming reads: x2).
//   hipcc --offload-arch=gfx950 -O3 tools/calib_traffic.hip -o gpurun_out/calib_traffic
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ... -- gpurun_out/calib_traffic     (and --pmc FETCH_SIZE)
// Every kernel moves exactly 64 MiB of payload over a 1 GiB region (each byte touched once: nothing can hit in a cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t PAYLOAD = 64ull << 20;

// one wave per "game": rows of 81 x 4 B like the N / W / Q / CH edge rows (324 B at a 384 B row stride)
__global__ void w_row4(uint32_t* p, size_t rows) {
    const size_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    uint32_t* row = p + r * 96;
    row[lane] = lane;
    if (lane + 64 < 81) row[lane + 64] = lane;
}
__global__ void r_row4(const uint32_t* p, size_t rows, uint32_t* out) {
    const size_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const uint32_t* row = p + r * 96;
    uint32_t v = row[lane];
    if (lane + 64 < 81) v += row[lane + 64];
    if (v == 0xdeadbeef) out[0] = v;
}
// lane 0 of every wave writes / reads ONE 4-byte word in its own 128 B line (per-game scalars: leaf_status, path_len ...)
__global__ void w_scalar(uint32_t* p, size_t n) {
    const size_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (r < n && (threadIdx.x & 63) == 0) p[r * 32] = 1;
}
__global__ void r_scalar(const uint32_t* p, size_t n, uint32_t* out) {
    const size_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (r < n && (threadIdx.x & 63) == 0 && p[r * 32] == 0xdeadbeef) out[0] = 1;
}
// wide streaming: 16 B per lane (the guide's calibrated case)
__global__ void w_wide(uint4* p, size_t n) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i < n) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ void r_wide(const uint4* p, size_t n, uint32_t* out) {
    const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    if (i < n) { const uint4 v = p[i]; if (v.x == 0xdeadbeef) out[0] = v.y; }
}
// one byte per lane, 81 contiguous bytes per wave at a 128 B stride (the bit planes)
__global__ void w_byte(uint8_t* p, size_t rows) {
    const size_t r = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    p[r * 128 + lane] = 1;
    if (lane + 64 < 81) p[r * 128 + lane + 64] = 1;
}

int main() {
    void* buf = nullptr;
    uint32_t* out = nullptr;
    if (hipMalloc(&buf, 1ull << 30) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 0, 1ull << 30);
    hipDeviceSynchronize();
    const size_t rows4 = PAYLOAD / 324;            // 324 payload bytes per row
    const size_t nsc = PAYLOAD / 4 / 16;           // 4 MiB of payload in scalars (16x fewer: they are slow)
    const size_t nwide = PAYLOAD / 16;
    const size_t rowsb = PAYLOAD / 81 / 8;         // 8 MiB of payload in bytes
    for (int rep = 0; rep < 3; ++rep) {
        w_row4<<<(rows4 + 3) / 4, 256>>>(static_cast<uint32_t*>(buf), rows4);
        r_row4<<<(rows4 + 3) / 4, 256>>>(static_cast<const uint32_t*>(buf), rows4, out);
        w_scalar<<<(nsc + 3) / 4, 256>>>(static_cast<uint32_t*>(buf), nsc);
        r_scalar<<<(nsc + 3) / 4, 256>>>(static_cast<const uint32_t*>(buf), nsc, out);
        w_wide<<<(nwide + 255) / 256, 256>>>(static_cast<uint4*>(buf), nwide);
        r_wide<<<(nwide + 255) / 256, 256>>>(static_cast<const uint4*>(buf), nwide, out);
        w_byte<<<(rowsb + 3) / 4, 256>>>(static_cast<uint8_t*>(buf), rowsb);
        hipDeviceSynchronize();
    }
    printf("payload bytes: row4 %zu scalar %zu wide %zu byte %zu\n", rows4 * 324, nsc * 4, nwide * 16, rowsb * 81);
    return 0;
}
