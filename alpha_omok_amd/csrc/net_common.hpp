// net_common.hpp -- vector types and buffer-descriptor helpers shared by the PVNet kernels (included by net.hip).
#pragma once

namespace ao {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct Frag {
    float v[4];
};

__device__ __forceinline__ Frag ld_frag(const float4* p) {
    const float4 t = *p;
    Frag f;
    f.v[0] = t.x; f.v[1] = t.y; f.v[2] = t.z; f.v[3] = t.w;
    return f;
}

// Buffer-descriptor loads/stores: address = descriptor base (SGPRs) + per-lane 32-bit voffset +
// wave-uniform soffset (an SGPR). A step's dozens of fragment loads then share ONE address VGPR.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}

__device__ __forceinline__ Frag buf_ld_frag(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    Frag f;
    f.v[0] = __uint_as_float(t.x); f.v[1] = __uint_as_float(t.y);
    f.v[2] = __uint_as_float(t.z); f.v[3] = __uint_as_float(t.w);
    return f;
}

// XCD-aware block id remap: consecutive virtual ids (rows of one group, neighbouring groups)
// run on one XCD and share its L2 (blocks are dispatched round-robin over the 8 XCDs).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Rows handed out per simulation (engine_types.hpp, TreeParams::live): the tree kernel counts the live rows of this
// simulation's evaluation batch in one device word, the network kernels are launched for the batch's CAPACITY and read the word:
// a 16-board group without a live row has nothing to do (no host read-back in the loop). Null = every group is live.
__device__ __forceinline__ int live_groups16(const unsigned* live, unsigned row_cap, int groups) {
    if (!live) return groups;
    unsigned n = *live;   // (written by the previous kernel of the stream: a scalar load)
    n = n < row_cap ? n : row_cap;
    const int lg = static_cast<int>((n + 15u) >> 4);
    return lg < groups ? lg : groups;
}

template <int NX>
struct StepRegs {
    Frag x[NX];
    Frag w[3];
};

}  // namespace ao
