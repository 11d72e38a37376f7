// net_w16.hip -- the two-product instantiations of the split-fp16 conv kernels and their launchers (net_w16.hpp).
//
// x * w = xh*wh + xh*wl + xl*wh is the fp32-equivalent contraction of the split-fp16 kernels (net_trunk_h16.hpp). When every conv
// weight of a network, scaled by its layer's power of two, IS an fp16 number, wl == 0 and the middle product adds exact zeros:
// these kernels leave it out. Same device code as the three-product kernels (a template flag), same bits on such weights, a third
// fewer MFMAs. The reference's network is model.py:6-31,76-104 (PVNet, ResBlock); which networks qualify is decided by
// ao_net_finalize (net.hip), never by the caller.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/omok_hip.h"
#include "engine_types.hpp"
#include "net_device.hpp"
#include "net_common.hpp"
#include "net_trunk_f32.hpp"
#include "net_trunk_h16.hpp"
#include "net_layer_ksplit.hpp"
#include "net_board_h16.hpp"
#include "net_w16.hpp"

namespace ao {

namespace {
constexpr int kDev = 16;
// dynamic-LDS attribute set once per (device, kernel)
bool g_attr_trunk[kDev][16][2], g_attr_layer[kDev][16][2], g_attr_board[kDev][16][2], g_attr_ksplit[kDev][16][2];

template <typename K>
hipError_t set_lds(bool* done, K kernel, size_t lds) {
    if (*done) return hipSuccess;
    const hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (st == hipSuccess) *done = true;
    return st;
}
}  // namespace

hipError_t launch_trunk16h_w16(int device, int B, int in_kind, int groups, hipStream_t s, const TrunkHArgs& a) {
    if (device < 0 || device >= kDev) return hipErrorInvalidDevice;
    const int k = in_kind == 2 ? 1 : 0;
    switch (B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t heads_ = (static_cast<size_t>(3) * 128 + 16 * 3 * W * W + 8 * 16 * W * W + 8 * 16 * 128) * 4; \
        constexpr size_t rows_ = static_cast<size_t>(2) * W * 4 * 2 * 1024 + 64;                                       \
        constexpr size_t lds_ = (rows_ > heads_) ? rows_ : heads_;                                                     \
        hipError_t st = k ? set_lds(&g_attr_trunk[device][W][1], &k_trunk16hb_w16<W, 4, 0>, lds_)                      \
                          : set_lds(&g_attr_trunk[device][W][0], &k_trunk16h_w16<W, 4, 0>, lds_);                      \
        if (st != hipSuccess) return st;                                                                               \
        if (k) hipLaunchKernelGGL((k_trunk16hb_w16<W, 4, 0>), dim3(groups), dim3(512), lds_, s, a);                    \
        else hipLaunchKernelGGL((k_trunk16h_w16<W, 4, 0>), dim3(groups), dim3(512), lds_, s, a);                       \
    } break;
        AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
#undef AO_BW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_layer16h_w16(int device, int B, int xt, dim3 grid, hipStream_t s, const LayerHArgs& a) {
    if (device < 0 || device >= kDev) return hipErrorInvalidDevice;
#define AO_LAYERH_LAUNCH(W, XT_)                                                                                       \
    do {                                                                                                               \
        constexpr int NX_ = (XT_ < W) ? XT_ + 2 : XT_;                                                                 \
        constexpr size_t lds_ = static_cast<size_t>(2) * NX_ * 4 * 2 * 1024;                                           \
        const hipError_t st = set_lds(&g_attr_layer[device][W][XT_ == 4], &k_layer16h_w16<W, XT_, 4, 0>, lds_);        \
        if (st != hipSuccess) return st;                                                                               \
        hipLaunchKernelGGL((k_layer16h_w16<W, XT_, 4, 0>), grid, dim3(512), lds_, s, a);                               \
    } while (0)
    switch (B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        if (W <= 9) AO_LAYERH_LAUNCH(W, (W <= 9 ? W : 5));                                                             \
        else if (xt == 4) AO_LAYERH_LAUNCH(W, (W <= 9 ? W : 4));                                                       \
        else AO_LAYERH_LAUNCH(W, (W <= 9 ? W : 5));                                                                    \
    } break;
        AO_BW_CASE(3) AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
        AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
#undef AO_LAYERH_LAUNCH
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_layer16hk_w16(int device, int B, dim3 grid, hipStream_t s, const LayerHArgs& a) {
    if (device < 0 || device >= kDev) return hipErrorInvalidDevice;
    switch (B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t lds_ = static_cast<size_t>(2) * W * 8 * 1024;                                                 \
        const hipError_t st = set_lds(&g_attr_ksplit[device][W][0], &k_layer16hk_w16<W, 4>, lds_);                     \
        if (st != hipSuccess) return st;                                                                               \
        hipLaunchKernelGGL((k_layer16hk_w16<W, 4>), grid, dim3(512), lds_, s, a);                                      \
    } break;
        AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
#undef AO_BW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_row16hk_w16(int device, int B, dim3 grid, hipStream_t s, const LayerHArgs& a) {
    if (device < 0 || device >= kDev) return hipErrorInvalidDevice;
    switch (B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t lds_ = static_cast<size_t>(W) * 8 * 1024;                                                     \
        const hipError_t st = set_lds(&g_attr_ksplit[device][W][1], &k_row16hk_w16<W>, lds_);                          \
        if (st != hipSuccess) return st;                                                                               \
        hipLaunchKernelGGL((k_row16hk_w16<W>), grid, dim3(512), lds_, s, a);                                           \
    } break;
        AO_BW_CASE(4) AO_BW_CASE(5) AO_BW_CASE(6) AO_BW_CASE(7) AO_BW_CASE(8) AO_BW_CASE(9)
#undef AO_BW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_boardh_w16(int device, int B, bool bits, dim3 grid, hipStream_t s, const BoardHArgs& a) {
    if (device < 0 || device >= kDev) return hipErrorInvalidDevice;
    switch (B) {
#define AO_BW_CASE(W)                                                                                                  \
    case W: {                                                                                                          \
        constexpr size_t lds_ = static_cast<size_t>(W) * 8 * 1024;                                                     \
        hipError_t st = bits ? set_lds(&g_attr_board[device][W][1], &k_boardh_w16<W, 2>, lds_)                         \
                             : set_lds(&g_attr_board[device][W][0], &k_boardh_w16<W, 1>, lds_);                        \
        if (st != hipSuccess) return st;                                                                               \
        if (bits) hipLaunchKernelGGL((k_boardh_w16<W, 2>), grid, dim3(512), lds_, s, a);                               \
        else hipLaunchKernelGGL((k_boardh_w16<W, 1>), grid, dim3(512), lds_, s, a);                                    \
    } break;
        AO_BW_CASE(10) AO_BW_CASE(11) AO_BW_CASE(12) AO_BW_CASE(13) AO_BW_CASE(14) AO_BW_CASE(15)
#undef AO_BW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ao
