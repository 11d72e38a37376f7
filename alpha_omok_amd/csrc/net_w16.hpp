// net_w16.hpp -- launchers of the TWO-product split-fp16 kernels (net_w16.hip; see TrunkHLayerFn, W16, in net_trunk_h16.hpp).
// The instantiations live in their own translation unit so that they compile beside net.hip instead of after it; the
// launchers take the argument blocks net.hip has already filled in and return the HIP status of the attribute call / launch.
#pragma once

#include <hip/hip_runtime.h>

namespace ao {

struct TrunkHArgs;
struct LayerHArgs;
struct BoardHArgs;

// resident trunk (k_trunk16h_w16 / k_trunk16hb_w16<B, 4, 0>), boards 3 .. 9; in_kind 2 = bit planes
hipError_t launch_trunk16h_w16(int device, int B, int in_kind, int groups, hipStream_t s, const TrunkHArgs& a);
// one trunk conv per launch (k_layer16h_w16<B, XT, 4, 0>), boards 3 .. 15; xt: the column tile net.hip planned (boards > 9: 5 or 4)
hipError_t launch_layer16h_w16(int device, int B, int xt, dim3 grid, hipStream_t s, const LayerHArgs& a);
// one trunk conv per launch for medium batches (k_layer16hk_w16<B, 4>: a group split over four workgroups by cout pairs) and small
// ones (k_row16hk_w16<B>: one workgroup per group x output row x cout pair), boards 4 .. 9 -- net_layer_ksplit.hpp
hipError_t launch_layer16hk_w16(int device, int B, dim3 grid, hipStream_t s, const LayerHArgs& a);
hipError_t launch_row16hk_w16(int device, int B, dim3 grid, hipStream_t s, const LayerHArgs& a);
// board-resident trunk (k_boardh_w16<B, 1 | 2>), boards 10 .. 15
hipError_t launch_boardh_w16(int device, int B, bool bits, dim3 grid, hipStream_t s, const BoardHArgs& a);

}  // namespace ao
