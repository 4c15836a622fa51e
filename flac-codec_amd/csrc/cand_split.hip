// cand_split.hip -- k_cand64s: the persistent candidate kernel with eight waves per stereo frame ({L, R, mid, side} x
// {FIXED, LPC}, kernels/wave_cand_split.inc), 4096-sample blocks with LPC.  A translation unit of its own to compile
// beside cand.hip / cand_direct.hip.
#include "kernels/types.h"

#include <stdlib.h>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
#include "kernels/wave_cand_split.inc"
}  // namespace

namespace flacgpu_k {
bool launch_cand64_split(const Params &p, uint32_t B, uint32_t frames, uint32_t grid_cap, hipStream_t st) {
    if (B != FN || !(p.stereo4 && p.ncand == 4) || p.max_lpc_order == 0) return false;
    const uint32_t grid = frames < grid_cap ? frames : grid_cap;   // default: two 8-wave workgroups per CU
    if (p.inter) {
        if (p.max_lpc_order > 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64s<64, 32, true>), dim3(grid), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64s<64, 16, true>), dim3(grid), dim3(512), 0, st, p);
    } else {
        if (p.max_lpc_order > 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64s<64, 32, false>), dim3(grid), dim3(512), 0, st, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64s<64, 16, false>), dim3(grid), dim3(512), 0, st, p);
    }
    return true;
}
}  // namespace flacgpu_k
