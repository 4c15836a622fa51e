// lpc.hip -- K4: Levinson-Durbin, order choice, quantisation (encode.rs:3536-3580, 3656-3702, 3334-3401).
// One of the translation units of libflacenc_amd.so (gfx950 only; built with -ffp-contract=off, see
// Makefile); the kernels are reached through the launchers declared in kernels/types.h.
#include "kernels/types.h"

#include <stdlib.h>

namespace {
#include "kernels/common.inc"
#include "kernels/lpc.inc"
}  // namespace

namespace flacgpu_k {
void launch_lpc(const Params &p, const Knobs &kn, uint32_t blocks, hipStream_t st) {
    if (p.max_lpc_order <= 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<8>), dim3(blocks), dim3(64), 0, st, p);
    else if (p.max_lpc_order <= 12) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<12>), dim3(blocks), dim3(64), 0, st, p);
    else if (p.max_lpc_order <= 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<16>), dim3(blocks), dim3(64), 0, st, p);
    else if (kn.lpc_dyn) hipLaunchKernelGGL(k_lpc, dim3(blocks), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<32>), dim3(blocks), dim3(64), 0, st, p);
}
}  // namespace flacgpu_k
