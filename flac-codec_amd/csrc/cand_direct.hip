// cand_direct.hip -- the DIRECT-input instantiations of the persistent candidate kernel k_cand64p: 4096-sample stereo
// frames read from the caller's interleaved PCM (Params::inter), channel choice included; the SELF variant for
// streams without LPC.  A translation unit of its own only to compile beside cand.hip.
#include "kernels/types.h"

#include <stdlib.h>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
}  // namespace

namespace flacgpu_k {
bool launch_cand64_direct(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st) {
    const bool lpc = p.max_lpc_order > 0;
    if (B == FN) {
        const bool split = kn.cand_split;   // experiment: eight waves per frame (profiles/r03_cand_split.json)
        const uint32_t cap = kn.cand_grid ? kn.cand_grid : (split ? 512u : 768u);
        if (split && lpc) return launch_cand64_split(p, B, blocks, cap, st);
        const uint32_t grid = blocks < cap ? blocks : cap;   // default: three workgroups per CU (165 VGPRs)
        if (!lpc)   // no k_autocorr4 / k_lpc before this kernel: it derives the candidate info itself
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 16, true, true, true>), dim3(grid), dim3(WG), 0, st, p);
        else if (p.max_lpc_order > 16)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 32, true, true>), dim3(grid), dim3(WG), 0, st, p);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 16, true, true>), dim3(grid), dim3(WG), 0, st, p);
        return true;
    }
    return launch_cand64_direct_short(p, kn, B, blocks, st);   // cand_direct_b.hip
}
}  // namespace flacgpu_k
