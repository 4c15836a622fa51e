// cand.hip -- K2+K5 wave kernel k_cand64: FIXED + LPC + Rice search + choice of one candidate per wave.
// One of the translation units of libflacenc_amd.so (gfx950 only; built with -ffp-contract=off, see
// Makefile); the kernels are reached through the launchers declared in kernels/types.h.
#include "kernels/types.h"

#include <stdlib.h>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
}  // namespace

namespace flacgpu_k {
bool launch_cand64(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st) {
    // persistent variant with LDS prefetch: 4096-sample blocks, order <= 16, enough groups to go round
    const bool no_persist = kn.no_persist;
    if (p.inter) {   // interleaved stereo input read in place: persistent kernels only (cand_direct.hip)
        return launch_cand64_direct(p, kn, B, blocks, st);
    }
    if (!no_persist && B == FN && p.max_lpc_order > 16) {
        const uint32_t cap = kn.cand_grid ? kn.cand_grid : 512u;
        const uint32_t grid = blocks < cap ? blocks : cap;
        const bool stereo = p.stereo4 && p.ncand == 4;
        if (stereo) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 32, true>), dim3(grid), dim3(WG), 0, st, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 32, false>), dim3(grid), dim3(WG), 0, st, p);
        return stereo;
    }
    // independent channels (mono, 3..8 channels, wide stereo): the persistent kernel's four 16 KB rows per workgroup
    // allow two workgroups per CU; k_cand64<64, 16, false> (152 VGPRs, no LDS) runs three waves per SIMD and is faster
    // (config 4: 0.50 -> 0.33 ms per batch).  FLACGPU_CAND_PERSIST_N=1 brings the persistent one back for A/B runs.
    const bool stereo_cands = p.stereo4 && p.ncand == 4;
    if (!no_persist && B == FN && p.max_lpc_order <= 16 && (stereo_cands || kn.cand_persist_n)) {
        const bool stereo = p.stereo4 && p.ncand == 4;
        // default: three workgroups per CU for the stereo kernel (165 VGPRs, 32 KB of LDS), two for independent
        // channels (four 16 KB rows per workgroup)
        const uint32_t cap = kn.cand_grid ? kn.cand_grid : (stereo ? 768u : 512u);
        const uint32_t grid = blocks < cap ? blocks : cap;
        if (stereo) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 16, true>), dim3(grid), dim3(WG), 0, st, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<64, 16, false>), dim3(grid), dim3(WG), 0, st, p);
        return stereo;
    }
    if (p.max_lpc_order > 16) {  // orders 17..32: 4096-sample blocks only
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<64, 32>), dim3(blocks), dim3(WG), 0, st, p);
        return false;
    }
    if (B == FN && !p.stereo4) {   // independent channels: the instantiation without mid / side
        if (p.xpose) {   // 3, 4 / 8 channels read in place from the interleaved batch (load_lane_xpose): CW waves per workgroup
            const uint32_t cands = p.fcount * p.ncand;
            if (p.channels == 3 || p.channels == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<64, 16, false, 3>), dim3(cands / 3), dim3(192), 0, st, p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<64, 16, false, 4>), dim3(cands / 4), dim3(WG), 0, st, p);
        }
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<64, 16, false>), dim3(blocks), dim3(WG), 0, st, p);
        return false;
    }
    switch (B) {
#define X(n, spl) case n: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<spl, 16>), dim3(blocks), dim3(WG), 0, st, p); break;
        FLACGPU_WAVE_SIZES(X)
#undef X
    default: break;
    }
    return false;
}
}  // namespace flacgpu_k
