// cand_direct.hip -- the DIRECT-input instantiations of the persistent candidate kernel k_cand64p: 4096-sample stereo
// frames read from the caller's interleaved PCM (Params::inter), channel choice included; the SELF variant for
// streams without LPC.  A translation unit of its own only to compile beside cand.hip.
#include "kernels/types.h"

#include <stdlib.h>

#include <atomic>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
}  // namespace

namespace {
// resident workgroups of one instantiation on the current device (its registers and LDS image decide: the order-<= 16
// kernel with LPC fits FOUR workgroups per CU since r05 -- 116 VGPRs, an image of 40960 bytes sharp --, the others three),
// asked once per device; the cache is per device and its fill is race-free
template <class Kernel>
uint32_t resident_workgroups(Kernel kernel, std::atomic<uint32_t> (&cache)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    uint32_t g = cache[dev].load(std::memory_order_acquire);
    if (!g) {
        int n = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, WG, 0) != hipSuccess || n < 1) n = 3;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        g = (uint32_t)cus * (uint32_t)(n > 8 ? 8 : n);
        cache[dev].store(g, std::memory_order_release);   // (two threads may both compute it: same value)
    }
    return g;
}
}  // namespace

namespace flacgpu_k {
bool launch_cand64_direct(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st) {
    const bool lpc = p.max_lpc_order > 0;
    if (B == FN) {
        static std::atomic<uint32_t> res_self[64], res_deep[64], res_lpc[64];
        const uint32_t resident = !lpc ? resident_workgroups(k_cand64p<64, 16, true, true, true>, res_self)
                                  : p.max_lpc_order > 16 ? resident_workgroups(k_cand64p<64, 32, true, true>, res_deep)
                                                         : resident_workgroups(k_cand64p<64, 16, true, true>, res_lpc);
        const uint32_t cap = kn.cand_grid ? kn.cand_grid : resident;
        const uint32_t grid = blocks < cap ? blocks : cap;   // every resident slot of the chip, and no more (a persistent kernel)
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
