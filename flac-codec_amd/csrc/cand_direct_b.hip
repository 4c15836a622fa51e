// cand_direct_b.hip -- the DIRECT-input instantiations of the persistent candidate kernel k_cand64p for the shorter
// wave block lengths (1024 / 1152 / 2048 / 2304 samples: Options::fast() and friends, encode.rs:1635-1644), with LPC
// (order <= 16) and without (SELF).  A translation unit of its own only to compile beside cand_direct.hip.
#include "kernels/types.h"

#include <stdlib.h>

#include <atomic>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"

// resident workgroups of one instantiation on the current device (registers and LDS image differ per block length;
// CU count from the device), asked once per device -- the cache is per device and its fill is race-free
template <int SPL, bool SELF, bool PAIR>
uint32_t resident_grid() {
    static std::atomic<uint32_t> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    uint32_t g = cache[dev].load(std::memory_order_acquire);
    if (!g) {
        int n = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cand64p<SPL, 16, true, true, SELF, PAIR>, PAIR ? 128 : WG, 0) !=
                hipSuccess || n < 1)
            n = 3;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        g = (uint32_t)cus * (uint32_t)(n > 16 ? 16 : n);
        cache[dev].store(g, std::memory_order_release);   // (two threads may both compute it: same value)
    }
    return g;
}
template <int SPL>
void launch_spl(const Params &p, const Knobs &kn, uint32_t blocks, hipStream_t st) {
    const bool lpc = p.max_lpc_order > 0;
    // without LPC and with the fast channel choice only two subframes per frame are analysed: two waves per frame
    const bool pair = !lpc && !p.exhaustive && !kn.no_cand_pair;
    const uint32_t resident = pair ? resident_grid<SPL, true, true>() : lpc ? resident_grid<SPL, false, false>()
                                                                            : resident_grid<SPL, true, false>();
    const uint32_t cap = kn.cand_grid ? kn.cand_grid : resident;
    const uint32_t units = pair ? p.fcount : blocks;   // workgroup turns: frames
    const uint32_t grid = units < cap ? units : cap;
    if (pair) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<SPL, 16, true, true, true, true>), dim3(grid), dim3(128), 0, st, p);
    else if (!lpc) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<SPL, 16, true, true, true>), dim3(grid), dim3(WG), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64p<SPL, 16, true, true>), dim3(grid), dim3(WG), 0, st, p);
}
}  // namespace

namespace flacgpu_k {
bool launch_cand64_direct_short(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st) {
    switch (B) {
    case 2304: launch_spl<36>(p, kn, blocks, st); return true;
    case 2048: launch_spl<32>(p, kn, blocks, st); return true;
    case 1152: launch_spl<18>(p, kn, blocks, st); return true;
    case 1024: launch_spl<16>(p, kn, blocks, st); return true;
    default: return false;
    }
}
}  // namespace flacgpu_k
