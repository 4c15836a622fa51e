// frame64_d.hip -- k_frame64 instantiations for DIRECT input: 4096-sample stereo frames assembled from the
// caller's interleaved PCM (Params::inter), LPC order <= 16 and <= 32 (see pack.hip).
#include "kernels/types.h"

#include <stdlib.h>

#include <type_traits>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
#include "kernels/pack.inc"
#include "kernels/frame64_launch.inc"
}  // namespace

namespace flacgpu_k {
void launch_frame64_direct(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds, hipStream_t st) {
    (void)B;   // 4096-sample blocks only (flacenc_gpu.hip: direct_input_ok)
    if (p.max_lpc_order > 16) launch_frame64_nt<128, 64, 32, true>(p, q, frames, lds, st);
    else launch_frame64_nt<128, 64, 16, true>(p, q, frames, lds, st);
}
}  // namespace flacgpu_k
