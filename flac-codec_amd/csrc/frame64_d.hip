// frame64_d.hip -- k_frame64 instantiations for DIRECT input: stereo frames of the wave block lengths assembled from
// the caller's interleaved PCM (Params::inter); LPC order <= 16, and <= 32 for 4096-sample blocks (see pack.hip).
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
    // the wave block lengths (flacenc_gpu.hip: direct_input_ok); LPC order > 16 with 4096-sample blocks only
    switch (B) {
    case 2304: launch_frame64_nt<128, 36, 16, true>(p, q, frames, lds, st); break;
    case 2048: launch_frame64_nt<128, 32, 16, true>(p, q, frames, lds, st); break;
    case 1152: launch_frame64_nt<128, 18, 16, true>(p, q, frames, lds, st); break;
    case 1024: launch_frame64_nt<128, 16, 16, true>(p, q, frames, lds, st); break;
    default:
        if (p.max_lpc_order > 16) launch_frame64_nt<128, 64, 32, true>(p, q, frames, lds, st);
        else launch_frame64_nt<128, 64, 16, true>(p, q, frames, lds, st);
        break;
    }
}
}  // namespace flacgpu_k
