// frame64_b.hip -- k_frame64 instantiations: 4096-sample frames, LPC orders 17..32, 1..4 channels (the family is split over three
// translation units only to compile in parallel; see pack.hip).
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
void launch_frame64_deep(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st) {
    switch (p.channels) {
    case 1: launch_frame64_nt<64, 64, 32>(p, q, frames, lds, st); break;
    case 2: launch_frame64_nt<128, 64, 32>(p, q, frames, lds, st); break;
    case 3: launch_frame64_nt<192, 64, 32>(p, q, frames, lds, st); break;
    default: launch_frame64_nt<256, 64, 32>(p, q, frames, lds, st); break;
    }
}
}  // namespace flacgpu_k
