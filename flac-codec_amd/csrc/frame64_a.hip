// frame64_a.hip -- k_frame64 instantiations: 4096-sample frames, LPC order <= 16, 1..8 channels (the family is split over three
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
void launch_frame64_4096(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st) {
    switch (p.channels) {
    case 1: launch_frame64_nt<64, 64>(p, q, frames, lds, st); break;
    case 2: launch_frame64_nt<128, 64>(p, q, frames, lds, st); break;
    case 3:
        if (p.xpose) launch_frame64_nt<192, 64, 16, false, true>(p, q, frames, lds, st);   // the interleaved batch read in place
        else launch_frame64_nt<192, 64>(p, q, frames, lds, st);
        break;
    case 4:
        if (p.xpose) launch_frame64_nt<256, 64, 16, false, true>(p, q, frames, lds, st);   // the interleaved batch read in place
        else launch_frame64_nt<256, 64>(p, q, frames, lds, st);
        break;
    case 5: launch_frame64_nt<320, 64>(p, q, frames, lds, st); break;
    case 6: launch_frame64_nt<384, 64>(p, q, frames, lds, st); break;
    case 7: launch_frame64_nt<448, 64>(p, q, frames, lds, st); break;
    default: launch_frame64_nt<512, 64>(p, q, frames, lds, st); break;
    }
}
}  // namespace flacgpu_k
