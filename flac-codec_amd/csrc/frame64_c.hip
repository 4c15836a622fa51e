// frame64_c.hip -- k_frame64 instantiations: frames of 2304 / 2048 / 1152 / 1024 samples, LPC order <= 16, 1..4 channels (the family is split over three
// translation units only to compile in parallel; see pack.hip).
#include "kernels/types.h"

#include <stdlib.h>

#include <type_traits>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
#include "kernels/pack.inc"
#include "kernels/frame64_launch.inc"
template <int SPL>
void launch_frame64_spl(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st) {
    switch (p.channels) {
    case 1: launch_frame64_nt<64, SPL>(p, q, frames, lds, st); break;
    case 2: launch_frame64_nt<128, SPL>(p, q, frames, lds, st); break;
    case 3: launch_frame64_nt<192, SPL>(p, q, frames, lds, st); break;
    default: launch_frame64_nt<256, SPL>(p, q, frames, lds, st); break;
    }
}
}  // namespace

namespace flacgpu_k {
void launch_frame64_short(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds,
                          hipStream_t st) {
    switch (B) {
    case 2304: launch_frame64_spl<36>(p, q, frames, lds, st); break;
    case 2048: launch_frame64_spl<32>(p, q, frames, lds, st); break;
    case 1152: launch_frame64_spl<18>(p, q, frames, lds, st); break;
    case 1024: launch_frame64_spl<16>(p, q, frames, lds, st); break;
    default: break;
    }
}
}  // namespace flacgpu_k
