// sub64.hip -- k_sub64: frame assembly with one workgroup per SUBFRAME (kernels/sub64.inc) for frames of 3..8 independent
// channels of 4096 samples; a translation unit of its own to compile beside the k_frame64 ones.
#include "kernels/types.h"

#include <stdlib.h>

#include <type_traits>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
#include "kernels/pack.inc"
#include "kernels/sub64.inc"
}  // namespace

namespace flacgpu_k {
void launch_sub64(const Params &p, const PackParams &q, uint32_t frames, hipStream_t st) {
    // the image: a subframe never exceeds its VERBATIM form (encode.rs:2970-2979), subframe 0 carries the <= 16-byte
    // frame header; + the lead bits, the (w0 & 3) shift that aligns LDS with the output, two guard words
    const uint32_t words = ((((FN * p.bps + 8u + 32u + 128u + 31u + 31u) / 32u) + 3u + 2u + 3u) & ~3u) + 4u;   // (+ a guard group in front)
    if (p.xpose && p.channels == 6) {   // two workgroups of three waves per frame (load_lane_xpose_half6)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sub64<64, 16, 3>), dim3(frames * 2), dim3(192), 3 * (size_t)words * sizeof(uint32_t), st, p, q,
                           words);
    } else if (p.xpose) {   // 8 channels: four channels of a frame per workgroup, fetched together from the interleaved batch
        // (<= 25-bit samples: the four images are < 64 KB; they also hold the 8 KB transposing buffer: bps >= 8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sub64<64, 16, 4>), dim3(frames * p.channels / 4), dim3(256),
                           4 * (size_t)words * sizeof(uint32_t), st, p, q, words);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sub64<64, 16>), dim3(frames * p.channels), dim3(64), (size_t)words * sizeof(uint32_t), st,
                           p, q, words);
    }
    hipLaunchKernelGGL(k_sub_finish, dim3((frames + 63) / 64), dim3(64), 0, st, p, q);
}
}  // namespace flacgpu_k
