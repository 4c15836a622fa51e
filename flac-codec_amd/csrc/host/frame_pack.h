// frame_pack.h -- host-side frame assembly: what stays on the CPU in the reference's design
// (north_star): frame header + CRC-8 (stream.rs:185-276), subframe headers (stream.rs:1375-1413),
// residual partition headers (stream.rs:1603-1619), Rice bit emission (encode.rs:3834-3863),
// byte alignment + CRC-16 (encode.rs:2408-2409).  Input: the decision records and residual
// rows produced by the GPU analysis (include/flacenc_gpu.h).
#pragma once
#include <cstddef>
#include <cstdint>

#include "flacenc_gpu.h"

namespace flacenc {

struct FrameParams {
    uint32_t sample_rate;
    uint32_t bits_per_sample;  // stream bps
    uint32_t channels;
    uint64_t frame_number;
};

// size in bytes of the frame header (incl. CRC-8) for the given parameters
size_t frame_header_size(const FrameParams &fp, uint32_t block_size);
// exact size of the whole frame: header + ceil(body_bits / 8) + 2
size_t frame_size(const FrameParams &fp, const flacgpu_frame_plan &plan);

// Writes the frame into dst[0..frame_size).  `rows` points at this frame's residual rows
// ([channels][row_stride] int32).  Returns bytes written, or 0 on an internal inconsistency
// (bit count of a subframe differs from its decision record).
size_t pack_frame(const FrameParams &fp, const flacgpu_frame_plan &plan,
                  const flacgpu_subframe_plan *subs, const int32_t *rows, size_t row_stride,
                  uint8_t *dst, size_t cap);

}  // namespace flacenc
