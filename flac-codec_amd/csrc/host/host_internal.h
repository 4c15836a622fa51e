// host_internal.h -- what the host driver's translation units share besides the public C ABI (stream_writer.cpp defines them).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

#include "flacenc_gpu.h"
#include "flacenc_stream.h"

namespace flacenc_host {

double now_ms();
int options_error(const flacenc_options &o);                                   // Options' own ranges (encode.rs:1418-1455)
flacgpu_options gpu_options(const flacenc_options &o, uint32_t block_size);   // EncoderOptions (encode.rs:1701-1709)
// interleaved int32 samples -> update_md5's byte string: little-endian samples of `width` bytes (encode.rs:1292-1318)
void pack_le(const int32_t *s, size_t count, unsigned width, uint8_t *d);
// the checks of FlacSampleWriter::new / Encoder::new for a stream whose total is known (encode.rs:487-531, 1882-1917) and
// the number of bytes in front of its first frame (fixed at `new`: the SEEKTABLE placeholder has its final size)
int stream_header_len(const flacenc_options &o, uint32_t sample_rate, uint32_t bits_per_sample, uint32_t channels,
                      uint64_t total_pcm_frames, size_t *len);
// `helpers` parked threads of the process-wide pool run fn() side by side with the caller; returns when all are done
void run_parallel(unsigned helpers, const std::function<void()> &fn);
// CPUs this process may really use: the cgroup's CPU quota when there is one (coalesce.cpp), else the hardware threads
unsigned usable_cpus();
// idle analysis lanes of the per-stream writers (flacenc_release_pools)
void release_lane_pool();

}  // namespace flacenc_host
