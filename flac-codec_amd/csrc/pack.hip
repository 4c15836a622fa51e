// pack.hip -- K8 k_layout, generic K9+K10 k_zero + k_pack + k_crc, and the dispatch of the wave-per-subframe
// k_frame64 (instantiated in frame64_a/b/c.hip): stream.rs:242-276, 1390-1413, 1603-1619; encode.rs:3078-3135,
// 3834-3907, 2408-2409; crc.rs:99-188.
// One of the translation units of libflacenc_amd.so (gfx950 only; built with -ffp-contract=off, see
// Makefile); the kernels are reached through the launchers declared in kernels/types.h.
#include <algorithm>
#include "kernels/types.h"

#include <stdlib.h>

#include <type_traits>

namespace {
#include "kernels/common.inc"
#include "kernels/wave_cand.inc"
#include "kernels/pack.inc"

}  // namespace

namespace flacgpu_k {
void launch_frame64(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds, hipStream_t st) {
    if (q.edges && B == FN && p.channels >= 5 && !p.stereo4 && p.max_lpc_order <= 16) launch_sub64(p, q, frames, st);   // one workgroup per subframe (3, 4 channels: k_frame64 measured faster)
    else if (p.inter) launch_frame64_direct(p, q, B, frames, lds, st);        // 4096-sample stereo blocks read in place
    else if (p.max_lpc_order > 16) launch_frame64_deep(p, q, frames, lds, st);   // 4096-sample blocks, <= 4 channels
    else if (B == FN) launch_frame64_4096(p, q, frames, lds, st);
    else launch_frame64_short(p, q, B, frames, lds, st);                   // <= 4 channels
}
void launch_layout(const Params &p, const PackParams &q, hipStream_t st) {
    // <= 256 tiles per launch (all of them resident at once: the look-back needs no dispatch order); more frames
    // continue from frame_off[f0] in the next launch of the same stream, with an epoch of their own (the context hands
    // out epochs in steps of 8: up to 2 M frames per call)
    constexpr uint32_t kMaxTiles = 256;
    Params r = p;
    PackParams e = q;
    // (flacgpu_create refuses max_frames beyond 8 chunks, so the epochs stay inside this call's range)
    for (uint32_t done = 0; done < p.fcount || done == 0; done += kMaxTiles * 1024u) {
        r.f0 = p.f0 + done;
        r.fcount = std::min(p.fcount - done, kMaxTiles * 1024u);
        hipLaunchKernelGGL(k_layout, dim3((r.fcount + 1023) / 1024), dim3(1024), 0, st, r, e);
        e.epoch++;
        if (p.fcount == 0) break;
    }
}
void launch_zero(const PackParams &q, uint32_t n_frames, hipStream_t st) {
    hipLaunchKernelGGL(k_zero, dim3(2048), dim3(WG), 0, st, q, n_frames);
}
void launch_pack(const Params &p, const PackParams &q, uint32_t blocks, size_t lds, hipStream_t st) {
    if (p.block_size > LDS_BLOCK_LIMIT) hipLaunchKernelGGL(k_pack_t<true>, dim3(blocks), dim3(WG), 0, st, p, q);
    else hipLaunchKernelGGL(k_pack_t<false>, dim3(blocks), dim3(WG), lds, st, p, q);
}
void launch_crc(bool verify, const Params &p, const PackParams &q, uint32_t frames, uint32_t *verify_counts,
                hipStream_t st) {
    if (verify) hipLaunchKernelGGL(k_crc<true>, dim3(frames), dim3(WG), 0, st, p, q, verify_counts);
    else hipLaunchKernelGGL(k_crc<false>, dim3(frames), dim3(WG), 0, st, p, q, verify_counts);
}
hipError_t pack_set_attributes(size_t pack_lds) {
    return hipFuncSetAttribute((const void *)k_pack_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pack_lds);
}
}  // namespace flacgpu_k
