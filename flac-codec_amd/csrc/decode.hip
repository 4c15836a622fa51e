// decode.hip -- device-side frame decoder + verifier (decode.rs:1388-1856), one lane per subframe.
// One of the translation units of libflacenc_amd.so (gfx950 only; built with -ffp-contract=off, see
// Makefile); the kernels are reached through the launchers declared in kernels/types.h.
#include "kernels/types.h"

#include <stdlib.h>

namespace {
#include "kernels/common.inc"
#include "kernels/decode.inc"
}  // namespace

namespace flacgpu_k {
// One lane per subframe.  Every lane reads and writes its own cache lines, so the limit is the
// CU's address path (lines per instruction x waves per CU), not the SIMD: measured on 4096 / 8192
// / 32768 stereo frames, 32-lane waves win over 64 (fewer lines per instruction) and over 16 or 8
// (fewer waves per CU): 1.17 / 1.44 / 2.93 ms.
void launch_decode(uint32_t mo, uint32_t units, uint32_t lanes, const Params &pd, const PackParams &q,
                   int32_t *decoded, uint32_t *verify_counts, hipStream_t st) {
    const dim3 grid((units + lanes - 1) / lanes), block(lanes);
    // FIXED needs 4; the ring is also the store batch
    if (mo <= 8) hipLaunchKernelGGL(k_decode<8>, grid, block, 0, st, pd, q, decoded, verify_counts);
    else if (mo <= 12) hipLaunchKernelGGL(k_decode<12>, grid, block, 0, st, pd, q, decoded, verify_counts);
    else if (mo <= 16) hipLaunchKernelGGL(k_decode<16>, grid, block, 0, st, pd, q, decoded, verify_counts);
    else hipLaunchKernelGGL(k_decode<32>, grid, block, 0, st, pd, q, decoded, verify_counts);
}
void launch_decode_finish(const Params &p, int32_t *decoded, const int32_t *expect, uint32_t *verify_counts,
                          hipStream_t st, const uint32_t *frame_n) {
    hipLaunchKernelGGL(k_decode_finish, dim3(p.n_frames), dim3(WG), 0, st, p, decoded, expect, verify_counts,
                       frame_n);
}
// stand-alone decode: one lane per frame, 32-lane waves (see launch_decode), any LPC order
void launch_decode_frames(const uint32_t *words, const uint64_t *frame_off, const uint32_t *frame_n,
                          uint64_t cap_bytes, uint32_t n_frames, uint32_t channels, uint32_t bps, uint32_t ldb,
                          int32_t *decoded, uint32_t *verify_counts, hipStream_t st) {
    DecodeParams dp{words, frame_off, frame_n, cap_bytes, n_frames, channels, bps, ldb};
    const uint32_t lanes = 32;
    hipLaunchKernelGGL(k_decode_frames<32>, dim3((n_frames + lanes - 1) / lanes), dim3(lanes), 0, st, dp, decoded,
                       verify_counts);
}
}  // namespace flacgpu_k
