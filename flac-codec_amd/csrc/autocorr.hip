// autocorr.hip -- K3: window + exact autocorrelation (encode.rs:1785-1801, 3478-3501) and the MFMA experiment.
// One of the translation units of libflacenc_amd.so (gfx950 only; built with -ffp-contract=off, see
// Makefile); the kernels are reached through the launchers declared in kernels/types.h.
#include "kernels/types.h"

#include <stdlib.h>

#include <type_traits>

namespace {
#include "kernels/common.inc"
#include "kernels/lpc.inc"        // lpc_candidate: the Levinson tail of the DIRECT k_autocorr4 instantiations
#include "kernels/autocorr.inc"

// returns true when the launched kernel also ran K4 (Levinson, order choice, quantisation, candidate info) in its tail
template <int NL, bool STEREO>
bool launch_autocorr3(const Params &p, const Knobs &kn, uint32_t frame0, uint32_t nframes, uint32_t n,
                      const double *win, hipStream_t st) {
    const uint32_t groups = (nframes * p.ncand + 63) / 64;
    // 4 waves per 64 candidates (lags split 4 ways) by default: the f64 stream needs two waves per
    // SIMD to issue at full rate and 8192 frames are only 512 candidate groups (0.23 ms against
    // 0.31 ms split 2 ways).  When other contexts keep the SIMDs busy anyway, the 2-way split wins:
    // the int -> f64 x window conversion is replicated 2x instead of 4x (71 M instead of 92 M
    // instructions).
    const bool private_tiles = kn.ac_private;  // previous kernel (A/B runs)
    if constexpr (STEREO) {
        if (p.inter) {   // interleaved input read in place (the host selects this only with the 4-way split)
            // small batches (<= 1024 frames: at most 64 workgroups for 256 CUs) are a latency problem -- the serial walk
            // over a frame's samples -- and take the eight-wave split of the lags, whose walk is shorter (512 frames:
            // 0.0555 -> 0.0441 ms per batch in the four-context loop, profiles/r04_batch_sweep.json); large batches are a
            // throughput problem and keep four waves (the eight-wave kernel reads the tile twice as often)
            const bool fuse = !kn.no_lpc_fuse;
            if ((kn.ac_eight_waves || groups <= 64) && NL == 13) {
                if (fuse) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<13, 8, true, true, 0, true>), dim3(groups), dim3(512), 0, st, p,
                                             frame0, nframes, n, win);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<13, 8, true, true>), dim3(groups), dim3(512), 0, st, p,
                                        frame0, nframes, n, win);
            } else {
                if (fuse) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<NL, 4, true, true, 0, true>), dim3(groups), dim3(256), 0, st, p,
                                             frame0, nframes, n, win);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<NL, 4, true, true>), dim3(groups), dim3(256), 0, st, p,
                                        frame0, nframes, n, win);
            }
            return fuse;
        }
    }
    if constexpr (!STEREO) {
        if (p.split_src) {   // interleaved independent channels: the producers split them on the way (no K0 pass)
            const bool fuse = !kn.no_lpc_fuse;
#define AC4_SPLIT(C)                                                                                                      \
    do {                                                                                                                  \
        if (fuse) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<NL, 4, false, true, C, true>), dim3(groups), dim3(256), 0, st, p, \
                                     frame0, nframes, n, win);                                                            \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<NL, 4, false, true, C>), dim3(groups), dim3(256), 0, st, p,   \
                                frame0, nframes, n, win);                                                                 \
    } while (0)
            if (p.channels == 8) AC4_SPLIT(8);
            else if (p.channels == 4) AC4_SPLIT(4);
            else AC4_SPLIT(0);
#undef AC4_SPLIT
            return fuse;
        }
    }
    if (!private_tiles) {  // shared conversion through LDS
        if (p.ac_split == 2)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<NL, 2, STEREO>), dim3(groups), dim3(128), 0, st, p,
                               frame0, nframes, n, win);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4<NL, 4, STEREO>), dim3(groups), dim3(256), 0, st, p,
                               frame0, nframes, n, win);
        return false;
    }
    if (p.ac_split == 2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3<NL, 2, STEREO>), dim3(groups), dim3(128), 0, st, p,
                           frame0, nframes, n, win);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3<NL, 4, STEREO>), dim3(groups), dim3(256), 0, st, p,
                           frame0, nframes, n, win);
    return false;
}
template <bool STEREO>
bool launch_autocorr3_nl(const Params &p, const Knobs &kn, uint32_t frame0, uint32_t nframes, uint32_t n, const double *win,
                         hipStream_t st) {
    const uint32_t nl = p.max_lpc_order + 1;
    if (nl <= 5) return launch_autocorr3<5, STEREO>(p, kn, frame0, nframes, n, win, st);
    if (nl <= 9) return launch_autocorr3<9, STEREO>(p, kn, frame0, nframes, n, win, st);
    if (nl <= 13) return launch_autocorr3<13, STEREO>(p, kn, frame0, nframes, n, win, st);
    return launch_autocorr3<17, STEREO>(p, kn, frame0, nframes, n, win, st);
}
// frame length a multiple of 32, order <= 16, and either stereo L/R/M/S candidates of <= 24-bit
// samples (mid/side formed with one v_mad_i32_i24) or independent channels of any width
// *fused: K4 ran in the kernel's tail (launch_autocorr3)
bool try_autocorr3(const Params &p, const Knobs &kn, uint32_t frame0, uint32_t nframes, uint32_t n, const double *win,
                   hipStream_t st, bool *fused) {
    *fused = false;
    if (n < 32 || n % 32 != 0 || kn.no_ac3) return false;
    const bool stereo = p.stereo4 && p.ncand == 4 && p.channels == 2 && p.bps <= 24;
    const bool indep = !p.stereo4 && p.ncand == p.channels;
    if (!stereo && !indep) return false;
    if (p.max_lpc_order > 16) {  // lags up to 32: two blocks of history, frame a multiple of 64
        if (n % 64 != 0) return false;
        const uint32_t groups = (nframes * p.ncand + 63) / 64;
        const bool private_deep = kn.ac_private;
        if (stereo && p.inter) {
            // (K4 in the tail of the deep kernel: measured and left off -- 228 instead of 153 VGPRs leave no room for another
            // kernel's wave beside two of these, and the four-context step of config 5 goes 0.566 -> 0.588 ms;
            // FLACGPU_LPC_FUSE_DEEP=1 for A/B runs, profiles/r04_autocorr_lpc_fuse.json)
            if (kn.lpc_fuse_deep && !kn.no_lpc_fuse) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4_deep<true, true, true>), dim3(groups), dim3(256), 0, st, p, frame0,
                                   nframes, n, win);
                *fused = true;
            } else {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4_deep<true, true>), dim3(groups), dim3(256), 0, st, p, frame0,
                                   nframes, n, win);
            }
        } else if (private_deep) {
            if (stereo)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3_deep<true>), dim3(groups), dim3(256), 0, st, p, frame0,
                                   nframes, n, win);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3_deep<false>), dim3(groups), dim3(256), 0, st, p, frame0,
                                   nframes, n, win);
        } else if (stereo) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4_deep<true>), dim3(groups), dim3(256), 0, st, p, frame0,
                               nframes, n, win);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr4_deep<false>), dim3(groups), dim3(256), 0, st, p, frame0,
                               nframes, n, win);
        }
        return true;
    }
    if (stereo) *fused = launch_autocorr3_nl<true>(p, kn, frame0, nframes, n, win, st);
    else *fused = launch_autocorr3_nl<false>(p, kn, frame0, nframes, n, win, st);
    return true;
}

template <int H>
void launch_autocorr(const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n,
                     const double *win, hipStream_t st) {
    const uint32_t lanes = nframes * p.ncand;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr2<H, 4>), dim3((lanes + 63) / 64), dim3(WG), 0, st, p,
                       frame0, nframes, n, win);
}
}  // namespace

namespace flacgpu_k {
bool dispatch_autocorr(uint32_t H, const Params &p_in, const Knobs &kn, uint32_t frame0, uint32_t nframes, uint32_t n,
                       const double *win, hipStream_t st) {
    Params p = p_in;
    if (n != p.block_size) p.fma_t0 = p.fma_t1 = 0;   // a short last frame has a window of its own: no fused tiles
    // EXPERIMENT switch (bench.py --experiment mfma_autocorr): the re-associating f64-MFMA kernel in
    // place of the exact one, to measure what a whole step costs with the autocorrelation off the
    // VALU pipe.  NOT bit-exact; never set in production.
    const bool mfma = kn.experiment_mfma_ac;
    if (mfma && p.max_lpc_order >= 1 && p.max_lpc_order <= 16 && frame0 == 0 && nframes == p.n_frames &&
        n == p.block_size) {
        hipLaunchKernelGGL(k_autocorr_mfma, dim3((nframes * p.ncand + 3) / 4), dim3(WG), 0, st, p, n, win, p.ac);
        return false;
    }
    bool fused = false;
    if (try_autocorr3(p, kn, frame0, nframes, n, win, st, &fused)) return fused;
    switch (H) {
    case 4: launch_autocorr<4>(p, frame0, nframes, n, win, st); break;
    case 8: launch_autocorr<8>(p, frame0, nframes, n, win, st); break;
    case 12: launch_autocorr<12>(p, frame0, nframes, n, win, st); break;
    case 16: launch_autocorr<16>(p, frame0, nframes, n, win, st); break;
    case 20: launch_autocorr<20>(p, frame0, nframes, n, win, st); break;
    case 24: launch_autocorr<24>(p, frame0, nframes, n, win, st); break;
    case 28: launch_autocorr<28>(p, frame0, nframes, n, win, st); break;
    case 32: launch_autocorr<32>(p, frame0, nframes, n, win, st); break;
    default: launch_autocorr<36>(p, frame0, nframes, n, win, st); break;
    }
    return false;
}
void launch_autocorr_mfma(const Params &p, uint32_t blocks, uint32_t n, const double *win, double *ac,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_autocorr_mfma, dim3(blocks), dim3(WG), 0, st, p, n, win, ac);
}
}  // namespace flacgpu_k
