// flacenc_gpu.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the FLAC encode hot path.
//
// Written for gfx950 only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).
// -ffp-contract=off is REQUIRED: the f64 analysis must round exactly like the
// reference (Rust never contracts a*b+c; it uses mul_add only where written).
//
// The kernels live in kernels/*.inc; the large families are compiled as translation units of their
// own (cand.hip, pack.hip, autocorr.hip, lpc.hip, decode.hip -- parallel build) and reached through
// the launchers of kernels/types.h.  This file holds the context, the small kernels and the C ABI
// (include/flacenc_gpu.h).
//
// Kernel inventory (reference function each one replaces, /root/reference/src):
//   K0    k_deinterleave2 / k_deinterleave_n / k_deinterleave + k_orbits, k_candinfo (k0_split.inc)
//           audio.rs:190-199 Frame::fill_from_samples; encode.rs:2870-2898 wasted bits / all-zero
//   K1    k_stereo_stats (generic_analysis.inc)  encode.rs:2463-2674 correlate_channels (fast mode)
//   K3    k_autocorr3 / k_autocorr3_deep / k_autocorr2 (autocorr.inc)
//           encode.rs:1785-1801 Window::apply + 3478-3501 autocorrelate (exact summation order)
//   K4    k_lpc_u / k_lpc (lpc.inc)  encode.rs:3536-3580 lp_coefficients, 3656-3702
//           compute_best_order, 3334-3401 quantize
//   K2+K5 k_cand64 (wave_cand.inc): FIXED + LPC + Rice search + choice of one candidate per wave
//           encode.rs:2849-2898, 3020-3088, 3174-3203, 3747-3962, 2929-2979;
//         generic: k_fixed (generic_analysis.inc), k_fir (generic_fir.inc), one workgroup per candidate
//   K6    k_decide (decide_emit.inc)  encode.rs:2747-2786 / 2803-2835 channel-assignment choice
//   K7    k_emit (decide_emit.inc)  residual rows of the chosen subframes (generic packing / fetch)
//   K8    k_layout (pack.inc)  frame sizes -> byte offsets
//   K9+10 k_frame64 (pack.inc): a frame assembled in LDS, wave per subframe, CRC-16, one write
//           stream.rs:242-276, 1390-1413, 1603-1619; encode.rs:3078-3135, 3834-3907, 2408-2409;
//         generic: k_zero + k_pack + k_crc
//   N3    k_decode (decode.inc)  decode.rs:1388-1856 read_frame .. predict, one lane per subframe
#include "kernels/types.h"
#include "checksums.h"
#include "lpc_host.h"

#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <mutex>
#include <thread>
#include <atomic>
#include <string>
#include <type_traits>
#include <vector>

thread_local std::string g_last_error;

// Every FLACGPU_* environment knob (kernels/types.h Knobs), read when a context is created.  The test-only knobs
// are ignored unless FLACGPU_TEST_KNOBS=1 accompanies them: a production process cannot be skewed by a stray variable.
static std::atomic<uint64_t> g_early_downloads{0}, g_early_remainders{0};
// (experiment counters: early downloads queued / of them completed by a second copy of the remainder)
extern "C" void flacgpu_early_download_counters(uint64_t *queued, uint64_t *with_remainder) {
    if (queued) *queued = g_early_downloads.load();
    if (with_remainder) *with_remainder = g_early_remainders.load();
}
// Params::defer_margin16 by default: the LPC estimate must undercut the FIXED bound by this many sixteenths of a bit per sample.
// compute_best_order's estimate (encode.rs:3675: ln(err / 2n) / 2 ln 2 = log2(sigma) - 0.5) lies 2.5-2.9 bits per sample under
// what Rice coding makes of the residual, so 3 bits defer only candidates that are then decided by the bound (measured,
// profiles/r05_defer_fixed.json: 0.6 % of the deferred candidates of the high-order input take the re-read, none of SURVEY's
// input defers at all; at 2 bits every candidate of SURVEY's input deferred and re-read: +11 % on the kernel)
static constexpr uint32_t kDeferMargin16 = 48;
static constexpr uint32_t kDeferWords = DEFER_SLOTS * DEFER_SLOT_WORDS;   // Params::defer_stats
static constexpr uint32_t kTurnWords = 4;   // Params::turn_counter words in front of the counters (16-byte aligned counters behind them)
Knobs read_knobs() {
    Knobs k;
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    k.no_direct = on("FLACGPU_NO_DIRECT");
    k.no_cand_pair = on("FLACGPU_NO_CAND_PAIR");
    k.force_fir_check = on("FLACGPU_FIR_CHECK");
    k.no_sub64 = on("FLACGPU_NO_SUB64");
    k.no_lpc_fuse = on("FLACGPU_NO_LPC_FUSE");
    k.no_xpose = on("FLACGPU_NO_XPOSE");
    k.no_chunk = on("FLACGPU_NO_CHUNK");
    k.no_ac_fma = on("FLACGPU_NO_AC_FMA");
    k.lpc_fuse_deep = on("FLACGPU_LPC_FUSE_DEEP");
    k.upload_by_kernel = on("FLACGPU_UPLOAD_KERNEL");
    k.no_hand = on("FLACGPU_NO_HAND");   // A/B: k_frame64 fetches the samples and runs the FIR again (Params::hand_meta off)
    k.no_direct_short = on("FLACGPU_NO_DIRECT_SHORT");   // A/B: the shorter wave block lengths through K0 + k_cand64
    k.no_fast = on("FLACGPU_NO_FAST");
    k.no_w64 = on("FLACGPU_NO_W64");
    k.no_persist = on("FLACGPU_NO_PERSIST");
    k.no_ac3 = on("FLACGPU_NO_AC3");
    k.ac_private = on("FLACGPU_AC_PRIVATE");
    k.no_fused_pack = on("FLACGPU_NO_FUSED_PACK");
    k.no_frame64 = on("FLACGPU_NO_FRAME64");
    k.no_fork = on("FLACGPU_NO_FORK");
    k.lpc_dyn = on("FLACGPU_LPC_DYN");
    k.ac_eight_waves = on("FLACGPU_AC_WAVES8");
    k.cand_persist_n = on("FLACGPU_CAND_PERSIST_N");
    k.early_download = on("FLACGPU_EARLY_DOWNLOAD");
    if (const char *e = getenv("FLACGPU_CAND_GRID")) k.cand_grid = (uint32_t)atoi(e);
    // A/B: Params::defer_fixed (0 never, 1 by the estimate, 2 whenever LPC parameters exist) and its margin in 1/16 bit per
    // sample -- neither changes a byte of output
    if (const char *e = getenv("FLACGPU_DEFER_FIXED")) k.defer_fixed = atoi(e);
    if (const char *e = getenv("FLACGPU_DEFER_MARGIN16")) k.defer_margin16 = atoi(e);
    const char *t = getenv("FLACGPU_TEST_KNOBS");
    if (t && t[0] == '1') {
        k.experiment_mfma_ac = on("FLACGPU_EXPERIMENT_MFMA_AC");
        if (const char *e = getenv("FLACGPU_TIE_BAND")) { k.has_tie_band = true; k.tie_band = atof(e); }
        if (const char *e = getenv("FLACGPU_TIE_PERTURB")) { k.has_tie_perturb = true; k.tie_perturb = atof(e); }
        if (const char *e = getenv("FLACGPU_DECODE_LANES")) k.decode_lanes = (uint32_t)atoi(e);
        if (const char *e = getenv("FLACGPU_FIR_SUSPECT_BITS")) k.fir_suspect_bits = (uint32_t)atoi(e);
    }
    return k;
}

namespace {
#include "kernels/common.inc"
#include "kernels/k0_split.inc"
#include "kernels/generic_analysis.inc"
#include "kernels/generic_fir.inc"
#include "kernels/decide_emit.inc"
}  // namespace
using namespace flacgpu_k;

struct flacgpu_ctx;
static uint32_t *defer_stats_of(const flacgpu_ctx *c);
static int ctx_sync(flacgpu_ctx *c);
static hipStream_t ctx_stream(flacgpu_ctx *c);

struct flacgpu_ctx {
    flacgpu_options opts;
    uint32_t bps, channels, max_frames, ldb, ncand, stereo4;
    int device;
    // device buffers
    int32_t *d_in = nullptr, *d_planar = nullptr, *d_resid = nullptr;
    double *d_window_full = nullptr, *d_window_last = nullptr, *d_log2_thr = nullptr, *d_ac = nullptr;
    CandInfo *d_cinfo = nullptr;
    SubPlan *d_fixed = nullptr, *d_cand = nullptr, *d_out = nullptr;
    LpcParams *d_lpc = nullptr;
    FrameInfo *d_finfo = nullptr;
    uint32_t *d_hand = nullptr;     // Params::hand_meta (stereo contexts of 4096-sample blocks with LPC; else nullptr)
    flacgpu_frame_plan *d_fplan = nullptr;
    uint32_t *d_stats_base = nullptr;   // Params::turn_counter words in front of d_stats (zeroed by the same memset)
    uint32_t *d_stats = nullptr;
    uint32_t *d_orbits = nullptr;   // OR of all samples per (frame, candidate); = d_stats + 4
    uint32_t *d_ties = nullptr;     // candidates whose LPC order estimates tie (k_lpc), [F * NC]
    double tie_band = 1e-9, tie_perturb = 0.0;
    Knobs knobs;                  // the FLACGPU_* environment, read once at flacgpu_create
    const uint8_t *packed_src = nullptr;   // upload by kernel (Knobs::upload_by_kernel): K0 reads the caller's pinned PCM itself
    // a batch made of SEGMENTS (flacgpu_encode_segments*): per-frame frame numbers and PCM addresses on the device, their
    // pinned staging copy, and the segments themselves (ensure_planar walks them)
    uint64_t *d_seg_fn = nullptr, *h_seg = nullptr;
    const int32_t **d_seg_ptr = nullptr;
    std::vector<flacgpu_segment> segs;
    bool seg_pending = false;   // the next analyze_impl belongs to a segments call
    bool seg_active = false;    // the batch in hand is made of segments (frame numbers from d_seg_fn)
    bool seg_direct = false;    // ... read in place (Params::inter_tab)
    bool drained = false;       // nothing of this context is in flight (its last batch was waited for, nothing submitted since):
                                // the asynchronous entries then skip their hipStreamSynchronize -- with more streams than
                                // hardware queues a synchronise of an IDLE stream still waits behind the other streams' work
    bool env_no_direct = false;     // FLACGPU_NO_DIRECT as read at creation; copy_input: FLACGPU_TUNE_COPY_INPUT.  knobs.no_direct
    bool copy_input = false;        //   is their OR
    bool ties_checked = true;       // the last analysis has been looked at by resolve_order_ties
    uint32_t ties_resolved = 0;     // candidates re-decided on the host for the last analysis
    uint32_t fir_rechecked = 0;     // candidate waves of the last analysis that asked for the checked FIR re-run
                                    // (Params::check_fir) and got it
    uint32_t last_n_fast = 0;       // frames of the last analysis that took the wave kernels
    uint64_t last_first_frame = 0;  // arguments of the last frame assembly (re-run after a re-decision)
    uint32_t last_rate = 0;
    uint32_t *h_stats = nullptr;    // pinned copy of the 4 counters (asynchronous host path)
    int32_t *d_big = nullptr;       // blocks > LDS_BLOCK_LIMIT: per-workgroup arrays of the generic kernels
    unsigned long long *d_abs = nullptr;   // stereo, fast channel choice: K0's sums of |l|, |r|, |mid|, |side| per frame
    bool abs_valid = false;         // K0 of the batch in hand filled d_abs (k_candinfo then makes the choice)
    int32_t *d_decoded = nullptr;   // [F][C][ldb] PCM decoded back from the packed frames (lazy)
    uint32_t *d_verify = nullptr;   // [4] verify counters
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_layout = nullptr;
    bool two_ranges = false;         // FLACGPU_TUNE_TWO_RANGES
    // flacgpu_encode_packed_async_host: k_frame64 writes the batch's frames straight into this pinned host
    // buffer (no d_packed, no download) when every frame of the batch takes it; out_in_host says it did
    uint8_t *host_out = nullptr;
    size_t host_out_cap = 0;
    bool out_in_host = false;
    bool blocking_wait = false;      // FLACGPU_TUNE_BLOCKING_WAIT
    int lag_split = 4;               // FLACGPU_TUNE_LAG_SPLIT
    uint32_t *d_packed = nullptr;   // packed frame bytes (as 32-bit words)
    uint64_t *d_frame_off = nullptr;
    unsigned long long *d_tile_sync = nullptr;   // k_layout's tile exchange (PackParams::tile_sync)
    PackParams::SubEdgeRec *d_edges = nullptr;   // k_sub64: [F][C] edge records (5..8 channels)
    uint32_t layout_epoch = 0;
    uint64_t packed_cap = 0;        // bytes
    bool packed_valid = false;
    bool resid_valid = false;       // d_resid holds the rows of the last analysed batch
    // a batch cut into frame ranges by flacgpu_encode_device (chunk_frames): analyze_impl / pack_impl launch their kernels for
    // frames [rng_f0, rng_f0 + rng_cnt) only; rng_cnt == 0: the whole batch
    uint32_t rng_f0 = 0, rng_cnt = 0;
    uint32_t flat_lo = 0, flat_hi = 0;    // samples [flat_lo, flat_hi) of a full block lie under window values of exactly 1.0
    uint32_t chunk_samples = 64u << 20;   // FLACGPU_TUNE_CHUNK_MSAMPLES (0: never cut)
    bool chunk_auto = true;               // nobody set the tuning: only stereo batches are cut (measured: +9 %; 8 channels: +-0)
    bool planar_valid = true;       // false: the last batch was analysed from the caller's interleaved PCM in
    const int32_t *direct_src = nullptr;   //   place (direct_src); d_planar is filled on demand (ensure_planar)
    Params last_params;
    uint32_t window_last_len = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;  // stream the last analysis / assembly was submitted to
    // asynchronous host path (flacgpu_encode_packed_async ...)
    uint64_t *h_off = nullptr;          // pinned: byte offsets of the frames of the batch in flight
    hipEvent_t ev_sizes = nullptr, ev_bytes = nullptr, ev_null = nullptr;
    bool sizes_pending = false, bytes_pending = false;
    // early download (FLACGPU_EARLY_DOWNLOAD): a copy of the previous batch's size (+ margin) queued right behind the
    // packing kernels, before this batch's sizes are known; the remainder (if any) follows once they are
    uint64_t prev_total = 0, early_bytes = 0;
    uint8_t *early_dst = nullptr;
    // last call
    uint32_t last_frames = 0, last_len = 0;
    bool timing = false;
    hipEvent_t ev[FLACGPU_N_KERNELS + 1];
    bool ev_ok = false;
    bool ev_used[FLACGPU_N_KERNELS];
    float last_ms[FLACGPU_N_KERNELS];
};

// Every entry point runs with the context's device current and restores the caller's device on
// the way out (a context may be created on one thread and used from another, whose current
// device is a different GPU).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define CTX_GUARD(c) DeviceGuard device_guard_((c)->device)

// Waits for the work this context submitted last (never for other contexts: no device-wide sync).
static hipStream_t ctx_stream(flacgpu_ctx *c) { return c->last_stream ? c->last_stream : c->own_stream; }
static int ctx_sync(flacgpu_ctx *c) {
    HIP_TRY(hipStreamSynchronize(ctx_stream(c)));
    return FLACGPU_OK;
}
// blocking copy ordered after this context's work only
static int copy_sync(flacgpu_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, ctx_stream(c)));
    return ctx_sync(c);
}

namespace {

// Window::generate (encode.rs:1725-1783) on the host: it needs cos(), whose last-ulp
// behaviour differs between libms; the table is built once per block length with the
// host libm (the one a Rust build on this box links) and uploaded.
void window_generate(int kind, float p, uint32_t n, std::vector<double> &w) {
    const double PI = 3.14159265358979323846264338327950288;
    w.assign(n, 1.0);
    if (kind == FLACGPU_WINDOW_RECTANGLE) return;
    auto hann = [&]() {
        double np = (double)(uint16_t)n - 1.0;
        for (uint32_t i = 0; i < n; i++) w[i] = 0.5 - 0.5 * cos(2.0 * PI * (double)i / np);
    };
    if (kind == FLACGPU_WINDOW_HANN) {
        hann();
        return;
    }
    if (p != p) p = 0.5f;  // NaN => Tukey(0.5)
    if (p <= 0.0f) return;
    if (p >= 1.0f) {
        hann();
        return;
    }
    double t = (double)p / 2.0 * (double)n;
    uint64_t tu = (uint64_t)t;
    if (tu == 0) return;
    uint64_t np = tu - 1;
    if (np > n || n - np < np) return;
    double npf = (double)(uint16_t)np;
    for (uint32_t i = 0; i < (uint32_t)np; i++) {
        double x = 0.5 - 0.5 * cos(PI * (double)i / npf);
        w[i] = x;
        w[n - 1 - i] = x;
    }
}

// thresholds for floor(log2(l)) (encode.rs:3360): for exponent e, the smallest double in
// [2^e, 2^(e+1)) whose host log2() already rounds up to e+1 (or 2^(e+1) if none)
void build_log2_thresholds(double *thr) {
    for (int e = -64; e < 64; e++) {
        double hi = ldexp(1.0, e + 1);
        double top = nextafter(hi, 0.0);
        if (floor(log2(top)) == (double)e) {
            thr[e + 64] = hi;
            continue;
        }
        uint64_t lo_b, hi_b;
        double lo = ldexp(1.0, e);
        memcpy(&lo_b, &lo, 8);
        memcpy(&hi_b, &top, 8);
        while (lo_b < hi_b) {  // first bit pattern with floor(log2) == e+1
            uint64_t mid = lo_b + (hi_b - lo_b) / 2;
            double m;
            memcpy(&m, &mid, 8);
            if (floor(log2(m)) == (double)e) lo_b = mid + 1;
            else hi_b = mid;
        }
        memcpy(&thr[e + 64], &lo_b, 8);
    }
}

int upload_window(flacgpu_ctx *c, uint32_t n, double *dst, hipStream_t st) {
    std::vector<double> w;
    window_generate(c->opts.window_kind, c->opts.window_param, n, w);
    if (dst == c->d_window_full) {
        // the run of window values that are EXACTLY 1.0 (the flat middle of a Tukey window; empty for the others): samples
        // under it stay integers when windowed, which is what lets k_autocorr4 fuse multiply and add there (Params::fma_t0)
        uint32_t lo = 0;
        while (lo < n && w[lo] != 1.0) lo++;
        uint32_t hi = lo;
        while (hi < n && w[hi] == 1.0) hi++;
        c->flat_lo = lo;
        c->flat_hi = hi;   // [lo, hi)
    }
    w.resize((size_t)n + 64, 0.0);
    HIP_TRY(hipMemcpyAsync(dst, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // `w` is a local
    return 0;
}

// dynamic LDS of k_pack: the residual row + the subframe's bit string (a chosen subframe is
// never longer than its VERBATIM form: <= 40 + 33 n bits, plus the 16-byte frame header)
size_t pack_lds_bytes(uint32_t block_size) { return (size_t)pack_sb_words(block_size) * 4; }

// K0 for frames [f0, f0 + fcount): split the channels into planar rows and OR every candidate's
// samples.  Returns true when the orbits are already accumulated.
constexpr uint32_t kGridY = 65535;   // frames per launch of the kernels that index frames with blockIdx.y

bool launch_k0(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames, uint32_t last_len,
               uint32_t f0, uint32_t fcount, hipStream_t st) {
    if (fcount > kGridY) {   // batches of more than 65535 frames: several launches
        bool r = false;
        for (uint32_t f = f0; f < f0 + fcount; f += kGridY)
            r = launch_k0(c, d_pcm, layout, n_frames, last_len, f, std::min(kGridY, f0 + fcount - f), st);
        return r;
    }
    const uint32_t B = c->opts.block_size;
    const dim3 grid(std::max<uint32_t>(1u, (B + WG * 8 - 1) / (WG * 8)), fcount);  // 8 samples per lane
    if (layout == FLACGPU_LAYOUT_INTERLEAVED && c->channels == 2) {
        unsigned long long *abs = (c->d_abs && c->ncand == 4) ? c->d_abs : nullptr;   // (zeroed by the caller)
        hipLaunchKernelGGL(k_deinterleave2, grid, dim3(WG), 0, st, (const int2 *)d_pcm, c->d_planar, B, c->ldb,
                           n_frames, last_len, c->d_orbits, c->ncand, f0, abs);
        c->abs_valid = abs != nullptr;
        return true;
    }
    if (layout == FLACGPU_LAYOUT_INTERLEAVED && c->ncand == c->channels) {
        switch (c->channels) {
#define X(C) case C: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deinterleave_n<C>), grid, dim3(WG), 0, st, d_pcm, \
                                        c->d_planar, B, c->ldb, n_frames, last_len, c->d_orbits, f0); return true;
            X(1) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
        default: break;
        }
    }
    hipLaunchKernelGGL(k_deinterleave, grid, dim3(WG), 0, st, d_pcm, c->d_planar, c->channels, B, c->ldb,
                       n_frames, last_len, layout == FLACGPU_LAYOUT_PLANAR, f0);
    return false;
}

// K0 for stream-width little-endian input (k_deinterleave_packed); false: no instantiation / layout
// the kernel cannot take (the caller widens on the host instead)
bool packed_k0_supported(uint32_t block_size, uint32_t channels, uint32_t bytes) {
    return bytes >= 1 && bytes <= 3 && block_size % 4 == 0 && ((size_t)block_size * channels * bytes) % 16 == 0;
}
template <int C>
void launch_k0_packed_c(flacgpu_ctx *c, uint32_t bytes, const dim3 &grid, uint32_t n_frames, uint32_t last_len,
                        uint32_t f0, hipStream_t st) {
    const uint32_t *in = c->packed_src ? reinterpret_cast<const uint32_t *>(c->packed_src)
                                       : reinterpret_cast<const uint32_t *>(c->d_in);
    const uint32_t B = c->opts.block_size;
    unsigned long long *abs = (C == 2 && c->d_abs && c->ncand == 4) ? c->d_abs : nullptr;   // (zeroed by the caller)
    switch (bytes) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deinterleave_packed<C, 1>), grid, dim3(WG), 0, st, in, c->d_planar, B,
                               c->ldb, n_frames, last_len, c->d_orbits, f0, abs); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deinterleave_packed<C, 2>), grid, dim3(WG), 0, st, in, c->d_planar, B,
                               c->ldb, n_frames, last_len, c->d_orbits, f0, abs); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deinterleave_packed<C, 3>), grid, dim3(WG), 0, st, in, c->d_planar, B,
                                c->ldb, n_frames, last_len, c->d_orbits, f0, abs); break;
    }
    c->abs_valid = abs != nullptr;
}
void launch_k0_packed(flacgpu_ctx *c, uint32_t bytes, uint32_t n_frames, uint32_t last_len, hipStream_t st) {
    const uint32_t B = c->opts.block_size;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += kGridY) {
        const dim3 grid(std::max<uint32_t>(1u, (B / 4 + WG - 1) / WG), std::min(kGridY, n_frames - f0));   // 4 PCM frames per lane
        switch (c->channels) {
#define X(C) case C: launch_k0_packed_c<C>(c, bytes, grid, n_frames, last_len, f0, st); break;
            X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
        default: break;
        }
    }
}
__global__ void __launch_bounds__(WG) k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (size_t)gridDim.x * WG) dst[i] = src[i];
}
void launch_orbits(const Params &p, uint32_t *orbits, hipStream_t st) {
    for (uint32_t f = p.f0; f < p.f0 + p.fcount; f += kGridY) {
        Params r = p;
        r.f0 = f;
        r.fcount = std::min(kGridY, p.f0 + p.fcount - f);
        hipLaunchKernelGGL(k_orbits, dim3(std::max<uint32_t>(1u, (p.block_size + WG * 8 - 1) / (WG * 8)), r.fcount),
                           dim3(WG), 0, st, r, orbits);
    }
}

const char *const kKernelNames[FLACGPU_N_KERNELS] = {
    "k_deinterleave", "k_stereo_stats", "k_fixed", "k_autocorr", "k_lpc",
    "k_fir",          "k_decide",       "k_emit",  "k_layout",   "k_pack",
    "k_crc",          "k_cand64"};

bool wave_block_size(uint32_t B) {
    switch (B) {
#define X(n, spl) case n:
        FLACGPU_WAVE_SIZES(X)
#undef X
        return true;
    default: return false;
    }
}

}  // namespace

extern "C" {

const char *flacgpu_last_error(void) { return g_last_error.c_str(); }
const char *flacgpu_kernel_name(int i) {
    return (i >= 0 && i < FLACGPU_N_KERNELS) ? kKernelNames[i] : "";
}

// dynamic-LDS opt-ins are per kernel and per device: set once per device, to the largest block the
// LDS-resident generic kernels accept (k_fixed / k_fir: 2 * block_size * 4 bytes and change)
static int set_kernel_attributes_once(int device) {
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64 || done[device]) return FLACGPU_OK;
    const size_t B = LDS_BLOCK_LIMIT;
    const int dyn = (int)((2 * B + B / 16 + 16) * sizeof(int32_t));
    HIP_TRY(hipFuncSetAttribute((const void *)k_fixed_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    HIP_TRY(hipFuncSetAttribute((const void *)k_fir_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    HIP_TRY(hipFuncSetAttribute((const void *)k_emit_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(B * sizeof(int32_t))));
    HIP_TRY(pack_set_attributes(pack_lds_bytes((uint32_t)B)));
    done[device] = true;
    return FLACGPU_OK;
}

static int create_impl(flacgpu_ctx *c, const flacgpu_options *o);

// k_layout's launches are cut into chunks of 256 tiles x 1024 frames, each with an epoch of its own, and the context hands
// epochs out in steps of 8 (next_layout_epoch, pack.hip launch_layout): a batch can hold 8 chunks and no more
static constexpr uint32_t kMaxBatchFrames = 8u * 256u * 1024u;
int flacgpu_create(const flacgpu_options *o, uint32_t bps, uint32_t channels, int device,
                   uint32_t max_frames, flacgpu_ctx **out) {
    if (!o || !out) return FLACGPU_ERR_INVALID_ARG;
    *out = nullptr;
    // Options validation, encode.rs:1418-1455; stream validation, :495, :1904
    if (o->block_size < 16 || o->block_size > 65535 || o->max_lpc_order > 32 ||
        o->max_partition_order > 15 || bps < 1 || bps > 32 || channels < 1 || channels > 8 ||
        max_frames == 0 || max_frames > kMaxBatchFrames) {
        g_last_error = "invalid option / stream parameter";
        return FLACGPU_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_last_error = "no HIP device";
        return FLACGPU_ERR_NO_DEVICE;
    }
    if (device >= ndev) {
        g_last_error = "no such HIP device";
        return FLACGPU_ERR_NO_DEVICE;
    }
    if (device < 0) HIP_TRY(hipGetDevice(&device));   // -1: the caller's current device
    flacgpu_ctx *c = new flacgpu_ctx();
    c->opts = *o;
    c->bps = bps;
    c->channels = channels;
    c->max_frames = max_frames;
    c->device = device;
    int rc;
    {
        CTX_GUARD(c);   // the caller's current device is left as it was
        rc = create_impl(c, o);
    }
    if (rc != FLACGPU_OK) {
        flacgpu_destroy(c);   // frees whatever was allocated before the failure
        return rc;
    }
    *out = c;
    return FLACGPU_OK;
}

static int create_impl(flacgpu_ctx *c, const flacgpu_options *o) {
    const uint32_t max_frames = c->max_frames, channels = c->channels, bps = c->bps;
    c->ldb = (o->block_size + 3u) & ~3u;
    c->stereo4 = (channels == 2 && bps < 32) ? 1u : 0u;  // side needs bps+1 <= 32 (encode.rs:2715)
    c->ncand = c->stereo4 ? 4u : channels;
    const size_t F = max_frames, B = o->block_size, C = channels, NC = c->ncand;
    const size_t slack = 64;
#define ALLOC(ptr, count) HIP_TRY(hipMalloc((void **)&(ptr), sizeof(*(ptr)) * (count)))
    ALLOC(c->d_in, F * B * C + slack);
    ALLOC(c->d_planar, F * C * c->ldb + slack);
    ALLOC(c->d_resid, F * C * B);
    ALLOC(c->d_window_full, B + slack);
    ALLOC(c->d_window_last, B + slack);
    ALLOC(c->d_log2_thr, 128);
    ALLOC(c->d_ac, F * NC * AC_LD);
    ALLOC(c->d_cinfo, F * NC);
    ALLOC(c->d_fixed, F * NC);
    ALLOC(c->d_cand, F * NC);
    ALLOC(c->d_out, F * C);
    ALLOC(c->d_lpc, F * NC);
    ALLOC(c->d_finfo, F);
    if (c->stereo4 && B == FN && o->max_lpc_order > 0 && !read_knobs().no_hand) {   // (c->knobs is filled further down)
        ALLOC(c->d_hand, F * 2);
        HIP_TRY(hipMemset(c->d_hand, 0, sizeof(uint32_t) * F * 2));
    }
    ALLOC(c->d_fplan, F);
    // Params::turn_counter (4 words), counters, then the per-candidate ORs (one memset per batch over all three), then Params::defer_stats
    ALLOC(c->d_stats_base, kTurnWords + 4 + F * NC + kDeferWords + 16);
    c->d_stats = c->d_stats_base + kTurnWords;
    HIP_TRY(hipMemset(c->d_stats_base, 0, kTurnWords * sizeof(uint32_t)));
    c->d_orbits = c->d_stats + 4;
    HIP_TRY(hipMemset(defer_stats_of(c), 0, kDeferWords * sizeof(uint32_t)));   // Params::defer_stats, cumulative
    // worst case: every subframe VERBATIM at 32 bits + headers
    c->packed_cap = (uint64_t)F * C * B * 4 + (uint64_t)F * (C * 8 + 64) + 256;
    ALLOC(c->d_packed, c->packed_cap / 4 + 8);
    ALLOC(c->d_frame_off, F + 1);
    ALLOC(c->d_tile_sync, F / 1024 + 2);
    HIP_TRY(hipMemset(c->d_tile_sync, 0, sizeof(unsigned long long) * (F / 1024 + 2)));
    if (channels >= 5 && !read_knobs().no_sub64) ALLOC(c->d_edges, F * channels);   // (c->knobs is filled further down)
    if (B > LDS_BLOCK_LIMIT) ALLOC(c->d_big, F * NC * (size_t)big_scratch_ints((uint32_t)B));
    ALLOC(c->d_ties, F * NC);
    if (c->stereo4 && !o->exhaustive_channel_correlation) ALLOC(c->d_abs, F * 4);
    c->knobs = read_knobs();
    c->env_no_direct = c->knobs.no_direct;
    if (c->knobs.has_tie_band) c->tie_band = c->knobs.tie_band;            // test knobs
    if (c->knobs.has_tie_perturb) c->tie_perturb = c->knobs.tie_perturb;
#undef ALLOC
    // non-blocking: contexts must not synchronise with each other through the legacy null stream
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_layout, hipEventDisableTiming));
    HIP_TRY(hipMemsetAsync(c->d_planar, 0, sizeof(int32_t) * (F * C * c->ldb + slack), c->own_stream));
    HIP_TRY(hipMemsetAsync(c->d_cinfo, 0, sizeof(CandInfo) * F * NC, c->own_stream));
    double thr[128];
    build_log2_thresholds(thr);
    HIP_TRY(hipMemcpyAsync(c->d_log2_thr, thr, sizeof thr, hipMemcpyHostToDevice, c->own_stream));
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    if (int rc = upload_window(c, o->block_size, c->d_window_full, c->own_stream)) return rc;
    if (int rc = set_kernel_attributes_once(c->device)) return rc;
    HIP_TRY(hipEventCreateWithFlags(&c->ev_sizes, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_bytes, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_null, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void **)&c->h_off, sizeof(uint64_t) * (F + 1), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->h_stats, sizeof(uint32_t) * 4, hipHostMallocDefault));
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    c->ev_ok = true;
    return FLACGPU_OK;
}

void flacgpu_destroy(flacgpu_ctx *c) {
    if (!c) return;
    CTX_GUARD(c);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->h_off) (void)hipHostFree(c->h_off);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    (void)hipFree(c->d_ties);
    (void)hipFree(c->d_abs);
    if (c->ev_sizes) (void)hipEventDestroy(c->ev_sizes);
    if (c->ev_bytes) (void)hipEventDestroy(c->ev_bytes);
    if (c->ev_null) (void)hipEventDestroy(c->ev_null);
    (void)hipFree(c->d_in); (void)hipFree(c->d_planar); (void)hipFree(c->d_resid); (void)hipFree(c->d_window_full);
    (void)hipFree(c->d_window_last); (void)hipFree(c->d_log2_thr); (void)hipFree(c->d_ac); (void)hipFree(c->d_cinfo);
    (void)hipFree(c->d_fixed); (void)hipFree(c->d_cand); (void)hipFree(c->d_out); (void)hipFree(c->d_lpc);
    (void)hipFree(c->d_hand);
    (void)hipFree(c->d_finfo); (void)hipFree(c->d_fplan); (void)hipFree(c->d_stats_base);
    (void)hipFree(c->d_packed); (void)hipFree(c->d_frame_off); (void)hipFree(c->d_tile_sync);
    if (c->d_seg_fn) (void)hipFree(c->d_seg_fn);
    if (c->d_seg_ptr) (void)hipFree(c->d_seg_ptr);
    if (c->h_seg) (void)hipHostFree(c->h_seg);
    (void)hipFree(c->d_edges);
    (void)hipFree(c->d_decoded); (void)hipFree(c->d_verify); (void)hipFree(c->d_big);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_layout) (void)hipEventDestroy(c->ev_layout);
    if (c->ev_ok) for (auto &e : c->ev) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int flacgpu_set_timing(flacgpu_ctx *c, int enable) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    c->timing = enable != 0;
    return 0;
}

// Params::defer_stats: behind the ORs, on a 64-byte line of its own
static uint32_t *defer_stats_of(const flacgpu_ctx *c) {
    const size_t off = (kTurnWords + 4 + (size_t)c->max_frames * c->ncand + 15u) & ~(size_t)15u;
    return c->d_stats_base + off;
}
static void fill_params(const flacgpu_ctx *c, uint32_t n_frames, uint32_t last_len, Params &p) {
    const uint32_t B = c->opts.block_size;
    memset(&p, 0, sizeof p);
    p.channels = c->channels;
    p.bps = c->bps;
    p.block_size = B;
    p.ldb = c->ldb;
    p.ncand = c->ncand;
    p.stereo4 = c->stereo4;
    p.mid_side = c->opts.mid_side;
    p.exhaustive = c->opts.exhaustive_channel_correlation;
    p.max_lpc_order = c->opts.max_lpc_order;
    p.max_po = c->opts.max_partition_order;
    p.use_rice2 = c->bps > 16;  // encode.rs:1965
    p.n_frames = n_frames;
    p.last_len = last_len;
    p.f0 = 0;
    p.fcount = n_frames;
    p.ac_split = (uint32_t)c->lag_split;
    p.fma_t0 = p.fma_t1 = 0;
    // integer samples below 2^26 (candidates of <= 25 bits + the side channel's extra bit) have exact f64 products
    if (!c->knobs.no_ac_fma && c->bps + (c->stereo4 ? 1u : 0u) <= 26u && c->flat_hi > c->flat_lo) {
        // (the margin is the most lags a launch can carry -- its lag count is the order + 1 rounded up to four --, not the order)
        const uint32_t maxlag = c->opts.max_lpc_order <= 16 ? 16u : 32u;
        p.fma_t0 = (c->flat_lo + maxlag + 31u) / 32u;      // first 32-sample tile whose every partner sample is flat too
        p.fma_t1 = c->flat_hi / 32u;                        // tiles t < fma_t1 end inside the flat run
    }
    p.planar = c->d_planar;
    p.inter = nullptr;
    p.split_src = nullptr;
    p.split_dst = nullptr;
    p.xpose = 0;
    p.window_full = c->d_window_full;
    p.window_last = c->d_window_last;
    p.log2_thr = c->d_log2_thr;
    p.cinfo = c->d_cinfo;
    p.fixed_plan = c->d_fixed;
    // independent channels: candidate c of a frame is its subframe c, so the candidate stage writes the subframe slots
    // directly and k_decide has nothing to copy (config 4: 18 MB each way per batch)
    p.cand_plan = c->stereo4 ? c->d_cand : c->d_out;
    p.out_plan = c->d_out;
    p.lpc = c->d_lpc;
    p.ac = c->d_ac;
    p.finfo = c->d_finfo;
    p.frame_plan = c->d_fplan;
    p.residuals = c->d_resid;
    p.stats = c->d_stats;
    p.big_scratch = c->d_big;
    p.big_stride = big_scratch_ints(B);
    p.tie_list = c->d_ties;
    p.tie_cap = c->max_frames * c->ncand;
    p.tie_band = c->tie_band;
    p.tie_perturb = c->tie_perturb;
    p.check_fir = c->knobs.force_fir_check ? 1u : 0u;
    p.fir_suspect_bits = c->knobs.fir_suspect_bits ? c->knobs.fir_suspect_bits : 30u;
    p.defer_fixed = c->knobs.defer_fixed >= 0 ? (uint32_t)c->knobs.defer_fixed : 1u;
    p.defer_margin16 = c->knobs.defer_margin16 >= 0 ? (uint32_t)c->knobs.defer_margin16 : kDeferMargin16;
    p.defer_stats = defer_stats_of(c);   // behind the ORs; cumulative, zeroed at creation
    p.turn_counter = c->d_stats_base;
    p.hand_meta = nullptr;   // (set by analyze_impl for the batches whose candidate kernel hands its residuals over)
}

// stream == NULL: the context's own (non-blocking) stream, ordered AFTER whatever the caller has
// already submitted to the legacy default stream (producers of d_pcm there are safe); results are
// ordered on the context's stream, which every fetch / wait entry point synchronises.
static int resolve_stream(flacgpu_ctx *c, void *stream, hipStream_t *out) {
    if (stream) {
        *out = (hipStream_t)stream;
        return FLACGPU_OK;
    }
    HIP_TRY(hipEventRecord(c->ev_null, nullptr));
    HIP_TRY(hipStreamWaitEvent(c->own_stream, c->ev_null, 0));
    *out = c->own_stream;
    return FLACGPU_OK;
}

// packed_bytes != 0: the PCM sits in c->d_in as interleaved little-endian samples of that many bytes
// Whether a batch of interleaved i32 PCM can be analysed and assembled in place (Params::inter): stereo with
// the four L/R/M/S candidates of <= 24-bit samples, whole blocks of one of the wave block lengths (4096; 1024 / 1152 /
// 2048 / 2304 with LPC order <= 16 -- the persistent candidate kernel stages the interleaved frame once per frame in
// LDS; the non-persistent one read it once per candidate wave and lost to the split pass, 0.114 -> 0.172 ms at 1152
// samples), and every stage on its wave kernel -- none of the knobs that select an older or generic kernel (they read
// Params::planar).
static bool direct_input_ok(const flacgpu_ctx *c, const Params &p, uint32_t last_len) {
    const Knobs &kn = c->knobs;
    const bool off = kn.no_direct || kn.no_fast || kn.no_w64 || kn.no_persist || kn.no_ac3 || kn.ac_private ||
                     kn.experiment_mfma_ac || kn.no_fused_pack || kn.no_frame64;
    const uint32_t B = p.block_size;
    const bool block_ok = B == FN || (wave_block_size(B) && p.max_lpc_order <= 16 && !kn.no_direct_short);
    return !off && c->stereo4 && c->channels == 2 && c->bps <= 24 && block_ok && last_len == B &&
           p.max_po <= 6 && p.ac_split != 2 &&
           (size_t)frame_fb_words(p.channels, c->bps, B) * sizeof(int32_t) <= 150 * 1024;
}

// interleaved INDEPENDENT channels whose batch k_autocorr4's producers read themselves (Params::split_src; analyze_impl adds
// the conditions on the call: layout, alignment, no stream-width upload)
static bool split_input_ok(const flacgpu_ctx *c, const Params &p, uint32_t last_len) {
    const uint32_t B = p.block_size;
    return !c->stereo4 && c->channels >= 1 && c->ncand == c->channels && p.max_lpc_order >= 1 && p.max_lpc_order <= 16 && B == FN &&
           last_len == B && p.ac_split != 2 && (c->bps <= 25u) && p.max_po <= 6 &&
           !(c->knobs.no_direct || c->knobs.no_fast || c->knobs.no_w64 || c->knobs.no_ac3 || c->knobs.ac_private ||
             c->knobs.experiment_mfma_ac);
}

// k_layout's tiles find each other through PackParams::tile_sync, whose words carry the launch's 24-bit epoch: a new
// one per launch, the words wiped (in stream order) when the counter wraps
static int next_layout_epoch(flacgpu_ctx *c, PackParams &q, hipStream_t st) {
    // (steps of 8: launch_layout cuts a launch of more than 256 tiles into chunks with epochs of their own)
    c->layout_epoch = (c->layout_epoch + 8) & 0xFFFFF8u;
    if (c->layout_epoch == 0) {
        HIP_TRY(hipMemsetAsync(c->d_tile_sync, 0, sizeof(unsigned long long) * (c->max_frames / 1024 + 2), st));
        c->layout_epoch = 8;
    }
    q.tile_sync = c->d_tile_sync;
    q.epoch = c->layout_epoch;
    q.edges = c->d_edges;
    return FLACGPU_OK;
}

// d_planar of the last batch, for the consumers outside the hot path (verification against the input,
// residual rows, flacgpu_device_buffer): a DIRECT batch never split its channels, so K0 runs now, from the
// caller's buffer
static int ensure_planar(flacgpu_ctx *c) {
    if (c->planar_valid) return FLACGPU_OK;
    if (!c->direct_src || c->last_frames == 0) return FLACGPU_OK;
    if (int rc = ctx_sync(c)) return rc;
    if (c->seg_active && c->seg_direct) {   // frame f0 + i of segment s is frame i of its own buffer
        uint32_t f0 = 0;
        const size_t frame_ints = (size_t)c->opts.block_size * c->channels;
        for (const flacgpu_segment &g : c->segs) {
            (void)launch_k0(c, g.pcm - (ptrdiff_t)(f0 * frame_ints), FLACGPU_LAYOUT_INTERLEAVED, c->last_frames, c->last_len, f0,
                            g.n_frames, ctx_stream(c));
            f0 += g.n_frames;
        }
    } else
    (void)launch_k0(c, c->direct_src, FLACGPU_LAYOUT_INTERLEAVED, c->last_frames, c->last_len, 0, c->last_frames,
                    ctx_stream(c));
    HIP_TRY(hipGetLastError());
    if (int rc = ctx_sync(c)) return rc;
    c->planar_valid = true;
    return FLACGPU_OK;
}

static int analyze_impl(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames,
                        uint32_t last_len, hipStream_t st, uint32_t packed_bytes);

// A batch that is NOT made of segments begins: the per-frame tables of an earlier flacgpu_encode_segments* batch on this
// context (frame numbers, addresses, the segments ensure_planar would walk) must not reach its frame assembly or verifier.
static void begin_plain_batch(flacgpu_ctx *c) {
    c->seg_active = c->seg_pending = c->seg_direct = false;
    c->segs.clear();
}

int flacgpu_analyze_device(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames,
                           uint32_t last_len, void *stream) {
    if (!c || !d_pcm) {
        g_last_error = "invalid analyze arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    hipStream_t st;
    if (int rc = resolve_stream(c, stream, &st)) return rc;
    return analyze_impl(c, d_pcm, layout, n_frames, last_len, st, 0);
}

static int analyze_impl(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames,
                        uint32_t last_len, hipStream_t st, uint32_t packed_bytes) {
    if (!c || !d_pcm || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size || (layout != 0 && layout != 1)) {
        g_last_error = "invalid analyze arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    // (a batch made of segments announces itself just before; the entry points that start a batch WITHOUT passing here --
    // the two-ranges branch of flacgpu_encode_device, flacgpu_pack_plans -- call begin_plain_batch themselves)
    if (c->rng_cnt == 0 || c->rng_f0 == 0) {
        if (c->seg_pending) {
            c->seg_active = true;
            c->seg_pending = false;
        } else {
            begin_plain_batch(c);
        }
    }
    const uint32_t B = c->opts.block_size;
    // the reference collects partitions into ArrayVec<_, 64> and panics beyond (encode.rs:3880)
    for (uint32_t n : {n_frames > 1 ? B : last_len, last_len}) {
        uint32_t tz = (uint32_t)__builtin_ctz(n);
        if ((tz < c->opts.max_partition_order ? tz : c->opts.max_partition_order) > (uint32_t)MAXP) {
            g_last_error = "effective partition order > 6: the reference panics (MAX_PARTITIONS = 64)";
            return FLACGPU_ERR_UNSUPPORTED;
        }
    }
    if (last_len != B && last_len != c->window_last_len) {
        if (int rc = upload_window(c, last_len, c->d_window_last, st)) return rc;
        c->window_last_len = last_len;
    }
    Params p;
    fill_params(c, n_frames, last_len, p);
    c->host_out = nullptr;   // a new batch: frames go to d_packed unless flacgpu_encode_packed_async_host says otherwise
    c->out_in_host = false;

    int evi = 0;
    for (auto &u : c->ev_used) u = false;
    auto mark = [&](int k) {  // event BEFORE kernel k; the next mark closes it
        if (c->timing) {
            (void)hipEventRecord(c->ev[evi], st);
            c->ev_used[k] = true;
            evi++;
        }
    };
    int order_of_marks[FLACGPU_N_KERNELS];
    int n_marks = 0;
    auto begin = [&](int k) {
        order_of_marks[n_marks++] = k;
        mark(k);
    };

    const uint32_t ncb = n_frames * c->ncand;
    const bool ranged = c->rng_cnt != 0;                 // (flacgpu_encode_device checked that this batch can be cut)
    const uint32_t rf0 = ranged ? c->rng_f0 : 0u, rcnt = ranged ? c->rng_cnt : n_frames;
    const bool first_range = rf0 == 0;
    if (first_range) HIP_TRY(hipMemsetAsync(c->d_stats_base, 0, sizeof(uint32_t) * (kTurnWords + 4 + (size_t)ncb), st));  // + d_orbits
    // DIRECT input: interleaved i32 stereo PCM of whole 4096-sample blocks is read in place by the
    // autocorrelation, candidate and frame kernels (no K0 split; the ORs come out of k_autocorr4, so
    // k_candinfo runs after it) -- see kernels/autocorr.inc.  The caller's buffer is then the only copy of
    // the input: it must stay valid until the batch's results were fetched (include/flacenc_gpu.h).
    // the in-place kernels fetch 16-byte pieces (vector loads, LDS-DMA): a batch that does not start on a 16-byte boundary
    // (a view into a larger device buffer) takes the copying K0 path, whose loads are dwords
    const bool aligned16 = ((uintptr_t)d_pcm & 15u) == 0;
    const bool direct = !packed_bytes && aligned16 && layout == FLACGPU_LAYOUT_INTERLEAVED && direct_input_ok(c, p, last_len);
    c->planar_valid = !direct;
    c->direct_src = direct ? d_pcm : nullptr;
    if (direct) p.inter = d_pcm;
    // the winners' LPC residuals handed from k_cand64p to k_frame64 (Params::hand_meta): 4096-sample direct stereo frames with
    // LPC -- the candidate kernel's instantiations that keep the frame's image through the turn
    if (direct && B == FN && p.max_lpc_order > 0 && c->d_hand) p.hand_meta = c->d_hand;
    if (direct && c->seg_active && c->seg_direct) p.inter_tab = c->d_seg_ptr;   // (frames found through the table; d_pcm = the first segment)
    c->abs_valid = false;
    if (c->d_abs && first_range) HIP_TRY(hipMemsetAsync(c->d_abs, 0, sizeof(unsigned long long) * 4 * n_frames, st));
    // K0 (+ OR of every candidate's samples -> wasted bits)
    // (one channel: interleaved and planar are the same bytes -- no copy either)
    const bool planar_direct = !packed_bytes && aligned16 && (layout == FLACGPU_LAYOUT_PLANAR || c->channels == 1) && (B % 4 == 0) &&
                               last_len == B && !c->knobs.no_direct;
    // Independent channels, interleaved, with LPC: k_autocorr4's producers split the batch into the planar rows while
    // they read it (Params::split_src) -- the K0 pass (8 B per sample at the HBM roofline) disappears.  Needs every
    // frame on the wave kernels (no generic-path frame reads the rows before the autocorrelation has written them).
    // (one channel: the input is its own planar row -- nothing to split, but the ORs still come out of the
    // autocorrelation instead of a k_orbits pass over the batch)
    const bool split = !direct && !packed_bytes && aligned16 && (layout == FLACGPU_LAYOUT_INTERLEAVED || c->channels == 1) &&
                       (c->channels >= 2 || planar_direct) && split_input_ok(c, p, last_len);
    // 3, 4, 6 or 8 channels: the candidate and subframe kernels read the interleaved batch in place as well (load_lane_xpose: a
    // workgroup fetches a whole frame -- or half of a 6- or 8-channel one -- together) -- the planar rows, half of the
    // autocorrelation kernel's HBM traffic, are not written at all; like a DIRECT stereo batch, the caller's buffer is then
    // the only copy of the input.  (Whole frames of 5 or 6 channels in one workgroup: measured -- five or six 152-register
    // waves do not pack onto the four SIMDs, the candidate and subframe kernels lose more (0.09 -> 0.13-0.15, 0.15 ->
    // 0.25-0.27 ms per 67 M samples) than the autocorrelation gains (0.21 -> 0.14); 5 and 7 channels keep the rows.)
    // (6, 8 channels: k_sub64, which needs its edge records; 3, 4: k_frame64)
    // (>= 8-bit samples: the transposing buffer lies in the LDS image areas of the frame kernels, sized by the sample width)
    const uint32_t C_ = c->channels;
    // (the frames must then be assembled by k_frame64 / k_sub64 too -- pack_impl's f64w gates --: the generic k_emit / k_pack
    // read the planar rows, which this path never writes)
    const bool xpose = split && !c->knobs.no_xpose && !c->knobs.no_fused_pack && !c->knobs.no_frame64 && c->bps >= 8 &&
                       (C_ == 3 || C_ == 4 || ((C_ == 8 || C_ == 6) && c->d_edges));
    if (split) {
        p.split_src = d_pcm;
        if (c->seg_active && c->seg_direct) p.inter_tab = c->d_seg_ptr;   // (segments: every frame's address from the table)
        p.split_dst = (c->channels == 1 || xpose) ? nullptr : c->d_planar;
        p.xpose = xpose ? 1u : 0u;
        if (c->channels == 1) p.planar = d_pcm;   // [frame][B] with ldb == B (planar_direct)
        if (xpose) {
            c->planar_valid = false;
            c->direct_src = d_pcm;
        }
    }
    if (c->seg_active && c->seg_direct && !direct && !split) {
        g_last_error = "internal: a batch of segments was announced as read in place, but the analysis takes a copying path";
        return FLACGPU_ERR_UNSUPPORTED;
    }
    if (!direct && !split) begin(0);
    bool have_orbits = direct || split;
    if (direct || split) {
    } else if (packed_bytes) {
        launch_k0_packed(c, packed_bytes, n_frames, last_len, st);
        have_orbits = true;
    } else if (planar_direct) {
        p.planar = d_pcm;  // [frame][ch][B] with ldb == B
    } else {
        have_orbits = launch_k0(c, d_pcm, layout, n_frames, last_len, 0, n_frames, st);
    }
    if (!have_orbits)
        launch_orbits(p, c->d_orbits, st);
    const size_t dyn2 = (2 * (size_t)B + B / 16 + 16) * sizeof(int32_t);
    // (K0 summed the magnitudes on the way otherwise; direct input without LPC: k_cand64p<..., SELF> makes the choice)
    if (c->stereo4 && !p.exhaustive && !c->abs_valid && !(direct && p.max_lpc_order == 0)) {
        begin(1);
        if (direct) hipLaunchKernelGGL(k_stereo_stats_t<true>, dim3(n_frames), dim3(WG), 0, st, p);
        else hipLaunchKernelGGL(k_stereo_stats_t<false>, dim3(n_frames), dim3(WG), 0, st, p);
    }
    if (!direct && !split)
        hipLaunchKernelGGL(k_candinfo, dim3((ncb + WG - 1) / WG), dim3(WG), 0, st, p, c->d_orbits,
                           c->abs_valid ? c->d_abs : nullptr);
    // blocks of exactly 4096 samples take the register-resident kernels; anything else (other
    // block sizes, a short last frame, candidates wider than 25 bits) the generic LDS ones
    const bool narrow = (c->bps + (c->stereo4 ? 1u : 0u) <= 25u) && !c->knobs.no_fast;
    // wave-per-candidate kernel (FIXED + LPC analysis of a candidate in one wave, after the LPC
    // parameters are known): block lengths 64 x {16, 18, 32, 36, 64}, LPC order <= 16
    const bool w64 = narrow && wave_block_size(B) && (p.max_lpc_order <= 16 || B == FN) && p.max_po <= 6 &&
                     !c->knobs.no_w64;
    const uint32_t n_fast = w64 ? ((last_len == B) ? n_frames : n_frames - 1) : 0;
    Params pf = p, pg = p;
    pf.f0 = 0;
    pf.fcount = n_fast;
    pg.f0 = n_fast;
    pg.fcount = n_frames - n_fast;
    if (ranged) {
        if (n_fast != n_frames || !(direct || split)) {
            g_last_error = "internal: a ranged batch must run the in-place wave kernels on every frame";
            return FLACGPU_ERR_UNSUPPORTED;
        }
        pf.f0 = rf0;
        pf.fcount = rcnt;
    }
    const bool lpc = p.max_lpc_order > 0;
    // Otherwise the FIXED analysis and the autocorrelation -> Levinson chain, which only share
    // their input, run concurrently on two HIP streams (fork after k_candinfo, join before
    // k_fir).  With per-kernel timing enabled everything is serialised on one stream.
    const bool fork = lpc && !c->timing && !c->knobs.no_fork && !(w64 && pg.fcount == 0);
    hipStream_t sf = fork ? c->aux_stream : st;
    if (fork) {
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(sf, c->ev_fork, 0));
    }
    if (pg.fcount) {
        begin(2);
        if (B > LDS_BLOCK_LIMIT) hipLaunchKernelGGL(k_fixed_t<true>, dim3(pg.fcount * c->ncand), dim3(WG), 0, sf, pg);
        else hipLaunchKernelGGL(k_fixed_t<false>, dim3(pg.fcount * c->ncand), dim3(WG), dyn2, sf, pg);
    }
    if (fork) HIP_TRY(hipEventRecord(c->ev_join, sf));
    if (lpc) {
        const uint32_t H = ((p.max_lpc_order + 1) + 3u) & ~3u;
        begin(3);
        const uint32_t full = (last_len == B) ? n_frames : n_frames - 1;
        bool k4_done = false;   // the direct autocorrelation kernels run K4 in their tail (whole blocks only: one launch)
        if (ranged) k4_done = dispatch_autocorr(H, p, c->knobs, rf0, rcnt, B, c->d_window_full, st);
        else if (full) k4_done = dispatch_autocorr(H, p, c->knobs, 0, full, B, c->d_window_full, st);
        if (full != n_frames) k4_done = dispatch_autocorr(H, p, c->knobs, full, 1, last_len, c->d_window_last, st) && k4_done;
        if (!k4_done || full != n_frames) {
            if (ranged) {
                g_last_error = "internal: a ranged batch needs K4 inside the autocorrelation kernel";
                return FLACGPU_ERR_UNSUPPORTED;
            }
            begin(4);
            launch_lpc(p, c->knobs, (ncb + 63) / 64, st);
        }
        if (fork) HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
        if (pg.fcount) {
            begin(5);
            if (B > LDS_BLOCK_LIMIT) hipLaunchKernelGGL(k_fir_t<true>, dim3(pg.fcount * c->ncand), dim3(WG), 0, st, pg);
            else hipLaunchKernelGGL(k_fir_t<false>, dim3(pg.fcount * c->ncand), dim3(WG), dyn2, st, pg);
        }
    }
    bool decided = false;   // the persistent stereo kernels choose the channel assignment themselves
    if (w64 && pf.fcount) {
        begin(11);
        decided = launch_cand64(pf, c->knobs, B, (pf.fcount * c->ncand + 3) / 4, st);
    }
    if (!decided && ranged) {
        begin(6);
        hipLaunchKernelGGL(k_decide, dim3(pf.fcount), dim3(64), 0, st, pf);
    } else if (!decided) {
        begin(6);
        hipLaunchKernelGGL(k_decide, dim3(n_frames), dim3(64), 0, st, p);
    } else if (pg.fcount) {
        begin(6);
        hipLaunchKernelGGL(k_decide, dim3(pg.fcount), dim3(64), 0, st, pg);
    }
    c->last_n_fast = ranged ? n_frames : pf.fcount;
    c->ties_checked = !lpc;
    c->ties_resolved = 0;
    c->fir_rechecked = 0;
    // the residual rows (k_emit) are produced lazily: flacgpu_fetch(residuals) / host packing
    // need them, the device-side packer recomputes residuals in registers instead
    c->resid_valid = false;
    if (c->timing) (void)hipEventRecord(c->ev[evi], st);
    HIP_TRY(hipGetLastError());
    c->last_frames = n_frames;
    c->last_len = last_len;
    c->last_params = p;
    c->last_stream = st;
    c->drained = false;
    c->packed_valid = false;
    if (c->timing) {
        HIP_TRY(hipStreamSynchronize(st));
        for (auto &m : c->last_ms) m = 0.f;
        for (int i = 0; i < n_marks; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]);
            c->last_ms[order_of_marks[i]] = ms;
        }
    }
    return FLACGPU_OK;
}

static int pack_impl(flacgpu_ctx *c, uint64_t first_frame_number, uint32_t sample_rate, hipStream_t st,
                     hipEvent_t after_layout);

// The candidates k_lpc listed (two best order estimates closer than the device's log() can be trusted
// to separate) are re-decided here with the HOST's libm -- Levinson, order choice and quantisation
// from the device's exact autocorrelation (host/lpc_host.cpp) --, their parameters replaced, and the
// candidate stage, the assignment decision and (when done before) the frame assembly run again, so
// that the output is what the reference produces with this host's libm (encode.rs:3656-3702).
static int resolve_order_ties(flacgpu_ctx *c) {
    if (c->ties_checked || c->last_frames == 0) return FLACGPU_OK;
    if (int rc = ctx_sync(c)) return rc;
    c->ties_checked = true;
    uint32_t s[4];
    if (int rc = copy_sync(c, s, c->d_stats, sizeof s, hipMemcpyDeviceToHost)) return rc;
    const uint32_t cap = c->max_frames * c->ncand;
    const uint32_t n_ties = std::min(s[1], cap);
    // s[3]: candidate waves whose unchecked FIR could not rule a ResidualOverflow out (Params::check_fir)
    const bool recheck_fir = s[3] != 0;
    if (n_ties == 0 && !recheck_fir) return FLACGPU_OK;
    static_assert(sizeof(flacenc::HostLpc) == sizeof(LpcParams), "same record on both sides");
    std::vector<uint32_t> list(n_ties);
    if (int rc = copy_sync(c, list.data(), c->d_ties, sizeof(uint32_t) * n_ties, hipMemcpyDeviceToHost)) return rc;
    Params p = c->last_params;
    if (recheck_fir) {
        p.check_fir = 1;   // the candidate stage below tests every FIR exactly (fir64_overflows)
        c->fir_rechecked = s[3];
    }
    const size_t nc = (size_t)p.n_frames * p.ncand;
    std::vector<double> ac;
    std::vector<CandInfo> ci;
    const bool bulk = n_ties > 32;
    if (bulk) {
        ac.resize(nc * AC_LD);
        ci.resize(nc);
        if (int rc = copy_sync(c, ac.data(), c->d_ac, sizeof(double) * ac.size(), hipMemcpyDeviceToHost)) return rc;
        if (int rc = copy_sync(c, ci.data(), c->d_cinfo, sizeof(CandInfo) * nc, hipMemcpyDeviceToHost)) return rc;
    }
    hipStream_t st = ctx_stream(c);
    for (uint32_t idx : list) {
        if (idx >= nc) continue;
        double row[AC_LD];
        CandInfo info;
        if (bulk) {
            memcpy(row, ac.data() + (size_t)idx * AC_LD, sizeof row);
            info = ci[idx];
        } else {
            if (int rc = copy_sync(c, row, c->d_ac + (size_t)idx * AC_LD, sizeof row, hipMemcpyDeviceToHost)) return rc;
            if (int rc = copy_sync(c, &info, c->d_cinfo + idx, sizeof info, hipMemcpyDeviceToHost)) return rc;
        }
        const uint32_t frame = idx / p.ncand;
        const uint32_t n = (frame + 1 == p.n_frames) ? p.last_len : p.block_size;
        flacenc::HostLpc h;
        flacenc::lpc_from_autocorr(row, p.max_lpc_order, n, info.bps, &h);
        HIP_TRY(hipMemcpyAsync(c->d_lpc + idx, &h, sizeof h, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));   // `h` is a local
    }
    // candidate stage + assignment again (k_fixed's results are still in place)
    const uint32_t B = p.block_size;
    Params pf = p, pg = p;
    pf.f0 = 0;
    pf.fcount = c->last_n_fast;
    pg.f0 = c->last_n_fast;
    pg.fcount = p.n_frames - c->last_n_fast;
    const size_t dyn2 = (2 * (size_t)B + B / 16 + 16) * sizeof(int32_t);
    if (pg.fcount) {
        if (B > LDS_BLOCK_LIMIT) hipLaunchKernelGGL(k_fir_t<true>, dim3(pg.fcount * c->ncand), dim3(WG), 0, st, pg);
        else hipLaunchKernelGGL(k_fir_t<false>, dim3(pg.fcount * c->ncand), dim3(WG), dyn2, st, pg);
    }
    bool decided = false;
    if (pf.fcount) decided = launch_cand64(pf, c->knobs, B, (pf.fcount * c->ncand + 3) / 4, st);
    if (!decided) hipLaunchKernelGGL(k_decide, dim3(p.n_frames), dim3(64), 0, st, p);
    else if (pg.fcount) hipLaunchKernelGGL(k_decide, dim3(pg.fcount), dim3(64), 0, st, pg);
    HIP_TRY(hipGetLastError());
    c->resid_valid = false;
    c->ties_resolved = n_ties;
    if (c->packed_valid)
        if (int rc = pack_impl(c, c->last_first_frame, c->last_rate, st, nullptr)) return rc;
    return ctx_sync(c);
}

static int ensure_residual_rows(flacgpu_ctx *c) {
    if (c->resid_valid) return FLACGPU_OK;
    // (the rows take the place of what the candidate kernel handed to k_frame64 -- Params::hand_meta --: a frame assembly that
    // follows fetches its samples itself)
    c->last_params.hand_meta = nullptr;
    if (int rc = ensure_planar(c)) return rc;   // k_emit reads planar rows
    if (int rc = ctx_sync(c)) return rc;
    const Params &p = c->last_params;
    if (p.block_size > LDS_BLOCK_LIMIT)
        hipLaunchKernelGGL(k_emit_t<true>, dim3(p.n_frames * p.channels), dim3(WG), 0, ctx_stream(c), p);
    else
        hipLaunchKernelGGL(k_emit_t<false>, dim3(p.n_frames * p.channels), dim3(WG),
                           (size_t)p.block_size * sizeof(int32_t), ctx_stream(c), p);
    HIP_TRY(hipGetLastError());
    if (int rc = ctx_sync(c)) return rc;
    c->resid_valid = true;
    return FLACGPU_OK;
}

int flacgpu_fetch(flacgpu_ctx *c, flacgpu_frame_plan *plans, flacgpu_subframe_plan *subs,
                  int32_t *residuals) {
    if (!c || c->last_frames == 0) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    const size_t F = c->last_frames;
    hipStream_t st = c->own_stream;
    if (int rc = ctx_sync(c)) return rc;
    if (int rc = resolve_order_ties(c)) return rc;
    if (residuals)
        if (int rc = ensure_residual_rows(c)) return rc;
    if (plans) HIP_TRY(hipMemcpyAsync(plans, c->d_fplan, sizeof(*plans) * F, hipMemcpyDeviceToHost, st));
    if (subs)
        HIP_TRY(hipMemcpyAsync(subs, c->d_out, sizeof(*subs) * F * c->channels, hipMemcpyDeviceToHost, st));
    if (residuals)
        HIP_TRY(hipMemcpyAsync(residuals, c->d_resid,
                               sizeof(int32_t) * F * c->channels * c->opts.block_size,
                               hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FLACGPU_OK;
}

int flacgpu_analyze(flacgpu_ctx *c, const int32_t *pcm, int layout, uint32_t n_frames,
                    uint32_t last_len, flacgpu_frame_plan *plans, flacgpu_subframe_plan *subs,
                    int32_t *residuals) {
    if (!c || !pcm || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size) {
        g_last_error = "invalid analyze arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    const size_t B = c->opts.block_size, C = c->channels;
    const size_t count = ((size_t)(n_frames - 1) * B + last_len) * C;
    HIP_TRY(hipMemcpyAsync(c->d_in, pcm, count * sizeof(int32_t), hipMemcpyHostToDevice, c->own_stream));
    // planar host input with a short last frame is laid out [frame][ch][len]; handled by K0
    int rc = analyze_impl(c, c->d_in, layout, n_frames, last_len, c->own_stream, 0);
    if (rc) return rc;
    return flacgpu_fetch(c, plans, subs, residuals);
}


// after_layout: an event recorded right behind k_layout (the frame sizes are known from there on)
static int pack_impl(flacgpu_ctx *c, uint64_t first_frame_number, uint32_t sample_rate, hipStream_t st,
                     hipEvent_t after_layout);

int flacgpu_pack_device(flacgpu_ctx *c, uint64_t first_frame_number, uint32_t sample_rate,
                        void *stream) {
    if (!c || c->last_frames == 0) {
        g_last_error = "flacgpu_pack_device: no analysed batch";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    hipStream_t st;
    if (int rc = resolve_stream(c, stream, &st)) return rc;
    return pack_impl(c, first_frame_number, sample_rate, st, nullptr);
}

static int pack_impl(flacgpu_ctx *c, uint64_t first_frame_number, uint32_t sample_rate, hipStream_t st,
                     hipEvent_t after_layout) {
    c->last_first_frame = first_frame_number;
    c->last_rate = sample_rate;
    const Params &p = c->last_params;
    PackParams q;
    q.first_frame_number = first_frame_number;
    if (c->seg_active) q.frame_numbers = c->d_seg_fn;
    q.sample_rate = sample_rate;
    q.out_words = c->d_packed;   // (k_layout does not look at it; replaced below for host output)
    q.frame_off = c->d_frame_off;
    q.cap_bytes = c->packed_cap;
    hipEvent_t *ev = c->ev;  // reuse the event pool: [0..3]
    if (c->timing) (void)hipEventRecord(ev[0], st);
    const bool ranged = c->rng_cnt != 0;
    if (int rc = next_layout_epoch(c, q, st)) return rc;
    if (ranged) {   // this range's frames only: their offsets continue from the previous range's end (frame_off[f0])
        Params pr = p;
        pr.f0 = c->rng_f0;
        pr.fcount = c->rng_cnt;
        launch_layout(pr, q, st);
    } else {
        launch_layout(p, q, st);
    }
    if (after_layout) HIP_TRY(hipEventRecord(after_layout, st));
    // frames of a wave block length are assembled whole in LDS by k_frame64 (residuals recomputed
    // from the PCM, CRC-16 from LDS, one write of the finished bytes); any other frame goes
    // through k_emit (residual rows) -> k_pack (one workgroup per subframe, zero-filled output,
    // atomic OR at shared words) -> k_crc
    const uint32_t B = p.block_size;
    const bool narrow = (c->bps + (c->stereo4 ? 1u : 0u) <= 25u) && !c->knobs.no_fast;
    uint32_t fbw = frame_fb_words(p.channels, c->bps, B);
    if (c->stereo4 && p.exhaustive) fbw = std::min(fbw, frame_fb_words_exhaustive_stereo(c->bps, B));
    // wave per subframe: block lengths 64 x {16, 18, 32, 36, 64}; orders 17..32 and 5..8 channels
    // for 4096-sample blocks only (and not both)
    const bool f64w = narrow && wave_block_size(B) &&
                      (p.max_lpc_order <= 16 || (B == FN && p.channels <= 4)) &&
                      (p.channels <= 4 || B == FN) && p.max_po <= 6 &&
                      (size_t)fbw * sizeof(int32_t) <= 150 * 1024 &&
                      !c->knobs.no_fused_pack && !c->knobs.no_frame64;
    const uint32_t n_fast = f64w ? (p.last_len == B ? p.n_frames : p.n_frames - 1) : 0;
    const bool fused = n_fast != 0;
    Params pf = p, pg = p;
    pf.f0 = 0;
    pf.fcount = n_fast;
    pg.f0 = n_fast;
    pg.fcount = p.n_frames - n_fast;
    if (ranged) {
        if (n_fast != p.n_frames) {
            g_last_error = "internal: a ranged batch must be assembled by the wave kernels";
            return FLACGPU_ERR_UNSUPPORTED;
        }
        pf.f0 = c->rng_f0;
        pf.fcount = c->rng_cnt;
    }
    // host output: k_frame64 only ever stores (dwords inside a frame, bytes at its ends), so it can write
    // over PCIe into pinned memory; the generic packer builds its frames with atomic ORs and cannot
    c->out_in_host = c->host_out && pf.fcount && pg.fcount == 0 && c->host_out_cap >= c->packed_cap;
    if (c->out_in_host) q.out_words = reinterpret_cast<uint32_t *>(c->host_out);
    if (pg.fcount) launch_zero(q, p.n_frames, st);
    if (c->timing) (void)hipEventRecord(ev[1], st);
    if (pf.fcount) launch_frame64(pf, q, B, pf.fcount, (size_t)fbw * sizeof(int32_t), st);
    if (pg.fcount) {
        if (!c->resid_valid) {  // residual rows of these frames
            if (p.block_size > LDS_BLOCK_LIMIT)
                hipLaunchKernelGGL(k_emit_t<true>, dim3(pg.fcount * p.channels), dim3(WG), 0, st, pg);
            else
                hipLaunchKernelGGL(k_emit_t<false>, dim3(pg.fcount * p.channels), dim3(WG),
                                   (size_t)p.block_size * sizeof(int32_t), st, pg);
        }
        launch_pack(pg, q, pg.fcount * p.channels, pack_lds_bytes(p.block_size), st);
    }
    if (c->timing) (void)hipEventRecord(ev[2], st);
    {   // CRC-16 of the frames that did not take the fused kernel
        Params pc = p;
        pc.f0 = fused ? n_fast : 0;
        pc.fcount = p.n_frames - pc.f0;   // (a ranged batch: n_fast == n_frames, nothing left)
        if (pc.fcount) launch_crc(false, pc, q, pc.fcount, nullptr, st);
    }
    if (c->timing) (void)hipEventRecord(ev[3], st);
    HIP_TRY(hipGetLastError());
    c->packed_valid = true;
    c->last_stream = st;
    c->drained = false;
    if (c->timing) {
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < 3; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            c->last_ms[8 + i] = ms;
        }
        if (fused && n_fast == p.n_frames) c->last_ms[10] = 0.f;  // no generic k_crc launch: empty slot
    }
    return FLACGPU_OK;
}

int flacgpu_set_tuning(flacgpu_ctx *c, int key, int value) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    switch (key) {
    case FLACGPU_TUNE_TWO_RANGES: c->two_ranges = value != 0; return FLACGPU_OK;
    case FLACGPU_TUNE_CHUNK_MSAMPLES:
        c->chunk_samples = value <= 0 ? 0u : (uint32_t)std::min(value, 2047) << 20;
        c->chunk_auto = false;
        return FLACGPU_OK;
    case FLACGPU_TUNE_COPY_INPUT:
        // its own flag: turning the tuning off must not cancel an environment FLACGPU_NO_DIRECT=1
        c->copy_input = value != 0;
        c->knobs.no_direct = c->env_no_direct || c->copy_input;
        return FLACGPU_OK;
    case FLACGPU_TUNE_LAG_SPLIT:
        if (value != 2 && value != 4) break;
        c->lag_split = value;
        return FLACGPU_OK;
    case FLACGPU_TUNE_BLOCKING_WAIT: {
        if ((value != 0) == c->blocking_wait) return FLACGPU_OK;
        CTX_GUARD(c);
        if (int rc = ctx_sync(c)) return rc;   // nothing in flight while the way of waiting changes
        c->blocking_wait = value != 0;
        return FLACGPU_OK;
    }
    }
    g_last_error = "flacgpu_set_tuning: unknown key / value";
    return FLACGPU_ERR_INVALID_ARG;
}

// Frames per range when flacgpu_encode_device cuts a batch (0: it does not).  Only batches whose every frame runs the
// in-place wave kernels with K4 inside the autocorrelation kernel are cut -- the conditions of `direct` / `split` in
// analyze_impl, whole 4096-sample blocks, exhaustive channel choice (the fast one sums the whole batch first).
static uint32_t chunk_frames(const flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames, uint32_t last_len) {
    const uint32_t B = c->opts.block_size;
    const Knobs &kn = c->knobs;
    if (!c->chunk_samples || kn.no_chunk || c->timing || c->lag_split == 2 || B != FN || last_len != B || layout != FLACGPU_LAYOUT_INTERLEAVED) return 0;
    if (c->chunk_auto && !c->stereo4) return 0;   // independent channels: only on request (config 4: no gain measured)
    if (((uintptr_t)d_pcm & 15u) || kn.no_direct || kn.no_fast || kn.no_w64 || kn.no_ac3 || kn.ac_private || kn.experiment_mfma_ac ||
        kn.no_lpc_fuse || kn.no_fused_pack || kn.no_frame64 || kn.no_persist)
        return 0;
    const uint32_t lo = c->opts.max_lpc_order;
    if (lo < 1 || lo > 16 || c->opts.max_partition_order > 6 || c->bps + (c->stereo4 ? 1u : 0u) > 25u) return 0;
    if (c->stereo4 ? !(c->channels == 2 && c->bps <= 24 && c->opts.exhaustive_channel_correlation && c->ncand == 4)
                   : !(c->channels >= 2 && c->ncand == c->channels))
        return 0;
    if ((size_t)frame_fb_words(c->channels, c->bps, B) * sizeof(int32_t) > 150 * 1024) return 0;
    const uint64_t per_frame = (uint64_t)B * c->channels;
    uint32_t chunk = (uint32_t)(c->chunk_samples / per_frame) & ~63u;   // whole candidate groups of the autocorrelation
    if (chunk < 256 || n_frames < chunk + chunk / 2) return 0;          // (nothing to cut, or ranges too small to fill the chip)
    return chunk;
}

// Analysis + frame assembly of one batch in one call.  With FLACGPU_TUNE_TWO_RANGES set, when
// every frame takes the wave kernels (4096-sample blocks, order <= 16, <= 4 channels), the batch
// is cut into two frame ranges that run
// the whole kernel chain on two HIP streams: the HBM-bound (K0, copy-out), latency-bound (Levinson,
// layout, CRC) and VALU-bound (autocorrelation, k_cand64, k_frame64) kernels of the two ranges
// overlap instead of running back to back.  The second range's byte offsets continue from the first
// range's total (k_layout), so the output is the same contiguous byte string.  Anything else runs
// flacgpu_analyze_device + flacgpu_pack_device.
int flacgpu_encode_device(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames,
                          uint32_t last_len, uint64_t first_frame_number, uint32_t sample_rate,
                          void *stream) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    const uint32_t B = c->opts.block_size;
    const uint32_t fbw = frame_fb_words(c->channels, c->bps, B);
    const bool eligible =
        d_pcm && n_frames >= 256 && n_frames <= c->max_frames && last_len == B && wave_block_size(B) &&
        (c->bps + (c->stereo4 ? 1u : 0u) <= 25u) &&
        (c->opts.max_lpc_order <= 16 || (B == FN && c->channels <= 4)) &&
        c->opts.max_partition_order <= 6 && (c->channels <= 4 || B == FN) && (layout == 0 || layout == 1) &&
        (size_t)fbw * sizeof(int32_t) <= 150 * 1024 && !c->timing && !c->knobs.no_fast &&
        !c->knobs.no_w64 && !c->knobs.no_fused_pack && !c->knobs.no_frame64 &&
        c->two_ranges;
    hipStream_t st0;
    if (int rc = resolve_stream(c, stream, &st0)) return rc;
    if (!eligible) {
        if (!d_pcm) {
            g_last_error = "invalid encode arguments";
            return FLACGPU_ERR_INVALID_ARG;
        }
        // A batch of more samples than the chip's last-level cache likes (256 MiB of Infinity Cache: the candidate and
        // frame kernels re-read what the autocorrelation just streamed) is run range by range -- the whole kernel chain for
        // ~64 Mi samples at a time, each range's frame offsets continuing from the one before (k_layout): a stereo batch of
        // 16384 frames (134 M samples) runs 9 % faster as two ranges of 8192 (profiles/r04_chunk_ab.json; 8-channel batches
        // of 8192 frames as four ranges: no difference, so independent channels are cut only when the tuning asks).
        const uint32_t chunk = chunk_frames(c, d_pcm, layout, n_frames, last_len);
        if (chunk) {
            int rc = FLACGPU_OK;
            for (uint32_t f0 = 0; f0 < n_frames && rc == FLACGPU_OK; f0 += chunk) {
                c->rng_f0 = f0;
                c->rng_cnt = std::min(chunk, n_frames - f0);
                rc = analyze_impl(c, d_pcm, layout, n_frames, last_len, st0, 0);
                if (rc == FLACGPU_OK) rc = pack_impl(c, first_frame_number, sample_rate, st0, nullptr);
            }
            c->rng_f0 = c->rng_cnt = 0;
            return rc;
        }
        if (int rc = analyze_impl(c, d_pcm, layout, n_frames, last_len, st0, 0)) return rc;
        return pack_impl(c, first_frame_number, sample_rate, st0, nullptr);
    }
    hipStream_t st1 = c->aux_stream;
    begin_plain_batch(c);   // (this branch never passes analyze_impl)
    Params p;
    fill_params(c, n_frames, last_len, p);
    PackParams q;
    q.first_frame_number = first_frame_number;
    q.sample_rate = sample_rate;
    q.out_words = c->d_packed;
    q.frame_off = c->d_frame_off;
    q.cap_bytes = c->packed_cap;
    // (planar rows are read in place unless the caller asked for a copy: FLACGPU_TUNE_COPY_INPUT / FLACGPU_NO_DIRECT)
    const bool planar_direct = (layout == FLACGPU_LAYOUT_PLANAR) && (B % 4 == 0) && !c->knobs.no_direct && ((uintptr_t)d_pcm & 15u) == 0;
    if (planar_direct) p.planar = d_pcm;
    HIP_TRY(hipMemsetAsync(c->d_stats_base, 0, sizeof(uint32_t) * (kTurnWords + 4 + (size_t)n_frames * c->ncand), st0));  // + d_orbits
    HIP_TRY(hipEventRecord(c->ev_fork, st0));
    HIP_TRY(hipStreamWaitEvent(st1, c->ev_fork, 0));
    const uint32_t h = ((n_frames / 2) + 15u) & ~15u;  // 16 frames = one autocorrelation wave group
    const bool lpc = p.max_lpc_order > 0;
    const uint32_t H = ((p.max_lpc_order + 1) + 3u) & ~3u;
    const size_t l64 = (size_t)fbw * sizeof(int32_t);
    for (int half = 0; half < 2; half++) {
        hipStream_t st = half ? st1 : st0;
        Params r = p;
        r.f0 = half ? h : 0;
        r.fcount = half ? n_frames - h : h;
        r.turn_counter = c->d_stats_base + half;   // the two ranges' candidate kernels may run side by side
        const uint32_t ncb = r.fcount * c->ncand;
        bool have_orbits = false;
        if (!planar_direct) have_orbits = launch_k0(c, d_pcm, layout, n_frames, last_len, r.f0, r.fcount, st);
        if (!have_orbits) launch_orbits(r, c->d_orbits, st);
        if (c->stereo4 && !p.exhaustive) hipLaunchKernelGGL(k_stereo_stats_t<false>, dim3(r.fcount), dim3(WG), 0, st, r);
        hipLaunchKernelGGL(k_candinfo, dim3((ncb + WG - 1) / WG), dim3(WG), 0, st, r, c->d_orbits, (const unsigned long long *)nullptr);
        if (lpc) {
            if (!dispatch_autocorr(H, r, c->knobs, r.f0, r.fcount, B, c->d_window_full, st))
                launch_lpc(r, c->knobs, (ncb + 63) / 64, st);
        }
        if (!launch_cand64(r, c->knobs, B, (ncb + 3) / 4, st)) hipLaunchKernelGGL(k_decide, dim3(r.fcount), dim3(64), 0, st, r);
        // frame assembly of this range; the second range's offsets continue from the first's
        if (half) HIP_TRY(hipStreamWaitEvent(st, c->ev_layout, 0));
        if (int rc = next_layout_epoch(c, q, st)) return rc;
        launch_layout(r, q, st);
        if (!half) HIP_TRY(hipEventRecord(c->ev_layout, st));
        launch_frame64(r, q, B, r.fcount, l64, st);
    }
    HIP_TRY(hipEventRecord(c->ev_join, st1));
    HIP_TRY(hipStreamWaitEvent(st0, c->ev_join, 0));
    HIP_TRY(hipGetLastError());
    c->resid_valid = false;
    c->planar_valid = true;
    c->direct_src = nullptr;
    c->last_frames = n_frames;
    c->last_len = last_len;
    c->last_params = p;
    c->last_stream = st0;
    c->packed_valid = true;
    c->last_n_fast = n_frames;
    c->ties_checked = !lpc;
    c->ties_resolved = 0;
    c->fir_rechecked = 0;
    c->last_first_frame = first_frame_number;
    c->last_rate = sample_rate;
    return FLACGPU_OK;
}

// ---- a batch made of SEGMENTS: runs of whole blocks of several streams of the context's shape analysed and assembled as ONE
// batch (the kernels' launches and their serial walk over a block are paid once, not once per stream).  Frame f of the batch
// is frame f - f0(s) of its segment s; its frame number and, for direct stereo input read in place, its address come from
// per-frame tables (PackParams::frame_numbers, Params::inter_tab).
static int segments_common(flacgpu_ctx *c, const flacgpu_segment *segs, uint32_t n_segs, bool on_device, uint32_t sample_rate,
                           hipStream_t st) {
    const uint32_t B = c->opts.block_size;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_segs; i++) {
        if (!segs[i].pcm || segs[i].n_frames == 0) {
            g_last_error = "flacgpu_encode_segments: an empty segment";
            return FLACGPU_ERR_INVALID_ARG;
        }
        total += segs[i].n_frames;
    }
    if (total == 0 || total > c->max_frames) {
        g_last_error = "flacgpu_encode_segments: more frames than the context was created for";
        return FLACGPU_ERR_INVALID_ARG;
    }
    const size_t F = c->max_frames;
    if (!c->d_seg_fn) {
        HIP_TRY(hipMalloc((void **)&c->d_seg_fn, sizeof(uint64_t) * F));
        HIP_TRY(hipMalloc((void **)&c->d_seg_ptr, sizeof(int32_t *) * F));
        HIP_TRY(hipHostMalloc((void **)&c->h_seg, sizeof(uint64_t) * 2 * F, hipHostMallocDefault));
    }
    // the staging arrays and the device tables of the batch before are free once BOTH streams are idle: the one this call
    // runs on and the one the context's last batch (or its re-assembly after a host re-decision) was submitted to
    HIP_TRY(hipStreamSynchronize(st));
    if (int rc = ctx_sync(c)) return rc;
    const size_t frame_ints = (size_t)B * c->channels;
    Params probe;
    fill_params(c, (uint32_t)total, B, probe);
    // read in place (stereo: Params::inter; 2..8 independent channels: Params::split_src) or gathered into d_in
    bool direct = on_device && (direct_input_ok(c, probe, B) || (c->channels >= 2 && split_input_ok(c, probe, B)));
    for (uint32_t i = 0; i < n_segs && direct; i++) direct = ((uintptr_t)segs[i].pcm & 15u) == 0;
    uint64_t *fn = c->h_seg;
    const int32_t **ptr = reinterpret_cast<const int32_t **>(c->h_seg + F);
    uint32_t f = 0;
    c->segs.assign(segs, segs + n_segs);
    for (uint32_t i = 0; i < n_segs; i++) {
        for (uint32_t k = 0; k < segs[i].n_frames; k++, f++) {
            fn[f] = segs[i].first_frame_number + k;
            ptr[f] = segs[i].pcm + (size_t)k * frame_ints;
        }
        if (!direct)   // gathered into the context's input buffer (host segments: the upload itself)
            HIP_TRY(hipMemcpyAsync(c->d_in + (size_t)(f - segs[i].n_frames) * frame_ints, segs[i].pcm,
                                   sizeof(int32_t) * frame_ints * segs[i].n_frames,
                                   on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(c->d_seg_fn, fn, sizeof(uint64_t) * total, hipMemcpyHostToDevice, st));
    if (direct) HIP_TRY(hipMemcpyAsync(c->d_seg_ptr, ptr, sizeof(int32_t *) * total, hipMemcpyHostToDevice, st));
    c->seg_pending = true;
    c->seg_direct = direct;
    c->rng_f0 = c->rng_cnt = 0;
    if (int rc = analyze_impl(c, direct ? segs[0].pcm : c->d_in, FLACGPU_LAYOUT_INTERLEAVED, (uint32_t)total, B, st, 0)) {
        c->seg_pending = false;   // (refused before the analysis took the announcement)
        return rc;
    }
    return pack_impl(c, segs[0].first_frame_number, sample_rate, st, nullptr);
}

int flacgpu_encode_segments_device(flacgpu_ctx *c, const flacgpu_segment *segs, uint32_t n_segs, uint32_t sample_rate,
                                   void *stream) {
    if (!c || !segs || n_segs == 0) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    hipStream_t st;
    if (int rc = resolve_stream(c, stream, &st)) return rc;
    return segments_common(c, segs, n_segs, true, sample_rate, st);
}

int flacgpu_encode_segments(flacgpu_ctx *c, const flacgpu_segment *segs, uint32_t n_segs, uint32_t sample_rate, uint8_t *out,
                            size_t cap, uint64_t *offsets, uint64_t *total) {
    if (!c || !segs || n_segs == 0) return FLACGPU_ERR_INVALID_ARG;
    {
        CTX_GUARD(c);
        if (int rc = segments_common(c, segs, n_segs, false, sample_rate, c->own_stream)) return rc;
    }
    return flacgpu_fetch_frames(c, out, cap, offsets, total);
}

int flacgpu_fetch_frames(flacgpu_ctx *c, uint8_t *out, size_t cap, uint64_t *offsets,
                         uint64_t *total) {
    if (!c || !c->packed_valid) {
        g_last_error = "flacgpu_fetch_frames: nothing packed";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    const size_t F = c->last_frames;
    if (int rc = ctx_sync(c)) return rc;
    if (int rc = resolve_order_ties(c)) return rc;
    std::vector<uint64_t> off;
    uint64_t *offp = offsets;
    if (!offp) {
        off.resize(F + 1);
        offp = off.data();
    }
    if (int rc = copy_sync(c, offp, c->d_frame_off, sizeof(uint64_t) * (F + 1), hipMemcpyDeviceToHost)) return rc;
    const uint64_t bytes = offp[F];
    if (total) *total = bytes;
    if (!out || cap < bytes) {
        g_last_error = "output buffer too small";
        return FLACGPU_ERR_BUFFER_TOO_SMALL;
    }
    if (c->out_in_host) {   // the batch was assembled into the caller's host buffer (ctx_sync above: complete)
        if (out != c->host_out) memcpy(out, c->host_out, bytes);
        return FLACGPU_OK;
    }
    if (int rc = copy_sync(c, out, c->d_packed, bytes, hipMemcpyDeviceToHost)) return rc;
    return FLACGPU_OK;
}

int flacgpu_encode_frames(flacgpu_ctx *c, const int32_t *pcm, int layout, uint32_t n_frames,
                          uint32_t last_len, uint64_t first_frame_number, uint32_t sample_rate,
                          uint8_t *out, size_t cap, uint64_t *offsets, uint64_t *total) {
    if (!c || !pcm || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size) {
        g_last_error = "invalid encode arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    const size_t B = c->opts.block_size, C = c->channels;
    const size_t count = ((size_t)(n_frames - 1) * B + last_len) * C;
    HIP_TRY(hipMemcpyAsync(c->d_in, pcm, count * sizeof(int32_t), hipMemcpyHostToDevice, c->own_stream));
    int rc = flacgpu_encode_device(c, c->d_in, layout, n_frames, last_len, first_frame_number, sample_rate,
                                   c->own_stream);
    if (rc) return rc;
    return flacgpu_fetch_frames(c, out, cap, offsets, total);
}

// Frame assembly of caller-supplied decisions: the device-side counterpart of flacenc_pack_frames.
// The PCM is split into planar rows as for an analysis, the plans replace the ones an analysis would
// have produced, and the packers (k_frame64, or k_emit + k_pack + k_crc) run on them.
int flacgpu_pack_plans(flacgpu_ctx *c, const int32_t *pcm, uint32_t n_frames, uint32_t last_len,
                       const flacgpu_frame_plan *plans, const flacgpu_subframe_plan *subs,
                       uint64_t first_frame_number, uint32_t sample_rate) {
    if (!c || !pcm || !plans || !subs || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size) {
        g_last_error = "flacgpu_pack_plans: invalid arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    hipStream_t st = c->own_stream;
    begin_plain_batch(c);   // (no analyze_impl here)
    const size_t B = c->opts.block_size, C = c->channels;
    const size_t count = ((size_t)(n_frames - 1) * B + last_len) * C;
    HIP_TRY(hipMemcpyAsync(c->d_in, pcm, count * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(c->d_stats_base, 0, sizeof(uint32_t) * (kTurnWords + 4 + (size_t)n_frames * c->ncand), st));
    (void)launch_k0(c, c->d_in, FLACGPU_LAYOUT_INTERLEAVED, n_frames, last_len, 0, n_frames, st);
    HIP_TRY(hipMemcpyAsync(c->d_fplan, plans, sizeof(*plans) * n_frames, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_out, subs, sizeof(*subs) * n_frames * C, hipMemcpyHostToDevice, st));
    Params p;
    fill_params(c, n_frames, last_len, p);
    c->last_params = p;
    c->last_frames = n_frames;
    c->last_len = last_len;
    c->last_stream = st;
    c->resid_valid = false;
    c->planar_valid = true;
    c->direct_src = nullptr;
    c->packed_valid = false;
    HIP_TRY(hipStreamSynchronize(st));   // the host arrays may go away
    c->host_out = nullptr;
    c->out_in_host = false;
    return pack_impl(c, first_frame_number, sample_rate, st, nullptr);
}

// ---- stand-alone decoder: any FLAC stream (fLaC marker, metadata, frames) --------------------
namespace {
struct HostFrameHead {
    uint32_t n = 0, header_bytes = 0, blocking = 0, acode = 0, bps_code = 0;
};
// FrameHeader::parse (stream.rs:214-240) on the host: used by the scan for frame starts
bool host_parse_header(const uint8_t *d, size_t avail, HostFrameHead &h) {
    if (avail < 6 || d[0] != 0xFF || (d[1] & 0xFE) != 0xF8) return false;
    h.blocking = d[1] & 1;
    const uint32_t bcode = d[2] >> 4, rcode = d[2] & 15;
    h.acode = d[3] >> 4;
    h.bps_code = (d[3] >> 1) & 7;
    if ((d[3] & 1) || h.acode > 10 || bcode == 0 || rcode == 15 || h.bps_code == 3) return false;
    size_t k = 4;
    {   // UTF-8 like number
        const uint8_t b0 = d[k];
        uint32_t ones = 0;
        while (ones < 8 && (b0 & (0x80 >> ones))) ones++;
        if (ones == 1 || ones > 7) return false;
        const uint32_t extra = ones ? ones - 1 : 0;
        if (k + 1 + extra > avail) return false;
        for (uint32_t i = 1; i <= extra; i++)
            if ((d[k + i] & 0xC0) != 0x80) return false;
        k += 1 + extra;
    }
    switch (bcode) {
    case 1: h.n = 192; break;
    case 2: h.n = 576; break;
    case 3: h.n = 1152; break;
    case 4: h.n = 2304; break;
    case 5: h.n = 4608; break;
    case 6:
        if (k + 1 > avail) return false;
        h.n = d[k] + 1u;
        k += 1;
        break;
    case 7:
        if (k + 2 > avail) return false;
        h.n = ((uint32_t)d[k] << 8 | d[k + 1]) + 1u;
        k += 2;
        break;
    default: h.n = 256u << (bcode - 8); break;
    }
    if (rcode == 12) k += 1;
    else if (rcode == 13 || rcode == 14) k += 2;
    if (k + 1 > avail) return false;
    if (flacenc::crc8(d, k) != d[k]) return false;
    h.header_bytes = (uint32_t)k + 1;
    return true;
}
}  // namespace

// Decodes a whole FLAC stream held in host memory.  The host finds the frame boundaries (a frame
// ends where the next valid header -- sync code, CRC-8 -- begins AND the two bytes before it are
// the CRC-16 of everything since the frame's start; the last frame ends with the stream), the GPU
// decodes the frames in parallel, one lane per frame, re-checks every CRC-16 and undoes the stereo
// decorrelation; the MD5 of the decoded PCM is compared with STREAMINFO's on the host
// (decode.rs:1282 `verify`, 1388-1436 read_frame, 1494-1856).
int flacgpu_decode_stream(const uint8_t *data, size_t len, int device, int32_t *out, size_t out_cap,
                          flacgpu_stream_info *info) {
    if (!data || !info) return FLACGPU_ERR_INVALID_ARG;
    memset(info, 0, sizeof *info);
    if (len < 42 || memcmp(data, "fLaC", 4) != 0) {
        g_last_error = "not a FLAC stream (no fLaC marker)";
        return FLACGPU_ERR_INVALID_ARG;
    }
    size_t pos = 4;
    bool have_si = false;
    uint32_t min_frame = 0;
    for (;;) {   // metadata blocks (metadata/mod.rs:257-266): last flag + type, 24-bit length
        if (pos + 4 > len) return FLACGPU_ERR_INVALID_ARG;
        const bool last = data[pos] & 0x80;
        const uint32_t type = data[pos] & 0x7F;
        const size_t blen = (size_t)data[pos + 1] << 16 | (size_t)data[pos + 2] << 8 | data[pos + 3];
        pos += 4;
        if (pos + blen > len) return FLACGPU_ERR_INVALID_ARG;
        if (type == 0 && blen == 34) {   // STREAMINFO, metadata/mod.rs:1599-1630
            const uint8_t *b = data + pos;
            info->min_block = b[0] << 8 | b[1];
            info->max_block = b[2] << 8 | b[3];
            min_frame = b[4] << 16 | b[5] << 8 | b[6];
            info->sample_rate = (uint32_t)b[10] << 12 | (uint32_t)b[11] << 4 | b[12] >> 4;
            info->channels = ((b[12] >> 1) & 7) + 1;
            info->bits_per_sample = (((uint32_t)b[12] & 1) << 4 | b[13] >> 4) + 1;
            info->total_samples = ((uint64_t)(b[13] & 15) << 32) | (uint64_t)b[14] << 24 | (uint64_t)b[15] << 16 |
                                  (uint64_t)b[16] << 8 | b[17];
            memcpy(info->md5, b + 18, 16);
            have_si = true;
        }
        pos += blen;
        if (last) break;
    }
    if (!have_si || info->channels > 8 || info->bits_per_sample > 32 || info->max_block < 1) {
        g_last_error = "no usable STREAMINFO block";
        return FLACGPU_ERR_INVALID_ARG;
    }
    // ---- scan for frame starts
    uint16_t T[256];
    for (int i = 0; i < 256; i++) {
        uint16_t c = (uint16_t)(i << 8);
        for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
        T[i] = c;
    }
    std::vector<uint64_t> off;
    std::vector<uint32_t> fn;
    size_t p = pos;
    uint64_t samples = 0;
    uint32_t bad_scan = 0;
    while (p < len) {
        HostFrameHead h;
        if (!host_parse_header(data + p, len - p, h)) {
            bad_scan++;
            break;   // lost synchronisation: what follows is not decoded
        }
        // running CRC-16 from p; crc_m2 = CRC over [p, i - 2)
        uint16_t crc = 0, d1 = 0, d2 = 0;   // crc after i bytes, i-1 bytes, i-2 bytes
        size_t q = p;
        size_t end = 0;
        const size_t min_end = p + std::max<size_t>(h.header_bytes + 2 + info->channels, min_frame ? min_frame : 0);
        for (; q < len; q++) {
            if (q >= min_end && q >= p + 2) {
                // candidate: a header starts at q and bytes [q-2, q) are the CRC-16 of [p, q-2)
                if (data[q] == 0xFF && (data[q + (q + 1 < len ? 1 : 0)] & 0xFE) == 0xF8 &&
                    (uint16_t)(data[q - 2] << 8 | data[q - 1]) == d2) {
                    HostFrameHead hn;
                    if (host_parse_header(data + q, len - q, hn) && hn.blocking == h.blocking) {
                        end = q;
                        break;
                    }
                }
            }
            d2 = d1;
            d1 = crc;
            crc = (uint16_t)(T[(crc >> 8) ^ data[q]] ^ (crc << 8));
        }
        if (!end) {   // the last frame ends with the stream
            if (q == len && len >= p + 2 && (uint16_t)(data[len - 2] << 8 | data[len - 1]) == d2) end = len;
            else {
                bad_scan++;
                break;
            }
        }
        off.push_back(p);
        fn.push_back(h.n);
        samples += h.n;
        p = end;
    }
    off.push_back(p);
    const size_t F = fn.size();
    info->frames = (uint32_t)F;
    info->bad_frames = bad_scan;
    info->decoded_samples = samples;
    const size_t C = info->channels;
    if (F == 0) return FLACGPU_OK;
    if (out && out_cap < samples * C) {
        g_last_error = "output buffer too small";
        return FLACGPU_ERR_BUFFER_TOO_SMALL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_last_error = "no HIP device";
        return FLACGPU_ERR_NO_DEVICE;
    }
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    DeviceGuard guard(device);
    uint32_t maxn = 0;
    for (uint32_t v : fn) maxn = std::max(maxn, v);
    const size_t ldb = (maxn + 3u) & ~3u;
    // ---- device buffers (freed on every path by the guard object below)
    struct Bufs {
        void *bytes = nullptr, *off = nullptr, *fn = nullptr, *pcm = nullptr, *counts = nullptr;
        hipStream_t st = nullptr;
        ~Bufs() {
            (void)hipFree(bytes); (void)hipFree(off); (void)hipFree(fn); (void)hipFree(pcm); (void)hipFree(counts);
            if (st) (void)hipStreamDestroy(st);
        }
    } b;
    const size_t bytes_cap = (len + 64 + 3) & ~(size_t)3;
    HIP_TRY(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&b.bytes, bytes_cap));
    HIP_TRY(hipMalloc(&b.off, sizeof(uint64_t) * (F + 1)));
    HIP_TRY(hipMalloc(&b.fn, sizeof(uint32_t) * F));
    HIP_TRY(hipMalloc(&b.pcm, sizeof(int32_t) * (F * C * ldb + 64)));
    HIP_TRY(hipMalloc(&b.counts, sizeof(uint32_t) * (4 + F)));
    HIP_TRY(hipMemsetAsync(b.bytes, 0, bytes_cap, b.st));
    HIP_TRY(hipMemcpyAsync(b.bytes, data, len, hipMemcpyHostToDevice, b.st));
    HIP_TRY(hipMemcpyAsync(b.off, off.data(), sizeof(uint64_t) * (F + 1), hipMemcpyHostToDevice, b.st));
    HIP_TRY(hipMemcpyAsync(b.fn, fn.data(), sizeof(uint32_t) * F, hipMemcpyHostToDevice, b.st));
    HIP_TRY(hipMemsetAsync(b.counts, 0, sizeof(uint32_t) * (4 + F), b.st));
    launch_decode_frames((const uint32_t *)b.bytes, (const uint64_t *)b.off, (const uint32_t *)b.fn, bytes_cap,
                         (uint32_t)F, (uint32_t)C, info->bits_per_sample, (uint32_t)ldb, (int32_t *)b.pcm,
                         (uint32_t *)b.counts, b.st);
    Params pp;
    memset(&pp, 0, sizeof pp);
    pp.channels = (uint32_t)C;
    pp.bps = info->bits_per_sample;
    pp.block_size = maxn;
    pp.ldb = (uint32_t)ldb;
    pp.n_frames = (uint32_t)F;
    pp.last_len = fn[F - 1];
    pp.fcount = (uint32_t)F;
    PackParams q;
    q.first_frame_number = 0;
    q.sample_rate = info->sample_rate;
    q.out_words = (uint32_t *)b.bytes;
    q.frame_off = (uint64_t *)b.off;
    q.cap_bytes = bytes_cap;
    launch_crc(true, pp, q, (uint32_t)F, (uint32_t *)b.counts, b.st);
    launch_decode_finish(pp, (int32_t *)b.pcm, nullptr, (uint32_t *)b.counts, b.st, (const uint32_t *)b.fn);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> planar;
    try {   // sizes come from the (untrusted) stream's headers: an allocation failure is an error code, not an exception
        planar.resize(F * C * ldb);
    } catch (const std::bad_alloc &) {
        g_last_error = "flacgpu_decode_stream: out of host memory for the decoded PCM";
        return FLACGPU_ERR_UNSUPPORTED;
    }
    uint32_t counts[4];
    HIP_TRY(hipMemcpyAsync(planar.data(), b.pcm, sizeof(int32_t) * planar.size(), hipMemcpyDeviceToHost, b.st));
    HIP_TRY(hipMemcpyAsync(counts, b.counts, sizeof counts, hipMemcpyDeviceToHost, b.st));
    HIP_TRY(hipStreamSynchronize(b.st));
    info->bad_frames += counts[0];
    info->bad_crc16 = counts[1];
    // interleave + MD5 over ceil(bps / 8)-byte little-endian samples (decode.rs:1282 verify)
    flacenc::Md5 md5;
    const unsigned width = (info->bits_per_sample + 7) / 8;
    std::vector<uint8_t> le(maxn * C * width);
    size_t o = 0;
    for (size_t f = 0; f < F; f++) {
        const size_t n = fn[f];
        size_t k = 0;
        for (size_t i = 0; i < n; i++)
            for (size_t ch = 0; ch < C; ch++) {
                const int32_t v = planar[(f * C + ch) * ldb + i];
                if (out) out[o++] = v;
                for (unsigned w = 0; w < width; w++) le[k++] = (uint8_t)((uint32_t)v >> (8 * w));
            }
        md5.update(le.data(), k);
    }
    uint8_t got[16], zero[16] = {0};
    md5.digest(got);
    memcpy(info->decoded_md5, got, 16);
    info->md5_status = memcmp(info->md5, zero, 16) == 0 ? 2 : (memcmp(info->md5, got, 16) == 0 ? 1 : 0);
    return FLACGPU_OK;
}

// ---- asynchronous host path ------------------------------------------------------------------
void *flacgpu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        g_last_error = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}
void flacgpu_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
// What the host link of `device` carries (GB/s, both directions summed): mode 0 = that direction idle, 1 = a copy engine
// (hipMemcpyAsync on a stream of its own), 2 = a kernel loading from / storing to pinned host memory.  Best of 4 runs
// of `bytes` each way.  A diagnostic for bench.py's end_to_end.pipelined_pcie (tools/ubench/pcie_duplex.hip is the
// stand-alone form).
int flacgpu_link_probe(int device, size_t bytes, int up_mode, int down_mode, double *sum_gbs) {
    if (!sum_gbs || bytes < 4096 || up_mode < 0 || up_mode > 2 || down_mode < 0 || down_mode > 2 || (!up_mode && !down_mode))
        return FLACGPU_ERR_INVALID_ARG;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return FLACGPU_ERR_NO_DEVICE;
    DeviceGuard guard(dev);
    bytes &= ~(size_t)15;
    void *h_up = nullptr, *h_down = nullptr, *d_up = nullptr, *d_down = nullptr;
    hipStream_t s0 = nullptr, s1 = nullptr;
    auto cleanup = [&] {
        if (s0) (void)hipStreamDestroy(s0);
        if (s1) (void)hipStreamDestroy(s1);
        if (h_up) (void)hipHostFree(h_up);
        if (h_down) (void)hipHostFree(h_down);
        if (d_up) (void)hipFree(d_up);
        if (d_down) (void)hipFree(d_down);
    };
    auto fail = [&](hipError_t e, const char *what) {
        g_last_error = std::string(what) + ": " + hipGetErrorString(e);
        cleanup();
        return FLACGPU_ERR_HIP;
    };
#define PROBE_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(e_, #x); } while (0)
    PROBE_TRY(hipHostMalloc(&h_up, bytes, hipHostMallocDefault));
    PROBE_TRY(hipHostMalloc(&h_down, bytes, hipHostMallocDefault));
    PROBE_TRY(hipMalloc(&d_up, bytes));
    PROBE_TRY(hipMalloc(&d_down, bytes));
    memset(h_up, 1, bytes);
    PROBE_TRY(hipMemset(d_down, 2, bytes));
    PROBE_TRY(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    PROBE_TRY(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    double best = 0.0;
    for (int it = 0; it < 4; it++) {
        PROBE_TRY(hipStreamSynchronize(s0));
        PROBE_TRY(hipStreamSynchronize(s1));
        const auto t = std::chrono::steady_clock::now();
        if (up_mode == 1) PROBE_TRY(hipMemcpyAsync(d_up, h_up, bytes, hipMemcpyHostToDevice, s0));
        if (up_mode == 2)
            hipLaunchKernelGGL(k_copy16, dim3(256), dim3(WG), 0, s0, (const uint4 *)h_up, (uint4 *)d_up, bytes / 16);
        if (down_mode == 1) PROBE_TRY(hipMemcpyAsync(h_down, d_down, bytes, hipMemcpyDeviceToHost, s1));
        if (down_mode == 2)
            hipLaunchKernelGGL(k_copy16, dim3(256), dim3(WG), 0, s1, (const uint4 *)d_down, (uint4 *)h_down, bytes / 16);
        PROBE_TRY(hipStreamSynchronize(s0));
        PROBE_TRY(hipStreamSynchronize(s1));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
        best = std::max(best, (double)bytes * ((up_mode != 0) + (down_mode != 0)) / dt / 1e9);
    }
#undef PROBE_TRY
    cleanup();
    *sum_gbs = best;
    return FLACGPU_OK;
}

int flacgpu_current_device(void) {
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}
// The host side of a device: the NUMA node its PCI function hangs off and the CPUs local to it (sysfs), for the host threads
// that feed it (host/multi_device.cpp).  node -1 / an empty list: unknown (no sysfs entry, a single-node machine's -1).
int flacgpu_device_numa_info(int device, int *numa_node, char *cpulist, size_t cap) {
    if (numa_node) *numa_node = -1;
    if (cpulist && cap) cpulist[0] = 0;
    char bus[64] = {0};
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return FLACGPU_ERR_INVALID_ARG;
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return FLACGPU_ERR_HIP;
    for (char *p = bus; *p; p++) *p = (char)tolower((unsigned char)*p);   // sysfs spells the address in lower case
    auto read_line = [&](const char *leaf, char *dst, size_t dcap) {
        const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/" + leaf;
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return false;
        const bool ok = fgets(dst, (int)dcap, f) != nullptr;
        fclose(f);
        if (ok)
            for (char *p = dst; *p; p++)
                if (*p == '\n') *p = 0;
        return ok;
    };
    char tmp[64];
    if (numa_node && read_line("numa_node", tmp, sizeof tmp)) *numa_node = atoi(tmp);
    if (cpulist && cap) (void)read_line("local_cpulist", cpulist, cap);
    return FLACGPU_OK;
}

int flacgpu_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
int flacgpu_packed_input_supported(const flacgpu_ctx *c, uint32_t bytes_per_sample) {
    return c && packed_k0_supported(c->opts.block_size, c->channels, bytes_per_sample) ? 1 : 0;
}
int flacgpu_packed_input_shape_supported(uint32_t block_size, uint32_t channels, uint32_t bytes_per_sample) {
    return packed_k0_supported(block_size, channels, bytes_per_sample) ? 1 : 0;
}

// record `ev` on `st` (the waits below are for it)
static int record_waitable(flacgpu_ctx *, hipEvent_t ev, hipStream_t st) {
    HIP_TRY(hipEventRecord(ev, st));
    return FLACGPU_OK;
}
// FLACGPU_TUNE_BLOCKING_WAIT: the thread sleeps between looks at the event instead of spinning in
// hipEventSynchronize (which spins also with hipEventBlockingSync; a hipLaunchHostFunc wake-up keeps runtime
// helper threads busy instead: 0.3 CPU-seconds per 40 ms call with 64 writers)
static int wait_waitable(flacgpu_ctx *c, hipEvent_t ev) {
    if (c->blocking_wait) {
        for (unsigned spins = 0;; spins++) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return FLACGPU_OK;
            if (e != hipErrorNotReady) HIP_TRY(e);
            std::this_thread::sleep_for(std::chrono::microseconds(spins < 8 ? 50 : 200));
        }
    }
    HIP_TRY(hipEventSynchronize(ev));
    return FLACGPU_OK;
}

size_t flacgpu_packed_cap(const flacgpu_ctx *c) { return c ? (size_t)c->packed_cap + 64 : 0; }

int flacgpu_encode_packed_async(flacgpu_ctx *c, const uint8_t *pcm_le, uint32_t bytes_per_sample,
                                uint32_t n_frames, uint32_t last_len, uint64_t first_frame_number,
                                uint32_t sample_rate) {
    return flacgpu_encode_packed_async_host(c, pcm_le, bytes_per_sample, n_frames, last_len, first_frame_number,
                                            sample_rate, nullptr, 0);
}

static bool packed_async_args_ok(const flacgpu_ctx *c, const uint8_t *pcm_le, uint32_t bytes_per_sample, uint64_t n_frames,
                                 uint32_t last_len) {
    return c && pcm_le && n_frames != 0 && n_frames <= c->max_frames && last_len != 0 && last_len <= c->opts.block_size &&
           (bytes_per_sample == 4 ||
            (bytes_per_sample == (c->bps + 7) / 8 && packed_k0_supported(c->opts.block_size, c->channels, bytes_per_sample)));
}

static int packed_async_impl(flacgpu_ctx *c, const uint8_t *pcm_le, uint32_t bytes_per_sample, uint32_t n_frames, uint32_t last_len,
                             uint64_t first_frame_number, uint32_t sample_rate, uint8_t *out_host, size_t out_cap);

int flacgpu_encode_packed_async_host(flacgpu_ctx *c, const uint8_t *pcm_le, uint32_t bytes_per_sample,
                                     uint32_t n_frames, uint32_t last_len, uint64_t first_frame_number,
                                     uint32_t sample_rate, uint8_t *out_host, size_t out_cap) {
    if (!packed_async_args_ok(c, pcm_le, bytes_per_sample, n_frames, last_len)) {
        g_last_error = "flacgpu_encode_packed_async: invalid arguments / unsupported sample width for this stream shape";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    return packed_async_impl(c, pcm_le, bytes_per_sample, n_frames, last_len, first_frame_number, sample_rate, out_host, out_cap);
}

// The same with the batch made of SEGMENTS (frames of several streams, each numbered by its own stream): the segments' whole
// blocks lie back to back in `pcm_le` (the caller packed them there: flacgpu_segment::pcm is not looked at), the frame numbers
// reach the frame assembly through the per-frame table.
int flacgpu_encode_segments_packed_async_host(flacgpu_ctx *c, const uint8_t *pcm_le, uint32_t bytes_per_sample,
                                              const flacgpu_segment *segs, uint32_t n_segs, uint32_t sample_rate,
                                              uint8_t *out_host, size_t out_cap) {
    if (!c || !segs || n_segs == 0) return FLACGPU_ERR_INVALID_ARG;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_segs; i++) {
        if (segs[i].n_frames == 0) {
            g_last_error = "flacgpu_encode_segments_packed_async_host: an empty segment";
            return FLACGPU_ERR_INVALID_ARG;
        }
        total += segs[i].n_frames;
    }
    if (!packed_async_args_ok(c, pcm_le, bytes_per_sample, total, c->opts.block_size)) {
        g_last_error = "flacgpu_encode_segments_packed_async_host: invalid arguments / more frames than the context holds / "
                       "unsupported sample width for this stream shape";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    const size_t F = c->max_frames;
    if (!c->d_seg_fn) {
        HIP_TRY(hipMalloc((void **)&c->d_seg_fn, sizeof(uint64_t) * F));
        HIP_TRY(hipMalloc((void **)&c->d_seg_ptr, sizeof(int32_t *) * F));
        HIP_TRY(hipHostMalloc((void **)&c->h_seg, sizeof(uint64_t) * 2 * F, hipHostMallocDefault));
    }
    if (!c->drained) {   // (the table's staging copy of the batch before must be free; one batch per context at a time)
        if (int rc = ctx_sync(c)) return rc;
        HIP_TRY(hipStreamSynchronize(c->own_stream));
    }
    uint64_t *fn = c->h_seg;
    uint32_t f = 0;
    for (uint32_t i = 0; i < n_segs; i++)
        for (uint32_t k = 0; k < segs[i].n_frames; k++) fn[f++] = segs[i].first_frame_number + k;
    HIP_TRY(hipMemcpyAsync(c->d_seg_fn, fn, sizeof(uint64_t) * total, hipMemcpyHostToDevice, c->own_stream));
    c->segs.clear();          // (nothing is read in place: the batch is one contiguous run of blocks in the context's input buffer)
    c->seg_pending = true;
    c->seg_direct = false;
    c->rng_f0 = c->rng_cnt = 0;
    const int rc = packed_async_impl(c, pcm_le, bytes_per_sample, (uint32_t)total, c->opts.block_size, segs[0].first_frame_number,
                                     sample_rate, out_host, out_cap);
    if (rc) c->seg_pending = false;   // (refused before the analysis took the announcement)
    return rc;
}

static int packed_async_impl(flacgpu_ctx *c, const uint8_t *pcm_le, uint32_t bytes_per_sample, uint32_t n_frames, uint32_t last_len,
                             uint64_t first_frame_number, uint32_t sample_rate, uint8_t *out_host, size_t out_cap) {
    hipStream_t st = c->own_stream;
    // FLACGPU_TRACE_SUBMIT=1: where a submission spends its host time (stderr)
    static const bool trace = [] {
        const char *e = getenv("FLACGPU_TRACE_SUBMIT");
        return e && e[0] && e[0] != '0';
    }();
    using clk = std::chrono::steady_clock;
    const clk::time_point tr0 = clk::now();
    double tr_up = 0, tr_an = 0, tr_pk = 0;
    auto since = [&] { return std::chrono::duration<double, std::milli>(clk::now() - tr0).count(); };
    const size_t B = c->opts.block_size, C = c->channels;
    const size_t bytes = ((size_t)(n_frames - 1) * B + last_len) * C * bytes_per_sample;
    // The upward leg.  Default: a copy engine (hipMemcpyAsync).  FLACGPU_UPLOAD_KERNEL=1 (A/B, pinned memory only): the
    // bytes cross the link as kernel loads -- K0 reads the stream-width samples straight out of the caller's pinned
    // buffer; int32 samples take a copy kernel into d_in
    c->packed_src = nullptr;
    if (c->knobs.upload_by_kernel && bytes_per_sample != 4) {
        c->packed_src = pcm_le;
    } else if (c->knobs.upload_by_kernel && bytes % 16 == 0) {   // (whole 16-byte pieces only: no read past the caller's buffer)
        hipLaunchKernelGGL(k_copy16, dim3(1024), dim3(WG), 0, st, reinterpret_cast<const uint4 *>(pcm_le),
                           reinterpret_cast<uint4 *>(c->d_in), bytes / 16);
    } else {
        HIP_TRY(hipMemcpyAsync(c->d_in, pcm_le, bytes, hipMemcpyHostToDevice, st));
    }
    if (trace) tr_up = since();
    const int arc = analyze_impl(c, c->d_in, FLACGPU_LAYOUT_INTERLEAVED, n_frames, last_len, st,
                                 bytes_per_sample == 4 ? 0 : bytes_per_sample);
    c->packed_src = nullptr;
    if (arc) return arc;
    if (trace) tr_an = since();
    // the frame sizes leave the device as soon as k_layout has run (second stream), so that the host
    // can size the copy of the bytes while k_frame64 is still assembling them
    c->host_out = (reinterpret_cast<uintptr_t>(out_host) & 3u) ? nullptr : out_host;
    c->host_out_cap = out_cap;
    if (int rc = pack_impl(c, first_frame_number, sample_rate, st, c->ev_layout)) return rc;
    if (trace) tr_pk = since();
    // frames written straight to the host buffer: they are there when the stream has drained
    if (c->out_in_host)
        if (int rc = record_waitable(c, c->ev_bytes, st)) return rc;
    c->early_bytes = 0;
    c->early_dst = nullptr;
    if (!c->out_in_host && c->knobs.early_download && c->host_out && c->prev_total) {
        const uint64_t guess = std::min<uint64_t>(std::min<uint64_t>(c->packed_cap, c->host_out_cap),
                                                  c->prev_total + c->prev_total / 8 + 4096);
        HIP_TRY(hipMemcpyAsync(c->host_out, c->d_packed, guess, hipMemcpyDeviceToHost, st));
        if (int rc = record_waitable(c, c->ev_bytes, st)) return rc;
        c->early_bytes = guess;
        c->early_dst = c->host_out;
        g_early_downloads.fetch_add(1, std::memory_order_relaxed);
    }
    HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_layout, 0));
    HIP_TRY(hipMemcpyAsync(c->h_off, c->d_frame_off, sizeof(uint64_t) * ((size_t)n_frames + 1),
                           hipMemcpyDeviceToHost, c->aux_stream));
    HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, c->aux_stream));
    if (int rc = record_waitable(c, c->ev_sizes, c->aux_stream)) return rc;
    c->sizes_pending = true;
    c->bytes_pending = false;
    if (trace)
        std::fprintf(stderr, "[submit] %u frames, %zu bytes up: upload queued %.3f ms, analysis %.3f, assembly %.3f, all %.3f\n", n_frames, bytes,
                     tr_up, tr_an - tr_up, tr_pk - tr_an, since());
    return FLACGPU_OK;
}

int flacgpu_frames_ready(flacgpu_ctx *c, const uint64_t **offsets, uint64_t *total) {
    if (!c || !c->sizes_pending) {
        g_last_error = "flacgpu_frames_ready: no asynchronous batch in flight";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    if (int rc = wait_waitable(c, c->ev_sizes)) return rc;
    if (!c->ties_checked) {
        if (c->h_stats[1] == 0 && c->h_stats[3] == 0) {
            c->ties_checked = true;   // the common case: nothing to re-decide, nothing to wait for
        } else {                      // order ties: host re-decision, then the sizes again
            if (int rc = resolve_order_ties(c)) return rc;
            if (c->out_in_host)   // assembled again, into the host buffer again
                if (int rc = record_waitable(c, c->ev_bytes, ctx_stream(c))) return rc;
            if (int rc = copy_sync(c, c->h_off, c->d_frame_off, sizeof(uint64_t) * ((size_t)c->last_frames + 1),
                                   hipMemcpyDeviceToHost))
                return rc;
        }
    }
    if (offsets) *offsets = c->h_off;
    if (total) *total = c->h_off[c->last_frames];
    return FLACGPU_OK;
}

int flacgpu_fetch_frames_async(flacgpu_ctx *c, uint8_t *out, size_t cap) {
    if (!c || !c->sizes_pending || !out) {
        g_last_error = "flacgpu_fetch_frames_async: no asynchronous batch in flight";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    if (int rc = wait_waitable(c, c->ev_sizes)) return rc;
    const uint64_t bytes = c->h_off[c->last_frames];
    if (cap < bytes) {
        g_last_error = "output buffer too small";
        return FLACGPU_ERR_BUFFER_TOO_SMALL;
    }
    if (c->out_in_host) {   // k_frame64 wrote them there; the event was recorded behind it
        if (out != c->host_out) {
            if (int rc = wait_waitable(c, c->ev_bytes)) return rc;
            memcpy(out, c->host_out, bytes);
        }
        c->bytes_pending = true;
        return FLACGPU_OK;
    }
    c->prev_total = bytes;
    if (c->early_bytes && out == c->early_dst && c->ties_resolved == 0 && c->fir_rechecked == 0) {
        // the early copy (queued behind the kernels at submission) holds the first early_bytes; the rest now
        if (bytes > c->early_bytes) {
            g_early_remainders.fetch_add(1, std::memory_order_relaxed);
            HIP_TRY(hipMemcpyAsync(out + c->early_bytes, reinterpret_cast<const uint8_t *>(c->d_packed) + c->early_bytes,
                                   bytes - c->early_bytes, hipMemcpyDeviceToHost, c->own_stream));
            if (int rc = record_waitable(c, c->ev_bytes, c->own_stream)) return rc;
        }
        c->bytes_pending = true;
        return FLACGPU_OK;
    }
    HIP_TRY(hipMemcpyAsync(out, c->d_packed, bytes, hipMemcpyDeviceToHost, c->own_stream));
    if (int rc = record_waitable(c, c->ev_bytes, c->own_stream)) return rc;
    c->bytes_pending = true;
    return FLACGPU_OK;
}

int flacgpu_wait(flacgpu_ctx *c) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    if (c->bytes_pending) {
        if (int rc = wait_waitable(c, c->ev_bytes)) return rc;
        c->bytes_pending = false;
        c->sizes_pending = false;
        // the frames were stored by the batch's last kernel (out_in_host) or copied by the last command of the context's own
        // stream, and the sizes' copy on the second stream was waited for by flacgpu_frames_ready: nothing is in flight
        c->drained = c->last_stream == c->own_stream || c->last_stream == nullptr;
        return FLACGPU_OK;
    }
    if (int rc = ctx_sync(c)) return rc;
    return FLACGPU_OK;
}

int flacgpu_experiment_mfma_autocorr(flacgpu_ctx *c, float *kernel_ms, uint32_t *compared,
                                     uint32_t *params_differ, double *max_rel_err) {
    if (!c || c->last_frames == 0 || c->opts.max_lpc_order == 0 || c->opts.max_lpc_order > 16 ||
        c->last_len != c->opts.block_size) {
        g_last_error = "mfma experiment: needs an analysed batch of full blocks with 1 <= max_lpc_order <= 16";
        return FLACGPU_ERR_INVALID_ARG;
    }
    CTX_GUARD(c);
    hipStream_t st = c->own_stream;
    Params p = c->last_params;
    const size_t nc = (size_t)p.n_frames * p.ncand;
    std::vector<LpcParams> exact(nc), mfma(nc);
    std::vector<double> ac_exact(nc * AC_LD), ac_mfma(nc * AC_LD);
    if (int rc = ctx_sync(c)) return rc;
    if (int rc = copy_sync(c, exact.data(), c->d_lpc, sizeof(LpcParams) * nc, hipMemcpyDeviceToHost)) return rc;
    if (int rc = copy_sync(c, ac_exact.data(), c->d_ac, sizeof(double) * nc * AC_LD, hipMemcpyDeviceToHost)) return rc;
    // warm-up + timed launch of the MFMA kernel, writing into the regular ac buffer
    for (int it = 0; it < 2; it++) {
        if (it == 1) (void)hipEventRecord(c->ev[0], st);
        launch_autocorr_mfma(p, (uint32_t)((nc + 3) / 4), p.block_size, c->d_window_full, c->d_ac, st);
        if (it == 1) (void)hipEventRecord(c->ev[1], st);
    }
    launch_lpc(p, c->knobs, (uint32_t)((nc + 63) / 64), st);   // the product's own K4 (no scratch, unlike the generic k_lpc)
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    if (int rc = copy_sync(c, mfma.data(), c->d_lpc, sizeof(LpcParams) * nc, hipMemcpyDeviceToHost)) return rc;
    if (int rc = copy_sync(c, ac_mfma.data(), c->d_ac, sizeof(double) * nc * AC_LD, hipMemcpyDeviceToHost)) return rc;
    // restore the exact results so that the context stays consistent
    if (int rc = copy_sync(c, c->d_lpc, exact.data(), sizeof(LpcParams) * nc, hipMemcpyHostToDevice)) return rc;
    if (int rc = copy_sync(c, c->d_ac, ac_exact.data(), sizeof(double) * nc * AC_LD, hipMemcpyHostToDevice)) return rc;
    uint32_t cmp = 0, diff = 0;
    double worst = 0.0;
    for (size_t i = 0; i < nc; i++) {
        if (exact[i].status != 0 && mfma[i].status != 0) continue;
        cmp++;
        bool d = exact[i].status != mfma[i].status || exact[i].order != mfma[i].order ||
                 exact[i].shift != mfma[i].shift;
        if (!d)
            for (uint32_t j = 0; j < exact[i].order; j++) d |= exact[i].qlp[j] != mfma[i].qlp[j];
        diff += d;
        for (uint32_t l = 0; l <= c->opts.max_lpc_order; l++) {
            double e = ac_exact[i * AC_LD + l], m = ac_mfma[i * AC_LD + l];
            if (e != 0.0) worst = std::max(worst, std::abs((m - e) / e));
        }
    }
    if (kernel_ms) *kernel_ms = ms;
    if (compared) *compared = cmp;
    if (params_differ) *params_differ = diff;
    if (max_rel_err) *max_rel_err = worst;
    return FLACGPU_OK;
}

int flacgpu_verify_device(flacgpu_ctx *c, uint32_t sample_rate, uint64_t first_frame_number,
                          flacgpu_verify_result *result, float *kernel_ms) {
    if (!c || !c->packed_valid || !result) {
        g_last_error = "flacgpu_verify_device: nothing packed";
        return FLACGPU_ERR_INVALID_ARG;
    }
    if (c->out_in_host) {
        g_last_error = "flacgpu_verify_device: the batch was assembled into the caller's host buffer";
        return FLACGPU_ERR_UNSUPPORTED;
    }
    CTX_GUARD(c);
    if (int rc = resolve_order_ties(c)) return rc;
    hipStream_t st = c->own_stream;
    Params p = c->last_params;
    const size_t F = c->max_frames, C = c->channels;
    if (!c->d_decoded) {
        HIP_TRY(hipMalloc((void **)&c->d_decoded, sizeof(int32_t) * (F * C * c->ldb + 64)));
        HIP_TRY(hipMalloc((void **)&c->d_verify, sizeof(uint32_t) * (4 + F)));  // counters + per-frame codes
    }
    PackParams q;
    q.first_frame_number = first_frame_number;
    if (c->seg_active) q.frame_numbers = c->d_seg_fn;
    q.sample_rate = sample_rate;
    q.out_words = c->d_packed;
    q.frame_off = c->d_frame_off;
    q.cap_bytes = c->packed_cap;
    if (int rc = ensure_planar(c)) return rc;
    if (int rc = ctx_sync(c)) return rc;
    HIP_TRY(hipMemsetAsync(c->d_verify, 0, sizeof(uint32_t) * (4 + (size_t)p.n_frames), st));
    // compare against the planar PCM the analysis consumed: the context's copy, or the rows it read in place
    // (planar / one-channel input: the caller's buffer is still valid here, include/flacenc_gpu.h)
    const int32_t *expect = (p.ldb == c->ldb) ? p.planar : nullptr;
    Params pd = p;
    pd.ldb = c->ldb;
    (void)hipEventRecord(c->ev[0], st);
    // the CRC-16 check only shares the input with the decoder, which leaves half the SIMDs idle:
    // it runs beside it on the context's second stream
    HIP_TRY(hipEventRecord(c->ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
    launch_crc(true, p, q, p.n_frames, c->d_verify, c->aux_stream);
    HIP_TRY(hipEventRecord(c->ev_join, c->aux_stream));
    {
        const uint32_t units = p.n_frames * p.channels;
        uint32_t lanes = 32;
        if (const uint32_t v = c->knobs.decode_lanes) {  // experiment knob
            if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) lanes = v;
        }
        launch_decode(c->opts.max_lpc_order, units, lanes, pd, q, c->d_decoded, c->d_verify, st);
        launch_decode_finish(pd, c->d_decoded, expect, c->d_verify, st);
    }
    HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
    (void)hipEventRecord(c->ev[1], st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    uint32_t counts[4];
    if (int rc = copy_sync(c, counts, c->d_verify, sizeof counts, hipMemcpyDeviceToHost)) return rc;
    result->frames = p.n_frames;
    result->bad_structure = counts[0];
    result->bad_crc16 = counts[1];
    result->frames_pcm_differs = counts[2];
    result->samples_differ = counts[3];
    result->compared_pcm = expect != nullptr;
    if (kernel_ms) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        *kernel_ms = ms;
    }
    return FLACGPU_OK;
}

int flacgpu_fetch_decoded(flacgpu_ctx *c, int32_t *interleaved) {
    if (!c || !c->d_decoded || !interleaved || c->last_frames == 0) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    const size_t F = c->last_frames, C = c->channels, B = c->opts.block_size, ldb = c->ldb;
    std::vector<int32_t> planar(F * C * ldb);
    if (int rc = ctx_sync(c)) return rc;
    if (int rc = copy_sync(c, planar.data(), c->d_decoded, sizeof(int32_t) * planar.size(), hipMemcpyDeviceToHost)) return rc;
    size_t o = 0;
    for (size_t f = 0; f < F; f++) {
        const size_t n = (f + 1 == F) ? c->last_len : B;
        for (size_t i = 0; i < n; i++)
            for (size_t ch = 0; ch < C; ch++) interleaved[o++] = planar[(f * C + ch) * ldb + i];
    }
    return FLACGPU_OK;
}

int flacgpu_resolve(flacgpu_ctx *c) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    if (int rc = ctx_sync(c)) return rc;
    return resolve_order_ties(c);
}

int flacgpu_get_stats(flacgpu_ctx *c, flacgpu_stats *out) {
    if (!c || !out) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    uint32_t s[4];
    if (int rc = ctx_sync(c)) return rc;
    if (int rc = copy_sync(c, s, c->d_stats, sizeof s, hipMemcpyDeviceToHost)) return rc;
    out->frames = c->last_frames;
    out->lpc_failed = s[0];
    out->order_ties = s[1];
    out->log2_edge = s[2];
    out->order_ties_resolved = c->ties_resolved;
    out->fir_recheck = s[3];
    out->fir_rechecked = c->fir_rechecked;
    uint32_t d[kDeferWords];
    if (int rc = copy_sync(c, d, defer_stats_of(c), sizeof d, hipMemcpyDeviceToHost)) return rc;
    uint64_t decided = 0, refetched = 0;   // (64 counter lines of 32 bits each: summed wide, handed out saturated)
    for (uint32_t i = 0; i < DEFER_SLOTS; i++) {
        decided += d[DEFER_SLOT_WORDS * i];
        refetched += d[DEFER_SLOT_WORDS * i + 1];
    }
    out->fixed_decided = (uint32_t)std::min<uint64_t>(decided, 0xFFFFFFFFull);
    out->fixed_refetched = (uint32_t)std::min<uint64_t>(refetched, 0xFFFFFFFFull);
    return FLACGPU_OK;
}

void *flacgpu_device_buffer(flacgpu_ctx *c, int which) {
    if (!c) return nullptr;
    CTX_GUARD(c);
    switch (which) {
    case 0: return c->d_fplan;
    case 1: return c->d_out;
    case 2: return ensure_residual_rows(c) == FLACGPU_OK ? c->d_resid : nullptr;
    case 3: return ensure_planar(c) == FLACGPU_OK ? c->d_planar : nullptr;
    case 4: return c->d_packed;
    case 5: return c->d_frame_off;
    case 6: return c->d_ac;
    default: return nullptr;
    }
}

// Diagnostic: of the last batch's subframes, how many k_cand64p handed to k_frame64 with their residual (Params::hand_meta).
// *enabled = 0 and zeros where the batch's shape has no hand-over (or FLACGPU_NO_HAND is set, or residual rows were fetched).
int flacgpu_handed_subframes(flacgpu_ctx *c, uint32_t *handed, uint32_t *subframes, int *enabled) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    CTX_GUARD(c);
    if (int rc = ctx_sync(c)) return rc;
    const bool on = c->last_frames && c->last_params.hand_meta;
    uint32_t n = 0;
    const uint32_t total = on ? c->last_frames * 2u : 0u;
    if (on) {
        std::vector<uint32_t> flags(total);
        if (int rc = copy_sync(c, flags.data(), c->d_hand, sizeof(uint32_t) * total, hipMemcpyDeviceToHost)) return rc;
        for (uint32_t v : flags) n += v ? 1u : 0u;
    }
    if (handed) *handed = n;
    if (subframes) *subframes = total;
    if (enabled) *enabled = on ? 1 : 0;
    return FLACGPU_OK;
}

int flacgpu_get_kernel_ms(flacgpu_ctx *c, float ms[FLACGPU_N_KERNELS]) {
    if (!c || !ms) return FLACGPU_ERR_INVALID_ARG;
    memcpy(ms, c->last_ms, sizeof(float) * FLACGPU_N_KERNELS);
    return FLACGPU_OK;
}

}  // extern "C"
