// types.h -- records, kernel parameters and launcher declarations shared by the translation units
// of libflacenc_amd.so.  The kernels are split over several .hip files only so that they compile in
// parallel; every kernel lives in an anonymous namespace of its own file and is reached through
// the launchers declared at the end of this header.
#ifndef FLACGPU_KERNEL_TYPES_H
#define FLACGPU_KERNEL_TYPES_H
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <string>
#include <type_traits>

#include "flacenc_gpu.h"


constexpr int WG = 256;          // threads per workgroup (4 wave64)
constexpr int MAXP = 6;          // max effective partition order (64 partitions, encode.rs:3756)
constexpr int NLEAF = 1 << MAXP;
constexpr int NNODE = 2 * NLEAF - 1;

extern thread_local std::string g_last_error;  // defined in flacenc_gpu.hip

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return FLACGPU_ERR_HIP;                                                     \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------
// device-side records
// ---------------------------------------------------------------------------------
typedef flacgpu_subframe_plan SubPlan;  // same layout on both sides of the ABI

struct CandInfo {        // per (frame, candidate)
    uint8_t active;      // 0: not a candidate for this frame (fast correlation / no mid)
    uint8_t wasted;
    uint8_t bps;         // effective bps after wasted-bit removal
    uint8_t is_const;    // all samples zero -> CONSTANT, nothing else to analyse
};

struct LpcParams {       // per (frame, candidate), output of k_lpc
    int32_t status;      // 0 ok; else the reference's error (1 Insufficient, 2 NoBestOrder,
                         // 3 ZeroCoeffs, 4 NegativeShift)
    uint8_t order, precision, shift;
    uint8_t est8;        // compute_best_order's size estimate of the chosen order in 1/8 bit per sample (1..255; 0: none --
                         // the host's re-decision leaves none); a HINT for Params::defer_fixed, never part of a decision
    int32_t qlp[FLACGPU_MAX_LPC_ORDER];
};

struct FrameInfo {       // per frame, from k_stereo_stats (fast mode) -- preset assignment
    uint8_t assignment;
    uint8_t pad[3];
};

struct Params {
    // stream shape / options
    uint32_t channels, bps, block_size, ldb;      // ldb = row stride of the planar buffer
    uint32_t ncand;                                // candidate slots per frame
    uint32_t stereo4;                              // 1: slots are L,R,M,S
    uint32_t mid_side, exhaustive;
    uint32_t max_lpc_order, max_po, use_rice2;
    uint32_t n_frames, last_len;
    uint32_t f0, fcount;                           // frames [f0, f0 + fcount) handled by this launch
    uint32_t ac_split;                             // waves the lags of k_autocorr3 are split over (2 or 4)
    // k_autocorr4 / _deep: 32-sample tiles [fma_t0, fma_t1) of a FULL block whose every product has both factors under window
    // values of exactly 1.0 (and integer samples below 2^26): there a term is one v_fma_f64, bit-identical to the reference's
    // multiply + add because the product is exact.  fma_t0 >= fma_t1: nowhere (short last frame, other windows, wide samples)
    uint32_t fma_t0, fma_t1;
    // buffers
    const int32_t *planar;
    // DIRECT input: the batch's interleaved stereo PCM ([frame][sample][l, r], every frame of block_size
    // samples) read by k_autocorr4 / k_cand64p / k_frame64 themselves -- no K0 split, `planar` is not
    // written.  nullptr: the kernels read `planar`
    const int32_t *inter;
    // ... or, a batch made of SEGMENTS (runs of whole blocks of several streams, flacgpu_encode_segments_device): the
    // address of every frame's PCM, [n_frames]; nullptr: frame f starts at inter + f * 2 * block_size (inter_frame below)
    const int32_t *const *inter_tab;
    // SPLIT input: a batch of interleaved INDEPENDENT channels ([frame][sample][channel]) that k_autocorr4's producers
    // split into the planar rows (split_dst = `planar`) while they read it -- no k_deinterleave_n pass; nullptr: not used
    // split_dst == nullptr with several channels: nobody wants the rows -- XPOSE below
    const int32_t *split_src;
    int32_t *split_dst;
    // XPOSE (with split_src; 3, 4, 6 or 8 channels): k_cand64 and k_sub64 read the interleaved batch in place as well
    // (load_lane_xpose), the planar rows are never written
    uint32_t xpose;
    const double *window_full, *window_last;
    const double *log2_thr;                        // [128], index e + 64
    CandInfo *cinfo;
    SubPlan *fixed_plan, *cand_plan, *out_plan;
    LpcParams *lpc;
    double *ac;                                    // [n_frames*ncand][36]
    FrameInfo *finfo;
    flacgpu_frame_plan *frame_plan;
    int32_t *residuals;                            // [n_frames][channels][block_size]
    uint32_t *stats;                               // [4]
    int32_t *big_scratch;                          // blocks > LDS_BLOCK_LIMIT: per-workgroup arrays in HBM
    uint32_t big_stride;                           // ints per workgroup in big_scratch
    // LPC order choice: candidates whose two best estimates lie within tie_band (relative) are listed
    // (stats[1] counts them) and re-decided on the host with its libm; tie_perturb is a TEST knob
    // (0 in production) that skews the device's estimates inside the band
    uint32_t *tie_list;
    uint32_t tie_cap;
    double tie_band, tie_perturb;
    // ResidualOverflow (encode.rs:3190-3197) in the wave kernels.  0 (every first analysis): the in-place FIR runs
    // unchecked and the fold of its residual tells whether an overflow was POSSIBLE -- a wrapped x - pred of a <= 25-bit
    // sample has magnitude >= 2^31 - 2^24, so a lane whose sum of folded residuals stays below 2^30 has none; a wave
    // that cannot rule it out counts itself in stats[3] and the host has the candidate stage run again with
    // check_fir = 1 (resolve_order_ties), where every candidate takes the exact read-only test fir64_overflows first.
    uint32_t check_fir;
    uint32_t fir_suspect_bits;   // 30; a TEST knob lowers it so that ordinary input exercises the re-run
    // The FIXED half's partition tree and exact bit count (encode.rs:3862-3947 for the FIXED residual) put off until the LPC
    // candidate's size is known (k_cand64p, direct stereo input) -- r05.  From the 64 leaf sums the order statistics hold a
    // LOWER BOUND of the FIXED residual block's size follows at any partition order (wave_cand_fixed.inc).  When k_lpc's size
    // estimate of the LPC candidate (LpcParams::est8, 1/8 bit per sample) undercuts that bound by defer_margin16 / 16 bit per
    // sample, the wave runs the LPC half first; an exact lpc_bits < bound then DECIDES encode.rs:2929-2934 for LPC without
    // tree or count, otherwise the wave builds the tree, re-fetches its samples and counts after all.  0: never (A/B), 1: by
    // the estimate, 2: whenever LPC parameters exist (TEST: every undecided candidate takes the re-fetch).
    // defer_stats: {waves that deferred, of those: re-fetched}, cumulative.
    uint32_t defer_fixed, defer_margin16;
    uint32_t *defer_stats;   // [DEFER_SLOTS][DEFER_SLOT_WORDS]: the pair {deferred, re-fetched} per slot
    // k_cand64p's DYNAMIC TURNS (r05): the launch's ticket counter -- 0 when a launch begins (zeroed with the batch's
    // counters by the host, and again by whoever draws a launch's last ticket); launches of one context that may run
    // concurrently (the two ranges of FLACGPU_TUNE_TWO_RANGES) use different words
    uint32_t *turn_counter;
    // THE RESIDUAL HANDED OVER (r06; 4096-sample direct stereo frames with LPC): the candidate wave that wins a subframe with its
    // LPC candidate still holds that candidate's folded residual in registers when the channel choice is made -- t = r ^ (r >> 31)
    // with r's sign kept in bit 31 (wave_rice_fold<SIGNS>): zigzag(r) rotated right by one.  It stores the 64 x 64 words into the
    // subframe's row of Params::residuals (sample e of lane l at dword 256 (e / 4) + 4 l + e % 4: a contiguous kilobyte per
    // instruction on both sides) and hand_meta[subframe] = 1; k_frame64 reads them back instead of fetching both channels, picking, shifting and
    // running the FIR a second time.  nullptr: off (every other shape; plans that came from the host).  A subframe whose winner
    // is FIXED / CONSTANT / VERBATIM, or whose wave had to re-fetch its samples for the exact FIXED count, gets 0 and takes
    // k_frame64's own path.
    uint32_t *hand_meta;   // [frame][2]
};


// the interleaved stereo PCM of frame `frame` of a DIRECT batch (frame_dwords = 2 * block_size)
__device__ __forceinline__ const int32_t *inter_frame(const Params &p, uint32_t frame, uint32_t frame_dwords) {
    return p.inter_tab ? p.inter_tab[frame] : p.inter + (size_t)frame * frame_dwords;
}
// ... of frame `frame` of a SPLIT / XPOSE batch of interleaved independent channels (frame_dwords = block_size * channels)
__device__ __forceinline__ const int32_t *split_frame(const Params &p, uint32_t frame, uint32_t frame_dwords) {
    return p.inter_tab ? p.inter_tab[frame] : p.split_src + (size_t)frame * frame_dwords;
}

// largest block the generic kernels keep whole in LDS (k_fixed / k_fir: two arrays of it); larger
// blocks (up to 65535, encode.rs:1418-1423) use per-workgroup arrays in HBM instead
constexpr uint32_t LDS_BLOCK_LIMIT = 16384;
__host__ __device__ constexpr uint32_t big_scratch_ints(uint32_t block_size) {
    return 2u * (block_size + block_size / 16u + 16u) + 64u;   // x[n] | r[n] with the RIDX padding
}
constexpr uint32_t DEFER_SLOTS = 64, DEFER_SLOT_WORDS = 16;   // Params::defer_stats: workgroup w adds to slot w % 64 (a 64-byte line each)
constexpr int AC_LD = 36;        // row stride of the ac buffer (max lag group count rounded up)
constexpr uint32_t FN = 4096;    // the block length of every preset but `fast`

struct PackParams {
    uint64_t first_frame_number;
    // a batch made of segments: the frame number of every frame, [n_frames]; nullptr: first_frame_number + f
    const uint64_t *frame_numbers = nullptr;
    uint32_t sample_rate;
    uint32_t *out_words;      // packed bytes, viewed as big-endian-filled 32-bit words
    uint64_t *frame_off;      // [n_frames + 1] byte offsets
    uint64_t cap_bytes;
    // k_layout's tile exchange: one word per 1024-frame tile, (epoch << 40) | bytes of the tile; the epoch is new for
    // every launch, so the words are never reset
    unsigned long long *tile_sync = nullptr;
    uint32_t epoch = 0;
    // k_sub64 (one workgroup per subframe, kernels/sub64.inc): the subframes' edge records [frame][channel], merged by
    // k_sub_finish; nullptr: the frame-per-workgroup k_frame64 assembles every wave-kernel frame
    struct SubEdgeRec { uint32_t w[8]; } *edges = nullptr;
    // k_frame64: words of the workgroup's LDS image as launched (a multiple of 4; set by launch_frame64_nt) -- the image is
    // zeroed whole, without waiting for the frame's length to arrive from memory; 0: zero the frame's own words only
    uint32_t fb_words = 0;
};
__device__ __forceinline__ uint64_t frame_number_of(const PackParams &q, uint32_t frame) {
    return q.frame_numbers ? q.frame_numbers[frame] : q.first_frame_number + frame;
}


// words reserved for a subframe's bit string: a chosen subframe is never longer than its
// VERBATIM form (<= 40 + 33 n bits) plus the 16-byte frame header; multiple of 4 words
__host__ __device__ constexpr uint32_t pack_sb_words(uint32_t block_size) {
    return ((block_size * 33u / 32u + 32u) + 3u) & ~3u;
}

// words of LDS a whole frame of 4096-sample subframes may need (VERBATIM everywhere + header +
// CRC-16 + one guard word for the funnel shifts); multiple of 4 words
__host__ __device__ constexpr uint32_t frame_fb_words(uint32_t channels, uint32_t bps, uint32_t n = FN) {
    return (((16u + 2u) * 8u + channels * (n * (bps + 1u) + 64u) + 31u) / 32u + 2u + 3u) & ~3u;
}


// Stereo frames whose channel assignment is chosen EXHAUSTIVELY (first minimum of the candidate pairs' bit counts,
// encode.rs:2747-2786): the chosen pair is never larger than L + R, and a chosen subframe never larger than its
// VERBATIM form, so the frame body is bounded by two bps-bit VERBATIM subframes -- not by the bps + 1 bits of a side
// channel.  For 24-bit stereo that is 24.6 KB instead of 25.6 KB of LDS per k_frame64 workgroup: six workgroups per
// CU instead of five.  (The fast channel choice picks its pair before any bit count exists: frame_fb_words.)
__host__ __device__ constexpr uint32_t frame_fb_words_exhaustive_stereo(uint32_t bps, uint32_t n = FN) {
    return (((16u + 2u) * 8u + 2u * (n * bps + 64u) + 31u) / 32u + 2u + 3u) & ~3u;
}

// block lengths the wave kernels are instantiated for: 64 lanes x SPL samples
#define FLACGPU_WAVE_SIZES(X) X(4096, 64) X(2304, 36) X(2048, 32) X(1152, 18) X(1024, 16)

// Every FLACGPU_* environment knob, resolved ONCE per context (flacgpu_create; flacenc_gpu.hip read_knobs) -- the
// dispatch path never calls getenv.  The A/B selectors keep an older or generic kernel reachable for measurements
// and parity tests; the TEST knobs (tie_band, tie_perturb, experiment_mfma_ac, decode_lanes) are honoured only
// when FLACGPU_TEST_KNOBS=1 is set as well.
struct Knobs {
    bool no_direct = false, no_fast = false, no_w64 = false, no_persist = false, no_ac3 = false, ac_private = false,
         no_fused_pack = false, no_frame64 = false, no_fork = false, lpc_dyn = false,
         ac_eight_waves = false,   // A/B: k_autocorr4<13, 8> (FLACGPU_AC_WAVES8)
         cand_persist_n = false,   // A/B: persistent candidate kernel for independent channels (FLACGPU_CAND_PERSIST_N)
         early_download = false,   // the frames' D2H copy queued before the sizes are known (FLACGPU_EARLY_DOWNLOAD)
         no_hand = false,          // A/B: Params::hand_meta off (FLACGPU_NO_HAND)
         no_direct_short = false,  // A/B: 1024 / 1152 / 2048 / 2304-sample blocks through K0 + k_cand64 (FLACGPU_NO_DIRECT_SHORT)
         lpc_fuse_deep = false,    // A/B: K4 in the tail of k_autocorr4_deep as well (FLACGPU_LPC_FUSE_DEEP; measured slower)
         no_ac_fma = false,        // A/B: every autocorrelation term a multiply and an add (FLACGPU_NO_AC_FMA)
         no_chunk = false,         // A/B: big in-place batches run as ONE range (FLACGPU_NO_CHUNK)
         no_xpose = false,         // A/B: 4 / 8 interleaved channels split into planar rows by k_autocorr4 instead of read in place (FLACGPU_NO_XPOSE)
         no_lpc_fuse = false,      // A/B: K4 as a launch of its own behind the direct autocorrelation (FLACGPU_NO_LPC_FUSE)
         no_sub64 = false,         // A/B: frames of 3..8 channels assembled by one workgroup per FRAME (k_frame64) (FLACGPU_NO_SUB64)
         no_cand_pair = false;     // A/B: four waves per frame also for the fast channel choice without LPC (FLACGPU_NO_CAND_PAIR)
    bool upload_by_kernel = false;      // A/B: the asynchronous host path reads the caller's pinned PCM with kernel loads (FLACGPU_UPLOAD_KERNEL)
    bool force_fir_check = false;       // A/B + TEST: every candidate takes the exact ResidualOverflow test first (FLACGPU_FIR_CHECK)
    uint32_t cand_grid = 0;             // resident workgroups of the persistent candidate kernels, 0: default
    bool experiment_mfma_ac = false;    // TEST: the re-associating MFMA autocorrelation (not bit-exact)
    bool has_tie_band = false, has_tie_perturb = false;
    double tie_band = 0.0, tie_perturb = 0.0;   // TEST
    uint32_t decode_lanes = 0;          // TEST: lanes per wave of the stand-alone decoder, 0: default
    uint32_t fir_suspect_bits = 0;      // TEST: Params::fir_suspect_bits, 0: default (30)
    int defer_fixed = -1;               // FLACGPU_DEFER_FIXED: 0 / 1 / 2 (Params::defer_fixed); -1: default (1)
    int defer_margin16 = -1;            // FLACGPU_DEFER_MARGIN16: Params::defer_margin16; -1: default
};
Knobs read_knobs();   // flacenc_gpu.hip

// ---- launchers (one per kernel family; defined in the .hip file that holds the kernels) ----
namespace flacgpu_k {
// lpc.hip
void launch_lpc(const Params &p, const Knobs &kn, uint32_t blocks, hipStream_t st);
// cand.hip
// returns true when the kernel also chose the channel assignment and wrote out_plan / frame_plan (the
// persistent stereo kernels: K6 in the workgroup) -- no k_decide launch for those frames then
bool launch_cand64(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st);
// cand_direct.hip
bool launch_cand64_direct(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st);   // true: channel choice made
bool launch_cand64_direct_short(const Params &p, const Knobs &kn, uint32_t B, uint32_t blocks, hipStream_t st);
// autocorr.hip
// returns true when the kernel also ran K4 in its tail (the DIRECT k_autocorr4 instantiations): no launch_lpc then
bool dispatch_autocorr(uint32_t H, const Params &p, const Knobs &kn, uint32_t frame0, uint32_t nframes, uint32_t n,
                       const double *win, hipStream_t st);
void launch_autocorr_mfma(const Params &p, uint32_t blocks, uint32_t n, const double *win, double *ac,
                          hipStream_t st);
// pack.hip
void launch_layout(const Params &p, const PackParams &q, hipStream_t st);
void launch_zero(const PackParams &q, uint32_t n_frames, hipStream_t st);
void launch_pack(const Params &p, const PackParams &q, uint32_t blocks, size_t lds, hipStream_t st);
void launch_crc(bool verify, const Params &p, const PackParams &q, uint32_t frames, uint32_t *verify_counts,
                hipStream_t st);
void launch_frame64(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds,
                    hipStream_t st);
hipError_t pack_set_attributes(size_t pack_lds);
// frame64_a.hip / frame64_b.hip / frame64_c.hip
void launch_frame64_4096(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st);
void launch_frame64_direct(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds, hipStream_t st);
void launch_frame64_deep(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st);
// sub64.hip: one workgroup per subframe (independent channels, 4096-sample blocks, LPC order <= 16)
void launch_sub64(const Params &p, const PackParams &q, uint32_t frames, hipStream_t st);
void launch_frame64_short(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds,
                          hipStream_t st);
// decode.hip
void launch_decode(uint32_t max_lpc_order, uint32_t units, uint32_t lanes, const Params &p, const PackParams &q,
                   int32_t *decoded, uint32_t *verify_counts, hipStream_t st);
void launch_decode_finish(const Params &p, int32_t *decoded, const int32_t *expect, uint32_t *verify_counts,
                          hipStream_t st, const uint32_t *frame_n = nullptr);
void launch_decode_frames(const uint32_t *words, const uint64_t *frame_off, const uint32_t *frame_n,
                          uint64_t cap_bytes, uint32_t n_frames, uint32_t channels, uint32_t bps, uint32_t ldb,
                          int32_t *decoded, uint32_t *verify_counts, hipStream_t st);
}  // namespace flacgpu_k
#endif
