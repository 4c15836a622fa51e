/*
 * flacenc_gpu.h -- C ABI of the MI355X-native FLAC encode hot path.
 *
 * This is the drop-in boundary.  The reference
 * (tuffy/flac-codec 1.3.2) is 100 % safe Rust with no FFI, so the boundary is
 * the narrowest internal call that separates per-frame ANALYSIS from stream
 * bookkeeping: the analysis half of `encode_frame`
 * (/root/reference/src/encode.rs:2259-2406) -- i.e. correlate_channels[_exhaustive]
 * (:2463-2847), encode_subframe (:2849-2980), encode_fixed_subframe (:3020-3088),
 * LpcParameters::best / encode_residuals (:3145-3332), write_residuals' partition
 * search (:3747-3962) -- for a BATCH of frames instead of one.  In the reference
 * the result of that half is up to 8 `BitRecorder`s per frame; here it is a
 * decision record per subframe plus the residual signal, or (flacgpu_pack_*) the
 * finished frame bytes.
 *
 * A Rust maintainer binds these with `extern "C"` inside `Encoder::encode`
 * (encode.rs:1997-2022); see INTEGRATION.md for the stub.
 *
 * Plain pointers and sizes only.  All functions return 0 on success or a
 * negative FLACGPU_ERR_*.  A context is NOT thread-safe (one per stream or per
 * GPU shard of a stream), exactly like the reference's `&mut Encoder`.
 * Internal analysis failures of the reference (InsufficientLpcSamples,
 * NoBestLpcOrder, ZeroLpCoefficients, LpNegativeShiftError, ResidualOverflow)
 * are never surfaced: as in encode.rs:2929-2968 they only steer the
 * FIXED / VERBATIM fallback.
 */
#ifndef FLACENC_GPU_H
#define FLACENC_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLACGPU_MAX_CHANNELS 8
#define FLACGPU_MAX_LPC_ORDER 32
#define FLACGPU_MAX_PARTITIONS 64
#define FLACGPU_MAX_BLOCK_SIZE 65535 /* every block size the reference accepts (encode.rs:1418-1423); blocks
                                        above 16384 take kernels that keep their arrays in HBM, not LDS */

enum {
    FLACGPU_OK = 0,
    FLACGPU_ERR_INVALID_ARG = -1,
    FLACGPU_ERR_UNSUPPORTED = -2, /* legal for the reference but not for this build (or the
                                     reference itself would panic, e.g. > 64 partitions,
                                     encode.rs:3756/3880) */
    FLACGPU_ERR_HIP = -3,         /* HIP runtime failure; flacgpu_last_error() has the text */
    FLACGPU_ERR_NO_DEVICE = -4,
    FLACGPU_ERR_BUFFER_TOO_SMALL = -5,
    FLACGPU_ERR_BUSY = -6,        /* flacgpu_pipeline_submit: every slot holds a batch, retire one first */
    FLACGPU_ERR_NO_RCCL = -7      /* flacgpu_rccl_allgather_counters: librccl could not be loaded (no fallback) */
};

/* encode.rs:1713-1720 `Window` */
enum { FLACGPU_WINDOW_RECTANGLE = 0, FLACGPU_WINDOW_HANN = 1, FLACGPU_WINDOW_TUKEY = 2 };

/* Mirror of `EncoderOptions` (encode.rs:1701-1709) + Options::block_size (:1366).
 * use_rice2 is derived exactly like encode.rs:1965 (stream bits_per_sample > 16). */
typedef struct {
    uint32_t block_size;          /* 16..=65535 */
    uint32_t max_partition_order; /* 0..=15 (effective order is capped at 6, see above) */
    uint32_t max_lpc_order;       /* 0 = None, else 1..=32 */
    uint8_t mid_side;
    uint8_t exhaustive_channel_correlation;
    uint8_t window_kind;          /* FLACGPU_WINDOW_* */
    uint8_t reserved;
    float window_param;           /* Tukey p */
} flacgpu_options;

/* subframe types (stream.rs:1452-1460) */
enum { FLACGPU_SUB_CONSTANT = 0, FLACGPU_SUB_VERBATIM = 1, FLACGPU_SUB_FIXED = 2, FLACGPU_SUB_LPC = 3 };

/* channel assignment as coded in the frame header (stream.rs:996-1010) */
enum {
    FLACGPU_ASSIGN_INDEPENDENT = 0,
    FLACGPU_ASSIGN_LEFT_SIDE = 8,
    FLACGPU_ASSIGN_SIDE_RIGHT = 9,
    FLACGPU_ASSIGN_MID_SIDE = 10
};

/* candidate a subframe was computed from */
enum { FLACGPU_SRC_MID = 8, FLACGPU_SRC_SIDE = 9 }; /* 0..7 = input channel */

/* Decision record of one subframe: everything `BitRecorder::playback`
 * (encode.rs:2332-2333) needs except the residual values themselves. */
typedef struct {
    uint8_t type;            /* FLACGPU_SUB_* */
    uint8_t wasted;          /* wasted bits per sample (encode.rs:2878-2898) */
    uint8_t bps;             /* effective bits per sample of this subframe */
    uint8_t order;           /* FIXED 0..4 / LPC 1..32 */
    uint8_t precision;       /* LPC coefficient precision (encode.rs:3305-3315) */
    uint8_t shift;           /* LPC shift (encode.rs:3360) */
    uint8_t coding_method;   /* 0 = RICE (4-bit), 1 = RICE2 (5-bit) (encode.rs:3929-3942) */
    uint8_t partition_order; /* as written: ilog2(n_partitions) (encode.rs:3902) */
    uint8_t source;          /* 0..7 input channel, FLACGPU_SRC_MID, FLACGPU_SRC_SIDE */
    uint8_t reserved[3];
    uint32_t n_partitions;   /* partitions actually emitted */
    uint32_t part_len;       /* residual chunk length; the first chunk is short (encode.rs:3876-3879) */
    uint32_t bits;           /* exact subframe length in bits == BitRecorder::written() */
    int32_t coeffs[FLACGPU_MAX_LPC_ORDER];
    uint8_t rice[FLACGPU_MAX_PARTITIONS];        /* Rice parameter; 0xFF = escaped / constant */
    uint8_t escape_bits[FLACGPU_MAX_PARTITIONS]; /* escape size, 0 = all-zero partition */
} flacgpu_subframe_plan;

typedef struct {
    uint8_t assignment; /* FLACGPU_ASSIGN_* */
    uint8_t channels;
    uint16_t block_size; /* samples per channel in this frame */
    uint32_t body_bits;  /* sum of the subframes' bits (frame = header + ceil(body/8) + 2) */
} flacgpu_frame_plan;

/* diagnostic counters of the last analyze call */
typedef struct {
    uint32_t frames;
    uint32_t lpc_failed;        /* candidates whose LPC path errored (fell back to FIXED) */
    uint32_t order_ties;        /* candidates whose two best LPC-order estimates were within
                                   1e-9 relative: inside that band the device's log() is not
                                   trusted to pick like the host libm the reference uses; they are
                                   re-decided on the host (see order_ties_resolved) */
    uint32_t log2_edge;         /* quantise calls where max|c| sat in the libm-sensitive band
                                   just below a power of two (handled by the host table) */
    uint32_t order_ties_resolved; /* of order_ties: candidates whose LPC parameters were recomputed on
                                   the host with its libm (encode.rs:3656-3702) before the results
                                   were handed out; every fetch / verify entry point does this */
    uint32_t fir_recheck;       /* candidates whose LPC residual, computed WITHOUT the per-sample overflow test of
                                   encode.rs:3190-3197, was too large to rule an overflow out (a lane's sum of
                                   folded residuals >= 2^30: garbage predictions only) */
    uint32_t fir_rechecked;     /* of fir_recheck: re-analysed with the exact test before the results were handed
                                   out (the same entry points as order_ties_resolved) */
    /* CUMULATIVE since the context was created (direct stereo input with LPC): candidates whose FIXED partition search and
     * exact bit count were put off behind the LPC half because the LPC size estimate undercut a lower bound of the
     * FIXED size -- fixed_decided: the exact LPC size then lay below the bound, encode.rs:2929-2934 was decided for LPC
     * without them; fixed_refetched: it did not, the samples were fetched again and counted.  Same bytes either way.
     * Both saturate at 2^32 - 1 (diagnostics of long-lived contexts, not arithmetic). */
    uint32_t fixed_decided;
    uint32_t fixed_refetched;
} flacgpu_stats;

typedef struct flacgpu_ctx flacgpu_ctx;

/* PCM layouts accepted by analyze */
enum {
    FLACGPU_LAYOUT_INTERLEAVED = 0, /* [pcm_frame][channel]  (FlacSampleWriter::write, encode.rs:558) */
    FLACGPU_LAYOUT_PLANAR = 1       /* [flac_frame][channel][block_size] (Frame, audio.rs:190-199) */
};

/* Create a context for one stream shape.  Replaces `EncodingCaches`
 * (encode.rs:1810-1851): all scratch lives in HBM, sized for `max_frames`
 * FLAC frames per call (at most 2 097 152: eight frame-layout chunks of 256 x 1024 frames).
 * device < 0 selects the current HIP device. */
int flacgpu_create(const flacgpu_options *opts, uint32_t bits_per_sample, uint32_t channels,
                   int device, uint32_t max_frames, flacgpu_ctx **out);
void flacgpu_destroy(flacgpu_ctx *ctx);
const char *flacgpu_last_error(void);

/* Analyse `n_frames` FLAC frames (all `block_size` long except the last, which is
 * `last_frame_len` samples per channel, 1..=block_size).
 *   pcm        host pointer, layout as given; samples must fit bits_per_sample
 *   plans      out, [n_frames]
 *   subframes  out, [n_frames * channels], subframe c of frame f at f*channels + c
 *   residuals  out, [n_frames][channels][block_size] int32: for FIXED/LPC subframes the
 *              `order` warm-up samples followed by the n-order residuals; for VERBATIM
 *              the n samples; for CONSTANT the sample at [0] (all after wasted-bit removal)
 * Synchronous, like the reference's encode_frame. */
int flacgpu_analyze(flacgpu_ctx *ctx, const int32_t *pcm, int layout, uint32_t n_frames,
                    uint32_t last_frame_len, flacgpu_frame_plan *plans,
                    flacgpu_subframe_plan *subframes, int32_t *residuals);

/* Same, PCM already resident in device memory (d_pcm is a device pointer).  Results stay
 * on the device until flacgpu_fetch().  Asynchronous with respect to the host.
 * `stream` is a hipStream_t.  NULL selects the context's own non-blocking stream, ordered AFTER
 * everything the caller has already submitted to the legacy default stream (so a d_pcm produced
 * there is safe to hand over); the results are then ordered on the context's stream, which every
 * flacgpu_fetch* / flacgpu_get_stats / flacgpu_wait call synchronises.  A caller that chains further
 * device work on the results passes its own stream instead.
 * Every entry point makes the context's device current for the duration of the call and restores
 * the caller's device afterwards.
 * Lifetime of d_pcm: interleaved stereo PCM of whole blocks of 1024, 1152, 2048, 2304 or 4096 samples (<= 24 bits), interleaved
 * 3-, 4-, 6- and 8-channel PCM of whole 4096-sample blocks (LPC order 1..16), one-channel PCM and planar PCM are read IN PLACE by the
 * analysis and frame kernels -- the
 * context keeps no copy (no K0 split pass; this is what the reference's `write(&[i32])` hands over,
 * encode.rs:558).  The buffer must therefore stay valid and unchanged until the batch's results have
 * been fetched (flacgpu_fetch* / flacgpu_verify_device / flacgpu_pack_device included) or the next
 * batch is submitted to this context.  flacgpu_set_tuning(ctx, FLACGPU_TUNE_COPY_INPUT, 1) (or FLACGPU_NO_DIRECT=1
 * in the environment) restores the copy and with it the old rule: free once the submitted work has run. */
int flacgpu_analyze_device(flacgpu_ctx *ctx, const int32_t *d_pcm, int layout, uint32_t n_frames,
                           uint32_t last_frame_len, void *stream);
/* Copy the results of the last flacgpu_analyze_device to host buffers (any may be NULL). */
int flacgpu_fetch(flacgpu_ctx *ctx, flacgpu_frame_plan *plans, flacgpu_subframe_plan *subframes,
                  int32_t *residuals);
int flacgpu_get_stats(flacgpu_ctx *ctx, flacgpu_stats *out);
/* Diagnostic (bench, tests): of the last batch's subframes, how many the candidate kernel handed to the frame kernel together
 * with their residual -- 4096-sample stereo frames read in place, LPC on: the wave that wins a subframe with an LPC candidate
 * stores the folded residual it still holds, and the frame kernel neither fetches the channels nor runs the FIR again
 * (encode.rs:3174-3203 is evaluated once per winning subframe instead of twice).  *enabled = 0 where the batch's shape has no
 * hand-over, FLACGPU_NO_HAND=1 is set, or the residual rows were fetched in between.  Synchronises the context. */
int flacgpu_handed_subframes(flacgpu_ctx *ctx, uint32_t *handed, uint32_t *subframes, int *enabled);

/* Device pointers of the last analysis (for callers that chain further device work):
 * which: 0 frame plans, 1 subframe plans, 2 residuals, 3 planar pcm, 4 packed frame bytes,
 * 5 frame byte offsets (uint64[n_frames + 1]), 6 the autocorrelation of every candidate (double[n_frames * candidates][36]:
 * lags 0 .. max_lpc_order of the windowed samples, encode.rs:3403-3413 -- bit for bit the reference's sums). */
void *flacgpu_device_buffer(flacgpu_ctx *ctx, int which);
/* The plans (0, 1) and the frame bytes / offsets (4, 5) an asynchronous call leaves in HBM are FINAL only after a resolving
 * entry point has run: flacgpu_fetch, flacgpu_fetch_frames, flacgpu_frames_ready, flacgpu_verify_device -- or this one, for
 * callers that consume the device buffers directly.  It waits for the context's stream, re-decides on the host the few
 * candidates whose LPC order estimate lies inside the libm-sensitive band (encode.rs:3656-3702) and re-runs the candidate
 * stage with the exact ResidualOverflow test (encode.rs:3189-3201) when the unchecked FIR could not rule an overflow out
 * (flacgpu_stats::fir_recheck != 0), then re-assembles the frames.  Nothing to do (the usual case): one stream sync and a
 * 16-byte read.  Synchronous. */
int flacgpu_resolve(flacgpu_ctx *ctx);

/* ---- device-side frame assembly ---------------------------------
 * Replaces, for the frames of the last flacgpu_analyze_device call, the host half of
 * encode_frame (encode.rs:2284-2294 header, :2332-2333 playback, :2408-2409 align + CRC-16)
 * and write_partitions / Partition::to_writer (encode.rs:3834-3907): frame header + CRC-8,
 * subframe headers, warm-up, LPC parameters, Rice/escaped residual codes (bit offsets by
 * prefix scan), byte alignment, CRC-16 -- producing the exact bytes `Encoder::encode` would
 * hand to its writer.  Frame f carries frame number first_frame_number + f.
 * Asynchronous; results stay in HBM until flacgpu_fetch_frames. */
int flacgpu_pack_device(flacgpu_ctx *ctx, uint64_t first_frame_number, uint32_t sample_rate,
                        void *stream);
/* flacgpu_analyze_device + flacgpu_pack_device of one batch in one asynchronous call: the device
 * half of `Encoder::encode` for every frame of the batch (encode.rs:2274-2440).
 * Overlap comes in two forms.  (1) Several contexts on several streams (double / triple buffering
 * of consecutive batches, what bench.py does): the HBM-, latency- and VALU-bound kernels of
 * different batches overlap; measured 0.86 -> 0.70 ms per 8192-frame batch with three contexts.
 * (2) FLACGPU_TUNE_TWO_RANGES: a single batch whose frames all take the 4096-sample wave kernels is
 * cut into two frame ranges whose kernel chains run on two HIP streams inside the context (worth
 * ~4 % for a lone batch, counter-productive together with (1), so off by default).
 * Either way the bytes, plans and counters are those of the two separate calls, and work submitted
 * to `stream` afterwards sees all of it. */
enum {
    FLACGPU_TUNE_TWO_RANGES = 1, /* 0 (default) / 1 */
    /* waves the autocorrelation lags of a 64-candidate group are split over: 4 (default; best for
     * a context that has the GPU to itself) or 2 (fewer instructions; best when several contexts
     * keep the SIMDs busy).  Results are identical. */
    FLACGPU_TUNE_LAG_SPLIT = 2,
    /* 1: flacgpu_frames_ready / flacgpu_fetch_frames_async / flacgpu_wait sleep between looks at their
     * event (50-200 us) instead of spinning in hipEventSynchronize.  For processes with more waiting threads
     * than CPUs (many concurrent writers): a spinning waiter burns the CPU time the others -- and the MD5
     * engines -- need.  0 (default) spins (lowest latency). */
    FLACGPU_TUNE_BLOCKING_WAIT = 3,
    /* 1: the context copies (splits) every batch of flacgpu_analyze_device / flacgpu_encode_device into its own
     * planar buffer at submission, as before r02: d_pcm may be refilled as soon as the submitted work has run,
     * whatever is fetched later (an LPC order tie re-decided at fetch time, flacgpu_verify_device, residual rows
     * all work from the copy).  0 (default): direct input, the lifetime rule at flacgpu_analyze_device applies.
     * For streaming callers that recycle one input buffer without waiting for their fetch. */
    FLACGPU_TUNE_COPY_INPUT = 4,
    /* flacgpu_encode_device runs a batch of in-place wave-kernel frames (interleaved whole 4096-sample blocks, LPC order
     * 1..16, exhaustive channel choice) range by range once it holds more than 1.5 x this many Mi samples: the whole kernel
     * chain for `value` Mi samples at a time.  Default: 64, stereo batches only (16384-frame batches: + 9 %; 8-channel batches
     * measured no gain and are cut only once this tuning has been set).  0: never cut.  The bytes, plans and counters are
     * those of the uncut batch. */
    FLACGPU_TUNE_CHUNK_MSAMPLES = 5
};
int flacgpu_set_tuning(flacgpu_ctx *ctx, int key, int value);
int flacgpu_encode_device(flacgpu_ctx *ctx, const int32_t *d_pcm, int layout, uint32_t n_frames,
                          uint32_t last_frame_len, uint64_t first_frame_number,
                          uint32_t sample_rate, void *stream);
/* Copies the packed frames of the last flacgpu_pack_device to the host.  offsets (may be NULL)
 * receives n_frames + 1 byte offsets into `out`; *total (may be NULL) the byte count.  When
 * `out` is NULL or cap is too small only offsets/total are filled and
 * FLACGPU_ERR_BUFFER_TOO_SMALL is returned. */
int flacgpu_fetch_frames(flacgpu_ctx *ctx, uint8_t *out, size_t cap, uint64_t *offsets,
                         uint64_t *total);
/* analyze + pack + fetch in one synchronous call on host PCM: the whole of the reference's
 * `Encoder::encode` loop for a batch of blocks, minus stream bookkeeping. */
int flacgpu_encode_frames(flacgpu_ctx *ctx, const int32_t *pcm, int layout, uint32_t n_frames,
                          uint32_t last_frame_len, uint64_t first_frame_number,
                          uint32_t sample_rate, uint8_t *out, size_t cap, uint64_t *offsets,
                          uint64_t *total);

/* Device-side frame assembly of CALLER-SUPPLIED decisions (the device counterpart of
 * flacenc_pack_frames, include/flacenc_stream.h): `pcm` (host, interleaved, as for flacgpu_analyze)
 * is split into rows, `plans` [n_frames] / `subframes` [n_frames * channels] replace what an analysis
 * would have decided, and the packers produce the frame bytes (flacgpu_fetch_frames).  The decisions
 * must be consistent with the PCM (residuals are recomputed from it). */
int flacgpu_pack_plans(flacgpu_ctx *ctx, const int32_t *pcm, uint32_t n_frames, uint32_t last_frame_len,
                       const flacgpu_frame_plan *plans, const flacgpu_subframe_plan *subframes,
                       uint64_t first_frame_number, uint32_t sample_rate);

/* ---- asynchronous host path: stream-width upload, sizes ahead of the bytes ---------------------
 * What a streaming front end (FlacSampleWriter / FlacByteWriter, encode.rs:359, 558) needs to keep
 * several batches in flight: PCM enters as the little-endian `bytes_per_sample = ceil(bps / 8)`-byte
 * samples that FlacByteWriter::write receives and update_md5 hashes (encode.rs:1292-1318,
 * byteorder.rs:60-72) -- 2 or 3 bytes per sample across PCIe instead of 4, widened by K0 on the
 * device --, ideally from pinned memory (flacgpu_host_alloc), so that the copy is asynchronous.
 *   flacgpu_encode_packed_async  H2D + analysis + frame assembly, all queued; returns at once
 *   flacgpu_frames_ready         waits for the frame SIZES only (they leave the device right after
 *                                k_layout, while the bytes are still being assembled); *offsets
 *                                (n_frames + 1 entries) stays valid until the context's next batch
 *   flacgpu_fetch_frames_async   queues the copy of exactly those bytes to `out`
 *   flacgpu_wait                 waits for that copy (or, without one, for the context's stream)
 * One batch per context at a time; several contexts overlap (H2D of one, kernels of another, D2H of
 * a third run concurrently on their own streams). */
void *flacgpu_host_alloc(size_t bytes); /* pinned host memory (hipHostMalloc), NULL on failure */
void flacgpu_host_free(void *p);
int flacgpu_current_device(void);       /* the caller's current HIP device, -1 without one */
/* Diagnostic: GB/s the host link of `device` (-1: current) carries, both directions summed, moving `bytes` each way with
 * up_mode / down_mode = 0 (idle), 1 (copy engine: hipMemcpyAsync on its own stream) or 2 (a kernel loading from /
 * storing to pinned host memory) at the same time.  The asynchronous host path uploads by copy engine and lets k_frame64
 * store the frames into pinned memory: (1, 2) is its ceiling. */
int flacgpu_link_probe(int device, size_t bytes, int up_mode, int down_mode, double *sum_gbs);
/* 1 when flacgpu_encode_packed_async takes this sample width for the context's stream shape (a block
 * must be a whole number of 16-byte groups); otherwise widen to int32 and pass bytes_per_sample 4 */
int flacgpu_packed_input_supported(const flacgpu_ctx *ctx, uint32_t bytes_per_sample);
/* the same question before a context exists (it depends on the block size, the channel count and the width only) */
int flacgpu_packed_input_shape_supported(uint32_t block_size, uint32_t channels, uint32_t bytes_per_sample);
int flacgpu_encode_packed_async(flacgpu_ctx *ctx, const uint8_t *pcm_le, uint32_t bytes_per_sample,
                                uint32_t n_frames, uint32_t last_frame_len, uint64_t first_frame_number,
                                uint32_t sample_rate);
int flacgpu_frames_ready(flacgpu_ctx *ctx, const uint64_t **offsets, uint64_t *total);
/* As flacgpu_encode_packed_async, with the frames assembled STRAIGHT INTO `out_host` (pinned memory from
 * flacgpu_host_alloc, at least flacgpu_packed_cap(ctx) bytes, 4-byte aligned): k_frame64 only stores -- dwords
 * inside a frame, bytes at its ends -- so it writes over PCIe while it works and no download follows.  Taken
 * when every frame of the batch is assembled by k_frame64 (whole blocks of a wave block length); otherwise
 * the frames stay in device memory as usual.  flacgpu_fetch_frames_async(out_host) then has nothing to copy;
 * flacgpu_wait returns when the frames are in `out_host`. */
int flacgpu_encode_packed_async_host(flacgpu_ctx *ctx, const uint8_t *pcm_le, uint32_t bytes_per_sample,
                                     uint32_t n_frames, uint32_t last_frame_len, uint64_t first_frame_number,
                                     uint32_t sample_rate, uint8_t *out_host, size_t out_cap);
size_t flacgpu_packed_cap(const flacgpu_ctx *ctx);   /* bytes a batch of max_frames frames can need */
int flacgpu_fetch_frames_async(flacgpu_ctx *ctx, uint8_t *out, size_t cap);
int flacgpu_wait(flacgpu_ctx *ctx);

/* ---- the pipelined batch loop: host PCM in, finished frames in host memory out, both link directions busy ----------
 * Replaces the per-block loop of `Encoder::encode` as FlacSampleWriter::write drives it (encode.rs:558-585) for callers
 * that bring whole batches of blocks and do their own stream bookkeeping: `depth` encoder contexts take consecutive
 * batches in rotation, each on its own HIP stream, so the upload of batch n, the kernels of batch n - 1 and the frames of
 * batch n - 2 (stored by k_frame64 straight into the slot's pinned buffer) are in flight together.
 *   submit   queues H2D + analysis + frame assembly of one batch and returns at once; FLACGPU_ERR_BUSY when all `depth`
 *            slots hold a batch.  `pcm`: interleaved samples in PINNED host memory (flacgpu_host_alloc), int32 when
 *            bytes_per_sample == 4, else the little-endian ceil(bps / 8)-byte samples of flacgpu_encode_packed_async;
 *            it must stay valid until that batch has been retired.
 *   retire   waits for the OLDEST batch in flight and hands out its frames: `*frames` (pinned memory owned by the
 *            pipeline), `*offsets` (n_frames + 1 byte offsets) and `*total` stay valid until the next submit, which
 *            reuses that slot.  Frames are byte-identical to flacgpu_encode_frames on the same batch.
 * Batches are independent (frame numbers come from the caller), so the output order is the submit order.
 * examples/c_abi_pipeline.c is the whole recipe; bench.py's end_to_end.pipelined_pcie measures it. */
typedef struct flacgpu_pipeline flacgpu_pipeline;
int flacgpu_pipeline_create(const flacgpu_options *opts, uint32_t bits_per_sample, uint32_t channels, int device,
                            uint32_t max_frames, uint32_t depth, flacgpu_pipeline **out);
void flacgpu_pipeline_destroy(flacgpu_pipeline *p);
int flacgpu_pipeline_submit(flacgpu_pipeline *p, const void *pcm, uint32_t bytes_per_sample, uint32_t n_frames,
                            uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate);
int flacgpu_pipeline_retire(flacgpu_pipeline *p, const uint8_t **frames, const uint64_t **offsets, uint32_t *n_frames,
                            uint64_t *total);
uint32_t flacgpu_pipeline_in_flight(const flacgpu_pipeline *p);
uint32_t flacgpu_pipeline_depth(const flacgpu_pipeline *p);

/* ---- a batch made of SEGMENTS: frames of SEVERAL streams in one analysis batch ------------------------------------------
 * The reference encodes every file with an `Encoder` of its own, block by block (encode.rs:487-627 once per file); a batch of
 * a few hundred frames leaves most of the GPU idle and still pays the kernels' launches and their serial walk over a block
 * (256 frames of 24-bit stereo run at 0.3 of the per-sample rate of 8192).  Here runs of WHOLE blocks of several streams of
 * the context's shape (bits per sample, channels, options) form one batch: segment s contributes n_frames frames numbered
 * first_frame_number, first_frame_number + 1, ...; the frames come out concatenated in segment order (offsets: one entry per
 * frame of the batch plus the end).  A stream's short last block is not a segment: it goes through flacgpu_encode_frames.
 *   flacgpu_encode_segments_device  PCM resident in device memory, asynchronous like flacgpu_encode_device (results through
 *                                   flacgpu_fetch_frames / flacgpu_device_buffer).  Every shape flacgpu_analyze_device reads in
 *                                   place -- interleaved stereo of 4096-, 2304-, 2048-, 1152- or 1024-sample blocks (<= 24
 *                                   bits), 2..8 interleaved independent channels of 4096-sample blocks (LPC order 1..16) --
 *                                   is read IN PLACE through a per-frame address table when every segment starts on a
 *                                   16-byte boundary (the buffers must stay valid until the results were fetched); other
 *                                   shapes (one channel, other block sizes, 32-bit samples) are gathered into the
 *                                   context's input buffer first.
 *   flacgpu_encode_segments         host PCM (int32): every segment is uploaded to its place in the context's input buffer --
 *                                   the gather costs nothing beyond the upload; synchronous, frames to `out`. */
typedef struct {
    const int32_t *pcm;            /* interleaved samples of n_frames whole blocks */
    uint32_t n_frames;
    uint32_t reserved;
    uint64_t first_frame_number;   /* of the segment's first frame (the frame number counts the stream's blocks) */
} flacgpu_segment;
int flacgpu_encode_segments_device(flacgpu_ctx *ctx, const flacgpu_segment *segments, uint32_t n_segments,
                                   uint32_t sample_rate, void *stream);
int flacgpu_encode_segments(flacgpu_ctx *ctx, const flacgpu_segment *segments, uint32_t n_segments, uint32_t sample_rate,
                            uint8_t *out, size_t cap, uint64_t *offsets, uint64_t *total);
/* The asynchronous, stream-width form (r06): what a front end that coalesces many streams needs to keep the link busy.  The
 * segments' whole blocks lie BACK TO BACK in `pcm_le` (pinned memory; int32 when bytes_per_sample == 4, else the little-endian
 * ceil(bps / 8)-byte samples of flacgpu_encode_packed_async -- the byte string update_md5 hashes, encode.rs:1292-1318) in
 * segment order; flacgpu_segment::pcm is ignored.  Upload, analysis and frame assembly are queued and the call returns;
 * k_frame64 stores the frames straight into `out_host` (pinned, >= flacgpu_packed_cap(ctx) bytes) when every frame takes it.
 * Results as for flacgpu_encode_packed_async_host: flacgpu_frames_ready (one offset per frame of the batch plus the end),
 * flacgpu_fetch_frames_async, flacgpu_wait.  `pcm_le` must stay valid until the batch has been waited for. */
int flacgpu_encode_segments_packed_async_host(flacgpu_ctx *ctx, const uint8_t *pcm_le, uint32_t bytes_per_sample,
                                              const flacgpu_segment *segments, uint32_t n_segments, uint32_t sample_rate,
                                              uint8_t *out_host, size_t out_cap);

/* ---- several GPUs ------------------------------------------------------------------------------------
 * A FLAC frame depends on its own samples, the options and its frame number only (encode.rs:2284-2294), so a stream shards
 * by CONTIGUOUS FRAME RANGES with no data-path exchange.  What crosses shards is bookkeeping -- the reference's single
 * process keeps it in `Encoder` (encode.rs:1997-2022): the seek points' byte offsets, a prefix sum of frame sizes
 * (:1999-2003), STREAMINFO's min / max frame size (:2414-2436) and the totals.  One record of four integers per shard
 * carries all of it. */
typedef struct {
    uint64_t frames;     /* frames of the shard */
    uint64_t bytes;      /* their bytes */
    uint64_t min_frame;  /* smallest / largest frame of the shard; 0 for a shard without frames, which takes no part in */
    uint64_t max_frame;  /* the merged min / max */
} flacgpu_shard_counters;
/* Stream-level bookkeeping from the per-shard records, in shard order: totals, min / max frame size and (optional, n
 * entries) the byte offset of every shard's first frame behind the stream's first frame (exclusive prefix sum). */
int flacgpu_merge_counters(const flacgpu_shard_counters *shards, uint32_t n, flacgpu_shard_counters *merged,
                           uint64_t *shard_byte_offsets);
/* The contiguous frame range [lo, hi) of shard `shard` of `shards`: [k F / G, (k + 1) F / G). */
void flacgpu_shard_range(uint64_t total_frames, uint32_t shards, uint32_t shard, uint64_t *lo, uint64_t *hi);
int flacgpu_device_count(void);   /* visible HIP devices */
/* The host side of a device: the NUMA node its PCI function hangs off (-1: unknown) and the CPUs local to it as sysfs spells
 * them ("0-63,128-191"; empty: unknown) -- where the host threads that feed the device belong. */
int flacgpu_device_numa_info(int device, int *numa_node, char *cpulist, size_t cap);

/* ONE PROCESS, SEVERAL DEVICES.  `devices` lists HIP ordinals, one shard each (NULL / 0: every visible device; an ordinal
 * may be listed more than once -- each listing is a shard with contexts of its own).  Every shard owns up to `depth`
 * encoder contexts sized for `max_frames` frames per batch.
 *   flacgpu_multi_encode         a run of blocks of ONE stream in host memory (int32, or the little-endian ceil(bps/8)-byte
 *                                samples of flacgpu_encode_packed_async; pinned memory keeps the uploads asynchronous): cut into
 *                                batches of <= max_frames frames that are DEALT to the shards in turn (batch j to shard j mod
 *                                shards), each shard keeping `depth` of its batches in flight on a parked host thread of its own
 *                                that is bound to the CPUs of its GPU's NUMA node; every frame carries the number the stream
 *                                gives it.  The frames come back concatenated in stream order in `out` -- each retired batch is
 *                                copied ONCE, from its pinned slot to its final place (a batch's place is the sum of the sizes of
 *                                the batches before it, published as they retire) --, `offsets` (n_frames + 1 entries),
 *                                `per_shard` (flacgpu_multi_shards entries: what each shard encoded) and `merged` as
 *                                flacgpu_merge_counters leaves them.  Any out pointer may be NULL (FLACGPU_ERR_BUFFER_TOO_SMALL
 *                                with *total set when `out` is too small).  Byte-identical to one context's
 *                                flacgpu_encode_frames over the same blocks.  Synchronous.
 *   flacgpu_multi_encode_device  PCM resident in shard `shard`'s device memory: flacgpu_encode_device on the shard's next
 *                                context in rotation, asynchronous; flacgpu_multi_wait drains every context of every shard;
 *                                flacgpu_multi_counters reads the records of every shard's LAST batch (and merges them);
 *                                flacgpu_multi_last_context hands that batch's context out (fetch / verify). */
/* A flacgpu_multi belongs to one calling thread at a time (like a flacgpu_ctx); flacgpu_multi_encode runs its shards on
 * threads of its own for the duration of the call. */
typedef struct flacgpu_multi flacgpu_multi;
int flacgpu_multi_create(const flacgpu_options *opts, uint32_t bits_per_sample, uint32_t channels, const int *devices,
                         uint32_t n_devices, uint32_t max_frames, uint32_t depth, flacgpu_multi **out);
void flacgpu_multi_destroy(flacgpu_multi *m);
uint32_t flacgpu_multi_shards(const flacgpu_multi *m);
/* Cumulative over the object's flacgpu_multi_encode calls: bytes handed out in `out`, bytes the host copied to put them there
 * (the same number: one copy per output byte), and how many of the shards' threads were bound to their GPU's local CPUs. */
int flacgpu_multi_host_copy_stats(const flacgpu_multi *m, uint64_t *bytes_out, uint64_t *bytes_copied, uint32_t *threads_near_gpu);
int flacgpu_multi_device_of(const flacgpu_multi *m, uint32_t shard);
int flacgpu_multi_encode(flacgpu_multi *m, const void *pcm, uint32_t bytes_per_sample, uint64_t n_frames,
                         uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate, uint8_t *out,
                         size_t cap, uint64_t *offsets, uint64_t *total, flacgpu_shard_counters *per_shard,
                         flacgpu_shard_counters *merged);
int flacgpu_multi_encode_device(flacgpu_multi *m, uint32_t shard, const int32_t *d_pcm, int layout, uint32_t n_frames,
                                uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate);
int flacgpu_multi_wait(flacgpu_multi *m);
int flacgpu_multi_counters(flacgpu_multi *m, flacgpu_shard_counters *per_shard, flacgpu_shard_counters *merged);
flacgpu_ctx *flacgpu_multi_last_context(flacgpu_multi *m, uint32_t shard);

/* ONE PROCESS PER GPU (torchrun / MPI): the same record all-gathered over the caller's RCCL communicator (an
 * `ncclComm_t`; `stream` a hipStream_t or NULL) -- the path's only collective, 32 bytes per rank.  `all` receives
 * *n_ranks records in rank order (cap_ranks entries provided).  librccl is loaded on first use; when it cannot be,
 * the call fails with FLACGPU_ERR_NO_RCCL -- there is no fallback.  Synchronous. */
int flacgpu_rccl_available(void);
int flacgpu_rccl_allgather_counters(void *nccl_comm, void *stream, const flacgpu_shard_counters *mine,
                                    flacgpu_shard_counters *all, uint32_t cap_ranks, uint32_t *n_ranks, uint32_t *my_rank);

/* ---- device-side decode + verify of the frames packed last ----------
 * The reference's frame decoder (decode.rs:1388-1436 read_frame, 1494-1633 read_subframes,
 * 1635-1752 read_subframe / predict, 1800-1856 read_residuals) run on the GPU, one lane per
 * subframe, over the bytes produced by flacgpu_pack_device, plus a CRC-16 check of every frame.
 * A lane starts where the encoder's plan says its subframe starts; the plan is proven, not
 * trusted: subframe 0 must start where the frame header ends, every subframe must end exactly
 * where the next one starts and the last one at the padding before the CRC-16 (otherwise the
 * frame counts as bad_structure) -- i.e. the frame parses front to back as a sequential decoder
 * would parse it.
 * When the analysed PCM is still in the context (host-input or interleaved device input), the
 * decoded samples are compared with it -- the round trip every encoder test of the reference
 * performs (tests/format.rs), without leaving HBM. */
typedef struct {
    uint32_t frames;
    uint32_t bad_structure;      /* header / CRC-8 / subframe structure errors */
    uint32_t bad_crc16;
    uint32_t frames_pcm_differs; /* only meaningful when compared_pcm != 0 */
    uint32_t samples_differ;
    uint32_t compared_pcm;
} flacgpu_verify_result;
int flacgpu_verify_device(flacgpu_ctx *ctx, uint32_t sample_rate, uint64_t first_frame_number,
                          flacgpu_verify_result *result, float *kernel_ms);
/* decoded PCM of the last flacgpu_verify_device, interleaved, to host memory
 * ((n_frames-1)*block_size + last_frame_len) * channels samples */
int flacgpu_fetch_decoded(flacgpu_ctx *ctx, int32_t *interleaved);

/* ---- stand-alone decoder: any FLAC stream, not only this encoder's output ---------------------
 * The reference's `verify` / frame reader (decode.rs:1282, 1388-1436 read_frame, 1494-1856) for a
 * whole stream held in host memory: the host parses the metadata and finds the frame boundaries
 * (next valid header preceded by the CRC-16 of everything since the frame's start), the GPU decodes
 * the frames in parallel (one lane per frame walks its subframes front to back), re-checks every
 * CRC-16 and undoes the stereo decorrelation; the MD5 of the decoded PCM is compared with
 * STREAMINFO's.  `out` (may be NULL: verify only) receives the interleaved samples. */
typedef struct {
    uint32_t sample_rate, channels, bits_per_sample, min_block, max_block;
    uint32_t frames;            /* frames found by the scan */
    uint32_t bad_frames;        /* frames that did not parse (+1 when the scan lost synchronisation) */
    uint32_t bad_crc16;
    uint64_t total_samples;     /* STREAMINFO: samples per channel, 0 = unknown */
    uint64_t decoded_samples;   /* samples per channel in the frames found */
    uint8_t md5[16];            /* STREAMINFO */
    uint8_t decoded_md5[16];    /* of the decoded PCM */
    uint32_t md5_status;        /* 1: equal, 0: different, 2: STREAMINFO holds no MD5 (all zero) */
    uint32_t reserved;
} flacgpu_stream_info;
int flacgpu_decode_stream(const uint8_t *data, size_t len, int device, int32_t *out, size_t out_cap_samples,
                          flacgpu_stream_info *info);

/* EXPERIMENT, not on the product path: recomputes the autocorrelation of the last analysed
 * batch on the f64 matrix cores (v_mfma_f64_16x16x4_f64, block-Gram form), times that kernel,
 * reruns Levinson/quantisation on it and reports how many candidates' quantised LPC parameters
 * (order, shift, coefficients) differ from the exact-summation-order path, and the largest
 * relative error of an autocorrelation lag.  The context's exact results are restored.
 * Requires full blocks and 1 <= max_lpc_order <= 16. */
int flacgpu_experiment_mfma_autocorr(flacgpu_ctx *ctx, float *kernel_ms, uint32_t *compared,
                                     uint32_t *params_differ, double *max_rel_err);

/* Duration in milliseconds of each kernel of the last analyze call, measured with HIP events
 * on the launch stream (names via flacgpu_kernel_name).  Requires flacgpu_set_timing(ctx, 1). */
#define FLACGPU_N_KERNELS 12
int flacgpu_set_timing(flacgpu_ctx *ctx, int enable);
int flacgpu_get_kernel_ms(flacgpu_ctx *ctx, float ms[FLACGPU_N_KERNELS]);
const char *flacgpu_kernel_name(int index);

/* Identity of this build: the first 16 hex digits of the SHA-256 over the library's sources (every file under
 * csrc/ but build/, in sorted path order: csrc/Makefile `BUILD_ID`).  Counter collections under profiles/ carry the
 * id of the library they were taken with; bench.py prints `counters_stale` when it differs from the running one. */
const char *flacgpu_build_id(void);

#ifdef __cplusplus
}
#endif
#endif
