/*
 * flacenc_stream.h -- C ABI of the host-side stream writers.
 *
 * Mirrors the public encode surface of the reference (tuffy/flac-codec 1.3.2):
 *   encode::Options                    /root/reference/src/encode.rs:1361-1672
 *   FlacSampleWriter::{new,write,finalize}           encode.rs:487, 558, 624
 *   FlacByteWriter::{new,write(io::Write),finalize}  encode.rs:143, 359, 258
 *   FlacChannelWriter::{new,write,finalize}          encode.rs:769, 832, 939
 *   FlacStreamWriter::{new,write}                    encode.rs:1094, 1142
 * Same argument meaning and the same error conditions (the FLACENC_ERR_* values
 * name the reference's `Error` / `OptionsError` variants, src/lib.rs:59-193,
 * encode.rs:1676-1698).  The per-frame analysis is done by the gfx950 kernels
 * behind include/flacenc_gpu.h; this layer does what stays on the host in the
 * reference's design: block buffering, MD5, frame headers, Rice bit-packing,
 * CRC-8/16, seek points and the metadata rewrite at finalize
 * (Encoder::new / encode / finalize_inner, encode.rs:1882-2110).
 *
 * Frames are analysed in batches (`batch_frames`), so a `write` may emit its
 * frames later than the reference would (at the latest at finalize); the bytes
 * are identical.
 */
#ifndef FLACENC_STREAM_H
#define FLACENC_STREAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FLACENC_OK = 0,
    /* OptionsError, encode.rs:1676-1698 */
    FLACENC_ERR_INVALID_BLOCK_SIZE = -101,
    FLACENC_ERR_INVALID_LPC_ORDER = -102,
    FLACENC_ERR_INVALID_MAX_PARTITIONS = -103,
    FLACENC_ERR_EXCESSIVE_PADDING = -104,
    /* Error, src/lib.rs:59-193 */
    FLACENC_ERR_INVALID_BITS_PER_SAMPLE = -110,        /* encode.rs:495 */
    FLACENC_ERR_INVALID_SAMPLE_RATE = -111,            /* encode.rs:1902 */
    FLACENC_ERR_EXCESSIVE_CHANNELS = -112,             /* encode.rs:1908 */
    FLACENC_ERR_SAMPLES_NOT_DIVISIBLE_BY_CHANNELS = -113, /* encode.rs:517 */
    FLACENC_ERR_INVALID_TOTAL_SAMPLES = -114,          /* encode.rs:518 */
    FLACENC_ERR_INVALID_TOTAL_BYTES = -115,            /* encode.rs:177 */
    FLACENC_ERR_EXCESSIVE_TOTAL_SAMPLES = -116,        /* encode.rs:1913, 2010, 2094 */
    FLACENC_ERR_SAMPLE_COUNT_MISMATCH = -117,          /* encode.rs:2083 */
    FLACENC_ERR_NO_SAMPLES = -118,                     /* encode.rs:2090 */
    FLACENC_ERR_EXCESSIVE_FRAME_NUMBER = -119,         /* stream.rs:1235 */
    FLACENC_ERR_CHANNEL_COUNT_MISMATCH = -120,         /* encode.rs:858 */
    FLACENC_ERR_CHANNEL_LENGTH_MISMATCH = -121,        /* encode.rs:854 */
    FLACENC_ERR_NON_SUBSET_SAMPLE_RATE = -122,         /* encode.rs:1181 */
    FLACENC_ERR_NON_SUBSET_BITS_PER_SAMPLE = -123,     /* encode.rs:1151, 1187 */
    FLACENC_ERR_IO = -130,
    /* not reference errors */
    FLACENC_ERR_INVALID_ARG = -140,
    FLACENC_ERR_FINALIZED = -141,
    FLACENC_ERR_GPU = -150,          /* flacgpu_* failed; flacenc_last_error() has the text */
    FLACENC_ERR_UNSUPPORTED = -151   /* legal upstream, not in this build (see flacenc_gpu.h) */
};

enum { FLACENC_SEEKTABLE_NONE = 0, FLACENC_SEEKTABLE_SECONDS = 1, FLACENC_SEEKTABLE_FRAMES = 2 };

/* encode::Options (encode.rs:1363-1374).  Metadata beyond STREAMINFO / SEEKTABLE /
 * PADDING / VORBIS_COMMENT is out of scope. */
typedef struct {
    uint32_t block_size;          /* Options::block_size, >= 16 (encode.rs:1418) */
    uint32_t max_partition_order; /* 0..=15 (encode.rs:1447) */
    uint32_t max_lpc_order;       /* 0 = None, 1..=32 (encode.rs:1430) */
    uint8_t mid_side;             /* encode.rs:1460 */
    uint8_t exhaustive_channel_correlation; /* !fast_channel_correlation (encode.rs:1472) */
    uint8_t window_kind;          /* 0 Rectangle, 1 Hann, 2 Tukey (encode.rs:1713) */
    uint8_t reserved0;
    float window_param;
    int64_t padding;              /* PADDING bytes; 0 removes the block (encode.rs:1486-1508) */
    int32_t seektable_mode;       /* FLACENC_SEEKTABLE_* (encode.rs:1568-1590) */
    uint32_t seektable_value;     /* seconds (u8) or frames */
    /* execution knobs (no reference equivalent) */
    uint32_t batch_frames;        /* FLAC frames analysed per GPU call; 0 = default (256: small enough that a stream's first batch reaches the MD5 thread and the GPU within ~2 ms) */
    int32_t device;               /* HIP device ordinal, -1 = current */
    uint32_t pack_threads;        /* host bit-pack threads; 0 = default */
    uint32_t host_pack;           /* 0: frames are assembled on the GPU (k_layout/k_pack/k_crc);
                                     1: Rice bit-packing + CRC stay on the host, as in the
                                     north-star split (same bytes either way) */
    /* VORBIS_COMMENT (Options::tag / Options::comment, encode.rs:1513-1528).  n_comment_fields
     * == 0 and vendor_string == NULL: no block.  Fields are "NAME=value" strings; a NULL
     * vendor_string with fields present means VorbisComment::default()'s "flac-codec 1.3.2"
     * (metadata/mod.rs:2225-2232).  The pointers are only read during *_writer_new. */
    const char *vendor_string;
    const char *const *comment_fields;
    uint32_t n_comment_fields;
    uint32_t pipeline_depth;      /* batches in flight per writer (device-side frame assembly): each on
                                     a context, HIP stream and pinned staging buffers of its own, so
                                     that upload, kernels, download and the MD5 of neighbouring
                                     batches overlap; 0 = default (2), at most 4 */
    uint32_t shared_md5;          /* 1: the stream's MD5 chain runs on the shared multi-stream engines
                                     (host/md5_mb.h: up to 16 streams per engine thread, one per AVX-512
                                     lane) instead of on a worker thread of its own -- for callers that
                                     encode many streams side by side; flacenc_encode_many sets it */
} flacenc_options;

void flacenc_options_default(flacenc_options *o); /* Options::default(), encode.rs:1376-1408 */
void flacenc_options_fast(flacenc_options *o);    /* Options::fast(),    encode.rs:1635-1644 */
void flacenc_options_best(flacenc_options *o);    /* Options::best(),    encode.rs:1649-1657 */
/* validation of the setters (returns the OptionsError the reference's builder would) */
int flacenc_options_validate(const flacenc_options *o);

/* Output sink = the reference's `W: Write + Seek`.  `seek` is only used by finalize
 * (rewind to the stream start to rewrite the metadata, encode.rs:2104-2106).
 * Return 0 on success. */
typedef int (*flacenc_write_fn)(void *user, const uint8_t *data, size_t len);
typedef int (*flacenc_seek_fn)(void *user, uint64_t absolute_offset);
typedef struct {
    flacenc_write_fn write;
    flacenc_seek_fn seek;
    void *user;
    uint64_t start; /* writer.stream_position() at creation (encode.rs:1941) */
} flacenc_sink;

typedef struct flacenc_writer flacenc_writer;

/* FlacSampleWriter::new (encode.rs:487).  has_total = 0 is `None`; total_samples counts
 * ALL channels like the reference (must divide by channels).  sink == NULL collects the
 * stream in memory (see flacenc_writer_data). */
int flacenc_sample_writer_new(const flacenc_options *opts, uint32_t sample_rate,
                              uint32_t bits_per_sample, uint32_t channels, int has_total,
                              uint64_t total_samples, const flacenc_sink *sink,
                              flacenc_writer **out);
/* FlacByteWriter::new (encode.rs:143); total_bytes like the reference. */
int flacenc_byte_writer_new(const flacenc_options *opts, uint32_t sample_rate,
                            uint32_t bits_per_sample, uint32_t channels, int has_total,
                            uint64_t total_bytes, int big_endian, const flacenc_sink *sink,
                            flacenc_writer **out);
/* FlacChannelWriter::new (encode.rs:769); total_samples is per channel. */
int flacenc_channel_writer_new(const flacenc_options *opts, uint32_t sample_rate,
                               uint32_t bits_per_sample, uint32_t channels, int has_total,
                               uint64_t total_samples, const flacenc_sink *sink,
                               flacenc_writer **out);

/* FlacSampleWriter::write (encode.rs:558): interleaved samples, any count. */
int flacenc_write_samples(flacenc_writer *w, const int32_t *samples, size_t count);
/* FlacByteWriter's io::Write::write (encode.rs:359): PCM bytes, any count. */
int flacenc_write_bytes(flacenc_writer *w, const uint8_t *bytes, size_t count);
/* FlacChannelWriter::write (encode.rs:832): n_channels arrays of `len` samples. */
int flacenc_write_channels(flacenc_writer *w, const int32_t *const *channels, uint32_t n_channels,
                           size_t len);
/* finalize (encode.rs:624 / 258 / 939): flush the last partial block, update SEEKTABLE,
 * STREAMINFO (sizes, total samples, MD5) and rewrite the metadata.  Idempotent. */
int flacenc_finalize(flacenc_writer *w);
/* Drop (encode.rs:399, 2113): finalizes, swallowing errors, then frees. */
void flacenc_writer_free(flacenc_writer *w);

/* memory sink access (sink == NULL at creation) */
const uint8_t *flacenc_writer_data(flacenc_writer *w, size_t *len);

/* Batch front end: many independent streams encoded concurrently by `threads` host workers (0 =
 * default), each one FlacSampleWriter::new(total known) / write / finalize (encode.rs:487, 558, 624)
 * with the .flac bytes written to the job's own buffer (FLACENC_ERR_IO in `status` when it is too
 * small; flacenc_worst_case_bytes() is always enough: every frame VERBATIM at bps + 1 bits with its headers, one seek
 * point per frame, the metadata blocks).  The workers share the
 * GPU through the pooled analysis lanes; every stream's MD5 chain runs on a thread of its own.
 * Returns 0 or the first job's error. */
/* Upper bound of the .flac size of a stream of `pcm_frames` samples per channel under `opts`. */
size_t flacenc_worst_case_bytes(const flacenc_options *opts, uint32_t bits_per_sample, uint32_t channels,
                                uint64_t pcm_frames);
typedef struct {
    const int32_t *samples;   /* interleaved, the whole stream */
    size_t count;             /* samples over all channels */
    uint32_t sample_rate, bits_per_sample, channels;
    uint32_t reserved;
    uint8_t *out;             /* caller-owned output buffer */
    size_t out_cap;
    size_t out_len;           /* out: bytes of the finished stream */
    int32_t status;           /* out: FLACENC_OK or the error of this stream */
    int32_t reserved1;
    /* out, diagnostics: wall time of this stream from its first phase to its last, and inside it the staging
     * (sample packing), the GPU calls incl. waiting for results, and the MD5 thread's busy time */
    double elapsed_ms, pack_ms, gpu_ms, md5_ms, start_ms;
} flacenc_job;
/* `threads` workers (0: half the hardware threads) share the streams' three phases -- first blocks, remaining samples,
 * finish --, at most 64 streams open at a time: the thread count need not match the stream count (a few threads per
 * usable CPU are enough), every stream's MD5 chain starts at once.  Returns the first failing stream's status. */
int flacenc_encode_many(const flacenc_options *opts, flacenc_job *jobs, size_t n_jobs, uint32_t threads);
/* The same over several GPUs of one process: stream i is encoded whole on devices[i mod n_devices] (streams are independent:
 * one `Encoder` per file in the reference, encode.rs:1882-1980 -- the natural shard; nothing crosses devices).  devices ==
 * NULL with n_devices == FLACENC_ALL_DEVICES: every visible device; NULL / 0: opts->device as flacenc_encode_many.  An
 * ordinal may be listed more than once.  Output is byte-identical to flacenc_encode_many's. */
/* The same with the streams of one shape (sample rate, bits per sample, channels) SHARING analysis batches: runs of whole
 * blocks of several streams form one batch (every frame with its own stream's frame number), and the batches travel through
 * a ring of pinned staging buffers -- samples packed to the stream's width on the way in (2 or 3 bytes across PCIe; the packed
 * bytes are what the MD5 lanes hash), upload of batch i + 1, kernels of batch i and the frames of batch i - 1 in flight
 * together, frames copied once from the pinned ring to `out`.  A library of short files runs at the per-sample rate of large
 * batches (a 256-frame batch of 24-bit stereo on its own: 0.3 of it), long streams at the link's.  A stream's short last
 * block is encoded by a one-frame call, its metadata is rebuilt from the frame sizes (flacenc_stream_header).  Output
 * byte-identical to flacenc_encode_many's, stream by stream (csrc/host/coalesce.cpp; measured: DESIGN.md section 6). */
int flacenc_encode_many_coalesced(const flacenc_options *opts, flacenc_job *jobs, size_t n_jobs, uint32_t threads);
/* Frees what the front ends keep between calls: idle analysis contexts and their pinned staging buffers (the writers' lane
 * pool, the coalescing ring).  Contexts in use are not touched. */
void flacenc_release_pools(void);
#define FLACENC_ALL_DEVICES 0xFFFFFFFFu
int flacenc_encode_many_devices(const flacenc_options *opts, flacenc_job *jobs, size_t n_jobs, uint32_t threads,
                                const int *devices, uint32_t n_devices);

/* FlacStreamWriter (encode.rs:1050-1290): header-less subset frames, parameters per call. */
typedef struct flacenc_stream_writer flacenc_stream_writer;
int flacenc_stream_writer_new(const flacenc_options *opts, const flacenc_sink *sink,
                              flacenc_stream_writer **out);
int flacenc_stream_writer_write(flacenc_stream_writer *w, uint32_t sample_rate, uint32_t channels,
                                uint32_t bits_per_sample, const int32_t *samples, size_t count);
const uint8_t *flacenc_stream_writer_data(flacenc_stream_writer *w, size_t *len);
void flacenc_stream_writer_free(flacenc_stream_writer *w);

/* statistics of a writer (after finalize) */
typedef struct {
    uint64_t frames, samples_per_channel, bytes_written;
    uint32_t min_frame_size, max_frame_size;
    uint8_t md5[16];
    double gpu_ms, pack_ms, md5_ms; /* accumulated host-side wall time per stage */
} flacenc_stats;
int flacenc_writer_stats(flacenc_writer *w, flacenc_stats *out);

/* Host bit-packing of one analysed batch (what Encoder::encode does after the analysis:
 * frame header + CRC-8, subframes, Rice residuals, CRC-16; encode.rs:2284-2409, 3834-3863).
 * Inputs are the outputs of flacgpu_analyze / flacgpu_fetch.  Frame f gets frame number
 * first_frame_number + f and lands at out[offsets[f] .. offsets[f+1]); offsets has
 * n_frames + 1 entries.  Returns 0, or FLACENC_ERR_INVALID_ARG when `cap` is too small
 * (offsets[n_frames] then holds the required size). */
int flacenc_pack_frames(uint32_t sample_rate, uint32_t bits_per_sample, uint32_t channels,
                        uint64_t first_frame_number, uint32_t n_frames, uint32_t row_stride,
                        const void *frame_plans, const void *subframe_plans,
                        const int32_t *residual_rows, uint32_t threads, uint8_t *out, size_t cap,
                        uint64_t *offsets);

/* Everything in front of the first frame (fLaC + STREAMINFO + SEEKTABLE + VORBIS_COMMENT + PADDING)
 * exactly as a writer that had emitted frames of these sizes leaves it at finalize: for the owner of a
 * stream whose contiguous frame ranges were encoded on several GPUs (encode.rs:1999-2003
 * seek points, 2414-2436 min/max frame size, 2024-2110 finalize).  `md5`: of the whole PCM, computed by
 * the owner.  *len receives the size; FLACENC_ERR_INVALID_ARG when `cap` is too small. */
int flacenc_stream_header(const flacenc_options *opts, uint32_t sample_rate, uint32_t bits_per_sample,
                          uint32_t channels, uint64_t total_pcm_frames, const uint8_t md5[16], uint64_t n_frames,
                          const uint32_t *frame_sizes, uint32_t last_frame_len, uint8_t *out, size_t cap,
                          size_t *len);

/* The 34-byte STREAMINFO body exactly as the writers serialise it (metadata/mod.rs:1599-1630). */
int flacenc_streaminfo_bytes(uint32_t min_block, uint32_t max_block, uint32_t min_frame, uint32_t max_frame,
                             uint32_t sample_rate, uint32_t channels, uint32_t bits_per_sample,
                             uint64_t total_samples, const uint8_t md5[16], uint8_t out[34]);

/* Test hooks of the multi-stream MD5 engine (host/md5_mb.cpp): digests of `streams` chains fed through the
 * pool in `runs` runs of assorted lengths against the scalar implementation (returns the mismatches), and
 * whether this host runs the 16-lane AVX-512 step. */
int flacenc_md5_selftest(uint32_t streams, uint32_t runs, uint32_t seed);
int flacenc_md5_simd_available(void);
/* Diagnostic: GB/s of `lanes` (1..48) independent MD5 chains advanced in lockstep by one thread over kib_per_lane KiB each;
 * divided by the lane count it is the speed of ONE chain -- the per-stream bound of every front end (a stream's MD5 is one
 * serial chain, encode.rs:571, 1292-1318). */
double flacenc_md5_probe(uint32_t lanes, uint32_t kib_per_lane);
/* Test hook of flacenc_encode_many_coalesced's batch planning (host/coalesce.cpp, tests/test_coalesce_plan.py): streams of
 * whole[k] whole blocks of one shape (samples_per_block = block size x channels), batch_frames as in flacenc_options (0: the
 * default) -> segment i is blocks [seg_first[i], seg_first[i] + seg_n[i]) of stream seg_stream[i] in batch seg_batch[i].
 * Returns the number of segments (the arrays hold up to `cap`); *batch_cap: the frames a batch may hold. */
size_t flacenc_coalesce_plan(const uint64_t *whole, size_t n_streams, uint32_t batch_frames, uint32_t samples_per_block, uint32_t *seg_stream,
                             uint64_t *seg_first, uint32_t *seg_n, uint32_t *seg_batch, size_t cap, uint32_t *batch_cap);

const char *flacenc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
