/*
 * flac_oracle.h -- CPU ORACLE for the FLAC encode hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it, and there only
 * as the checker / the timed CPU baseline.  The product path
 * (flac-codec_amd/) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference's encoder algorithm
 * (tuffy/flac-codec 1.3.2, /root/reference/src/encode.rs and friends); every
 * function cites the reference file:line it follows.  The reference is pure
 * Rust and there is no Rust toolchain in this image, so the reference itself
 * cannot be compiled here (no oracle/_ref).  Pinning:
 *   - per-stage results are pinned by the reference's own known-answer unit
 *     tests (encode.rs:3216-3745, decode.rs:1754, stream.rs doc-test bytes),
 *     reproduced in tests/test_oracle_kat.py;
 *   - the bitstream is additionally pinned by decoding the oracle's output
 *     with an independent decoder restatement (decode.rs:1388-1856) and by
 *     decoding the reference's own .flac fixtures (the .flac files under the reference tests/data);
 *   - whole-bitstream ENCODER output is NOT pinned by any reference test
 *     (no golden .flac from this encoder exists upstream): "bitstream parity
 *     unpinned" beyond the stage KATs -- see DESIGN.md.
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_CHANNELS 8
#define ORC_MAX_LPC 32
#define ORC_MAX_PARTITIONS 64

/* encode.rs:1713-1720 */
enum { ORC_WINDOW_RECTANGLE = 0, ORC_WINDOW_HANN = 1, ORC_WINDOW_TUKEY = 2 };

/* Options (encode.rs:1363-1374) restricted to what reaches the hot path
 * (EncoderOptions, encode.rs:1701-1709) plus the container knobs. */
typedef struct {
    uint32_t block_size;          /* encode.rs:1366 (u16, >= 16) */
    uint32_t max_partition_order; /* 0..=15 */
    int32_t mid_side;
    int32_t max_lpc_order;        /* 0 = None, else 1..=32 */
    int32_t window_kind;
    float window_param;           /* Tukey p */
    int32_t exhaustive;           /* exhaustive_channel_correlation */
    /* container */
    int32_t padding;              /* bytes of PADDING block, <0 = none */
    int32_t seektable_mode;       /* 0 none, 1 seconds, 2 frames */
    uint32_t seektable_value;
} orc_options;

enum {
    ORC_SUB_CONSTANT = 0,
    ORC_SUB_VERBATIM = 1,
    ORC_SUB_FIXED = 2,
    ORC_SUB_LPC = 3
};

/* channel assignment codes as written in the frame header (stream.rs:938) */
enum {
    ORC_ASSIGN_INDEPENDENT = 0,
    ORC_ASSIGN_LEFT_SIDE = 8,
    ORC_ASSIGN_SIDE_RIGHT = 9,
    ORC_ASSIGN_MID_SIDE = 10
};

/* decision record of one encoded subframe (what the reference holds in a
 * BitRecorder, encode.rs:1832-1836, expressed as data) */
typedef struct {
    uint8_t type;            /* ORC_SUB_* */
    uint8_t wasted;          /* wasted bits per sample */
    uint8_t bps;             /* effective bps after wasted-bit removal */
    uint8_t order;           /* predictor order */
    uint8_t precision;       /* LPC coefficient precision (bits) */
    uint8_t shift;           /* LPC shift */
    uint8_t coding_method;   /* 0 = RICE, 1 = RICE2 */
    uint8_t partition_order; /* as written: ilog2(#partitions) */
    uint32_t n_partitions;   /* number of partitions actually emitted */
    uint32_t bits;           /* BitRecorder::written() */
    int32_t coeffs[ORC_MAX_LPC];
    uint8_t rice[ORC_MAX_PARTITIONS];        /* 0xFF = escaped/constant */
    uint8_t escape_bits[ORC_MAX_PARTITIONS]; /* escape size (0 = constant) */
    uint16_t part_len[ORC_MAX_PARTITIONS];   /* residuals per partition */
} orc_subframe_plan;

typedef struct {
    uint8_t assignment;      /* ORC_ASSIGN_* (independent: 0) */
    uint8_t channels;
    uint16_t block_size;
    uint32_t frame_bytes;
    /* per emitted subframe, the candidate it came from:
     * 0..7 = input channel, 8 = mid, 9 = side */
    uint8_t source[ORC_MAX_CHANNELS];
    orc_subframe_plan sub[ORC_MAX_CHANNELS];
} orc_frame_plan;

/* ---- stage functions (pinned by the reference's KATs) ---- */

/* encode.rs:1725-1783 Window::generate */
void orc_window_generate(int kind, float p, uint32_t n, double *window);
/* encode.rs:3478-3501 autocorrelate; returns number of lags produced */
int orc_autocorrelate(const double *windowed, uint32_t n, uint32_t max_lpc_order, double *out);
/* encode.rs:3536-3580 lp_coefficients; coeffs is [32][32] row = order-1; returns #orders */
int orc_lp_coefficients(const double *ac, int n_ac, double coeffs[ORC_MAX_LPC][ORC_MAX_LPC],
                        double *errors);
/* encode.rs:3656-3684 subframe_bits_by_order; returns #orders emitted */
int orc_subframe_bits_by_order(uint32_t bps, uint32_t precision, uint32_t sample_count,
                               const double *errors, int n_orders, double *bits);
/* encode.rs:3688-3702 compute_best_order; returns order (>=1) or 0 = NoBestLpcOrder */
int orc_compute_best_order(uint32_t bps, uint32_t precision, uint32_t sample_count,
                           const double *errors, int n_orders);
/* encode.rs:3334-3401 LpcParameters::quantize.
 * returns 0 ok, 1 ZeroLpCoefficients, 2 LpNegativeShiftError */
int orc_quantize(int order, const double *coeffs, uint32_t precision, int32_t *qlp, uint32_t *shift);
/* encode.rs:3174-3203 encode_residuals; returns 0 ok, 1 ResidualOverflow */
int orc_encode_residuals(int order, const int32_t *qlp, uint32_t shift, const int32_t *channel,
                         uint32_t n, int32_t *residuals);
/* encode.rs:3305-3315 precision table */
uint32_t orc_lpc_precision(uint32_t n);

/* ---- frame level ---- */

/* encode.rs:2259-2439 encode_frame for one block of planar samples
 * (channels[c] points at n samples).  Appends the frame bytes to *out
 * (realloc'd) and fills *plan if non-NULL.  Returns 0 or a negative error. */
int orc_encode_frame(const orc_options *opts, uint32_t sample_rate, uint32_t bps,
                     uint32_t n_channels, const int32_t *const *channels, uint32_t n,
                     uint64_t frame_number, int subset_header, uint8_t **out, size_t *out_len,
                     size_t *out_cap, orc_frame_plan *plan);

/* residuals of the subframe `sub_index` of the last frame encoded with `plan`
 * recomputed from its decision record (for tests): returns count */
int orc_subframe_residuals(const orc_subframe_plan *sp, const int32_t *cand_samples, uint32_t n,
                           int32_t *residuals);

/* ---- stream level: FlacSampleWriter::new/write/finalize semantics
 * (encode.rs:487-627, 1882-2110) into a memory buffer ---- */
typedef struct {
    uint64_t frames;
    uint64_t samples_written;   /* per channel */
    uint32_t min_frame_size, max_frame_size;
    uint8_t md5[16];
    uint64_t first_frame_offset;
} orc_stream_stats;

enum {
    ORC_OK = 0,
    ORC_ERR_INVALID_BPS = -1,
    ORC_ERR_INVALID_SAMPLE_RATE = -2,
    ORC_ERR_EXCESSIVE_CHANNELS = -3,
    ORC_ERR_NOT_DIVISIBLE = -4,
    ORC_ERR_INVALID_TOTAL = -5,
    ORC_ERR_EXCESSIVE_TOTAL = -6,
    ORC_ERR_SAMPLE_COUNT_MISMATCH = -7,
    ORC_ERR_NO_SAMPLES = -8,
    ORC_ERR_IO = -9,
    ORC_ERR_OPTIONS = -10,
    ORC_ERR_UNSUPPORTED = -11 /* reference would panic (e.g. >64 partitions) */
};

/* Encode `n_interleaved` interleaved samples in one go.  total_known != 0
 * passes Some(n_interleaved) as total_samples (placeholder SEEKTABLE).
 * threads: 1 = sequential; >1 = frame-parallel workers (NOT something the
 * reference does; output is identical, used only as a stronger CPU point). */
/* VORBIS_COMMENT block (metadata/mod.rs:2218-2232, 2512-2536).  vendor == NULL means
 * VorbisComment::default()'s "flac-codec 1.3.2"; fields are "NAME=value" strings as built by
 * Options::tag (encode.rs:1513-1520 -> VorbisComment::insert, metadata/mod.rs:2353-2360). */
typedef struct {
    const char *vendor;
    const char *const *fields;
    uint32_t n_fields;
} orc_vorbis_comment;

/* orc_encode_stream with an optional VORBIS_COMMENT (NULL = none) */
int orc_encode_stream_vc(const orc_options *opts, const orc_vorbis_comment *vc,
                         uint32_t sample_rate, uint32_t bps, uint32_t channels,
                         const int32_t *interleaved, uint64_t n_interleaved, int total_known,
                         int threads, uint8_t **out, size_t *out_len, orc_stream_stats *stats);

int orc_encode_stream(const orc_options *opts, uint32_t sample_rate, uint32_t bps,
                      uint32_t channels, const int32_t *interleaved, uint64_t n_interleaved,
                      int total_known, int threads, uint8_t **out, size_t *out_len,
                      orc_stream_stats *stats);

/* Decode a whole .flac (decode.rs:1388-1856 restated) to interleaved i32.
 * Verifies CRC-8/CRC-16 and the STREAMINFO MD5 (if non-zero).
 * Returns 0 or negative error. */
typedef struct {
    uint32_t sample_rate, channels, bps;
    uint32_t min_block, max_block, min_frame, max_frame;
    uint64_t total_samples;
    uint8_t md5[16];
    int md5_ok;      /* 1 match, 0 mismatch, -1 not present */
    uint64_t frames;
    uint32_t n_seekpoints;
} orc_decoded_info;

int orc_decode_stream(const uint8_t *data, size_t len, int32_t **out_interleaved,
                      uint64_t *out_count, orc_decoded_info *info);

/* helpers exposed for tests */
void orc_md5(const uint8_t *data, size_t len, uint8_t out[16]);
uint8_t orc_crc8(const uint8_t *data, size_t len);
uint16_t orc_crc16(const uint8_t *data, size_t len);
void orc_free(void *p);

/* default option sets (encode.rs:1376-1408, 1635-1657) */
void orc_options_default(orc_options *o);
void orc_options_fast(orc_options *o);
void orc_options_best(orc_options *o);

#ifdef __cplusplus
}
#endif
#endif
