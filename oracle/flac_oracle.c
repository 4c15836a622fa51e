/*
 * flac_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on
 * the product path).  Plain-C restatement of the reference encoder
 * (tuffy/flac-codec 1.3.2).  See flac_oracle.h for the pinning statement.
 *
 * Build: see oracle/Makefile  (-O2 -ffp-contract=off, no -ffast-math: the
 * f64 steps must round exactly like the Rust reference, which never
 * contracts a*b+c into an FMA unless `mul_add` is written).
 *
 * Rust release-mode integer semantics are mirrored: + - * wrap silently,
 * `as` casts truncate (int) or saturate (float->int), >> on signed is
 * arithmetic.
 */
#define _GNU_SOURCE
#include "flac_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* bit writer: bitstream-io BitWriter<_, BigEndian> / BitRecorder      */
/* (third-party, un-vendored; MSB-first packing per RFC 9639 sec. 5)   */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t *buf;
    size_t len, cap;
    uint64_t acc;   /* pending bits, right-aligned */
    unsigned accn;  /* number of pending bits (< 8 between calls) */
    uint32_t total; /* BitRecorder<u32,_>::written() (wrapping u32) */
    int err;
} bitw;

static void bw_init(bitw *w) { memset(w, 0, sizeof *w); }
static void bw_clear(bitw *w) {
    w->len = 0;
    w->acc = 0;
    w->accn = 0;
    w->total = 0;
    w->err = 0;
}
static void bw_free(bitw *w) {
    free(w->buf);
    memset(w, 0, sizeof *w);
}
static void bw_reserve(bitw *w, size_t extra) {
    if (w->len + extra > w->cap) {
        size_t nc = w->cap ? w->cap * 2 : 4096;
        while (nc < w->len + extra) nc *= 2;
        w->buf = (uint8_t *)realloc(w->buf, nc);
        w->cap = nc;
    }
}
/* write the low n (<= 32) bits of v, MSB first */
static inline void bw_put(bitw *w, unsigned n, uint32_t v) {
    if (n == 0) return;
    uint64_t mask = (n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1);
    w->acc = (w->acc << n) | ((uint64_t)v & mask);
    w->accn += n;
    w->total += n;
    if (w->accn >= 8) {
        bw_reserve(w, 8);
        while (w->accn >= 8) {
            w->buf[w->len++] = (uint8_t)(w->acc >> (w->accn - 8));
            w->accn -= 8;
        }
    }
}
static inline void bw_put64(bitw *w, unsigned n, uint64_t v) {
    if (n > 32) {
        bw_put(w, n - 32, (uint32_t)(v >> 32));
        bw_put(w, 32, (uint32_t)v);
    } else
        bw_put(w, n, (uint32_t)v);
}
/* write_unary::<1>(q): q zeros then a one (stop bit 1) */
static inline void bw_unary1(bitw *w, uint32_t q) {
    while (q >= 32) {
        bw_put(w, 32, 0);
        q -= 32;
    }
    bw_put(w, q + 1, 1);
}
/* write_unary::<0>(q): q ones then a zero (stop bit 0) */
static inline void bw_unary0(bitw *w, uint32_t q) {
    while (q >= 31) {
        bw_put(w, 31, 0x7FFFFFFFu);
        q -= 31;
    }
    bw_put(w, q + 1, ((1u << q) - 1) << 1);
}
/* write_signed_counted(bits, v): two's complement in `bits` bits; errors when
 * the value does not fit (bitstream-io "excessive value for bits written") */
static inline int bw_signed(bitw *w, unsigned bits, int64_t v) {
    int64_t lo = -((int64_t)1 << (bits - 1)), hi = ((int64_t)1 << (bits - 1)) - 1;
    if (v < lo || v > hi) {
        w->err = 1;
        return -1;
    }
    bw_put64(w, bits, (uint64_t)v & ((bits >= 64) ? ~0ull : (((uint64_t)1 << bits) - 1)));
    return 0;
}
/* byte_align(): pad with zero bits */
static inline void bw_align(bitw *w) {
    if (w->accn) bw_put(w, 8 - w->accn, 0);
}
/* BitRecorder::playback(): replay all recorded bits into dst */
static void bw_playback(const bitw *src, bitw *dst) {
    size_t i = 0;
    if (dst->accn == 0) {
        bw_reserve(dst, src->len + 8);
        memcpy(dst->buf + dst->len, src->buf, src->len);
        dst->len += src->len;
        dst->total += (uint32_t)(src->len * 8);
        i = src->len;
    }
    for (; i + 4 <= src->len; i += 4) {
        uint32_t v = ((uint32_t)src->buf[i] << 24) | ((uint32_t)src->buf[i + 1] << 16) |
                     ((uint32_t)src->buf[i + 2] << 8) | src->buf[i + 3];
        bw_put(dst, 32, v);
    }
    for (; i < src->len; i++) bw_put(dst, 8, src->buf[i]);
    if (src->accn) bw_put(dst, src->accn, (uint32_t)src->acc);
}

/* ------------------------------------------------------------------ */
/* CRC-8 (poly 0x07) and CRC-16 (poly 0x8005), crc.rs:99-188.          */
/* Tables regenerated from the polynomials (MSB-first, init 0).        */
/* ------------------------------------------------------------------ */
static uint8_t crc8_table[256];
static uint16_t crc16_table[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void) {
    for (int i = 0; i < 256; i++) {
        uint8_t c = (uint8_t)i;
        for (int b = 0; b < 8; b++) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
        crc8_table[i] = c;
        uint16_t d = (uint16_t)(i << 8);
        for (int b = 0; b < 8; b++)
            d = (d & 0x8000) ? (uint16_t)((d << 1) ^ 0x8005) : (uint16_t)(d << 1);
        crc16_table[i] = d;
    }
}
uint8_t orc_crc8(const uint8_t *data, size_t len) {
    pthread_once(&crc_once, crc_init);
    uint8_t c = 0;
    for (size_t i = 0; i < len; i++) c = crc8_table[c ^ data[i]]; /* crc.rs:128 */
    return c;
}
uint16_t orc_crc16(const uint8_t *data, size_t len) {
    pthread_once(&crc_once, crc_init);
    uint16_t c = 0;
    for (size_t i = 0; i < len; i++)
        c = (uint16_t)(crc16_table[(uint8_t)(c >> 8) ^ data[i]] ^ (uint16_t)(c << 8)); /* crc.rs:182 */
    return c;
}

/* ------------------------------------------------------------------ */
/* MD5 (RFC 1321) -- crate `md5` 0.8, un-vendored                       */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t s[4];
    uint64_t n;
    uint8_t buf[64];
} md5_ctx;
static const uint32_t md5_k[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t md5_r[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                  5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
static void md5_block(md5_ctx *c, const uint8_t *p) {
    uint32_t m[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
               ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = c->s[0], b = c->s[1], cc = c->s[2], d = c->s[3];
    for (int i = 0; i < 64; i++) {
        uint32_t f;
        int g;
        if (i < 16) {
            f = (b & cc) | (~b & d);
            g = i;
        } else if (i < 32) {
            f = (d & b) | (~d & cc);
            g = (5 * i + 1) & 15;
        } else if (i < 48) {
            f = b ^ cc ^ d;
            g = (3 * i + 5) & 15;
        } else {
            f = cc ^ (b | ~d);
            g = (7 * i) & 15;
        }
        uint32_t t = a + f + md5_k[i] + m[g];
        a = d;
        d = cc;
        cc = b;
        b = b + ((t << md5_r[i]) | (t >> (32 - md5_r[i])));
    }
    c->s[0] += a;
    c->s[1] += b;
    c->s[2] += cc;
    c->s[3] += d;
}
static void md5_init(md5_ctx *c) {
    c->s[0] = 0x67452301;
    c->s[1] = 0xefcdab89;
    c->s[2] = 0x98badcfe;
    c->s[3] = 0x10325476;
    c->n = 0;
}
static void md5_update(md5_ctx *c, const uint8_t *p, size_t len) {
    size_t have = (size_t)(c->n & 63);
    c->n += len;
    if (have) {
        size_t need = 64 - have;
        if (len < need) {
            memcpy(c->buf + have, p, len);
            return;
        }
        memcpy(c->buf + have, p, need);
        md5_block(c, c->buf);
        p += need;
        len -= need;
    }
    while (len >= 64) {
        md5_block(c, p);
        p += 64;
        len -= 64;
    }
    if (len) memcpy(c->buf, p, len);
}
static void md5_final(const md5_ctx *c0, uint8_t out[16]) {
    md5_ctx c = *c0; /* the reference finalizes a clone: encode.rs:2100 */
    uint64_t bits = c.n * 8;
    uint8_t pad[72];
    size_t have = (size_t)(c.n & 63);
    size_t padlen = (have < 56) ? (56 - have) : (120 - have);
    memset(pad, 0, sizeof pad);
    pad[0] = 0x80;
    md5_update(&c, pad, padlen);
    uint8_t lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
    md5_update(&c, lenb, 8);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(c.s[i] >> (8 * j));
}
void orc_md5(const uint8_t *data, size_t len, uint8_t out[16]) {
    md5_ctx c;
    md5_init(&c);
    md5_update(&c, data, len);
    md5_final(&c, out);
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ */
/* options: encode.rs:1376-1408 (default), 1635-1644 (fast), 1649-1657 */
/* ------------------------------------------------------------------ */
void orc_options_default(orc_options *o) {
    o->block_size = 4096;
    o->max_partition_order = 5;
    o->mid_side = 1;
    o->max_lpc_order = 8;
    o->window_kind = ORC_WINDOW_TUKEY;
    o->window_param = 0.5f;
    o->exhaustive = 1;
    o->padding = 4096;
    o->seektable_mode = 1;
    o->seektable_value = 10;
}
void orc_options_fast(orc_options *o) {
    orc_options_default(o);
    o->block_size = 1152;
    o->mid_side = 0;
    o->max_partition_order = 3;
    o->max_lpc_order = 0;
    o->exhaustive = 0;
}
void orc_options_best(orc_options *o) {
    orc_options_default(o);
    o->block_size = 4096;
    o->mid_side = 1;
    o->max_partition_order = 6;
    o->max_lpc_order = 12;
}

/* ------------------------------------------------------------------ */
/* Window::generate, encode.rs:1725-1783                               */
/* ------------------------------------------------------------------ */
void orc_window_generate(int kind, float p, uint32_t n, double *window) {
    const double PI = 3.14159265358979323846264338327950288; /* std::f64::consts::PI */
    uint32_t i;
    if (kind == ORC_WINDOW_RECTANGLE) {
        for (i = 0; i < n; i++) window[i] = 1.0;
        return;
    }
    if (kind == ORC_WINDOW_HANN) { /* :1730-1740 */
        double np = (double)(uint16_t)n - 1.0;
        for (i = 0; i < n; i++) window[i] = 0.5 - 0.5 * cos(2.0 * PI * (double)i / np);
        return;
    }
    /* Tukey(p), :1741-1781; match arms in source order */
    if (p <= 0.0f) { /* ..=0.0 */
        for (i = 0; i < n; i++) window[i] = 1.0;
    } else if (p >= 1.0f) { /* 1.0.. */
        orc_window_generate(ORC_WINDOW_HANN, 0, n, window);
    } else if (p > 0.0f && p < 1.0f) { /* 0.0..1.0 */
        /* ((p as f64) / 2.0 * len as f64) as usize, then checked_sub(1) */
        double t = (double)p / 2.0 * (double)n;
        uint64_t tu = (t >= 18446744073709551615.0) ? UINT64_MAX : (uint64_t)t;
        if (tu == 0) { /* None => rectangle (:1773) */
            for (i = 0; i < n; i++) window[i] = 1.0;
            return;
        }
        uint64_t np = tu - 1;
        /* get_disjoint_mut([0..np, np..len-np, len-np..len]) :1752-1756:
         * fails (=> rectangle) when out of bounds / overlapping */
        if (np > n || n - np < np) {
            for (i = 0; i < n; i++) window[i] = 1.0;
            return;
        }
        for (i = 0; i < n; i++) window[i] = 1.0; /* mid.fill(1.0) */
        double npf = (double)(uint16_t)np;
        for (i = 0; i < (uint32_t)np; i++) {
            double x = 0.5 - 0.5 * cos(PI * (double)i / npf); /* :1764 */
            window[i] = x;
            window[n - 1 - i] = x;
        }
    } else { /* NaN => Tukey(0.5) (:1778-1780) */
        orc_window_generate(ORC_WINDOW_TUKEY, 0.5f, n, window);
    }
}

/* ------------------------------------------------------------------ */
/* autocorrelate, encode.rs:3478-3501.  f64 `sum()` is a left fold      */
/* starting from -0.0; every product is rounded before the add.        */
/* ------------------------------------------------------------------ */
int orc_autocorrelate(const double *windowed, uint32_t n, uint32_t max_lpc_order, double *out) {
    int count = 0;
    for (uint32_t lag = 0; lag <= max_lpc_order; lag++) {
        if (lag >= n) return count; /* tail.is_empty() */
        double s = -0.0;
        const double *tail = windowed + lag;
        uint32_t m = n - lag;
        for (uint32_t i = 0; i < m; i++) {
            double prod = windowed[i] * tail[i];
            s = s + prod;
        }
        out[count++] = s;
    }
    return count;
}

/* ------------------------------------------------------------------ */
/* lp_coefficients (Levinson-Durbin), encode.rs:3536-3580               */
/* ------------------------------------------------------------------ */
int orc_lp_coefficients(const double *ac, int n_ac, double coeffs[ORC_MAX_LPC][ORC_MAX_LPC],
                        double *errors) {
    if (n_ac < 2) return 0; /* the reference panics (:3543) */
    double k = ac[1] / ac[0];
    coeffs[0][0] = k;
    errors[0] = ac[0] * (1.0 - k * k);
    int count = 1;
    for (int i = 1; i < n_ac - 1; i++) {
        const double *c = coeffs[i - 1];
        double err = errors[i - 1];
        /* q = next - sum_{j<i} ac[i-j]*c[j]   (prev.iter().rev().zip(coeffs)) */
        double s = -0.0;
        for (int j = 0; j < i; j++) {
            double prod = ac[i - j] * c[j];
            s = s + prod;
        }
        double q = ac[i + 1] - s;
        double kk = q / err;
        for (int j = 0; j < i; j++) {
            double t = kk * c[i - 1 - j];
            coeffs[i][j] = c[j] - t;
        }
        coeffs[i][i] = kk;
        errors[i] = err * (1.0 - kk * kk);
        count++;
    }
    return count;
}

/* encode.rs:3305-3315 */
uint32_t orc_lpc_precision(uint32_t n) {
    if (n <= 192) return 7;
    if (n <= 384) return 8;
    if (n <= 576) return 9;
    if (n <= 1152) return 10;
    if (n <= 2304) return 11;
    if (n <= 4608) return 12;
    return 13;
}

/* ------------------------------------------------------------------ */
/* subframe_bits_by_order / compute_best_order, encode.rs:3656-3702     */
/* Quirk: `.max(0.0)` binds to the constant (2*LN_2), so bits-per-      */
/* residual is NOT clamped at 0 (:3675).                                */
/* ------------------------------------------------------------------ */
int orc_subframe_bits_by_order(uint32_t bps, uint32_t precision, uint32_t sample_count,
                               const double *errors, int n_orders, double *bits) {
    const double LN_2 = 0.693147180559945309417232121458176568;
    double error_scale = 0.5 / (double)sample_count;
    double denom = fmax(2.0 * LN_2, 0.0);
    int count = 0;
    for (int i = 0; i < n_orders; i++) {
        if (!(errors[i] > 0.0)) break; /* take_while(error > 0.0) */
        uint32_t order = (uint32_t)(i + 1);
        uint32_t header_bits = order * (bps + precision);
        double bpr = log(errors[i] * error_scale) / denom;
        bits[count++] = fma(bpr, (double)(uint16_t)(sample_count - order), (double)header_bits);
    }
    return count;
}

/* f64::total_cmp ordering key (core::f64::total_cmp) */
static inline int64_t total_key(double x) {
    int64_t b;
    memcpy(&b, &x, 8);
    b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
    return b;
}

int orc_compute_best_order(uint32_t bps, uint32_t precision, uint32_t sample_count,
                           const double *errors, int n_orders) {
    double bits[ORC_MAX_LPC];
    int cnt = orc_subframe_bits_by_order(bps, precision, sample_count, errors, n_orders, bits);
    if (cnt == 0) return 0; /* NoBestLpcOrder */
    int best = 0;
    for (int i = 1; i < cnt; i++)
        if (total_key(bits[i]) < total_key(bits[best])) best = i; /* min_by: first min wins */
    return best + 1;
}

/* ------------------------------------------------------------------ */
/* LpcParameters::quantize, encode.rs:3334-3401                         */
/* ------------------------------------------------------------------ */
static inline int32_t f64_to_i32_sat(double x) { /* Rust `as i32` */
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}

int orc_quantize(int order, const double *coeffs, uint32_t precision, int32_t *qlp,
                 uint32_t *shift_out) {
    const int32_t MAX_SHIFT = 15, MIN_SHIFT = -16;
    int32_t max_coeff = (1 << (precision - 1)) - 1;
    int32_t min_coeff = -(1 << (precision - 1));
    /* l = max |c| by total_cmp, must be > 0.0 */
    double l = fabs(coeffs[0]);
    for (int i = 1; i < order; i++) {
        double a = fabs(coeffs[i]);
        if (total_key(a) >= total_key(l)) l = a; /* max_by: last max wins (same value) */
    }
    if (!(l > 0.0)) return 1; /* ZeroLpCoefficients */
    double error = 0.0;
    int32_t fl = f64_to_i32_sat(floor(log2(l)));
    /* (precision-1) as i32 - floor(log2 l) as i32 - 1, wrapping */
    int32_t sh = (int32_t)((uint32_t)(int32_t)(precision - 1) - (uint32_t)fl - 1u);
    if (sh > MAX_SHIFT) sh = MAX_SHIFT;
    if (sh >= 0) {
        double scale = (double)(1 << sh);
        for (int i = 0; i < order; i++) {
            double sum = fma(coeffs[i], scale, error); /* mul_add :3372 */
            int32_t q = f64_to_i32_sat(round(sum));
            if (q < min_coeff) q = min_coeff;
            if (q > max_coeff) q = max_coeff;
            error = sum - (double)q;
            qlp[i] = q;
        }
        *shift_out = (uint32_t)sh;
        return 0;
    } else if (sh >= MIN_SHIFT) {
        double scale = (double)(1 << (-sh));
        for (int i = 0; i < order; i++) {
            double sum = (coeffs[i] / scale) + error; /* :3391 */
            int32_t q = f64_to_i32_sat(round(sum));
            if (q < min_coeff) q = min_coeff;
            if (q > max_coeff) q = max_coeff;
            error = sum - (double)q;
            qlp[i] = q;
        }
        *shift_out = 0;
        return 0;
    }
    return 2; /* LpNegativeShiftError */
}

/* ------------------------------------------------------------------ */
/* LpcSubframeParameters::encode_residuals, encode.rs:3174-3203         */
/* ------------------------------------------------------------------ */
int orc_encode_residuals(int order, const int32_t *qlp, uint32_t shift, const int32_t *channel,
                         uint32_t n, int32_t *residuals) {
    for (uint32_t i = (uint32_t)order; i < n; i++) {
        int64_t sum = 0;
        for (int j = 0; j < order; j++) sum += (int64_t)channel[i - 1 - j] * (int64_t)qlp[j];
        int32_t pred = (int32_t)(sum >> shift); /* `as i32` truncates */
        int64_t r = (int64_t)channel[i] - (int64_t)pred;
        if (r < INT32_MIN || r > INT32_MAX) return 1; /* checked_sub -> ResidualOverflow */
        residuals[i - order] = (int32_t)r;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* per-channel caches, encode.rs:1810-1851                              */
/* ------------------------------------------------------------------ */
typedef struct {
    bitw w;
    orc_subframe_plan plan;
} recorder;

typedef struct {
    int32_t *fixed_buf[4];
    recorder fixed_out, lpc_out, const_out, verb_out;
    double *window;
    uint32_t window_len; /* Window::apply regenerates iff len changed (:1791) */
    double *windowed;
    int32_t *residuals;
    int32_t *wasted;
    uint32_t cap;
} chan_cache;

typedef struct {
    chan_cache channels[ORC_MAX_CHANNELS];
    int32_t *average, *difference;
    uint32_t cap;
    chan_cache left, right, avg, diff;
} enc_caches;

/* ------------------------------------------------------------------ */
/* join / try_join / vec_map, encode.rs:3964-4010: the reference forks   */
/* at most channel-level tasks per frame with rayon (L || R, then        */
/* M || S, each subframe FIXED || LPC; frames stay sequential).  A small  */
/* work-helping pool restates that task structure for the timed CPU       */
/* baseline (threads < 0 in orc_encode_stream: |threads| threads); the    */
/* bytes are the same either way.                                         */
/* ------------------------------------------------------------------ */
typedef struct fj_task {
    void (*fn)(void *);
    void *arg;
    volatile int done;
} fj_task;
static struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    fj_task *q[64];
    int nq, stop, nworkers;
    pthread_t th[16];
} g_fj = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, {0}};
static volatile int g_fj_on = 0;

static inline void fj_pause(void) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
static fj_task *fj_pop_locked(void) { return g_fj.nq ? g_fj.q[--g_fj.nq] : NULL; }
static void *fj_worker(void *unused) {
    (void)unused;
    pthread_mutex_lock(&g_fj.mu);
    for (;;) {
        fj_task *t = fj_pop_locked();
        if (!t) {
            if (g_fj.stop) break;
            /* like rayon's workers: spin briefly for the next fork before going to sleep */
            pthread_mutex_unlock(&g_fj.mu);
            int seen = 0;
            for (int spin = 0; spin < 20000 && !seen; spin++) {
                seen = *(volatile int *)&g_fj.nq || *(volatile int *)&g_fj.stop;
                fj_pause();
            }
            pthread_mutex_lock(&g_fj.mu);
            if (!seen && !g_fj.nq && !g_fj.stop) pthread_cond_wait(&g_fj.cv, &g_fj.mu);
            continue;
        }
        pthread_mutex_unlock(&g_fj.mu);
        t->fn(t->arg);
        pthread_mutex_lock(&g_fj.mu);
        t->done = 1;
        pthread_cond_broadcast(&g_fj.cv);
    }
    pthread_mutex_unlock(&g_fj.mu);
    return NULL;
}
static void fj_start(int threads) { /* `threads` in total, the caller included */
    g_fj.stop = 0;
    g_fj.nworkers = threads - 1 > 15 ? 15 : threads - 1;
    for (int i = 0; i < g_fj.nworkers; i++) pthread_create(&g_fj.th[i], NULL, fj_worker, NULL);
    g_fj_on = 1;
}
static void fj_stop(void) {
    pthread_mutex_lock(&g_fj.mu);
    g_fj.stop = 1;
    pthread_cond_broadcast(&g_fj.cv);
    pthread_mutex_unlock(&g_fj.mu);
    for (int i = 0; i < g_fj.nworkers; i++) pthread_join(g_fj.th[i], NULL);
    g_fj.nworkers = 0;
    g_fj_on = 0;
}
/* run n tasks: the first inline, the rest offered to the pool; while waiting, run queued tasks */
static void fj_run(fj_task *tasks, int n) {
    if (!g_fj_on || n <= 1) {
        for (int i = 0; i < n; i++) tasks[i].fn(tasks[i].arg);
        return;
    }
    pthread_mutex_lock(&g_fj.mu);
    for (int i = n - 1; i >= 1; i--) {
        tasks[i].done = 0;
        g_fj.q[g_fj.nq++] = &tasks[i];
    }
    pthread_cond_broadcast(&g_fj.cv);
    pthread_mutex_unlock(&g_fj.mu);
    tasks[0].fn(tasks[0].arg);
    pthread_mutex_lock(&g_fj.mu);
    for (;;) {
        int pending = 0;
        for (int i = 1; i < n; i++) pending += !tasks[i].done;
        if (!pending) break;
        fj_task *t = fj_pop_locked();
        if (t) {
            pthread_mutex_unlock(&g_fj.mu);
            t->fn(t->arg);
            pthread_mutex_lock(&g_fj.mu);
            t->done = 1;
            pthread_cond_broadcast(&g_fj.cv);
        } else {
            pthread_mutex_unlock(&g_fj.mu);
            for (int spin = 0; spin < 2000; spin++) {
                int all = 1;
                for (int i = 1; i < n; i++) all &= tasks[i].done;
                if (all || *(volatile int *)&g_fj.nq) break;
                fj_pause();
            }
            pthread_mutex_lock(&g_fj.mu);
        }
    }
    pthread_mutex_unlock(&g_fj.mu);
}

static void cc_reserve(chan_cache *c, uint32_t n) {
    if (n <= c->cap) return;
    for (int i = 0; i < 4; i++) c->fixed_buf[i] = (int32_t *)realloc(c->fixed_buf[i], 4u * n);
    c->window = (double *)realloc(c->window, 8u * n);
    c->windowed = (double *)realloc(c->windowed, 8u * n);
    c->residuals = (int32_t *)realloc(c->residuals, 4u * n);
    c->wasted = (int32_t *)realloc(c->wasted, 4u * n);
    c->cap = n;
    c->window_len = 0;
}
static void cc_free(chan_cache *c) {
    for (int i = 0; i < 4; i++) free(c->fixed_buf[i]);
    free(c->window);
    free(c->windowed);
    free(c->residuals);
    free(c->wasted);
    bw_free(&c->fixed_out.w);
    bw_free(&c->lpc_out.w);
    bw_free(&c->const_out.w);
    bw_free(&c->verb_out.w);
    memset(c, 0, sizeof *c);
}
static void caches_free(enc_caches *e) {
    for (int i = 0; i < ORC_MAX_CHANNELS; i++) cc_free(&e->channels[i]);
    cc_free(&e->left);
    cc_free(&e->right);
    cc_free(&e->avg);
    cc_free(&e->diff);
    free(e->average);
    free(e->difference);
    memset(e, 0, sizeof *e);
}

/* ------------------------------------------------------------------ */
/* SubframeHeader, stream.rs:1375-1413 + 1527-1567                      */
/* ------------------------------------------------------------------ */
static void write_subframe_header(bitw *w, int type, unsigned order, uint32_t wasted) {
    bw_put(w, 1, 0);
    uint32_t code = 0;
    switch (type) {
    case ORC_SUB_CONSTANT: code = 0; break;
    case ORC_SUB_VERBATIM: code = 1; break;
    case ORC_SUB_FIXED: code = 8 + order; break;
    case ORC_SUB_LPC: code = order + 31; break;
    }
    bw_put(w, 6, code);
    if (wasted == 0)
        bw_put(w, 1, 0);
    else {
        bw_put(w, 1, 1);
        bw_unary1(w, wasted - 1);
    }
}

static void plan_reset(orc_subframe_plan *p, int type, uint32_t wasted, uint32_t bps) {
    memset(p, 0, sizeof *p);
    p->type = (uint8_t)type;
    p->wasted = (uint8_t)wasted;
    p->bps = (uint8_t)bps;
}

/* encode.rs:2982-2998 */
static int encode_constant_subframe(recorder *r, int32_t sample, uint32_t bps, uint32_t wasted) {
    plan_reset(&r->plan, ORC_SUB_CONSTANT, wasted, bps);
    write_subframe_header(&r->w, ORC_SUB_CONSTANT, 0, wasted);
    if (bw_signed(&r->w, bps, sample)) return -1;
    r->plan.bits = r->w.total;
    return 0;
}
/* encode.rs:3000-3018 */
static int encode_verbatim_subframe(recorder *r, const int32_t *ch, uint32_t n, uint32_t bps,
                                    uint32_t wasted) {
    plan_reset(&r->plan, ORC_SUB_VERBATIM, wasted, bps);
    write_subframe_header(&r->w, ORC_SUB_VERBATIM, 0, wasted);
    for (uint32_t i = 0; i < n; i++)
        if (bw_signed(&r->w, bps, ch[i])) return -1;
    r->plan.bits = r->w.total;
    return 0;
}

/* ------------------------------------------------------------------ */
/* write_residuals, encode.rs:3747-3962                                 */
/* ------------------------------------------------------------------ */
enum { PH_STANDARD = 0, PH_ESCAPED = 1, PH_CONSTANT = 2 };
typedef struct {
    uint8_t kind, rice, escape;
    uint32_t off, len;
} part;

/* Partition::new, :3765-3831.  returns 0 = None */
static int partition_new(const int32_t *res, uint32_t len, uint32_t rice_max, uint32_t *est,
                         part *out) {
    uint16_t n_p = (uint16_t)len;
    if (n_p == 0) return 0;
    uint64_t sum = 0;
    for (uint32_t i = 0; i < len; i++) {
        int32_t v = res[i];
        sum += (uint64_t)(v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v); /* unsigned_abs */
    }
    if (sum > 0) {
        uint32_t rice;
        if (sum > (uint64_t)n_p) {
            double needed = ceil(log2((double)sum / (double)n_p));
            uint32_t bits_needed = (needed >= 4294967295.0) ? UINT32_MAX
                                   : (needed <= 0.0)        ? 0u
                                                            : (uint32_t)needed;
            if (bits_needed < rice_max) {
                rice = bits_needed;
            } else {
                uint32_t ilog = 63u - (uint32_t)__builtin_clzll(sum);
                uint32_t escape = ilog + 2;
                if (escape > 31) return 0; /* try_into SignedBitCount<31> fails */
                *est += escape * (uint32_t)n_p;
                out->kind = PH_ESCAPED;
                out->rice = 0xFF;
                out->escape = (uint8_t)escape;
                return 1;
            }
        } else
            rice = 0;
        uint64_t t = (rice > 0) ? (sum >> (rice - 1)) : (sum << 1);
        if (t > 0xFFFFFFFFull) return 0; /* u32::try_from fails */
        uint32_t psize = 4u + ((1u + rice) * (uint32_t)n_p) + (uint32_t)t - ((uint32_t)n_p / 2u);
        *est += psize;
        out->kind = PH_STANDARD;
        out->rice = (uint8_t)rice;
        out->escape = 0;
        return 1;
    }
    out->kind = PH_CONSTANT;
    out->rice = 0xFF;
    out->escape = 0;
    return 1;
}

/* best_partitions, :3865-3896.  returns #partitions (>=1) or <0 when the
 * reference would panic (ArrayVec overflow past MAX_PARTITIONS, :3880) */
static int best_partitions(const orc_options *o, uint32_t rice_max, uint32_t block_size,
                           const int32_t *res, uint32_t nres, part *best) {
    uint32_t tz = (uint32_t)__builtin_ctz(block_size);
    uint32_t max_po = tz < o->max_partition_order ? tz : o->max_partition_order;
    int have = 0, best_n = 0;
    uint32_t best_est = 0;
    part cur[ORC_MAX_PARTITIONS];
    for (uint32_t po = 0; po <= max_po; po++) {
        uint32_t plen = block_size >> po; /* block_size / partition_count */
        if (nres == 0) continue;          /* no chunks -> p.is_empty() -> dropped */
        uint32_t count = (nres + plen - 1) / plen;
        uint32_t first = nres - (count - 1) * plen; /* rchunks().rev(): short chunk first */
        uint32_t est = 0, off = 0;
        int ok = 1;
        for (uint32_t k = 0; k < count; k++) {
            uint32_t len = (k == 0) ? first : plen;
            if (k >= ORC_MAX_PARTITIONS) return -1; /* ArrayVec::extend panics */
            if (!partition_new(res + off, len, rice_max, &est, &cur[k])) {
                ok = 0;
                break;
            }
            cur[k].off = off;
            cur[k].len = len;
            off += len;
        }
        if (!ok) continue;
        if ((count & (count - 1)) != 0) continue; /* !is_power_of_two */
        if (!have || est < best_est) {            /* min_by_key: first min wins */
            have = 1;
            best_est = est;
            best_n = (int)count;
            memcpy(best, cur, sizeof(part) * count);
        }
    }
    if (!have) { /* :3887-3895 */
        best[0].kind = PH_ESCAPED;
        best[0].rice = 0xFF;
        best[0].escape = 31;
        best[0].off = 0;
        best[0].len = nres;
        return 1;
    }
    return best_n;
}

static inline uint32_t zigzag(int32_t s) { /* :3845-3849 */
    return (s < 0) ? ((((uint32_t)0 - (uint32_t)s) - 1u) << 1) + 1u : ((uint32_t)s) << 1;
}

static int write_residuals(const orc_options *o, int use_rice2, recorder *r, uint32_t order,
                           const int32_t *res, uint32_t nres) {
    bitw *w = &r->w;
    part parts[ORC_MAX_PARTITIONS];
    uint32_t block_size = order + nres;
    uint32_t rice_max = use_rice2 ? 31u : 15u;
    int np = best_partitions(o, rice_max, block_size, res, nres, parts);
    if (np < 0) return ORC_ERR_UNSUPPORTED;
    int method = 0;
    if (use_rice2) { /* try_reduce_rice, :3929-3942 */
        for (int i = 0; i < np; i++)
            if (parts[i].kind == PH_STANDARD && parts[i].rice >= 15) method = 1;
    }
    unsigned hb = method ? 5 : 4;
    uint32_t all_ones = method ? 31u : 15u;
    bw_put(w, 2, (uint32_t)method);
    uint32_t porder = 31u - (uint32_t)__builtin_clz((uint32_t)np); /* partitions.len().ilog2() */
    bw_put(w, 4, porder);
    r->plan.coding_method = (uint8_t)method;
    r->plan.partition_order = (uint8_t)porder;
    r->plan.n_partitions = (uint32_t)np;
    for (int i = 0; i < np; i++) {
        const part *p = &parts[i];
        const int32_t *pr = res + p->off;
        r->plan.rice[i] = p->rice;
        r->plan.escape_bits[i] = p->escape;
        r->plan.part_len[i] = (uint16_t)p->len;
        if (p->kind == PH_STANDARD) {
            bw_put(w, hb, p->rice);
            unsigned k = p->rice;
            uint32_t mask = k ? ((1u << k) - 1u) : 0u;
            for (uint32_t j = 0; j < p->len; j++) {
                uint32_t u = zigzag(pr[j]);
                bw_unary1(w, u >> k);
                bw_put(w, k, u & mask);
            }
        } else if (p->kind == PH_ESCAPED) {
            bw_put(w, hb, all_ones);
            bw_put(w, 5, p->escape);
            for (uint32_t j = 0; j < p->len; j++)
                if (bw_signed(w, p->escape, pr[j])) return ORC_ERR_IO;
        } else {
            bw_put(w, hb, all_ones);
            bw_put(w, 5, 0);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* encode_fixed_subframe, encode.rs:3020-3088                           */
/* ------------------------------------------------------------------ */
static int encode_fixed_subframe(const orc_options *o, int use_rice2, chan_cache *c, recorder *r,
                                 const int32_t *ch, uint32_t n, uint32_t bps, uint32_t wasted) {
    const int32_t *orders[5];
    uint32_t lens[5];
    int n_orders = 1;
    orders[0] = ch;
    lens[0] = n;
    for (int b = 0; b < 4; b++) {
        const int32_t *prev = orders[n_orders - 1];
        uint32_t plen = lens[n_orders - 1];
        if (plen < 1) break; /* split_at_checked(1) -> None */
        int32_t *buf = c->fixed_buf[b];
        uint32_t m = plen - 1;
        int overflow = 0;
        for (uint32_t i = 0; i < m; i++) {
            int64_t v = (int64_t)prev[i + 1] - (int64_t)prev[i];
            if (v < INT32_MIN || v > INT32_MAX) { /* checked_sub None -> break 'outer */
                overflow = 1;
                break;
            }
            buf[i] = (int32_t)v;
        }
        if (overflow) break;
        if (m == 0) break; /* buf.is_empty() */
        orders[n_orders] = buf;
        lens[n_orders] = m;
        n_orders++;
    }
    uint32_t min_fixed = lens[n_orders - 1];
    int best = 0;
    uint64_t best_sum = 0;
    for (int k = 0; k < n_orders; k++) {
        uint64_t s = 0;
        const int32_t *p = orders[k] + (lens[k] - min_fixed);
        for (uint32_t i = 0; i < min_fixed; i++) {
            int32_t v = p[i];
            s += (uint64_t)(v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v);
        }
        if (k == 0 || s < best_sum) { /* min_by_key: first min wins */
            best = k;
            best_sum = s;
        }
    }
    plan_reset(&r->plan, ORC_SUB_FIXED, wasted, bps);
    r->plan.order = (uint8_t)best;
    write_subframe_header(&r->w, ORC_SUB_FIXED, (unsigned)best, wasted);
    for (int i = 0; i < best; i++)
        if (bw_signed(&r->w, bps, ch[i])) return ORC_ERR_IO;
    int rc = write_residuals(o, use_rice2, r, (uint32_t)best, orders[best], lens[best]);
    r->plan.bits = r->w.total;
    return rc;
}

/* ------------------------------------------------------------------ */
/* LpcParameters::best + encode_lpc_subframe, encode.rs:3090-3136,      */
/* 3145-3172, 3292-3332                                                 */
/* ------------------------------------------------------------------ */
static int encode_lpc_subframe(const orc_options *o, int use_rice2, chan_cache *c, recorder *r,
                               const int32_t *ch, uint32_t n, uint32_t bps, uint32_t wasted) {
    uint32_t max_order = (uint32_t)o->max_lpc_order;
    if (n <= max_order) return -100; /* InsufficientLpcSamples :3300 */
    uint32_t precision = orc_lpc_precision(n);
    /* Window::apply :1785-1801 */
    if (c->window_len != n) {
        orc_window_generate(o->window_kind, o->window_param, n, c->window);
        c->window_len = n;
    }
    for (uint32_t i = 0; i < n; i++) c->windowed[i] = (double)ch[i] * c->window[i];
    double ac[ORC_MAX_LPC + 1];
    int n_ac = orc_autocorrelate(c->windowed, n, max_order, ac);
    static __thread double coeffs[ORC_MAX_LPC][ORC_MAX_LPC];
    double errors[ORC_MAX_LPC];
    int n_orders = orc_lp_coefficients(ac, n_ac, coeffs, errors);
    int order = orc_compute_best_order(bps, precision, n, errors, n_orders);
    if (order == 0) return -101; /* NoBestLpcOrder */
    int32_t qlp[ORC_MAX_LPC];
    uint32_t shift;
    int q = orc_quantize(order, coeffs[order - 1], precision, qlp, &shift);
    if (q) return -102 - q; /* ZeroLpCoefficients / LpNegativeShiftError */
    if (orc_encode_residuals(order, qlp, shift, ch, n, c->residuals)) return -105; /* overflow */

    plan_reset(&r->plan, ORC_SUB_LPC, wasted, bps);
    r->plan.order = (uint8_t)order;
    r->plan.precision = (uint8_t)precision;
    r->plan.shift = (uint8_t)shift;
    memcpy(r->plan.coeffs, qlp, sizeof(int32_t) * (size_t)order);
    bitw *w = &r->w;
    write_subframe_header(w, ORC_SUB_LPC, (unsigned)order, wasted);
    for (int i = 0; i < order; i++)
        if (bw_signed(w, bps, ch[i])) return ORC_ERR_IO;
    bw_put(w, 4, precision - 1);       /* write_count::<0b1111> :3122 */
    bw_put(w, 5, shift & 31u);         /* write::<5,i32>(shift) :3129 */
    for (int i = 0; i < order; i++)
        if (bw_signed(w, precision, qlp[i])) return ORC_ERR_IO;
    int rc = write_residuals(o, use_rice2, r, (uint32_t)order, c->residuals, n - (uint32_t)order);
    r->plan.bits = w->total;
    return rc;
}

typedef struct {
    const orc_options *o;
    int use_rice2;
    chan_cache *c;
    recorder *r;
    const int32_t *ch;
    uint32_t n, bps, wasted;
    int rc;
} sub_job;
static void run_fixed_job(void *a) {
    sub_job *j = (sub_job *)a;
    j->rc = encode_fixed_subframe(j->o, j->use_rice2, j->c, j->r, j->ch, j->n, j->bps, j->wasted);
}
static void run_lpc_job(void *a) {
    sub_job *j = (sub_job *)a;
    j->rc = encode_lpc_subframe(j->o, j->use_rice2, j->c, j->r, j->ch, j->n, j->bps, j->wasted);
}

/* ------------------------------------------------------------------ */
/* encode_subframe, encode.rs:2849-2980.  *status: 0 ok, <0 fatal       */
/* ------------------------------------------------------------------ */
static recorder *encode_subframe(const orc_options *o, int use_rice2, chan_cache *c,
                                 const int32_t *ch, uint32_t n, uint32_t bps, int all_0,
                                 int *status) {
    *status = 0;
    cc_reserve(c, n);
    if (all_0) { /* :2870-2875 */
        bw_clear(&c->const_out.w);
        if (encode_constant_subframe(&c->const_out, ch[0], bps, 0)) *status = ORC_ERR_IO;
        return &c->const_out;
    }
    /* wasted bits :2878-2898 */
    uint32_t wasted = 32;
    int none = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t tz = ch[i] ? (uint32_t)__builtin_ctz((uint32_t)ch[i]) : 32u;
        if (tz == 0) {
            none = 1;
            break;
        }
        if (tz < wasted) wasted = tz;
    }
    if (none)
        wasted = 0;
    else if (wasted == 32) {
        bw_clear(&c->const_out.w);
        if (encode_constant_subframe(&c->const_out, ch[0], bps, 0)) *status = ORC_ERR_IO;
        return &c->const_out;
    } else {
        for (uint32_t i = 0; i < n; i++) c->wasted[i] = ch[i] >> wasted;
        ch = c->wasted;
        bps -= wasted; /* checked_sub(..).unwrap() */
    }

    bw_clear(&c->fixed_out.w);
    recorder *best;
    int fatal = 0;
    if (o->max_lpc_order > 0) {
        bw_clear(&c->lpc_out.w);
        /* join(fixed, lpc), encode.rs:2906-2928 (disjoint scratch: fixed_buf / fixed_out vs
           window, windowed, residuals / lpc_out) */
        sub_job jf = {o, use_rice2, c, &c->fixed_out, ch, n, bps, wasted, 0}, jl = jf;
        jl.r = &c->lpc_out;
        fj_task tk[2] = {{run_fixed_job, &jf, 0}, {run_lpc_job, &jl, 0}};
        fj_run(tk, 2);
        int f = jf.rc, l = jl.rc;
        if (f == ORC_ERR_UNSUPPORTED || l == ORC_ERR_UNSUPPORTED) fatal = ORC_ERR_UNSUPPORTED;
        if (f == 0 && l == 0)
            best = (c->lpc_out.w.total < c->fixed_out.w.total) ? &c->lpc_out : &c->fixed_out;
        else if (f != 0 && l == 0)
            best = &c->lpc_out;
        else if (f == 0)
            best = &c->fixed_out;
        else
            best = NULL;
    } else {
        int f = encode_fixed_subframe(o, use_rice2, c, &c->fixed_out, ch, n, bps, wasted);
        if (f == ORC_ERR_UNSUPPORTED) fatal = ORC_ERR_UNSUPPORTED;
        best = (f == 0) ? &c->fixed_out : NULL;
    }
    if (fatal) {
        *status = fatal;
        return &c->fixed_out;
    }
    uint32_t verbatim_len = n * bps; /* :2971 (header bits not counted) */
    if (best && best->w.total < verbatim_len) return best;
    bw_clear(&c->verb_out.w);
    if (encode_verbatim_subframe(&c->verb_out, ch, n, bps, wasted)) *status = ORC_ERR_IO;
    return &c->verb_out;
}

/* ------------------------------------------------------------------ */
/* frame header, stream.rs:242-276 + code tables                        */
/* ------------------------------------------------------------------ */
static void write_frame_header(bitw *w, uint32_t block_size, uint32_t sample_rate, uint32_t bps,
                               uint32_t assignment_code, uint64_t frame_number, int subset) {
    size_t start = w->len;
    bw_put(w, 15, 0x7FFC); /* SYNC_CODE 0b111111111111100 */
    bw_put(w, 1, 0);       /* blocking_strategy false */
    /* BlockSize::try_from(u16) stream.rs:531-558 */
    uint32_t bcode;
    int bextra = 0;
    switch (block_size) {
    case 192: bcode = 1; break;
    case 576: bcode = 2; break;
    case 1152: bcode = 3; break;
    case 2304: bcode = 4; break;
    case 4608: bcode = 5; break;
    case 256: bcode = 8; break;
    case 512: bcode = 9; break;
    case 1024: bcode = 10; break;
    case 2048: bcode = 11; break;
    case 4096: bcode = 12; break;
    case 8192: bcode = 13; break;
    case 16384: bcode = 14; break;
    case 32768: bcode = 15; break;
    default:
        if (block_size <= 256) {
            bcode = 6;
            bextra = 8;
        } else {
            bcode = 7;
            bextra = 16;
        }
    }
    bw_put(w, 4, bcode);
    /* SampleRate::try_from(u32) stream.rs:782-800 (arm order matters) */
    uint32_t rcode;
    int rextra = 0;
    uint32_t rval = 0;
    switch (sample_rate) {
    case 88200: rcode = 1; break;
    case 176400: rcode = 2; break;
    case 192000: rcode = 3; break;
    case 8000: rcode = 4; break;
    case 16000: rcode = 5; break;
    case 22050: rcode = 6; break;
    case 24000: rcode = 7; break;
    case 32000: rcode = 8; break;
    case 44100: rcode = 9; break;
    case 48000: rcode = 10; break;
    case 96000: rcode = 11; break;
    default:
        if (sample_rate % 1000 == 0 && sample_rate / 1000 < 255) {
            rcode = 12;
            rextra = 8;
            rval = sample_rate / 1000;
        } else if (sample_rate % 10 == 0 && sample_rate / 10 < 65535) {
            rcode = 14;
            rextra = 16;
            rval = sample_rate / 10;
        } else if (sample_rate < 65535) {
            rcode = 13;
            rextra = 16;
            rval = sample_rate;
        } else
            rcode = 0; /* Streaminfo */
    }
    (void)subset; /* subset streams reject rcode 0 / bps code 0 before getting here */
    bw_put(w, 4, rcode);
    bw_put(w, 4, assignment_code);
    uint32_t pcode;
    switch (bps) { /* stream.rs:1086-1098, 1185-1197 */
    case 8: pcode = 1; break;
    case 12: pcode = 2; break;
    case 16: pcode = 4; break;
    case 20: pcode = 5; break;
    case 24: pcode = 6; break;
    case 32: pcode = 7; break;
    default: pcode = 0;
    }
    bw_put(w, 3, pcode);
    bw_put(w, 1, 0); /* pad(1) */
    /* FrameNumber::to_writer stream.rs:1264-1325 */
    uint64_t v = frame_number;
#define FN_BYTE(b) (0x80u | (uint32_t)((v >> (6 * (b))) & 0x3F))
    if (v <= 0x7F) {
        bw_unary0(w, 0);
        bw_put(w, 7, (uint32_t)v);
    } else if (v <= 0x7FF) {
        bw_unary0(w, 2);
        bw_put(w, 5, (uint32_t)(v >> 6));
        bw_put(w, 8, FN_BYTE(0));
    } else if (v <= 0xFFFF) {
        bw_unary0(w, 3);
        bw_put(w, 4, (uint32_t)(v >> 12));
        bw_put(w, 8, FN_BYTE(1));
        bw_put(w, 8, FN_BYTE(0));
    } else if (v <= 0x1FFFFF) {
        bw_unary0(w, 4);
        bw_put(w, 3, (uint32_t)(v >> 18));
        for (int b = 2; b >= 0; b--) bw_put(w, 8, FN_BYTE(b));
    } else if (v <= 0x3FFFFFF) {
        bw_unary0(w, 5);
        bw_put(w, 2, (uint32_t)(v >> 24));
        for (int b = 3; b >= 0; b--) bw_put(w, 8, FN_BYTE(b));
    } else if (v <= 0x7FFFFFFFull) {
        bw_unary0(w, 6);
        bw_put(w, 1, (uint32_t)(v >> 30));
        for (int b = 4; b >= 0; b--) bw_put(w, 8, FN_BYTE(b));
    } else {
        bw_unary0(w, 7);
        for (int b = 5; b >= 0; b--) bw_put(w, 8, FN_BYTE(b));
    }
#undef FN_BYTE
    if (bextra) bw_put(w, (unsigned)bextra, block_size - 1);
    if (rextra) bw_put(w, (unsigned)rextra, rval);
    /* CRC-8 over the header bytes (stream.rs:194-197) */
    uint8_t c = orc_crc8(w->buf + start, w->len - start);
    bw_put(w, 8, c);
}

/* ------------------------------------------------------------------ */
/* encode_frame, encode.rs:2259-2439 (+ correlate_channels :2463-2674, */
/* correlate_channels_exhaustive :2676-2847)                           */
/* ------------------------------------------------------------------ */
static inline uint64_t abs_u64(int32_t v) {
    return (uint64_t)(v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v);
}

typedef struct {
    const orc_options *o;
    int use_rice2;
    chan_cache *c;
    const int32_t *ch;
    uint32_t n, bps;
    int all0;
    recorder *out;
    int st;
} chan_job;
static void run_chan_job(void *a) {
    chan_job *j = (chan_job *)a;
    j->out = encode_subframe(j->o, j->use_rice2, j->c, j->ch, j->n, j->bps, j->all0, &j->st);
}

static int encode_frame_inner(const orc_options *o, enc_caches *cache, uint32_t sample_rate,
                              uint32_t bps, uint32_t n_channels, const int32_t *const *chs,
                              uint32_t n, uint64_t frame_number, int subset, int use_rice2,
                              bitw *out, orc_frame_plan *plan) {
    recorder *subs[ORC_MAX_CHANNELS];
    uint8_t source[ORC_MAX_CHANNELS];
    uint32_t assignment = ORC_ASSIGN_INDEPENDENT;
    int st = 0;
    if (n == 0 || n > 65535) return ORC_ERR_OPTIONS;

    if (n_channels == 1) {
        int all0 = 1;
        for (uint32_t i = 0; i < n; i++)
            if (chs[0][i]) {
                all0 = 0;
                break;
            }
        subs[0] = encode_subframe(o, use_rice2, &cache->channels[0], chs[0], n, bps, all0, &st);
        if (st) return st;
        source[0] = 0;
    } else if (n_channels == 2) {
        const int32_t *left = chs[0], *right = chs[1];
        if (cache->cap < n) {
            cache->average = (int32_t *)realloc(cache->average, 4u * n);
            cache->difference = (int32_t *)realloc(cache->difference, 4u * n);
            cache->cap = n;
        }
        if (o->exhaustive) { /* :2676-2847 */
            /* try_join(left, right), encode.rs:2690-2712 */
            chan_job jl = {o, use_rice2, &cache->left, left, n, bps, 0, NULL, 0};
            chan_job jr = {o, use_rice2, &cache->right, right, n, bps, 0, NULL, 0};
            fj_task tlr[2] = {{run_chan_job, &jl, 0}, {run_chan_job, &jr, 0}};
            fj_run(tlr, 2);
            if (jl.st) return jl.st;
            if (jr.st) return jr.st;
            recorder *lr = jl.out, *rr = jr.out;
            if (bps + 1 <= 32 && o->mid_side) {
                for (uint32_t i = 0; i < n; i++) {
                    cache->average[i] = (int32_t)((uint32_t)left[i] + (uint32_t)right[i]) >> 1;
                    cache->difference[i] = (int32_t)((uint32_t)left[i] - (uint32_t)right[i]);
                }
                /* try_join(average, difference), encode.rs:2715-2745 */
                chan_job ja = {o, use_rice2, &cache->avg, cache->average, n, bps, 0, NULL, 0};
                chan_job jd = {o, use_rice2, &cache->diff, cache->difference, n, bps + 1, 0, NULL, 0};
                fj_task tad[2] = {{run_chan_job, &ja, 0}, {run_chan_job, &jd, 0}};
                fj_run(tad, 2);
                if (ja.st) return ja.st;
                if (jd.st) return jd.st;
                recorder *ar = ja.out, *dr = jd.out;
                uint32_t tot[4] = {lr->w.total + rr->w.total, lr->w.total + dr->w.total,
                                   dr->w.total + rr->w.total, ar->w.total + dr->w.total};
                int b = 0;
                for (int i = 1; i < 4; i++)
                    if (tot[i] < tot[b]) b = i;
                switch (b) {
                case 0: subs[0] = lr; subs[1] = rr; source[0] = 0; source[1] = 1; break;
                case 1: assignment = ORC_ASSIGN_LEFT_SIDE; subs[0] = lr; subs[1] = dr;
                        source[0] = 0; source[1] = 9; break;
                case 2: assignment = ORC_ASSIGN_SIDE_RIGHT; subs[0] = dr; subs[1] = rr;
                        source[0] = 9; source[1] = 1; break;
                default: assignment = ORC_ASSIGN_MID_SIDE; subs[0] = ar; subs[1] = dr;
                        source[0] = 8; source[1] = 9; break;
                }
            } else if (bps + 1 <= 32) {
                for (uint32_t i = 0; i < n; i++)
                    cache->difference[i] = (int32_t)((uint32_t)left[i] - (uint32_t)right[i]);
                recorder *dr = encode_subframe(o, use_rice2, &cache->diff, cache->difference, n,
                                               bps + 1, 0, &st);
                if (st) return st;
                uint32_t tot[3] = {lr->w.total + rr->w.total, lr->w.total + dr->w.total,
                                   dr->w.total + rr->w.total};
                int b = 0;
                for (int i = 1; i < 3; i++)
                    if (tot[i] < tot[b]) b = i;
                switch (b) {
                case 0: subs[0] = lr; subs[1] = rr; source[0] = 0; source[1] = 1; break;
                case 1: assignment = ORC_ASSIGN_LEFT_SIDE; subs[0] = lr; subs[1] = dr;
                        source[0] = 0; source[1] = 9; break;
                default: assignment = ORC_ASSIGN_SIDE_RIGHT; subs[0] = dr; subs[1] = rr;
                        source[0] = 9; source[1] = 1; break;
                }
            } else {
                subs[0] = lr; subs[1] = rr; source[0] = 0; source[1] = 1;
            }
        } else { /* correlate_channels :2463-2674 */
            const int32_t *c0 = left, *c1 = right;
            uint32_t b0 = bps, b1 = bps;
            int a0, a1;
            source[0] = 0; source[1] = 1;
            if (bps + 1 <= 32 && o->mid_side) {
                uint64_t ls = 0, rs = 0, ms = 0, ss = 0;
                for (uint32_t i = 0; i < n; i++) {
                    ls += abs_u64(left[i]);
                    rs += abs_u64(right[i]);
                    int32_t m = (int32_t)((uint32_t)left[i] + (uint32_t)right[i]) >> 1;
                    int32_t s = (int32_t)((uint32_t)left[i] - (uint32_t)right[i]);
                    cache->average[i] = m;
                    cache->difference[i] = s;
                    ms += abs_u64(m);
                    ss += abs_u64(s);
                }
                uint64_t tot[4] = {ls + rs, ls + ss, ss + rs, ms + ss};
                int b = 0;
                for (int i = 1; i < 4; i++)
                    if (tot[i] < tot[b]) b = i;
                a0 = (ls == 0); a1 = (rs == 0);
                if (b == 1) { assignment = ORC_ASSIGN_LEFT_SIDE; c1 = cache->difference; b1 = bps + 1;
                              a1 = (ss == 0); source[1] = 9; }
                else if (b == 2) { assignment = ORC_ASSIGN_SIDE_RIGHT; c0 = cache->difference;
                              b0 = bps + 1; a0 = (ss == 0); source[0] = 9; }
                else if (b == 3) { assignment = ORC_ASSIGN_MID_SIDE; c0 = cache->average;
                              c1 = cache->difference; b1 = bps + 1; a0 = (ms == 0); a1 = (ss == 0);
                              source[0] = 8; source[1] = 9; }
            } else if (bps + 1 <= 32) {
                uint64_t ls = 0, rs = 0, ss = 0;
                for (uint32_t i = 0; i < n; i++) {
                    ls += abs_u64(left[i]);
                    rs += abs_u64(right[i]);
                    int32_t s = (int32_t)((uint32_t)left[i] - (uint32_t)right[i]);
                    cache->difference[i] = s;
                    ss += abs_u64(s);
                }
                /* candidate order :2600-2607: LeftSide, SideRight, Independent */
                uint64_t tot[3] = {ls + ss, ss + rs, ls + rs};
                int b = 0;
                for (int i = 1; i < 3; i++)
                    if (tot[i] < tot[b]) b = i;
                a0 = (ls == 0); a1 = (rs == 0);
                if (b == 0) { assignment = ORC_ASSIGN_LEFT_SIDE; c1 = cache->difference; b1 = bps + 1;
                              a1 = (ss == 0); source[1] = 9; }
                else if (b == 1) { assignment = ORC_ASSIGN_SIDE_RIGHT; c0 = cache->difference;
                              b0 = bps + 1; a0 = (ss == 0); source[0] = 9; }
            } else {
                a0 = 1; a1 = 1;
                for (uint32_t i = 0; i < n; i++) { if (left[i]) a0 = 0; if (right[i]) a1 = 0; }
            }
            subs[0] = encode_subframe(o, use_rice2, &cache->channels[0], c0, n, b0, a0, &st);
            if (st) return st;
            subs[1] = encode_subframe(o, use_rice2, &cache->channels[1], c1, n, b1, a1, &st);
            if (st) return st;
        }
    } else { /* vec_map over the channels, encode.rs:2393-2402 */
        chan_job jc[ORC_MAX_CHANNELS];
        fj_task tc[ORC_MAX_CHANNELS];
        for (uint32_t c = 0; c < n_channels; c++) {
            int all0 = 1;
            for (uint32_t i = 0; i < n; i++)
                if (chs[c][i]) { all0 = 0; break; }
            chan_job j = {o, use_rice2, &cache->channels[c], chs[c], n, bps, all0, NULL, 0};
            jc[c] = j;
            tc[c].fn = run_chan_job;
            tc[c].arg = &jc[c];
            tc[c].done = 0;
        }
        fj_run(tc, (int)n_channels);
        for (uint32_t c = 0; c < n_channels; c++) {
            if (jc[c].st) return jc[c].st;
            subs[c] = jc[c].out;
            source[c] = (uint8_t)c;
        }
    }

    uint32_t code = (assignment == ORC_ASSIGN_INDEPENDENT) ? (n_channels - 1) : assignment;
    size_t start = out->len;
    write_frame_header(out, n, sample_rate, bps, code, frame_number, subset);
    for (uint32_t c = 0; c < n_channels; c++) bw_playback(&subs[c]->w, out);
    bw_align(out);
    uint16_t crc = orc_crc16(out->buf + start, out->len - start);
    bw_put(out, 16, crc);
    if (plan) {
        memset(plan, 0, sizeof *plan);
        plan->assignment = (uint8_t)assignment;
        plan->channels = (uint8_t)n_channels;
        plan->block_size = (uint16_t)n;
        plan->frame_bytes = (uint32_t)(out->len - start);
        for (uint32_t c = 0; c < n_channels; c++) {
            plan->source[c] = source[c];
            plan->sub[c] = subs[c]->plan;
        }
    }
    return 0;
}

static int validate_options(const orc_options *o) {
    if (o->block_size < 16 || o->block_size > 65535) return ORC_ERR_OPTIONS; /* :1418 */
    if (o->max_lpc_order < 0 || o->max_lpc_order > 32) return ORC_ERR_OPTIONS; /* :1430 */
    if (o->max_partition_order > 15) return ORC_ERR_OPTIONS;                   /* :1447 */
    return 0;
}

int orc_encode_frame(const orc_options *opts, uint32_t sample_rate, uint32_t bps,
                     uint32_t n_channels, const int32_t *const *channels, uint32_t n,
                     uint64_t frame_number, int subset_header, uint8_t **out, size_t *out_len,
                     size_t *out_cap, orc_frame_plan *plan) {
    if (bps < 1 || bps > 32) return ORC_ERR_INVALID_BPS;
    if (n_channels < 1 || n_channels > 8) return ORC_ERR_EXCESSIVE_CHANNELS;
    enc_caches *cache = (enc_caches *)calloc(1, sizeof *cache);
    bitw w;
    bw_init(&w);
    w.buf = *out;
    w.len = *out_len;
    w.cap = *out_cap;
    int rc = encode_frame_inner(opts, cache, sample_rate, bps, n_channels, channels, n,
                                frame_number, subset_header, bps > 16, &w, plan);
    *out = w.buf;
    *out_len = w.len;
    *out_cap = w.cap;
    caches_free(cache);
    free(cache);
    return rc;
}

/* residual recomputation from a decision record (tests only) */
int orc_subframe_residuals(const orc_subframe_plan *sp, const int32_t *cand, uint32_t n,
                           int32_t *residuals) {
    int32_t *tmp = (int32_t *)malloc(4u * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) tmp[i] = cand[i] >> sp->wasted;
    int count = 0;
    if (sp->type == ORC_SUB_LPC) {
        if (orc_encode_residuals(sp->order, sp->coeffs, sp->shift, tmp, n, residuals) == 0)
            count = (int)(n - sp->order);
    } else if (sp->type == ORC_SUB_FIXED) {
        static const int64_t binom[5][5] = {
            {1, 0, 0, 0, 0}, {1, -1, 0, 0, 0}, {1, -2, 1, 0, 0}, {1, -3, 3, -1, 0}, {1, -4, 6, -4, 1}};
        for (uint32_t i = sp->order; i < n; i++) {
            int64_t v = 0;
            for (int j = 0; j <= sp->order; j++) v += binom[sp->order][j] * (int64_t)tmp[i - j];
            residuals[i - sp->order] = (int32_t)v;
        }
        count = (int)(n - sp->order);
    }
    free(tmp);
    return count;
}

/* ------------------------------------------------------------------ */
/* stream level: Encoder::new / encode / finalize_inner                 */
/* (encode.rs:1882-2110) + FlacSampleWriter (encode.rs:487-627)         */
/* metadata bytes: metadata/mod.rs:257-266, 904-960, 1740-1760,         */
/* 1826-1831, 2010-2036, 2118-2139                                      */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t sample_offset, byte_offset;
    uint16_t frame_samples;
    int defined;
} seekpoint;

typedef struct {
    uint32_t min_block, max_block, min_frame, max_frame, sample_rate, channels, bps;
    uint64_t total_samples;
    uint8_t md5[16];
    int has_md5;
} streaminfo;

static void write_block_header(bitw *w, int last, uint32_t type, uint32_t size) {
    bw_put(w, 1, (uint32_t)last);
    bw_put(w, 7, type);
    bw_put(w, 24, size);
}
static void write_streaminfo(bitw *w, const streaminfo *s) { /* metadata/mod.rs:1743-1759 */
    bw_put(w, 16, s->min_block);
    bw_put(w, 16, s->max_block);
    bw_put(w, 24, s->min_frame);
    bw_put(w, 24, s->max_frame);
    bw_put(w, 20, s->sample_rate);
    bw_put(w, 3, s->channels - 1);
    bw_put(w, 5, s->bps - 1);
    bw_put64(w, 36, s->total_samples);
    for (int i = 0; i < 16; i++) bw_put(w, 8, s->has_md5 ? s->md5[i] : 0);
}

/* SeekTableInterval::filter, encode.rs:1338-1358 */
static size_t filter_seekpoints(const orc_options *o, uint32_t sample_rate, const seekpoint *in,
                                size_t n_in, seekpoint *out) {
    size_t cnt = 0;
    if (o->seektable_mode == 1) {
        uint64_t nth = (uint64_t)(uint32_t)((o->seektable_value & 0xFF) * sample_rate);
        uint64_t offset = 0;
        for (size_t i = 0; i < n_in; i++) {
            uint64_t lo = in[i].sample_offset, hi = lo + in[i].frame_samples;
            if (offset >= lo && offset < hi) {
                offset += nth;
                out[cnt++] = in[i];
            }
        }
    } else if (o->seektable_mode == 2) {
        size_t step = o->seektable_value ? o->seektable_value : 1;
        for (size_t i = 0; i < n_in; i += step) out[cnt++] = in[i];
    }
    return cnt;
}

#define SEEKTABLE_MAX_POINTS ((1u << 24) / 18u)

typedef struct {
    orc_options o;
    streaminfo si;
    int has_seektable; /* placeholder block present */
    size_t n_placeholder;
    int has_padding;
    uint32_t padding;
    uint8_t *vc_body;  /* serialized VORBIS_COMMENT body, NULL = no block */
    size_t vc_len;
    bitw out;          /* whole file in memory */
    size_t frames_start;
    uint64_t frame_number, samples_written;
    seekpoint *seekpoints;
    size_t n_seek, cap_seek;
    md5_ctx md5;
    enc_caches caches;
    int use_rice2;
} encoder;

static void write_all_blocks(encoder *e, bitw *w, const seekpoint *pts, size_t npts,
                             int seektable_after_padding) {
    /* "fLaC" + STREAMINFO + [SEEKTABLE] + [PADDING]; order after sort_by
     * (encode.rs:1944-1951): SeekTable < Padding; a SEEKTABLE inserted at
     * finalize is pushed after PADDING (metadata/mod.rs:4425-4441) */
    bw_put(w, 32, 0x664C6143u);
    int has_st = (pts != NULL);
    int n_after = (has_st ? 1 : 0) + (e->has_padding ? 1 : 0) + (e->vc_body ? 1 : 0);
    write_block_header(w, n_after == 0, 0, 34);
    write_streaminfo(w, &e->si);
    if (e->vc_body) { /* VorbisComment sorts first (encode.rs:1945) */
        n_after--;
        write_block_header(w, n_after == 0, 4, (uint32_t)e->vc_len);
        for (size_t i = 0; i < e->vc_len; i++) bw_put(w, 8, e->vc_body[i]);
    }
    for (int pass = 0; pass < 2; pass++) {
        int do_st = has_st && ((pass == 0) != (seektable_after_padding != 0));
        int do_pad = e->has_padding && ((pass == 0) == (seektable_after_padding != 0));
        if (do_st) {
            n_after--;
            write_block_header(w, n_after == 0, 3, (uint32_t)(npts * 18));
            for (size_t i = 0; i < npts; i++) {
                if (pts[i].defined) {
                    bw_put64(w, 64, pts[i].sample_offset);
                    bw_put64(w, 64, pts[i].byte_offset);
                    bw_put(w, 16, pts[i].frame_samples);
                } else { /* metadata/mod.rs:2133-2137 */
                    bw_put64(w, 64, UINT64_MAX);
                    bw_put64(w, 64, 0);
                    bw_put(w, 16, 0);
                }
            }
        }
        if (do_pad) {
            n_after--;
            write_block_header(w, n_after == 0, 1, e->padding);
            bw_reserve(w, e->padding + 8);
            for (uint32_t i = 0; i < e->padding; i++) bw_put(w, 8, 0);
        }
    }
}

static void put_le32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
/* VorbisComment::to_writer, metadata/mod.rs:2512-2536 */
static uint8_t *serialize_vc(const orc_vorbis_comment *vc, size_t *len) {
    const char *vendor = vc->vendor ? vc->vendor : "flac-codec 1.3.2";
    size_t n = 4 + strlen(vendor) + 4;
    for (uint32_t i = 0; i < vc->n_fields; i++) n += 4 + strlen(vc->fields[i]);
    uint8_t *b = (uint8_t *)malloc(n), *p = b;
    put_le32(p, (uint32_t)strlen(vendor)); p += 4;
    memcpy(p, vendor, strlen(vendor)); p += strlen(vendor);
    put_le32(p, vc->n_fields); p += 4;
    for (uint32_t i = 0; i < vc->n_fields; i++) {
        size_t l = strlen(vc->fields[i]);
        put_le32(p, (uint32_t)l); p += 4;
        memcpy(p, vc->fields[i], l); p += l;
    }
    *len = n;
    return b;
}

static int encoder_new(encoder *e, const orc_options *o, const orc_vorbis_comment *vc,
                       uint32_t sample_rate, uint32_t bps,
                       uint32_t channels, uint64_t total_pcm_frames /* 0 = None */) {
    memset(e, 0, sizeof *e);
    e->o = *o;
    if (vc) e->vc_body = serialize_vc(vc, &e->vc_len);
    if (validate_options(o)) return ORC_ERR_OPTIONS;
    if (sample_rate >= 1048576) return ORC_ERR_INVALID_SAMPLE_RATE; /* :1899 */
    if (channels < 1 || channels > 8) return ORC_ERR_EXCESSIVE_CHANNELS; /* :1904 */
    if (total_pcm_frames >= 68719476736ull) return ORC_ERR_EXCESSIVE_TOTAL; /* :1912 */
    e->si.min_block = e->si.max_block = o->block_size;
    e->si.sample_rate = sample_rate;
    e->si.channels = channels;
    e->si.bps = bps;
    e->si.total_samples = total_pcm_frames;
    e->has_padding = o->padding > 0; /* padding(0) removes the block (:1493) */
    e->padding = o->padding > 0 ? (uint32_t)o->padding : 0;
    e->use_rice2 = bps > 16; /* :1965 */
    md5_init(&e->md5);
    bw_init(&e->out);
    seekpoint *ph = NULL;
    size_t nph = 0;
    if (total_pcm_frames && o->seektable_mode) { /* :1920-1939 */
        size_t nall = (size_t)((total_pcm_frames + o->block_size - 1) / o->block_size);
        seekpoint *all = (seekpoint *)malloc(sizeof(seekpoint) * (nall ? nall : 1));
        for (size_t i = 0; i < nall; i++) { /* placeholders :2131-2141 */
            uint64_t off = (uint64_t)i * o->block_size;
            uint64_t rem = total_pcm_frames - off;
            all[i].sample_offset = off;
            all[i].byte_offset = 0;
            all[i].defined = 0;
            all[i].frame_samples = (uint16_t)(rem > 65535 ? o->block_size
                                              : (rem < o->block_size ? rem : o->block_size));
        }
        ph = (seekpoint *)malloc(sizeof(seekpoint) * (nall ? nall : 1));
        nph = filter_seekpoints(o, sample_rate, all, nall, ph);
        if (nph > SEEKTABLE_MAX_POINTS) nph = SEEKTABLE_MAX_POINTS;
        free(all);
        e->has_seektable = 1;
        e->n_placeholder = nph;
    }
    write_all_blocks(e, &e->out, e->has_seektable ? ph : NULL, nph, 0);
    free(ph);
    e->frames_start = e->out.len;
    return 0;
}

/* Encoder::encode :1997-2022 */
static int encoder_encode(encoder *e, const int32_t *const *chs, uint32_t n) {
    if (e->n_seek == e->cap_seek) {
        e->cap_seek = e->cap_seek ? e->cap_seek * 2 : 1024;
        e->seekpoints = (seekpoint *)realloc(e->seekpoints, sizeof(seekpoint) * e->cap_seek);
    }
    seekpoint *sp = &e->seekpoints[e->n_seek++];
    sp->sample_offset = e->samples_written;
    sp->byte_offset = e->out.len - e->frames_start;
    sp->frame_samples = (uint16_t)n;
    sp->defined = 1;
    e->samples_written += n;
    if (e->si.total_samples && e->samples_written > e->si.total_samples)
        return ORC_ERR_EXCESSIVE_TOTAL;
    size_t start = e->out.len;
    int rc = encode_frame_inner(&e->o, &e->caches, e->si.sample_rate, e->si.bps, e->si.channels,
                                chs, n, e->frame_number, 0, e->use_rice2, &e->out, NULL);
    if (rc) return rc;
    e->frame_number++;
    uint32_t size = (uint32_t)(e->out.len - start);
    if (size != 0 && size < (1u << 24) - 1) { /* :2414-2436 */
        if (e->si.min_frame == 0 || size < e->si.min_frame) e->si.min_frame = size;
        if (e->si.max_frame == 0 || size > e->si.max_frame) e->si.max_frame = size;
    }
    return 0;
}

/* Encoder::finalize_inner :2024-2110 */
static int encoder_finalize(encoder *e) {
    seekpoint *pts = NULL;
    size_t npts = 0;
    int st_after_padding = 0, write_st = 0;
    if (e->o.seektable_mode) {
        seekpoint *f = (seekpoint *)malloc(sizeof(seekpoint) * (e->n_seek ? e->n_seek : 1));
        size_t nf = filter_seekpoints(&e->o, e->si.sample_rate, e->seekpoints, e->n_seek, f);
        if (e->has_seektable) { /* placeholder already in place :2035-2052 */
            npts = e->n_placeholder;
            pts = (seekpoint *)calloc(npts ? npts : 1, sizeof(seekpoint));
            for (size_t i = 0; i < npts; i++)
                if (i < nf) pts[i] = f[i]; /* else Placeholder (defined = 0) */
            write_st = 1;
        } else if (e->has_padding) { /* :2053-2073 */
            if (nf > SEEKTABLE_MAX_POINTS) { free(f); return ORC_ERR_UNSUPPORTED; }
            uint64_t st_size = (uint64_t)nf * 18 + 4;
            if (nf * 18 < (1u << 24) && e->padding >= st_size) {
                e->padding -= (uint32_t)st_size;
                /* a PADDING block shrunk to 0 stays in the list (size-0 block) */
                pts = (seekpoint *)malloc(sizeof(seekpoint) * (nf ? nf : 1));
                memcpy(pts, f, sizeof(seekpoint) * nf);
                npts = nf;
                write_st = 1;
                st_after_padding = 1;
            }
        }
        free(f);
    }
    int rc = 0;
    if (e->si.total_samples) { /* :2079-2097 */
        if (e->si.total_samples != e->samples_written) rc = ORC_ERR_SAMPLE_COUNT_MISMATCH;
    } else {
        if (e->samples_written >= 68719476736ull) rc = ORC_ERR_EXCESSIVE_TOTAL;
        else if (e->samples_written == 0) rc = ORC_ERR_NO_SAMPLES;
        else e->si.total_samples = e->samples_written;
    }
    if (rc == 0) {
        md5_final(&e->md5, e->si.md5);
        e->si.has_md5 = 1;
        bitw hdr;
        bw_init(&hdr);
        write_all_blocks(e, &hdr, write_st ? pts : NULL, npts, st_after_padding);
        if (hdr.len != e->frames_start) rc = ORC_ERR_IO; /* must rewrite in place */
        else memcpy(e->out.buf, hdr.buf, hdr.len);
        bw_free(&hdr);
    }
    free(pts);
    return rc;
}

static void encoder_free(encoder *e) {
    caches_free(&e->caches);
    free(e->seekpoints);
    free(e->vc_body);
}

/* update_md5, encode.rs:1292-1318 (+ byteorder.rs:60-72 for 24-bit) */
static void update_md5_samples(md5_ctx *m, const int32_t *s, size_t count, unsigned bytes) {
    uint8_t buf[4096];
    size_t fill = 0;
    for (size_t i = 0; i < count; i++) {
        uint32_t v = (uint32_t)s[i];
        for (unsigned b = 0; b < bytes; b++) buf[fill++] = (uint8_t)(v >> (8 * b));
        if (fill + 4 > sizeof buf) {
            md5_update(m, buf, fill);
            fill = 0;
        }
    }
    if (fill) md5_update(m, buf, fill);
}

/* frame-parallel helper (NOT reference behaviour; identical output) */
typedef struct {
    const orc_options *o;
    uint32_t sample_rate, bps, channels, block;
    const int32_t *interleaved;
    uint64_t n_blocks_total, last_len;
    int tid, nthreads;
    uint8_t **bufs;
    size_t *lens;
    int rc;
} par_job;

static void *par_worker(void *arg) {
    par_job *j = (par_job *)arg;
    enc_caches *cache = (enc_caches *)calloc(1, sizeof *cache);
    int32_t *planar = (int32_t *)malloc(4u * (size_t)j->block * j->channels);
    const int32_t *chs[ORC_MAX_CHANNELS];
    for (uint64_t f = (uint64_t)j->tid; f < j->n_blocks_total; f += (uint64_t)j->nthreads) {
        uint32_t n = (f == j->n_blocks_total - 1) ? (uint32_t)j->last_len : j->block;
        const int32_t *src = j->interleaved + f * (uint64_t)j->block * j->channels;
        for (uint32_t c = 0; c < j->channels; c++) {
            int32_t *dst = planar + (size_t)c * n;
            for (uint32_t i = 0; i < n; i++) dst[i] = src[(size_t)i * j->channels + c];
            chs[c] = dst;
        }
        bitw w;
        bw_init(&w);
        int rc = encode_frame_inner(j->o, cache, j->sample_rate, j->bps, j->channels, chs, n, f, 0,
                                    j->bps > 16, &w, NULL);
        if (rc) j->rc = rc;
        j->bufs[f] = w.buf;
        j->lens[f] = w.len;
    }
    free(planar);
    caches_free(cache);
    free(cache);
    return NULL;
}

int orc_encode_stream(const orc_options *opts, uint32_t sample_rate, uint32_t bps,
                      uint32_t channels, const int32_t *interleaved, uint64_t n_interleaved,
                      int total_known, int threads, uint8_t **out, size_t *out_len,
                      orc_stream_stats *stats) {
    return orc_encode_stream_vc(opts, NULL, sample_rate, bps, channels, interleaved, n_interleaved,
                                total_known, threads, out, out_len, stats);
}

int orc_encode_stream_vc(const orc_options *opts, const orc_vorbis_comment *vc,
                         uint32_t sample_rate, uint32_t bps, uint32_t channels,
                         const int32_t *interleaved, uint64_t n_interleaved, int total_known,
                         int threads, uint8_t **out, size_t *out_len, orc_stream_stats *stats) {
    if (bps < 1 || bps > 32) return ORC_ERR_INVALID_BPS; /* :495 */
    if (channels < 1 || channels > 8) return ORC_ERR_EXCESSIVE_CHANNELS;
    uint64_t total = 0;
    if (total_known) { /* :511-520 */
        if (n_interleaved % channels) return ORC_ERR_NOT_DIVISIBLE;
        total = n_interleaved / channels;
        if (total == 0) return ORC_ERR_INVALID_TOTAL;
    }
    encoder e;
    int rc = encoder_new(&e, opts, vc, sample_rate, bps, channels, total);
    if (rc) return rc;
    unsigned bytes = (bps + 7) / 8;
    uint32_t block = opts->block_size;
    size_t frame_samples = (size_t)block * channels;
    uint64_t whole = n_interleaved / frame_samples;
    uint64_t rem = n_interleaved - whole * frame_samples;
    rem -= rem % channels; /* truncate to whole PCM frames :593-595 */
    int32_t *planar = (int32_t *)malloc(4u * frame_samples);
    const int32_t *chs[ORC_MAX_CHANNELS];

    const int fork_join = threads < -1;
    if (fork_join) fj_start(-threads); /* reference-shaped: frames sequential, tasks forked per frame */
    if (threads > 1) {
        uint64_t nblocks = whole + ((n_interleaved - whole * frame_samples) ? 1 : 0);
        uint64_t last_len = (n_interleaved - whole * frame_samples) ? rem / channels : block;
        if (nblocks && last_len == 0) { rc = ORC_ERR_UNSUPPORTED; goto done; }
        uint8_t **bufs = (uint8_t **)calloc(nblocks ? nblocks : 1, sizeof *bufs);
        size_t *lens = (size_t *)calloc(nblocks ? nblocks : 1, sizeof *lens);
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        par_job *jobs = (par_job *)calloc((size_t)threads, sizeof *jobs);
        for (int t = 0; t < threads; t++) {
            par_job j = {opts, sample_rate, bps, channels, block, interleaved, nblocks, last_len,
                         t, threads, bufs, lens, 0};
            jobs[t] = j;
            pthread_create(&th[t], NULL, par_worker, &jobs[t]);
        }
        /* MD5 on this thread meanwhile (serial chain) */
        update_md5_samples(&e.md5, interleaved, (size_t)(whole * frame_samples + rem), bytes);
        for (int t = 0; t < threads; t++) {
            pthread_join(th[t], NULL);
            if (jobs[t].rc) rc = jobs[t].rc;
        }
        for (uint64_t f = 0; f < nblocks && rc == 0; f++) {
            uint32_t n = (f == nblocks - 1) ? (uint32_t)last_len : block;
            if (e.n_seek == e.cap_seek) {
                e.cap_seek = e.cap_seek ? e.cap_seek * 2 : 1024;
                e.seekpoints = (seekpoint *)realloc(e.seekpoints, sizeof(seekpoint) * e.cap_seek);
            }
            seekpoint *sp = &e.seekpoints[e.n_seek++];
            sp->sample_offset = e.samples_written;
            sp->byte_offset = e.out.len - e.frames_start;
            sp->frame_samples = (uint16_t)n;
            sp->defined = 1;
            e.samples_written += n;
            bw_reserve(&e.out, lens[f]);
            memcpy(e.out.buf + e.out.len, bufs[f], lens[f]);
            e.out.len += lens[f];
            uint32_t size = (uint32_t)lens[f];
            if (e.si.min_frame == 0 || size < e.si.min_frame) e.si.min_frame = size;
            if (e.si.max_frame == 0 || size > e.si.max_frame) e.si.max_frame = size;
            e.frame_number++;
        }
        for (uint64_t f = 0; f < nblocks; f++) free(bufs[f]);
        free(bufs); free(lens); free(th); free(jobs);
    } else {
        for (uint64_t f = 0; f < whole && rc == 0; f++) { /* FlacSampleWriter::write :558-585 */
            const int32_t *src = interleaved + f * frame_samples;
            update_md5_samples(&e.md5, src, frame_samples, bytes);
            for (uint32_t c = 0; c < channels; c++) { /* Frame::fill_from_samples audio.rs:190 */
                int32_t *dst = planar + (size_t)c * block;
                for (uint32_t i = 0; i < block; i++) dst[i] = src[(size_t)i * channels + c];
                chs[c] = dst;
            }
            rc = encoder_encode(&e, chs, block);
        }
        if (rc == 0 && n_interleaved - whole * frame_samples) { /* finalize_inner :588-611 */
            const int32_t *src = interleaved + whole * frame_samples;
            uint32_t n = (uint32_t)(rem / channels);
            if (n == 0) rc = ORC_ERR_UNSUPPORTED; /* reference panics on chunks_exact(0) */
            else {
                update_md5_samples(&e.md5, src, (size_t)rem, bytes);
                for (uint32_t c = 0; c < channels; c++) {
                    int32_t *dst = planar + (size_t)c * n;
                    for (uint32_t i = 0; i < n; i++) dst[i] = src[(size_t)i * channels + c];
                    chs[c] = dst;
                }
                rc = encoder_encode(&e, chs, n);
            }
        }
    }
    if (rc == 0) rc = encoder_finalize(&e);
done:
    if (fork_join) fj_stop();
    free(planar);
    if (stats) {
        stats->frames = e.frame_number;
        stats->samples_written = e.samples_written;
        stats->min_frame_size = e.si.min_frame;
        stats->max_frame_size = e.si.max_frame;
        memcpy(stats->md5, e.si.md5, 16);
        stats->first_frame_offset = e.frames_start;
    }
    if (rc == 0) {
        *out = e.out.buf;
        *out_len = e.out.len;
        e.out.buf = NULL;
    } else {
        *out = NULL;
        *out_len = 0;
        bw_free(&e.out);
    }
    encoder_free(&e);
    return rc;
}

/* ------------------------------------------------------------------ */
/* decoder restatement (decode.rs:1388-1856), used as the round-trip    */
/* verifier exactly like every encoder test in tests/format.rs          */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *p;
    size_t len;
    size_t pos; /* bit position */
    int err;
} bitr;
static inline uint32_t br_bit(bitr *r) {
    if ((r->pos >> 3) >= r->len) {
        r->err = 1;
        return 0;
    }
    uint32_t b = (r->p[r->pos >> 3] >> (7 - (r->pos & 7))) & 1u;
    r->pos++;
    return b;
}
static inline uint64_t br_bits(bitr *r, unsigned n) {
    uint64_t v = 0;
    while (n) {
        if ((r->pos >> 3) >= r->len) {
            r->err = 1;
            return 0;
        }
        unsigned avail = 8 - (unsigned)(r->pos & 7);
        unsigned take = n < avail ? n : avail;
        uint32_t byte = r->p[r->pos >> 3];
        uint32_t chunk = (byte >> (avail - take)) & ((1u << take) - 1u);
        v = (v << take) | chunk;
        r->pos += take;
        n -= take;
    }
    return v;
}
static inline int64_t br_signed(bitr *r, unsigned n) {
    uint64_t v = br_bits(r, n);
    if (n < 64 && (v >> (n - 1)) & 1) v |= ~(((uint64_t)1 << n) - 1);
    return (int64_t)v;
}
static inline uint32_t br_unary1(bitr *r) { /* count zeros until a 1 */
    uint32_t q = 0;
    while (!r->err && br_bit(r) == 0) q++;
    return q;
}

static int read_residuals(bitr *r, uint32_t order, uint32_t block, int64_t *res) {
    uint32_t method = (uint32_t)br_bits(r, 2);
    if (method > 1) return -1;
    unsigned hb = method ? 5 : 4;
    uint32_t esc = method ? 31 : 15;
    uint32_t po = (uint32_t)br_bits(r, 4);
    uint32_t nparts = 1u << po;
    uint32_t plen = block >> po;
    uint32_t idx = 0;
    for (uint32_t p = 0; p < nparts; p++) {
        uint32_t cnt;
        if (p == 0) {
            if (plen < order) return -1;
            cnt = plen - order;
        } else
            cnt = plen;
        uint32_t k = (uint32_t)br_bits(r, hb);
        if (k == esc) {
            uint32_t eb = (uint32_t)br_bits(r, 5);
            for (uint32_t i = 0; i < cnt; i++) res[idx++] = eb ? br_signed(r, eb) : 0;
        } else {
            for (uint32_t i = 0; i < cnt; i++) {
                uint32_t q = br_unary1(r);
                uint32_t u = (q << k) | (uint32_t)br_bits(r, k);
                res[idx++] = (u & 1) ? -(int64_t)(u >> 1) - 1 : (int64_t)(u >> 1);
            }
        }
        if (r->err) return -1;
    }
    return (idx == block - order) ? 0 : -1;
}

static int read_subframe(bitr *r, uint32_t bps, uint32_t n, int64_t *ch) {
    if (br_bit(r)) return -1;
    uint32_t type = (uint32_t)br_bits(r, 6);
    uint32_t wasted = 0;
    if (br_bit(r)) wasted = br_unary1(r) + 1;
    if (wasted >= bps) return -1;
    uint32_t eb = bps - wasted;
    static const int64_t fixed_coeffs[5][4] = {
        {0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
    if (type == 0) {
        int64_t v = br_signed(r, eb);
        for (uint32_t i = 0; i < n; i++) ch[i] = v;
    } else if (type == 1) {
        for (uint32_t i = 0; i < n; i++) ch[i] = br_signed(r, eb);
    } else if (type >= 8 && type <= 12) {
        uint32_t order = type - 8;
        if (order > n) return -1;
        for (uint32_t i = 0; i < order; i++) ch[i] = br_signed(r, eb);
        if (read_residuals(r, order, n, ch + order)) return -1;
        for (uint32_t i = order; i < n; i++) {
            int64_t s = 0;
            for (uint32_t j = 0; j < order; j++) s += fixed_coeffs[order][j] * ch[i - 1 - j];
            ch[i] += s;
        }
    } else if (type >= 32) {
        uint32_t order = type - 31;
        if (order > n) return -1;
        for (uint32_t i = 0; i < order; i++) ch[i] = br_signed(r, eb);
        uint32_t prec = (uint32_t)br_bits(r, 4) + 1;
        if (prec == 16) return -1;
        int32_t shift = (int32_t)br_signed(r, 5);
        if (shift < 0) return -1;
        int64_t coef[32];
        for (uint32_t i = 0; i < order; i++) coef[i] = br_signed(r, prec);
        if (read_residuals(r, order, n, ch + order)) return -1;
        for (uint32_t i = order; i < n; i++) { /* predict decode.rs:1738-1752 */
            int64_t s = 0;
            for (uint32_t j = 0; j < order; j++) s += ch[i - 1 - j] * coef[j];
            ch[i] += (s >> shift);
        }
    } else
        return -1;
    if (wasted)
        for (uint32_t i = 0; i < n; i++) ch[i] = (int64_t)((uint64_t)ch[i] << wasted);
    return r->err ? -1 : 0;
}

int orc_decode_stream(const uint8_t *data, size_t len, int32_t **out_interleaved,
                      uint64_t *out_count, orc_decoded_info *info) {
    memset(info, 0, sizeof *info);
    *out_interleaved = NULL;
    *out_count = 0;
    if (len < 42 || memcmp(data, "fLaC", 4)) return -1;
    size_t pos = 4;
    int last = 0, have_si = 0;
    while (!last) {
        if (pos + 4 > len) return -2;
        last = data[pos] >> 7;
        uint32_t type = data[pos] & 0x7F;
        uint32_t size = ((uint32_t)data[pos + 1] << 16) | ((uint32_t)data[pos + 2] << 8) | data[pos + 3];
        pos += 4;
        if (pos + size > len) return -2;
        if (type == 0) {
            if (size != 34) return -2;
            bitr r = {data + pos, size, 0, 0};
            info->min_block = (uint32_t)br_bits(&r, 16);
            info->max_block = (uint32_t)br_bits(&r, 16);
            info->min_frame = (uint32_t)br_bits(&r, 24);
            info->max_frame = (uint32_t)br_bits(&r, 24);
            info->sample_rate = (uint32_t)br_bits(&r, 20);
            info->channels = (uint32_t)br_bits(&r, 3) + 1;
            info->bps = (uint32_t)br_bits(&r, 5) + 1;
            info->total_samples = br_bits(&r, 36);
            memcpy(info->md5, data + pos + 18, 16);
            have_si = 1;
        } else if (type == 3) {
            info->n_seekpoints = size / 18;
        }
        pos += size;
    }
    if (!have_si) return -2;
    uint32_t C = info->channels;
    size_t cap = (size_t)(info->total_samples ? info->total_samples : 65536) * C;
    int32_t *pcm = (int32_t *)malloc(4u * (cap ? cap : 1));
    uint64_t count = 0;
    int64_t *chbuf = (int64_t *)malloc(sizeof(int64_t) * 65536u * C);
    md5_ctx md5;
    md5_init(&md5);
    unsigned bytes = (info->bps + 7) / 8;
    int rc = 0;
    while (pos < len) {
        size_t fstart = pos;
        bitr r = {data + pos, len - pos, 0, 0};
        if (br_bits(&r, 15) != 0x7FFC) { rc = -3; break; }
        br_bit(&r);
        uint32_t bcode = (uint32_t)br_bits(&r, 4);
        uint32_t rcode = (uint32_t)br_bits(&r, 4);
        uint32_t acode = (uint32_t)br_bits(&r, 4);
        uint32_t pcode = (uint32_t)br_bits(&r, 3);
        br_bit(&r);
        /* frame number (UTF-8 like) */
        uint32_t ones = 0;
        while (br_bit(&r) == 1 && !r.err) ones++;
        uint64_t fn;
        if (ones == 0) fn = br_bits(&r, 7);
        else if (ones == 1 || ones > 7) { rc = -3; break; }
        else {
            fn = br_bits(&r, 7 - ones);
            for (uint32_t i = 1; i < ones; i++) {
                if (br_bits(&r, 2) != 2) { rc = -3; break; }
                fn = (fn << 6) | br_bits(&r, 6);
            }
        }
        (void)fn;
        uint32_t bs;
        static const uint32_t bs_table[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0,
                                              256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
        if (bcode == 6) bs = (uint32_t)br_bits(&r, 8) + 1;
        else if (bcode == 7) bs = (uint32_t)br_bits(&r, 16) + 1;
        else bs = bs_table[bcode];
        if (bs == 0) { rc = -3; break; }
        if (rcode == 12) br_bits(&r, 8);
        else if (rcode == 13 || rcode == 14) br_bits(&r, 16);
        size_t hdr_bytes = r.pos >> 3;
        uint8_t crc8 = (uint8_t)br_bits(&r, 8);
        if (r.err || orc_crc8(data + fstart, hdr_bytes) != crc8) { rc = -4; break; }
        static const uint32_t bps_table[8] = {0, 8, 12, 0, 16, 20, 24, 32};
        uint32_t bps = pcode ? bps_table[pcode] : info->bps;
        if (bps != info->bps) { rc = -3; break; }
        uint32_t nch = (acode < 8) ? acode + 1 : 2;
        if (nch != C || acode > 10) { rc = -3; break; }
        for (uint32_t c = 0; c < nch && rc == 0; c++) {
            uint32_t sb = bps;
            if ((acode == 8 && c == 1) || (acode == 9 && c == 0) || (acode == 10 && c == 1)) sb++;
            if (read_subframe(&r, sb, bs, chbuf + (size_t)c * bs)) rc = -5;
        }
        if (rc) break;
        int64_t *c0 = chbuf, *c1 = chbuf + bs;
        if (acode == 8) for (uint32_t i = 0; i < bs; i++) c1[i] = c0[i] - c1[i];
        else if (acode == 9) for (uint32_t i = 0; i < bs; i++) c0[i] += c1[i];
        else if (acode == 10)
            for (uint32_t i = 0; i < bs; i++) { /* decode.rs:1598-1602 */
                int64_t side = c1[i];
                int64_t sum = c0[i] * 2 + ((side < 0 ? -side : side) % 2);
                c0[i] = (sum + side) >> 1;
                c1[i] = (sum - side) >> 1;
            }
        r.pos = (r.pos + 7) & ~(size_t)7;
        size_t body = r.pos >> 3;
        uint16_t crc16 = (uint16_t)br_bits(&r, 16);
        if (r.err || orc_crc16(data + fstart, body) != crc16) { rc = -6; break; }
        if ((count + (uint64_t)bs) * C > cap) {
            cap = (size_t)((count + bs) * C * 2);
            pcm = (int32_t *)realloc(pcm, 4u * cap);
        }
        int32_t *dst = pcm + count * C;
        for (uint32_t i = 0; i < bs; i++)
            for (uint32_t c = 0; c < C; c++) dst[(size_t)i * C + c] = (int32_t)chbuf[(size_t)c * bs + i];
        update_md5_samples(&md5, dst, (size_t)bs * C, bytes);
        count += bs;
        info->frames++;
        pos = fstart + (r.pos >> 3);
    }
    free(chbuf);
    if (rc) {
        free(pcm);
        return rc;
    }
    uint8_t digest[16], zero[16] = {0};
    md5_final(&md5, digest);
    if (memcmp(info->md5, zero, 16) == 0) info->md5_ok = -1;
    else info->md5_ok = memcmp(info->md5, digest, 16) == 0;
    *out_interleaved = pcm;
    *out_count = count * C;
    return 0;
}
