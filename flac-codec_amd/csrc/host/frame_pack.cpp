#include "frame_pack.h"

#include "bitsink.h"
#include "checksums.h"

namespace flacenc {

namespace {

// BlockSize::try_from(u16), stream.rs:531-558
inline void block_size_code(uint32_t n, uint32_t &code, unsigned &extra_bits) {
    extra_bits = 0;
    switch (n) {
    case 192: code = 0x1; return;
    case 576: code = 0x2; return;
    case 1152: code = 0x3; return;
    case 2304: code = 0x4; return;
    case 4608: code = 0x5; return;
    case 256: code = 0x8; return;
    case 512: code = 0x9; return;
    case 1024: code = 0xA; return;
    case 2048: code = 0xB; return;
    case 4096: code = 0xC; return;
    case 8192: code = 0xD; return;
    case 16384: code = 0xE; return;
    case 32768: code = 0xF; return;
    default: break;
    }
    if (n <= 256) {
        code = 0x6;
        extra_bits = 8;
    } else {
        code = 0x7;
        extra_bits = 16;
    }
}

// SampleRate::try_from(u32), stream.rs:767-800 (the order of the arms matters)
inline void sample_rate_code(uint32_t r, uint32_t &code, unsigned &extra_bits, uint32_t &extra) {
    extra_bits = 0;
    extra = 0;
    switch (r) {
    case 88200: code = 0x1; return;
    case 176400: code = 0x2; return;
    case 192000: code = 0x3; return;
    case 8000: code = 0x4; return;
    case 16000: code = 0x5; return;
    case 22050: code = 0x6; return;
    case 24000: code = 0x7; return;
    case 32000: code = 0x8; return;
    case 44100: code = 0x9; return;
    case 48000: code = 0xA; return;
    case 96000: code = 0xB; return;
    default: break;
    }
    if (r % 1000 == 0 && r / 1000 < 255) {
        code = 0xC; extra_bits = 8; extra = r / 1000;
    } else if (r % 10 == 0 && r / 10 < 65535) {
        code = 0xE; extra_bits = 16; extra = r / 10;
    } else if (r < 65535) {
        code = 0xD; extra_bits = 16; extra = r;
    } else {
        code = 0x0;  // Streaminfo
    }
}

inline uint32_t bps_code(uint32_t bps) {  // stream.rs:1086-1098
    switch (bps) {
    case 8: return 1;
    case 12: return 2;
    case 16: return 4;
    case 20: return 5;
    case 24: return 6;
    case 32: return 7;
    default: return 0;
    }
}

inline unsigned frame_number_bytes(uint64_t v) {  // stream.rs:1270-1321
    if (v <= 0x7F) return 1;
    if (v <= 0x7FF) return 2;
    if (v <= 0xFFFF) return 3;
    if (v <= 0x1FFFFF) return 4;
    if (v <= 0x3FFFFFF) return 5;
    if (v <= 0x7FFFFFFFull) return 6;
    return 7;
}

void write_header(BitSink &w, const FrameParams &fp, uint32_t n, uint32_t assignment) {
    uint32_t bcode, rcode, rextra;
    unsigned bbits, rbits;
    block_size_code(n, bcode, bbits);
    sample_rate_code(fp.sample_rate, rcode, rbits, rextra);
    w.put(0x7FFC, 15);  // sync code 0b111111111111100
    w.put(0, 1);        // fixed block size strategy (encode.rs:2285)
    w.put(bcode, 4);
    w.put(rcode, 4);
    w.put(assignment == FLACGPU_ASSIGN_INDEPENDENT ? fp.channels - 1 : assignment, 4);
    w.put(bps_code(fp.bits_per_sample), 3);
    w.put(0, 1);
    const uint64_t v = fp.frame_number;
    const unsigned nb = frame_number_bytes(v);
    if (nb == 1) {
        w.put(static_cast<uint32_t>(v), 8);  // 0xxxxxxx
    } else {
        w.put_ones_then_zero(nb);                                       // 1..10
        w.put(static_cast<uint32_t>(v >> (6 * (nb - 1))), 7 - nb);      // leading payload bits
        for (int b = static_cast<int>(nb) - 2; b >= 0; b--)
            w.put(0x80u | static_cast<uint32_t>((v >> (6 * b)) & 0x3F), 8);
    }
    if (bbits) w.put(n - 1, bbits);
    if (rbits) w.put(rextra, rbits);
}

inline uint32_t zigzag(int32_t s) { return (static_cast<uint32_t>(s) << 1) ^ static_cast<uint32_t>(s >> 31); }

void write_subframe(BitSink &w, const flacgpu_subframe_plan &sp, const int32_t *row, uint32_t n) {
    uint32_t type_code;
    switch (sp.type) {
    case FLACGPU_SUB_CONSTANT: type_code = 0; break;
    case FLACGPU_SUB_VERBATIM: type_code = 1; break;
    case FLACGPU_SUB_FIXED: type_code = 8u + sp.order; break;
    default: type_code = 31u + sp.order; break;
    }
    w.put(type_code, 7);  // pad bit 0 + 6-bit type
    if (sp.wasted) {
        w.put(1, 1);
        w.put_unary_then(sp.wasted - 1u, 0, 0);
    } else {
        w.put(0, 1);
    }
    const unsigned bps = sp.bps;
    if (sp.type == FLACGPU_SUB_CONSTANT) {
        w.put_signed(row[0], bps);
        return;
    }
    if (sp.type == FLACGPU_SUB_VERBATIM) {
        for (uint32_t i = 0; i < n; i++) w.put_signed(row[i], bps);
        return;
    }
    const uint32_t order = sp.order;
    for (uint32_t i = 0; i < order; i++) w.put_signed(row[i], bps);
    if (sp.type == FLACGPU_SUB_LPC) {
        w.put(sp.precision - 1u, 4);
        w.put(sp.shift, 5);
        for (uint32_t i = 0; i < order; i++) w.put_signed(sp.coeffs[i], sp.precision);
    }
    // residual block, encode.rs:3898-3907 + 3834-3863
    const unsigned hb = sp.coding_method ? 5 : 4;
    const uint32_t escape_code = sp.coding_method ? 31u : 15u;
    w.put(sp.coding_method, 2);
    w.put(sp.partition_order, 4);
    const uint32_t nres = n - order;
    const uint32_t np = sp.n_partitions;
    const int32_t *r = row + order;
    uint32_t first_len = np > 1 ? nres - (np - 1) * sp.part_len : nres;
    for (uint32_t q = 0; q < np; q++) {
        const uint32_t len = q == 0 ? first_len : sp.part_len;
        const uint8_t k8 = sp.rice[q];
        if (k8 != 0xFF) {
            const unsigned k = k8;
            w.put(k, hb);
            const uint32_t mask = k ? ((1u << k) - 1u) : 0u;
            for (uint32_t i = 0; i < len; i++) {
                const uint32_t u = zigzag(r[i]);
                w.put_unary_then(u >> k, u & mask, k);
            }
        } else {
            const unsigned eb = sp.escape_bits[q];
            w.put(escape_code, hb);
            w.put(eb, 5);
            if (eb)
                for (uint32_t i = 0; i < len; i++) w.put_signed(r[i], eb);
        }
        r += len;
    }
}

}  // namespace

size_t frame_header_size(const FrameParams &fp, uint32_t n) {
    uint32_t c, e;
    unsigned bb, rb;
    block_size_code(n, c, bb);
    sample_rate_code(fp.sample_rate, c, rb, e);
    return 4 + frame_number_bytes(fp.frame_number) + bb / 8 + rb / 8 + 1;
}

size_t frame_size(const FrameParams &fp, const flacgpu_frame_plan &plan) {
    return frame_header_size(fp, plan.block_size) + (static_cast<size_t>(plan.body_bits) + 7) / 8 + 2;
}

size_t pack_frame(const FrameParams &fp, const flacgpu_frame_plan &plan,
                  const flacgpu_subframe_plan *subs, const int32_t *rows, size_t row_stride,
                  uint8_t *dst, size_t cap) {
    const uint32_t n = plan.block_size;
    // BitSink stores 32 bits at a time: it may touch up to 3 bytes past the logical end
    BitSink w(dst, cap);
    write_header(w, fp, n, plan.assignment);
    const size_t hdr_bytes = static_cast<size_t>(w.bits() / 8);
    w.finish();
    const uint8_t c8 = crc8(dst, hdr_bytes);
    w.put(c8, 8);
    for (uint32_t c = 0; c < fp.channels; c++) {
        const uint64_t before = w.bits();
        write_subframe(w, subs[c], rows + c * row_stride, n);
        if (w.bits() - before != subs[c].bits) return 0;  // decision record and emission disagree
    }
    w.align();
    const size_t body = w.finish();
    if (w.overflowed() || body + 2 > cap) return 0;
    const uint16_t c16 = crc16(dst, body);
    dst[body] = static_cast<uint8_t>(c16 >> 8);
    dst[body + 1] = static_cast<uint8_t>(c16);
    return body + 2;
}

}  // namespace flacenc
