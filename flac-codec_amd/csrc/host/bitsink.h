// bitsink.h -- MSB-first bit packer (what bitstream-io's BitWriter<_, BigEndian> does for the
// reference, encode.rs:27) writing into caller-provided memory.  Host side of the product.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace flacenc {

class BitSink {
public:
    BitSink(uint8_t *dst, size_t capacity) : base_(dst), cur_(dst), end_(dst + capacity) {}

    // append the low `n` (0..32) bits of v
    inline void put(uint32_t v, unsigned n) {
        if (n == 0) return;
        acc_ = (acc_ << n) | (static_cast<uint64_t>(v) & ((n == 32) ? 0xFFFFFFFFull : ((1ull << n) - 1)));
        nacc_ += n;
        if (nacc_ >= 32) {
            nacc_ -= 32;
            uint32_t w = static_cast<uint32_t>(acc_ >> nacc_);
            if (cur_ + 4 <= end_) {
                cur_[0] = static_cast<uint8_t>(w >> 24);
                cur_[1] = static_cast<uint8_t>(w >> 16);
                cur_[2] = static_cast<uint8_t>(w >> 8);
                cur_[3] = static_cast<uint8_t>(w);
            } else {
                overflow_ = true;
            }
            cur_ += 4;
        }
        total_ += n;
    }
    inline void put64(uint64_t v, unsigned n) {
        if (n > 32) {
            put(static_cast<uint32_t>(v >> 32), n - 32);
            put(static_cast<uint32_t>(v), 32);
        } else {
            put(static_cast<uint32_t>(v), n);
        }
    }
    // two's-complement signed value in `n` bits (write_signed_counted)
    inline void put_signed(int32_t v, unsigned n) { put(static_cast<uint32_t>(v), n); }
    // `q` zero bits followed by a one (write_unary::<1>)
    inline void put_unary_then(uint32_t q, uint32_t low, unsigned k) {
        // q zeros, stop bit 1, then the k low bits -- the common short case in one put
        if (q + 1 + k <= 32) {
            put((1u << k) | low, q + 1 + k);
            return;
        }
        while (q >= 32) {
            put(0, 32);
            q -= 32;
        }
        put(1, q + 1);
        put(low, k);
    }
    // `q` one bits followed by a zero (write_unary::<0>)
    inline void put_ones_then_zero(unsigned q) { put(((1u << q) - 1u) << 1, q + 1); }

    inline void align() {
        if (total_ & 7) put(0, 8 - static_cast<unsigned>(total_ & 7));
    }
    // flush pending whole bytes; only valid when byte aligned
    inline size_t finish() {
        unsigned n = nacc_;
        while (n >= 8) {
            n -= 8;
            if (cur_ < end_) *cur_ = static_cast<uint8_t>(acc_ >> n);
            else overflow_ = true;
            cur_++;
        }
        nacc_ = n;
        return static_cast<size_t>(cur_ - base_);
    }
    inline uint64_t bits() const { return total_; }
    inline bool overflowed() const { return overflow_; }
    inline uint8_t *base() const { return base_; }

private:
    uint8_t *base_, *cur_, *end_;
    uint64_t acc_ = 0;
    unsigned nacc_ = 0;
    uint64_t total_ = 0;
    bool overflow_ = false;
};

}  // namespace flacenc
