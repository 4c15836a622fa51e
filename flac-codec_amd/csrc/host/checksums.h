// checksums.h -- CRC-8 (poly 0x07), CRC-16 (poly 0x8005, MSB first, init 0) as used by FLAC
// frames (reference: src/crc.rs:99-188) and MD5 (RFC 1321; crate md5 0.8 in the reference,
// encode.rs:1874, 2100).  Host side of the product.
#pragma once
#include <cstddef>
#include <cstdint>

namespace flacenc {

uint8_t crc8(const uint8_t *p, size_t n);
uint16_t crc16(const uint8_t *p, size_t n);

class Md5 {
public:
    Md5();
    void update(const void *data, size_t len);
    void digest(uint8_t out[16]) const;  // does not disturb the running state (clone + finalize)

    // for the multi-buffer engine (md5_mb.cpp): one block on a bare state, the state words of this
    // object, and whole blocks hashed outside update() (only while no partial block is buffered)
    static void transform(uint32_t state[4], const uint8_t *p);
    // nblocks blocks of TWO independent chains, interleaved (a chain is latency-bound: the second one is almost free)
    static void transform2(uint32_t s0[4], const uint8_t *p0, uint32_t s1[4], const uint8_t *p1, size_t nblocks);
    void get_state(uint32_t s[4]) const { s[0] = a_; s[1] = b_; s[2] = c_; s[3] = d_; }
    void set_state(const uint32_t s[4]) { a_ = s[0]; b_ = s[1]; c_ = s[2]; d_ = s[3]; }
    void add_blocks(uint64_t nblocks) { len_ += 64 * nblocks; }
    size_t buffered() const { return static_cast<size_t>(len_ & 63); }

private:
    void block(const uint8_t *p);
    uint32_t a_, b_, c_, d_;
    uint64_t len_ = 0;
    uint8_t buf_[64];
};

}  // namespace flacenc
