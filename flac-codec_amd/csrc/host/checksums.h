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

private:
    void block(const uint8_t *p);
    uint32_t a_, b_, c_, d_;
    uint64_t len_ = 0;
    uint8_t buf_[64];
};

}  // namespace flacenc
