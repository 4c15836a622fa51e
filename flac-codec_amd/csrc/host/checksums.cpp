#include "checksums.h"

#include <cstring>

namespace flacenc {

namespace {
struct Crc8Table {
    uint8_t t[256];
    constexpr Crc8Table() : t() {
        for (int i = 0; i < 256; i++) {
            uint8_t c = static_cast<uint8_t>(i);
            for (int b = 0; b < 8; b++) c = static_cast<uint8_t>((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
            t[i] = c;
        }
    }
};
// slicing-by-4 tables for the MSB-first CRC-16: T[k][b] = crc of byte b followed by k zero bytes
struct Crc16Tables {
    uint16_t t[4][256];
    constexpr Crc16Tables() : t() {
        for (int i = 0; i < 256; i++) {
            uint16_t c = static_cast<uint16_t>(i << 8);
            for (int b = 0; b < 8; b++) c = static_cast<uint16_t>((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
            t[0][i] = c;
        }
        for (int k = 1; k < 4; k++)
            for (int i = 0; i < 256; i++) {
                uint16_t c = t[k - 1][i];
                t[k][i] = static_cast<uint16_t>(t[0][c >> 8] ^ (c << 8));
            }
    }
};
constexpr Crc8Table kCrc8{};
constexpr Crc16Tables kCrc16{};
}  // namespace

uint8_t crc8(const uint8_t *p, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) c = kCrc8.t[c ^ p[i]];
    return c;
}

uint16_t crc16(const uint8_t *p, size_t n) {
    uint16_t c = 0;
    size_t i = 0;
    // 4 bytes per step: the 16-bit state only touches the first two of them
    for (; i + 4 <= n; i += 4) {
        uint8_t b0 = static_cast<uint8_t>(p[i] ^ (c >> 8));
        uint8_t b1 = static_cast<uint8_t>(p[i + 1] ^ (c & 0xFF));
        c = static_cast<uint16_t>(kCrc16.t[3][b0] ^ kCrc16.t[2][b1] ^ kCrc16.t[1][p[i + 2]] ^ kCrc16.t[0][p[i + 3]]);
    }
    for (; i < n; i++) c = static_cast<uint16_t>(kCrc16.t[0][(c >> 8) ^ p[i]] ^ (c << 8));
    return c;
}

// ---- MD5, RFC 1321 ----
namespace {
inline uint32_t rotl(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
inline uint32_t load_le(const uint8_t *p) {
    return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) |
           (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24);
}
}  // namespace

Md5::Md5() : a_(0x67452301u), b_(0xefcdab89u), c_(0x98badcfeu), d_(0x10325476u) {}

#define MD5_F(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define MD5_G(x, y, z) (((x) & (z)) + ((y) & ~(z)))
#define MD5_H(x, y, z) ((x) ^ (y) ^ (z))
#define MD5_I(x, y, z) ((y) ^ ((x) | ~(z)))
#define MD5_STEP(f, a, b, c, d, x, t, s) \
    (a) += f((b), (c), (d)) + (x) + (t); \
    (a) = rotl((a), (s)) + (b)

void Md5::block(const uint8_t *p) {
    uint32_t s[4] = {a_, b_, c_, d_};
    transform(s, p);
    a_ = s[0]; b_ = s[1]; c_ = s[2]; d_ = s[3];
}

void Md5::transform(uint32_t state[4], const uint8_t *p) {
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = load_le(p + 4 * i);
    uint32_t a = state[0], b = state[1], c = state[2], d = state[3];
    MD5_STEP(MD5_F, a, b, c, d, x[0], 0xd76aa478u, 7);   MD5_STEP(MD5_F, d, a, b, c, x[1], 0xe8c7b756u, 12);
    MD5_STEP(MD5_F, c, d, a, b, x[2], 0x242070dbu, 17);  MD5_STEP(MD5_F, b, c, d, a, x[3], 0xc1bdceeeu, 22);
    MD5_STEP(MD5_F, a, b, c, d, x[4], 0xf57c0fafu, 7);   MD5_STEP(MD5_F, d, a, b, c, x[5], 0x4787c62au, 12);
    MD5_STEP(MD5_F, c, d, a, b, x[6], 0xa8304613u, 17);  MD5_STEP(MD5_F, b, c, d, a, x[7], 0xfd469501u, 22);
    MD5_STEP(MD5_F, a, b, c, d, x[8], 0x698098d8u, 7);   MD5_STEP(MD5_F, d, a, b, c, x[9], 0x8b44f7afu, 12);
    MD5_STEP(MD5_F, c, d, a, b, x[10], 0xffff5bb1u, 17); MD5_STEP(MD5_F, b, c, d, a, x[11], 0x895cd7beu, 22);
    MD5_STEP(MD5_F, a, b, c, d, x[12], 0x6b901122u, 7);  MD5_STEP(MD5_F, d, a, b, c, x[13], 0xfd987193u, 12);
    MD5_STEP(MD5_F, c, d, a, b, x[14], 0xa679438eu, 17); MD5_STEP(MD5_F, b, c, d, a, x[15], 0x49b40821u, 22);
    MD5_STEP(MD5_G, a, b, c, d, x[1], 0xf61e2562u, 5);   MD5_STEP(MD5_G, d, a, b, c, x[6], 0xc040b340u, 9);
    MD5_STEP(MD5_G, c, d, a, b, x[11], 0x265e5a51u, 14); MD5_STEP(MD5_G, b, c, d, a, x[0], 0xe9b6c7aau, 20);
    MD5_STEP(MD5_G, a, b, c, d, x[5], 0xd62f105du, 5);   MD5_STEP(MD5_G, d, a, b, c, x[10], 0x02441453u, 9);
    MD5_STEP(MD5_G, c, d, a, b, x[15], 0xd8a1e681u, 14); MD5_STEP(MD5_G, b, c, d, a, x[4], 0xe7d3fbc8u, 20);
    MD5_STEP(MD5_G, a, b, c, d, x[9], 0x21e1cde6u, 5);   MD5_STEP(MD5_G, d, a, b, c, x[14], 0xc33707d6u, 9);
    MD5_STEP(MD5_G, c, d, a, b, x[3], 0xf4d50d87u, 14);  MD5_STEP(MD5_G, b, c, d, a, x[8], 0x455a14edu, 20);
    MD5_STEP(MD5_G, a, b, c, d, x[13], 0xa9e3e905u, 5);  MD5_STEP(MD5_G, d, a, b, c, x[2], 0xfcefa3f8u, 9);
    MD5_STEP(MD5_G, c, d, a, b, x[7], 0x676f02d9u, 14);  MD5_STEP(MD5_G, b, c, d, a, x[12], 0x8d2a4c8au, 20);
    MD5_STEP(MD5_H, a, b, c, d, x[5], 0xfffa3942u, 4);   MD5_STEP(MD5_H, d, a, b, c, x[8], 0x8771f681u, 11);
    MD5_STEP(MD5_H, c, d, a, b, x[11], 0x6d9d6122u, 16); MD5_STEP(MD5_H, b, c, d, a, x[14], 0xfde5380cu, 23);
    MD5_STEP(MD5_H, a, b, c, d, x[1], 0xa4beea44u, 4);   MD5_STEP(MD5_H, d, a, b, c, x[4], 0x4bdecfa9u, 11);
    MD5_STEP(MD5_H, c, d, a, b, x[7], 0xf6bb4b60u, 16);  MD5_STEP(MD5_H, b, c, d, a, x[10], 0xbebfbc70u, 23);
    MD5_STEP(MD5_H, a, b, c, d, x[13], 0x289b7ec6u, 4);  MD5_STEP(MD5_H, d, a, b, c, x[0], 0xeaa127fau, 11);
    MD5_STEP(MD5_H, c, d, a, b, x[3], 0xd4ef3085u, 16);  MD5_STEP(MD5_H, b, c, d, a, x[6], 0x04881d05u, 23);
    MD5_STEP(MD5_H, a, b, c, d, x[9], 0xd9d4d039u, 4);   MD5_STEP(MD5_H, d, a, b, c, x[12], 0xe6db99e5u, 11);
    MD5_STEP(MD5_H, c, d, a, b, x[15], 0x1fa27cf8u, 16); MD5_STEP(MD5_H, b, c, d, a, x[2], 0xc4ac5665u, 23);
    MD5_STEP(MD5_I, a, b, c, d, x[0], 0xf4292244u, 6);   MD5_STEP(MD5_I, d, a, b, c, x[7], 0x432aff97u, 10);
    MD5_STEP(MD5_I, c, d, a, b, x[14], 0xab9423a7u, 15); MD5_STEP(MD5_I, b, c, d, a, x[5], 0xfc93a039u, 21);
    MD5_STEP(MD5_I, a, b, c, d, x[12], 0x655b59c3u, 6);  MD5_STEP(MD5_I, d, a, b, c, x[3], 0x8f0ccc92u, 10);
    MD5_STEP(MD5_I, c, d, a, b, x[10], 0xffeff47du, 15); MD5_STEP(MD5_I, b, c, d, a, x[1], 0x85845dd1u, 21);
    MD5_STEP(MD5_I, a, b, c, d, x[8], 0x6fa87e4fu, 6);   MD5_STEP(MD5_I, d, a, b, c, x[15], 0xfe2ce6e0u, 10);
    MD5_STEP(MD5_I, c, d, a, b, x[6], 0xa3014314u, 15);  MD5_STEP(MD5_I, b, c, d, a, x[13], 0x4e0811a1u, 21);
    MD5_STEP(MD5_I, a, b, c, d, x[4], 0xf7537e82u, 6);   MD5_STEP(MD5_I, d, a, b, c, x[11], 0xbd3af235u, 10);
    MD5_STEP(MD5_I, c, d, a, b, x[2], 0x2ad7d2bbu, 15);  MD5_STEP(MD5_I, b, c, d, a, x[9], 0xeb86d391u, 21);
    state[0] += a; state[1] += b; state[2] += c; state[3] += d;
}

// Two independent chains advanced step by step in ONE instruction stream: a scalar chain is bound by the latency of its ~4.5
// dependent operations per step and leaves most of the core's integer units idle, so a second chain rides along at (nearly)
// no cost -- about 1 GB/s EACH on the bench host, where a lane of the 16-wide AVX-512 step makes 0.6 (its vector integer
// operations take two cycles).  For a handful of long streams (md5_mb.cpp's engines, two chains per thread).
void Md5::transform2(uint32_t s0[4], const uint8_t *p0, uint32_t s1[4], const uint8_t *p1, size_t nblocks) {
    auto ld32 = [](const uint8_t *q) {
        uint32_t v;
        std::memcpy(&v, q, 4);   // (little-endian hosts: x86-64; load_le elsewhere)
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_BIG_ENDIAN__
        v = __builtin_bswap32(v);
#endif
        return v;
    };
    uint32_t A0 = s0[0], B0 = s0[1], C0 = s0[2], D0 = s0[3], A1 = s1[0], B1 = s1[1], C1 = s1[2], D1 = s1[3];
    for (size_t blk = 0; blk < nblocks; blk++, p0 += 64, p1 += 64) {
        uint32_t a0 = A0, b0 = B0, c0 = C0, d0 = D0, a1 = A1, b1 = B1, c1 = C1, d1 = D1;
        MD5_STEP(MD5_F, a0, b0, c0, d0, ld32(p0 + 0), 0xd76aa478u, 7); MD5_STEP(MD5_F, a1, b1, c1, d1, ld32(p1 + 0), 0xd76aa478u, 7);
        MD5_STEP(MD5_F, d0, a0, b0, c0, ld32(p0 + 4), 0xe8c7b756u, 12); MD5_STEP(MD5_F, d1, a1, b1, c1, ld32(p1 + 4), 0xe8c7b756u, 12);
        MD5_STEP(MD5_F, c0, d0, a0, b0, ld32(p0 + 8), 0x242070dbu, 17); MD5_STEP(MD5_F, c1, d1, a1, b1, ld32(p1 + 8), 0x242070dbu, 17);
        MD5_STEP(MD5_F, b0, c0, d0, a0, ld32(p0 + 12), 0xc1bdceeeu, 22); MD5_STEP(MD5_F, b1, c1, d1, a1, ld32(p1 + 12), 0xc1bdceeeu, 22);
        MD5_STEP(MD5_F, a0, b0, c0, d0, ld32(p0 + 16), 0xf57c0fafu, 7); MD5_STEP(MD5_F, a1, b1, c1, d1, ld32(p1 + 16), 0xf57c0fafu, 7);
        MD5_STEP(MD5_F, d0, a0, b0, c0, ld32(p0 + 20), 0x4787c62au, 12); MD5_STEP(MD5_F, d1, a1, b1, c1, ld32(p1 + 20), 0x4787c62au, 12);
        MD5_STEP(MD5_F, c0, d0, a0, b0, ld32(p0 + 24), 0xa8304613u, 17); MD5_STEP(MD5_F, c1, d1, a1, b1, ld32(p1 + 24), 0xa8304613u, 17);
        MD5_STEP(MD5_F, b0, c0, d0, a0, ld32(p0 + 28), 0xfd469501u, 22); MD5_STEP(MD5_F, b1, c1, d1, a1, ld32(p1 + 28), 0xfd469501u, 22);
        MD5_STEP(MD5_F, a0, b0, c0, d0, ld32(p0 + 32), 0x698098d8u, 7); MD5_STEP(MD5_F, a1, b1, c1, d1, ld32(p1 + 32), 0x698098d8u, 7);
        MD5_STEP(MD5_F, d0, a0, b0, c0, ld32(p0 + 36), 0x8b44f7afu, 12); MD5_STEP(MD5_F, d1, a1, b1, c1, ld32(p1 + 36), 0x8b44f7afu, 12);
        MD5_STEP(MD5_F, c0, d0, a0, b0, ld32(p0 + 40), 0xffff5bb1u, 17); MD5_STEP(MD5_F, c1, d1, a1, b1, ld32(p1 + 40), 0xffff5bb1u, 17);
        MD5_STEP(MD5_F, b0, c0, d0, a0, ld32(p0 + 44), 0x895cd7beu, 22); MD5_STEP(MD5_F, b1, c1, d1, a1, ld32(p1 + 44), 0x895cd7beu, 22);
        MD5_STEP(MD5_F, a0, b0, c0, d0, ld32(p0 + 48), 0x6b901122u, 7); MD5_STEP(MD5_F, a1, b1, c1, d1, ld32(p1 + 48), 0x6b901122u, 7);
        MD5_STEP(MD5_F, d0, a0, b0, c0, ld32(p0 + 52), 0xfd987193u, 12); MD5_STEP(MD5_F, d1, a1, b1, c1, ld32(p1 + 52), 0xfd987193u, 12);
        MD5_STEP(MD5_F, c0, d0, a0, b0, ld32(p0 + 56), 0xa679438eu, 17); MD5_STEP(MD5_F, c1, d1, a1, b1, ld32(p1 + 56), 0xa679438eu, 17);
        MD5_STEP(MD5_F, b0, c0, d0, a0, ld32(p0 + 60), 0x49b40821u, 22); MD5_STEP(MD5_F, b1, c1, d1, a1, ld32(p1 + 60), 0x49b40821u, 22);
        MD5_STEP(MD5_G, a0, b0, c0, d0, ld32(p0 + 4), 0xf61e2562u, 5); MD5_STEP(MD5_G, a1, b1, c1, d1, ld32(p1 + 4), 0xf61e2562u, 5);
        MD5_STEP(MD5_G, d0, a0, b0, c0, ld32(p0 + 24), 0xc040b340u, 9); MD5_STEP(MD5_G, d1, a1, b1, c1, ld32(p1 + 24), 0xc040b340u, 9);
        MD5_STEP(MD5_G, c0, d0, a0, b0, ld32(p0 + 44), 0x265e5a51u, 14); MD5_STEP(MD5_G, c1, d1, a1, b1, ld32(p1 + 44), 0x265e5a51u, 14);
        MD5_STEP(MD5_G, b0, c0, d0, a0, ld32(p0 + 0), 0xe9b6c7aau, 20); MD5_STEP(MD5_G, b1, c1, d1, a1, ld32(p1 + 0), 0xe9b6c7aau, 20);
        MD5_STEP(MD5_G, a0, b0, c0, d0, ld32(p0 + 20), 0xd62f105du, 5); MD5_STEP(MD5_G, a1, b1, c1, d1, ld32(p1 + 20), 0xd62f105du, 5);
        MD5_STEP(MD5_G, d0, a0, b0, c0, ld32(p0 + 40), 0x02441453u, 9); MD5_STEP(MD5_G, d1, a1, b1, c1, ld32(p1 + 40), 0x02441453u, 9);
        MD5_STEP(MD5_G, c0, d0, a0, b0, ld32(p0 + 60), 0xd8a1e681u, 14); MD5_STEP(MD5_G, c1, d1, a1, b1, ld32(p1 + 60), 0xd8a1e681u, 14);
        MD5_STEP(MD5_G, b0, c0, d0, a0, ld32(p0 + 16), 0xe7d3fbc8u, 20); MD5_STEP(MD5_G, b1, c1, d1, a1, ld32(p1 + 16), 0xe7d3fbc8u, 20);
        MD5_STEP(MD5_G, a0, b0, c0, d0, ld32(p0 + 36), 0x21e1cde6u, 5); MD5_STEP(MD5_G, a1, b1, c1, d1, ld32(p1 + 36), 0x21e1cde6u, 5);
        MD5_STEP(MD5_G, d0, a0, b0, c0, ld32(p0 + 56), 0xc33707d6u, 9); MD5_STEP(MD5_G, d1, a1, b1, c1, ld32(p1 + 56), 0xc33707d6u, 9);
        MD5_STEP(MD5_G, c0, d0, a0, b0, ld32(p0 + 12), 0xf4d50d87u, 14); MD5_STEP(MD5_G, c1, d1, a1, b1, ld32(p1 + 12), 0xf4d50d87u, 14);
        MD5_STEP(MD5_G, b0, c0, d0, a0, ld32(p0 + 32), 0x455a14edu, 20); MD5_STEP(MD5_G, b1, c1, d1, a1, ld32(p1 + 32), 0x455a14edu, 20);
        MD5_STEP(MD5_G, a0, b0, c0, d0, ld32(p0 + 52), 0xa9e3e905u, 5); MD5_STEP(MD5_G, a1, b1, c1, d1, ld32(p1 + 52), 0xa9e3e905u, 5);
        MD5_STEP(MD5_G, d0, a0, b0, c0, ld32(p0 + 8), 0xfcefa3f8u, 9); MD5_STEP(MD5_G, d1, a1, b1, c1, ld32(p1 + 8), 0xfcefa3f8u, 9);
        MD5_STEP(MD5_G, c0, d0, a0, b0, ld32(p0 + 28), 0x676f02d9u, 14); MD5_STEP(MD5_G, c1, d1, a1, b1, ld32(p1 + 28), 0x676f02d9u, 14);
        MD5_STEP(MD5_G, b0, c0, d0, a0, ld32(p0 + 48), 0x8d2a4c8au, 20); MD5_STEP(MD5_G, b1, c1, d1, a1, ld32(p1 + 48), 0x8d2a4c8au, 20);
        MD5_STEP(MD5_H, a0, b0, c0, d0, ld32(p0 + 20), 0xfffa3942u, 4); MD5_STEP(MD5_H, a1, b1, c1, d1, ld32(p1 + 20), 0xfffa3942u, 4);
        MD5_STEP(MD5_H, d0, a0, b0, c0, ld32(p0 + 32), 0x8771f681u, 11); MD5_STEP(MD5_H, d1, a1, b1, c1, ld32(p1 + 32), 0x8771f681u, 11);
        MD5_STEP(MD5_H, c0, d0, a0, b0, ld32(p0 + 44), 0x6d9d6122u, 16); MD5_STEP(MD5_H, c1, d1, a1, b1, ld32(p1 + 44), 0x6d9d6122u, 16);
        MD5_STEP(MD5_H, b0, c0, d0, a0, ld32(p0 + 56), 0xfde5380cu, 23); MD5_STEP(MD5_H, b1, c1, d1, a1, ld32(p1 + 56), 0xfde5380cu, 23);
        MD5_STEP(MD5_H, a0, b0, c0, d0, ld32(p0 + 4), 0xa4beea44u, 4); MD5_STEP(MD5_H, a1, b1, c1, d1, ld32(p1 + 4), 0xa4beea44u, 4);
        MD5_STEP(MD5_H, d0, a0, b0, c0, ld32(p0 + 16), 0x4bdecfa9u, 11); MD5_STEP(MD5_H, d1, a1, b1, c1, ld32(p1 + 16), 0x4bdecfa9u, 11);
        MD5_STEP(MD5_H, c0, d0, a0, b0, ld32(p0 + 28), 0xf6bb4b60u, 16); MD5_STEP(MD5_H, c1, d1, a1, b1, ld32(p1 + 28), 0xf6bb4b60u, 16);
        MD5_STEP(MD5_H, b0, c0, d0, a0, ld32(p0 + 40), 0xbebfbc70u, 23); MD5_STEP(MD5_H, b1, c1, d1, a1, ld32(p1 + 40), 0xbebfbc70u, 23);
        MD5_STEP(MD5_H, a0, b0, c0, d0, ld32(p0 + 52), 0x289b7ec6u, 4); MD5_STEP(MD5_H, a1, b1, c1, d1, ld32(p1 + 52), 0x289b7ec6u, 4);
        MD5_STEP(MD5_H, d0, a0, b0, c0, ld32(p0 + 0), 0xeaa127fau, 11); MD5_STEP(MD5_H, d1, a1, b1, c1, ld32(p1 + 0), 0xeaa127fau, 11);
        MD5_STEP(MD5_H, c0, d0, a0, b0, ld32(p0 + 12), 0xd4ef3085u, 16); MD5_STEP(MD5_H, c1, d1, a1, b1, ld32(p1 + 12), 0xd4ef3085u, 16);
        MD5_STEP(MD5_H, b0, c0, d0, a0, ld32(p0 + 24), 0x04881d05u, 23); MD5_STEP(MD5_H, b1, c1, d1, a1, ld32(p1 + 24), 0x04881d05u, 23);
        MD5_STEP(MD5_H, a0, b0, c0, d0, ld32(p0 + 36), 0xd9d4d039u, 4); MD5_STEP(MD5_H, a1, b1, c1, d1, ld32(p1 + 36), 0xd9d4d039u, 4);
        MD5_STEP(MD5_H, d0, a0, b0, c0, ld32(p0 + 48), 0xe6db99e5u, 11); MD5_STEP(MD5_H, d1, a1, b1, c1, ld32(p1 + 48), 0xe6db99e5u, 11);
        MD5_STEP(MD5_H, c0, d0, a0, b0, ld32(p0 + 60), 0x1fa27cf8u, 16); MD5_STEP(MD5_H, c1, d1, a1, b1, ld32(p1 + 60), 0x1fa27cf8u, 16);
        MD5_STEP(MD5_H, b0, c0, d0, a0, ld32(p0 + 8), 0xc4ac5665u, 23); MD5_STEP(MD5_H, b1, c1, d1, a1, ld32(p1 + 8), 0xc4ac5665u, 23);
        MD5_STEP(MD5_I, a0, b0, c0, d0, ld32(p0 + 0), 0xf4292244u, 6); MD5_STEP(MD5_I, a1, b1, c1, d1, ld32(p1 + 0), 0xf4292244u, 6);
        MD5_STEP(MD5_I, d0, a0, b0, c0, ld32(p0 + 28), 0x432aff97u, 10); MD5_STEP(MD5_I, d1, a1, b1, c1, ld32(p1 + 28), 0x432aff97u, 10);
        MD5_STEP(MD5_I, c0, d0, a0, b0, ld32(p0 + 56), 0xab9423a7u, 15); MD5_STEP(MD5_I, c1, d1, a1, b1, ld32(p1 + 56), 0xab9423a7u, 15);
        MD5_STEP(MD5_I, b0, c0, d0, a0, ld32(p0 + 20), 0xfc93a039u, 21); MD5_STEP(MD5_I, b1, c1, d1, a1, ld32(p1 + 20), 0xfc93a039u, 21);
        MD5_STEP(MD5_I, a0, b0, c0, d0, ld32(p0 + 48), 0x655b59c3u, 6); MD5_STEP(MD5_I, a1, b1, c1, d1, ld32(p1 + 48), 0x655b59c3u, 6);
        MD5_STEP(MD5_I, d0, a0, b0, c0, ld32(p0 + 12), 0x8f0ccc92u, 10); MD5_STEP(MD5_I, d1, a1, b1, c1, ld32(p1 + 12), 0x8f0ccc92u, 10);
        MD5_STEP(MD5_I, c0, d0, a0, b0, ld32(p0 + 40), 0xffeff47du, 15); MD5_STEP(MD5_I, c1, d1, a1, b1, ld32(p1 + 40), 0xffeff47du, 15);
        MD5_STEP(MD5_I, b0, c0, d0, a0, ld32(p0 + 4), 0x85845dd1u, 21); MD5_STEP(MD5_I, b1, c1, d1, a1, ld32(p1 + 4), 0x85845dd1u, 21);
        MD5_STEP(MD5_I, a0, b0, c0, d0, ld32(p0 + 32), 0x6fa87e4fu, 6); MD5_STEP(MD5_I, a1, b1, c1, d1, ld32(p1 + 32), 0x6fa87e4fu, 6);
        MD5_STEP(MD5_I, d0, a0, b0, c0, ld32(p0 + 60), 0xfe2ce6e0u, 10); MD5_STEP(MD5_I, d1, a1, b1, c1, ld32(p1 + 60), 0xfe2ce6e0u, 10);
        MD5_STEP(MD5_I, c0, d0, a0, b0, ld32(p0 + 24), 0xa3014314u, 15); MD5_STEP(MD5_I, c1, d1, a1, b1, ld32(p1 + 24), 0xa3014314u, 15);
        MD5_STEP(MD5_I, b0, c0, d0, a0, ld32(p0 + 52), 0x4e0811a1u, 21); MD5_STEP(MD5_I, b1, c1, d1, a1, ld32(p1 + 52), 0x4e0811a1u, 21);
        MD5_STEP(MD5_I, a0, b0, c0, d0, ld32(p0 + 16), 0xf7537e82u, 6); MD5_STEP(MD5_I, a1, b1, c1, d1, ld32(p1 + 16), 0xf7537e82u, 6);
        MD5_STEP(MD5_I, d0, a0, b0, c0, ld32(p0 + 44), 0xbd3af235u, 10); MD5_STEP(MD5_I, d1, a1, b1, c1, ld32(p1 + 44), 0xbd3af235u, 10);
        MD5_STEP(MD5_I, c0, d0, a0, b0, ld32(p0 + 8), 0x2ad7d2bbu, 15); MD5_STEP(MD5_I, c1, d1, a1, b1, ld32(p1 + 8), 0x2ad7d2bbu, 15);
        MD5_STEP(MD5_I, b0, c0, d0, a0, ld32(p0 + 36), 0xeb86d391u, 21); MD5_STEP(MD5_I, b1, c1, d1, a1, ld32(p1 + 36), 0xeb86d391u, 21);
        A0 += a0; B0 += b0; C0 += c0; D0 += d0;
        A1 += a1; B1 += b1; C1 += c1; D1 += d1;
    }
    s0[0] = A0; s0[1] = B0; s0[2] = C0; s0[3] = D0;
    s1[0] = A1; s1[1] = B1; s1[2] = C1; s1[3] = D1;
}

void Md5::update(const void *data, size_t len) {
    const uint8_t *p = static_cast<const uint8_t *>(data);
    size_t have = static_cast<size_t>(len_ & 63);
    len_ += len;
    if (have) {
        size_t need = 64 - have;
        if (len < need) {
            std::memcpy(buf_ + have, p, len);
            return;
        }
        std::memcpy(buf_ + have, p, need);
        block(buf_);
        p += need;
        len -= need;
    }
    for (; len >= 64; p += 64, len -= 64) block(p);
    if (len) std::memcpy(buf_, p, len);
}

void Md5::digest(uint8_t out[16]) const {
    Md5 t = *this;
    uint8_t pad[72] = {0x80};
    uint64_t bits = t.len_ * 8;
    size_t have = static_cast<size_t>(t.len_ & 63);
    size_t padlen = have < 56 ? 56 - have : 120 - have;
    t.update(pad, padlen);
    uint8_t lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = static_cast<uint8_t>(bits >> (8 * i));
    t.update(lenb, 8);
    const uint32_t s[4] = {t.a_, t.b_, t.c_, t.d_};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = static_cast<uint8_t>(s[i] >> (8 * j));
}

}  // namespace flacenc
