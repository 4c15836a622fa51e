// md5_mb.h -- MD5 of many streams at once.
//
// A stream's MD5 (STREAMINFO; encode.rs:571, 1292-1318, 2100) is one serial chain: 64 dependent steps per
// 64-byte block, about 1 GB/s on one core whatever its width.  Chains of DIFFERENT streams are independent,
// so an engine thread advances up to 16 of them in lockstep, one per 32-bit lane of an AVX-512 register
// (RFC 1321's step is add / ternary logic / rotate, all lane-wise): the blocks of 16 streams cost what one
// costs.  Writers that encode many streams side by side (flacenc_encode_many) hand their byte runs to the
// pool instead of to a worker thread of their own; the digest is bit for bit the scalar one (same state
// words, same padding), and a host without AVX-512 runs the scalar code on the engine threads.
#pragma once
#include <cstddef>
#include <cstdint>

#include "checksums.h"

namespace flacenc {

struct Md5Lane;   // one attached stream

class Md5Pool {
public:
    static Md5Pool &get();
    // `state` must outlive the lane; an engine serves any number of streams, 16 at a time
    Md5Lane *attach(Md5 *state);
    void detach(Md5Lane *lane);                      // waits for the lane's queued runs
    // queue [p, p + n) behind the lane's earlier runs; the bytes must stay valid until wait(ticket) returns
    uint64_t push(Md5Lane *lane, const uint8_t *p, size_t n);
    void wait(Md5Lane *lane, uint64_t ticket);
    uint64_t pushed(const Md5Lane *lane) const;
    double busy_ms(const Md5Lane *lane) const;
    static bool simd_available();                    // AVX-512 F + BW + VL on this host

private:
    Md5Pool();
    ~Md5Pool();
    struct Impl;
    Impl *impl_;
};

// whole 64-byte blocks of up to 16 independent chains in lockstep: state[w][lane], ptr[lane] -> that lane's
// next block (advanced by 64 bytes per block by the callee's caller); lanes whose bit in `mask` is clear are
// ignored (their state is left alone).  Scalar fallback inside when AVX-512 is missing.
void md5_blocks_x16(uint32_t state[4][16], const uint8_t *const ptr[16], size_t nblocks, uint32_t mask);
// the same for `groups` (1..4) groups of sixteen chains whose steps are interleaved: a chain's speed is bound by the
// latency of its 64 dependent steps, so up to four groups run in about the time of one
void md5_blocks_groups(uint32_t (*state)[4][16], const uint8_t *const (*ptr)[16], size_t nblocks, const uint32_t *mask, int groups);

}  // namespace flacenc
