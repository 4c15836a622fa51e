// md5_mb.cpp -- MD5 of up to 16 streams in lockstep (AVX-512), and the engine threads that feed it.
// See md5_mb.h.  RFC 1321 is restated here only as its step function on sixteen independent 32-bit lanes;
// the result is the scalar one word for word (host/checksums.cpp keeps padding and length handling).
#include "md5_mb.h"
#include "host_internal.h"

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define FLACENC_MD5_X86 1
#endif

namespace flacenc {

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

alignas(64) const uint8_t kDummy[64] = {0};

#ifdef FLACENC_MD5_X86
// x[w] = message word w of the sixteen blocks (one per lane): a 16 x 16 transpose of 32-bit words
__attribute__((target("avx512f,avx512bw,avx512vl"))) inline void transpose16(__m512i (&r)[16]) {
    __m512i t[16], u[16];
#pragma GCC unroll 8
    for (int i = 0; i < 8; i++) {
        t[2 * i] = _mm512_unpacklo_epi32(r[2 * i], r[2 * i + 1]);
        t[2 * i + 1] = _mm512_unpackhi_epi32(r[2 * i], r[2 * i + 1]);
    }
#pragma GCC unroll 4
    for (int g = 0; g < 4; g++) {   // u[4 g + j], 128-bit lane q: word 4 q + j of rows 4 g .. 4 g + 3
        u[4 * g + 0] = _mm512_unpacklo_epi64(t[4 * g + 0], t[4 * g + 2]);
        u[4 * g + 1] = _mm512_unpackhi_epi64(t[4 * g + 0], t[4 * g + 2]);
        u[4 * g + 2] = _mm512_unpacklo_epi64(t[4 * g + 1], t[4 * g + 3]);
        u[4 * g + 3] = _mm512_unpackhi_epi64(t[4 * g + 1], t[4 * g + 3]);
    }
#pragma GCC unroll 4
    for (int j = 0; j < 4; j++) {   // a 4 x 4 transpose of 128-bit lanes per j
        const __m512i v0 = _mm512_shuffle_i32x4(u[j], u[4 + j], 0x88);
        const __m512i v1 = _mm512_shuffle_i32x4(u[j], u[4 + j], 0xdd);
        const __m512i v2 = _mm512_shuffle_i32x4(u[8 + j], u[12 + j], 0x88);
        const __m512i v3 = _mm512_shuffle_i32x4(u[8 + j], u[12 + j], 0xdd);
        r[0 + j] = _mm512_shuffle_i32x4(v0, v2, 0x88);
        r[8 + j] = _mm512_shuffle_i32x4(v0, v2, 0xdd);
        r[4 + j] = _mm512_shuffle_i32x4(v1, v3, 0x88);
        r[12 + j] = _mm512_shuffle_i32x4(v1, v3, 0xdd);
    }
}

// The 64 steps on G groups of sixteen lanes (G = 1..4: up to 64 chains per call).  A step's dependent chain is
// f -> add -> rotate -> add (four vector operations: 4 cycles per step where they take one cycle each, 8 on cores whose
// vector integer units take two, e.g. Zen 5) while its six operations keep one of several vector pipes busy for a cycle: one
// group alone leaves most issue slots empty, so G independent groups advance side by side, step by step (measured on an EPYC
// 9575F, one thread: 9.8 GB/s with one group -- 7.9 before the transposes were moved between the rounds --, 17.3 with two,
// 18.1 with three, 13.7 with four, whose sixteen state registers leave too few for the rest).  The message words live in memory (x[group][word], L1-resident: the steps read them as memory operands, so four groups'
// states fit the register file), written by the 16 x 16 transposes -- the NEXT block's, group by group, between the rounds of
// the current one, where the out-of-order window finds them beside the waiting chains.
// F = b ? c : d, G = d ? b : c, H = b ^ c ^ d, I = c ^ (b | ~d) as truth tables of vpternlogd on (b, c, d).
// (the two additions that do not depend on the previous step are inline asm with memory operands: the compiler otherwise
// re-associates the sum so that F + T sits on the dependent chain, and copies every message word into spill slots of its own)
alignas(64) const uint32_t kMd5T[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
#define MD5G_STEP(IMM, A, B, C, D, W, I, S)                                                                              \
    _Pragma("GCC unroll 4") for (int g = 0; g < G; g++) {                                                                \
        __m512i t_;                                                                                                      \
        asm("vpaddd %2, %1, %0" : "=v"(t_) : "v"(A[g]), "m"(xc[g][W]));                                                  \
        asm("vpaddd %2%{1to16%}, %1, %0" : "=v"(t_) : "v"(t_), "m"(kMd5T[I]));                                           \
        t_ = _mm512_add_epi32(t_, _mm512_ternarylogic_epi32(B[g], C[g], D[g], IMM));                                     \
        A[g] = _mm512_add_epi32(B[g], _mm512_rol_epi32(t_, S));                                                          \
    }

// a group's next block: sixteen rows loaded (and their readers advanced), transposed into x[word]
__attribute__((target("avx512f,avx512bw,avx512vl"), always_inline)) inline void md5_fetch_block(const uint8_t *(&p)[16], const size_t (&step)[16],
                                                                                             __m512i *dst) {
    __m512i r[16];
#pragma GCC unroll 16
    for (int l = 0; l < 16; l++) {
        r[l] = _mm512_loadu_si512(p[l]);
        _mm_prefetch(reinterpret_cast<const char *>(p[l]) + 1024, _MM_HINT_T0);   // sixteen sequential readers
        p[l] += step[l];
    }
    transpose16(r);
#pragma GCC unroll 16
    for (int l = 0; l < 16; l++) dst[l] = r[l];
}

template <int G>
__attribute__((target("avx512f,avx512bw,avx512vl"))) void md5_groups_avx512(uint32_t (*state)[4][16], const uint8_t *const (*ptr)[16],
                                                                            size_t nblocks, const uint32_t *mask) {
    const uint8_t *p[G][16];
    size_t step[G][16];
    for (int g = 0; g < G; g++)
        for (int l = 0; l < 16; l++) {
            const bool on = (mask[g] >> l) & 1u;
            p[g][l] = on ? ptr[g][l] : kDummy;
            step[g][l] = on ? 64 : 0;
        }
    __m512i SA[G], SB[G], SC[G], SD[G];
    for (int g = 0; g < G; g++) {
        SA[g] = _mm512_loadu_si512(state[g][0]);
        SB[g] = _mm512_loadu_si512(state[g][1]);
        SC[g] = _mm512_loadu_si512(state[g][2]);
        SD[g] = _mm512_loadu_si512(state[g][3]);
    }
    // (the two buffers 4 KB + 192 bytes apart at most: a word being stored for the next block and the word being read at the
    // same position of the current one never share their low 12 address bits)
    alignas(64) __m512i xbuf[2][G * 16 + 3];
#define fetch(g, dst) md5_fetch_block(p[g], step[g], dst)
    if (nblocks)
        for (int g = 0; g < G; g++) fetch(g, xbuf[0] + 16 * g);
    for (size_t blk = 0; blk < nblocks; blk++) {
        const __m512i(*xc)[16] = reinterpret_cast<const __m512i(*)[16]>(xbuf[blk & 1]);
        __m512i(*xn)[16] = reinterpret_cast<__m512i(*)[16]>(xbuf[(blk & 1) ^ 1]);
        const bool more = blk + 1 < nblocks;
        __m512i a[G], b[G], c[G], d[G];
        for (int g = 0; g < G; g++) {
            a[g] = SA[g];
            b[g] = SB[g];
            c[g] = SC[g];
            d[g] = SD[g];
        }
        MD5G_STEP(0xCA, a, b, c, d,  0,  0,  7)
        MD5G_STEP(0xCA, d, a, b, c,  1,  1, 12)
        MD5G_STEP(0xCA, c, d, a, b,  2,  2, 17)
        MD5G_STEP(0xCA, b, c, d, a,  3,  3, 22)
        MD5G_STEP(0xCA, a, b, c, d,  4,  4,  7)
        MD5G_STEP(0xCA, d, a, b, c,  5,  5, 12)
        MD5G_STEP(0xCA, c, d, a, b,  6,  6, 17)
        MD5G_STEP(0xCA, b, c, d, a,  7,  7, 22)
        MD5G_STEP(0xCA, a, b, c, d,  8,  8,  7)
        MD5G_STEP(0xCA, d, a, b, c,  9,  9, 12)
        MD5G_STEP(0xCA, c, d, a, b, 10, 10, 17)
        MD5G_STEP(0xCA, b, c, d, a, 11, 11, 22)
        MD5G_STEP(0xCA, a, b, c, d, 12, 12,  7)
        MD5G_STEP(0xCA, d, a, b, c, 13, 13, 12)
        MD5G_STEP(0xCA, c, d, a, b, 14, 14, 17)
        MD5G_STEP(0xCA, b, c, d, a, 15, 15, 22)
        if (more) fetch(0, xn[0]);
        MD5G_STEP(0xE4, a, b, c, d,  1, 16,  5)
        MD5G_STEP(0xE4, d, a, b, c,  6, 17,  9)
        MD5G_STEP(0xE4, c, d, a, b, 11, 18, 14)
        MD5G_STEP(0xE4, b, c, d, a,  0, 19, 20)
        MD5G_STEP(0xE4, a, b, c, d,  5, 20,  5)
        MD5G_STEP(0xE4, d, a, b, c, 10, 21,  9)
        MD5G_STEP(0xE4, c, d, a, b, 15, 22, 14)
        MD5G_STEP(0xE4, b, c, d, a,  4, 23, 20)
        MD5G_STEP(0xE4, a, b, c, d,  9, 24,  5)
        MD5G_STEP(0xE4, d, a, b, c, 14, 25,  9)
        MD5G_STEP(0xE4, c, d, a, b,  3, 26, 14)
        MD5G_STEP(0xE4, b, c, d, a,  8, 27, 20)
        MD5G_STEP(0xE4, a, b, c, d, 13, 28,  5)
        MD5G_STEP(0xE4, d, a, b, c,  2, 29,  9)
        MD5G_STEP(0xE4, c, d, a, b,  7, 30, 14)
        MD5G_STEP(0xE4, b, c, d, a, 12, 31, 20)
        if (G > 1 && more) fetch(G > 1 ? 1 : 0, xn[G > 1 ? 1 : 0]);
        MD5G_STEP(0x96, a, b, c, d,  5, 32,  4)
        MD5G_STEP(0x96, d, a, b, c,  8, 33, 11)
        MD5G_STEP(0x96, c, d, a, b, 11, 34, 16)
        MD5G_STEP(0x96, b, c, d, a, 14, 35, 23)
        MD5G_STEP(0x96, a, b, c, d,  1, 36,  4)
        MD5G_STEP(0x96, d, a, b, c,  4, 37, 11)
        MD5G_STEP(0x96, c, d, a, b,  7, 38, 16)
        MD5G_STEP(0x96, b, c, d, a, 10, 39, 23)
        MD5G_STEP(0x96, a, b, c, d, 13, 40,  4)
        MD5G_STEP(0x96, d, a, b, c,  0, 41, 11)
        MD5G_STEP(0x96, c, d, a, b,  3, 42, 16)
        MD5G_STEP(0x96, b, c, d, a,  6, 43, 23)
        MD5G_STEP(0x96, a, b, c, d,  9, 44,  4)
        MD5G_STEP(0x96, d, a, b, c, 12, 45, 11)
        MD5G_STEP(0x96, c, d, a, b, 15, 46, 16)
        MD5G_STEP(0x96, b, c, d, a,  2, 47, 23)
        if (G > 2 && more) fetch(G > 2 ? 2 : 0, xn[G > 2 ? 2 : 0]);
        MD5G_STEP(0x39, a, b, c, d,  0, 48,  6)
        MD5G_STEP(0x39, d, a, b, c,  7, 49, 10)
        MD5G_STEP(0x39, c, d, a, b, 14, 50, 15)
        MD5G_STEP(0x39, b, c, d, a,  5, 51, 21)
        MD5G_STEP(0x39, a, b, c, d, 12, 52,  6)
        MD5G_STEP(0x39, d, a, b, c,  3, 53, 10)
        MD5G_STEP(0x39, c, d, a, b, 10, 54, 15)
        MD5G_STEP(0x39, b, c, d, a,  1, 55, 21)
        MD5G_STEP(0x39, a, b, c, d,  8, 56,  6)
        MD5G_STEP(0x39, d, a, b, c, 15, 57, 10)
        MD5G_STEP(0x39, c, d, a, b,  6, 58, 15)
        MD5G_STEP(0x39, b, c, d, a, 13, 59, 21)
        MD5G_STEP(0x39, a, b, c, d,  4, 60,  6)
        MD5G_STEP(0x39, d, a, b, c, 11, 61, 10)
        MD5G_STEP(0x39, c, d, a, b,  2, 62, 15)
        MD5G_STEP(0x39, b, c, d, a,  9, 63, 21)
        if (G > 3 && more) fetch(G > 3 ? 3 : 0, xn[G > 3 ? 3 : 0]);
        for (int g = 0; g < G; g++) {
            SA[g] = _mm512_add_epi32(SA[g], a[g]);
            SB[g] = _mm512_add_epi32(SB[g], b[g]);
            SC[g] = _mm512_add_epi32(SC[g], c[g]);
            SD[g] = _mm512_add_epi32(SD[g], d[g]);
        }
    }
    for (int g = 0; g < G; g++) {
        const __mmask16 k = (__mmask16)mask[g];
        _mm512_mask_storeu_epi32(state[g][0], k, SA[g]);
        _mm512_mask_storeu_epi32(state[g][1], k, SB[g]);
        _mm512_mask_storeu_epi32(state[g][2], k, SC[g]);
        _mm512_mask_storeu_epi32(state[g][3], k, SD[g]);
    }
}
#undef fetch
#undef MD5G_STEP
#endif

bool detect_simd() {
#ifdef FLACENC_MD5_X86
    if (std::getenv("FLACENC_MD5_SCALAR")) return false;
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
#else
    return false;
#endif
}

}  // namespace

bool Md5Pool::simd_available() {
    static const bool have = detect_simd();
    return have;
}

void md5_blocks_groups(uint32_t (*state)[4][16], const uint8_t *const (*ptr)[16], size_t nblocks, const uint32_t *mask, int groups) {
#ifdef FLACENC_MD5_X86
    if (Md5Pool::simd_available()) {
        switch (groups) {
        case 1: md5_groups_avx512<1>(state, ptr, nblocks, mask); return;
        case 2: md5_groups_avx512<2>(state, ptr, nblocks, mask); return;
        case 3: md5_groups_avx512<3>(state, ptr, nblocks, mask); return;
        case 4: md5_groups_avx512<4>(state, ptr, nblocks, mask); return;
        default: break;
        }
    }
#endif
    for (int g = 0; g < groups; g++)
        for (int l = 0; l < 16; l++) {
            if (!((mask[g] >> l) & 1u)) continue;
            uint32_t s[4] = {state[g][0][l], state[g][1][l], state[g][2][l], state[g][3][l]};
            for (size_t b = 0; b < nblocks; b++) Md5::transform(s, ptr[g][l] + 64 * b);
            for (int w = 0; w < 4; w++) state[g][w][l] = s[w];
        }
}

void md5_blocks_x16(uint32_t state[4][16], const uint8_t *const ptr[16], size_t nblocks, uint32_t mask) {
    md5_blocks_groups(reinterpret_cast<uint32_t(*)[4][16]>(state), reinterpret_cast<const uint8_t *const(*)[16]>(ptr), nblocks, &mask, 1);
}

// ---- the engines ---------------------------------------------------------------------------------------
// An engine thread serves any number of attached streams; in every pass it takes up to 48 of those that
// have bytes waiting (round robin, so that nobody starves) -- three groups of sixteen lanes whose steps are
// interleaved (md5_groups_avx512) --, advances them together by the whole blocks the shortest of them has (at
// most kMaxBlocks), and goes back for the next pick.  The more streams wait, the fuller the registers and the
// issue slots: a chain's speed is set by the latency of its dependent steps, so forty-eight of them cost little
// more than sixteen.
struct Md5Engine;
struct Md5Lane {
    Md5Engine *engine = nullptr;
    Md5 *md5 = nullptr;
    std::deque<std::pair<const uint8_t *, size_t>> q;   // runs not yet started (engine mutex)
    uint64_t pushed = 0, done = 0;
    double busy_ms = 0;
    // the run in hand (engine thread only)
    bool busy = false;
    const uint8_t *p = nullptr;
    size_t blocks = 0, tail = 0;
};

struct Md5Engine {
    static constexpr int kLanes = 16;            // lanes of a group; an engine is given its first 16 streams before the next engine starts
    static constexpr int kGroups = 3;            // groups per pass (a fourth loses: register pressure)
    static constexpr int kAct = kLanes * kGroups;
    static constexpr size_t kMaxBlocks = 2048;   // blocks per lockstep pass (128 KB per lane)
    static constexpr size_t kShort = 128;        // remainders below 8 KB do not join a pass
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Md5Lane *> lanes;    // attached streams
    size_t next = 0;                 // round-robin start of the next pick
    bool stop = false;
    std::thread th;

    void run() {
        std::unique_lock<std::mutex> lock(mu);
        alignas(64) uint32_t st[kGroups][4][kLanes];
        for (;;) {
            // pick: up to 48 streams with a run in hand or waiting
            Md5Lane *act[kAct];
            int na = 0;
            std::vector<Md5Lane *> started;
            const size_t n = lanes.size();
            for (size_t k = 0; k < n && na < kAct; k++) {
                Md5Lane *l = lanes[(next + k) % n];
                if (!l->busy && !l->q.empty()) {
                    l->p = l->q.front().first;
                    l->blocks = l->q.front().second;   // bytes for now; split below, outside the lock
                    l->q.pop_front();
                    l->busy = true;
                    started.push_back(l);
                }
                if (l->busy) act[na++] = l;
            }
            if (!na) {
                if (stop) return;
                cv.wait(lock);
                continue;
            }
            if (n) next = (next + 1) % n;
            lock.unlock();
            const double t0 = now_ms();
            for (Md5Lane *l : started) {   // the bytes that complete a buffered partial block go the scalar way
                size_t bytes = l->blocks;
                const size_t have = l->md5->buffered();
                if (have) {
                    const size_t take = std::min(bytes, 64 - have);
                    l->md5->update(l->p, take);
                    l->p += take;
                    bytes -= take;
                }
                l->blocks = bytes / 64;
                l->tail = bytes % 64;
            }
            // a run's short remainder would cut the whole pass down to its length: finish it on the scalar code
            for (int i = 0; i < na; i++)
                if (act[i]->blocks && act[i]->blocks < kShort) {
                    act[i]->md5->update(act[i]->p, 64 * act[i]->blocks);
                    act[i]->p += 64 * act[i]->blocks;
                    act[i]->blocks = 0;
                }
            Md5Lane *work[kAct];
            int nw = 0;
            size_t common = kMaxBlocks;
            for (int i = 0; i < na; i++)
                if (act[i]->blocks) {
                    work[nw++] = act[i];
                    common = std::min(common, act[i]->blocks);
                }
            if (nw >= 3 && Md5Pool::simd_available()) {
                // lanes dealt round the groups: every group about equally full
                const int groups = (nw + kLanes - 1) / kLanes;
                const uint8_t *ptr[kGroups][kLanes];
                uint32_t mask[kGroups] = {};
                for (int g = 0; g < kGroups; g++)
                    for (int i = 0; i < kLanes; i++) ptr[g][i] = kDummy;
                for (int i = 0; i < nw; i++) {
                    const int g = i % groups, l = i / groups;
                    uint32_t w4[4];
                    work[i]->md5->get_state(w4);
                    for (int w = 0; w < 4; w++) st[g][w][l] = w4[w];
                    ptr[g][l] = work[i]->p;
                    mask[g] |= 1u << l;
                }
                md5_blocks_groups(st, ptr, common, mask, groups);
                for (int i = 0; i < nw; i++) {
                    const int g = i % groups, l = i / groups;
                    const uint32_t w4[4] = {st[g][0][l], st[g][1][l], st[g][2][l], st[g][3][l]};
                    work[i]->md5->set_state(w4);
                    work[i]->md5->add_blocks(common);
                    work[i]->p += 64 * common;
                    work[i]->blocks -= common;
                }
            } else if (nw == 2) {
                // TWO chains: the scalar code with both interleaved in one instruction stream (Md5::transform2) -- each runs
                // at the scalar chain's speed (1.05 GB/s on the bench host, a SIMD lane 0.6): what a handful of long streams
                // spread two to an engine get
                uint32_t s0[4], s1[4];
                work[0]->md5->get_state(s0);
                work[1]->md5->get_state(s1);
                Md5::transform2(s0, work[0]->p, s1, work[1]->p, common);
                work[0]->md5->set_state(s0);
                work[1]->md5->set_state(s1);
                for (int i = 0; i < 2; i++) {
                    work[i]->md5->add_blocks(common);
                    work[i]->p += 64 * common;
                    work[i]->blocks -= common;
                }
            } else {   // a lone chain is faster on the scalar code (shorter dependency chain per step)
                for (int i = 0; i < nw; i++) {
                    const size_t m = std::min(work[i]->blocks, kMaxBlocks);
                    work[i]->md5->update(work[i]->p, 64 * m);
                    work[i]->p += 64 * m;
                    work[i]->blocks -= m;
                }
            }
            Md5Lane *finished[kAct];
            int nf = 0;
            for (int i = 0; i < na; i++)
                if (act[i]->blocks == 0) {
                    if (act[i]->tail) act[i]->md5->update(act[i]->p, act[i]->tail);
                    finished[nf++] = act[i];
                }
            const double dt = now_ms() - t0;
            lock.lock();
            for (int i = 0; i < na; i++) act[i]->busy_ms += dt / na;
            for (int i = 0; i < nf; i++) {
                finished[i]->busy = false;
                finished[i]->done++;
            }
            if (nf) cv.notify_all();
        }
    }
};

struct Md5Pool::Impl {
    std::mutex mu;
    std::vector<std::unique_ptr<Md5Engine>> engines;
    unsigned max_engines = 4;
};

Md5Pool::Md5Pool() : impl_(new Impl) {
    // half the CPUs the process may use (its cgroup quota), four to eight
    impl_->max_engines = std::min(8u, std::max(4u, flacenc_host::usable_cpus() / 2));
    if (const char *e = std::getenv("FLACENC_MD5_ENGINES")) impl_->max_engines = (unsigned)std::max(1, std::atoi(e));
}
Md5Pool::~Md5Pool() {
    for (auto &e : impl_->engines) {
        {
            std::lock_guard<std::mutex> lock(e->mu);
            e->stop = true;
            e->cv.notify_all();
        }
        if (e->th.joinable()) e->th.join();
    }
    delete impl_;
}
Md5Pool &Md5Pool::get() {
    static Md5Pool pool;
    return pool;
}

Md5Lane *Md5Pool::attach(Md5 *state) {
    std::lock_guard<std::mutex> plock(impl_->mu);
    // Streams are SPREAD: an engine is given two (a pair runs on the scalar code at 1.05 GB/s per chain, where a lane of the
    // lockstep SIMD step makes 0.6), then the next engine thread is started; once every engine the pool may have is running,
    // a newcomer joins the one with the fewest streams -- a handful of long streams get scalar speed, hundreds fill the
    // engines' three lockstep groups evenly.  (Idle engines sleep: their number costs nothing but when they work.)
    Md5Engine *best = nullptr;
    size_t best_n = 0;
    for (auto &e : impl_->engines) {
        std::lock_guard<std::mutex> lock(e->mu);
        const size_t n = e->lanes.size();
        if (!best || n < best_n) {
            best = e.get();
            best_n = n;
        }
    }
    if ((!best || best_n >= 2) && impl_->engines.size() < impl_->max_engines) {
        impl_->engines.emplace_back(new Md5Engine());
        best = impl_->engines.back().get();
        best->th = std::thread([best] { best->run(); });
    }
    if (!best) return nullptr;
    Md5Lane *l = new Md5Lane();
    l->engine = best;
    l->md5 = state;
    std::lock_guard<std::mutex> lock(best->mu);
    best->lanes.push_back(l);
    return l;
}

void Md5Pool::detach(Md5Lane *lane) {
    if (!lane) return;
    Md5Engine *e = lane->engine;
    {
        std::unique_lock<std::mutex> lock(e->mu);
        e->cv.wait(lock, [&] { return lane->done >= lane->pushed; });
        for (size_t i = 0; i < e->lanes.size(); i++)
            if (e->lanes[i] == lane) {
                e->lanes.erase(e->lanes.begin() + (long)i);
                break;
            }
    }
    delete lane;
}

uint64_t Md5Pool::push(Md5Lane *lane, const uint8_t *p, size_t n) {
    Md5Engine *e = lane->engine;
    std::lock_guard<std::mutex> lock(e->mu);
    lane->q.emplace_back(p, n);
    const uint64_t ticket = ++lane->pushed;
    e->cv.notify_all();
    return ticket;
}

void Md5Pool::wait(Md5Lane *lane, uint64_t ticket) {
    Md5Engine *e = lane->engine;
    std::unique_lock<std::mutex> lock(e->mu);
    e->cv.wait(lock, [&] { return lane->done >= ticket; });
}

uint64_t Md5Pool::pushed(const Md5Lane *lane) const {
    std::lock_guard<std::mutex> lock(lane->engine->mu);
    return lane->pushed;
}
double Md5Pool::busy_ms(const Md5Lane *lane) const {
    std::lock_guard<std::mutex> lock(lane->engine->mu);
    return lane->busy_ms;
}

}  // namespace flacenc

// Test hook (tests/test_md5_pool.py): `streams` chains fed through the pool in runs of assorted lengths,
// digests compared with the scalar class on the same bytes.  Returns the number of mismatching digests.
extern "C" int flacenc_md5_selftest(uint32_t streams, uint32_t runs, uint32_t seed) {
    using namespace flacenc;
    struct S {
        std::vector<uint8_t> data;
        std::vector<size_t> cuts;
        Md5 pooled, scalar;
        Md5Lane *lane = nullptr;
    };
    std::vector<std::unique_ptr<S>> ss;
    uint64_t x = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&]() {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        return x;
    };
    for (uint32_t i = 0; i < streams; i++) {
        std::unique_ptr<S> s(new S());
        size_t total = 0;
        for (uint32_t r = 0; r < runs; r++) {
            const uint64_t k = rnd() % 6;
            const size_t n = k == 0 ? rnd() % 64 : k == 1 ? 64 * (rnd() % 40) : k == 2 ? 0 : rnd() % 200000;
            s->cuts.push_back(n);
            total += n;
        }
        s->data.resize(total + 1);
        for (size_t j = 0; j < total; j += 8) {
            const uint64_t v = rnd();
            std::memcpy(s->data.data() + j, &v, std::min<size_t>(8, total - j));
        }
        s->lane = Md5Pool::get().attach(&s->pooled);
        ss.push_back(std::move(s));
    }
    for (uint32_t r = 0; r < runs; r++)   // runs of all streams interleaved, as concurrent writers push them
        for (auto &s : ss) {
            size_t off = 0;
            for (uint32_t q = 0; q < r; q++) off += s->cuts[q];
            if (s->lane) Md5Pool::get().push(s->lane, s->data.data() + off, s->cuts[r]);
            else s->pooled.update(s->data.data() + off, s->cuts[r]);
        }
    int bad = 0;
    for (auto &s : ss) {
        if (s->lane) Md5Pool::get().detach(s->lane);
        size_t total = 0;
        for (size_t n : s->cuts) total += n;
        s->scalar.update(s->data.data(), total);
        uint8_t a[16], b[16];
        s->pooled.digest(a);
        s->scalar.digest(b);
        if (std::memcmp(a, b, 16)) bad++;
    }
    return bad;
}
// Diagnostic (bench.py): GB/s of `lanes` (1..48) independent chains advanced in lockstep on one thread over kib_per_lane KiB
// each -- divided by the lane count, the speed of ONE chain: the per-stream bound of every front end.
extern "C" double flacenc_md5_probe(uint32_t lanes, uint32_t kib_per_lane) {
    using namespace flacenc;
    if (lanes < 1 || lanes > 48 || kib_per_lane == 0) return 0.0;
    const size_t per = (size_t)kib_per_lane << 10;
    std::vector<uint8_t> buf(per * lanes);
    for (size_t i = 0; i < buf.size(); i++) buf[i] = (uint8_t)(i * 2654435761u >> 11);
    const int groups = (int)((lanes + 15) / 16);
    alignas(64) uint32_t st[3][4][16] = {};
    const uint8_t *ptr[3][16] = {};
    uint32_t mask[3] = {0, 0, 0};
    for (uint32_t i = 0; i < lanes; i++) {
        const int g = (int)(i % groups), l = (int)(i / groups);
        ptr[g][l] = buf.data() + per * i;
        mask[g] |= 1u << l;
    }
    double best = 0.0;
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now_ms();
        md5_blocks_groups(st, ptr, per / 64, mask, groups);
        const double dt = now_ms() - t0;
        if (dt > 0) best = std::max(best, (double)buf.size() / dt * 1e-6);
    }
    return best;
}
extern "C" int flacenc_md5_simd_available(void) { return flacenc::Md5Pool::simd_available() ? 1 : 0; }
