// md5_mb.cpp -- MD5 of up to 16 streams in lockstep (AVX-512), and the engine threads that feed it.
// See md5_mb.h.  RFC 1321 is restated here only as its step function on sixteen independent 32-bit lanes;
// the result is the scalar one word for word (host/checksums.cpp keeps padding and length handling).
#include "md5_mb.h"

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

// the 64 steps on sixteen lanes, written out (constant rotations and message words: the compiler keeps the
// sixteen words and the state in registers); F = b ? c : d, G = d ? b : c, H = b ^ c ^ d, I = c ^ (b | ~d) as
// truth tables of vpternlogd on (b, c, d).  (A + x + t) does not wait for the previous step: the chain is
// f -> add -> rotate -> add.
#define MD5X_STEP(IMM, A, B, C, D, W, T, S)                                                          \
    A = _mm512_add_epi32(B, _mm512_rol_epi32(                                                         \
               _mm512_add_epi32(_mm512_add_epi32(A, _mm512_add_epi32(x[W], _mm512_set1_epi32((int)(T)))), \
                                _mm512_ternarylogic_epi32(B, C, D, IMM)), S));
__attribute__((target("avx512f,avx512bw,avx512vl"))) inline void md5_64steps(__m512i &a, __m512i &b, __m512i &c, __m512i &d,
                                                                            const __m512i (&x)[16]) {
    MD5X_STEP(0xCA, a, b, c, d,  0, 0xd76aa478u,  7)
    MD5X_STEP(0xCA, d, a, b, c,  1, 0xe8c7b756u, 12)
    MD5X_STEP(0xCA, c, d, a, b,  2, 0x242070dbu, 17)
    MD5X_STEP(0xCA, b, c, d, a,  3, 0xc1bdceeeu, 22)
    MD5X_STEP(0xCA, a, b, c, d,  4, 0xf57c0fafu,  7)
    MD5X_STEP(0xCA, d, a, b, c,  5, 0x4787c62au, 12)
    MD5X_STEP(0xCA, c, d, a, b,  6, 0xa8304613u, 17)
    MD5X_STEP(0xCA, b, c, d, a,  7, 0xfd469501u, 22)
    MD5X_STEP(0xCA, a, b, c, d,  8, 0x698098d8u,  7)
    MD5X_STEP(0xCA, d, a, b, c,  9, 0x8b44f7afu, 12)
    MD5X_STEP(0xCA, c, d, a, b, 10, 0xffff5bb1u, 17)
    MD5X_STEP(0xCA, b, c, d, a, 11, 0x895cd7beu, 22)
    MD5X_STEP(0xCA, a, b, c, d, 12, 0x6b901122u,  7)
    MD5X_STEP(0xCA, d, a, b, c, 13, 0xfd987193u, 12)
    MD5X_STEP(0xCA, c, d, a, b, 14, 0xa679438eu, 17)
    MD5X_STEP(0xCA, b, c, d, a, 15, 0x49b40821u, 22)
    MD5X_STEP(0xE4, a, b, c, d,  1, 0xf61e2562u,  5)
    MD5X_STEP(0xE4, d, a, b, c,  6, 0xc040b340u,  9)
    MD5X_STEP(0xE4, c, d, a, b, 11, 0x265e5a51u, 14)
    MD5X_STEP(0xE4, b, c, d, a,  0, 0xe9b6c7aau, 20)
    MD5X_STEP(0xE4, a, b, c, d,  5, 0xd62f105du,  5)
    MD5X_STEP(0xE4, d, a, b, c, 10, 0x02441453u,  9)
    MD5X_STEP(0xE4, c, d, a, b, 15, 0xd8a1e681u, 14)
    MD5X_STEP(0xE4, b, c, d, a,  4, 0xe7d3fbc8u, 20)
    MD5X_STEP(0xE4, a, b, c, d,  9, 0x21e1cde6u,  5)
    MD5X_STEP(0xE4, d, a, b, c, 14, 0xc33707d6u,  9)
    MD5X_STEP(0xE4, c, d, a, b,  3, 0xf4d50d87u, 14)
    MD5X_STEP(0xE4, b, c, d, a,  8, 0x455a14edu, 20)
    MD5X_STEP(0xE4, a, b, c, d, 13, 0xa9e3e905u,  5)
    MD5X_STEP(0xE4, d, a, b, c,  2, 0xfcefa3f8u,  9)
    MD5X_STEP(0xE4, c, d, a, b,  7, 0x676f02d9u, 14)
    MD5X_STEP(0xE4, b, c, d, a, 12, 0x8d2a4c8au, 20)
    MD5X_STEP(0x96, a, b, c, d,  5, 0xfffa3942u,  4)
    MD5X_STEP(0x96, d, a, b, c,  8, 0x8771f681u, 11)
    MD5X_STEP(0x96, c, d, a, b, 11, 0x6d9d6122u, 16)
    MD5X_STEP(0x96, b, c, d, a, 14, 0xfde5380cu, 23)
    MD5X_STEP(0x96, a, b, c, d,  1, 0xa4beea44u,  4)
    MD5X_STEP(0x96, d, a, b, c,  4, 0x4bdecfa9u, 11)
    MD5X_STEP(0x96, c, d, a, b,  7, 0xf6bb4b60u, 16)
    MD5X_STEP(0x96, b, c, d, a, 10, 0xbebfbc70u, 23)
    MD5X_STEP(0x96, a, b, c, d, 13, 0x289b7ec6u,  4)
    MD5X_STEP(0x96, d, a, b, c,  0, 0xeaa127fau, 11)
    MD5X_STEP(0x96, c, d, a, b,  3, 0xd4ef3085u, 16)
    MD5X_STEP(0x96, b, c, d, a,  6, 0x04881d05u, 23)
    MD5X_STEP(0x96, a, b, c, d,  9, 0xd9d4d039u,  4)
    MD5X_STEP(0x96, d, a, b, c, 12, 0xe6db99e5u, 11)
    MD5X_STEP(0x96, c, d, a, b, 15, 0x1fa27cf8u, 16)
    MD5X_STEP(0x96, b, c, d, a,  2, 0xc4ac5665u, 23)
    MD5X_STEP(0x39, a, b, c, d,  0, 0xf4292244u,  6)
    MD5X_STEP(0x39, d, a, b, c,  7, 0x432aff97u, 10)
    MD5X_STEP(0x39, c, d, a, b, 14, 0xab9423a7u, 15)
    MD5X_STEP(0x39, b, c, d, a,  5, 0xfc93a039u, 21)
    MD5X_STEP(0x39, a, b, c, d, 12, 0x655b59c3u,  6)
    MD5X_STEP(0x39, d, a, b, c,  3, 0x8f0ccc92u, 10)
    MD5X_STEP(0x39, c, d, a, b, 10, 0xffeff47du, 15)
    MD5X_STEP(0x39, b, c, d, a,  1, 0x85845dd1u, 21)
    MD5X_STEP(0x39, a, b, c, d,  8, 0x6fa87e4fu,  6)
    MD5X_STEP(0x39, d, a, b, c, 15, 0xfe2ce6e0u, 10)
    MD5X_STEP(0x39, c, d, a, b,  6, 0xa3014314u, 15)
    MD5X_STEP(0x39, b, c, d, a, 13, 0x4e0811a1u, 21)
    MD5X_STEP(0x39, a, b, c, d,  4, 0xf7537e82u,  6)
    MD5X_STEP(0x39, d, a, b, c, 11, 0xbd3af235u, 10)
    MD5X_STEP(0x39, c, d, a, b,  2, 0x2ad7d2bbu, 15)
    MD5X_STEP(0x39, b, c, d, a,  9, 0xeb86d391u, 21)
}
#undef MD5X_STEP

__attribute__((target("avx512f,avx512bw,avx512vl"))) void md5_x16_avx512(uint32_t state[4][16], const uint8_t *const ptr[16],
                                                                         size_t nblocks, uint32_t mask) {
    const uint8_t *p[16];
    size_t step[16];
    for (int l = 0; l < 16; l++) {
        const bool on = (mask >> l) & 1u;
        p[l] = on ? ptr[l] : kDummy;
        step[l] = on ? 64 : 0;
    }
    __m512i A = _mm512_loadu_si512(state[0]), B = _mm512_loadu_si512(state[1]);
    __m512i C = _mm512_loadu_si512(state[2]), D = _mm512_loadu_si512(state[3]);
    for (size_t blk = 0; blk < nblocks; blk++) {
        __m512i x[16];
#pragma GCC unroll 16
        for (int l = 0; l < 16; l++) {
            x[l] = _mm512_loadu_si512(p[l]);
            _mm_prefetch(reinterpret_cast<const char *>(p[l]) + 1024, _MM_HINT_T0);   // sixteen sequential readers
            p[l] += step[l];
        }
        transpose16(x);
        __m512i a = A, b = B, c = C, d = D;
        md5_64steps(a, b, c, d, x);
        A = _mm512_add_epi32(A, a);
        B = _mm512_add_epi32(B, b);
        C = _mm512_add_epi32(C, c);
        D = _mm512_add_epi32(D, d);
    }
    const __mmask16 k = (__mmask16)mask;
    _mm512_mask_storeu_epi32(state[0], k, A);
    _mm512_mask_storeu_epi32(state[1], k, B);
    _mm512_mask_storeu_epi32(state[2], k, C);
    _mm512_mask_storeu_epi32(state[3], k, D);
}
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

void md5_blocks_x16(uint32_t state[4][16], const uint8_t *const ptr[16], size_t nblocks, uint32_t mask) {
#ifdef FLACENC_MD5_X86
    if (Md5Pool::simd_available()) {
        md5_x16_avx512(state, ptr, nblocks, mask);
        return;
    }
#endif
    for (int l = 0; l < 16; l++) {
        if (!((mask >> l) & 1u)) continue;
        uint32_t s[4] = {state[0][l], state[1][l], state[2][l], state[3][l]};
        for (size_t b = 0; b < nblocks; b++) Md5::transform(s, ptr[l] + 64 * b);
        for (int w = 0; w < 4; w++) state[w][l] = s[w];
    }
}

// ---- the engines ---------------------------------------------------------------------------------------
// An engine thread serves any number of attached streams; in every pass it takes up to 16 of those that
// have bytes waiting (round robin, so that nobody starves), advances them together by the whole blocks the
// shortest of them has (at most kMaxBlocks), and goes back for the next pick.  The more streams wait, the
// fuller the register: one engine at 16 lanes hashes what ten scalar threads hash.
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
    static constexpr int kLanes = 16;
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
        alignas(64) uint32_t st[4][kLanes];
        for (;;) {
            // pick: up to 16 streams with a run in hand or waiting
            Md5Lane *act[kLanes];
            int na = 0;
            std::vector<Md5Lane *> started;
            const size_t n = lanes.size();
            for (size_t k = 0; k < n && na < kLanes; k++) {
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
            Md5Lane *work[kLanes];
            int nw = 0;
            size_t common = kMaxBlocks;
            for (int i = 0; i < na; i++)
                if (act[i]->blocks) {
                    work[nw++] = act[i];
                    common = std::min(common, act[i]->blocks);
                }
            if (nw >= 2 && Md5Pool::simd_available()) {
                const uint8_t *ptr[kLanes];
                for (int i = 0; i < kLanes; i++) ptr[i] = kDummy;
                for (int i = 0; i < nw; i++) {
                    uint32_t w4[4];
                    work[i]->md5->get_state(w4);
                    for (int w = 0; w < 4; w++) st[w][i] = w4[w];
                    ptr[i] = work[i]->p;
                }
                md5_blocks_x16(st, ptr, common, (1u << nw) - 1u);
                for (int i = 0; i < nw; i++) {
                    const uint32_t w4[4] = {st[0][i], st[1][i], st[2][i], st[3][i]};
                    work[i]->md5->set_state(w4);
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
            Md5Lane *finished[kLanes];
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
    // an engine fills its 16 lanes before the next engine thread is started; after that the least loaded
    Md5Engine *best = nullptr;
    for (auto &e : impl_->engines) {
        std::lock_guard<std::mutex> lock(e->mu);
        if (e->lanes.size() < (size_t)Md5Engine::kLanes && (!best || e->lanes.size() > best->lanes.size())) best = e.get();
    }
    if (!best && impl_->engines.size() < impl_->max_engines) {
        impl_->engines.emplace_back(new Md5Engine());
        best = impl_->engines.back().get();
        best->th = std::thread([best] { best->run(); });
    }
    if (!best)
        for (auto &e : impl_->engines) {
            std::lock_guard<std::mutex> lock(e->mu);
            if (!best || e->lanes.size() < best->lanes.size()) best = e.get();
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
extern "C" int flacenc_md5_simd_available(void) { return flacenc::Md5Pool::simd_available() ? 1 : 0; }
