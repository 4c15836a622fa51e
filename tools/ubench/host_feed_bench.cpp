// What the host side of the feeding paths can move on this box: int32 -> 3-byte packing (AVX2 shuffle), plain memcpy and
// the 16-lane MD5 step, each on 1..T threads at once (aggregate GB/s).  The many-stream front ends
// (flacenc_encode_many*, csrc/host/stream_writer.cpp) are bound by these and by the CPU quota of the cgroup.
//   g++ -O3 -std=c++17 -Iflac-codec_amd/csrc -Iflac-codec_amd/csrc/host tools/ubench/host_feed_bench.cpp \
//       flac-codec_amd/csrc/host/md5_mb.cpp flac-codec_amd/csrc/host/checksums.cpp flac-codec_amd/csrc/host/cpu_quota.cpp -lpthread -o tools/ubench/bin/host_feed_bench
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "host/md5_mb.h"
using namespace flacenc;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__attribute__((target("avx2"))) static void pack3(const int32_t *s, size_t count, uint8_t *d) {
    const __m256i sh = _mm256_setr_epi8(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1, 0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1,
                                        -1, -1, -1);
    size_t i = 0;
    for (; i + 16 <= count; i += 8) {
        const __m256i v = _mm256_shuffle_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i)), sh);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(d + 3 * i), _mm256_castsi256_si128(v));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(d + 3 * i + 12), _mm256_extracti128_si256(v, 1));
    }
}

template <class F>
static double run_threads(int T, F f) {
    std::vector<std::thread> th;
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            ready++;
            while (!go.load()) std::this_thread::yield();
            f(t);
        });
    while (ready.load() < T) std::this_thread::yield();
    const double t0 = now();
    go = true;
    for (auto &x : th) x.join();
    return now() - t0;
}

int main(int argc, char **argv) {
    const int maxT = argc > 1 ? atoi(argv[1]) : 16;
    const size_t N = 16u << 20;   // samples per thread per pass (64 MB in, 48 MB out)
    const int reps = 4;
    for (int T = 1; T <= maxT; T *= 2) {
        std::vector<std::vector<int32_t>> in(T);
        std::vector<std::vector<uint8_t>> out(T);
        for (int t = 0; t < T; t++) {
            in[t].assign(N, 0x123456);
            out[t].assign(N * 4 + 64, 0);
        }
        double dt = run_threads(T, [&](int t) { for (int r = 0; r < reps; r++) pack3(in[t].data(), N, out[t].data()); });
        printf("T=%2d pack3  : %7.2f Gsamples/s (%6.1f GB/s read + %6.1f written)\n", T, T * reps * (double)N / dt / 1e9,
               T * reps * 4.0 * N / dt / 1e9, T * reps * 3.0 * N / dt / 1e9);
        dt = run_threads(T, [&](int t) { for (int r = 0; r < reps; r++) memcpy(out[t].data(), in[t].data(), N * 4); });
        printf("T=%2d memcpy : %7.2f GB/s copied\n", T, T * reps * 4.0 * N / dt / 1e9);
        dt = run_threads(T, [&](int t) {
            alignas(64) uint32_t st[4][16] = {};
            const uint8_t *ptr[16];
            const size_t per = N * 4 / 16;
            for (int l = 0; l < 16; l++) ptr[l] = reinterpret_cast<const uint8_t *>(in[t].data()) + l * per;
            for (int r = 0; r < reps; r++) md5_blocks_x16(st, ptr, per / 64, 0xFFFF);
        });
        printf("T=%2d md5 x16: %7.2f GB/s hashed (%5.2f per engine thread)\n", T, T * reps * 4.0 * N / dt / 1e9, reps * 4.0 * N / dt / 1e9);
        fflush(stdout);
    }
    return 0;
}
