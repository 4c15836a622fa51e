// Throughput of the 16-lane MD5 step (csrc/host/md5_mb.cpp) against the scalar chain on this host:
//   g++ -O2 -std=c++17 -Iflac-codec_amd/csrc -Iflac-codec_amd/csrc/host tools/ubench/md5_mb_bench.cpp flac-codec_amd/csrc/host/md5_mb.cpp flac-codec_amd/csrc/host/checksums.cpp flac-codec_amd/csrc/host/cpu_quota.cpp -lpthread -o /tmp/md5_mb_bench
#include "host/md5_mb.h"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace flacenc;
int main() {
    const size_t per = 16 << 20;
    std::vector<uint8_t> buf(16 * per, 7);
    alignas(64) uint32_t st[4][16] = {};
    const uint8_t *ptr[16];
    for (int l = 0; l < 16; l++) ptr[l] = buf.data() + l * per;
    for (uint32_t mask : {0xFFFFu, 0x00FFu, 0x000Fu, 0x0003u}) {
        auto t0 = std::chrono::steady_clock::now();
        md5_blocks_x16(st, ptr, per / 64, mask);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int lanes = __builtin_popcount(mask);
        printf("lanes %2d: %.2f GB/s useful (%.2f GB/s per lane)\n", lanes, lanes * per / dt / 1e9, per / dt / 1e9);
    }
    Md5 m;
    auto t0 = std::chrono::steady_clock::now();
    m.update(buf.data(), 64 << 20);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("scalar: %.2f GB/s\n", (64 << 20) / dt / 1e9);
}
