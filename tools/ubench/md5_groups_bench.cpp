#include "host/md5_mb.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace flacenc;
int main() {
    const size_t per = 1 << 20;
    std::vector<uint8_t> buf(64 * per);
    for (size_t i = 0; i < buf.size(); i++) buf[i] = (uint8_t)(i * 2654435761u >> 13);
    for (int G = 1; G <= 4; G++) {
        alignas(64) uint32_t st[4][4][16];
        const uint8_t *ptr[4][16];
        uint32_t mask[4] = {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF};
        Md5 ref[64];
        for (int g = 0; g < G; g++) for (int l = 0; l < 16; l++) { ptr[g][l] = buf.data() + (g * 16 + l) * per; uint32_t w[4]; ref[g*16+l].get_state(w); for (int k=0;k<4;k++) st[g][k][l]=w[k]; }
        const int reps = 8;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) md5_blocks_groups(st, ptr, per / 64, mask, G);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // check against scalar
        int bad = 0;
        for (int g = 0; g < G; g++) for (int l = 0; l < 16; l++) {
            Md5 m; for (int r = 0; r < reps; r++) m.update(buf.data() + (g * 16 + l) * per, per);
            uint32_t w[4]; m.get_state(w);
            for (int k = 0; k < 4; k++) if (w[k] != st[g][k][l]) bad++;
        }
        printf("G=%d: %.2f GB/s (%.3f GB/s per lane)  mismatches %d\n", G, reps * G * 16.0 * per / dt / 1e9, reps * (double)per / dt / 1e9, bad);
    }
}
