/* c_abi_pipeline.c -- the pipelined batch loop of include/flacenc_gpu.h from plain C: the recipe a binding's
 * `encode_blocks` follows.  Host PCM in pinned memory goes up, finished FLAC frames come
 * back into pinned memory, `depth` batches in flight, no MD5 / container (the caller's stream state, as in the
 * reference's Encoder::encode loop, encode.rs:558-585 / 1997-2022).
 *
 *   gcc -O2 -Iinclude examples/c_abi_pipeline.c -Lflac-codec_amd -lflacenc_amd -Wl,-rpath,flac-codec_amd
 *   ./a.out [batches] [frames_per_batch] [bytes_per_sample: 4 | 3]
 * prints: total frame bytes, FNV-1a hash of all frames in order, frames, and the Msamples/s of the loop.
 * tests/test_gpu_pipeline.py compares the hash with the oracle's frames for the same signal. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "flacenc_gpu.h"

#define BLOCK 4096u
#define CH 2u

/* deterministic 24-bit test signal (restated in the test): 2-pole resonator driven by an LCG; right = 3/4 left + noise */
static void make_signal(int32_t *pcm, size_t frames, uint32_t seed) {
    uint32_t s = seed;
    int64_t y1 = 0, y2 = 0;
    for (size_t i = 0; i < frames; i++) {
        s = s * 1103515245u + 12345u;
        int32_t e = (int32_t)((s >> 10) & 0x3FFFF) - 131072;
        int64_t y = ((58000 * y1 - 29491 * y2) >> 15) + e;
        if (y > 8000000) y = 8000000;
        if (y < -8000000) y = -8000000;
        y2 = y1;
        y1 = y;
        s = s * 1103515245u + 12345u;
        int32_t e2 = (int32_t)((s >> 14) & 0xFFF) - 2048;
        pcm[2 * i] = (int32_t)y;
        pcm[2 * i + 1] = (int32_t)((3 * y) >> 2) + e2;
    }
}

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static uint64_t g_hash = 1469598103934665603ull, g_bytes = 0, g_frames = 0;
/* the caller's side of `retire`: here the frames are only hashed; a writer would append them to its file and keep the
 * sizes for the seek table (flacenc_stream_header rebuilds the metadata from them) */
static int drain_one(flacgpu_pipeline *p, int hash) {
    const uint8_t *frames;
    const uint64_t *off;
    uint32_t n;
    uint64_t total;
    int rc = flacgpu_pipeline_retire(p, &frames, &off, &n, &total);
    if (rc) return rc;
    if (hash)
        for (uint64_t i = 0; i < total; i++) g_hash = (g_hash ^ frames[i]) * 1099511628211ull;
    g_bytes += total;
    g_frames += n;
    return 0;
}

int main(int argc, char **argv) {
    uint32_t batches = argc > 1 ? (uint32_t)atoi(argv[1]) : 8;
    uint32_t fpb = argc > 2 ? (uint32_t)atoi(argv[2]) : 256;
    uint32_t width = argc > 3 ? (uint32_t)atoi(argv[3]) : 4;
    const uint32_t depth = 4, rate = 48000, bps = 24;
    flacgpu_options o;
    memset(&o, 0, sizeof o);
    o.block_size = BLOCK;              /* Options::best(), encode.rs:1646-1656 */
    o.max_partition_order = 6;
    o.max_lpc_order = 12;
    o.mid_side = 1;
    o.exhaustive_channel_correlation = 1;
    o.window_kind = FLACGPU_WINDOW_TUKEY;
    o.window_param = 0.5f;
    flacgpu_pipeline *p = NULL;
    int rc = flacgpu_pipeline_create(&o, bps, CH, -1, fpb, depth, &p);
    if (rc) { fprintf(stderr, "create: %d %s\n", rc, flacgpu_last_error()); return 1; }

    /* one pinned input buffer per slot: a batch's PCM must stay put until the batch has been retired */
    const size_t samples = (size_t)fpb * BLOCK * CH;
    uint8_t *in[4];
    int32_t *tmp = (int32_t *)malloc(samples * sizeof(int32_t));
    for (uint32_t i = 0; i < depth; i++) in[i] = (uint8_t *)flacgpu_host_alloc(samples * 4);
    if (!tmp || !in[0] || !in[1] || !in[2] || !in[3]) return 2;

    double t0 = 0;
    for (int pass = 0; pass < 2; pass++) {   /* pass 0: hashed (and warms the contexts); pass 1: timed */
        g_hash = 1469598103934665603ull; g_bytes = 0; g_frames = 0;
        t0 = now();
        for (uint32_t b = 0; b < batches; b++) {
            if (flacgpu_pipeline_in_flight(p) == depth && (rc = drain_one(p, pass == 0))) goto fail;
            uint8_t *dst = in[b % depth];
            if (pass == 0) {                  /* (the timed pass re-sends the buffers as they are) */
                make_signal(tmp, samples / CH, 1000u + (b % depth));
                if (width == 4) memcpy(dst, tmp, samples * 4);
                else
                    for (size_t i = 0; i < samples; i++) {   /* little-endian 3-byte samples, as FlacByteWriter gets them */
                        dst[3 * i] = (uint8_t)tmp[i]; dst[3 * i + 1] = (uint8_t)(tmp[i] >> 8); dst[3 * i + 2] = (uint8_t)(tmp[i] >> 16);
                    }
            }
            rc = flacgpu_pipeline_submit(p, dst, width, fpb, BLOCK, (uint64_t)b * fpb, rate);
            if (rc) goto fail;
        }
        while (flacgpu_pipeline_in_flight(p))
            if ((rc = drain_one(p, pass == 0))) goto fail;
        if (pass == 0)
            printf("%llu %016llx %llu ", (unsigned long long)g_bytes, (unsigned long long)g_hash, (unsigned long long)g_frames);
    }
    {
        const double dt = now() - t0;
        printf("%.1f\n", (double)batches * samples / dt / 1e6);
    }
    for (uint32_t i = 0; i < depth; i++) flacgpu_host_free(in[i]);
    free(tmp);
    flacgpu_pipeline_destroy(p);
    return 0;
fail:
    fprintf(stderr, "pipeline: %d %s\n", rc, flacgpu_last_error());
    return 1;
}
