/* c_abi_demo.c -- the drop-in boundary used from plain C (what a cgo / Rust `extern "C"` binding
 * does): encode a synthetic stereo signal through include/flacenc_stream.h and print the size and
 * an FNV-1a hash of the .flac bytes.  tests/test_gpu_c_abi.py builds this with gcc, runs it on the
 * GPU box and compares the hash with the oracle's stream for the same signal.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -Lflac-codec_amd -lflacenc_amd -Wl,-rpath,flac-codec_amd
 *   ./a.out [pcm_frames] [best|default|fast]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flacenc_stream.h"

/* deterministic test signal: a 2-pole resonator driven by an LCG, right = 3/4 left + its own noise;
 * tests/test_gpu_c_abi.py restates these few lines in Python */
static void make_signal(int32_t *pcm, size_t frames) {
    uint32_t s = 12345u;
    int64_t y1 = 0, y2 = 0;
    for (size_t i = 0; i < frames; i++) {
        s = s * 1103515245u + 12345u;
        int32_t e = (int32_t)((s >> 16) & 0x3FFF) - 8192;
        int64_t y = ((58000 * y1 - 29491 * y2) >> 15) + e;
        if (y > 30000) y = 30000;
        if (y < -30000) y = -30000;
        y2 = y1;
        y1 = y;
        s = s * 1103515245u + 12345u;
        int32_t e2 = (int32_t)((s >> 16) & 0x7FF) - 1024;
        pcm[2 * i] = (int32_t)y;
        pcm[2 * i + 1] = (int32_t)((3 * y) >> 2) + e2;
    }
}

int main(int argc, char **argv) {
    size_t frames = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : 100000;
    const char *preset = argc > 2 ? argv[2] : "best";
    int32_t *pcm = (int32_t *)malloc(frames * 2 * sizeof(int32_t));
    if (!pcm) return 2;
    make_signal(pcm, frames);

    flacenc_options o;
    if (!strcmp(preset, "fast")) flacenc_options_fast(&o);
    else if (!strcmp(preset, "default")) flacenc_options_default(&o);
    else flacenc_options_best(&o);

    flacenc_writer *w = NULL;
    int rc = flacenc_sample_writer_new(&o, 44100, 16, 2, 1, frames * 2, NULL, &w);
    if (rc) { fprintf(stderr, "new: %d\n", rc); return 1; }
    /* two writes of odd sizes, like a caller streaming a file */
    size_t first = (frames / 3) * 2 + 2;
    rc = flacenc_write_samples(w, pcm, first);
    if (!rc) rc = flacenc_write_samples(w, pcm + first, frames * 2 - first);
    if (!rc) rc = flacenc_finalize(w);
    if (rc) { fprintf(stderr, "encode: %d\n", rc); return 1; }
    size_t len = 0;
    const uint8_t *data = flacenc_writer_data(w, &len);
    uint64_t h = 1469598103934665603ull; /* FNV-1a 64 */
    for (size_t i = 0; i < len; i++) h = (h ^ data[i]) * 1099511628211ull;
    flacenc_stats st;
    flacenc_writer_stats(w, &st);
    printf("%zu %016llx %llu\n", len, (unsigned long long)h, (unsigned long long)st.frames);
    flacenc_writer_free(w);
    free(pcm);
    return 0;
}
