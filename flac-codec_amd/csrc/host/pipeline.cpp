// pipeline.cpp -- flacgpu_pipeline_*: the full-duplex host -> host batch loop behind one small C-ABI object.
//
// What the reference's `Encoder::encode` loop does per block on the calling thread (encode.rs:558-585: take a block
// of PCM, encode it, append the frame) is here a rotation over `depth` encoder contexts, each with its own HIP stream:
// while batch n's PCM travels up (H2D on its context's stream), batch n - 1's kernels run and batch n - 2's frames
// travel down -- written by k_frame64 straight into the slot's pinned host buffer (flacgpu_encode_packed_async_host),
// so the downward leg is the kernel's own stores, not a copy engine.  No MD5 and no container: callers that want the
// stream bookkeeping use include/flacenc_stream.h; this is the batch boundary of include/flacenc_gpu.h made
// overlap-by-default.  Built from the public entry points only (no HIP headers here).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "flacenc_gpu.h"

struct flacgpu_pipeline {
    struct Slot {
        flacgpu_ctx *ctx = nullptr;
        uint8_t *out = nullptr;      // pinned, flacgpu_packed_cap(ctx) bytes
        size_t cap = 0;
        uint32_t n_frames = 0;
        bool busy = false;
    };
    std::vector<Slot> slots;
    uint32_t head = 0, tail = 0, in_flight = 0;   // submit at head, retire at tail
    bool engine_down = false;   // A/B (FLACGPU_PIPELINE_ENGINE_DOWN=1): frames assembled in HBM, copied down by a copy engine
    std::string error;
};

extern "C" {

int flacgpu_pipeline_create(const flacgpu_options *opts, uint32_t bits_per_sample, uint32_t channels, int device,
                            uint32_t max_frames, uint32_t depth, flacgpu_pipeline **out) {
    if (!opts || !out || depth < 1 || depth > 16 || max_frames == 0) return FLACGPU_ERR_INVALID_ARG;
    flacgpu_pipeline *p = new (std::nothrow) flacgpu_pipeline();
    if (!p) return FLACGPU_ERR_HIP;
    p->slots.resize(depth);
    if (const char *e = getenv("FLACGPU_PIPELINE_ENGINE_DOWN")) p->engine_down = e[0] && e[0] != '0';
    for (auto &s : p->slots) {
        int rc = flacgpu_create(opts, bits_per_sample, channels, device, max_frames, &s.ctx);
        if (rc == FLACGPU_OK) {
            s.cap = flacgpu_packed_cap(s.ctx);
            s.out = static_cast<uint8_t *>(flacgpu_host_alloc(s.cap));
            if (!s.out) rc = FLACGPU_ERR_HIP;
        }
        if (rc != FLACGPU_OK) {
            flacgpu_pipeline_destroy(p);
            return rc;
        }
    }
    *out = p;
    return FLACGPU_OK;
}

void flacgpu_pipeline_destroy(flacgpu_pipeline *p) {
    if (!p) return;
    for (auto &s : p->slots) {
        if (s.ctx) {
            if (s.busy) (void)flacgpu_wait(s.ctx);
            flacgpu_destroy(s.ctx);
        }
        if (s.out) flacgpu_host_free(s.out);
    }
    delete p;
}

uint32_t flacgpu_pipeline_in_flight(const flacgpu_pipeline *p) { return p ? p->in_flight : 0; }
uint32_t flacgpu_pipeline_depth(const flacgpu_pipeline *p) { return p ? (uint32_t)p->slots.size() : 0; }

int flacgpu_pipeline_submit(flacgpu_pipeline *p, const void *pcm, uint32_t bytes_per_sample, uint32_t n_frames,
                            uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate) {
    if (!p || !pcm) return FLACGPU_ERR_INVALID_ARG;
    if (p->in_flight == p->slots.size()) return FLACGPU_ERR_BUSY;   // retire one first
    flacgpu_pipeline::Slot &s = p->slots[p->head];
    const int rc = flacgpu_encode_packed_async_host(s.ctx, static_cast<const uint8_t *>(pcm), bytes_per_sample, n_frames,
                                                    last_frame_len, first_frame_number, sample_rate,
                                                    p->engine_down ? nullptr : s.out, p->engine_down ? 0 : s.cap);
    if (rc != FLACGPU_OK) return rc;
    s.n_frames = n_frames;
    s.busy = true;
    p->head = (p->head + 1) % p->slots.size();
    p->in_flight++;
    return FLACGPU_OK;
}

int flacgpu_pipeline_retire(flacgpu_pipeline *p, const uint8_t **frames, const uint64_t **offsets, uint32_t *n_frames,
                            uint64_t *total) {
    if (!p || p->in_flight == 0) return FLACGPU_ERR_INVALID_ARG;
    flacgpu_pipeline::Slot &s = p->slots[p->tail];
    const uint64_t *off = nullptr;
    uint64_t bytes = 0;
    int rc = flacgpu_frames_ready(s.ctx, &off, &bytes);               // the sizes (and any host re-decision)
    if (rc == FLACGPU_OK) rc = flacgpu_fetch_frames_async(s.ctx, s.out, s.cap);   // nothing to copy when k_frame64 wrote them here
    if (rc == FLACGPU_OK) rc = flacgpu_wait(s.ctx);
    s.busy = false;
    p->tail = (p->tail + 1) % p->slots.size();
    p->in_flight--;
    if (rc != FLACGPU_OK) return rc;
    if (frames) *frames = s.out;
    if (offsets) *offsets = off;
    if (n_frames) *n_frames = s.n_frames;
    if (total) *total = bytes;
    return FLACGPU_OK;
}

}  // extern "C"
