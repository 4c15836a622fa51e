// multi_device.cpp -- flacgpu_multi_*: ONE process driving several GPUs behind the C ABI.
//
// The reference's host is one process that owns the whole stream (Encoder::encode, encode.rs:1997-2022; its
// fork-join over a frame's channels, :3964-4010).  What crosses shards of a stream is bookkeeping only: the seek
// points' byte offsets -- a prefix sum of frame sizes (:1999-2003) -- and STREAMINFO's min / max frame size
// (:2414-2436).  So a run of blocks is cut into CONTIGUOUS FRAME RANGES, one per listed device ("shard"), each range
// runs through a pipeline of that device's own contexts (host/pipeline.cpp: upload, kernels and the frames' way down
// overlap), and the host merges {frames, bytes, min_frame, max_frame} and the per-frame sizes exactly as
// flac_codec_amd/parallel.py's merge_counters does for the one-process-per-GPU shape.  No data-path collective, no
// peer access: shards never read each other's memory.  Built from the public entry points only (no HIP headers here).
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "flacenc_gpu.h"

namespace {
struct Shard {
    int device = 0;
    flacgpu_pipeline *pipe = nullptr;          // host -> host batches of this shard (created on first use)
    std::vector<flacgpu_ctx *> ctx;            // device-resident batches: contexts in rotation
    uint32_t next = 0;                         // rotation cursor
    flacgpu_ctx *last = nullptr;               // context of the shard's last resident batch
    uint32_t last_frames = 0;
    // results of the last flacgpu_multi_encode call
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off;                 // local offsets, frames + 1 entries
    flacgpu_shard_counters counters{};
    int rc = FLACGPU_OK;
};
}  // namespace

struct flacgpu_multi {
    flacgpu_options opts{};
    uint32_t bps = 0, channels = 0, max_frames = 0, depth = 0;
    std::vector<Shard> shards;
};

static void count_frames(const uint64_t *off, uint64_t n, flacgpu_shard_counters *c) {
    c->frames = n;
    c->bytes = n ? off[n] - off[0] : 0;
    c->min_frame = c->max_frame = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t s = off[i + 1] - off[i];
        c->min_frame = (i == 0) ? s : std::min(c->min_frame, s);
        c->max_frame = std::max(c->max_frame, s);
    }
}

extern "C" {

int flacgpu_merge_counters(const flacgpu_shard_counters *shards, uint32_t n, flacgpu_shard_counters *merged,
                           uint64_t *shard_byte_offsets) {
    if (!shards || !merged || n == 0) return FLACGPU_ERR_INVALID_ARG;
    flacgpu_shard_counters m{};
    bool any = false;
    for (uint32_t i = 0; i < n; i++) {
        if (shard_byte_offsets) shard_byte_offsets[i] = m.bytes;   // exclusive prefix sum: where shard i's first frame lands
        m.frames += shards[i].frames;
        m.bytes += shards[i].bytes;
        if (shards[i].frames == 0) continue;                       // an idle shard takes no part in min / max
        m.min_frame = any ? std::min(m.min_frame, shards[i].min_frame) : shards[i].min_frame;
        m.max_frame = std::max(m.max_frame, shards[i].max_frame);
        any = true;
    }
    *merged = m;
    return FLACGPU_OK;
}

void flacgpu_shard_range(uint64_t total_frames, uint32_t shards, uint32_t shard, uint64_t *lo, uint64_t *hi) {
    // [k F / G, (k + 1) F / G): the same cut as parallel.shard_range (128-bit products: F can be 2^36 frames)
    const unsigned __int128 f = total_frames;
    if (lo) *lo = static_cast<uint64_t>(f * shard / (shards ? shards : 1));
    if (hi) *hi = static_cast<uint64_t>(f * (shard + 1) / (shards ? shards : 1));
}

int flacgpu_multi_create(const flacgpu_options *opts, uint32_t bits_per_sample, uint32_t channels, const int *devices,
                         uint32_t n_devices, uint32_t max_frames, uint32_t depth, flacgpu_multi **out) {
    if (!opts || !out || max_frames == 0 || depth < 1 || depth > 16) return FLACGPU_ERR_INVALID_ARG;
    *out = nullptr;
    std::vector<int> list;
    if (devices && n_devices) {
        list.assign(devices, devices + n_devices);
    } else {                       // every visible device
        const int n = flacgpu_device_count();
        if (n <= 0) return FLACGPU_ERR_HIP;
        for (int d = 0; d < n; d++) list.push_back(d);
    }
    const int visible = flacgpu_device_count();
    for (int d : list)
        if (d < 0 || d >= visible) return FLACGPU_ERR_INVALID_ARG;
    flacgpu_multi *m = new (std::nothrow) flacgpu_multi();
    if (!m) return FLACGPU_ERR_HIP;
    m->opts = *opts;
    m->bps = bits_per_sample;
    m->channels = channels;
    m->max_frames = max_frames;
    m->depth = depth;
    m->shards.resize(list.size());
    for (size_t i = 0; i < list.size(); i++) m->shards[i].device = list[i];
    // a first context per shard now, so that an unsupported stream shape or a dead device fails here
    for (auto &s : m->shards) {
        flacgpu_ctx *c = nullptr;
        const int rc = flacgpu_create(opts, bits_per_sample, channels, s.device, max_frames, &c);
        if (rc != FLACGPU_OK) {
            flacgpu_multi_destroy(m);
            return rc;
        }
        s.ctx.push_back(c);
    }
    *out = m;
    return FLACGPU_OK;
}

void flacgpu_multi_destroy(flacgpu_multi *m) {
    if (!m) return;
    for (auto &s : m->shards) {
        if (s.pipe) flacgpu_pipeline_destroy(s.pipe);
        for (flacgpu_ctx *c : s.ctx) {
            (void)flacgpu_wait(c);
            flacgpu_destroy(c);
        }
    }
    delete m;
}

uint32_t flacgpu_multi_shards(const flacgpu_multi *m) { return m ? static_cast<uint32_t>(m->shards.size()) : 0; }
int flacgpu_multi_device_of(const flacgpu_multi *m, uint32_t shard) {
    return (m && shard < m->shards.size()) ? m->shards[shard].device : -1;
}

// One shard's range [lo, hi) of the call, through its pipeline in sub-batches of <= max_frames frames.
static void run_shard_body(flacgpu_multi *m, Shard &s, const uint8_t *pcm, uint32_t bytes_per_sample, uint64_t lo, uint64_t hi,
                           uint64_t n_frames, uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate);
// (a shard thread must not let an exception -- std::bad_alloc of its staging vectors -- cross the C boundary)
static void run_shard(flacgpu_multi *m, Shard &s, const uint8_t *pcm, uint32_t bytes_per_sample, uint64_t lo, uint64_t hi,
                      uint64_t n_frames, uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate) {
    try {
        run_shard_body(m, s, pcm, bytes_per_sample, lo, hi, n_frames, last_frame_len, first_frame_number, sample_rate);
    } catch (...) {
        while (s.pipe && flacgpu_pipeline_in_flight(s.pipe)) {
            const uint8_t *fr;
            const uint64_t *off;
            uint32_t nf;
            uint64_t total;
            if (flacgpu_pipeline_retire(s.pipe, &fr, &off, &nf, &total) != FLACGPU_OK) break;
        }
        s.rc = FLACGPU_ERR_HIP;
    }
}
static void run_shard_body(flacgpu_multi *m, Shard &s, const uint8_t *pcm, uint32_t bytes_per_sample, uint64_t lo, uint64_t hi,
                           uint64_t n_frames, uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate) {
    s.rc = FLACGPU_OK;
    s.bytes.clear();
    s.off.assign(1, 0);
    s.counters = flacgpu_shard_counters{};
    if (hi <= lo) return;
    if (!s.pipe) {
        s.rc = flacgpu_pipeline_create(&m->opts, m->bps, m->channels, s.device, m->max_frames, m->depth, &s.pipe);
        if (s.rc != FLACGPU_OK) return;
    }
    const size_t frame_bytes = static_cast<size_t>(m->opts.block_size) * m->channels * bytes_per_sample;
    s.off.reserve(hi - lo + 1);
    uint64_t next = lo;            // next frame to submit
    auto retire = [&]() -> int {
        const uint8_t *fr = nullptr;
        const uint64_t *off = nullptr;
        uint32_t nf = 0;
        uint64_t total = 0;
        const int rc = flacgpu_pipeline_retire(s.pipe, &fr, &off, &nf, &total);
        if (rc != FLACGPU_OK) return rc;
        const uint64_t base = s.bytes.size();
        s.bytes.insert(s.bytes.end(), fr, fr + total);
        for (uint32_t i = 1; i <= nf; i++) s.off.push_back(base + off[i] - off[0]);
        return FLACGPU_OK;
    };
    while (next < hi || flacgpu_pipeline_in_flight(s.pipe)) {
        if (next < hi && flacgpu_pipeline_in_flight(s.pipe) < flacgpu_pipeline_depth(s.pipe)) {
            const uint32_t take = static_cast<uint32_t>(std::min<uint64_t>(m->max_frames, hi - next));
            const uint32_t ll = (next + take == n_frames) ? last_frame_len : m->opts.block_size;
            const int rc = flacgpu_pipeline_submit(s.pipe, pcm + next * frame_bytes, bytes_per_sample, take, ll,
                                                   first_frame_number + next, sample_rate);
            if (rc != FLACGPU_OK) {
                s.rc = rc;
                break;
            }
            next += take;
            continue;
        }
        if (int rc = retire()) {
            s.rc = rc;
            break;
        }
    }
    while (flacgpu_pipeline_in_flight(s.pipe)) {   // (after an error: drain what is still in flight)
        const int rc = retire();
        if (s.rc == FLACGPU_OK) s.rc = rc;
        if (rc != FLACGPU_OK) break;
    }
    if (s.rc == FLACGPU_OK) count_frames(s.off.data(), hi - lo, &s.counters);
}

int flacgpu_multi_encode(flacgpu_multi *m, const void *pcm, uint32_t bytes_per_sample, uint64_t n_frames,
                         uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate, uint8_t *out,
                         size_t cap, uint64_t *offsets, uint64_t *total, flacgpu_shard_counters *per_shard,
                         flacgpu_shard_counters *merged) {
    if (!m || !pcm || n_frames == 0 || last_frame_len == 0 || last_frame_len > m->opts.block_size ||
        !(bytes_per_sample == 4 || bytes_per_sample == (m->bps + 7) / 8))
        return FLACGPU_ERR_INVALID_ARG;
    const uint32_t G = static_cast<uint32_t>(m->shards.size());
    const uint8_t *p = static_cast<const uint8_t *>(pcm);
    // one host thread per shard: each drives its own device's pipeline (the entry points make their context's device
    // current for the duration of the call); shard 0 runs on the calling thread
    std::vector<std::thread> th;
    std::vector<uint64_t> lo(G), hi(G);
    for (uint32_t k = 0; k < G; k++) flacgpu_shard_range(n_frames, G, k, &lo[k], &hi[k]);
    for (uint32_t k = 1; k < G; k++)
        th.emplace_back(run_shard, m, std::ref(m->shards[k]), p, bytes_per_sample, lo[k], hi[k], n_frames, last_frame_len,
                        first_frame_number, sample_rate);
    run_shard(m, m->shards[0], p, bytes_per_sample, lo[0], hi[0], n_frames, last_frame_len, first_frame_number,
              sample_rate);
    for (auto &t : th) t.join();
    for (auto &s : m->shards)
        if (s.rc != FLACGPU_OK) return s.rc;
    // the merge: shard byte offsets (exclusive prefix sum), totals, min / max frame size
    std::vector<flacgpu_shard_counters> cs(G);
    std::vector<uint64_t> base(G);
    for (uint32_t k = 0; k < G; k++) cs[k] = m->shards[k].counters;
    flacgpu_shard_counters all{};
    flacgpu_merge_counters(cs.data(), G, &all, base.data());
    if (per_shard) std::memcpy(per_shard, cs.data(), sizeof(flacgpu_shard_counters) * G);
    if (merged) *merged = all;
    if (total) *total = all.bytes;
    if (offsets) {
        for (uint32_t k = 0; k < G; k++)
            for (uint64_t i = 0; i + lo[k] < hi[k]; i++) offsets[lo[k] + i] = base[k] + m->shards[k].off[i];
        offsets[n_frames] = all.bytes;
    }
    if (out) {
        if (cap < all.bytes) return FLACGPU_ERR_BUFFER_TOO_SMALL;
        for (uint32_t k = 0; k < G; k++)
            if (!m->shards[k].bytes.empty()) std::memcpy(out + base[k], m->shards[k].bytes.data(), m->shards[k].bytes.size());
    }
    return FLACGPU_OK;
}

int flacgpu_multi_encode_device(flacgpu_multi *m, uint32_t shard, const int32_t *d_pcm, int layout, uint32_t n_frames,
                                uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate) {
    if (!m || shard >= m->shards.size() || !d_pcm) return FLACGPU_ERR_INVALID_ARG;
    Shard &s = m->shards[shard];
    while (s.ctx.size() < m->depth) {   // the shard's other contexts, on first use
        flacgpu_ctx *c = nullptr;
        const int rc = flacgpu_create(&m->opts, m->bps, m->channels, s.device, m->max_frames, &c);
        if (rc != FLACGPU_OK) return rc;
        s.ctx.push_back(c);
    }
    flacgpu_ctx *c = s.ctx[s.next];
    s.next = (s.next + 1) % s.ctx.size();
    const int rc = flacgpu_encode_device(c, d_pcm, layout, n_frames, last_frame_len, first_frame_number, sample_rate,
                                         nullptr);
    if (rc == FLACGPU_OK) {
        s.last = c;
        s.last_frames = n_frames;
    }
    return rc;
}

int flacgpu_multi_wait(flacgpu_multi *m) {
    if (!m) return FLACGPU_ERR_INVALID_ARG;
    int rc = FLACGPU_OK;
    for (auto &s : m->shards)
        for (flacgpu_ctx *c : s.ctx) {
            const int r = flacgpu_wait(c);
            if (rc == FLACGPU_OK) rc = r;
        }
    return rc;
}

flacgpu_ctx *flacgpu_multi_last_context(flacgpu_multi *m, uint32_t shard) {
    return (m && shard < m->shards.size()) ? m->shards[shard].last : nullptr;
}

int flacgpu_multi_counters(flacgpu_multi *m, flacgpu_shard_counters *per_shard, flacgpu_shard_counters *merged) {
    if (!m) return FLACGPU_ERR_INVALID_ARG;
    const uint32_t G = static_cast<uint32_t>(m->shards.size());
    std::vector<flacgpu_shard_counters> cs(G);
    for (uint32_t k = 0; k < G; k++) {
        Shard &s = m->shards[k];
        if (!s.last) continue;   // a shard that got no batch: idle
        std::vector<uint64_t> off(static_cast<size_t>(s.last_frames) + 1);
        uint64_t total = 0;
        const int rc = flacgpu_fetch_frames(s.last, nullptr, 0, off.data(), &total);   // sizes only (resolves order ties first)
        if (rc != FLACGPU_OK && rc != FLACGPU_ERR_BUFFER_TOO_SMALL) return rc;
        count_frames(off.data(), s.last_frames, &cs[k]);
    }
    if (per_shard) std::memcpy(per_shard, cs.data(), sizeof(flacgpu_shard_counters) * G);
    if (merged) return flacgpu_merge_counters(cs.data(), G, merged, nullptr);
    return FLACGPU_OK;
}

}  // extern "C"
