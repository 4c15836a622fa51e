// multi_device.cpp -- flacgpu_multi_*: ONE process driving several GPUs behind the C ABI.
//
// The reference's host is one process that owns the whole stream (Encoder::encode, encode.rs:1997-2022; its
// fork-join over a frame's channels, :3964-4010).  What crosses shards of a stream is bookkeeping only: the seek
// points' byte offsets -- a prefix sum of frame sizes (:1999-2003) -- and STREAMINFO's min / max frame size
// (:2414-2436).  So a run of blocks is cut into CONTIGUOUS FRAME RANGES -- batches of <= max_frames frames, dealt to the
// listed devices ("shards") in turn --, each shard runs its batches through its own contexts (upload, kernels and the frames'
// way down overlap: depth batches in flight, k_frame64 storing the frames into the slot's pinned buffer), and every retired
// batch is copied ONCE, from the pinned slot straight to its final place in the caller's `out`: a batch's place is the sum of
// the sizes of the batches before it, published by whoever retires them (Placement) -- dealing batches instead of one long
// range per shard is what makes that sum known in time (r06; r05 staged every shard's bytes in a growing vector and copied
// them again at the end: three host copies per output byte).  The per-shard records {frames, bytes, min_frame, max_frame}
// merge as flac_codec_amd/parallel.py's merge_counters does for the one-process-per-GPU shape.  No data-path collective, no
// peer access: shards never read each other's memory.  Built from the public entry points only (no HIP headers here).
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "flacenc_gpu.h"

namespace {
// One listed device: its contexts, its pinned frame buffers, and a parked host thread that drives them -- created with the
// shard's first flacgpu_multi_encode and kept, bound to the CPUs of the GPU's own NUMA node (its PCI function's
// local_cpulist): the thread that copies frames out of pinned memory and rings doorbells sits next to the root complex
// the GPU hangs off, not on the other socket.
struct Shard {
    int device = 0;
    std::vector<flacgpu_ctx *> ctx;            // contexts in rotation (device-resident batches AND the host path's slots)
    std::vector<uint8_t *> pin;                // the host path: a pinned frame buffer per context (k_frame64 stores into it)
    size_t pin_cap = 0;
    uint32_t next = 0;                         // rotation cursor of flacgpu_multi_encode_device
    flacgpu_ctx *last = nullptr;               // context of the shard's last resident batch
    uint32_t last_frames = 0;
    flacgpu_shard_counters counters{};         // of the last flacgpu_multi_encode call
    int rc = FLACGPU_OK;
    // the parked thread
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, busy = false, stop = false, pinned_near_gpu = false;
    void run() {
        pinned_near_gpu = pin_thread_near(device);
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
            cv.wait(l, [&] { return has_job || stop; });
            if (stop) return;
            has_job = false;
            std::function<void()> f = std::move(job);
            l.unlock();
            try {
                f();
            } catch (...) {   // (nothing may cross the C boundary, least of all on a thread of our own)
                rc = FLACGPU_ERR_HIP;
            }
            l.lock();
            busy = false;
            cv.notify_all();
        }
    }
    static bool pin_thread_near(int device) {
        char list[1024];
        int node = -1;
        if (flacgpu_device_numa_info(device, &node, list, sizeof list) != FLACGPU_OK || !list[0]) return false;
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = 0;
        for (const char *p = list; *p;) {   // "0-63,128-191"
            char *e = nullptr;
            long a = std::strtol(p, &e, 10), b = a;
            if (e == p) break;
            if (*e == '-') b = std::strtol(e + 1, &e, 10);
            for (long c = a; c <= b && c < CPU_SETSIZE; c++, n++) CPU_SET((int)c, &set);
            p = (*e == ',') ? e + 1 : e;
            if (*e && *e != ',') break;
        }
        return n > 0 && sched_setaffinity(0, sizeof set, &set) == 0;
    }
    void start(std::function<void()> f) {
        std::lock_guard<std::mutex> l(mu);
        if (!th.joinable()) th = std::thread([this] { run(); });
        job = std::move(f);
        has_job = busy = true;
        cv.notify_all();
    }
    void join_job() {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return !busy; });
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// Where the batches of one flacgpu_multi_encode call land: batch j's first byte follows the bytes of the batches before it,
// so its place is known as soon as THEIR sizes are -- each shard publishes a batch's size when it retires it, and waits (the
// batches before were submitted earlier, on the other shards) for its own base
struct Placement {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint64_t> total, base;
    std::vector<uint8_t> have;
    size_t next = 0;
    uint64_t running = 0;
    bool failed = false;
    explicit Placement(size_t n) : total(n, 0), base(n, 0), have(n, 0) {}
    void publish(size_t j, uint64_t bytes) {
        std::lock_guard<std::mutex> l(mu);
        total[j] = bytes;
        have[j] = 1;
        while (next < have.size() && have[next]) {
            base[next] = running;
            running += total[next];
            next++;
        }
        cv.notify_all();
    }
    bool wait_base(size_t j, uint64_t *b) {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return next > j || failed; });
        if (next <= j) return false;
        *b = base[j];
        return true;
    }
    void fail() {
        std::lock_guard<std::mutex> l(mu);
        failed = true;
        cv.notify_all();
    }
};
}  // namespace

struct flacgpu_multi {
    flacgpu_options opts{};
    uint32_t bps = 0, channels = 0, max_frames = 0, depth = 0;
    std::deque<Shard> shards;                  // (a Shard holds a mutex and a thread: never moved)
    uint64_t host_bytes_out = 0, host_bytes_copied = 0;   // flacgpu_multi_host_copy_stats: cumulative
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

// (every entry point: nothing may cross the C boundary -- std::bad_alloc of a table, std::system_error of a thread)
#define MULTI_GUARDED(...)               \
    try {                                \
        __VA_ARGS__                      \
    } catch (...) {                      \
        return FLACGPU_ERR_HIP;          \
    }

int flacgpu_multi_create(const flacgpu_options *opts, uint32_t bits_per_sample, uint32_t channels, const int *devices,
                         uint32_t n_devices, uint32_t max_frames, uint32_t depth, flacgpu_multi **out) {
    if (!opts || !out || max_frames == 0 || depth < 1 || depth > 16) return FLACGPU_ERR_INVALID_ARG;
    *out = nullptr;
    MULTI_GUARDED(
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
    )
}

void flacgpu_multi_destroy(flacgpu_multi *m) {
    if (!m) return;
    for (auto &s : m->shards) {
        s.shutdown();
        for (flacgpu_ctx *c : s.ctx) {
            (void)flacgpu_wait(c);
            flacgpu_destroy(c);
        }
        for (uint8_t *p : s.pin) flacgpu_host_free(p);
    }
    delete m;
}

uint32_t flacgpu_multi_shards(const flacgpu_multi *m) { return m ? static_cast<uint32_t>(m->shards.size()) : 0; }
int flacgpu_multi_device_of(const flacgpu_multi *m, uint32_t shard) {
    return (m && shard < m->shards.size()) ? m->shards[shard].device : -1;
}
int flacgpu_multi_host_copy_stats(const flacgpu_multi *m, uint64_t *bytes_out, uint64_t *bytes_copied, uint32_t *threads_near_gpu) {
    if (!m) return FLACGPU_ERR_INVALID_ARG;
    if (bytes_out) *bytes_out = m->host_bytes_out;
    if (bytes_copied) *bytes_copied = m->host_bytes_copied;
    if (threads_near_gpu) {
        *threads_near_gpu = 0;
        for (const auto &s : m->shards) *threads_near_gpu += s.pinned_near_gpu ? 1u : 0u;
    }
    return FLACGPU_OK;
}

namespace {
struct Call {   // one flacgpu_multi_encode
    const uint8_t *pcm;
    uint32_t bytes_per_sample, last_frame_len, sample_rate;
    uint64_t n_frames, first_frame_number;
    uint8_t *out;
    size_t cap;
    uint64_t *offsets;
    size_t n_batches;
    Placement *place;
    std::atomic<uint64_t> copied{0};
};

// Shard k's share of the call: batches k, k + G, k + 2 G, ... through its `depth` slots.
void run_shard(flacgpu_multi *m, Shard &s, uint32_t k, uint32_t G, Call &c) {
    s.rc = FLACGPU_OK;
    s.counters = flacgpu_shard_counters{};
    const uint32_t M = m->max_frames, depth = m->depth;
    while (s.ctx.size() < depth && s.rc == FLACGPU_OK) {   // the shard's other contexts, on first use
        flacgpu_ctx *x = nullptr;
        s.rc = flacgpu_create(&m->opts, m->bps, m->channels, s.device, M, &x);
        if (s.rc == FLACGPU_OK) s.ctx.push_back(x);
    }
    while (s.pin.size() < depth && s.rc == FLACGPU_OK) {
        s.pin_cap = flacgpu_packed_cap(s.ctx[0]);
        uint8_t *p = static_cast<uint8_t *>(flacgpu_host_alloc(s.pin_cap));
        if (!p) s.rc = FLACGPU_ERR_HIP;
        else s.pin.push_back(p);
    }
    if (s.rc != FLACGPU_OK) {
        c.place->fail();
        return;
    }
    const size_t frame_bytes = static_cast<size_t>(m->opts.block_size) * m->channels * c.bytes_per_sample;
    struct InFlight {
        size_t batch;
        uint32_t slot, frames;
    };
    std::deque<InFlight> fl;
    bool any = false;
    auto retire = [&]() -> int {
        const InFlight f = fl.front();
        fl.pop_front();
        flacgpu_ctx *x = s.ctx[f.slot];
        const uint64_t *off = nullptr;
        uint64_t total = 0;
        int rc = flacgpu_frames_ready(x, &off, &total);          // the sizes (and any host re-decision)
        if (rc != FLACGPU_OK) return rc;
        c.place->publish(f.batch, total);                        // ... tell the others where the next batch starts
        for (uint32_t i = 0; i < f.frames; i++) {
            const uint64_t z = off[i + 1] - off[i];
            s.counters.min_frame = any ? std::min(s.counters.min_frame, z) : z;
            s.counters.max_frame = std::max(s.counters.max_frame, z);
            any = true;
        }
        s.counters.frames += f.frames;
        s.counters.bytes += total;
        rc = flacgpu_fetch_frames_async(x, s.pin[f.slot], s.pin_cap);   // (nothing to copy when k_frame64 stored them there)
        if (rc == FLACGPU_OK) rc = flacgpu_wait(x);
        if (rc != FLACGPU_OK) return rc;
        uint64_t base = 0;
        if (!c.place->wait_base(f.batch, &base)) return FLACGPU_ERR_HIP;   // (another shard failed)
        const uint64_t first = (uint64_t)f.batch * M;
        if (c.offsets)
            for (uint32_t i = 0; i < f.frames; i++) c.offsets[first + i] = base + (off[i] - off[0]);
        if (c.out && base + total <= c.cap) {    // the ONE host copy of these bytes: pinned slot -> their place in `out`
            std::memcpy(c.out + base, s.pin[f.slot], total);
            c.copied.fetch_add(total, std::memory_order_relaxed);
        }
        return FLACGPU_OK;
    };
    uint32_t slot = 0;
    for (size_t j = k; j < c.n_batches && s.rc == FLACGPU_OK; j += G) {
        if (fl.size() == depth) s.rc = retire();
        if (s.rc != FLACGPU_OK) break;
        const uint64_t first = (uint64_t)j * M;
        const uint32_t take = static_cast<uint32_t>(std::min<uint64_t>(M, c.n_frames - first));
        const uint32_t ll = (first + take == c.n_frames) ? c.last_frame_len : m->opts.block_size;
        s.rc = flacgpu_encode_packed_async_host(s.ctx[slot], c.pcm + first * frame_bytes, c.bytes_per_sample, take, ll,
                                                c.first_frame_number + first, c.sample_rate, s.pin[slot], s.pin_cap);
        if (s.rc != FLACGPU_OK) break;
        fl.push_back(InFlight{j, slot, take});
        slot = (slot + 1) % depth;
    }
    while (!fl.empty()) {   // drain (after an error too: nothing may stay in flight)
        if (s.rc == FLACGPU_OK) {
            s.rc = retire();
        } else {
            (void)flacgpu_wait(s.ctx[fl.front().slot]);
            fl.pop_front();
        }
    }
    if (s.rc != FLACGPU_OK) c.place->fail();
}
}  // namespace

int flacgpu_multi_encode(flacgpu_multi *m, const void *pcm, uint32_t bytes_per_sample, uint64_t n_frames,
                         uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate, uint8_t *out,
                         size_t cap, uint64_t *offsets, uint64_t *total, flacgpu_shard_counters *per_shard,
                         flacgpu_shard_counters *merged) {
    if (!m || !pcm || n_frames == 0 || last_frame_len == 0 || last_frame_len > m->opts.block_size ||
        !(bytes_per_sample == 4 || bytes_per_sample == (m->bps + 7) / 8))
        return FLACGPU_ERR_INVALID_ARG;
    MULTI_GUARDED(
        const uint32_t G = static_cast<uint32_t>(m->shards.size());
        const size_t n_batches = static_cast<size_t>((n_frames + m->max_frames - 1) / m->max_frames);
        Placement place(n_batches);
        Call c;
        c.pcm = static_cast<const uint8_t *>(pcm);
        c.bytes_per_sample = bytes_per_sample;
        c.last_frame_len = last_frame_len;
        c.sample_rate = sample_rate;
        c.n_frames = n_frames;
        c.first_frame_number = first_frame_number;
        c.out = out;
        c.cap = cap;
        c.offsets = offsets;
        c.n_batches = n_batches;
        c.place = &place;
        // the shards' parked threads drive their devices (the entry points make their context's device current for the
        // duration of a call); shard 0's share runs on the calling thread
        for (uint32_t k = 1; k < G; k++) {
            Shard *sp = &m->shards[k];
            sp->start([m, sp, k, G, &c] { run_shard(m, *sp, k, G, c); });
        }
        try {
            run_shard(m, m->shards[0], 0, G, c);
        } catch (...) {
            m->shards[0].rc = FLACGPU_ERR_HIP;
            place.fail();
        }
        for (uint32_t k = 1; k < G; k++) m->shards[k].join_job();
        for (auto &s : m->shards)
            if (s.rc != FLACGPU_OK) return s.rc;
        std::vector<flacgpu_shard_counters> cs(G);
        for (uint32_t k = 0; k < G; k++) cs[k] = m->shards[k].counters;
        flacgpu_shard_counters all{};
        flacgpu_merge_counters(cs.data(), G, &all, nullptr);
        if (per_shard) std::memcpy(per_shard, cs.data(), sizeof(flacgpu_shard_counters) * G);
        if (merged) *merged = all;
        if (total) *total = all.bytes;
        if (offsets) offsets[n_frames] = all.bytes;
        m->host_bytes_out += out ? all.bytes : 0;
        m->host_bytes_copied += c.copied.load();
        if (out && cap < all.bytes) return FLACGPU_ERR_BUFFER_TOO_SMALL;
        return FLACGPU_OK;
    )
}

int flacgpu_multi_encode_device(flacgpu_multi *m, uint32_t shard, const int32_t *d_pcm, int layout, uint32_t n_frames,
                                uint32_t last_frame_len, uint64_t first_frame_number, uint32_t sample_rate) {
    if (!m || shard >= m->shards.size() || !d_pcm) return FLACGPU_ERR_INVALID_ARG;
    Shard &s = m->shards[shard];
    try {
        while (s.ctx.size() < m->depth) {   // the shard's other contexts, on first use
            flacgpu_ctx *c = nullptr;
            const int rc = flacgpu_create(&m->opts, m->bps, m->channels, s.device, m->max_frames, &c);
            if (rc != FLACGPU_OK) return rc;
            s.ctx.push_back(c);
        }
    } catch (...) {
        return FLACGPU_ERR_HIP;
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
    MULTI_GUARDED(
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
    )
}

}  // extern "C"
