// stream_writer.cpp -- host driver: the reference's Encoder<W> + FlacSampleWriter /
// FlacByteWriter / FlacChannelWriter / FlacStreamWriter (encode.rs:48-1290, 1853-2160),
// batching PCM blocks into the gfx950 analysis (include/flacenc_gpu.h) and doing what the
// reference keeps sequential: MD5, frame headers, Rice bit-packing, CRC, seek points,
// metadata.  C ABI: include/flacenc_stream.h.
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <tuple>
#include <mutex>
#include <vector>

#include "bitsink.h"
#include "checksums.h"
#include "md5_mb.h"
#include "host_internal.h"
#include "flacenc_gpu.h"
#include "flacenc_stream.h"
#include "frame_pack.h"

namespace {

using flacenc::Md5;
using flacenc::Md5Lane;
using flacenc::Md5Pool;

thread_local std::string g_err;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

constexpr uint64_t kMaxSamples = 68719476736ull;         // Encoder::MAX_SAMPLES, encode.rs:1880
constexpr uint32_t kMaxFrameSize = (1u << 24) - 1;       // Streaminfo::MAX_FRAME_SIZE
constexpr size_t kMaxSeekPoints = (1u << 24) / 18;       // SeekTable::MAX_POINTS
constexpr uint64_t kMaxFrameNumber = (1ull << 36) - 1;   // FrameNumber::MAX_FRAME_NUMBER
constexpr size_t kDirectFrames = 64;   // blocks in one write call from which they are encoded without staging

// ---- output sink (W: Write + Seek) --------------------------------------------------
struct Sink {
    flacenc_sink cb{};
    bool memory = true;
    std::vector<uint8_t> mem;
    size_t mem_pos = 0;
    // caller-owned buffer (flacenc_encode_many): no growth, no page faults of our own
    uint8_t *fixed = nullptr;
    size_t fixed_cap = 0, fixed_len = 0;

    int write(const uint8_t *p, size_t n) {
        if (fixed) {
            if (mem_pos + n > fixed_cap) return FLACENC_ERR_IO;
            std::memcpy(fixed + mem_pos, p, n);
            mem_pos += n;
            fixed_len = std::max(fixed_len, mem_pos);
            return 0;
        }
        if (!memory) return cb.write(cb.user, p, n) ? FLACENC_ERR_IO : 0;
        if (mem_pos + n > mem.size()) mem.resize(mem_pos + n);
        std::memcpy(mem.data() + mem_pos, p, n);
        mem_pos += n;
        return 0;
    }
    int seek(uint64_t off) {
        if (fixed) {
            mem_pos = static_cast<size_t>(off);
            return 0;
        }
        if (!memory) return (cb.seek && cb.seek(cb.user, off) == 0) ? 0 : FLACENC_ERR_IO;
        mem_pos = static_cast<size_t>(off);
        return 0;
    }
    uint64_t start() const { return (memory || fixed) ? 0 : cb.start; }
};

struct SeekPoint {
    uint64_t sample_offset, byte_offset;
    uint16_t frame_samples;
    bool defined;
};

// SeekTableInterval::filter, encode.rs:1338-1358
std::vector<SeekPoint> filter_points(const flacenc_options &o, uint32_t sample_rate,
                                     const std::vector<SeekPoint> &in) {
    std::vector<SeekPoint> out;
    if (o.seektable_mode == FLACENC_SEEKTABLE_SECONDS) {
        const uint64_t nth = static_cast<uint32_t>((o.seektable_value & 0xFFu) * sample_rate);
        uint64_t offset = 0;
        for (const auto &p : in)
            if (offset >= p.sample_offset && offset < p.sample_offset + p.frame_samples) {
                offset += nth;
                out.push_back(p);
            }
    } else if (o.seektable_mode == FLACENC_SEEKTABLE_FRAMES) {
        const size_t step = o.seektable_value ? o.seektable_value : 1;
        for (size_t i = 0; i < in.size(); i += step) out.push_back(in[i]);
    }
    return out;
}

// ---- metadata serialisation (metadata/mod.rs:257-266, 904-960, 1740-1760, 1826-1831,
//      2010-2036, 2118-2139) ----------------------------------------------------------
struct StreamInfo {
    uint32_t min_block = 0, max_block = 0, min_frame = 0, max_frame = 0;
    uint32_t sample_rate = 0, channels = 0, bps = 0;
    uint64_t total_samples = 0;  // 0 = None
    uint8_t md5[16] = {0};
};

void put_be(std::vector<uint8_t> &v, uint64_t x, int bytes) {
    for (int i = bytes - 1; i >= 0; i--) v.push_back(static_cast<uint8_t>(x >> (8 * i)));
}
void put_block_header(std::vector<uint8_t> &v, bool last, uint8_t type, uint32_t size) {
    v.push_back(static_cast<uint8_t>((last ? 0x80 : 0) | type));
    put_be(v, size, 3);
}

struct MetaLayout {
    bool vorbis = false;              // a VORBIS_COMMENT block is present (sorts first)
    std::vector<uint8_t> vorbis_body;
    bool seektable = false;           // a SEEKTABLE block is present
    bool seektable_after_padding = false;  // inserted at finalize => pushed last
    std::vector<SeekPoint> points;
    bool padding = false;
    uint32_t padding_size = 0;
};

std::vector<uint8_t> build_metadata(const StreamInfo &si, const MetaLayout &m) {
    std::vector<uint8_t> v;
    v.insert(v.end(), {'f', 'L', 'a', 'C'});
    int remaining = (m.seektable ? 1 : 0) + (m.padding ? 1 : 0) + (m.vorbis ? 1 : 0);
    put_block_header(v, remaining == 0, 0, 34);
    put_be(v, si.min_block, 2);
    put_be(v, si.max_block, 2);
    put_be(v, si.min_frame, 3);
    put_be(v, si.max_frame, 3);
    // 20 bits rate | 3 bits channels-1 | 5 bits bps-1 | 36 bits total samples
    uint64_t packed = (static_cast<uint64_t>(si.sample_rate) << 44) |
                      (static_cast<uint64_t>(si.channels - 1) << 41) |
                      (static_cast<uint64_t>(si.bps - 1) << 36) | (si.total_samples & 0xFFFFFFFFFull);
    put_be(v, packed, 8);
    v.insert(v.end(), si.md5, si.md5 + 16);
    auto emit_seektable = [&]() {
        remaining--;
        put_block_header(v, remaining == 0, 3, static_cast<uint32_t>(m.points.size() * 18));
        for (const auto &p : m.points) {
            if (p.defined) {
                put_be(v, p.sample_offset, 8);
                put_be(v, p.byte_offset, 8);
                put_be(v, p.frame_samples, 2);
            } else {  // SeekPoint::Placeholder
                put_be(v, ~0ull, 8);
                put_be(v, 0, 8);
                put_be(v, 0, 2);
            }
        }
    };
    auto emit_padding = [&]() {
        remaining--;
        put_block_header(v, remaining == 0, 1, m.padding_size);
        v.insert(v.end(), m.padding_size, 0);
    };
    if (m.vorbis) {  // VorbisComment::to_writer, metadata/mod.rs:2512-2536
        remaining--;
        put_block_header(v, remaining == 0, 4, static_cast<uint32_t>(m.vorbis_body.size()));
        v.insert(v.end(), m.vorbis_body.begin(), m.vorbis_body.end());
    }
    // block order after Encoder::new's sort (encode.rs:1944-1951): VORBIS_COMMENT < SEEKTABLE < PADDING;
    // a SEEKTABLE created at finalize is pushed behind PADDING (metadata/mod.rs:4425-4441)
    if (m.seektable && !m.seektable_after_padding) emit_seektable();
    if (m.padding) emit_padding();
    if (m.seektable && m.seektable_after_padding) emit_seektable();
    return v;
}

int options_error(const flacenc_options &o) {
    if (o.block_size < 16 || o.block_size > 65535) return FLACENC_ERR_INVALID_BLOCK_SIZE;
    if (o.max_lpc_order > 32) return FLACENC_ERR_INVALID_LPC_ORDER;
    if (o.max_partition_order > 15) return FLACENC_ERR_INVALID_MAX_PARTITIONS;
    if (o.padding < 0 || o.padding >= (1 << 24)) return FLACENC_ERR_EXCESSIVE_PADDING;
    return 0;
}

flacgpu_options gpu_options(const flacenc_options &o, uint32_t block_size) {
    flacgpu_options g{};
    g.block_size = block_size;
    g.max_partition_order = o.max_partition_order;
    g.max_lpc_order = o.max_lpc_order;
    g.mid_side = o.mid_side;
    g.exhaustive_channel_correlation = o.exhaustive_channel_correlation;
    g.window_kind = o.window_kind;
    g.window_param = o.window_param;
    return g;
}

int map_gpu_error(int rc) {
    g_err = std::string("gpu: ") + flacgpu_last_error();
    return rc == FLACGPU_ERR_UNSUPPORTED ? FLACENC_ERR_UNSUPPORTED : FLACENC_ERR_GPU;
}

// Packs the frames of one analysed batch in parallel (frames are independent once their
// frame numbers and sizes are known) and appends them to `out`.
struct PackedBatch {
    std::vector<uint8_t> bytes;
    std::vector<size_t> offsets;  // n_frames + 1
};

int pack_batch(uint32_t sample_rate, uint32_t bps, uint32_t channels, uint64_t first_frame_number,
               uint32_t n_frames, uint32_t row_stride, const flacgpu_frame_plan *plans,
               const flacgpu_subframe_plan *subs, const int32_t *rows, unsigned threads,
               PackedBatch &out) {
    out.offsets.assign(n_frames + 1, 0);
    for (uint32_t f = 0; f < n_frames; f++) {
        flacenc::FrameParams fp{sample_rate, bps, channels, first_frame_number + f};
        out.offsets[f + 1] = out.offsets[f] + flacenc::frame_size(fp, plans[f]);
    }
    out.bytes.resize(out.offsets[n_frames] + 16);
    std::vector<int> status(std::max(1u, threads), 0);
    auto work = [&](unsigned t, unsigned nt) {
        // contiguous frame ranges per thread; the 32-bit stores of BitSink may spill up to 3
        // bytes into the next frame's region, so frames are packed into a scratch buffer
        std::vector<uint8_t> scratch;
        uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(n_frames) * t / nt);
        uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(n_frames) * (t + 1) / nt);
        for (uint32_t f = lo; f < hi; f++) {
            flacenc::FrameParams fp{sample_rate, bps, channels, first_frame_number + f};
            const size_t sz = out.offsets[f + 1] - out.offsets[f];
            scratch.resize(sz + 8);
            size_t got = flacenc::pack_frame(fp, plans[f], subs + static_cast<size_t>(f) * channels,
                                             rows + static_cast<size_t>(f) * channels * row_stride,
                                             row_stride, scratch.data(), scratch.size());
            if (got != sz) {
                status[t] = 1;
                return;
            }
            std::memcpy(out.bytes.data() + out.offsets[f], scratch.data(), sz);
        }
    };
    unsigned nt = std::max(1u, std::min<unsigned>(threads, n_frames));
    if (nt == 1) {
        work(0, 1);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; t++) pool.emplace_back(work, t, nt);
        for (auto &th : pool) th.join();
    }
    for (int s : status)
        if (s) {
            g_err = "internal: packed frame size differs from its decision record";
            return FLACENC_ERR_GPU;
        }
    out.bytes.resize(out.offsets[n_frames]);
    return 0;
}

}  // namespace

// =====================================================================================
// Encoder<W>, encode.rs:1853-2110
// =====================================================================================
// Idle analysis lanes, keyed by everything flacgpu_create depends on.  A lane is what one batch in
// flight needs: an analysis context (a dozen hipMalloc calls to create, as many hipFree calls to
// destroy, each a device-wide synchronisation: with many short streams -- a music library -- they
// dominate) and two pinned staging buffers (PCM at stream width in, frame bytes out; pinning memory
// is slower still).  Writers return their lanes here; at most kPoolCap idle lanes are kept (the rest
// are destroyed), the process exit reclaims.
struct CtxKey {
    flacgpu_options g;
    uint32_t bps, channels, batch;
    int device;   // resolved ordinal, never -1: a pooled context must not change GPUs
    bool operator==(const CtxKey &o) const {
        return std::memcmp(&g, &o.g, sizeof g) == 0 && bps == o.bps && channels == o.channels &&
               batch == o.batch && device == o.device;
    }
};
struct Lane {
    flacgpu_ctx *gpu = nullptr;
    uint8_t *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0;
    int reserve_in(size_t n) {
        if (n <= pin_in_cap) return 0;
        flacgpu_host_free(pin_in);
        pin_in = static_cast<uint8_t *>(flacgpu_host_alloc(n));
        pin_in_cap = pin_in ? n : 0;
        return pin_in ? 0 : FLACENC_ERR_GPU;
    }
    int reserve_out(size_t n) {
        if (n <= pin_out_cap) return 0;
        flacgpu_host_free(pin_out);
        n += n / 8;   // a little head room: the next batch of the stream is about as large
        pin_out = static_cast<uint8_t *>(flacgpu_host_alloc(n));
        pin_out_cap = pin_out ? n : 0;
        return pin_out ? 0 : FLACENC_ERR_GPU;
    }
    void destroy() {
        if (gpu) flacgpu_destroy(gpu);
        flacgpu_host_free(pin_in);
        flacgpu_host_free(pin_out);
        delete this;
    }
};
struct LanePool {
    static constexpr size_t kPoolCap = 192;
    std::mutex mu;
    std::vector<std::pair<CtxKey, Lane *>> idle;
    Lane *take(const CtxKey &k) {
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < idle.size(); i++) {
            if (idle[i].first == k) {
                Lane *l = idle[i].second;
                idle.erase(idle.begin() + static_cast<ptrdiff_t>(i));
                return l;
            }
        }
        return nullptr;
    }
    // a full pool gives up its OLDEST idle lane (least recently returned) for the one coming back: a process that has
    // moved on to other stream shapes keeps pooling the shapes it uses now (the r02 pool destroyed the newcomer instead,
    // and every later stream of a new shape paid a context creation)
    void give(const CtxKey &k, Lane *l) {
        Lane *evict = nullptr;
        {
            std::lock_guard<std::mutex> lock(mu);
            if (idle.size() >= kPoolCap) {
                evict = idle.front().second;
                idle.erase(idle.begin());
            }
            idle.emplace_back(k, l);
        }
        if (evict) evict->destroy();
    }
};
LanePool &lane_pool() {
    static LanePool *p = new LanePool();  // intentionally leaked: no HIP calls during static destruction
    return *p;
}

// The stream MD5 is one serial chain over the PCM bytes (encode.rs:571, 1292-1318): it runs on a
// thread of its own, fed in stream order with the very buffers that are being uploaded.
// The stream's MD5 chain, off the writer's thread: a worker thread of its own, or (use_pool) a lane of the
// shared multi-stream engines (md5_mb.h) when many writers run side by side.
struct Md5Worker {
    Md5 *md5 = nullptr;
    bool use_pool = false, pool_tried = false;
    Md5Lane *lane = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<const uint8_t *, size_t>> q;
    uint64_t pushed = 0, done = 0;
    bool stop = false;
    double busy_ms = 0;
    uint64_t push(const uint8_t *p, size_t n) {
        if (use_pool && !pool_tried) {   // nullptr when every engine is full: the private thread then
            pool_tried = true;
            lane = Md5Pool::get().attach(md5);
        }
        if (lane) {
            pushed = Md5Pool::get().push(lane, p, n);
            return pushed;
        }
        std::unique_lock<std::mutex> lock(mu);
        if (!th.joinable()) th = std::thread([this] { run(); });
        q.emplace_back(p, n);
        const uint64_t ticket = ++pushed;
        cv.notify_all();
        return ticket;
    }
    void wait(uint64_t ticket) {
        if (lane) {
            Md5Pool::get().wait(lane, ticket);
            return;
        }
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return done >= ticket; });
    }
    double total_busy_ms() {
        if (lane) return Md5Pool::get().busy_ms(lane);
        std::lock_guard<std::mutex> lock(mu);
        return busy_ms;
    }
    void run() {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv.wait(lock, [&] { return stop || !q.empty(); });
            if (q.empty()) return;
            auto job = q.front();
            q.pop_front();
            lock.unlock();
            const double t0 = now_ms();
            md5->update(job.first, job.second);
            const double dt = now_ms() - t0;
            lock.lock();
            busy_ms += dt;
            done++;
            cv.notify_all();
        }
    }
    ~Md5Worker() {
        if (lane) Md5Pool::get().detach(lane);
        {
            std::lock_guard<std::mutex> lock(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// interleaved int32 samples -> little-endian samples of `width` bytes (update_md5's byte string,
// encode.rs:1292-1318; byteorder.rs:60-72).  x86-64 hosts with AVX2 take the shuffle versions
// (picked once at run time); everything else the portable loops.
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target("avx2"))) size_t pack_le3_avx2(const int32_t *s, size_t count, uint8_t *d) {
    // 8 samples -> 24 bytes: per 128-bit half, bytes 0-2, 4-6, 8-10, 12-14 move to the front
    const __m256i sh = _mm256_setr_epi8(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1,
                                        0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1);
    size_t i = 0;
    // the 16-byte stores overrun each 12-byte group by 4 bytes: stop one group early
    for (; i + 16 <= count; i += 8) {
        const __m256i v = _mm256_shuffle_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i)), sh);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(d + 3 * i), _mm256_castsi256_si128(v));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(d + 3 * i + 12), _mm256_extracti128_si256(v, 1));
    }
    return i;
}
__attribute__((target("avx2"))) size_t pack_le2_avx2(const int32_t *s, size_t count, uint8_t *d) {
    const __m256i sh = _mm256_setr_epi8(0, 1, 4, 5, 8, 9, 12, 13, -1, -1, -1, -1, -1, -1, -1, -1,
                                        0, 1, 4, 5, 8, 9, 12, 13, -1, -1, -1, -1, -1, -1, -1, -1);
    size_t i = 0;
    for (; i + 8 <= count; i += 8) {
        const __m256i v = _mm256_shuffle_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i)), sh);
        _mm_storel_epi64(reinterpret_cast<__m128i *>(d + 2 * i), _mm256_castsi256_si128(v));
        _mm_storel_epi64(reinterpret_cast<__m128i *>(d + 2 * i + 8), _mm256_extracti128_si256(v, 1));
    }
    return i;
}
bool have_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
#else
size_t pack_le3_avx2(const int32_t *, size_t, uint8_t *) { return 0; }
size_t pack_le2_avx2(const int32_t *, size_t, uint8_t *) { return 0; }
bool have_avx2() { return false; }
#endif

void pack_le(const int32_t *s, size_t count, unsigned width, uint8_t *d) {
    switch (width) {
    case 1:
        for (size_t i = 0; i < count; i++) d[i] = static_cast<uint8_t>(s[i]);
        break;
    case 2: {
        size_t i = have_avx2() ? pack_le2_avx2(s, count, d) : 0;
        for (; i < count; i++) {
            const uint32_t v = static_cast<uint32_t>(s[i]);
            d[2 * i] = static_cast<uint8_t>(v);
            d[2 * i + 1] = static_cast<uint8_t>(v >> 8);
        }
        break;
    }
    case 3: {
        size_t i = have_avx2() ? pack_le3_avx2(s, count, d) : 0;
        for (; i < count; i++) {
            const uint32_t v = static_cast<uint32_t>(s[i]);
            d[3 * i] = static_cast<uint8_t>(v);
            d[3 * i + 1] = static_cast<uint8_t>(v >> 8);
            d[3 * i + 2] = static_cast<uint8_t>(v >> 16);
        }
        break;
    }
    default: std::memcpy(d, s, count * 4); break;
    }
}

struct flacenc_writer {
    enum Kind { SAMPLE, BYTE, CHANNEL } kind = SAMPLE;
    flacenc_options o{};
    Sink sink;
    StreamInfo si;
    MetaLayout meta;
    size_t metadata_len = 0;
    uint64_t frame_number = 0, samples_written = 0, byte_count = 0;  // Counter::count
    uint64_t samples_submitted = 0;   // samples_written + the batches still in flight
    std::vector<SeekPoint> seekpoints;
    Md5 md5;
    bool finalized = false;
    flacgpu_ctx *gpu = nullptr;   // = lanes[0]->gpu (the synchronous paths use this one)
    // batches in flight (device-side frame assembly): lanes taken from the pool, the oldest retired first
    struct InFlight {
        Lane *lane;
        uint32_t n_frames, last_len;
        uint64_t md5_ticket;
    };
    std::vector<Lane *> lanes;
    std::deque<InFlight> inflight;
    size_t next_lane = 0;
    unsigned depth = 2;           // batches in flight
    unsigned upload_width = 4;    // bytes per sample across PCIe
    Md5Worker md5_worker;
    int failed = 0;               // first error of an asynchronous batch
    uint32_t batch_frames = 256;
    unsigned pack_threads = 1;
    // backlog of interleaved samples not yet cut into blocks (FlacSampleWriter::sample_buf)
    std::vector<int32_t> backlog;
    // byte writer state
    bool big_endian = false;
    unsigned bytes_per_sample = 2;
    std::vector<uint8_t> byte_backlog;
    // per-batch scratch
    std::vector<flacgpu_frame_plan> plans;
    std::vector<flacgpu_subframe_plan> subs;
    std::vector<int32_t> rows;
    std::vector<uint8_t> md5_bytes;
    std::vector<uint8_t> outbuf;  // frames fetched from the device
    flacenc_stats stats{};

    bool header_only = false;
    CtxKey gpu_key{};
    ~flacenc_writer() {
        for (auto &f : inflight) (void)flacgpu_wait(f.lane->gpu);   // nothing may still write into a pooled lane
        // ... and nothing may still READ one: queued MD5 runs point at the lanes' pinned staging buffers (abandoned or
        // failed streams reach this point with runs pending), which the next owner overwrites or the pool frees
        md5_worker.wait(md5_worker.pushed);
        for (Lane *l : lanes) lane_pool().give(gpu_key, l);
    }

    void use_fixed_sink(uint8_t *buf, size_t cap) {
        sink.fixed = buf;
        sink.fixed_cap = cap;
    }

    // one more lane (context + staging) for this stream shape, from the pool or new
    int add_lane() {
        Lane *l = lane_pool().take(gpu_key);
        if (!l) {
            l = new Lane();
            int rc = flacgpu_create(&gpu_key.g, gpu_key.bps, gpu_key.channels, gpu_key.device, batch_frames, &l->gpu);
            if (rc) {
                delete l;
                return map_gpu_error(rc);
            }
        }
        // many writers side by side (shared MD5 engines): waiting threads sleep instead of spinning
        (void)flacgpu_set_tuning(l->gpu, FLACGPU_TUNE_BLOCKING_WAIT, o.shared_md5 ? 1 : 0);
        lanes.push_back(l);
        return 0;
    }

    // Encoder::new, encode.rs:1882-1980
    int init(const flacenc_options &opts, uint32_t rate, uint32_t bps, uint32_t channels,
             bool has_total, uint64_t total_pcm_frames, const flacenc_sink *s) {
        o = opts;
        if (int e = options_error(o)) return e;
        if (rate >= 1048576u) return FLACENC_ERR_INVALID_SAMPLE_RATE;
        if (channels < 1 || channels > 8) return FLACENC_ERR_EXCESSIVE_CHANNELS;
        if (has_total && total_pcm_frames >= kMaxSamples) return FLACENC_ERR_EXCESSIVE_TOTAL_SAMPLES;
        if (s) {
            sink.memory = false;
            sink.cb = *s;
        }
        si.min_block = si.max_block = o.block_size;
        si.sample_rate = rate;
        si.channels = channels;
        si.bps = bps;
        si.total_samples = has_total ? total_pcm_frames : 0;
        bytes_per_sample = (bps + 7) / 8;
        meta.padding = o.padding > 0;
        meta.padding_size = static_cast<uint32_t>(o.padding);
        if (o.n_comment_fields || o.vendor_string) {
            auto put_le32 = [&](uint32_t x) {
                for (int i = 0; i < 4; i++) meta.vorbis_body.push_back(static_cast<uint8_t>(x >> (8 * i)));
            };
            const std::string vendor = o.vendor_string ? o.vendor_string : "flac-codec 1.3.2";
            put_le32(static_cast<uint32_t>(vendor.size()));
            meta.vorbis_body.insert(meta.vorbis_body.end(), vendor.begin(), vendor.end());
            put_le32(o.n_comment_fields);
            for (uint32_t i = 0; i < o.n_comment_fields; i++) {
                const std::string f = o.comment_fields[i];
                put_le32(static_cast<uint32_t>(f.size()));
                meta.vorbis_body.insert(meta.vorbis_body.end(), f.begin(), f.end());
            }
            meta.vorbis = true;
        }
        // the writer keeps no pointers into caller memory
        o.vendor_string = nullptr;
        o.comment_fields = nullptr;
        if (has_total && o.seektable_mode != FLACENC_SEEKTABLE_NONE) {
            // placeholder SEEKTABLE, encode.rs:1920-1939 + EncoderSeekPoint::placeholders :2131
            std::vector<SeekPoint> all;
            for (uint64_t off = 0; off < total_pcm_frames; off += o.block_size) {
                uint64_t rem = total_pcm_frames - off;
                uint16_t fs = static_cast<uint16_t>(rem > 65535 ? o.block_size
                                                                 : std::min<uint64_t>(rem, o.block_size));
                all.push_back({off, 0, fs, false});
            }
            meta.points = filter_points(o, rate, all);
            if (meta.points.size() > kMaxSeekPoints) meta.points.resize(kMaxSeekPoints);
            for (auto &p : meta.points) p.defined = false;
            meta.seektable = true;
        }
        std::vector<uint8_t> hdr = build_metadata(si, meta);
        metadata_len = hdr.size();
        if (sink.memory && !sink.fixed && has_total && !header_only)   // one allocation instead of doublings
            sink.mem.reserve(hdr.size() + static_cast<size_t>(total_pcm_frames) * channels * bytes_per_sample * 3 / 4);
        if (int e = sink.write(hdr.data(), hdr.size())) return e;
        if (header_only) return 0;   // flacenc_stream_header: bookkeeping without an analysis lane
        batch_frames = o.batch_frames ? o.batch_frames : 256;
        unsigned hw = std::thread::hardware_concurrency();
        pack_threads = o.pack_threads ? o.pack_threads : std::max(1u, std::min(hw, 16u));
        flacgpu_options g = gpu_options(o, o.block_size);
        std::memset(&gpu_key, 0, sizeof gpu_key);
        gpu_key.g = g;
        gpu_key.bps = bps;
        gpu_key.channels = channels;
        gpu_key.batch = batch_frames;
        gpu_key.device = o.device >= 0 ? o.device : flacgpu_current_device();
        if (int rc = add_lane()) return rc;
        gpu = lanes[0]->gpu;
        depth = o.pipeline_depth ? std::min<uint32_t>(o.pipeline_depth, 4u) : 2u;
        upload_width = flacgpu_packed_input_supported(gpu, bytes_per_sample) ? bytes_per_sample : 4u;
        md5_worker.md5 = &md5;
        md5_worker.use_pool = o.shared_md5 != 0 && !getenv("FLACENC_NO_SHARED_MD5");
        return 0;
    }

    // update_md5, encode.rs:1292-1318: little-endian, ceil(bps/8) bytes per sample
    // MD5 over the samples as little-endian bytes_per_sample-byte integers (encode.rs:571), converted
    // through a small fixed buffer (a whole-call buffer costs as many page faults as the PCM itself)
    void md5_samples(const int32_t *s, size_t count) {
        double t0 = now_ms();
        constexpr size_t CHUNK = 1 << 16;  // samples
        if (md5_bytes.size() < CHUNK * 4) md5_bytes.resize(CHUNK * 4);
        uint8_t *d = md5_bytes.data();
        for (size_t base = 0; base < count; base += CHUNK) {
            const size_t m = std::min(CHUNK, count - base);
            const int32_t *q = s + base;
            switch (bytes_per_sample) {
            case 1: for (size_t i = 0; i < m; i++) d[i] = static_cast<uint8_t>(q[i]); break;
            case 2:
                for (size_t i = 0; i < m; i++) {
                    uint32_t v = static_cast<uint32_t>(q[i]);
                    d[2 * i] = static_cast<uint8_t>(v);
                    d[2 * i + 1] = static_cast<uint8_t>(v >> 8);
                }
                break;
            case 3:
                for (size_t i = 0; i < m; i++) {
                    uint32_t v = static_cast<uint32_t>(q[i]);
                    d[3 * i] = static_cast<uint8_t>(v);
                    d[3 * i + 1] = static_cast<uint8_t>(v >> 8);
                    d[3 * i + 2] = static_cast<uint8_t>(v >> 16);
                }
                break;
            default: std::memcpy(d, q, m * 4); break;
            }
            md5.update(d, m * bytes_per_sample);
        }
        stats.md5_ms += now_ms() - t0;
    }

    // Encoder::encode for a run of blocks (encode.rs:1997-2022 + encode_frame's tail
    // :2408-2436), `n_frames` blocks of block_size, the last one `last_len` long.
    int encode_blocks(const int32_t *interleaved, uint32_t n_frames, uint32_t last_len) {
        const uint32_t B = o.block_size, C = si.channels;
        if (o.host_pack) {
            plans.resize(n_frames);
            subs.resize(static_cast<size_t>(n_frames) * C);
            rows.resize(static_cast<size_t>(n_frames) * C * B);
        }
        // ExcessiveTotalSamples is raised BEFORE the offending frame is encoded (:2006-2011)
        uint32_t usable = n_frames;
        int deferred = 0;
        if (si.total_samples) {
            uint64_t w = samples_written;
            for (uint32_t f = 0; f < n_frames; f++) {
                w += (f + 1 == n_frames) ? last_len : B;
                if (w > si.total_samples) {
                    usable = f;
                    deferred = FLACENC_ERR_EXCESSIVE_TOTAL_SAMPLES;
                    break;
                }
            }
        }
        if (usable) {
            const uint32_t ll = (usable == n_frames) ? last_len : B;
            if (frame_number + usable - 1 > kMaxFrameNumber) return FLACENC_ERR_EXCESSIVE_FRAME_NUMBER;
            PackedBatch pb;
            size_t dev_total = 0;
            double t0 = now_ms();
            if (o.host_pack) {
                int rc = flacgpu_analyze(gpu, interleaved, FLACGPU_LAYOUT_INTERLEAVED, usable, ll,
                                         plans.data(), subs.data(), rows.data());
                stats.gpu_ms += now_ms() - t0;
                if (rc) return map_gpu_error(rc);
                t0 = now_ms();
                if (int e = pack_batch(si.sample_rate, si.bps, C, frame_number, usable, B, plans.data(),
                                       subs.data(), rows.data(), pack_threads, pb))
                    return e;
                stats.pack_ms += now_ms() - t0;
            } else {
                // frames assembled on the device: only the finished bytes cross PCIe
                std::vector<uint64_t> off(usable + 1);
                uint64_t total = 0;
                // first the sizes, then a buffer of exactly that size (grown, reused by every batch)
                int rc = flacgpu_encode_frames(gpu, interleaved, FLACGPU_LAYOUT_INTERLEAVED, usable, ll,
                                               frame_number, si.sample_rate, nullptr, 0, off.data(), &total);
                if (rc == FLACGPU_ERR_BUFFER_TOO_SMALL) {
                    if (outbuf.size() < total) outbuf.resize(total);
                    rc = flacgpu_fetch_frames(gpu, outbuf.data(), outbuf.size(), off.data(), &total);
                }
                stats.gpu_ms += now_ms() - t0;
                if (rc) return map_gpu_error(rc);
                pb.offsets.assign(off.begin(), off.end());
                dev_total = total;
            }
            for (uint32_t f = 0; f < usable; f++) {
                const uint32_t n = (f + 1 == usable) ? ll : B;
                seekpoints.push_back({samples_written, byte_count + pb.offsets[f], static_cast<uint16_t>(n), true});
                samples_written += n;
                const uint32_t size = static_cast<uint32_t>(pb.offsets[f + 1] - pb.offsets[f]);
                if (size != 0 && size < kMaxFrameSize) {  // :2414-2436
                    si.min_frame = si.min_frame ? std::min(si.min_frame, size) : size;
                    si.max_frame = std::max(si.max_frame, size);
                }
            }
            frame_number += usable;
            const uint8_t *obytes = o.host_pack ? pb.bytes.data() : outbuf.data();
            const size_t olen = o.host_pack ? pb.bytes.size() : dev_total;
            byte_count += olen;
            if (int e = sink.write(obytes, olen)) return e;
        }
        if (deferred) {
            // the reference pushes the seekpoint and bumps samples_written before failing
            seekpoints.push_back({samples_written, byte_count, static_cast<uint16_t>(B), true});
            samples_written += (usable + 1 == n_frames) ? last_len : B;
        }
        return deferred;
    }

    // ---- device-side frame assembly, pipelined: Encoder::encode for a run of blocks (encode.rs:
    // 1997-2022) is split into SUBMIT (stage the PCM at stream width in pinned memory, queue upload +
    // kernels, hand the same bytes to the MD5 thread) and RETIRE (sizes, copy of the bytes, seek
    // points / STREAMINFO bookkeeping and the sink write, encode.rs:1999-2003, 2414-2436), with up
    // to `depth` batches in flight on their own contexts and streams.
    int submit_blocks(const int32_t *interleaved, uint32_t n_frames, uint32_t last_len) {
        if (failed) return failed;
        const uint32_t B = o.block_size, C = si.channels;
        // ExcessiveTotalSamples is raised BEFORE the offending frame is encoded (:2006-2011)
        uint32_t usable = n_frames;
        int deferred = 0;
        if (si.total_samples) {
            uint64_t w = samples_submitted;
            for (uint32_t f = 0; f < n_frames; f++) {
                w += (f + 1 == n_frames) ? last_len : B;
                if (w > si.total_samples) {
                    usable = f;
                    deferred = FLACENC_ERR_EXCESSIVE_TOTAL_SAMPLES;
                    break;
                }
            }
        }
        const size_t all_samples = (static_cast<size_t>(n_frames - 1) * B + last_len) * C;
        if (usable) {
            const uint32_t ll = (usable == n_frames) ? last_len : B;
            if (frame_number + usable - 1 > kMaxFrameNumber) return FLACENC_ERR_EXCESSIVE_FRAME_NUMBER;
            while (inflight.size() >= depth)
                if (int rc = retire_oldest()) return rc;
            // lanes are used round-robin and retired in order: lane (n mod depth) is free again
            const size_t idx = next_lane;
            next_lane = (next_lane + 1) % depth;
            while (lanes.size() <= idx)
                if (int rc = add_lane()) return rc;
            Lane *lane = lanes[idx];
            const size_t count = (static_cast<size_t>(usable - 1) * B + ll) * C;
            const double t0 = now_ms();
            if (int rc = lane->reserve_in(static_cast<size_t>(batch_frames) * B * C * 4 + 64)) return rc;
            // MD5 of the whole call's samples (the frame that trips ExcessiveTotalSamples included:
            // the reference hashes before it encodes, encode.rs:571-577)
            uint64_t ticket;
            if (upload_width == bytes_per_sample && usable == n_frames) {
                // staged and handed to the MD5 chain piece by piece (runs of whole 64-byte blocks): the hash of a
                // large batch starts after its first piece, not after the whole batch has been packed -- in a burst
                // of many streams the chain's latency (one AVX-512 lane: ~0.5 GB/s) is what bounds the call
                const size_t frame_bytes = static_cast<size_t>(B) * C * bytes_per_sample;
                size_t piece_frames = 32;
                while (piece_frames < usable && (piece_frames * frame_bytes) % 64) piece_frames *= 2;
                if ((piece_frames * frame_bytes) % 64) piece_frames = usable;
                ticket = 0;
                for (size_t f0 = 0; f0 < usable; f0 += piece_frames) {
                    const size_t f1 = std::min<size_t>(usable, f0 + piece_frames);
                    const size_t s0 = f0 * B * C, s1 = (f1 == usable) ? count : f1 * B * C;
                    pack_le(interleaved + s0, s1 - s0, upload_width, lane->pin_in + s0 * upload_width);
                    ticket = md5_worker.push(lane->pin_in + s0 * bytes_per_sample, (s1 - s0) * bytes_per_sample);
                }
                stats.pack_ms += now_ms() - t0;
            } else {
                pack_le(interleaved, count, upload_width, lane->pin_in);
                stats.pack_ms += now_ms() - t0;
                md5_worker.wait(md5_worker.pushed);   // keep the chain in stream order
                md5_samples(interleaved, all_samples);
                ticket = md5_worker.pushed;
            }
            const double t1 = now_ms();
            // the frames come back by themselves: k_frame64 writes them into the lane's pinned buffer (which must
            // then hold the worst case of a batch -- not pinned for shapes where that is hundreds of megabytes)
            const size_t worst = flacgpu_packed_cap(lane->gpu);
            if (worst <= (size_t(64) << 20))
                if (int rc = lane->reserve_out(worst)) return rc;
            int rc = flacgpu_encode_packed_async_host(lane->gpu, lane->pin_in, upload_width, usable, ll, frame_number,
                                                      si.sample_rate, lane->pin_out, lane->pin_out_cap);
            stats.gpu_ms += now_ms() - t1;
            if (rc) return map_gpu_error(rc);
            inflight.push_back({lane, usable, ll, ticket});
            frame_number += usable;
            samples_submitted += static_cast<uint64_t>(usable - 1) * B + ll;
        } else {
            md5_worker.wait(md5_worker.pushed);
            md5_samples(interleaved, all_samples);
        }
        if (deferred) {
            if (int rc = retire_all()) return rc;
            // the reference pushes the seekpoint and bumps samples_written before failing
            seekpoints.push_back({samples_written, byte_count, static_cast<uint16_t>(B), true});
            samples_written += (usable + 1 == n_frames) ? last_len : B;
            samples_submitted = samples_written;
        }
        return deferred;
    }

    int retire_oldest() {
        InFlight f = inflight.front();
        inflight.pop_front();
        const uint32_t B = o.block_size;
        const double t0 = now_ms();
        const uint64_t *off = nullptr;
        uint64_t total = 0;
        int rc = flacgpu_frames_ready(f.lane->gpu, &off, &total);
        if (!rc) rc = f.lane->reserve_out(total + 64) ? FLACGPU_ERR_HIP : 0;
        if (!rc) rc = flacgpu_fetch_frames_async(f.lane->gpu, f.lane->pin_out, f.lane->pin_out_cap);
        if (!rc) rc = flacgpu_wait(f.lane->gpu);
        stats.gpu_ms += now_ms() - t0;
        md5_worker.wait(f.md5_ticket);   // the staging buffer is free again only after its hash
        if (rc) {
            failed = map_gpu_error(rc);
            return failed;
        }
        for (uint32_t k = 0; k < f.n_frames; k++) {
            const uint32_t n = (k + 1 == f.n_frames) ? f.last_len : B;
            seekpoints.push_back({samples_written, byte_count + off[k], static_cast<uint16_t>(n), true});
            samples_written += n;
            const uint32_t size = static_cast<uint32_t>(off[k + 1] - off[k]);
            if (size != 0 && size < kMaxFrameSize) {  // :2414-2436
                si.min_frame = si.min_frame ? std::min(si.min_frame, size) : size;
                si.max_frame = std::max(si.max_frame, size);
            }
        }
        byte_count += total;
        if (int e = sink.write(f.lane->pin_out, total)) {
            failed = e;
            return e;
        }
        return 0;
    }
    int retire_all() {
        while (!inflight.empty())
            if (int rc = retire_oldest()) return rc;
        return failed;
    }

    // Cut whole blocks out of `data` (count samples), batch by batch; returns the samples consumed
    // through *consumed_out: whole batches, or every whole block at the final flush.
    int process(const int32_t *data, size_t count, bool final_flush, size_t *consumed_out) {
        const size_t frame_samples = static_cast<size_t>(o.block_size) * si.channels;
        size_t consumed = 0;
        int rc = 0;
        size_t whole_all = count / frame_samples;
        if (!final_flush) whole_all -= whole_all % batch_frames;
        if (!o.host_pack) {   // batches in flight; the MD5 follows the staging buffers
            while (rc == 0 && consumed < whole_all * frame_samples) {
                const size_t whole = whole_all - consumed / frame_samples;
                const uint32_t take = static_cast<uint32_t>(std::min<size_t>(whole, batch_frames));
                rc = submit_blocks(data + consumed, take, o.block_size);
                consumed += take * frame_samples;
            }
            *consumed_out = consumed;
            return rc;
        }
        // The stream MD5 is one serial chain over the PCM (encode.rs:571): it runs on its own
        // host thread over the samples of this call while the GPU batches are in flight
        std::future<void> md5_job;
        if (whole_all)
            md5_job = std::async(std::launch::async, [this, data, n = whole_all * frame_samples]() {
                md5_samples(data, n);
            });
        while (rc == 0 && consumed < whole_all * frame_samples) {
            size_t whole = whole_all - consumed / frame_samples;
            uint32_t take = static_cast<uint32_t>(std::min<size_t>(whole, batch_frames));
            rc = encode_blocks(data + consumed, take, o.block_size);
            consumed += take * frame_samples;
        }
        if (md5_job.valid()) md5_job.get();
        *consumed_out = consumed;
        return rc;
    }

    // cut whole blocks out of the backlog
    int drain(bool final_flush) {
        size_t consumed = 0;
        int rc = process(backlog.data(), backlog.size(), final_flush, &consumed);
        if (consumed) backlog.erase(backlog.begin(), backlog.begin() + static_cast<ptrdiff_t>(consumed));
        return rc;
    }

    // FlacSampleWriter::write (encode.rs:558-585) without staging the caller's samples: once the
    // backlog has been topped up to a whole batch and drained, whole batches are encoded straight
    // from the caller's buffer and only the tail (< one batch) is kept
    int write_direct(const int32_t *samples, size_t count) {
        const size_t frame_samples = static_cast<size_t>(o.block_size) * si.channels;
        const size_t batch_samples = static_cast<size_t>(batch_frames) * frame_samples;
        // a call that brings many blocks at once is encoded from the caller's buffer right away, in
        // batches of at most batch_frames (the last one smaller); small writes accumulate to a batch
        const bool large = !o.host_pack && count >= kDirectFrames * frame_samples;
        if (!backlog.empty()) {
            size_t room;
            if (large) room = (frame_samples - backlog.size() % frame_samples) % frame_samples;  // finish the open block
            else room = backlog.size() < batch_samples ? batch_samples - backlog.size() : 0;
            const size_t take = std::min(room, count);
            backlog.insert(backlog.end(), samples, samples + take);
            samples += take;
            count -= take;
            if (int rc = drain(large)) return rc;
        }
        if (backlog.empty() && (large || count >= batch_samples)) {
            size_t consumed = 0;
            if (int rc = process(samples, count, large, &consumed)) return rc;
            samples += consumed;
            count -= consumed;
        }
        backlog.insert(backlog.end(), samples, samples + count);
        return drain(false);
    }

    // FlacSampleWriter::finalize_inner, encode.rs:588-611, then Encoder::finalize_inner :2024
    int finalize() {
        if (finalized) return 0;
        finalized = true;
        // FlacByteWriter keeps raw bytes; bytes that do not make a whole sample are dropped by
        // the truncation to whole PCM frames (encode.rs:263-265)
        const bool had_partial_sample = !byte_backlog.empty();
        byte_backlog.clear();
        int rc = drain(true);
        if (rc) return rc;
        if (backlog.empty() && had_partial_sample) {
            g_err = "final partial block holds less than one PCM frame (the reference panics)";
            return FLACENC_ERR_UNSUPPORTED;
        }
        if (!backlog.empty()) {
            size_t usable = backlog.size() - backlog.size() % si.channels;
            if (usable == 0) {
                // Frame with zero PCM frames: `chunks_exact(0)` panics in the reference
                // (audio.rs:177 via encode.rs:2020); surfaced as an error here
                g_err = "final partial block holds less than one PCM frame (the reference panics)";
                return FLACENC_ERR_UNSUPPORTED;
            }
            if (o.host_pack) {
                md5_samples(backlog.data(), usable);
                rc = encode_blocks(backlog.data(), 1, static_cast<uint32_t>(usable / si.channels));
            } else {
                rc = submit_blocks(backlog.data(), 1, static_cast<uint32_t>(usable / si.channels));
            }
            backlog.clear();
            if (rc) return rc;
        }
        if (int e = retire_all()) return e;
        md5_worker.wait(md5_worker.pushed);
        stats.md5_ms += md5_worker.total_busy_ms();
        return finalize_encoder();
    }

    int finalize_encoder() {
        // SEEKTABLE, encode.rs:2029-2076
        if (o.seektable_mode != FLACENC_SEEKTABLE_NONE) {
            std::vector<SeekPoint> enc = filter_points(o, si.sample_rate, seekpoints);
            if (meta.seektable) {
                const size_t n = meta.points.size();
                for (size_t i = 0; i < n; i++) {
                    if (i < enc.size()) meta.points[i] = enc[i];
                    else meta.points[i] = SeekPoint{0, 0, 0, false};
                }
            } else if (meta.padding) {
                if (enc.size() > kMaxSeekPoints) {
                    g_err = "more seek points than a SEEKTABLE holds (the reference panics)";
                    return FLACENC_ERR_UNSUPPORTED;
                }
                const uint64_t st_total = static_cast<uint64_t>(enc.size()) * 18 + 4;
                if (enc.size() * 18 < (1u << 24) && meta.padding_size >= st_total) {
                    meta.padding_size -= static_cast<uint32_t>(st_total);
                    meta.points = enc;
                    meta.seektable = true;
                    meta.seektable_after_padding = true;
                }
            }
        }
        // total samples, encode.rs:2079-2097
        if (si.total_samples) {
            if (si.total_samples != samples_written) return FLACENC_ERR_SAMPLE_COUNT_MISMATCH;
        } else {
            if (samples_written >= kMaxSamples) return FLACENC_ERR_EXCESSIVE_TOTAL_SAMPLES;
            if (samples_written == 0) return FLACENC_ERR_NO_SAMPLES;
            si.total_samples = samples_written;
        }
        if (!header_only) md5.digest(si.md5);
        std::vector<uint8_t> hdr = build_metadata(si, meta);
        if (hdr.size() != metadata_len) {
            g_err = "internal: metadata size changed at finalize";
            return FLACENC_ERR_IO;
        }
        if (int e = sink.seek(sink.start())) return e;
        if (int e = sink.write(hdr.data(), hdr.size())) return e;
        if (sink.fixed) sink.mem_pos = sink.fixed_len;
        else if (sink.memory) sink.mem_pos = sink.mem.size();
        stats.frames = frame_number;
        stats.samples_per_channel = samples_written;
        stats.bytes_written = metadata_len + byte_count;
        stats.min_frame_size = si.min_frame;
        stats.max_frame_size = si.max_frame;
        std::memcpy(stats.md5, si.md5, 16);
        return 0;
    }

    // Endianness::bytes_to_le + Frame::fill_from_buf (byteorder.rs, audio.rs:149-188)
    void append_bytes_as_samples(const uint8_t *p, size_t nbytes) {
        const unsigned b = bytes_per_sample;
        const size_t count = nbytes / b;
        const size_t base = backlog.size();
        backlog.resize(base + count);
        int32_t *d = backlog.data() + base;
        for (size_t i = 0; i < count; i++) {
            uint32_t v = 0;
            if (big_endian)
                for (unsigned k = 0; k < b; k++) v = (v << 8) | p[i * b + k];
            else
                for (unsigned k = 0; k < b; k++) v |= static_cast<uint32_t>(p[i * b + k]) << (8 * k);
            const unsigned sh = 32 - 8 * b;
            d[i] = static_cast<int32_t>(v << sh) >> sh;  // sign-extend
        }
    }
};

namespace {
// Parked helper threads for flacenc_encode_many: run(k, fn) has k helpers execute fn() concurrently with the caller
// (which runs it too) and returns when all of them are done.  Threads are created on demand and never exit (a
// detached, intentionally leaked pool: no joins during static destruction).
struct WorkerPool {
    // One run(): the work, and its completion state.  Shared (not on the caller's stack): a helper that has just
    // reported completion may still be inside its call when run() returns (ADVICE r03 -- the mutex, the condition
    // variable and the closure used to live in run()'s frame, and the last helper decremented the counter before it
    // took the lock: run() could return, and its frame die, between those two steps).
    struct Job {
        std::function<void()> fn;
        std::mutex m;
        std::condition_variable cv;
        unsigned left = 0;      // helpers that have not finished fn() yet; guarded by m
    };
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<std::shared_ptr<Job>> tasks;
    unsigned idle = 0, total = 0;
    static WorkerPool &get() {
        static WorkerPool *p = new WorkerPool();
        return *p;
    }
    void worker() {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            idle++;
            cv_work.wait(lock, [&] { return !tasks.empty(); });
            idle--;
            std::shared_ptr<Job> job = std::move(tasks.front());
            tasks.pop_front();
            lock.unlock();
            job->fn();
            {   // the whole hand-off under the job's mutex; `job` keeps it alive past run()'s return
                std::lock_guard<std::mutex> l(job->m);
                if (--job->left == 0) job->cv.notify_all();
            }
            job.reset();
            lock.lock();
        }
    }
    template <class F>
    void run(unsigned helpers, F &fn) {
        auto job = std::make_shared<Job>();
        job->fn = [&fn] { fn(); };     // fn lives in the caller's frame: every call of it ends before `left` reaches 0
        job->left = helpers;
        {
            std::lock_guard<std::mutex> lock(mu);
            for (unsigned i = 0; i < helpers; i++) tasks.push_back(job);
            const unsigned need = helpers > idle ? helpers - idle : 0;
            for (unsigned i = 0; i < need && total < 1024; i++, total++) std::thread([this] { worker(); }).detach();
        }
        cv_work.notify_all();
        fn();
        std::unique_lock<std::mutex> l(job->m);
        job->cv.wait(l, [&] { return job->left == 0; });
    }
};
}  // namespace

extern "C" {

const char *flacenc_last_error(void) { return g_err.c_str(); }

void flacenc_options_default(flacenc_options *o) {  // encode.rs:1376-1408
    std::memset(o, 0, sizeof *o);
    o->block_size = 4096;
    o->max_partition_order = 5;
    o->max_lpc_order = 8;
    o->mid_side = 1;
    o->exhaustive_channel_correlation = 1;
    o->window_kind = FLACGPU_WINDOW_TUKEY;
    o->window_param = 0.5f;
    o->padding = 4096;
    o->seektable_mode = FLACENC_SEEKTABLE_SECONDS;
    o->seektable_value = 10;
    o->device = -1;
}
void flacenc_options_fast(flacenc_options *o) {  // encode.rs:1635-1644
    flacenc_options_default(o);
    o->block_size = 1152;
    o->mid_side = 0;
    o->max_partition_order = 3;
    o->max_lpc_order = 0;
    o->exhaustive_channel_correlation = 0;
}
void flacenc_options_best(flacenc_options *o) {  // encode.rs:1649-1657
    flacenc_options_default(o);
    o->block_size = 4096;
    o->mid_side = 1;
    o->max_partition_order = 6;
    o->max_lpc_order = 12;
}
int flacenc_options_validate(const flacenc_options *o) { return o ? options_error(*o) : FLACENC_ERR_INVALID_ARG; }

static int new_writer(flacenc_writer::Kind kind, const flacenc_options *opts, uint32_t rate,
                      uint32_t bps, uint32_t channels, bool has_total, uint64_t total_pcm_frames,
                      bool big_endian, const flacenc_sink *sink, flacenc_writer **out) {
    std::unique_ptr<flacenc_writer> w(new flacenc_writer());
    w->kind = kind;
    w->big_endian = big_endian;
    int rc = w->init(*opts, rate, bps, channels, has_total, total_pcm_frames, sink);
    if (rc) return rc;
    *out = w.release();
    return 0;
}

int flacenc_sample_writer_new(const flacenc_options *opts, uint32_t rate, uint32_t bps,
                              uint32_t channels, int has_total, uint64_t total_samples,
                              const flacenc_sink *sink, flacenc_writer **out) {
    if (!opts || !out) return FLACENC_ERR_INVALID_ARG;
    *out = nullptr;
    if (bps < 1 || bps > 32) return FLACENC_ERR_INVALID_BITS_PER_SAMPLE;  // encode.rs:495
    uint64_t pcm_frames = 0;
    if (has_total) {  // encode.rs:513-519
        if (channels == 0 || total_samples % channels) return FLACENC_ERR_SAMPLES_NOT_DIVISIBLE_BY_CHANNELS;
        pcm_frames = total_samples / channels;
        if (pcm_frames == 0) return FLACENC_ERR_INVALID_TOTAL_SAMPLES;
    }
    return new_writer(flacenc_writer::SAMPLE, opts, rate, bps, channels, has_total, pcm_frames, false, sink, out);
}

int flacenc_byte_writer_new(const flacenc_options *opts, uint32_t rate, uint32_t bps,
                            uint32_t channels, int has_total, uint64_t total_bytes, int big_endian,
                            const flacenc_sink *sink, flacenc_writer **out) {
    if (!opts || !out) return FLACENC_ERR_INVALID_ARG;
    *out = nullptr;
    if (bps < 1 || bps > 32) return FLACENC_ERR_INVALID_BITS_PER_SAMPLE;
    uint64_t pcm_frames = 0;
    if (has_total) {  // encode.rs:171-178
        const uint64_t bytes_per_sample = (bps + 7) / 8;
        if (channels == 0 || total_bytes % channels || (total_bytes / channels) % bytes_per_sample)
            return FLACENC_ERR_SAMPLES_NOT_DIVISIBLE_BY_CHANNELS;
        pcm_frames = total_bytes / channels / bytes_per_sample;
        if (pcm_frames == 0) return FLACENC_ERR_INVALID_TOTAL_BYTES;
    }
    return new_writer(flacenc_writer::BYTE, opts, rate, bps, channels, has_total, pcm_frames,
                      big_endian != 0, sink, out);
}

int flacenc_channel_writer_new(const flacenc_options *opts, uint32_t rate, uint32_t bps,
                               uint32_t channels, int has_total, uint64_t total_samples,
                               const flacenc_sink *sink, flacenc_writer **out) {
    if (!opts || !out) return FLACENC_ERR_INVALID_ARG;
    *out = nullptr;
    if (bps < 1 || bps > 32) return FLACENC_ERR_INVALID_BITS_PER_SAMPLE;
    // total_samples.and_then(NonZero::new): Some(0) behaves like None (encode.rs:797)
    bool ht = has_total && total_samples != 0;
    return new_writer(flacenc_writer::CHANNEL, opts, rate, bps, channels, ht, total_samples, false, sink, out);
}

int flacenc_write_samples(flacenc_writer *w, const int32_t *samples, size_t count) {
    if (!w || (!samples && count)) return FLACENC_ERR_INVALID_ARG;
    if (w->finalized) return FLACENC_ERR_FINALIZED;
    return w->write_direct(samples, count);
}

int flacenc_write_bytes(flacenc_writer *w, const uint8_t *bytes, size_t count) {
    if (!w || (!bytes && count)) return FLACENC_ERR_INVALID_ARG;
    if (w->finalized) return FLACENC_ERR_FINALIZED;
    // keep bytes that do not yet make a whole sample; everything else becomes samples
    w->byte_backlog.insert(w->byte_backlog.end(), bytes, bytes + count);
    const size_t usable = w->byte_backlog.size() - w->byte_backlog.size() % w->bytes_per_sample;
    w->append_bytes_as_samples(w->byte_backlog.data(), usable);
    w->byte_backlog.erase(w->byte_backlog.begin(), w->byte_backlog.begin() + static_cast<ptrdiff_t>(usable));
    return w->drain(false);
}

int flacenc_write_channels(flacenc_writer *w, const int32_t *const *channels, uint32_t n_channels,
                           size_t len) {
    if (!w || !channels) return FLACENC_ERR_INVALID_ARG;
    if (w->finalized) return FLACENC_ERR_FINALIZED;
    if (n_channels != w->si.channels) return FLACENC_ERR_CHANNEL_COUNT_MISMATCH;  // encode.rs:856-859
    const size_t base = w->backlog.size();
    w->backlog.resize(base + len * n_channels);
    int32_t *d = w->backlog.data() + base;
    for (uint32_t c = 0; c < n_channels; c++)
        for (size_t i = 0; i < len; i++) d[i * n_channels + c] = channels[c][i];
    return w->drain(false);
}

int flacenc_finalize(flacenc_writer *w) {
    if (!w) return FLACENC_ERR_INVALID_ARG;
    return w->finalize();
}

void flacenc_writer_free(flacenc_writer *w) {
    if (!w) return;
    (void)w->finalize();  // Drop: finalize, ignoring errors (encode.rs:399-405, 2113-2117)
    delete w;
}

const uint8_t *flacenc_writer_data(flacenc_writer *w, size_t *len) {
    if (!w || !w->sink.memory) {
        if (len) *len = 0;
        return nullptr;
    }
    if (len) *len = w->sink.mem.size();
    return w->sink.mem.data();
}

int flacenc_pack_frames(uint32_t sample_rate, uint32_t bps, uint32_t channels,
                        uint64_t first_frame_number, uint32_t n_frames, uint32_t row_stride,
                        const void *frame_plans, const void *subframe_plans,
                        const int32_t *residual_rows, uint32_t threads, uint8_t *out, size_t cap,
                        uint64_t *offsets) {
    if (!frame_plans || !subframe_plans || !residual_rows || !offsets || n_frames == 0)
        return FLACENC_ERR_INVALID_ARG;
    PackedBatch pb;
    int rc = pack_batch(sample_rate, bps, channels, first_frame_number, n_frames, row_stride,
                        static_cast<const flacgpu_frame_plan *>(frame_plans),
                        static_cast<const flacgpu_subframe_plan *>(subframe_plans), residual_rows,
                        threads ? threads : 1, pb);
    if (rc) return rc;
    for (uint32_t f = 0; f <= n_frames; f++) offsets[f] = pb.offsets[f];
    if (!out || cap < pb.bytes.size()) return FLACENC_ERR_INVALID_ARG;
    std::memcpy(out, pb.bytes.data(), pb.bytes.size());
    return 0;
}

// Many independent streams at once (a music library): `threads` workers, each driving one stream at
// a time through FlacSampleWriter::new / write / finalize (encode.rs:487, 558, 624) into the job's own
// output buffer; the GPU is shared through the pooled lanes, the MD5 chains run on the writers'
// worker threads.
size_t flacenc_worst_case_bytes(const flacenc_options *opts, uint32_t bits_per_sample, uint32_t channels,
                                uint64_t pcm_frames) {
    if (!opts || opts->block_size == 0 || channels == 0) return 0;
    const uint64_t B = opts->block_size;
    const uint64_t frames = (pcm_frames + B - 1) / B + 1;
    // a frame: header <= 16 bytes + CRC-16, every subframe VERBATIM (header byte + wasted-bits escape) at bps + 1 bits
    // (a side channel), rounded up; one 18-byte seek point per frame at most; fLaC + STREAMINFO (4 + 38), block headers,
    // the vendor comment and the padding
    const uint64_t frame_bytes = 18 + channels * 2 + (B * channels * (bits_per_sample + 1) + 7) / 8;
    // VORBIS_COMMENT as the writer emits it (metadata/mod.rs:2010-2139): vendor length + vendor + field count + (length +
    // field) per field -- the caller's strings, whatever their size
    uint64_t vorbis = 0;
    if (opts->n_comment_fields || opts->vendor_string) {
        vorbis = 4 + (opts->vendor_string ? std::strlen(opts->vendor_string) : 16) + 4;
        for (uint32_t i = 0; i < opts->n_comment_fields; i++)
            vorbis += 4 + (opts->comment_fields && opts->comment_fields[i] ? std::strlen(opts->comment_fields[i]) : 0);
    }
    const uint64_t meta = 42 + 3 * 4 + 64 + vorbis + (opts->padding > 0 ? (uint64_t)opts->padding : 0);
    return (size_t)(frames * (frame_bytes + 18) + meta + 64);
}

int flacenc_encode_many(const flacenc_options *opts_in, flacenc_job *jobs, size_t n_jobs, uint32_t threads) {
    return flacenc_encode_many_devices(opts_in, jobs, n_jobs, threads, nullptr, 0);
}

// The same front end over SEVERAL devices: streams are independent (the natural shard -- no cross-stream state at all in
// the reference: one Encoder per file, encode.rs:1882-1980), so stream i goes whole to devices[i mod n_devices]; the
// pooled analysis lanes are keyed by device, the host workers and the MD5 engines are shared.
int flacenc_encode_many_devices(const flacenc_options *opts_in, flacenc_job *jobs, size_t n_jobs, uint32_t threads,
                                const int *devices, uint32_t n_devices) {
    if (!opts_in || (!jobs && n_jobs)) return FLACENC_ERR_INVALID_ARG;
    if (int e = options_error(*opts_in)) return e;
    std::vector<int> devs;
    if (devices && n_devices) {
        devs.assign(devices, devices + n_devices);
    } else if (n_devices == FLACENC_ALL_DEVICES) {
        for (int d = 0, n = flacgpu_device_count(); d < n; d++) devs.push_back(d);
        if (devs.empty()) return FLACENC_ERR_GPU;
    }
    for (int d : devs)
        if (d < 0 || d >= flacgpu_device_count()) return FLACENC_ERR_INVALID_ARG;
    // several streams at a time: their MD5 chains share the multi-stream engines (one AVX-512 lane each)
    flacenc_options shared = *opts_in;
    if (threads >= 3 && n_jobs >= 3) {
        shared.shared_md5 = 1;
        // whole streams are at hand and the parallelism comes from the streams: bigger batches (fewer, larger
        // kernels and copies per stream) than a lone streaming writer wants
        if (shared.batch_frames == 0) shared.batch_frames = 512;
    }
    const flacenc_options *opts = &shared;
    // Three phases per stream, claimed separately by the workers: PRIME (writer, the first kPrime blocks staged, hashed
    // and queued on the GPU), REST (the remaining samples) and FINISH (wait for the hash and the frames, metadata).
    // A worker primes as long as fewer than kOpen streams are open, then takes the oldest primed stream's rest, then
    // finishes the oldest submitted one -- so every stream's MD5 chain (26 ms for 512 frames of 24-bit stereo: the
    // burst's floor) starts within the first millisecond and never runs dry, whatever the thread count.  (One stream
    // per worker from start to end had 16 workers take 64 streams in four rounds of 28 ms, and 64 workers on a
    // 16-CPU quota burn twice the CPU time and meet the throttle; without the priming the last of a worker's four
    // streams started its hash 4.5-5 ms into the call.)  kOpen = the shared MD5 engines' lanes.
    constexpr size_t kOpen = 64;
    constexpr size_t kPrime = kDirectFrames;   // blocks: the smallest write that is encoded without staging
    std::mutex claim_mu;
    size_t next = 0, next_rest = 0, next_fin = 0;
    std::vector<std::unique_ptr<flacenc_writer>> open_writers(n_jobs);
    std::unique_ptr<std::atomic<int>[]> stage(new std::atomic<int>[n_jobs ? n_jobs : 1]);   // 1 primed, 2 submitted
    std::vector<double> t_start(n_jobs, 0.0);
    std::vector<size_t> primed(n_jobs, 0);   // samples the first phase took
    for (size_t i = 0; i < n_jobs; i++) stage[i].store(0, std::memory_order_relaxed);
    const double t_begin = now_ms();
    auto prime = [&](size_t i) {
        flacenc_job &j = jobs[i];
        j.out_len = 0;
        t_start[i] = now_ms();
        if (!j.samples || !j.out || j.bits_per_sample < 1 || j.bits_per_sample > 32 || j.channels == 0 ||
            j.count % j.channels || j.count == 0) {
            j.status = FLACENC_ERR_INVALID_ARG;
            return;
        }
        std::unique_ptr<flacenc_writer> w(new flacenc_writer());
        w->kind = flacenc_writer::SAMPLE;
        w->use_fixed_sink(j.out, j.out_cap);
        flacenc_options mine = *opts;
        if (!devs.empty()) mine.device = devs[i % devs.size()];
        int rc = w->init(mine, j.sample_rate, j.bits_per_sample, j.channels, true, j.count / j.channels, nullptr);
        const size_t head = kPrime * static_cast<size_t>(opts->block_size) * j.channels;
        if (!rc && opts->shared_md5 && j.count >= 2 * head) {
            rc = w->write_direct(j.samples, head);
            primed[i] = head;
        }
        j.status = rc;
        j.start_ms = t_start[i] - t_begin;
        if (!rc) open_writers[i] = std::move(w);
    };
    auto rest = [&](size_t i) {
        while (stage[i].load(std::memory_order_acquire) < 1) std::this_thread::yield();   // (being primed)
        flacenc_writer *w = open_writers[i].get();
        if (!w) return;   // rejected or failed in its first phase: status says so
        flacenc_job &j = jobs[i];
        j.status = w->write_direct(j.samples + primed[i], j.count - primed[i]);
        if (j.status) open_writers[i].reset();
    };
    auto finish = [&](size_t i) {
        while (stage[i].load(std::memory_order_acquire) < 2) std::this_thread::yield();   // (its rest is at work)
        std::unique_ptr<flacenc_writer> w = std::move(open_writers[i]);
        if (!w) return;
        flacenc_job &j = jobs[i];
        j.status = w->finalize();
        j.out_len = w->sink.fixed_len;
        j.elapsed_ms = now_ms() - t_start[i];
        j.pack_ms = w->stats.pack_ms;
        j.gpu_ms = w->stats.gpu_ms;
        j.md5_ms = w->stats.md5_ms;
    };
    auto work = [&]() {
        for (;;) {
            size_t idx;
            int what;
            {
                std::lock_guard<std::mutex> lock(claim_mu);
                if (next < n_jobs && next - next_fin < kOpen) {
                    what = 0;
                    idx = next++;
                } else if (next_rest < next) {
                    what = 1;
                    idx = next_rest++;
                } else if (next_fin < next_rest) {
                    what = 2;
                    idx = next_fin++;
                } else {
                    return;   // everything claimed
                }
            }
            if (what == 0) {
                prime(idx);
                stage[idx].store(1, std::memory_order_release);
            } else if (what == 1) {
                rest(idx);
                stage[idx].store(2, std::memory_order_release);
            } else {
                finish(idx);
            }
        }
    };
    // (the workers mostly wait -- for the GPU, for their stream's turn --: two per CPU the process may really use; a container
    // with a CPU quota far below the host's thread count is throttled for whole periods when more threads than that spin)
    unsigned nt = threads ? threads : std::min<unsigned>(std::max(4u, 2 * flacenc_host::usable_cpus()), 64);
    nt = static_cast<unsigned>(std::min<size_t>(nt, std::max<size_t>(1, n_jobs)));
    // the helpers come from a process-wide pool of parked threads (creating 63 threads per call cost a 64-stream
    // burst 2-3 ms before its last stream had even started)
    WorkerPool::get().run(nt - 1, work);
    for (size_t i = 0; i < n_jobs; i++)
        if (jobs[i].status) return jobs[i].status;
    return 0;
}

// (many SMALL streams -- flacenc_encode_many_coalesced -- live in coalesce.cpp)

// The bytes in front of the first frame (fLaC marker + STREAMINFO + SEEKTABLE + VORBIS_COMMENT +
// PADDING) exactly as a writer that had produced frames of these sizes would leave them at finalize
// (Encoder::new + encode's seek points + finalize_inner, encode.rs:1882-1980, 1999-2003, 2024-2110):
// what the owner of a stream whose frame ranges were encoded elsewhere (other GPUs) needs.
int flacenc_stream_header(const flacenc_options *opts, uint32_t sample_rate, uint32_t bits_per_sample,
                          uint32_t channels, uint64_t total_pcm_frames, const uint8_t md5[16], uint64_t n_frames,
                          const uint32_t *frame_sizes, uint32_t last_frame_len, uint8_t *out, size_t cap,
                          size_t *len) {
    if (!opts || !md5 || !frame_sizes || !len || n_frames == 0 || last_frame_len == 0 ||
        last_frame_len > opts->block_size)
        return FLACENC_ERR_INVALID_ARG;
    if (bits_per_sample < 1 || bits_per_sample > 32) return FLACENC_ERR_INVALID_BITS_PER_SAMPLE;
    flacenc_writer w;
    w.header_only = true;
    if (int rc = w.init(*opts, sample_rate, bits_per_sample, channels, true, total_pcm_frames, nullptr)) return rc;
    const uint32_t B = opts->block_size;
    for (uint64_t f = 0; f < n_frames; f++) {
        const uint32_t n = (f + 1 == n_frames) ? last_frame_len : B;
        w.seekpoints.push_back({w.samples_written, w.byte_count, static_cast<uint16_t>(n), true});
        w.samples_written += n;
        const uint32_t size = frame_sizes[f];
        if (size != 0 && size < kMaxFrameSize) {
            w.si.min_frame = w.si.min_frame ? std::min(w.si.min_frame, size) : size;
            w.si.max_frame = std::max(w.si.max_frame, size);
        }
        w.byte_count += size;
    }
    w.frame_number = n_frames;
    std::memcpy(w.si.md5, md5, 16);
    w.finalized = true;
    if (int rc = w.finalize_encoder()) return rc;
    *len = w.metadata_len;
    if (!out || cap < w.metadata_len) return FLACENC_ERR_INVALID_ARG;
    std::memcpy(out, w.sink.mem.data(), w.metadata_len);
    return 0;
}

// The 34 bytes of a STREAMINFO block body (metadata/mod.rs:1599-1630, 1740-1760) as the writers
// serialise them: exposed so that the layout can be checked against the reference's literal bytes.
int flacenc_streaminfo_bytes(uint32_t min_block, uint32_t max_block, uint32_t min_frame, uint32_t max_frame,
                             uint32_t sample_rate, uint32_t channels, uint32_t bits_per_sample,
                             uint64_t total_samples, const uint8_t md5[16], uint8_t out[34]) {
    if (!md5 || !out) return FLACENC_ERR_INVALID_ARG;
    StreamInfo si;
    si.min_block = min_block;
    si.max_block = max_block;
    si.min_frame = min_frame;
    si.max_frame = max_frame;
    si.sample_rate = sample_rate;
    si.channels = channels;
    si.bps = bits_per_sample;
    si.total_samples = total_samples;
    std::memcpy(si.md5, md5, 16);
    MetaLayout m;
    const std::vector<uint8_t> v = build_metadata(si, m);   // "fLaC" + header (4) + body (34)
    if (v.size() != 42) return FLACENC_ERR_IO;
    std::memcpy(out, v.data() + 8, 34);
    return 0;
}

int flacenc_writer_stats(flacenc_writer *w, flacenc_stats *out) {
    if (!w || !out) return FLACENC_ERR_INVALID_ARG;
    *out = w->stats;
    return 0;
}

}  // extern "C"

// =====================================================================================
// FlacStreamWriter, encode.rs:1050-1290: subset frames, no metadata, parameters per call
// =====================================================================================
struct flacenc_stream_writer {
    flacenc_options o{};
    Sink sink;
    uint64_t frame_number = 0;
    struct Key {
        uint32_t bps, channels, cap;
        bool operator<(const Key &k) const {
            return std::tie(bps, channels, cap) < std::tie(k.bps, k.channels, k.cap);
        }
    };
    std::map<Key, flacgpu_ctx *> ctxs;
    ~flacenc_stream_writer() {
        for (auto &kv : ctxs) flacgpu_destroy(kv.second);
    }
};

extern "C" {

int flacenc_stream_writer_new(const flacenc_options *opts, const flacenc_sink *sink,
                              flacenc_stream_writer **out) {
    if (!opts || !out) return FLACENC_ERR_INVALID_ARG;
    *out = nullptr;
    if (int e = options_error(*opts)) return e;
    auto *w = new flacenc_stream_writer();
    w->o = *opts;
    if (sink) {
        w->sink.memory = false;
        w->sink.cb = *sink;
    }
    *out = w;
    return 0;
}

int flacenc_stream_writer_write(flacenc_stream_writer *w, uint32_t rate, uint32_t channels,
                                uint32_t bps, const int32_t *samples, size_t count) {
    if (!w || (!samples && count)) return FLACENC_ERR_INVALID_ARG;
    // validation order of encode.rs:1151-1191
    if (bps < 1 || bps > 32) return FLACENC_ERR_NON_SUBSET_BITS_PER_SAMPLE;
    if (channels == 0 || count % channels) return FLACENC_ERR_SAMPLES_NOT_DIVISIBLE_BY_CHANNELS;
    if (channels > 8) return FLACENC_ERR_EXCESSIVE_CHANNELS;
    const size_t n = count / channels;
    if (n == 0 || n > 65535) return FLACENC_ERR_INVALID_BLOCK_SIZE;
    {  // SampleRate::try_from + "Streaminfo => NonSubsetSampleRate"
        bool named = false;
        for (uint32_t r : {88200u, 176400u, 192000u, 8000u, 16000u, 22050u, 24000u, 32000u, 44100u, 48000u, 96000u})
            named |= (r == rate);
        bool coded = named || (rate % 1000 == 0 && rate / 1000 < 255) ||
                     (rate % 10 == 0 && rate / 10 < 65535) || rate < 65535;
        if (!coded) return rate < (1u << 20) ? FLACENC_ERR_NON_SUBSET_SAMPLE_RATE : FLACENC_ERR_INVALID_SAMPLE_RATE;
    }
    if (!(bps == 8 || bps == 12 || bps == 16 || bps == 20 || bps == 24 || bps == 32))
        return FLACENC_ERR_NON_SUBSET_BITS_PER_SAMPLE;
    uint32_t cap = 16;
    while (cap < n) cap <<= 1;
    cap = std::min<uint32_t>(cap, FLACGPU_MAX_BLOCK_SIZE);
    flacenc_stream_writer::Key key{bps, channels, cap};
    flacgpu_ctx *ctx = nullptr;
    auto it = w->ctxs.find(key);
    if (it == w->ctxs.end()) {
        flacgpu_options g = gpu_options(w->o, cap);
        int rc = flacgpu_create(&g, bps, channels, w->o.device, 1, &ctx);
        if (rc) return map_gpu_error(rc);
        w->ctxs[key] = ctx;
    } else {
        ctx = it->second;
    }
    flacgpu_frame_plan plan;
    std::vector<flacgpu_subframe_plan> subs(channels);
    std::vector<int32_t> rows(static_cast<size_t>(channels) * cap);
    int rc = flacgpu_analyze(ctx, samples, FLACGPU_LAYOUT_INTERLEAVED, 1, static_cast<uint32_t>(n),
                             &plan, subs.data(), rows.data());
    if (rc) return map_gpu_error(rc);
    PackedBatch pb;
    if (int e = pack_batch(rate, bps, channels, w->frame_number, 1, cap, &plan, subs.data(), rows.data(), 1, pb))
        return e;
    if (int e = w->sink.write(pb.bytes.data(), pb.bytes.size())) return e;
    if (w->frame_number >= kMaxFrameNumber) return FLACENC_ERR_EXCESSIVE_FRAME_NUMBER;
    w->frame_number++;
    return 0;
}

const uint8_t *flacenc_stream_writer_data(flacenc_stream_writer *w, size_t *len) {
    if (!w || !w->sink.memory) {
        if (len) *len = 0;
        return nullptr;
    }
    if (len) *len = w->sink.mem.size();
    return w->sink.mem.data();
}

void flacenc_stream_writer_free(flacenc_stream_writer *w) { delete w; }

}  // extern "C"

// ---- what coalesce.cpp shares with this file (host_internal.h) -----------------------------------------------------------
namespace flacenc_host {
double now_ms() { return ::now_ms(); }
int options_error(const flacenc_options &o) { return ::options_error(o); }
flacgpu_options gpu_options(const flacenc_options &o, uint32_t block_size) { return ::gpu_options(o, block_size); }
void pack_le(const int32_t *s, size_t count, unsigned width, uint8_t *d) { ::pack_le(s, count, width, d); }
int stream_header_len(const flacenc_options &o, uint32_t sample_rate, uint32_t bits_per_sample, uint32_t channels,
                      uint64_t total_pcm_frames, size_t *len) {
    flacenc_writer w;
    w.header_only = true;
    const int rc = w.init(o, sample_rate, bits_per_sample, channels, true, total_pcm_frames, nullptr);
    w.finalized = true;   // (nothing to flush: the destructor must not try)
    if (!rc && len) *len = w.metadata_len;
    return rc;
}
void run_parallel(unsigned helpers, const std::function<void()> &fn) { WorkerPool::get().run(helpers, fn); }
void release_lane_pool() {
    std::vector<Lane *> all;
    {
        std::lock_guard<std::mutex> lock(lane_pool().mu);
        for (auto &e : lane_pool().idle) all.push_back(e.second);
        lane_pool().idle.clear();
    }
    for (Lane *l : all) l->destroy();
}
}  // namespace flacenc_host
