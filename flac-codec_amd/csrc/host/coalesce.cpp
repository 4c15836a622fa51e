// coalesce.cpp -- flacenc_encode_many_coalesced: many whole streams in host memory -> .flac bytes, the frames of all
// streams of one shape travelling through ONE ring of pinned staging buffers at the link's rate.
//
// The reference opens an `Encoder` per file and walks its blocks one by one (encode.rs:487-627): a writer per stream gives
// every stream batches of its own frames, each paying the kernels' launches and their serial walk over a block, and small
// streams never fill the GPU.  Here (r06; r05's first form uploaded int32 from the callers' pageable buffers, one
// synchronous batch per worker, and reached 0.09 of the link):
//
//   * batches are made of SEGMENTS -- runs of whole blocks of several streams, taken from the streams in turn so that all
//     their MD5 chains advance together -- and every frame keeps its own stream's frame number
//     (flacgpu_encode_segments_packed_async_host);
//   * a ring of `depth` slots, each an analysis context with a pinned input and a pinned output buffer: the workers PACK the
//     callers' int32 samples into the slot's input at stream width (2 or 3 bytes per sample across PCIe instead of 4; the
//     packed bytes are update_md5's byte string, encode.rs:1292-1318, so the MD5 lanes hash them where they lie), the last
//     packer SUBMITS the batch (upload, kernels and frame assembly queued, k_frame64 storing the frames straight into the
//     slot's pinned output), one worker at a time RETIRES the oldest batch and the frames are COPIED to their places in the
//     callers' output buffers (the only host copy of an output byte); a slot is reused once its upload, its frames and the
//     MD5 runs that read its input are done.  Upload of batch i + 1, kernels of batch i and the frames of batch i - 1 are in
//     flight together;
//   * MD5 (one serial chain per stream; a chain's speed is the latency of its 64 dependent steps per 64 bytes: 0.6 GB/s on the
//     bench host whatever the width) goes two ways.  A stream of up to 32 blocks travels as ONE segment and its chain is run by
//     a worker: HASH tasks take up to 48 such streams of a batch and advance them in lockstep, three interleaved groups of 16
//     AVX-512 lanes (md5_mb.cpp: 18 GB/s per thread) -- no engine thread, no queue, no lock.  Longer streams are cut into
//     segments of a quantum that lets every batch visit every stream, so that all chains advance together, and their runs go
//     to the shared engine threads in stream order;
//   * the work is tasks on one queue set -- COPY > RETIRE > PACK > HASH > TAIL > FIN (digest + metadata of a stream whose
//     parts are through: streams finish while later batches are still in flight) -- drawn by a few workers: the box's CPU
//     QUOTA (cgroup cpu.max), not its thread count, sizes them, and the first and last batches are quarter-size (the first
//     upload starts early, little is left behind the last);
//   * a stream's short last block is one synchronous one-frame call on a small context of its own kind; everything in front
//     of the first frame is rebuilt from the frame sizes (flacenc_stream_header) -- the bytes Encoder::new / encode /
//     finalize_inner leave (encode.rs:1882-2110), stream by stream.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <sys/resource.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "checksums.h"
#include "host_internal.h"
#include "md5_mb.h"

using namespace flacenc;
using namespace flacenc_host;

namespace {

// ---- ring slots, kept between calls (a context for 2048 stereo frames is a few hundred MB of device memory and tens of
// milliseconds to create; pinning its staging buffers costs more) ------------------------------------------------------------
struct RingSlot {
    flacgpu_ctx *ctx = nullptr;
    uint8_t *in = nullptr, *out = nullptr;   // pinned
    size_t in_cap = 0, out_cap = 0;
    size_t bytes() const { return in_cap + out_cap; }
    void destroy() {
        if (ctx) flacgpu_destroy(ctx);
        flacgpu_host_free(in);
        flacgpu_host_free(out);
        ctx = nullptr;
        in = out = nullptr;
    }
};
struct RingKey {
    flacgpu_options g;
    uint32_t bps, ch, frames;
    int device;
    bool operator==(const RingKey &k) const {   // field by field: the structs carry padding
        return g.block_size == k.g.block_size && g.max_partition_order == k.g.max_partition_order &&
               g.max_lpc_order == k.g.max_lpc_order && g.mid_side == k.g.mid_side &&
               g.exhaustive_channel_correlation == k.g.exhaustive_channel_correlation && g.window_kind == k.g.window_kind &&
               g.window_param == k.g.window_param && bps == k.bps && ch == k.ch && frames == k.frames && device == k.device;
    }
};
struct RingPool {
    static constexpr size_t kIdleBytes = size_t(1) << 30;   // pinned bytes kept idle
    static constexpr size_t kIdleSlots = 12;
    std::mutex mu;
    std::vector<std::pair<RingKey, RingSlot>> idle;
    static RingPool &get() {
        static RingPool *p = new RingPool();   // leaked on purpose: no HIP calls during static destruction
        return *p;
    }
    int take(const RingKey &k, size_t in_bytes, RingSlot *s) {
        {
            std::lock_guard<std::mutex> l(mu);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == k && idle[i].second.in_cap >= in_bytes) {
                    *s = idle[i].second;
                    idle.erase(idle.begin() + (ptrdiff_t)i);
                    return 0;
                }
        }
        int rc = flacgpu_create(&k.g, k.bps, k.ch, k.device, k.frames, &s->ctx);
        if (rc) return rc;
        s->out_cap = flacgpu_packed_cap(s->ctx);
        s->out = static_cast<uint8_t *>(flacgpu_host_alloc(s->out_cap));
        s->in_cap = in_bytes + 64;
        s->in = static_cast<uint8_t *>(flacgpu_host_alloc(s->in_cap));
        if (!s->out || !s->in) {
            s->destroy();
            return FLACGPU_ERR_HIP;
        }
        return 0;
    }
    void give(const RingKey &k, const RingSlot &s) {
        std::vector<RingSlot> drop;
        {
            std::lock_guard<std::mutex> l(mu);
            idle.emplace_back(k, s);
            size_t total = 0;
            for (auto &e : idle) total += e.second.bytes();
            while (idle.size() > 1 && (idle.size() > kIdleSlots || total > kIdleBytes)) {   // the oldest go
                total -= idle.front().second.bytes();
                drop.push_back(idle.front().second);
                idle.erase(idle.begin());
            }
        }
        for (auto &d : drop) d.destroy();
    }
    void release_all() {
        std::vector<std::pair<RingKey, RingSlot>> all;
        {
            std::lock_guard<std::mutex> l(mu);
            all.swap(idle);
        }
        for (auto &e : all) e.second.destroy();
    }
};

constexpr uint32_t kSolo = 32;   // blocks: streams up to this long are "solo"

struct Stream {
    size_t job = 0;
    uint64_t pcm_frames = 0, whole = 0;   // samples per channel, whole blocks
    uint32_t tail = 0;                    // samples of the short last block (0: none)
    size_t hlen = 0;                      // bytes in front of the first frame
    std::vector<uint32_t> sizes;          // per frame
    uint64_t pos = 0;                     // bytes of the frames placed so far (retired in stream order)
    std::vector<uint8_t> tail_bytes;      // the short last block's frame
    std::vector<uint8_t> tail_le;         // ... and its samples as update_md5's bytes
    std::vector<uint8_t> le;              // the whole byte string, only when the upload cannot take the stream's width
    Md5 md5;
    Md5Lane *lane = nullptr;
    bool attach_tried = false;            // the lane is attached by the first run (under mu)
    bool solo = false;                    // a stream of a few blocks: ONE segment, hashed by the worker that packs it
    std::atomic<int> parts_left{0};       // the run of whole blocks, the short last block: the stream is finished when both are through
    size_t last_batch = 0;                // batch of its last whole-block segment
    // MD5 runs are pushed in stream order whatever the order the packers finish in
    std::mutex mu;
    uint64_t next_md5_frame = 0;
    struct Pending {
        const uint8_t *p;
        size_t n;
        uint32_t frames;
        std::atomic<uint64_t> *ticket;
    };
    std::map<uint64_t, Pending> pending;
    uint64_t last_ticket = 0;
};

struct Seg {
    size_t stream;
    uint64_t first;   // first frame (block index in its stream)
    uint32_t n, boff; // frames, frame offset inside the batch
    std::atomic<uint64_t> ticket{0};   // of its MD5 run (0: not pushed yet)
    Seg(size_t s, uint64_t f, uint32_t n_, uint32_t b) : stream(s), first(f), n(n_), boff(b) {}
    Seg(const Seg &o) : stream(o.stream), first(o.first), n(o.n), boff(o.boff), ticket(o.ticket.load()) {}
};
struct Batch {
    std::vector<Seg> segs;
    uint32_t frames = 0;
    int slot = -1;
    std::atomic<uint32_t> pack_left{0}, copy_left{0}, hash_left{0};
    std::atomic<uint32_t> release_left{2};   // the frames copied out, the solo chains hashed: the second one through frees the slot
    std::vector<uint32_t> solo_idx;          // segments hashed by HASH tasks (filled at submission)
    bool submitted = false, failed = false;
    Batch() = default;
    Batch(Batch &&o) noexcept : segs(std::move(o.segs)), frames(o.frames), slot(o.slot), submitted(o.submitted), failed(o.failed) {}
};
struct CopyJob {
    const uint8_t *src;
    uint8_t *dst;
    size_t n;
};
struct Task {
    enum Kind { NONE, PACK, COPY, RETIRE, TAIL, FIN, HASH } kind = NONE;
    size_t batch = 0, i0 = 0, i1 = 0;   // PACK: segments [i0, i1) of the batch; COPY: copy jobs [i0, i1); TAIL: stream i0; FIN: fin_list[i0, i1); HASH: solo_idx[i0, i1)
};

}  // namespace

static double cpu_ms() {   // user + system time of the whole process
    rusage r;
    getrusage(RUSAGE_SELF, &r);
    return (r.ru_utime.tv_sec + r.ru_stime.tv_sec) * 1e3 + (r.ru_utime.tv_usec + r.ru_stime.tv_usec) * 1e-3;
}

// FLACENC_TRACE=1: the phases of a call on stderr (milliseconds since its start)
static bool trace_on() {
    static const bool on = [] {
        const char *e = std::getenv("FLACENC_TRACE");
        return e && e[0] && e[0] != '0';
    }();
    return on;
}

// frames per batch: the options' batch_frames (at least 64), else about 16 Mi samples; never more than the group holds
static uint32_t coalesce_batch_cap(uint32_t batch_frames, size_t samples_per_block, uint64_t total_whole) {
    uint32_t cap = batch_frames ? std::max(batch_frames, 64u) : (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(64, (16u << 20) / samples_per_block));
    return (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(total_whole, 1));
}

// ---- the batches of one shape group: which run of whole blocks of which stream goes into which batch ---------------------------
// `whole[k]` = whole blocks of stream k; batch_cap = frames a slot's context takes.  Invariants (tests/test_coalesce_plan.py):
// every block of every stream exactly once, a stream's segments in stream order and in non-decreasing batches, no batch above
// batch_cap, and a stream of up to kSolo blocks in ONE segment (its MD5 chain is one run hashed by one task).
struct PlanSeg {
    uint32_t stream;   // index into `whole`
    uint64_t first;    // first block of the segment in its stream
    uint32_t n;        // blocks
};
static std::vector<std::vector<PlanSeg>> plan_batches(const std::vector<uint64_t> &whole, uint32_t batch_cap) {
    uint64_t total_whole = 0;
    size_t n_active = 0;
    for (uint64_t w : whole) {
        total_whole += w;
        n_active += w ? 1 : 0;
    }
    std::vector<std::vector<PlanSeg>> batches;
    if (!total_whole || !batch_cap) return batches;
    // batch sizes: the first and the last ones smaller (a quarter, a half of the cap) -- the first upload starts after a
    // quarter of a batch has been packed, and behind the last upload only a quarter of a batch is left to analyse,
    // assemble and copy
    std::vector<uint32_t> plan;
    {
        uint64_t left = total_whole;
        std::vector<uint32_t> head, tailv;
        // (not where the streams' MD5 chains bound the call -- a chain makes ~0.2 Gsamples/s, fewer than ~70 streams
        // cannot use the link --: there the chains only need their bytes EARLY, and every batch costs them a pass)
        const bool chain_bound = n_active < 70;
        for (uint32_t z : {batch_cap / 4, batch_cap / 2}) {
            if (!chain_bound && z >= 64 && left >= 4ull * z) {
                head.push_back(z);
                tailv.push_back(z);
                left -= 2ull * z;
            }
        }
        // (chain-bound: ONE small batch in front -- a few blocks of every stream, packed in a tenth of a millisecond --
        // starts all the chains together; a full first batch starts the last of them a millisecond late)
        if (chain_bound && n_active && left >= 4ull * batch_cap) {
            const uint32_t z = (uint32_t)std::min<uint64_t>(batch_cap / 4, 8ull * n_active);
            if (z >= 8) {
                head.push_back(z);
                left -= z;
            }
        }
        plan = head;
        while (left) {
            const uint32_t z = (uint32_t)std::min<uint64_t>(left, batch_cap);
            plan.push_back(z);
            left -= z;
        }
        for (size_t i = tailv.size(); i-- > 0;) plan.push_back(tailv[i]);
    }
    std::vector<uint64_t> done(whole.size(), 0);
    std::deque<uint32_t> active;
    for (size_t k = 0; k < whole.size(); k++)
        if (whole[k]) active.push_back((uint32_t)k);
    for (size_t pi = 0; !active.empty(); pi++) {
        uint32_t cap = pi < plan.size() ? plan[pi] : batch_cap;   // (batches closed early in front of a solo stream: more of them)
        std::vector<PlanSeg> b;
        uint32_t frames = 0;
        // a quantum that lets a batch visit every active stream, but no less than kSolo blocks (solo streams are taken whole
        // whatever the quantum; the floor keeps a long stream's segments from getting tiny -- except in the small batch a
        // chain-bound call starts with)
        const uint32_t q_floor = (pi == 0 && n_active < 70 && cap < batch_cap) ? 1u : kSolo;
        const uint32_t q = std::max<uint32_t>(q_floor, (uint32_t)((cap + active.size() - 1) / active.size()));
        while (frames < cap && !active.empty()) {
            const uint32_t k = active.front();
            const bool solo = whole[k] <= kSolo;
            const uint32_t want = solo ? (uint32_t)whole[k] : (uint32_t)std::min<uint64_t>(q, whole[k] - done[k]);
            if (want > cap - frames && solo) {   // a solo stream is NEVER cut (its chain is one run, hashed by one task):
                if (frames) break;               //   it opens the next batch,
                cap = want;                      //   or has a small batch (the plan's remainder) made room (<= kSolo <= batch_cap)
            }
            active.pop_front();
            const uint32_t n = std::min<uint32_t>(want, cap - frames);
            b.push_back(PlanSeg{k, done[k], n});
            frames += n;
            done[k] += n;
            if (done[k] < whole[k]) {
                if (n < want) active.push_front(k);   // (cut by the batch's end: it goes on first in the next one)
                else active.push_back(k);
            }
        }
        batches.push_back(std::move(b));
    }
    return batches;
}

extern "C" {

// Test hook (tests/test_coalesce_plan.py): the batch plan of a shape group as flat arrays -- segment i is blocks
// [seg_first[i], seg_first[i] + seg_n[i]) of stream seg_stream[i] in batch seg_batch[i].  Returns the number of segments (the
// arrays hold up to `cap`), *batch_cap_out the frames per batch the front end would size its contexts for.
size_t flacenc_coalesce_plan(const uint64_t *whole, size_t n_streams, uint32_t batch_frames, uint32_t samples_per_block, uint32_t *seg_stream,
                             uint64_t *seg_first, uint32_t *seg_n, uint32_t *seg_batch, size_t cap, uint32_t *batch_cap_out) {
    std::vector<uint64_t> w(whole, whole + n_streams);
    uint64_t total = 0;
    for (uint64_t v : w) total += v;
    const uint32_t bc = coalesce_batch_cap(batch_frames, samples_per_block ? samples_per_block : 1, total);
    if (batch_cap_out) *batch_cap_out = bc;
    size_t n = 0;
    const auto plan = plan_batches(w, bc);
    for (size_t b = 0; b < plan.size(); b++)
        for (const PlanSeg &g : plan[b]) {
            if (n < cap) {
                seg_stream[n] = g.stream;
                seg_first[n] = g.first;
                seg_n[n] = g.n;
                seg_batch[n] = (uint32_t)b;
            }
            n++;
        }
    return n;
}

void flacenc_release_pools(void) {
    RingPool::get().release_all();
    release_lane_pool();
}

int flacenc_encode_many_coalesced(const flacenc_options *opts_in, flacenc_job *jobs, size_t n_jobs, uint32_t threads) {
    if (!opts_in || (!jobs && n_jobs)) return FLACENC_ERR_INVALID_ARG;
    if (int e = options_error(*opts_in)) return e;
    const flacenc_options &o = *opts_in;
    const uint32_t B = o.block_size;
    struct Shape {
        uint32_t rate, bps, ch;
        bool operator<(const Shape &k) const { return std::tie(rate, bps, ch) < std::tie(k.rate, k.bps, k.ch); }
    };
    std::map<Shape, std::vector<size_t>> groups;
    std::vector<std::unique_ptr<Stream>> st(n_jobs);
    const double t_begin = now_ms();
    const double cpu_begin = trace_on() ? cpu_ms() : 0.0;
    // (streams of one length share their header's size: the check constructs a header-only writer)
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint64_t>, std::pair<int, size_t>> header_memo;
    for (size_t i = 0; i < n_jobs; i++) {
        flacenc_job &j = jobs[i];
        j.out_len = 0;
        j.status = 0;
        j.start_ms = j.elapsed_ms = j.pack_ms = j.gpu_ms = j.md5_ms = 0.0;
        if (!j.samples || !j.out || j.bits_per_sample < 1 || j.bits_per_sample > 32 || j.channels == 0 || j.channels > 8 ||
            j.count % j.channels || j.count == 0) {
            j.status = FLACENC_ERR_INVALID_ARG;
            continue;
        }
        size_t hlen = 0;
        {
            const auto mk = std::make_tuple(j.sample_rate, j.bits_per_sample, j.channels, (uint64_t)(j.count / j.channels));
            auto it = header_memo.find(mk);
            if (it == header_memo.end()) {
                size_t l = 0;
                const int rc = stream_header_len(o, j.sample_rate, j.bits_per_sample, j.channels, j.count / j.channels, &l);
                it = header_memo.emplace(mk, std::make_pair(rc, l)).first;
            }
            if (it->second.first) {
                j.status = it->second.first;
                continue;
            }
            hlen = it->second.second;
        }
        st[i].reset(new Stream());
        Stream &s = *st[i];
        s.job = i;
        s.hlen = hlen;
        s.pcm_frames = j.count / j.channels;
        s.whole = s.pcm_frames / B;
        s.tail = (uint32_t)(s.pcm_frames % B);
        s.sizes.assign(s.whole + (s.tail ? 1 : 0), 0);
        groups[Shape{j.sample_rate, j.bits_per_sample, j.channels}].push_back(i);
    }
    if (trace_on()) std::fprintf(stderr, "[coalesce] %zu streams checked at %.2f ms\n", n_jobs, now_ms() - t_begin);
    // workers: packing, copying and one waiting for the GPU -- a few; the MD5 engines and the runtime's own threads need CPUs too
    const unsigned cpus = usable_cpus();
    const unsigned nt = std::max(1u, std::min<unsigned>(threads ? threads : std::max(2u, std::min(cpus > 6 ? cpus - 6 : 2u, 10u)), 64u));
    std::mutex err_mu;
    auto fail = [&](size_t job, int rc) {
        std::lock_guard<std::mutex> l(err_mu);
        if (!jobs[job].status) jobs[job].status = rc;
    };
    auto has_failed = [&](size_t job) {
        std::lock_guard<std::mutex> l(err_mu);
        return jobs[job].status != 0;
    };
    const int device = o.device >= 0 ? o.device : flacgpu_current_device();

    for (auto &kv : groups) {
        const Shape sh = kv.first;
        const std::vector<size_t> &ids = kv.second;
        const unsigned width = (sh.bps + 7) / 8;
        const unsigned up_width = (width < 4 && flacgpu_packed_input_shape_supported(B, sh.ch, width)) ? width : 4u;
        const size_t per = (size_t)B * sh.ch;
        const flacgpu_options g = gpu_options(o, B);

        // ---- batches: about 16 Mi samples each, the streams taken in turn, a quantum of whole blocks at a time
        uint64_t total_whole = 0;
        for (size_t i : ids) total_whole += st[i]->whole;
        const uint32_t batch_cap = coalesce_batch_cap(o.batch_frames, per, total_whole);
        std::vector<Batch> batches;
        {
            std::vector<uint64_t> whole_of(ids.size());
            for (size_t k = 0; k < ids.size(); k++) whole_of[k] = st[ids[k]]->whole;
            const std::vector<std::vector<PlanSeg>> plan = plan_batches(whole_of, batch_cap);
            for (size_t k = 0; k < ids.size(); k++) {
                // streams of up to kSolo blocks travel as ONE segment and are hashed by HASH tasks (below)
                st[ids[k]]->solo = whole_of[k] && whole_of[k] <= kSolo;
                st[ids[k]]->attach_tried = whole_of[k] <= kSolo;   // (no engine lane: their short last block is hashed where it is packed)
            }
            for (const auto &pb : plan) {
                Batch b;
                for (const PlanSeg &g : pb) {
                    const size_t i = ids[g.stream];
                    b.segs.emplace_back(i, g.first, g.n, b.frames);
                    b.frames += g.n;
                    if (g.first + g.n == st[i]->whole) st[i]->last_batch = batches.size();
                }
                batches.push_back(std::move(b));
            }
        }
        std::vector<size_t> tails;
        for (size_t i : ids)
            if (st[i]->tail) tails.push_back(i);

        // ---- the ring
        const RingKey key{g, sh.bps, sh.ch, batch_cap, device};
        const RingKey tail_key{g, sh.bps, sh.ch, 1u, device};
        const size_t in_bytes = (size_t)batch_cap * per * up_width;
        const unsigned depth = (unsigned)std::min<size_t>(std::max<size_t>(batches.size(), 1), o.pipeline_depth ? std::min<uint32_t>(o.pipeline_depth, 8u) : 6u);
        std::vector<RingSlot> slots;
        int ring_rc = 0;
        for (unsigned d = 0; d < depth && !batches.empty(); d++) {
            RingSlot s;
            const int rc = RingPool::get().take(key, in_bytes, &s);
            if (rc) {
                ring_rc = rc;
                break;
            }
            slots.push_back(s);
        }
        if (slots.empty() && !batches.empty()) {   // no context at all: every stream of the group fails
            for (size_t i : ids) fail(i, ring_rc == FLACGPU_ERR_UNSUPPORTED ? FLACENC_ERR_UNSUPPORTED : FLACENC_ERR_GPU);
            continue;
        }

        if (trace_on())
            std::fprintf(stderr, "[coalesce] %zu batches of <= %u frames, %u slots, width %u -> %u, ring ready at %.2f ms\n", batches.size(),
                         batch_cap, (unsigned)slots.size(), width, up_width, now_ms() - t_begin);
        // ---- scheduler state
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Task> copyq, packq, hashq, tailq, finq;
        std::function<void(size_t)> release_slot_fn;   // (release_slot, defined below the tasks that end with it)
        std::vector<size_t> fin_list;     // streams ready to be finished, in the order they became ready (mu)
        size_t fin_queued = 0, fin_left = ids.size();
        std::vector<int> free_slots;
        for (int d = (int)slots.size() - 1; d >= 0; d--) free_slots.push_back(d);
        size_t next_assign = 0, retire_next = 0, batches_done = 0, tails_left = tails.size();
        bool retiring = false;
        std::vector<std::vector<CopyJob>> copy_jobs(batches.size());
        for (size_t t : tails) {
            Task k;
            k.kind = Task::TAIL;
            k.i0 = t;
            tailq.push_back(k);
        }
        for (size_t i : ids) st[i]->parts_left.store((st[i]->whole ? 1 : 0) + (st[i]->tail ? 1 : 0));
        fin_list.reserve(ids.size());
        // (mu held) a part of the stream is through; the last one makes it ready to be finished -- tasks of up to 32 streams
        auto part_done = [&](size_t id, bool flush) {
            if (id != (size_t)-1 && st[id]->parts_left.fetch_sub(1) == 1) fin_list.push_back(id);
            while (fin_list.size() - fin_queued >= 32 || (flush && fin_list.size() > fin_queued)) {
                Task k;
                k.kind = Task::FIN;
                k.i0 = fin_queued;
                k.i1 = std::min(fin_list.size(), fin_queued + 32);
                fin_queued = k.i1;
                finq.push_back(k);
            }
        };
        // (mu held) hand the free slots to the next batches and queue their packing, a few segments per task
        auto assign = [&]() {
            while (!free_slots.empty() && next_assign < batches.size()) {
                Batch &b = batches[next_assign];
                b.slot = free_slots.back();
                free_slots.pop_back();
                const size_t ns = b.segs.size();
                size_t i0 = 0;
                uint32_t ntasks = 0;
                std::vector<Task> ts;
                // packing tasks: a batch is cut into about twice as many as there are workers (memory-bound work: no gain from size; some
                // workers wait for the GPU or an MD5 chain at any time), 32 ..
                // 256 frames each
                const uint32_t fr_target = std::min(256u, std::max(32u, (b.frames + 2 * nt - 1) / (2 * nt)));
                while (i0 < ns) {
                    size_t i1 = i0;
                    uint32_t fr = 0;
                    while (i1 < ns && fr < fr_target) fr += b.segs[i1++].n;
                    Task k;
                    k.kind = Task::PACK;
                    k.batch = next_assign;
                    k.i0 = i0;
                    k.i1 = i1;
                    ts.push_back(k);
                    ntasks++;
                    i0 = i1;
                }
                b.pack_left.store(ntasks);
                for (auto &k : ts) packq.push_back(k);
                next_assign++;
            }
        };
        // a stream's MD5 runs in stream order: a segment packed ahead of its predecessors waits in the stream's list
        auto md5_push = [&](Stream &s, uint64_t first, uint32_t frames, const uint8_t *p, size_t n, std::atomic<uint64_t> *ticket) {
            std::lock_guard<std::mutex> l(s.mu);
            if (!s.attach_tried) {   // the chain's lane on the shared engines, with its first run
                s.attach_tried = true;
                s.lane = Md5Pool::get().attach(&s.md5);
            }
            if (!s.lane) {   // (no engine lane: hash here, in order all the same)
                s.pending.emplace(first, Stream::Pending{p, n, frames, ticket});
                for (auto it = s.pending.begin(); it != s.pending.end() && it->first == s.next_md5_frame; it = s.pending.erase(it)) {
                    s.md5.update(it->second.p, it->second.n);
                    s.next_md5_frame += it->second.frames;
                    if (it->second.ticket) it->second.ticket->store(~0ull, std::memory_order_release);
                }
                return;
            }
            s.pending.emplace(first, Stream::Pending{p, n, frames, ticket});
            for (auto it = s.pending.begin(); it != s.pending.end() && it->first == s.next_md5_frame; it = s.pending.erase(it)) {
                s.last_ticket = Md5Pool::get().push(s.lane, it->second.p, it->second.n);
                s.next_md5_frame += it->second.frames;
                if (it->second.ticket) it->second.ticket->store(s.last_ticket, std::memory_order_release);
            }
        };
        const bool no_md5 = trace_on() && std::getenv("FLACENC_TIMING_NO_MD5");
        // PACK: the callers' samples of a few segments into the slot's input at the upload's width; the runs of the longer
        // streams go to the MD5 engines as they are packed, the last packer of the batch submits it and queues its HASH tasks
        auto run_pack = [&](const Task &k) {
            Batch &b = batches[k.batch];
            RingSlot &slot = slots[(size_t)b.slot];
            const double t0 = now_ms();
            for (size_t i = k.i0; i < k.i1; i++) {
                Seg &sg = b.segs[i];
                Stream &s = *st[sg.stream];
                const flacenc_job &j = jobs[s.job];
                const int32_t *src = j.samples + sg.first * per;
                const size_t count = (size_t)sg.n * per;
                uint8_t *dst = slot.in + (size_t)sg.boff * per * up_width;
                const uint8_t *le = dst;
                if (up_width == width) {
                    pack_le(src, count, width, dst);
                } else {   // the upload takes int32 only: the MD5 bytes are packed beside it, into the stream's own string
                    std::memcpy(dst, src, count * 4);
                    uint8_t *l2 = s.le.data() + sg.first * per * width;
                    pack_le(src, count, width, l2);
                    le = l2;
                }
                if (no_md5) sg.ticket.store(~0ull, std::memory_order_release);   // (FLACENC_TIMING_NO_MD5: wrong digests, timing only)
                else if (!s.solo) md5_push(s, sg.first, sg.n, le, count * width, up_width == width ? &sg.ticket : nullptr);
                if (!no_md5 && !s.solo && up_width != width) sg.ticket.store(~0ull, std::memory_order_release);   // (the slot's input is not what the MD5 reads)
            }
            const double dt = now_ms() - t0;
            for (size_t i = k.i0; i < k.i1; i++) {
                Stream &s = *st[b.segs[i].stream];
                std::lock_guard<std::mutex> l(s.mu);
                jobs[s.job].pack_ms += dt / (double)(k.i1 - k.i0);
            }
            if (b.pack_left.fetch_sub(1) != 1) return;
            // the last packer of the batch submits it
            std::vector<flacgpu_segment> gs(b.segs.size());
            for (size_t i = 0; i < b.segs.size(); i++) {
                gs[i].pcm = nullptr;
                gs[i].n_frames = b.segs[i].n;
                gs[i].reserved = 0;
                gs[i].first_frame_number = b.segs[i].first;
                if (!no_md5 && st[b.segs[i].stream]->solo) b.solo_idx.push_back((uint32_t)i);
            }
            const int rc = flacgpu_encode_segments_packed_async_host(slot.ctx, slot.in, up_width, gs.data(), (uint32_t)gs.size(), sh.rate,
                                                                     slot.out, slot.out_cap);
            if (trace_on()) std::fprintf(stderr, "[coalesce] batch %zu (%u frames, %zu segments) submitted at %.2f ms\n", k.batch, b.frames, b.segs.size(), now_ms() - t_begin);
            // HASH tasks: the batch's solo streams, up to 48 chains each
            std::vector<Task> hs;
            for (size_t j0 = 0; j0 < b.solo_idx.size(); j0 += 48) {
                Task h;
                h.kind = Task::HASH;
                h.batch = k.batch;
                h.i0 = j0;
                h.i1 = std::min(b.solo_idx.size(), j0 + 48);
                hs.push_back(h);
            }
            b.hash_left.store((uint32_t)hs.size());
            bool release_now = false;
            {
                std::lock_guard<std::mutex> l(mu);
                b.failed = rc != 0;
                b.submitted = true;
                for (auto &h : hs) hashq.push_back(h);
            }
            if (hs.empty()) release_now = b.release_left.fetch_sub(1) == 1;
            cv.notify_all();
            if (release_now) release_slot_fn(k.batch);
        };
        // HASH: the chains of up to 48 solo streams of a batch in lockstep (md5_blocks_groups) over the bytes packed into the
        // slot -- each chain is ONE run from the state RFC 1321 starts with: no engine thread, no queue, no lock.  A chain's
        // speed is bound by the latency of its 64 dependent steps per block, so the lanes count, not the bytes: the tasks are
        // as wide as the registers allow and take about 40 us per block of the longest stream.  Lanes of different lengths
        // drop out as they end.
        auto run_hash = [&](const Task &k) {
            Batch &b = batches[k.batch];
            RingSlot &slot = slots[(size_t)b.slot];
            const size_t frame_bytes = per * width;
            struct Lane {
                Seg *sg;
                const uint8_t *p;
                uint32_t done;
            };
            Lane lanes[48];
            Lane *live[48];
            size_t nl = 0;
            for (size_t j = k.i0; j < k.i1; j++) {
                Seg &sg = b.segs[b.solo_idx[j]];
                const uint8_t *p = up_width == width ? slot.in + (size_t)sg.boff * per * up_width
                                                     : st[sg.stream]->le.data() + sg.first * per * width;
                lanes[nl] = Lane{&sg, p, 0};
                live[nl] = &lanes[nl];
                nl++;
            }
            alignas(64) uint32_t stt[3][4][16];
            while (nl) {
                // as many blocks at a time as every live lane still has
                uint32_t common = ~0u;
                for (size_t i = 0; i < nl; i++) common = std::min(common, live[i]->sg->n - live[i]->done);
                const size_t bytes = (size_t)common * frame_bytes;
                if (nl >= 2 && bytes % 64 == 0) {
                    const int groups = (int)((nl + 15) / 16);
                    const uint8_t *ptr[3][16];
                    uint32_t mask[3] = {0, 0, 0};
                    for (size_t i = 0; i < nl; i++) {
                        const int g = (int)(i % groups), l = (int)(i / groups);
                        uint32_t w4[4];
                        st[live[i]->sg->stream]->md5.get_state(w4);
                        for (int w = 0; w < 4; w++) stt[g][w][l] = w4[w];
                        ptr[g][l] = live[i]->p + (size_t)live[i]->done * frame_bytes;
                        mask[g] |= 1u << l;
                    }
                    md5_blocks_groups(stt, ptr, bytes / 64, mask, groups);
                    for (size_t i = 0; i < nl; i++) {
                        const int g = (int)(i % groups), l = (int)(i / groups);
                        const uint32_t w4[4] = {stt[g][0][l], stt[g][1][l], stt[g][2][l], stt[g][3][l]};
                        Md5 &m = st[live[i]->sg->stream]->md5;
                        m.set_state(w4);
                        m.add_blocks(bytes / 64);
                    }
                } else {   // a lone chain, or runs that are not whole MD5 blocks: the scalar code
                    for (size_t i = 0; i < nl; i++) st[live[i]->sg->stream]->md5.update(live[i]->p + (size_t)live[i]->done * frame_bytes, bytes);
                }
                size_t w = 0;
                for (size_t i = 0; i < nl; i++) {
                    Lane *l = live[i];
                    l->done += common;
                    if (l->done < l->sg->n) {
                        live[w++] = l;
                        continue;
                    }
                    Stream &s = *st[l->sg->stream];
                    {   // the stream's short last block (if its task ran first) follows in order
                        std::lock_guard<std::mutex> lk(s.mu);
                        s.next_md5_frame += l->sg->n;
                        for (auto it = s.pending.begin(); it != s.pending.end() && it->first == s.next_md5_frame; it = s.pending.erase(it)) {
                            s.md5.update(it->second.p, it->second.n);
                            s.next_md5_frame += it->second.frames;
                        }
                    }
                    l->sg->ticket.store(~0ull, std::memory_order_release);
                }
                nl = w;
            }
            if (b.hash_left.fetch_sub(1) == 1 && b.release_left.fetch_sub(1) == 1) release_slot_fn(k.batch);
        };
        // (any thread) the batch's frames are in their places: its slot is free once the MD5 runs that read its input are done
        auto release_slot = [&](size_t bi) {
            Batch &b = batches[bi];
            for (Seg &sg : b.segs) {
                uint64_t t;
                while ((t = sg.ticket.load(std::memory_order_acquire)) == 0) std::this_thread::yield();   // (pushed by now: every earlier batch is packed)
                Stream &s = *st[sg.stream];
                if (t != ~0ull && s.lane) Md5Pool::get().wait(s.lane, t);
            }
            if (trace_on()) std::fprintf(stderr, "[coalesce] batch %zu's slot free at %.2f ms\n", bi, now_ms() - t_begin);
            {
                std::lock_guard<std::mutex> l(mu);
                free_slots.push_back(b.slot);
                batches_done++;
                assign();
                // the streams whose last whole block was in this batch: frames in place, MD5 runs of the whole blocks done
                for (Seg &sg : b.segs)
                    if (st[sg.stream]->last_batch == bi && sg.first + sg.n == st[sg.stream]->whole) part_done(sg.stream, false);
                part_done((size_t)-1, batches_done == batches.size());
            }
            cv.notify_all();
        };
        release_slot_fn = release_slot;
        auto run_retire = [&](size_t bi) {
            Batch &b = batches[bi];
            RingSlot &slot = slots[(size_t)b.slot];
            const double t0 = now_ms();
            const uint64_t *off = nullptr;
            uint64_t total = 0;
            int rc = b.failed ? FLACGPU_ERR_HIP : flacgpu_frames_ready(slot.ctx, &off, &total);
            if (!rc) rc = flacgpu_fetch_frames_async(slot.ctx, slot.out, slot.out_cap);   // (nothing to copy when k_frame64 stored them there)
            if (!rc) rc = flacgpu_wait(slot.ctx);
            else if (!b.failed) (void)flacgpu_wait(slot.ctx);
            const double dt = now_ms() - t0;
            if (trace_on()) std::fprintf(stderr, "[coalesce] batch %zu retired at %.2f ms (waited %.2f)\n", bi, now_ms() - t_begin, dt);
            std::vector<CopyJob> &cj = copy_jobs[bi];
            uint32_t f = 0;
            for (Seg &sg : b.segs) {
                Stream &s = *st[sg.stream];
                flacenc_job &j = jobs[s.job];
                if (rc) {
                    fail(s.job, FLACENC_ERR_GPU);
                } else {
                    const uint64_t bytes = off[f + sg.n] - off[f];
                    for (uint32_t k = 0; k < sg.n; k++) s.sizes[sg.first + k] = (uint32_t)(off[f + k + 1] - off[f + k]);
                    if (s.hlen + s.pos + bytes > j.out_cap) {
                        fail(s.job, FLACENC_ERR_IO);
                    } else if (!has_failed(s.job)) {
                        cj.push_back(CopyJob{slot.out + off[f], j.out + s.hlen + s.pos, (size_t)bytes});
                    }
                    s.pos += bytes;
                    j.gpu_ms += dt * sg.n / (double)b.frames;
                }
                f += sg.n;
            }
            // ~1 MB of frames per copy task
            std::vector<Task> ts;
            size_t i0 = 0;
            while (i0 < cj.size()) {
                size_t i1 = i0, bytes = 0;
                while (i1 < cj.size() && bytes < (size_t(1) << 20)) bytes += cj[i1++].n;
                Task k;
                k.kind = Task::COPY;
                k.batch = bi;
                k.i0 = i0;
                k.i1 = i1;
                ts.push_back(k);
                i0 = i1;
            }
            b.copy_left.store((uint32_t)ts.size());
            {
                std::lock_guard<std::mutex> l(mu);
                for (auto &k : ts) copyq.push_back(k);
                retiring = false;
                retire_next++;
            }
            cv.notify_all();
            if (ts.empty() && b.release_left.fetch_sub(1) == 1) release_slot(bi);
        };
        auto run_copy = [&](const Task &k) {
            const std::vector<CopyJob> &cj = copy_jobs[k.batch];
            for (size_t i = k.i0; i < k.i1; i++) std::memcpy(cj[i].dst, cj[i].src, cj[i].n);
            if (batches[k.batch].copy_left.fetch_sub(1) == 1 && batches[k.batch].release_left.fetch_sub(1) == 1) release_slot(k.batch);
        };
        // a stream's short last block: one frame, synchronously, on a one-frame context (one per worker that meets a tail)
        auto run_tail = [&](size_t id, RingSlot &tslot, bool &have) {
            Stream &s = *st[id];
            const flacenc_job &j = jobs[s.job];
            const int32_t *src = j.samples + s.whole * per;
            const size_t count = (size_t)s.tail * sh.ch;
            s.tail_le.resize(count * width);
            pack_le(src, count, width, s.tail_le.data());
            md5_push(s, s.whole, 1, s.tail_le.data(), s.tail_le.size(), nullptr);
            int rc = 0;
            if (!have) {
                rc = RingPool::get().take(tail_key, per * 4, &tslot);
                have = rc == 0;
            }
            uint64_t total = 0, off2[2] = {0, 0};
            if (!rc)
                rc = flacgpu_encode_frames(tslot.ctx, src, FLACGPU_LAYOUT_INTERLEAVED, 1, s.tail, s.whole, sh.rate, tslot.out, tslot.out_cap,
                                           off2, &total);
            if (rc) {
                fail(s.job, FLACENC_ERR_GPU);
            } else {
                s.sizes[s.whole] = (uint32_t)total;
                s.tail_bytes.assign(tslot.out, tslot.out + total);
            }
            {
                std::lock_guard<std::mutex> l(mu);
                tails_left--;
                part_done(id, tails_left == 0);
            }
            cv.notify_all();
        };

        // ---- a stream whose parts are through: digest, metadata from the frame sizes, the tail frame behind the whole blocks' frames
        auto finish_stream = [&](size_t id) {
            Stream &s = *st[id];
            flacenc_job &j = jobs[s.job];
            Md5Lane *lane;
            uint64_t ticket;
            {
                std::lock_guard<std::mutex> l(s.mu);
                lane = s.lane;
                ticket = s.last_ticket;
                s.lane = nullptr;
            }
            if (lane) {
                Md5Pool::get().wait(lane, ticket);   // the whole chain
                j.md5_ms = Md5Pool::get().busy_ms(lane);
                Md5Pool::get().detach(lane);
            }
            if (has_failed(s.job)) return;
            uint8_t digest[16];
            s.md5.digest(digest);
            size_t hlen = 0;
            const uint32_t last_len = s.tail ? s.tail : B;
            int rc = flacenc_stream_header(&o, sh.rate, sh.bps, sh.ch, s.pcm_frames, digest, s.sizes.size(), s.sizes.data(), last_len, j.out,
                                           j.out_cap, &hlen);
            if (!rc && hlen != s.hlen) rc = FLACENC_ERR_IO;   // (cannot happen: the header's size is fixed at `new`)
            if (rc || s.hlen + s.pos + s.tail_bytes.size() > j.out_cap) {
                fail(s.job, rc && rc != FLACENC_ERR_INVALID_ARG ? rc : FLACENC_ERR_IO);
                return;
            }
            if (!s.tail_bytes.empty()) std::memcpy(j.out + s.hlen + s.pos, s.tail_bytes.data(), s.tail_bytes.size());
            j.out_len = s.hlen + s.pos + s.tail_bytes.size();
            j.elapsed_ms = now_ms() - t_begin;
        };
        // ---- the workers (a stream's MD5 lane is attached with its first run)
        if (up_width != width)
            for (size_t i : ids) st[i]->le.resize((size_t)st[i]->whole * per * width);
        {
            std::lock_guard<std::mutex> l(mu);
            assign();
        }
        if (trace_on()) std::fprintf(stderr, "[coalesce] workers start at %.2f ms\n", now_ms() - t_begin);
        std::atomic<unsigned> next_wid{0};
        auto work = [&]() {
            const unsigned wid = next_wid.fetch_add(1);
            RingSlot tslot;
            bool have_tslot = false;
            for (;;) {
                Task k;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    for (;;) {
                        if (!copyq.empty()) {
                            k = copyq.front();
                            copyq.pop_front();
                            break;
                        }
                        if (!retiring && retire_next < batches.size() && batches[retire_next].submitted) {
                            retiring = true;
                            k.kind = Task::RETIRE;
                            k.batch = retire_next;
                            break;
                        }
                        if (!packq.empty()) {
                            k = packq.front();
                            packq.pop_front();
                            break;
                        }
                        if (!hashq.empty()) {
                            k = hashq.front();
                            hashq.pop_front();
                            break;
                        }
                        if (!tailq.empty()) {
                            k = tailq.front();
                            tailq.pop_front();
                            break;
                        }
                        if (!finq.empty()) {
                            k = finq.front();
                            finq.pop_front();
                            break;
                        }
                        if (fin_left == 0) {
                            k.kind = Task::NONE;
                            break;
                        }
                        cv.wait(lk);
                    }
                }
                if (k.kind == Task::NONE) break;
                const double tt0 = trace_on() ? now_ms() : 0.0;
                if (k.kind == Task::PACK) run_pack(k);
                else if (k.kind == Task::COPY) run_copy(k);
                else if (k.kind == Task::RETIRE) run_retire(k.batch);
                else if (k.kind == Task::HASH) run_hash(k);
                else if (k.kind == Task::TAIL) run_tail(k.i0, tslot, have_tslot);
                else {
                    for (size_t i = k.i0; i < k.i1; i++) {
                        size_t id;
                        {
                            std::lock_guard<std::mutex> l(mu);   // (fin_list grows under mu)
                            id = fin_list[i];
                        }
                        finish_stream(id);
                    }
                    bool last;
                    {
                        std::lock_guard<std::mutex> l(mu);
                        fin_left -= k.i1 - k.i0;
                        last = fin_left == 0;
                    }
                    if (last) cv.notify_all();
                }
                if (trace_on() && std::getenv("FLACENC_TRACE_TASKS"))
                    std::fprintf(stderr, "[task] w%u %s b%zu [%zu,%zu) %.3f -> %.3f\n", wid, k.kind == Task::PACK ? "pack" : k.kind == Task::COPY ? "copy" : k.kind == Task::RETIRE ? "retire" : k.kind == Task::TAIL ? "tail" : k.kind == Task::HASH ? "hash" : "fin",
                                 k.batch, k.i0, k.i1, tt0 - t_begin, now_ms() - t_begin);
            }
            if (have_tslot) RingPool::get().give(tail_key, tslot);
        };
        const unsigned workers = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, total_whole / 64 + tails.size() + 1));
        run_parallel(workers - 1, work);
        if (trace_on()) std::fprintf(stderr, "[coalesce] %u workers done at %.2f ms\n", workers, now_ms() - t_begin);
        for (auto &s : slots) RingPool::get().give(key, s);

        if (trace_on()) std::fprintf(stderr, "[coalesce] finished at %.2f ms\n", now_ms() - t_begin);
    }
    if (trace_on())
        std::fprintf(stderr, "[coalesce] call done at %.2f ms, %.1f CPU-ms of the process (%u usable CPUs, %u workers)\n", now_ms() - t_begin,
                     cpu_ms() - cpu_begin, cpus, nt);
    int first_error = 0;
    for (size_t i = 0; i < n_jobs; i++)
        if (jobs[i].status && !first_error) first_error = jobs[i].status;
    return first_error;
}

}  // extern "C"
