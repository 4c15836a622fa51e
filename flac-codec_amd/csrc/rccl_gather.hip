// rccl_gather.hip -- the ONE cross-GPU exchange of the path for the one-process-per-GPU shape, behind the C ABI:
// an all-gather of every rank's {frames, bytes, min_frame, max_frame} record over a caller-supplied RCCL communicator
// (what flac_codec_amd/parallel.py's all_gather_counters does through torch.distributed; encode.rs:1999-2003 seek-point
// offsets, :2414-2436 min / max frame size).  librccl is loaded lazily with dlopen: a process that never calls this
// never maps it, and its absence is an error code (FLACGPU_ERR_NO_RCCL), not a fallback.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>

#include "flacenc_gpu.h"
#include "kernels/types.h"

namespace {
// the few RCCL entry points used, by their C signatures (rccl.h: ncclResult_t is an int enum, ncclSuccess == 0,
// ncclUint64 == 5, ncclComm_t an opaque pointer)
using nccl_allgather_fn = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
using nccl_count_fn = int (*)(void *, int *);
using nccl_device_fn = int (*)(void *, int *);
using nccl_errstr_fn = const char *(*)(int);
struct Rccl {
    void *handle = nullptr;
    nccl_allgather_fn all_gather = nullptr;
    nccl_count_fn count = nullptr, rank = nullptr;
    nccl_device_fn device = nullptr;
    nccl_errstr_fn errstr = nullptr;
    std::string why;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char *e = dlerror();
            r.why = std::string("librccl not found: ") + (e ? e : "?");
            return;
        }
        r.all_gather = reinterpret_cast<nccl_allgather_fn>(dlsym(r.handle, "ncclAllGather"));
        r.count = reinterpret_cast<nccl_count_fn>(dlsym(r.handle, "ncclCommCount"));
        r.rank = reinterpret_cast<nccl_count_fn>(dlsym(r.handle, "ncclCommUserRank"));
        r.device = reinterpret_cast<nccl_device_fn>(dlsym(r.handle, "ncclCommCuDevice"));
        r.errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.all_gather || !r.count || !r.rank || !r.device) {
            r.why = "librccl lacks ncclAllGather / ncclCommCount / ncclCommUserRank / ncclCommCuDevice";
            r.all_gather = nullptr;
        }
    });
    return r;
}
constexpr int kNcclUint64 = 5;
// the call's device scratch -- the receive area and this rank's record -- kept per device between calls (a hipMalloc /
// hipFree pair per call is two device-wide synchronisations for a 32-byte exchange); grown when a larger world shows up
struct Scratch {
    uint64_t *d = nullptr;
    size_t records = 0;
};
std::mutex g_scratch_mu;
std::map<int, Scratch> g_scratch;
}  // namespace

#define RCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        const int e_ = (expr);                                                                          \
        if (e_ != 0) {                                                                                  \
            g_last_error = std::string(#expr) + ": " + (R.errstr ? R.errstr(e_) : "RCCL error");        \
            return FLACGPU_ERR_HIP;                                                                     \
        }                                                                                               \
    } while (0)

extern "C" {

int flacgpu_rccl_available(void) { return rccl().all_gather ? 1 : 0; }

int flacgpu_rccl_allgather_counters(void *nccl_comm, void *stream, const flacgpu_shard_counters *mine,
                                    flacgpu_shard_counters *all, uint32_t cap_ranks, uint32_t *n_ranks, uint32_t *my_rank) {
    if (!nccl_comm || !mine || !all) return FLACGPU_ERR_INVALID_ARG;
    Rccl &R = rccl();
    if (!R.all_gather) {
        g_last_error = R.why;
        return FLACGPU_ERR_NO_RCCL;
    }
    int world = 0, rank = 0, dev = 0;
    RCCL_TRY(R.count(nccl_comm, &world));
    RCCL_TRY(R.rank(nccl_comm, &rank));
    RCCL_TRY(R.device(nccl_comm, &dev));
    if (n_ranks) *n_ranks = (uint32_t)world;
    if (my_rank) *my_rank = (uint32_t)rank;
    if (world <= 0 || (uint32_t)world > cap_ranks) return FLACGPU_ERR_BUFFER_TOO_SMALL;
    // the communicator's device is made current for the call (and the caller's restored), like every context entry point
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    if (prev != dev) HIP_TRY(hipSetDevice(dev));
    struct Restore {
        int prev, dev;
        ~Restore() {
            if (prev != dev && prev >= 0) (void)hipSetDevice(prev);
        }
    } restore{prev, dev};
    static_assert(sizeof(flacgpu_shard_counters) == 4 * sizeof(uint64_t), "four 64-bit integers per rank");
    const size_t rec = sizeof(flacgpu_shard_counters);
    std::lock_guard<std::mutex> scratch_lock(g_scratch_mu);   // (one exchange per process at a time: the scratch is shared)
    Scratch &sc = g_scratch[dev];
    if (sc.records < (size_t)world + 1) {
        if (sc.d) (void)hipFree(sc.d);
        sc.d = nullptr;
        sc.records = 0;
        const size_t want = std::max<size_t>((size_t)world + 1, 65);
        HIP_TRY(hipMalloc(&sc.d, rec * want));
        sc.records = want;
    }
    uint64_t *d_buf = sc.d;    // [world + 1] records: the receive area, then this rank's record
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = FLACGPU_OK;
    do {
        uint64_t *d_mine = d_buf + 4 * (size_t)world;
        if (hipMemcpyAsync(d_mine, mine, rec, hipMemcpyHostToDevice, st) != hipSuccess) { rc = FLACGPU_ERR_HIP; break; }
        const int e = R.all_gather(d_mine, d_buf, 4, kNcclUint64, nccl_comm, st);
        if (e != 0) {
            g_last_error = std::string("ncclAllGather: ") + (R.errstr ? R.errstr(e) : "RCCL error");
            rc = FLACGPU_ERR_HIP;
            break;
        }
        if (hipMemcpyAsync(all, d_buf, rec * (size_t)world, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = FLACGPU_ERR_HIP; break; }
        if (hipStreamSynchronize(st) != hipSuccess) { rc = FLACGPU_ERR_HIP; break; }
    } while (0);
    if (rc == FLACGPU_ERR_HIP && g_last_error.empty()) g_last_error = "flacgpu_rccl_allgather_counters: HIP copy failed";
    return rc;
}

}  // extern "C"
