// flacenc_gpu.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the FLAC encode hot path.
//
// Written for gfx950 only: wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).
// -ffp-contract=off is REQUIRED: the f64 analysis must round exactly like the
// reference (Rust never contracts a*b+c; it uses mul_add only where written).
//
// Kernel inventory (reference function each one replaces, /root/reference/src):
//   K0 k_deinterleave  audio.rs:190-199  Frame::fill_from_samples
//   K1 k_stereo_stats  encode.rs:2463-2674 correlate_channels (fast mode only)
//   K2 k_fixed         encode.rs:2849-2898 (wasted bits), 3020-3088 encode_fixed_subframe,
//                      3747-3962 write_residuals' search (exact bit count instead of recording)
//   K3 k_autocorr      encode.rs:1785-1801 Window::apply + 3478-3501 autocorrelate
//                      (exact reference summation order: one sequential f64 chain per lag)
//   K4 k_lpc           encode.rs:3536-3580 lp_coefficients, 3656-3702 compute_best_order,
//                      3334-3401 quantize
//   K5 k_fir           encode.rs:3174-3203 encode_residuals + write_residuals' search +
//                      2929-2979 fixed/LPC/verbatim choice
//   K6 k_decide        encode.rs:2747-2786 / 2803-2835 channel-assignment choice
//   K7 k_emit          residual signal of the chosen subframes (FIR / fixed), the data
//                      `BitRecorder::playback` would replay
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <type_traits>
#include <vector>

#include "flacenc_gpu.h"

namespace {

constexpr int WG = 256;          // threads per workgroup (4 wave64)
constexpr int MAXP = 6;          // max effective partition order (64 partitions, encode.rs:3756)
constexpr int NLEAF = 1 << MAXP;
constexpr int NNODE = 2 * NLEAF - 1;

thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);           \
            return FLACGPU_ERR_HIP;                                                     \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------
// device-side records
// ---------------------------------------------------------------------------------
typedef flacgpu_subframe_plan SubPlan;  // same layout on both sides of the ABI

struct CandInfo {        // per (frame, candidate)
    uint8_t active;      // 0: not a candidate for this frame (fast correlation / no mid)
    uint8_t wasted;
    uint8_t bps;         // effective bps after wasted-bit removal
    uint8_t is_const;    // all samples zero -> CONSTANT, nothing else to analyse
};

struct LpcParams {       // per (frame, candidate), output of k_lpc
    int32_t status;      // 0 ok; else the reference's error (1 Insufficient, 2 NoBestOrder,
                         // 3 ZeroCoeffs, 4 NegativeShift)
    uint8_t order, precision, shift, pad;
    int32_t qlp[FLACGPU_MAX_LPC_ORDER];
};

struct FrameInfo {       // per frame, from k_stereo_stats (fast mode) -- preset assignment
    uint8_t assignment;
    uint8_t pad[3];
};

struct Params {
    // stream shape / options
    uint32_t channels, bps, block_size, ldb;      // ldb = row stride of the planar buffer
    uint32_t ncand;                                // candidate slots per frame
    uint32_t stereo4;                              // 1: slots are L,R,M,S
    uint32_t mid_side, exhaustive;
    uint32_t max_lpc_order, max_po, use_rice2;
    uint32_t n_frames, last_len;
    uint32_t f0, fcount;                           // frames [f0, f0 + fcount) handled by this launch
    uint32_t dbg;                                  // timing experiments only (FLACGPU_DEBUG)
    uint32_t ac_split;                             // waves the lags of k_autocorr3 are split over (2 or 4)
    // buffers
    const int32_t *planar;
    const double *window_full, *window_last;
    const double *log2_thr;                        // [128], index e + 64
    CandInfo *cinfo;
    SubPlan *fixed_plan, *cand_plan, *out_plan;
    LpcParams *lpc;
    double *ac;                                    // [n_frames*ncand][36]
    FrameInfo *finfo;
    flacgpu_frame_plan *frame_plan;
    int32_t *residuals;                            // [n_frames][channels][block_size]
    uint32_t *stats;                               // [4]
};

__device__ __forceinline__ uint32_t frame_len(const Params &p, uint32_t frame) {
    return (frame + 1 == p.n_frames) ? p.last_len : p.block_size;
}

// XCD-aware (frame, candidate) <- blockIdx mapping: consecutive blockIdx values are dealt
// round-robin over the 8 XCDs, so put the candidates of one frame 8 blocks apart: they
// then share an XCD L2 and the frame's L/R rows are fetched from HBM once.
__device__ __forceinline__ void map_block(uint32_t bid, uint32_t nc, uint32_t nframes,
                                          uint32_t &frame, uint32_t &cand) {
    uint32_t full = nframes / 8u;
    uint32_t per = 8u * nc;
    if (bid < full * per) {
        uint32_t g = bid / per, rem = bid - g * per;
        cand = rem / 8u;
        frame = g * 8u + (rem & 7u);
    } else {
        uint32_t rem = bid - full * per, tail = nframes - full * 8u;
        cand = rem / tail;
        frame = full * 8u + (rem - cand * tail);
    }
}

// candidate -> source rows.  mode 0: a;  1: (a+b)>>1 (mid);  2: a-b (side)
struct CandSrc {
    const int32_t *a, *b;
    int mode;
    uint32_t bps;
    uint8_t source;
};
__device__ __forceinline__ CandSrc cand_src(const Params &p, uint32_t frame, uint32_t cand) {
    CandSrc s;
    const int32_t *base = p.planar + (size_t)frame * p.channels * p.ldb;
    s.bps = p.bps;
    if (p.stereo4 && cand >= 2) {
        s.a = base;
        s.b = base + p.ldb;
        if (cand == 2) {
            s.mode = 1;
            s.source = FLACGPU_SRC_MID;
        } else {
            s.mode = 2;
            s.bps = p.bps + 1;
            s.source = FLACGPU_SRC_SIDE;
        }
    } else {
        s.a = base + (size_t)cand * p.ldb;
        s.b = s.a;
        s.mode = 0;
        s.source = (uint8_t)cand;
    }
    return s;
}
__device__ __forceinline__ int32_t combine(int mode, int32_t a, int32_t b) {
    // encode.rs:2721 `(l + r) >> 1`, :2734 `l - r` (i32, wrapping in release builds)
    if (mode == 1) return (int32_t)((uint32_t)a + (uint32_t)b) >> 1;
    if (mode == 2) return (int32_t)((uint32_t)a - (uint32_t)b);
    return a;
}

// ---------------------------------------------------------------------------------
// workgroup reductions (wave64 shuffles, then 4-way LDS combine)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_down(v, off, 64);
    return v;
}
// all threads get the result; `scratch` holds >= 4 u64
__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t *scratch) {
    v = wave_sum_u64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ uint32_t block_or_u32(uint32_t v, uint64_t *scratch) {
    v = wave_or_u32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (uint32_t)(scratch[0] | scratch[1] | scratch[2] | scratch[3]);
}

__device__ __forceinline__ uint32_t uabs(int32_t v) {
    return v < 0 ? 0u - (uint32_t)v : (uint32_t)v;  // i32::unsigned_abs
}
__device__ __forceinline__ uint32_t zigzag(int32_t s) {  // encode.rs:3845-3849
    return ((uint32_t)s << 1) ^ (uint32_t)(s >> 31);
}

// ---------------------------------------------------------------------------------
// Rice partition search (encode.rs:3865-3896 best_partitions + 3765-3831 Partition::new)
// and exact residual-block bit count (replaces BitRecorder::written()).
// Cooperative over the workgroup.  `r` is indexed by SAMPLE POSITION: residual of sample i
// lives at r[i], valid for i in [order, n).  Partitions are the reference's
// `residuals.rchunks(n >> porder).rev()`, i.e. cuts at multiples of the partition length.
// The f64 `ceil(log2(sum / n))` of encode.rs:3778-3780 is replaced by its exact integer
// equivalent (smallest k with n * 2^k >= sum; valid because sum < 2^53).
// ---------------------------------------------------------------------------------
// residual arrays in LDS use a padded index (one extra dword per 16) so that lanes walking
// contiguous 16-element runs (stride 17 dwords) hit distinct banks
__device__ __forceinline__ uint32_t RIDX(uint32_t i) { return i + (i >> 4); }

struct RiceShared {
    unsigned long long leaf[NLEAF];
    unsigned long long pre[NLEAF + 1];   // exclusive prefix of the leaf sums
    uint32_t nd_cnt[NNODE + 1];
    uint8_t nd_kind[NNODE + 1], nd_rice[NNODE + 1], nd_esc[NNODE + 1];
    uint32_t lv_est[MAXP + 1], lv_count[MAXP + 1], lv_bad[MAXP + 1], lv_hi[MAXP + 1];
    uint64_t red[4];
};
enum { PK_STANDARD = 0, PK_ESCAPED = 1, PK_CONSTANT = 2 };

// Partition::new (encode.rs:3765-3831) for one partition of `cnt` residuals with sum of
// absolute values `sum`: Rice parameter / escape decision and the size ESTIMATE the reference
// ranks partition orders by.  The f64 `ceil(log2(sum / n))` of :3778-3780 is replaced by its
// exact integer equivalent (smallest k with n * 2^k >= sum; sum < 2^53).
struct PartEval {
    uint32_t est;
    uint8_t kind, rice, esc, bad;
};
__device__ __forceinline__ PartEval partition_eval(uint32_t cnt, unsigned long long sum,
                                                   uint32_t rice_max) {
    PartEval r;
    r.est = 0;
    r.kind = PK_CONSTANT;
    r.rice = 0xFF;
    r.esc = 0;
    r.bad = 0;
    if (cnt > 0 && sum > 0) {
        uint32_t k = 0;
        bool standard = true;
        if (sum > (unsigned long long)cnt) {
            const uint32_t bs = 64u - (uint32_t)__clzll((long long)sum);
            const uint32_t bc = 32u - (uint32_t)__builtin_clz(cnt);
            k = bs > bc ? bs - bc - 1 : 0;
            while (((unsigned long long)cnt << k) < sum) k++;
            if (k >= rice_max) {
                standard = false;
                const uint32_t e = (bs - 1) + 2u;  // ilog2(sum) + 2
                if (e > 31u) r.bad = 1;
                r.kind = PK_ESCAPED;
                r.esc = (uint8_t)e;
                r.est = e * cnt;
            }
        }
        if (standard) {
            const unsigned long long t = k ? (sum >> (k - 1)) : (sum << 1);
            if (t > 0xFFFFFFFFull) r.bad = 1;  // u32::try_from fails -> candidate dropped
            r.kind = PK_STANDARD;
            r.rice = (uint8_t)k;
            r.est = 4u + (1u + k) * cnt + (uint32_t)t - cnt / 2u;  // wrapping u32
        }
    }
    return r;
}

__device__ __forceinline__ uint32_t rice_levels(uint32_t n, const Params &p) {
    uint32_t tz = (uint32_t)__builtin_ctz(n);
    uint32_t P = tz < p.max_po ? tz : p.max_po;
    return P > MAXP ? MAXP : P;  // host rejects calls with P > 6; clamp defensively
}
__device__ __forceinline__ void rice_init(RiceShared &S) {
    const uint32_t tid = threadIdx.x;
    if (tid < NLEAF) S.leaf[tid] = 0ull;
    if (tid <= MAXP) {
        S.lv_est[tid] = 0;
        S.lv_count[tid] = 0;
        S.lv_bad[tid] = 0;
        S.lv_hi[tid] = 0;
    }
}

// Partition tree evaluation.  Precondition: S.leaf[] holds the sums of |residual| of the
// 2^P finest partitions and a barrier has been passed.  One lane per tree node runs
// Partition::new (encode.rs:3765-3831); the f64 `ceil(log2(sum / n))` of :3778-3780 is
// replaced by its exact integer equivalent (smallest k with n * 2^k >= sum; sum < 2^53).
// Ends with a barrier; afterwards rice_pick() gives every lane the chosen level.
__device__ __forceinline__ void rice_tree(RiceShared &S, uint32_t n, uint32_t order, uint32_t P,
                                          uint32_t rice_max) {
    const uint32_t tid = threadIdx.x;
    // heap numbering: node = 2^level + j.  Nodes 1..63 (levels 0..5) are lanes of wave 0, which
    // first builds the inclusive prefix of the leaf sums with shuffles (no barrier) and reads
    // partition sums as prefix differences; nodes 64..127 exist only for P == 6 and are the
    // leaves themselves (wave 1).
    if (tid < 128) {
        unsigned long long incl = 0;
        if (tid < 64) {
            incl = S.leaf[tid];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                unsigned long long t = __shfl_up(incl, off, 64);
                if (tid >= (uint32_t)off) incl += t;
            }
        }
        const uint32_t node = tid;
        const bool is_node = node >= 1 && node < (2u << P);
        const uint32_t lvl = is_node ? 31u - (uint32_t)__builtin_clz(node) : 0u;
        const uint32_t j = node - (1u << lvl);
        unsigned long long sum;
        if (tid < 64) {
            const uint32_t span = 1u << (P - (lvl <= P ? lvl : P));
            const uint32_t hi = (j + 1) * span - 1, lo = j * span;  // leaves [lo, hi]
            const unsigned long long top = __shfl(incl, (int)(hi & 63), 64);
            const unsigned long long bot = __shfl(incl, (int)((lo ? lo - 1 : 0) & 63), 64);
            sum = top - (lo ? bot : 0ull);
        } else {
            sum = S.leaf[node & 63];
        }
        if (is_node) {
            const uint32_t plen = n >> lvl;
            const uint32_t start = j * plen, end = start + plen;
            const uint32_t cnt = (end > order) ? end - (start > order ? start : order) : 0u;
            const PartEval pe = partition_eval(cnt, sum, rice_max);
            const uint8_t kind = pe.kind, rice = pe.rice, esc = pe.esc;
            const uint32_t est = pe.est, bad = pe.bad;
            S.nd_kind[node] = kind;
            S.nd_rice[node] = rice;
            S.nd_esc[node] = esc;
            S.nd_cnt[node] = cnt;
            if (cnt > 0) {  // chunks lying inside the warm-up are not partitions
                atomicAdd(&S.lv_est[lvl], est);
                atomicAdd(&S.lv_count[lvl], 1u);
                if (bad) atomicOr(&S.lv_bad[lvl], 1u);
                if (kind == PK_STANDARD && rice >= 15) atomicOr(&S.lv_hi[lvl], 1u);
            }
        }
    }
    __syncthreads();
}

struct RicePick {
    int bp;            // chosen partition level, -1 = 31-bit escaped fallback (encode.rs:3887)
    uint32_t count;    // partitions emitted
    uint32_t method;   // 0 RICE, 1 RICE2
    uint32_t first_j;  // chunks of level bp lying wholly inside the warm-up
};
// first minimum estimate among valid levels (encode.rs:3881 `!p.is_empty() &&
// p.len().is_power_of_two()`, :3885 min_by_key); try_reduce_rice (:3929-3942)
__device__ __forceinline__ RicePick rice_pick(const RiceShared &S, uint32_t P, uint32_t use_rice2) {
    RicePick r;
    r.bp = -1;
    uint32_t best_est = 0;
    for (uint32_t lvl = 0; lvl <= P; lvl++) {
        const uint32_t c = S.lv_count[lvl];
        const bool ok = !S.lv_bad[lvl] && c > 0 && (c & (c - 1)) == 0;
        if (ok && (r.bp < 0 || S.lv_est[lvl] < best_est)) {
            r.bp = (int)lvl;
            best_est = S.lv_est[lvl];
        }
    }
    r.count = r.bp >= 0 ? S.lv_count[r.bp] : 1u;
    r.method = (r.bp >= 0 && use_rice2 && S.lv_hi[r.bp]) ? 1u : 0u;
    r.first_j = r.bp >= 0 ? (1u << r.bp) - r.count : 0u;
    return r;
}
// header + fixed part of partition `tid` of the chosen level, and the plan's parameter arrays
__device__ __forceinline__ unsigned long long rice_partition_fixed_bits(const RiceShared &S,
                                                                        const RicePick &pk,
                                                                        SubPlan &plan) {
    const uint32_t tid = threadIdx.x, hb = pk.method ? 5u : 4u;
    if (tid >= pk.count) return 0;
    const uint32_t nd = (1u << pk.bp) + pk.first_j + tid;
    const uint32_t c = S.nd_cnt[nd];
    plan.rice[tid] = S.nd_rice[nd];
    plan.escape_bits[tid] = S.nd_esc[nd];
    if (S.nd_kind[nd] == PK_STANDARD) return hb + (1u + S.nd_rice[nd]) * c;
    if (S.nd_kind[nd] == PK_ESCAPED) return hb + 5u + (uint32_t)S.nd_esc[nd] * c;
    return hb + 5u;
}
__device__ __forceinline__ bool rice_finish(RiceShared &S, const RicePick &pk, uint32_t n,
                                            unsigned long long mine, SubPlan &plan,
                                            uint32_t &resid_bits) {
    const unsigned long long tot = block_sum_u64(mine, S.red);
    if (threadIdx.x == 0) {
        plan.part_len = pk.bp >= 0 ? n >> pk.bp : n;
        plan.coding_method = (uint8_t)pk.method;
        plan.n_partitions = pk.count;
        plan.partition_order = (uint8_t)(31u - (uint32_t)__builtin_clz(pk.count));
    }
    // coding method (2) + partition order (4) (encode.rs:3949, 3902) + partitions
    resid_bits = 6u + (uint32_t)(tot & 0xFFFFFFFFull);
    return (tot >> 52) == 0;
}

// Generic Rice search over a residual array in LDS (any block length).
// Returns false when the reference's write_residuals would fail: only possible for the
// 31-bit escaped fallback partition (encode.rs:3887-3895) when a residual does not fit 31
// bits (`write_signed_counted` errors, :3857) -- the subframe candidate is then an Err.
__device__ bool rice_search(const int32_t *r, uint32_t n, uint32_t order, const Params &p,
                            RiceShared &S, SubPlan &plan /* LDS */, uint32_t &resid_bits) {
    const uint32_t tid = threadIdx.x;
    const uint32_t P = rice_levels(n, p);
    const uint32_t leaf_len = n >> P;
    const uint32_t ept = (n + WG - 1) / WG;
    const uint32_t lo = tid * ept > order ? tid * ept : order;
    const uint32_t hi = (tid + 1) * ept < n ? (tid + 1) * ept : n;
    rice_init(S);
    __syncthreads();
    if (lo < hi) {  // this lane's contiguous run of residuals -> leaf sums
        uint32_t i = lo;
        uint32_t cur = i / leaf_len;
        uint32_t bound = (cur + 1) * leaf_len;
        unsigned long long acc = 0;
        for (; i < hi; i++) {
            if (i == bound) {
                atomicAdd(&S.leaf[cur], acc);
                acc = 0;
                cur++;
                bound += leaf_len;
            }
            acc += uabs(r[RIDX(i)]);
        }
        atomicAdd(&S.leaf[cur], acc);
    }
    __syncthreads();
    rice_tree(S, n, order, P, p.use_rice2 ? 31u : 15u);
    const RicePick pk = rice_pick(S, P, p.use_rice2);
    unsigned long long mine = 0;  // this lane's share of the residual block's bit count
    if (pk.bp >= 0) {
        mine += rice_partition_fixed_bits(S, pk, plan);
        if (lo < hi) {  // exact body bits: sum over standard partitions of (u >> k)
            const uint32_t plen = n >> pk.bp;
            uint32_t i = lo;
            uint32_t cur = i / plen;
            uint32_t bound = (cur + 1) * plen;
            uint32_t k = S.nd_rice[(1u << pk.bp) + cur];
            for (; i < hi; i++) {
                if (i == bound) {
                    cur++;
                    bound += plen;
                    k = S.nd_rice[(1u << pk.bp) + cur];
                }
                if (k != 0xFF) mine += zigzag(r[RIDX(i)]) >> k;
            }
        }
    } else {
        if (tid == 0) {
            plan.rice[0] = 0xFF;
            plan.escape_bits[0] = 31;  // encode.rs:3887-3895
            mine += 4u + 5u + 31u * (n - order);
        }
        for (uint32_t i = lo; i < hi; i++)
            if (r[RIDX(i)] < -(1 << 30) || r[RIDX(i)] >= (1 << 30)) mine |= 1ull << 52;  // > 31 bits
    }
    return rice_finish(S, pk, n, mine, plan, resid_bits);
}

constexpr uint32_t FN = 4096;  // the block length of every preset but `fast`

__device__ __forceinline__ void plan_clear(SubPlan &plan) {
    uint32_t *w = reinterpret_cast<uint32_t *>(&plan);
    for (uint32_t i = threadIdx.x; i < sizeof(SubPlan) / 4; i += WG) w[i] = 0;
}
__device__ __forceinline__ void plan_store(SubPlan *dst, const SubPlan &src) {
    const uint32_t *s = reinterpret_cast<const uint32_t *>(&src);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    for (uint32_t i = threadIdx.x; i < sizeof(SubPlan) / 4; i += WG) d[i] = s[i];
}

// fixed / LPC / verbatim choice of encode_subframe (encode.rs:2929-2979), thread 0 only.
// returns 0: keep `best`, 1: VERBATIM
__device__ __forceinline__ void make_verbatim(SubPlan &plan, uint32_t n, uint32_t bps_eff,
                                              uint32_t wasted, uint8_t source) {
    plan.type = FLACGPU_SUB_VERBATIM;
    plan.wasted = (uint8_t)wasted;
    plan.bps = (uint8_t)bps_eff;
    plan.order = 0;
    plan.precision = 0;
    plan.shift = 0;
    plan.coding_method = 0;
    plan.partition_order = 0;
    plan.source = source;
    plan.n_partitions = 0;
    plan.part_len = 0;
    plan.bits = 8u + wasted + n * bps_eff;
}

// ---------------------------------------------------------------------------------
// K0: de-interleave (audio.rs:190-199)  [pcm_frame][ch] -> [flac_frame][ch][ldb]
// also used to re-stride planar input whose block size is not a multiple of 4
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_deinterleave(const int32_t *__restrict__ in,
                                                     int32_t *__restrict__ out, uint32_t channels,
                                                     uint32_t block_size, uint32_t ldb,
                                                     uint32_t n_frames, uint32_t last_len,
                                                     int planar_in, uint32_t f0) {
    const uint32_t frame = f0 + blockIdx.y;
    const uint32_t n = (frame + 1 == n_frames) ? last_len : block_size;
    const size_t in_base = (size_t)frame * block_size * channels;
    int32_t *o = out + (size_t)frame * channels * ldb;
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
        for (uint32_t c = 0; c < channels; c++) {
            int32_t v = planar_in ? in[in_base + (size_t)c * n + i]
                                  : in[in_base + (size_t)i * channels + c];
            o[(size_t)c * ldb + i] = v;
        }
    }
}

// stereo fast path: both channels of an interleaved pair in one 8-byte load.  Also ORs together
// all samples of L, R, mid, side per frame (orbits[frame*ncand + c]): trailing zeros of the OR =
// wasted bits (encode.rs:2878-2898), OR == 0 = all-zero candidate.
__global__ void __launch_bounds__(WG) k_deinterleave2(const int2 *__restrict__ in,
                                                      int32_t *__restrict__ out,
                                                      uint32_t block_size, uint32_t ldb,
                                                      uint32_t n_frames, uint32_t last_len,
                                                      uint32_t *__restrict__ orbits, uint32_t ncand,
                                                      uint32_t f0) {
    const uint32_t frame = f0 + blockIdx.y;
    const uint32_t n = (frame + 1 == n_frames) ? last_len : block_size;
    const int2 *src = in + (size_t)frame * block_size;
    int32_t *o = out + (size_t)frame * 2 * ldb;
    uint32_t ol = 0, orr = 0, om = 0, os = 0;
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
        int2 v = src[i];
        o[i] = v.x;
        o[ldb + i] = v.y;
        ol |= (uint32_t)v.x;
        orr |= (uint32_t)v.y;
        om |= (uint32_t)combine(1, v.x, v.y);
        os |= (uint32_t)combine(2, v.x, v.y);
    }
    ol = wave_or_u32(ol);
    orr = wave_or_u32(orr);
    om = wave_or_u32(om);
    os = wave_or_u32(os);
    if ((threadIdx.x & 63) == 0) {
        uint32_t *ob = orbits + (size_t)frame * ncand;
        if (ol) atomicOr(&ob[0], ol);
        if (orr) atomicOr(&ob[1], orr);
        if (ncand == 4) {
            if (om) atomicOr(&ob[2], om);
            if (os) atomicOr(&ob[3], os);
        }
    }
}

// the same ORs from the planar buffer (every layout but interleaved stereo)
__global__ void __launch_bounds__(WG) k_orbits(Params p, uint32_t *__restrict__ orbits) {
    const uint32_t frame = p.f0 + blockIdx.y;
    const uint32_t n = frame_len(p, frame);
    const int32_t *base = p.planar + (size_t)frame * p.channels * p.ldb;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
        if (p.stereo4) {
            const int32_t l = base[i], r = base[p.ldb + i];
            acc[0] |= (uint32_t)l;
            acc[1] |= (uint32_t)r;
            acc[2] |= (uint32_t)combine(1, l, r);
            acc[3] |= (uint32_t)combine(2, l, r);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < 8; c++)
                if (c < p.channels) acc[c] |= (uint32_t)base[(size_t)c * p.ldb + i];
        }
    }
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t v = wave_or_u32(acc[c]);
        if ((threadIdx.x & 63) == 0 && c < p.ncand && v) atomicOr(&orbits[(size_t)frame * p.ncand + c], v);
    }
}

// interleaved input with C independent channels (ncand == C): one pass splits the channels into
// planar rows and ORs every channel's samples (orbits -> wasted bits / all-zero)
template <int C>
__global__ void __launch_bounds__(WG) k_deinterleave_n(const int32_t *__restrict__ in,
                                                       int32_t *__restrict__ out, uint32_t block_size,
                                                       uint32_t ldb, uint32_t n_frames, uint32_t last_len,
                                                       uint32_t *__restrict__ orbits, uint32_t f0) {
    const uint32_t frame = f0 + blockIdx.y;
    const uint32_t n = (frame + 1 == n_frames) ? last_len : block_size;
    const int32_t *src = in + (size_t)frame * block_size * C;
    int32_t *o = out + (size_t)frame * C * ldb;
    uint32_t acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0;
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) {
        int32_t v[C];
        if constexpr (C % 4 == 0) {
#pragma unroll
            for (int q = 0; q < C / 4; q++) {
                const int4 t = reinterpret_cast<const int4 *>(src + (size_t)i * C)[q];
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
        } else if constexpr (C % 2 == 0) {
#pragma unroll
            for (int q = 0; q < C / 2; q++) {
                const int2 t = reinterpret_cast<const int2 *>(src + (size_t)i * C)[q];
                v[2 * q] = t.x; v[2 * q + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) v[c] = src[(size_t)i * C + c];
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            o[(size_t)c * ldb + i] = v[c];
            acc[c] |= (uint32_t)v[c];
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
        const uint32_t v = wave_or_u32(acc[c]);
        if ((threadIdx.x & 63) == 0 && v) atomicOr(&orbits[(size_t)frame * C + c], v);
    }
}

// per (frame, candidate): activity, wasted bits, effective bps (encode.rs:2870-2898)
__global__ void __launch_bounds__(WG) k_candinfo(Params p, const uint32_t *__restrict__ orbits) {
    const uint32_t idx = p.f0 * p.ncand + blockIdx.x * WG + threadIdx.x;
    if (idx >= (p.f0 + p.fcount) * p.ncand) return;
    const uint32_t cand = idx % p.ncand;
    CandInfo ci;
    if (p.exhaustive || !p.stereo4)
        // exhaustive: L, R always; S if bps+1 <= 32 (guaranteed by stereo4); M iff mid_side
        ci.active = !(p.stereo4 && cand == 2 && !p.mid_side);
    else
        ci.active = p.cinfo[idx].active;  // chosen by k_stereo_stats
    const uint32_t cbps = p.bps + ((p.stereo4 && cand == 3) ? 1u : 0u);
    const uint32_t orv = orbits[idx];
    if (orv == 0) {  // all zero -> CONSTANT(0) at the candidate's bps, wasted 0
        ci.is_const = 1;
        ci.wasted = 0;
        ci.bps = (uint8_t)cbps;
    } else {
        const uint32_t w = (uint32_t)__builtin_ctz(orv);
        ci.is_const = 0;
        ci.wasted = (uint8_t)w;
        ci.bps = (uint8_t)(cbps - w);
    }
    p.cinfo[idx] = ci;
}

// ---------------------------------------------------------------------------------
// K1: correlate_channels (fast, non-exhaustive), encode.rs:2463-2674
// one workgroup per frame: abs sums of L, R, M, S -> assignment + active candidates
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_stereo_stats(Params p) {
    __shared__ uint64_t red[4];
    const uint32_t frame = p.f0 + blockIdx.x;
    const uint32_t n = frame_len(p, frame);
    const int32_t *L = p.planar + (size_t)frame * 2 * p.ldb;
    const int32_t *R = L + p.ldb;
    uint64_t ls = 0, rs = 0, ms = 0, ss = 0;
    for (uint32_t i = threadIdx.x; i < n; i += WG) {
        int32_t l = L[i], r = R[i];
        ls += uabs(l);
        rs += uabs(r);
        ms += uabs(combine(1, l, r));
        ss += uabs(combine(2, l, r));
    }
    ls = block_sum_u64(ls, red);
    rs = block_sum_u64(rs, red);
    ms = block_sum_u64(ms, red);
    ss = block_sum_u64(ss, red);
    if (threadIdx.x == 0) {
        uint8_t assign;
        if (p.mid_side) {  // candidate order :2506-2514
            uint64_t tot[4] = {ls + rs, ls + ss, ss + rs, ms + ss};
            int b = 0;
            for (int i = 1; i < 4; i++)
                if (tot[i] < tot[b]) b = i;
            assign = b == 0 ? FLACGPU_ASSIGN_INDEPENDENT
                   : b == 1 ? FLACGPU_ASSIGN_LEFT_SIDE
                   : b == 2 ? FLACGPU_ASSIGN_SIDE_RIGHT : FLACGPU_ASSIGN_MID_SIDE;
        } else {  // candidate order :2600-2607: LeftSide, SideRight, Independent
            uint64_t tot[3] = {ls + ss, ss + rs, ls + rs};
            int b = 0;
            for (int i = 1; i < 3; i++)
                if (tot[i] < tot[b]) b = i;
            assign = b == 0 ? FLACGPU_ASSIGN_LEFT_SIDE
                   : b == 1 ? FLACGPU_ASSIGN_SIDE_RIGHT : FLACGPU_ASSIGN_INDEPENDENT;
        }
        p.finfo[frame].assignment = assign;
        CandInfo *ci = p.cinfo + (size_t)frame * p.ncand;
        ci[0].active = (assign == FLACGPU_ASSIGN_INDEPENDENT || assign == FLACGPU_ASSIGN_LEFT_SIDE);
        ci[1].active = (assign == FLACGPU_ASSIGN_INDEPENDENT || assign == FLACGPU_ASSIGN_SIDE_RIGHT);
        ci[2].active = (assign == FLACGPU_ASSIGN_MID_SIDE);
        ci[3].active = (assign != FLACGPU_ASSIGN_INDEPENDENT);
    }
}

// ---------------------------------------------------------------------------------
// K2: wasted bits + FIXED predictor analysis, one workgroup per (frame, candidate)
// dynamic LDS: x[n] | r[n]
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_fixed(Params p) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ RiceShared RS;
    __shared__ SubPlan plan;
    __shared__ uint64_t red[4];
    __shared__ uint64_t sums[5];

    uint32_t frame, cand;
    map_block(blockIdx.x, p.ncand, p.fcount, frame, cand);
    frame += p.f0;
    const uint32_t n = frame_len(p, frame);
    const size_t cidx = (size_t)frame * p.ncand + cand;
    const uint32_t tid = threadIdx.x;
    const CandInfo ci = p.cinfo[cidx];  // written by k_candinfo (wasted bits, activity)
    if (!ci.active) return;
    int32_t *x = lds;
    int32_t *r = lds + p.block_size;
    const CandSrc src = cand_src(p, frame, cand);

    for (uint32_t i = tid; i < n; i += WG) x[i] = combine(src.mode, src.a[i], src.b[i]);
    plan_clear(plan);
    __syncthreads();
    // encode.rs:2878-2898: min trailing zeros over all samples (zero counts as 32), from k_candinfo
    const uint32_t wasted = ci.is_const ? 32u : ci.wasted;
    if (wasted == 32u) {  // all zero -> CONSTANT(0) at the candidate's bps, wasted 0 (:2883-2887)
        if (tid == 0) {
            plan.type = FLACGPU_SUB_CONSTANT;
            plan.bps = (uint8_t)src.bps;
            plan.source = src.source;
            plan.bits = 8u + src.bps;
        }
        __syncthreads();
        plan_store(p.fixed_plan + cidx, plan);
        plan_store(p.cand_plan + cidx, plan);
        return;
    }
    const uint32_t bps_eff = src.bps - wasted;
    if (wasted)
        for (uint32_t i = tid; i < n; i += WG) x[i] >>= wasted;
    __syncthreads();

    // available orders, encode.rs:3039-3060
    uint32_t maxo = n - 1 < 4u ? n - 1 : 4u;
    if (bps_eff >= 28 && maxo > 0) {  // first differences can overflow i32 only then
        uint32_t ovf = 0;
        for (uint32_t i = tid; i < n; i += WG) {
            long long x0 = x[i];
            if (i >= 1) {
                long long d1 = x0 - x[i - 1];
                if (d1 < INT32_MIN || d1 > INT32_MAX) ovf |= 1u;
                if (i >= 2) {
                    long long d2 = x0 - 2ll * x[i - 1] + x[i - 2];
                    if (d2 < INT32_MIN || d2 > INT32_MAX) ovf |= 2u;
                    if (i >= 3) {
                        long long d3 = x0 - 3ll * x[i - 1] + 3ll * x[i - 2] - x[i - 3];
                        if (d3 < INT32_MIN || d3 > INT32_MAX) ovf |= 4u;
                        if (i >= 4) {
                            long long d4 = x0 - 4ll * x[i - 1] + 6ll * x[i - 2] - 4ll * x[i - 3] +
                                           x[i - 4];
                            if (d4 < INT32_MIN || d4 > INT32_MAX) ovf |= 8u;
                        }
                    }
                }
            }
        }
        ovf = block_or_u32(ovf, red);
        for (uint32_t k = 1; k <= maxo; k++)
            if (ovf & (1u << (k - 1))) {
                maxo = k - 1;
                break;
            }
    }
    // abs sums over the common tail [maxo, n), encode.rs:3062-3073
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (uint32_t i = maxo + tid; i < n; i += WG) {
        long long x0 = x[i];
        s0 += uabs((int32_t)x0);
        if (maxo >= 1) {
            long long xm1 = x[i - 1];
            s1 += uabs((int32_t)(x0 - xm1));
            if (maxo >= 2) {
                long long xm2 = x[i - 2];
                s2 += uabs((int32_t)(x0 - 2 * xm1 + xm2));
                if (maxo >= 3) {
                    long long xm3 = x[i - 3];
                    s3 += uabs((int32_t)(x0 - 3 * xm1 + 3 * xm2 - xm3));
                    if (maxo >= 4) {
                        long long xm4 = x[i - 4];
                        s4 += uabs((int32_t)(x0 - 4 * xm1 + 6 * xm2 - 4 * xm3 + xm4));
                    }
                }
            }
        }
    }
    s0 = block_sum_u64(s0, red);
    s1 = block_sum_u64(s1, red);
    s2 = block_sum_u64(s2, red);
    s3 = block_sum_u64(s3, red);
    s4 = block_sum_u64(s4, red);
    if (tid == 0) {
        sums[0] = s0; sums[1] = s1; sums[2] = s2; sums[3] = s3; sums[4] = s4;
    }
    __syncthreads();
    uint32_t order = 0;
    for (uint32_t k = 1; k <= maxo; k++)
        if (sums[k] < sums[order]) order = k;  // min_by_key: first minimum wins
    // residuals of the chosen order, indexed by sample position
    for (uint32_t i = order + tid; i < n; i += WG) {
        long long v = x[i];
        if (order == 1) v = v - x[i - 1];
        else if (order == 2) v = v - 2ll * x[i - 1] + x[i - 2];
        else if (order == 3) v = v - 3ll * x[i - 1] + 3ll * x[i - 2] - x[i - 3];
        else if (order == 4) v = v - 4ll * x[i - 1] + 6ll * x[i - 2] - 4ll * x[i - 3] + x[i - 4];
        r[RIDX(i)] = (int32_t)v;
    }
    __syncthreads();
    uint32_t rbits;
    uint32_t rb_ = 0;
    const bool fixed_ok = (p.dbg & 1) ? true : rice_search(r, n, order, p, RS, plan, rb_);
    rbits = rb_;
    if (tid == 0) {
        plan.reserved[0] = fixed_ok ? 0 : 1;  // internal: FIXED candidate is an Err
        plan.type = FLACGPU_SUB_FIXED;
        plan.wasted = (uint8_t)wasted;
        plan.bps = (uint8_t)bps_eff;
        plan.order = (uint8_t)order;
        plan.source = src.source;
        plan.bits = 8u + wasted + order * bps_eff + rbits;
    }
    __syncthreads();
    plan_store(p.fixed_plan + cidx, plan);
    if (p.max_lpc_order == 0) {  // no LPC candidate: final choice here (encode.rs:2947-2979)
        __syncthreads();
        const bool verbatim = !fixed_ok || !(plan.bits < n * bps_eff);
        __syncthreads();
        if (verbatim) plan_clear(plan);
        __syncthreads();
        if (tid == 0 && verbatim) make_verbatim(plan, n, bps_eff, wasted, src.source);
        __syncthreads();
        plan_store(p.cand_plan + cidx, plan);
    }
}

// ---------------------------------------------------------------------------------
// K3: window + autocorrelation in the reference's exact summation order.
// ac[lag] = sum_{i=0}^{n-1-lag} w[i]*w[i+lag] is a LEFT FOLD with every product rounded
// first (encode.rs:3495).  Equivalently, for j = lag..n-1: acc[lag] += w[j]*w[j-lag] in
// increasing j -- the same products in the same order.  One lane owns one candidate and a
// group of LG lags; it walks its candidate's samples sequentially, keeps the last H
// windowed samples in a statically indexed register ring and updates its LG chains
// (independent => full f64 pipeline with one wave per SIMD).  blockIdx.y selects the lag
// group so the ring indexing stays static.  The window value is wave-uniform (scalar load).
// H lags are computed (H = max order + 1 rounded up to a multiple of 4); extra lags are
// simply not read by k_lpc.
// ---------------------------------------------------------------------------------
constexpr int AC_LD = 36;  // row stride of the ac buffer (max H)

// ---------------------------------------------------------------------------------
// K3 for stereo frames (L, R, mid, side candidates) whose length is a multiple of 32, lags <= 16,
// samples <= 24 bits: every wave is self-contained -- no workgroup barrier at all.
//   lane = candidate (16 frames x 4 candidates per wave), wave pairs/quads split the lags.
//   The wave stages the raw L/R rows of its 16 frames through a private, double-buffered LDS
//   tile (32 rows x 32 samples), each lane reads its two source rows 16 samples at a time
//   (ds_read_b128), forms (a + cb * b) >> sh (one v_mad_i32_i24 + one shift covers L, R,
//   mid and side), converts to f64, multiplies by the window (wave-uniform, scalar loads) and
//   accumulates its lags strictly in sample order (the reference's left fold per lag,
//   encode.rs:3403-3413).  History is the previous 16-sample block, kept in registers and
//   statically indexed (the loop is unrolled over two blocks).
// ---------------------------------------------------------------------------------
constexpr int AC3_TS = 32;            // samples per tile
constexpr int AC3_LD = AC3_TS + 4;    // int row stride: 16-byte aligned, rows spread over banks

template <int A, int LG, bool FIRST>
__device__ __forceinline__ void ac3_block(const double (&w)[16], const double (&prev)[16],
                                          double (&acc)[LG]) {
#pragma unroll
    for (int s = 0; s < 16; s++) {
        // all products of a sample first, then the adds: a dependent f64 pair issued back to
        // back stalls the wave (7.3 instead of 5 cycles per instruction with one wave per SIMD)
        double prod[LG];
#pragma unroll
        for (int k = 0; k < LG; k++) {
            const int lag = A + k;
            const double o = (s - lag >= 0) ? w[(s - lag >= 0) ? s - lag : 0]
                                            : prev[(s - lag >= 0) ? 0 : 16 + s - lag];
            prod[k] = w[s] * o;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < LG; k++) {
            const int lag = A + k;
            if (FIRST && s < lag) continue;  // i >= lag only (first block of the frame)
            acc[k] = acc[k] + prod[k];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// raw operands of one 16-sample block: the lane's two source rows + the window slice
struct Ac3Raw {
    int4 a[4], b[4];
    double2 w[8];
};
__device__ __forceinline__ void ac3_load(const int32_t *ra, const int32_t *rb, const double *wt,
                                         uint32_t col, Ac3Raw &r) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        r.a[q] = *reinterpret_cast<const int4 *>(ra + col + 4 * q);
        r.b[q] = *reinterpret_cast<const int4 *>(rb + col + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) r.w[q] = *reinterpret_cast<const double2 *>(wt + col + 2 * q);  // broadcast
}
__device__ __forceinline__ void ac3_convert(const Ac3Raw &r, int32_t cb, uint32_t sh, double (&w)[16]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int32_t a[4] = {r.a[q].x, r.a[q].y, r.a[q].z, r.a[q].w};
        const int32_t b[4] = {r.b[q].x, r.b[q].y, r.b[q].z, r.b[q].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int s = 4 * q + e;
            const int32_t v = (__mul24(b[e], cb) + a[e]) >> sh;
            const double win = (s & 1) ? r.w[s >> 1].y : r.w[s >> 1].x;
            w[s] = (double)v * win;
        }
    }
}


// per-lane constants + staging registers of one wave; ROWS = staged rows per tile (32: the L/R rows
// of 16 stereo frames, 64: one row per candidate for independent channels)
template <int ROWS>
struct Ac3Lane {
    static constexpr int NIT = ROWS / 8;
    static constexpr int RAW = ROWS * AC3_LD;       // ints of raw rows per buffer
    static constexpr int BUF = RAW + 2 * AC3_TS;    // + the window slice (f64)
    const int32_t *gsrc[NIT];  // global source of this lane's staged int4s (tile 0)
    const double *wsrc;        // window slice source (2 f64 per lane of each 16-lane group)
    uint32_t sdst[NIT];        // LDS destinations (ints, within a buffer)
    uint32_t wdst;
    uint32_t off_a, off_b;    // LDS offsets of the candidate's two source rows
    int32_t cb;
    uint32_t sh;
    uint32_t ntiles;
    int4 stage[NIT];
    double2 wstage;
};
template <int ROWS>
__device__ __forceinline__ void ac3_fetch(Ac3Lane<ROWS> &L, uint32_t t) {
    t = t < L.ntiles ? t : L.ntiles - 1;
#pragma unroll
    for (int it = 0; it < ROWS / 8; it++) L.stage[it] = *reinterpret_cast<const int4 *>(L.gsrc[it] + t * AC3_TS);
    L.wstage = *reinterpret_cast<const double2 *>(L.wsrc + t * AC3_TS);
}
template <int ROWS>
__device__ __forceinline__ void ac3_commit(const Ac3Lane<ROWS> &L, int32_t *dst) {
#pragma unroll
    for (int it = 0; it < ROWS / 8; it++) *reinterpret_cast<int4 *>(dst + L.sdst[it]) = L.stage[it];
    *reinterpret_cast<double2 *>(dst + L.wdst) = L.wstage;
}
__device__ __forceinline__ void ac3_sync() {  // LDS operations of one wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Software pipeline per tile of 32 samples (two blocks of 16):
//   block 0: prefetch block 1's operands from LDS, convert + accumulate block 0
//   block 1: commit tile t+1 (fetched one whole tile ago) into the other buffer, start the
//            global fetch of tile t+2, prefetch the next tile's block 0, accumulate block 1
template <int A, int LG, bool FIRST, int ROWS>
__device__ __forceinline__ void ac3_tile(Ac3Lane<ROWS> &L, int32_t *tile, uint32_t t, Ac3Raw &r0, Ac3Raw &r1,
                                         double (&w0)[16], double (&w1)[16], double (&acc)[LG]) {
    const uint32_t buf = t & 1;
    constexpr int AC3_RAW = Ac3Lane<ROWS>::RAW, AC3_BUF = Ac3Lane<ROWS>::BUF;
    int32_t *cur = tile + buf * AC3_BUF, *nxt = tile + (buf ^ 1) * AC3_BUF;
    ac3_load(cur + L.off_a, cur + L.off_b, reinterpret_cast<const double *>(cur + AC3_RAW), 16, r1);
    __builtin_amdgcn_sched_barrier(0);
    ac3_convert(r0, L.cb, L.sh, w0);
    ac3_block<A, LG, FIRST>(w0, w1, acc);
    __builtin_amdgcn_sched_barrier(0);
    ac3_commit(L, nxt);
    ac3_sync();
    ac3_fetch(L, t + 2);
    ac3_load(nxt + L.off_a, nxt + L.off_b, reinterpret_cast<const double *>(nxt + AC3_RAW), 0, r0);
    __builtin_amdgcn_sched_barrier(0);
    ac3_convert(r1, L.cb, L.sh, w1);
    ac3_block<A, LG, false>(w1, w0, acc);
    __builtin_amdgcn_sched_barrier(0);
}

// per-lane constants of a wave: which candidate, which staged rows, where to stage from
template <bool STEREO>
__device__ __forceinline__ bool ac3_setup(const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n,
                                          const double *__restrict__ win, uint32_t group,
                                          Ac3Lane<STEREO ? 32 : 64> &L, uint32_t &frame_out,
                                          uint32_t &cand_out) {
    constexpr int ROWS = STEREO ? 32 : 64;
    constexpr int AC3_RAW = Ac3Lane<ROWS>::RAW;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t total = nframes * p.ncand;            // candidates of this launch
    const uint32_t cand0 = group * 64;
    const bool live = cand0 + lane < total;
    const uint32_t cc = live ? cand0 + lane : total - 1;  // clamped: results of dead lanes are dropped
    const uint32_t frame = frame0 + cc / p.ncand, cand = cc % p.ncand;
    frame_out = frame;
    cand_out = cand;
    const CandInfo ci = p.cinfo[(size_t)frame * p.ncand + cand];
    const uint32_t wasted = (ci.active && !ci.is_const) ? ci.wasted : 0;
    const uint32_t srow = lane >> 3, scol = (lane & 7) * 4;
    if constexpr (STEREO) {
        // candidate = (a + cb * b) >> sh over the frame's rows L (2 fl) and R (2 fl + 1)
        const uint32_t f_lo = cand0 / 4, fl = lane >> 2, f_last = nframes - 1;
        L.off_a = (2 * fl + (cand == 1 ? 1u : 0u)) * AC3_LD;
        L.off_b = (2 * fl + 1) * AC3_LD;
        L.cb = cand == 2 ? 1 : cand == 3 ? -1 : 0;
        L.sh = wasted + (cand == 2 ? 1u : 0u);
        // staging: lane -> (row = it * 8 + lane / 8, 4 samples at column 4 * (lane % 8))
#pragma unroll
        for (int it = 0; it < ROWS / 8; it++) {
            const uint32_t r = it * 8 + srow;              // 0..31: frame r / 2, channel r & 1
            const uint32_t fr = f_lo + r / 2 < nframes ? f_lo + r / 2 : f_last;
            L.gsrc[it] = p.planar + ((size_t)(frame0 + fr) * 2 + (r & 1)) * p.ldb + scol;
            L.sdst[it] = r * AC3_LD + scol;
        }
    } else {
        // independent channels: candidate g of the launch is planar row frame0 * C + g
        L.off_a = lane * AC3_LD;
        L.off_b = L.off_a;
        L.cb = 0;
        L.sh = wasted;
#pragma unroll
        for (int it = 0; it < ROWS / 8; it++) {
            const uint32_t r = it * 8 + srow;              // 0..63: candidate cand0 + r
            const uint32_t g = cand0 + r < total ? cand0 + r : total - 1;
            L.gsrc[it] = p.planar + ((size_t)frame0 * p.channels + g) * p.ldb + scol;
            L.sdst[it] = r * AC3_LD + scol;
        }
    }
    L.ntiles = n / AC3_TS;
    // the window slice (32 f64) is staged by every group of 16 lanes (identical data, same addresses)
    L.wsrc = win + 2 * (lane & 15);
    L.wdst = AC3_RAW + 4 * (lane & 15);
    return live;
}

template <int A, int LG, bool STEREO>
__device__ __forceinline__ void ac3_wave(const Params &p, int32_t *tile /* [2][BUF] */,
                                         uint32_t frame0, uint32_t nframes, uint32_t n,
                                         const double *__restrict__ win, uint32_t group) {
    constexpr int ROWS = STEREO ? 32 : 64;
    constexpr int AC3_RAW = Ac3Lane<ROWS>::RAW;
    Ac3Lane<ROWS> L;
    uint32_t frame, cand;
    const bool live = ac3_setup<STEREO>(p, frame0, nframes, n, win, group, L, frame, cand);
    double acc[LG];
#pragma unroll
    for (int k = 0; k < LG; k++) acc[k] = -0.0;  // f64 `sum()` identity
    double w0[16], w1[16];
    Ac3Raw r0, r1;
    ac3_fetch(L, 0);
    ac3_commit(L, tile);
    ac3_sync();
    ac3_fetch(L, 1);
    ac3_load(tile + L.off_a, tile + L.off_b, reinterpret_cast<const double *>(tile + AC3_RAW), 0, r0);
    ac3_tile<A, LG, true>(L, tile, 0, r0, r1, w0, w1, acc);
#pragma unroll 1
    for (uint32_t t = 1; t < L.ntiles; t++) ac3_tile<A, LG, false>(L, tile, t, r0, r1, w0, w1, acc);
    if (live) {
        double *out = p.ac + ((size_t)frame * p.ncand + cand) * AC_LD + A;
#pragma unroll
        for (int k = 0; k < LG; k++) out[k] = acc[k];
    }
}

// ---- lags up to 32 (LPC orders 17..32): the history of a 16-sample block is the TWO blocks before
// it, so four f64 block buffers rotate (period: two tiles).  FIRST: 0 steady state, 1 / 2 the first /
// second block of the frame (terms whose partner lies before the frame start are not formed).
template <int A, int LG, int FIRST>
__device__ __forceinline__ void ac3_block_deep(const double (&w)[16], const double (&p1)[16],
                                               const double (&p2)[16], double (&acc)[LG]) {
#pragma unroll
    for (int s = 0; s < 16; s++) {
        double prod[LG];
#pragma unroll
        for (int k = 0; k < LG; k++) {
            const int idx = s - (A + k);
            const double o = idx >= 0 ? w[idx >= 0 ? idx : 0]
                           : idx >= -16 ? p1[(idx < 0 && idx >= -16) ? 16 + idx : 0]
                                        : p2[(idx < -16) ? 32 + idx : 0];
            prod[k] = w[s] * o;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < LG; k++) {
            const int lag = A + k;
            if (FIRST == 1 && s < lag) continue;
            if (FIRST == 2 && 16 + s < lag) continue;
            acc[k] = acc[k] + prod[k];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one tile (two blocks): X0, X1 receive the tile's blocks; Y0, Y1 hold the previous tile's
template <int A, int LG, bool FIRSTT, int ROWS>
__device__ __forceinline__ void ac3_tile_deep(Ac3Lane<ROWS> &L, int32_t *tile, uint32_t t, Ac3Raw &raw,
                                              double (&X0)[16], double (&X1)[16], double (&Y0)[16],
                                              double (&Y1)[16], double (&acc)[LG]) {
    const uint32_t buf = t & 1;
    constexpr int AC3_RAW = Ac3Lane<ROWS>::RAW, AC3_BUF = Ac3Lane<ROWS>::BUF;
    int32_t *cur = tile + buf * AC3_BUF, *nxt = tile + (buf ^ 1) * AC3_BUF;
    ac3_load(cur + L.off_a, cur + L.off_b, reinterpret_cast<const double *>(cur + AC3_RAW), 0, raw);
    ac3_convert(raw, L.cb, L.sh, X0);
    ac3_block_deep<A, LG, FIRSTT ? 1 : 0>(X0, Y1, Y0, acc);
    __builtin_amdgcn_sched_barrier(0);
    ac3_load(cur + L.off_a, cur + L.off_b, reinterpret_cast<const double *>(cur + AC3_RAW), 16, raw);
    ac3_commit(L, nxt);   // tile t+1, fetched one whole tile ago
    ac3_sync();
    ac3_fetch(L, t + 2);
    __builtin_amdgcn_sched_barrier(0);
    ac3_convert(raw, L.cb, L.sh, X1);
    ac3_block_deep<A, LG, FIRSTT ? 2 : 0>(X1, X0, Y1, acc);
    __builtin_amdgcn_sched_barrier(0);
}

template <int A, int LG, bool STEREO>
__device__ __forceinline__ void ac3_wave_deep(const Params &p, int32_t *tile, uint32_t frame0,
                                              uint32_t nframes, uint32_t n,
                                              const double *__restrict__ win, uint32_t group) {
    constexpr int ROWS = STEREO ? 32 : 64;
    Ac3Lane<ROWS> L;
    uint32_t frame, cand;
    const bool live = ac3_setup<STEREO>(p, frame0, nframes, n, win, group, L, frame, cand);
    double acc[LG];
#pragma unroll
    for (int k = 0; k < LG; k++) acc[k] = -0.0;  // f64 `sum()` identity
    double wa[16], wb[16], wc[16], wd[16];
    Ac3Raw raw;
    ac3_fetch(L, 0);
    ac3_commit(L, tile);
    ac3_sync();
    ac3_fetch(L, 1);
    ac3_tile_deep<A, LG, true>(L, tile, 0, raw, wa, wb, wc, wd, acc);
    uint32_t t = 1;
#pragma unroll 1
    for (; t + 1 < L.ntiles; t += 2) {  // n is a multiple of 64: an even number of tiles
        ac3_tile_deep<A, LG, false>(L, tile, t, raw, wc, wd, wa, wb, acc);
        ac3_tile_deep<A, LG, false>(L, tile, t + 1, raw, wa, wb, wc, wd, acc);
    }
    if (t < L.ntiles) ac3_tile_deep<A, LG, false>(L, tile, t, raw, wc, wd, wa, wb, acc);
    if (live) {
        double *out = p.ac + ((size_t)frame * p.ncand + cand) * AC_LD + A;
#pragma unroll
        for (int k = 0; k < LG; k++) out[k] = acc[k];
    }
}

template <bool STEREO>
__global__ void __launch_bounds__(256)
k_autocorr3_deep(Params p, uint32_t frame0, uint32_t nframes, uint32_t n, const double *__restrict__ win) {
    __shared__ __attribute__((aligned(16))) int32_t tiles[4][2 * Ac3Lane<STEREO ? 32 : 64>::BUF];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int32_t *tile = tiles[wave];
    switch (wave) {  // 33 lags: 8 + 8 + 8 + 9
    case 0: ac3_wave_deep<0, 8, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
    case 1: ac3_wave_deep<8, 8, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
    case 2: ac3_wave_deep<16, 8, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
    default: ac3_wave_deep<24, 9, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
    }
}

// NL lags split over NS waves; lag ranges are [NL * w / NS, NL * (w + 1) / NS)
template <int NL, int NS, bool STEREO>
__global__ void __launch_bounds__(64 * NS)
k_autocorr3(Params p, uint32_t frame0, uint32_t nframes, uint32_t n, const double *__restrict__ win) {
    __shared__ __attribute__((aligned(16))) int32_t tiles[NS][2 * Ac3Lane<STEREO ? 32 : 64>::BUF];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int32_t *tile = tiles[wave];
    constexpr int B0 = 0, B1 = NL * 1 / NS, B2 = NL * 2 / NS, B3 = NL * 3 / NS, B4 = NL * 4 / NS;
    if constexpr (NS == 2) {
        if (wave == 0) ac3_wave<B0, B1 - B0, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x);
        else ac3_wave<B1, B2 - B1, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x);
    } else {
        switch (wave) {
        case 0: ac3_wave<B0, B1 - B0, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
        case 1: ac3_wave<B1, B2 - B1, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
        case 2: ac3_wave<B2, B3 - B2, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
        default: ac3_wave<B3, B4 - B3, STEREO>(p, tile, frame0, nframes, n, win, blockIdx.x); break;
        }
    }
}

// ---------------------------------------------------------------------------------
// K3 (tiled): generic kernel (any candidate set, block length and order); the samples reach the lanes
// through LDS.  A 256-thread workgroup serves 64 candidates; its 4 waves are the 4 lag groups.
// Per tile of TS samples:
//   1. the planar rows those candidates need (<= 72) are staged with coalesced 16-byte loads
//      (global -> registers for tile i+1 is issued while tile i is consumed);
//   2. all 256 lanes together turn them into the candidates' WINDOWED f64 samples
//      (mid/side derivation, wasted-bit shift, (f64)x * w[i], encode.rs:1799) exactly once;
//   3. every wave walks the 64 f64 columns for its lag group: one LDS read + LG mul/add pairs
//      per sample, ring of the last H values in registers, statically indexed.
// ---------------------------------------------------------------------------------
constexpr int AC_MAXROWS = 72;

template <int H, int A, int LG, bool FIRST, bool GUARDED>
__device__ __forceinline__ void ac_block_w(double (&hist)[H], double (&acc)[LG], const double *tw,
                                           uint32_t col0, uint32_t base, uint32_t n) {
    double w[H];
#pragma unroll
    for (int s = 0; s < H; s++) w[s] = tw[col0 + s];
#pragma unroll
    for (int s = 0; s < H; s++) {
        hist[s] = w[s];
        if (!GUARDED || base + s < n) {
#pragma unroll
            for (int k = 0; k < LG; k++) {
                const int lag = A + k;
                if (!FIRST || s >= lag) {
                    const double prod = w[s] * hist[(s - lag + 2 * H) % H];
                    acc[k] = acc[k] + prod;
                }
            }
        }
    }
}

template <int H, int A, int LG, int KB, int LDT, int LDW, int NW>
__device__ __forceinline__ void ac_wave(const Params &p, int32_t (*tile)[AC_MAXROWS * LDT],
                                        double *wt, uint32_t frame0, uint32_t nframes, uint32_t n,
                                        const double *__restrict__ win) {
    constexpr int TS = H * KB;          // samples per tile
    constexpr int Q = TS / 4;           // int4 per row per tile
    constexpr int NT = 64 * NW;         // threads per workgroup
    constexpr int NLOAD = (AC_MAXROWS * Q + NT - 1) / NT;
    constexpr int CW = (TS + NW - 1) / NW;  // columns converted per wave
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t total = nframes * p.ncand;
    const uint32_t cand0 = blockIdx.x * 64;
    const uint32_t last_c = cand0 + 63 < total ? cand0 + 63 : total - 1;
    const bool live = cand0 + lane < total;
    const uint32_t cc = live ? cand0 + lane : last_c;
    const uint32_t f_lo = cand0 / p.ncand, f_hi = last_c / p.ncand;
    const uint32_t frame = frame0 + cc / p.ncand;
    const uint32_t cand = cc % p.ncand;
    const uint32_t nrows = (f_hi - f_lo + 1) * p.channels;
    const int32_t *gbase = p.planar + (size_t)(frame0 + f_lo) * p.channels * p.ldb;
    // this lane's two source rows inside the staged set
    uint32_t ca = cand, cb = cand;
    int mode = 0;
    if (p.stereo4) {
        if (cand >= 2) { ca = 0; cb = 1; mode = cand == 2 ? 1 : 2; }
    }
    const uint32_t ra = (cc / p.ncand - f_lo) * p.channels + ca;
    const uint32_t rb = (cc / p.ncand - f_lo) * p.channels + cb;
    const CandInfo ci = p.cinfo[(size_t)frame * p.ncand + cand];
    const uint32_t wasted = (ci.active && !ci.is_const) ? ci.wasted : 0;

    double hist[H], acc[LG];
#pragma unroll
    for (int k = 0; k < LG; k++) acc[k] = -0.0;  // f64 `sum()` identity
#pragma unroll
    for (int s = 0; s < H; s++) hist[s] = 0.0;

    const uint32_t ntiles = (n + TS - 1) / TS;
    int4 stage[NLOAD];
    auto fetch = [&](uint32_t t) {
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
            const uint32_t idx = tid + k * NT;
            const uint32_t r = idx / Q, c4 = idx - r * Q;
            const uint32_t col = t * TS + 4 * c4;
            stage[k] = (r < nrows && col < p.ldb)
                           ? *reinterpret_cast<const int4 *>(gbase + (size_t)r * p.ldb + col)
                           : make_int4(0, 0, 0, 0);
        }
    };
    auto commit = [&](uint32_t buf) {
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
            const uint32_t idx = tid + k * NT;
            const uint32_t r = idx / Q, c4 = idx - r * Q;
            if (r < AC_MAXROWS) *reinterpret_cast<int4 *>(&tile[buf][r * LDT + 4 * c4]) = stage[k];
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();
    double *tw = wt + lane * LDW;  // this lane's candidate row of windowed samples
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t buf = t & 1;
        const uint32_t tbase = t * TS;
        if (t + 1 < ntiles) fetch(t + 1);
        {   // step 2: wave `wave` converts columns [wave*CW, wave*CW + CW) of all 64 candidates
            const int32_t *ta = &tile[buf][ra * LDT];
            const int32_t *tb = &tile[buf][rb * LDT];
#pragma unroll
            for (int c = 0; c < CW; c++) {
                const uint32_t col = wave * CW + c;
                if (col < (uint32_t)TS) {
                    const int32_t v = combine(mode, ta[col], tb[col]) >> wasted;
                    tw[col] = (double)v * win[tbase + col];
                }
            }
        }
        __syncthreads();
        if (tbase + TS <= n && t > 0) {
#pragma unroll 1
            for (int b = 0; b < KB; b++)
                ac_block_w<H, A, LG, false, false>(hist, acc, tw, b * H, tbase + b * H, n);
        } else {
#pragma unroll 1
            for (int b = 0; b < KB; b++) {
                const uint32_t base = tbase + b * H;
                if (base >= n) break;
                if (base == 0) ac_block_w<H, A, LG, true, true>(hist, acc, tw, b * H, base, n);
                else ac_block_w<H, A, LG, false, true>(hist, acc, tw, b * H, base, n);
            }
        }
        if (t + 1 < ntiles) commit(buf ^ 1);
        __syncthreads();
    }
    if (live) {
        double *out = p.ac + ((size_t)frame * p.ncand + cand) * AC_LD + A;
#pragma unroll
        for (int k = 0; k < LG; k++) out[k] = acc[k];
    }
}

#ifndef AC_TILE
#define AC_TILE 64   // samples per LDS tile (32 + a 128-VGPR cap overlapped better with other kernels
                     // but ran 0.55 instead of 0.35 ms on its own: net loss)
#endif
template <int H, int NW>
__global__ void __launch_bounds__(64 * NW)
k_autocorr2(Params p, uint32_t frame0, uint32_t nframes, uint32_t n, const double *__restrict__ win) {
    constexpr int LG = H / NW;       // lags per wave (lag group)
    constexpr int KB = (AC_TILE / H) > 0 ? (AC_TILE / H) : 1;
    constexpr int LDT = H * KB + 4;  // int row stride: 16-byte aligned rows
    constexpr int LDW = H * KB + 1;  // f64 row stride: odd => lanes (= rows) hit distinct banks
    __shared__ __attribute__((aligned(16))) int32_t tile[2][AC_MAXROWS * LDT];
    __shared__ double wt[64 * LDW];
    // wave = lag group; each wave runs its own statically indexed code
    switch (threadIdx.x >> 6) {
    case 0: ac_wave<H, 0 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
    case 1: ac_wave<H, 1 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
    case 2: ac_wave<H, 2 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
    case 3: ac_wave<H, 3 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
    default:
        if constexpr (NW == 8) {
            switch (threadIdx.x >> 6) {
            case 4: ac_wave<H, 4 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
            case 5: ac_wave<H, 5 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
            case 6: ac_wave<H, 6 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
            default: ac_wave<H, 7 * LG, LG, KB, LDT, LDW, NW>(p, tile, wt, frame0, nframes, n, win); break;
            }
        }
        break;
    }
}

// ---------------------------------------------------------------------------------
// EXPERIMENT (not on the product path): autocorrelation on the f64 matrix cores.
// Block-Gram form: with X[a][m] = w[16 a + m] (a = 0..n/16-1, m = 0..15),
//   G1 = X^T X            (pairs inside one 16-sample block)
//   G2[m][m'] = sum_a X[a][m] X[a+1][m']   (pairs straddling two consecutive blocks)
//   ac[lag] = sum_{m'-m=lag} G1[m][m'] + sum_{m'+16-m=lag} G2[m][m'],   0 <= lag <= 16.
// One wave per candidate; per 64 samples two v_mfma_f64_16x16x4_f64 (lane l feeds
// A[l&15][l>>4] = B[l>>4][l&15] = w[64 s + l] for G1, B = w[64 s + 16 + l] for G2).
// The summation order differs from the reference's left fold, so the result is NOT
// bit-exact; flacgpu_experiment_mfma_autocorr() measures both its speed and how many
// candidates' quantised LPC parameters change.  On MI355X the f64 MFMA peak equals the f64
// VALU peak, so this buys no time either.
// ---------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(WG) k_autocorr_mfma(Params p, uint32_t n,
                                                      const double *__restrict__ win,
                                                      double *__restrict__ ac_out) {
    __shared__ double acc[4][20];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t total = p.n_frames * p.ncand;
    const uint32_t cidx = blockIdx.x * 4 + wave;
    const bool live = cidx < total;
    const uint32_t cc = live ? cidx : 0;
    const uint32_t frame = cc / p.ncand, cand = cc % p.ncand;
    const CandSrc src = cand_src(p, frame, cand);
    const CandInfo ci = p.cinfo[cc];
    const uint32_t wasted = (ci.active && !ci.is_const) ? ci.wasted : 0;
    if (lane < 20) acc[wave][lane] = 0.0;
    auto load = [&](uint32_t s) -> double {
        const uint32_t i = 64 * s + lane;
        if (i >= n) return 0.0;
        return (double)(combine(src.mode, src.a[i], src.b[i]) >> wasted) * win[i];
    };
    double4_t g1 = {0, 0, 0, 0}, g2 = {0, 0, 0, 0};
    const uint32_t nsteps = (n + 63) / 64;
    double cur = load(0);
    for (uint32_t s = 0; s < nsteps; s++) {
        const double nxt = (s + 1 < nsteps) ? load(s + 1) : 0.0;
        const double from_cur = __shfl(cur, (int)((lane + 16) & 63), 64);
        const double from_nxt = __shfl(nxt, (int)((lane + 16) & 63), 64);
        const double wb = lane < 48 ? from_cur : from_nxt;
        g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur, cur, g1, 0, 0, 0);
        g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur, wb, g2, 0, 0, 0);
        cur = nxt;
    }
    __syncthreads();
    // lane l holds D[row = (l >> 4) + 4 r][col = l & 15] (f64 C/D layout)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = (int)(lane >> 4) + 4 * r, j = (int)(lane & 15);
        if (j - i >= 0) atomicAdd(&acc[wave][j - i], g1[r]);
        if (j + 16 - i <= 16) atomicAdd(&acc[wave][j + 16 - i], g2[r]);
    }
    __syncthreads();
    if (live && lane <= 16) ac_out[(size_t)cc * AC_LD + lane] = acc[wave][lane];
}

// ---------------------------------------------------------------------------------
// K4: Levinson-Durbin + order estimate + quantisation, one lane per candidate
// ---------------------------------------------------------------------------------
__device__ __forceinline__ long long total_key(double x) {  // f64::total_cmp key
    long long b = __double_as_longlong(x);
    b ^= (long long)((unsigned long long)(b >> 63) >> 1);
    return b;
}

__global__ void __launch_bounds__(64) k_lpc(Params p) {
    const uint32_t idx = p.f0 * p.ncand + blockIdx.x * 64 + threadIdx.x;
    if (idx >= (p.f0 + p.fcount) * p.ncand) return;
    const uint32_t frame = idx / p.ncand;
    const uint32_t n = frame_len(p, frame);
    const CandInfo ci = p.cinfo[idx];
    LpcParams *out = p.lpc + idx;
    if (!ci.active || ci.is_const) {
        out->status = 1;
        return;
    }
    const uint32_t L = p.max_lpc_order;
    if (n <= L) {  // InsufficientLpcSamples, encode.rs:3300
        out->status = 1;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    // precision table, encode.rs:3305-3315
    const uint32_t precision = n <= 192 ? 7 : n <= 384 ? 8 : n <= 576 ? 9 : n <= 1152 ? 10
                               : n <= 2304 ? 11 : n <= 4608 ? 12 : 13;
    const double *ac = p.ac + (size_t)idx * AC_LD;
    double c[FLACGPU_MAX_LPC_ORDER], cn[FLACGPU_MAX_LPC_ORDER], errs[FLACGPU_MAX_LPC_ORDER];
    // pass 1: errors of every order (lp_coefficients, encode.rs:3536-3580)
    {
        double k = ac[1] / ac[0];
        c[0] = k;
        double err = ac[0] * (1.0 - k * k);
        errs[0] = err;
        for (uint32_t i = 1; i < L; i++) {
            double s = -0.0;
            for (uint32_t j = 0; j < i; j++) {
                double prod = ac[i - j] * c[j];
                s = s + prod;
            }
            double q = ac[i + 1] - s;
            double kk = q / err;
            for (uint32_t j = 0; j < i; j++) {
                double t = kk * c[i - 1 - j];
                cn[j] = c[j] - t;
            }
            cn[i] = kk;
            for (uint32_t j = 0; j <= i; j++) c[j] = cn[j];
            err = err * (1.0 - kk * kk);
            errs[i] = err;
        }
    }
    // compute_best_order, encode.rs:3656-3702 (bits-per-residual NOT clamped, :3675)
    const double LN_2 = 0.693147180559945309417232121458176568;
    const double error_scale = 0.5 / (double)n;
    const double denom = 2.0 * LN_2;
    int best = -1;
    double best_bits = 0.0, second = 0.0;
    bool have_second = false;
    for (uint32_t i = 0; i < L; i++) {
        if (!(errs[i] > 0.0)) break;  // take_while(error > 0.0)
        uint32_t order = i + 1;
        double header_bits = (double)(order * ((uint32_t)ci.bps + precision));
        double bpr = log(errs[i] * error_scale) / denom;
        double bits = __builtin_fma(bpr, (double)(n - order), header_bits);
        if (best < 0) {
            best = (int)i;
            best_bits = bits;
        } else if (total_key(bits) < total_key(best_bits)) {
            second = best_bits;
            have_second = true;
            best = (int)i;
            best_bits = bits;
        } else if (!have_second || total_key(bits) < total_key(second)) {
            second = bits;
            have_second = true;
        }
    }
    if (best < 0) {  // NoBestLpcOrder
        out->status = 2;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    if (have_second && fabs(second - best_bits) <= 1e-9 * fabs(best_bits)) atomicAdd(&p.stats[1], 1u);
    const uint32_t order = (uint32_t)best + 1;
    // pass 2: coefficients of the chosen order (same recursion, same roundings)
    {
        double k = ac[1] / ac[0];
        c[0] = k;
        double err = ac[0] * (1.0 - k * k);
        for (uint32_t i = 1; i < order; i++) {
            double s = -0.0;
            for (uint32_t j = 0; j < i; j++) {
                double prod = ac[i - j] * c[j];
                s = s + prod;
            }
            double q = ac[i + 1] - s;
            double kk = q / err;
            for (uint32_t j = 0; j < i; j++) {
                double t = kk * c[i - 1 - j];
                cn[j] = c[j] - t;
            }
            cn[i] = kk;
            for (uint32_t j = 0; j <= i; j++) c[j] = cn[j];
            err = err * (1.0 - kk * kk);
        }
    }
    // quantize, encode.rs:3334-3401
    const int32_t max_coeff = (1 << (precision - 1)) - 1, min_coeff = -(1 << (precision - 1));
    double l = fabs(c[0]);
    for (uint32_t i = 1; i < order; i++) {
        double a = fabs(c[i]);
        if (total_key(a) >= total_key(l)) l = a;
    }
    if (!(l > 0.0)) {  // ZeroLpCoefficients (also NaN)
        out->status = 3;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    // floor(log2(l)) as the host libm computes it: exponent, bumped when l sits in the
    // (few-ulp) band below 2^(e+1) where log2() rounds up to e+1 (table built on the host)
    int32_t fl;
    if (isinf(l)) {
        fl = INT32_MAX;
    } else {
        int e = ilogb(l);
        fl = e;
        if (e >= -64 && e < 64 && l >= p.log2_thr[e + 64]) {
            fl = e + 1;
            atomicAdd(&p.stats[2], 1u);
        }
    }
    int32_t sh = (int32_t)((uint32_t)(int32_t)(precision - 1) - (uint32_t)fl - 1u);
    if (sh > 15) sh = 15;
    if (sh < -16) {  // LpNegativeShiftError
        out->status = 4;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    double error = 0.0;
    const double scale = (double)(1 << (sh >= 0 ? sh : -sh));
    for (uint32_t i = 0; i < order; i++) {
        double sum = sh >= 0 ? __builtin_fma(c[i], scale, error) : (c[i] / scale) + error;
        double rr = round(sum);
        int32_t q = (rr != rr) ? 0 : rr >= 2147483647.0 ? INT32_MAX : rr <= -2147483648.0 ? INT32_MIN
                                                                                         : (int32_t)rr;
        q = q < min_coeff ? min_coeff : q > max_coeff ? max_coeff : q;
        error = sum - (double)q;
        out->qlp[i] = q;
    }
    out->status = 0;
    out->order = (uint8_t)order;
    out->precision = (uint8_t)precision;
    out->shift = (uint8_t)(sh >= 0 ? sh : 0);
}

// Levinson-Durbin recursion up to `upto` orders (lp_coefficients, encode.rs:3536-3580), every loop
// unrolled and guarded so that the arrays stay in registers
template <int LMAX>
__device__ __forceinline__ void levinson_u(const double (&acr)[LMAX + 1], uint32_t upto, double (&c)[LMAX],
                                           double (&errs)[LMAX]) {
    double cn[LMAX];
    double k = acr[1] / acr[0];
    c[0] = k;
    double err = acr[0] * (1.0 - k * k);
    errs[0] = err;
#pragma unroll
    for (int i = 1; i < LMAX; i++) {
        if ((uint32_t)i < upto) {
            double s = -0.0;
#pragma unroll
            for (int j = 0; j < i; j++) {
                double prod = acr[i - j] * c[j];
                s = s + prod;
            }
            double q = acr[i + 1] - s;
            double kk = q / err;
#pragma unroll
            for (int j = 0; j < i; j++) {
                double t = kk * c[i - 1 - j];
                cn[j] = c[j] - t;
            }
            cn[i] = kk;
#pragma unroll
            for (int j = 0; j <= i; j++) c[j] = cn[j];
            err = err * (1.0 - kk * kk);
            errs[i] = err;
        }
    }
}

// K4 with max_lpc_order <= LMAX known at compile time: with run-time loop bounds the
// coefficient arrays sit in scratch memory and every access is a memory round trip
template <int LMAX>
__global__ void __launch_bounds__(64) k_lpc_u(Params p) {
    static_assert(LMAX <= FLACGPU_MAX_LPC_ORDER, "order");
    const uint32_t idx = p.f0 * p.ncand + blockIdx.x * 64 + threadIdx.x;
    if (idx >= (p.f0 + p.fcount) * p.ncand) return;
    const uint32_t frame = idx / p.ncand;
    const uint32_t n = frame_len(p, frame);
    const CandInfo ci = p.cinfo[idx];
    LpcParams *out = p.lpc + idx;
    if (!ci.active || ci.is_const) {
        out->status = 1;
        return;
    }
    const uint32_t L = p.max_lpc_order;
    if (n <= L) {  // InsufficientLpcSamples, encode.rs:3300
        out->status = 1;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    // precision table, encode.rs:3305-3315
    const uint32_t precision = n <= 192 ? 7 : n <= 384 ? 8 : n <= 576 ? 9 : n <= 1152 ? 10
                               : n <= 2304 ? 11 : n <= 4608 ? 12 : 13;
    const double *ac = p.ac + (size_t)idx * AC_LD;
    double acr[LMAX + 1], c[LMAX], errs[LMAX];
#pragma unroll
    for (int i = 0; i <= LMAX; i++) acr[i] = (uint32_t)i <= L ? ac[i] : 0.0;
    // pass 1: errors of every order
    levinson_u<LMAX>(acr, L, c, errs);
    // compute_best_order, encode.rs:3656-3702 (bits-per-residual NOT clamped, :3675)
    const double LN_2 = 0.693147180559945309417232121458176568;
    const double error_scale = 0.5 / (double)n;
    const double denom = 2.0 * LN_2;
    int best = -1;
    double best_bits = 0.0, second = 0.0;
    bool have_second = false;
    bool going = true;
#pragma unroll
    for (int i = 0; i < LMAX; i++) {
        going = going && (uint32_t)i < L && errs[i] > 0.0;  // take_while(error > 0.0)
        if (!going) continue;
        uint32_t order = i + 1;
        double header_bits = (double)(order * ((uint32_t)ci.bps + precision));
        double bpr = log(errs[i] * error_scale) / denom;
        double bits = __builtin_fma(bpr, (double)(n - order), header_bits);
        if (best < 0) {
            best = (int)i;
            best_bits = bits;
        } else if (total_key(bits) < total_key(best_bits)) {
            second = best_bits;
            have_second = true;
            best = (int)i;
            best_bits = bits;
        } else if (!have_second || total_key(bits) < total_key(second)) {
            second = bits;
            have_second = true;
        }
    }
    if (best < 0) {  // NoBestLpcOrder
        out->status = 2;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    if (have_second && fabs(second - best_bits) <= 1e-9 * fabs(best_bits)) atomicAdd(&p.stats[1], 1u);
    const uint32_t order = (uint32_t)best + 1;
    // pass 2: coefficients of the chosen order (same recursion, same roundings)
    levinson_u<LMAX>(acr, order, c, errs);
    // quantize, encode.rs:3334-3401
    const int32_t max_coeff = (1 << (precision - 1)) - 1, min_coeff = -(1 << (precision - 1));
    double l = fabs(c[0]);
#pragma unroll
    for (int i = 1; i < LMAX; i++) {
        if ((uint32_t)i < order) {
            double a = fabs(c[i]);
            if (total_key(a) >= total_key(l)) l = a;
        }
    }
    if (!(l > 0.0)) {  // ZeroLpCoefficients (also NaN)
        out->status = 3;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    // floor(log2(l)) as the host libm computes it: exponent, bumped when l sits in the
    // (few-ulp) band below 2^(e+1) where log2() rounds up to e+1 (table built on the host)
    int32_t fl;
    if (isinf(l)) {
        fl = INT32_MAX;
    } else {
        int e = ilogb(l);
        fl = e;
        if (e >= -64 && e < 64 && l >= p.log2_thr[e + 64]) {
            fl = e + 1;
            atomicAdd(&p.stats[2], 1u);
        }
    }
    int32_t sh = (int32_t)((uint32_t)(int32_t)(precision - 1) - (uint32_t)fl - 1u);
    if (sh > 15) sh = 15;
    if (sh < -16) {  // LpNegativeShiftError
        out->status = 4;
        atomicAdd(&p.stats[0], 1u);
        return;
    }
    double error = 0.0;
    const double scale = (double)(1 << (sh >= 0 ? sh : -sh));
#pragma unroll
    for (int i = 0; i < LMAX; i++) {
        if ((uint32_t)i >= order) continue;
        double sum = sh >= 0 ? __builtin_fma(c[i], scale, error) : (c[i] / scale) + error;
        double rr = round(sum);
        int32_t q = (rr != rr) ? 0 : rr >= 2147483647.0 ? INT32_MAX : rr <= -2147483648.0 ? INT32_MIN
                                                                                         : (int32_t)rr;
        q = q < min_coeff ? min_coeff : q > max_coeff ? max_coeff : q;
        error = sum - (double)q;
        out->qlp[i] = q;
    }
    out->status = 0;
    out->order = (uint8_t)order;
    out->precision = (uint8_t)precision;
    out->shift = (uint8_t)(sh >= 0 ? sh : 0);
}

// ---------------------------------------------------------------------------------
// K5: LPC FIR residual + Rice search + fixed/LPC/verbatim choice, one workgroup per
// (frame, candidate).  dynamic LDS: x[n] | r[n]
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_fir(Params p) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ RiceShared RS;
    __shared__ SubPlan plan;
    __shared__ uint64_t red[4];
    __shared__ int32_t qlp[FLACGPU_MAX_LPC_ORDER];

    uint32_t frame, cand;
    map_block(blockIdx.x, p.ncand, p.fcount, frame, cand);
    frame += p.f0;
    const uint32_t n = frame_len(p, frame);
    const size_t cidx = (size_t)frame * p.ncand + cand;
    const uint32_t tid = threadIdx.x;
    const CandInfo ci = p.cinfo[cidx];
    if (!ci.active || ci.is_const) return;  // CONSTANT already final (k_fixed)
    const LpcParams *lp = p.lpc + cidx;
    const CandSrc src = cand_src(p, frame, cand);
    const uint32_t wasted = ci.wasted, bps_eff = ci.bps;
    const int32_t status = lp->status;
    bool lpc_ok = status == 0;
    uint32_t lpc_bits = 0;
    plan_clear(plan);
    if (lpc_ok) {
        int32_t *x = lds;
        int32_t *r = lds + p.block_size;
        const uint32_t order = lp->order, shift = lp->shift;
        if (tid < FLACGPU_MAX_LPC_ORDER) qlp[tid] = tid < order ? lp->qlp[tid] : 0;
        for (uint32_t i = tid; i < n; i += WG) x[i] = combine(src.mode, src.a[i], src.b[i]) >> wasted;
        __syncthreads();
        // encode_residuals, encode.rs:3181-3197
        uint32_t ovf = 0;
        for (uint32_t i = order + tid; i < n; i += WG) {
            long long sum = 0;
            for (uint32_t j = 0; j < order; j++) sum += (long long)x[i - 1 - j] * (long long)qlp[j];
            int32_t pred = (int32_t)(sum >> shift);
            long long d = (long long)x[i] - (long long)pred;
            if (d < INT32_MIN || d > INT32_MAX) ovf = 1;  // checked_sub -> ResidualOverflow
            r[RIDX(i)] = (int32_t)d;
        }
        ovf = block_or_u32(ovf, red);
        if (ovf) {
            lpc_ok = false;
            if (tid == 0) atomicAdd(&p.stats[0], 1u);
        } else {
            uint32_t rbits = 0;
            if (!(p.dbg & 1) && !rice_search(r, n, order, p, RS, plan, rbits)) {
                lpc_ok = false;
                if (tid == 0) atomicAdd(&p.stats[0], 1u);
            }
            lpc_bits = 8u + wasted + order * bps_eff + 4u + 5u + order * lp->precision + rbits;
        }
    }
    __syncthreads();
    const SubPlan *fx = p.fixed_plan + cidx;
    const uint32_t fixed_bits = fx->bits;
    const bool fixed_ok = fx->reserved[0] == 0;
    // (Ok,Ok) -> min_by_key(written) with FIXED first: tie keeps FIXED; (Err,Ok) -> LPC;
    // (Ok,Err) -> FIXED; (Err,Err) -> VERBATIM (encode.rs:2929-2945)
    const bool use_lpc = lpc_ok && (!fixed_ok || lpc_bits < fixed_bits);
    const uint32_t best_bits = use_lpc ? lpc_bits : fixed_bits;
    const bool verbatim = (!fixed_ok && !lpc_ok) || !(best_bits < n * bps_eff);  // :2971-2979
    if (verbatim) {
        __syncthreads();
        plan_clear(plan);
        __syncthreads();
        if (tid == 0) make_verbatim(plan, n, bps_eff, wasted, src.source);
        __syncthreads();
        plan_store(p.cand_plan + cidx, plan);
    } else if (use_lpc) {
        if (tid == 0) {
            plan.type = FLACGPU_SUB_LPC;
            plan.wasted = (uint8_t)wasted;
            plan.bps = (uint8_t)bps_eff;
            plan.order = lp->order;
            plan.precision = lp->precision;
            plan.shift = lp->shift;
            plan.source = src.source;
            plan.bits = lpc_bits;
        }
        if (tid < FLACGPU_MAX_LPC_ORDER) plan.coeffs[tid] = qlp[tid];
        __syncthreads();
        plan_store(p.cand_plan + cidx, plan);
    } else {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(fx);
        uint32_t *d = reinterpret_cast<uint32_t *>(p.cand_plan + cidx);
        for (uint32_t i = tid; i < sizeof(SubPlan) / 4; i += WG) d[i] = s[i];
    }
}

// =================================================================================
// Wave-per-candidate kernel for blocks of exactly 4096 samples (candidates <= 25 bits, LPC
// order <= 16): ONE wave64 does the whole FIXED + LPC analysis of one candidate.  Lane l owns
// samples [64 l, 64 l + 64) in registers, which is exactly one finest Rice partition, so the
// partition tree needs no atomics; neighbours come through wave shuffles; there is NO workgroup
// barrier in the kernel (the 4 waves of a workgroup are the 4 candidates of a frame and share
// L1/L2).  Same decisions as the generic k_fixed + k_fir (same helpers).
// =================================================================================
struct WaveRice {
    uint32_t bits;      // residual block bits (method + order + partitions)
    bool ok;            // false: the 31-bit fallback partition cannot hold a residual
    int bp;             // chosen partition level, -1 = 31-bit escaped fallback
    uint32_t count, method;
    uint8_t price, pesc;  // parameters of partition `lane` (valid for lane < count)
};

// v + (v moved across lanes by one DPP control); lanes without a source lane add 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
// inclusive prefix sum over the 64 lanes (wrapping u32), six DPP adds and no LDS traffic:
// row_shr 1/2/4/8 scan each row of 16, row_bcast:15 / :31 carry the row totals forward
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v = dpp_add<0x111>(v);
    v = dpp_add<0x112>(v);
    v = dpp_add<0x114>(v);
    v = dpp_add<0x118>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_u32(v), 63);
}
// total of per-lane values < 2^48 (two u32 limbs)
__device__ __forceinline__ uint64_t wave_total_u48(uint64_t v) {
    const uint32_t lo = wave_total_u32((uint32_t)v & 0xFFFFFFu);
    const uint32_t hi = wave_total_u32((uint32_t)(v >> 24));
    return ((uint64_t)hi << 24) + lo;
}
// the previous lane's value, 0 on lane 0 (wave_shr:1)
__device__ __forceinline__ int32_t lane_prev(int32_t v) {
    int32_t r = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
    // keep it a v_mov_b32_dpp: folded into a consumer (v_sub_u32_dpp ... wave_shr:1) the halo of
    // the FIXED differences came out wrong on gfx950 (tests/test_gpu_pack.py, FIXED-only cases)
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ uint32_t sread(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// Residual sources of wave_rice.  resid(e) is called with e = 0..63 ascending after reset().
// StoredSrc: residuals held in a register array (the LPC residual, computed in place over the
// samples); they are folded in place by the first pass.
template <int SPL>
struct StoredSrc {
    static constexpr bool STORED = true;
    int32_t (&a)[SPL];
    __device__ __forceinline__ void reset() {}
    __device__ __forceinline__ int32_t resid(int e) { return a[e]; }
};
// FixedSrc<K>: residual of fixed order K generated on the fly from the samples (encode.rs:
// 3039-3060); nothing is stored, the second pass of the search simply regenerates it (K
// subtractions per sample) -- this keeps the kernel at 64 instead of 128 array registers.
template <int K, int SPL>
struct FixedSrc {
    static constexpr bool STORED = false;
    int32_t (&x)[SPL];
    int32_t (&h)[4];  // the 4 samples before x[0]
    int32_t q0, q1, q2, q3;
    __device__ __forceinline__ void reset() {
        // make the samples opaque: otherwise the compiler shares this pass's differences with
        // the previous pass (and with the order statistics), i.e. keeps 64..256 of them live
#pragma unroll
        for (int e = 0; e < SPL; e++) asm volatile("" : "+v"(x[e]));
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(h[k]));
        q0 = h[3];
        q1 = h[3] - h[2];
        q2 = q1 - (h[2] - h[1]);
        q3 = q2 - ((h[2] - h[1]) - (h[1] - h[0]));
    }
    __device__ __forceinline__ int32_t resid(int e) {
        const int32_t d1 = x[e] - q0, d2 = d1 - q1, d3 = d2 - q2, d4 = d3 - q3;
        q0 = x[e]; q1 = d1; q2 = d2; q3 = d3;
        return K == 0 ? x[e] : K == 1 ? d1 : K == 2 ? d2 : K == 3 ? d3 : d4;
    }
};

// Rice search of one wave's residual (encode.rs:3862-3947), lane l = samples [SPL l, SPL l + SPL)
// of a block of 64 SPL samples (SPL >= 16: one finest partition of a partition-order-6 block).
// Warm-up entries (lane 0, e < order <= MAXORD) count as zero and every value is folded to
// t = r ^ (r >> 31) = zigzag(r) >> 1, so that
//   sum |r| = sum t + #negative,  zigzag(r) >> k = t >> (k - 1) for k >= 1,
//   sum zigzag(r) = 2 sum t + #negative.
// Level totals come from one DPP scan of the node estimates plus ballots; nothing touches LDS.
// folded value of sample e in the second and later passes
template <int SPL, int MAXORD, class Src>
__device__ __forceinline__ uint32_t rice_folded(Src &src, int e, uint32_t first) {
    if constexpr (Src::STORED) {
        return (uint32_t)src.a[e];
    } else {
        int32_t r = src.resid(e);
        if (e < MAXORD) r = (uint32_t)e >= first ? r : 0;
        return (uint32_t)(r ^ (r >> 31));
    }
}
// leaf_sum: for a generated source, this lane's sum |r| (the order statistics already have it:
// the first pass is skipped); ignored for a stored source.
template <int SPL, int MAXORD, class Src>
__device__ __forceinline__ WaveRice wave_rice(Src src, uint32_t order, const Params &p,
                                              uint64_t leaf_sum = 0) {
    const uint32_t lane = threadIdx.x & 63;
    constexpr uint32_t N = 64u * SPL;  // block length
    const uint32_t P = rice_levels(N, p);
    const uint32_t rice_max = p.use_rice2 ? 31u : 15u;
    const uint32_t first = lane == 0 ? order : 0u;
    uint64_t sum_t = leaf_sum;
    uint32_t neg = 0;
    if constexpr (Src::STORED) {
        sum_t = 0;
#pragma unroll
        for (int e = 0; e < SPL; e += 2) {
            int32_t r0 = src.resid(e), r1 = src.resid(e + 1);
            if (e < MAXORD) r0 = (uint32_t)e >= first ? r0 : 0;
            if (e + 1 < MAXORD) r1 = (uint32_t)(e + 1) >= first ? r1 : 0;
            const int32_t s0 = r0 >> 31, s1 = r1 >> 31;
            const uint32_t t0 = (uint32_t)(r0 ^ s0), t1 = (uint32_t)(r1 ^ s1);
            src.a[e] = (int32_t)t0;
            src.a[e + 1] = (int32_t)t1;
            neg -= (uint32_t)s0;
            neg -= (uint32_t)s1;
            sum_t += t0 + t1;  // each < 2^31
            if ((e & 7) == 6) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t mysum = sum_t + neg;  // sum |r| of this lane's 64 samples, < 2^37
    // inclusive prefix of the leaf sums in two limbs
    const uint32_t sc_lo = wave_scan_u32((uint32_t)mysum & 0xFFFFFu);
    const uint32_t sc_hi = wave_scan_u32((uint32_t)(mysum >> 20));
    const uint64_t incl = ((uint64_t)sc_hi << 20) + sc_lo;
    // leaf node 64 + lane (level 6) and internal node `lane` (levels 0..5), heap numbering
    const uint32_t leaf_cnt = (uint32_t)SPL - first;
    const PartEval le = partition_eval(leaf_cnt, mysum, rice_max);
    const uint32_t node = lane;
    const uint32_t lvl = node ? 31u - (uint32_t)__builtin_clz(node) : 0u;
    const uint32_t j = node - (1u << lvl);
    const uint32_t span = 64u >> lvl;
    const uint32_t hi = (j + 1) * span - 1, lo = j * span;
    const uint64_t top = __shfl(incl, (int)(hi & 63), 64);
    const uint64_t bot = __shfl(incl, (int)((lo ? lo - 1 : 0) & 63), 64);
    const uint64_t isum = top - (lo ? bot : 0ull);
    const uint32_t plen = N >> lvl;
    const uint32_t istart = j * plen, iend = istart + plen;
    const uint32_t icnt = (node && iend > order) ? iend - (istart > order ? istart : order) : 0u;
    const PartEval ie = partition_eval(icnt, isum, rice_max);
    // per-level aggregates: estimate sums from a scan over the node lanes, the rest from ballots
    const uint32_t esc = wave_scan_u32(icnt ? ie.est : 0u);
    const uint64_t m_cnt = __ballot(icnt > 0);
    const uint64_t m_bad = __ballot(icnt > 0 && ie.bad);
    const uint64_t m_hi = __ballot(icnt > 0 && ie.kind == PK_STANDARD && ie.rice >= 15);
    WaveRice w;
    w.bp = -1;
    uint32_t best_est = 0, best_cnt = 1, best_hi = 0;
#pragma unroll
    for (uint32_t l = 0; l <= 6; l++) {
        if (l > P) break;
        uint32_t est, c;
        bool bad, hi15;
        if (l < 6) {
            const uint64_t lm = ((1ull << (1u << l)) - 1ull) << (1u << l);  // lanes 2^l .. 2^(l+1)-1
            est = sread(esc, (2 << l) - 1) - sread(esc, (1 << l) - 1);
            c = (uint32_t)__popcll(m_cnt & lm);
            bad = (m_bad & lm) != 0;
            hi15 = (m_hi & lm) != 0;
        } else {
            est = wave_total_u32(leaf_cnt ? le.est : 0u);
            c = (uint32_t)__popcll(__ballot(leaf_cnt > 0));
            bad = __ballot(leaf_cnt > 0 && le.bad) != 0;
            hi15 = __ballot(leaf_cnt > 0 && le.kind == PK_STANDARD && le.rice >= 15) != 0;
        }
        // `!p.is_empty() && p.len().is_power_of_two()` (encode.rs:3881), first minimum (:3885)
        const bool ok = !bad && c > 0 && (c & (c - 1)) == 0;
        if (ok && (w.bp < 0 || est < best_est)) {
            w.bp = (int)l;
            best_est = est;
            best_cnt = c;
            best_hi = hi15 ? 1u : 0u;
        }
    }
    w.count = best_cnt;
    w.method = (w.bp >= 0 && p.use_rice2 && best_hi) ? 1u : 0u;  // try_reduce_rice, :3929-3942
    const uint32_t hb = w.method ? 5u : 4u;
    // packed evaluation records to pass between lanes: cnt | kind << 16 | rice << 18 | esc << 26
    const uint32_t lrec = leaf_cnt | ((uint32_t)le.kind << 16) | ((uint32_t)le.rice << 18) | ((uint32_t)le.esc << 26);
    const uint32_t irec = icnt | ((uint32_t)ie.kind << 16) | ((uint32_t)ie.rice << 18) | ((uint32_t)ie.esc << 26);
    uint32_t mine = 0;  // bits, wrapping u32 like the reference's counter
    w.price = 0;
    w.pesc = 0;
    bool wide = false;
    if (w.bp >= 0) {
        const uint32_t bp = (uint32_t)w.bp;
        const uint32_t first_j = (1u << bp) - w.count;
        // partition q = lane of the chosen level: its record lives on lane first_j + q (leaf
        // level) or on lane 2^bp + first_j + q (internal node)
        const uint32_t qsrc = bp == 6 ? first_j + lane : (1u << bp) + first_j + lane;
        const uint32_t qrec = __shfl(bp == 6 ? lrec : irec, (int)(qsrc & 63), 64);
        if (lane < w.count) {
            const uint32_t c = qrec & 0xFFFF, kind = (qrec >> 16) & 3, rice = (qrec >> 18) & 0xFF, esc2 = qrec >> 26;
            w.price = (uint8_t)rice;
            w.pesc = (uint8_t)esc2;
            if (kind == PK_STANDARD) mine += hb + (1u + rice) * c;
            else if (kind == PK_ESCAPED) mine += hb + 5u + esc2 * c;
            else mine += hb + 5u;
        }
        // Rice parameter of the partition this lane's 64 samples fall in
        const uint32_t msrc = (1u << bp) + (lane >> (6 - bp));
        const uint32_t mrec = bp == 6 ? lrec : (uint32_t)__shfl(irec, (int)(msrc & 63), 64);
        const uint32_t k = (mrec >> 18) & 0xFF;
        uint32_t q = 0;
        if constexpr (Src::STORED) {
            const uint32_t sh = (k == 0 || k == 0xFF) ? 0u : k - 1u;
#pragma unroll
            for (int e = 0; e < SPL; e++) {
                q += (uint32_t)src.a[e] >> sh;
                if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            mine += k == 0xFF ? 0u : k == 0 ? 2u * (uint32_t)sum_t + neg : q;
        } else {  // regenerate the residual: zigzag(r) >> k directly (k = 0 included)
            const uint32_t ks = k == 0xFF ? 0u : k;
            src.reset();
#pragma unroll
            for (int e = 0; e < SPL; e++) {
                int32_t r = src.resid(e);
                if (e < MAXORD) r = (uint32_t)e >= first ? r : 0;
                q += zigzag(r) >> ks;
                if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            mine += k == 0xFF ? 0u : q;
        }
    } else {
        if (lane == 0) {
            w.price = 0xFF;
            w.pesc = 31;  // one escaped 31-bit partition, encode.rs:3887-3895
            mine += 4u + 5u + 31u * (N - order);
        }
        // t >= 2^30  <=>  r outside [-2^30, 2^30): write_signed_counted(31) fails (:3857)
        src.reset();
#pragma unroll
        for (int e = 0; e < SPL; e++) wide |= rice_folded<SPL, MAXORD>(src, e, first) >= (1u << 30);
    }
    w.bits = 6u + wave_total_u32(mine);  // method (2) + partition order (4), :3949, :3902
    w.ok = !__any(wide);
    return w;
}

// FIR of one lane's 64 samples, IN PLACE and descending (x[e] is dead once its residual exists);
// hp[MAXO] = the MAXO samples before them (zeros for lane 0); coefficients are wave-uniform.
// Returns the sign-bit OR of every i32 subtraction overflow outside the warm-up
// (ResidualOverflow, encode.rs:3190-3197).
struct KeepResidual {  // fir64 consumer: just store the residual
    __device__ __forceinline__ int32_t operator()(int, int32_t d) { return d; }
};
// f(e, residual) -> value stored in x[e]: lets a caller consume each residual where it is
// produced (k_frame64 sums the code lengths there)
template <int T, int SPL, int MAXO = 16, int CBASE = 2, class F = KeepResidual>
__device__ __forceinline__ uint32_t fir64(int32_t (&x)[SPL], const int32_t (&hp)[MAXO], uint32_t lpw,
                                          uint32_t order, uint32_t shift, F &&f = F()) {
    int32_t c[T];  // wave-uniform (SGPRs): coefficient j was loaded by lane CBASE + j
#pragma unroll
    for (int j = 0; j < T; j++) c[j] = (uint32_t)j < order ? (int32_t)sread(lpw, CBASE + j) : 0;
    // bit e set: sample e of this lane is warm-up (lane 0 only)
    const uint32_t warm = (threadIdx.x & 63) == 0 ? (order >= 32 ? 0xFFFFFFFFu : (1u << order) - 1u) : 0u;
    uint32_t ovf = 0;
#pragma unroll
    for (int e = SPL - 1; e >= 0; e--) {
        long long sum = 0;
#pragma unroll
        for (int j = 0; j < T; j++) {
            const int i = e - 1 - j;
            const int32_t v = i >= 0 ? x[i >= 0 ? i : 0] : hp[i >= 0 ? 0 : MAXO + i];
            sum += (long long)v * (long long)c[j];
        }
        const int32_t pred = (int32_t)(sum >> shift);
        const int32_t d = (int32_t)((uint32_t)x[e] - (uint32_t)pred);
        uint32_t o = (uint32_t)(x[e] ^ pred) & (uint32_t)(x[e] ^ d);  // sign bit: x - pred overflowed
        if (e < MAXO) o &= ~(warm << (31 - e));
        ovf |= o;
        x[e] = f(e, d);
        if ((e & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    return ovf >> 31;
}

__device__ __forceinline__ void store_plan_wave(SubPlan *dst, uint32_t type, uint32_t wasted,
                                                uint32_t bps, uint32_t order, uint32_t precision,
                                                uint32_t shift, uint32_t source, uint32_t bits,
                                                const WaveRice *w, uint32_t coeff /* of `lane` */,
                                                uint32_t N) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    if (lane == 0) {
        const uint32_t method = w ? w->method : 0u;
        const uint32_t count = w ? w->count : 0u;
        const uint32_t porder = count ? 31u - (uint32_t)__builtin_clz(count) : 0u;
        d[0] = type | (wasted << 8) | (bps << 16) | (order << 24);
        d[1] = precision | (shift << 8) | (method << 16) | (porder << 24);
        d[2] = source;                                       // source, reserved[3]
        d[3] = count;                                        // n_partitions
        d[4] = w ? (w->bp >= 0 ? N >> w->bp : N) : 0u;       // part_len
        d[5] = bits;
    }
    if (lane < 32) d[6 + lane] = lane < order ? coeff : 0u;  // coeffs
    uint8_t *b = reinterpret_cast<uint8_t *>(dst);
    const bool live = w && lane < w->count;
    b[24 + 128 + lane] = live ? w->price : 0;        // rice[lane]
    b[24 + 128 + 64 + lane] = live ? w->pesc : 0;    // escape_bits[lane]
}

// this lane's SPL consecutive samples of a row (16-byte loads when SPL is a multiple of 4,
// 8-byte loads otherwise: a lane starts SPL * 4 bytes after its neighbour)
template <int SPL>
__device__ __forceinline__ void load_lane(const int32_t *row, uint32_t lane, int32_t (&v)[SPL]) {
    if constexpr (SPL % 4 == 0) {
        const int4 *pp = reinterpret_cast<const int4 *>(row) + (SPL / 4) * lane;
#pragma unroll
        for (int q = 0; q < SPL / 4; q++) {
            const int4 a = pp[q];
            v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
        }
    } else {
        static_assert(SPL % 2 == 0, "samples per lane must be even");
        const int2 *pp = reinterpret_cast<const int2 *>(row) + (SPL / 2) * lane;
#pragma unroll
        for (int q = 0; q < SPL / 2; q++) {
            const int2 a = pp[q];
            v[2 * q] = a.x; v[2 * q + 1] = a.y;
        }
    }
}

template <int SPL, int MAXO>
__global__ void __launch_bounds__(WG, 2) k_cand64(Params p) {
    static_assert(MAXO <= SPL && (MAXO == 16 || MAXO == 32), "history comes from the previous lane only");
    constexpr uint32_t N = 64u * SPL;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t local = blockIdx.x * 4 + wave;
    if (local >= p.fcount * p.ncand) return;
    const uint32_t frame = p.f0 + local / p.ncand, cand = local % p.ncand;
    const size_t cidx = (size_t)frame * p.ncand + cand;
    // Everything the wave needs from memory is requested up front, so that one round trip is
    // exposed instead of four: candidate info, the LPC parameters (one coalesced dword per lane:
    // [0] status, [1] order | precision << 8 | shift << 16, [2..33] coefficients; handed out
    // with v_readlane later) and the samples.
    static_assert(sizeof(CandInfo) == 4 && sizeof(LpcParams) == 136, "layout");
    const uint32_t ci_raw = *reinterpret_cast<const uint32_t *>(p.cinfo + cidx);
    const uint32_t *lpw_p = reinterpret_cast<const uint32_t *>(p.lpc + cidx);
    const bool want_lpc = p.max_lpc_order > 0;
    const uint32_t lpw = (want_lpc && lane < 34) ? lpw_p[lane] : 1u;       // status 1 = no LPC
    const uint32_t qv = (want_lpc && lane < 32) ? lpw_p[2 + lane] : 0u;     // coefficient `lane`
    const CandSrc src = cand_src(p, frame, cand);
    int32_t x[SPL];
    load_lane<SPL>(src.a, lane, x);
    if (src.mode) {  // mid = (l + r) >> 1 (the shift joins the wasted-bits shift), side = l - r
        int32_t xb[SPL];
        load_lane<SPL>(src.b, lane, xb);
#pragma unroll
        for (int e = 0; e < SPL; e++)
            x[e] = src.mode == 1 ? (int32_t)((uint32_t)x[e] + (uint32_t)xb[e]) : combine(2, x[e], xb[e]);
    }
    CandInfo ci;
    memcpy(&ci, &ci_raw, 4);
    if (!ci.active) return;
    SubPlan *out = p.cand_plan + cidx;
    if (ci.is_const) {  // all zero -> CONSTANT(0) at the candidate's bps (encode.rs:2883-2887)
        store_plan_wave(out, FLACGPU_SUB_CONSTANT, 0, src.bps, 0, 0, 0, src.source, 8u + src.bps, nullptr, 0, N);
        return;
    }
    const uint32_t wasted = __builtin_amdgcn_readfirstlane((uint32_t)ci.wasted);
    const uint32_t bps_eff = __builtin_amdgcn_readfirstlane((uint32_t)ci.bps);
    {
        const uint32_t sh = wasted + (src.mode == 1 ? 1u : 0u);
#pragma unroll
        for (int e = 0; e < SPL; e++) x[e] >>= sh;
    }
    // the previous lane's last 4 samples (zeros before the block start)
    int32_t h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = lane_prev(x[SPL - 4 + k]);
    // ---- FIXED: abs sums of the iterated differences over [4, n) (encode.rs:3039-3073).
    // Values are biased by 2^30 (unsigned), |a - b| + acc is one v_sad_u32; u32 partial sums are
    // flushed every 8 terms (|d4| < 2^28 for <= 25-bit candidates).
    constexpr uint32_t BIAS = 1u << 30;
    uint64_t sm[5] = {0, 0, 0, 0, 0}, leaf[5];
    uint32_t w1 = 0, w2 = 0, w3 = 0;
    {
        const uint32_t hb0 = (uint32_t)h[0] + BIAS, hb1 = (uint32_t)h[1] + BIAS, hb2 = (uint32_t)h[2] + BIAS,
                       hb3 = (uint32_t)h[3] + BIAS;
        const uint32_t d1m3 = hb1 - hb0 + BIAS, d1m2 = hb2 - hb1 + BIAS, d1m1 = hb3 - hb2 + BIAS;
        const uint32_t d2m2 = d1m2 - d1m3 + BIAS, d2m1 = d1m1 - d1m2 + BIAS;
        uint32_t p0 = hb3, p1 = d1m1, p2 = d2m1, p3 = d2m1 - d2m2 + BIAS;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;  // contributions of samples 0..3
#pragma unroll
        for (int e = 0; e < SPL; e++) {
            const uint32_t xb = (uint32_t)x[e] + BIAS;
            const uint32_t d1 = xb - p0 + BIAS, d2 = d1 - p1 + BIAS, d3 = d2 - p2 + BIAS;
            a0 = __usad(xb, BIAS, a0);
            a1 = __usad(xb, p0, a1);
            a2 = __usad(d1, p1, a2);
            a3 = __usad(d2, p2, a3);
            a4 = __usad(d3, p3, a4);
            p0 = xb; p1 = d1; p2 = d2; p3 = d3;
            if (e == 0) w1 = a1;   // order K's warm-up = samples 0..K-1: sum |d_K| over them
            if (e == 1) w2 = a2;
            if (e == 2) w3 = a3;
            if (e == 3) { c0 = a0; c1 = a1; c2 = a2; c3 = a3; c4 = a4; }
            if ((e & 7) == 7 || e == SPL - 1) {
                sm[0] += a0; sm[1] += a1; sm[2] += a2; sm[3] += a3; sm[4] += a4;
                a0 = a1 = a2 = a3 = a4 = 0;
                __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting all 64 chains
            }
        }
        // sm: this lane's sum |d_K| over all 64 samples = the Rice search's leaf sums once the
        // warm-up of the chosen order is taken out (lane 0); the order choice sums [4, n)
        if (lane == 0) {
            leaf[0] = sm[0]; leaf[1] = sm[1] - w1; leaf[2] = sm[2] - w2; leaf[3] = sm[3] - w3; leaf[4] = sm[4] - c4;
            sm[0] -= c0; sm[1] -= c1; sm[2] -= c2; sm[3] -= c3; sm[4] -= c4;
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) leaf[k] = sm[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) sm[k] = wave_total_u48(sm[k]);
    uint32_t forder = 0;
#pragma unroll
    for (uint32_t k = 1; k <= 4; k++)
        if (sm[k] < sm[forder]) forder = k;  // min_by_key: first minimum wins
    WaveRice fw;
    switch (forder) {
    case 0: fw = wave_rice<SPL, 4>(FixedSrc<0, SPL>{x, h}, 0, p, leaf[0]); break;
    case 1: fw = wave_rice<SPL, 4>(FixedSrc<1, SPL>{x, h}, 1, p, leaf[1]); break;
    case 2: fw = wave_rice<SPL, 4>(FixedSrc<2, SPL>{x, h}, 2, p, leaf[2]); break;
    case 3: fw = wave_rice<SPL, 4>(FixedSrc<3, SPL>{x, h}, 3, p, leaf[3]); break;
    default: fw = wave_rice<SPL, 4>(FixedSrc<4, SPL>{x, h}, 4, p, leaf[4]); break;
    }
    const uint32_t fixed_bits = 8u + wasted + forder * bps_eff + fw.bits;
    const bool fixed_ok = fw.ok;
    // ---- LPC (encode.rs:3174-3203 + the same residual coding)
    bool lpc_ok = false;
    uint32_t lpc_bits = 0;
    WaveRice lw = fw;
    const uint32_t lstatus = sread(lpw, 0), lmeta = sread(lpw, 1);
    const uint32_t lprec = (lmeta >> 8) & 0xFF, lshift = (lmeta >> 16) & 0xFF;
    uint32_t lorder = 0;
    if (lstatus == 0) {
        lorder = lmeta & 0xFF;
        int32_t hp[MAXO];
#pragma unroll
        for (int k = 0; k < MAXO; k++) hp[k] = lane_prev(x[SPL - MAXO + k]);
        uint32_t ovf;
        switch ((lorder + 3) >> 2) {
        case 1: ovf = fir64<4, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
        case 2: ovf = fir64<8, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
        case 3: ovf = fir64<12, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
        case 4: ovf = fir64<16, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
        default:
            if constexpr (MAXO == 32) {
                switch ((lorder + 3) >> 2) {
                case 5: ovf = fir64<20, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
                case 6: ovf = fir64<24, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
                case 7: ovf = fir64<28, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
                default: ovf = fir64<32, SPL, MAXO>(x, hp, lpw, lorder, lshift); break;
                }
            } else {
                ovf = fir64<16, SPL, MAXO>(x, hp, lpw, lorder, lshift);
            }
            break;
        }
        if (__any(ovf)) {
            if (lane == 0) atomicAdd(&p.stats[0], 1u);
        } else {
            lw = wave_rice<SPL, MAXO>(StoredSrc<SPL>{x}, lorder, p);
            lpc_ok = lw.ok;
            if (!lpc_ok && lane == 0) atomicAdd(&p.stats[0], 1u);
            lpc_bits = 8u + wasted + lorder * bps_eff + 4u + 5u + lorder * lprec + lw.bits;
        }
    }
    // (Ok,Ok) -> min_by_key(written) with FIXED first; (Err,Ok) -> LPC; (Ok,Err) -> FIXED;
    // (Err,Err) -> VERBATIM; then the verbatim threshold (encode.rs:2929-2979)
    const bool use_lpc = lpc_ok && (!fixed_ok || lpc_bits < fixed_bits);
    const uint32_t best_bits = use_lpc ? lpc_bits : fixed_bits;
    const bool verbatim = (!fixed_ok && !lpc_ok) || !(best_bits < N * bps_eff);
    if (verbatim)
        store_plan_wave(out, FLACGPU_SUB_VERBATIM, wasted, bps_eff, 0, 0, 0, src.source,
                        8u + wasted + N * bps_eff, nullptr, 0, N);
    else if (use_lpc)
        store_plan_wave(out, FLACGPU_SUB_LPC, wasted, bps_eff, lorder, lprec, lshift, src.source, lpc_bits,
                        &lw, qv, N);
    else
        store_plan_wave(out, FLACGPU_SUB_FIXED, wasted, bps_eff, forder, 0, 0, src.source, fixed_bits,
                        &fw, 0, N);
}

// ---------------------------------------------------------------------------------
// K6: channel-assignment choice, one wave per frame (lane 0 decides, the wave copies)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_decide(Params p) {
    const uint32_t frame = p.f0 + blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t n = frame_len(p, frame);
    const SubPlan *cp = p.cand_plan + (size_t)frame * p.ncand;
    SubPlan *op = p.out_plan + (size_t)frame * p.channels;
    uint32_t sel[FLACGPU_MAX_CHANNELS];
    uint8_t assign = FLACGPU_ASSIGN_INDEPENDENT;
    if (p.stereo4) {
        if (p.exhaustive) {
            const uint32_t l = cp[0].bits, r = cp[1].bits, s = cp[3].bits;
            int b = 0;
            if (p.mid_side) {  // encode.rs:2747-2768
                const uint32_t m = cp[2].bits;
                uint32_t tot[4] = {l + r, l + s, s + r, m + s};
                for (int i = 1; i < 4; i++)
                    if (tot[i] < tot[b]) b = i;
            } else {  // encode.rs:2803-2820
                uint32_t tot[3] = {l + r, l + s, s + r};
                for (int i = 1; i < 3; i++)
                    if (tot[i] < tot[b]) b = i;
            }
            assign = b == 0 ? FLACGPU_ASSIGN_INDEPENDENT
                   : b == 1 ? FLACGPU_ASSIGN_LEFT_SIDE
                   : b == 2 ? FLACGPU_ASSIGN_SIDE_RIGHT : FLACGPU_ASSIGN_MID_SIDE;
        } else {
            assign = p.finfo[frame].assignment;
        }
        switch (assign) {
        case FLACGPU_ASSIGN_LEFT_SIDE: sel[0] = 0; sel[1] = 3; break;
        case FLACGPU_ASSIGN_SIDE_RIGHT: sel[0] = 3; sel[1] = 1; break;
        case FLACGPU_ASSIGN_MID_SIDE: sel[0] = 2; sel[1] = 3; break;
        default: sel[0] = 0; sel[1] = 1; break;
        }
    } else {
        for (uint32_t c = 0; c < p.channels; c++) sel[c] = c;
    }
    uint32_t body = 0;
    for (uint32_t c = 0; c < p.channels; c++) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(cp + sel[c]);
        uint32_t *d = reinterpret_cast<uint32_t *>(op + c);
        for (uint32_t i = lane; i < sizeof(SubPlan) / 4; i += 64) d[i] = s[i];
        body += cp[sel[c]].bits;
    }
    if (lane == 0) {
        flacgpu_frame_plan fp;
        fp.assignment = assign;
        fp.channels = (uint8_t)p.channels;
        fp.block_size = (uint16_t)n;
        fp.body_bits = body;
        p.frame_plan[frame] = fp;
    }
}

// ---------------------------------------------------------------------------------
// K7: residual signal of the chosen subframes, one workgroup per (frame, output channel).
// Out row: `order` warm-up samples then the residuals (VERBATIM: the samples; CONSTANT: [0]).
// dynamic LDS: x[n]
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG) k_emit(Params p) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ int32_t qlp[FLACGPU_MAX_LPC_ORDER];
    uint32_t frame, ch;
    map_block(blockIdx.x, p.channels, p.fcount, frame, ch);
    frame += p.f0;
    const uint32_t n = frame_len(p, frame);
    const uint32_t tid = threadIdx.x;
    const SubPlan *sp = p.out_plan + (size_t)frame * p.channels + ch;
    const uint32_t type = sp->type, order = sp->order, wasted = sp->wasted, shift = sp->shift;
    const uint32_t source = sp->source;
    uint32_t cand = source;
    if (p.stereo4) cand = source == FLACGPU_SRC_MID ? 2u : source == FLACGPU_SRC_SIDE ? 3u : source;
    const CandSrc src = cand_src(p, frame, cand);
    int32_t *out = p.residuals + ((size_t)frame * p.channels + ch) * p.block_size;
    int32_t *x = lds;
    if (type == FLACGPU_SUB_LPC && tid < FLACGPU_MAX_LPC_ORDER) qlp[tid] = sp->coeffs[tid];
    for (uint32_t i = tid; i < n; i += WG) x[i] = combine(src.mode, src.a[i], src.b[i]) >> wasted;
    __syncthreads();
    if (type == FLACGPU_SUB_LPC) {
        for (uint32_t i = tid; i < n; i += WG) {
            int32_t v = x[i];
            if (i >= order) {
                long long sum = 0;
                for (uint32_t j = 0; j < order; j++) sum += (long long)x[i - 1 - j] * (long long)qlp[j];
                v = (int32_t)((long long)v - (long long)(int32_t)(sum >> shift));
            }
            out[i] = v;
        }
    } else if (type == FLACGPU_SUB_FIXED) {
        for (uint32_t i = tid; i < n; i += WG) {
            long long v = x[i];
            if (i >= order) {
                if (order == 1) v = v - x[i - 1];
                else if (order == 2) v = v - 2ll * x[i - 1] + x[i - 2];
                else if (order == 3) v = v - 3ll * x[i - 1] + 3ll * x[i - 2] - x[i - 3];
                else if (order == 4) v = v - 4ll * x[i - 1] + 6ll * x[i - 2] - 4ll * x[i - 3] + x[i - 4];
            }
            out[i] = (int32_t)v;
        }
    } else {
        for (uint32_t i = tid; i < n; i += WG) out[i] = x[i];
    }
}


// ---------------------------------------------------------------------------------
// Device-side frame assembly.
//   K8  k_layout  frame sizes (header + ceil(body/8) + 2) -> exclusive prefix sum
//   K9  k_pack    one workgroup per (frame, subframe): the subframe's bits are built in LDS
//                 (header, warm-up, LPC parameters by a few lanes; residual codes at bit
//                 offsets from a workgroup prefix scan of the code lengths) and copied to
//                 the frame's position in HBM with a funnel shift; words shared with a
//                 neighbouring subframe / frame are OR-ed atomically (the buffer is zeroed
//                 first, so byte-alignment padding is free)
//   K10 k_crc     CRC-16 of every frame: 256 lanes each fold a slice, partial CRCs are
//                 combined with x^(8*len) mod P multiplications (GF(2) linearity)
// Replaces stream.rs:242-276 (FrameHeader::build + CRC-8), stream.rs:1390-1413,
// 1603-1619, encode.rs:3078-3135 (subframe serialisation), 3834-3907 (residual block),
// 2408-2409 (align + CRC-16).
// ---------------------------------------------------------------------------------
struct PackParams {
    uint64_t first_frame_number;
    uint32_t sample_rate;
    uint32_t *out_words;      // packed bytes, viewed as big-endian-filled 32-bit words
    uint64_t *frame_off;      // [n_frames + 1] byte offsets
    uint64_t cap_bytes;
};

struct HeaderCodes {
    uint32_t bcode, bextra_bits, rcode, rextra_bits, rextra, fn_bytes;
};
__device__ __forceinline__ HeaderCodes header_codes(uint32_t n, uint32_t rate, uint64_t fn) {
    HeaderCodes h;
    h.bextra_bits = 0;
    switch (n) {  // BlockSize::try_from, stream.rs:531-558
    case 192: h.bcode = 1; break;
    case 576: h.bcode = 2; break;
    case 1152: h.bcode = 3; break;
    case 2304: h.bcode = 4; break;
    case 4608: h.bcode = 5; break;
    case 256: h.bcode = 8; break;
    case 512: h.bcode = 9; break;
    case 1024: h.bcode = 10; break;
    case 2048: h.bcode = 11; break;
    case 4096: h.bcode = 12; break;
    case 8192: h.bcode = 13; break;
    case 16384: h.bcode = 14; break;
    case 32768: h.bcode = 15; break;
    default:
        if (n <= 256) { h.bcode = 6; h.bextra_bits = 8; }
        else { h.bcode = 7; h.bextra_bits = 16; }
    }
    h.rextra_bits = 0;
    h.rextra = 0;
    switch (rate) {  // SampleRate::try_from, stream.rs:767-800
    case 88200: h.rcode = 1; break;
    case 176400: h.rcode = 2; break;
    case 192000: h.rcode = 3; break;
    case 8000: h.rcode = 4; break;
    case 16000: h.rcode = 5; break;
    case 22050: h.rcode = 6; break;
    case 24000: h.rcode = 7; break;
    case 32000: h.rcode = 8; break;
    case 44100: h.rcode = 9; break;
    case 48000: h.rcode = 10; break;
    case 96000: h.rcode = 11; break;
    default:
        if (rate % 1000 == 0 && rate / 1000 < 255) { h.rcode = 12; h.rextra_bits = 8; h.rextra = rate / 1000; }
        else if (rate % 10 == 0 && rate / 10 < 65535) { h.rcode = 14; h.rextra_bits = 16; h.rextra = rate / 10; }
        else if (rate < 65535) { h.rcode = 13; h.rextra_bits = 16; h.rextra = rate; }
        else h.rcode = 0;
    }
    h.fn_bytes = fn <= 0x7F ? 1 : fn <= 0x7FF ? 2 : fn <= 0xFFFF ? 3 : fn <= 0x1FFFFF ? 4
               : fn <= 0x3FFFFFF ? 5 : fn <= 0x7FFFFFFFull ? 6 : 7;
    return h;
}
__device__ __forceinline__ uint32_t header_bytes(const HeaderCodes &h) {
    return 4 + h.fn_bytes + h.bextra_bits / 8 + h.rextra_bits / 8 + 1;
}

// frames [p.f0, p.f0 + p.fcount): byte offsets continue from frame_off[p.f0] (0 for the first
// range; written by the previous range's launch otherwise)
__global__ void __launch_bounds__(1024) k_layout(Params p, PackParams q) {
    __shared__ uint64_t wave_tot[16];
    const uint32_t tid = threadIdx.x;
    // lane t owns the contiguous frames [t*chunk, (t+1)*chunk): serial sum, one block scan
    const uint32_t chunk = (p.fcount + 1023) / 1024;
    const uint32_t end = p.f0 + p.fcount;
    const uint32_t lo = p.f0 + tid * chunk < end ? p.f0 + tid * chunk : end;
    const uint32_t hi = lo + chunk < end ? lo + chunk : end;
    auto frame_bytes = [&](uint32_t f) -> uint64_t {
        const flacgpu_frame_plan fp = p.frame_plan[f];
        HeaderCodes h = header_codes(fp.block_size, q.sample_rate, q.first_frame_number + f);
        return header_bytes(h) + ((uint64_t)fp.body_bits + 7) / 8 + 2;
    };
    const uint64_t base = p.f0 ? q.frame_off[p.f0] : 0ull;
    uint64_t mine = 0;
    for (uint32_t f = lo; f < hi; f++) mine += frame_bytes(f);
    uint64_t v = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t t = __shfl_up(v, off, 64);
        if ((tid & 63) >= (uint32_t)off) v += t;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = v;
    __syncthreads();
    uint64_t prefix = base + v - mine;
    for (uint32_t w = 0; w < (tid >> 6); w++) prefix += wave_tot[w];
    for (uint32_t f = lo; f < hi; f++) {
        if (f != p.f0 || p.f0 == 0) q.frame_off[f] = prefix;
        prefix += frame_bytes(f);
    }
    if (tid == 1023) q.frame_off[end] = prefix;
}

// zero the part of the output buffer the frames will occupy (16 bytes per lane)
__global__ void __launch_bounds__(WG) k_zero(PackParams q, uint32_t n_frames) {
    const uint64_t total = q.frame_off[n_frames];
    const uint64_t n16 = (total + 15) / 16 + 1;
    uint4 *o = reinterpret_cast<uint4 *>(q.out_words);
    for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * WG)
        o[i] = make_uint4(0, 0, 0, 0);
}

// words reserved for a subframe's bit string: a chosen subframe is never longer than its
// VERBATIM form (<= 40 + 33 n bits) plus the 16-byte frame header; multiple of 4 words
__host__ __device__ constexpr uint32_t pack_sb_words(uint32_t block_size) {
    return ((block_size * 33u / 32u + 32u) + 3u) & ~3u;
}

// OR a field of nbits (<= 32) at bit position pos of an MSB-first bit string held in LDS words
__device__ __forceinline__ void lds_put(uint32_t *sb, uint32_t pos, uint32_t v, uint32_t nbits) {
    if (nbits == 0) return;
    if (nbits < 32) v &= (1u << nbits) - 1u;
    const uint32_t w = pos >> 5, off = pos & 31;
    if (off + nbits <= 32) {
        atomicOr(&sb[w], v << (32 - off - nbits));
    } else {
        const uint32_t lo = off + nbits - 32;  // bits spilling into the next word
        atomicOr(&sb[w], v >> lo);
        atomicOr(&sb[w + 1], v << (32 - lo));
    }
}

// Generic subframe writer (any block length): residual rows read from HBM (written by k_emit).
// One subframe's bits are written into the MSB-first LDS bit string `sb` (zeroed by the caller, which
// has NOT synchronised yet) starting at bit `base`; `with_header` also writes the frame header at
// bit 0.  Ends WITHOUT a barrier.
__device__ __forceinline__ void pack_subframe(const Params &p, const PackParams &q, uint32_t frame,
                                              uint32_t ch, uint32_t n, uint32_t *sb, uint32_t base,
                                              bool with_header, const HeaderCodes &hc, uint64_t fn) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint8_t hdr[16];
    const uint32_t tid = threadIdx.x;
    const SubPlan *sp = p.out_plan + (size_t)frame * p.channels + ch;
    const uint32_t type = sp->type, order = sp->order, wasted = sp->wasted, bps = sp->bps;
    // residual row: every lane reads its own contiguous run straight from HBM/L2 (twice:
    // lengths, then codes; the second pass hits L1/L2)
    const int32_t *__restrict__ r = p.residuals + ((size_t)frame * p.channels + ch) * p.block_size;
    auto rd = [&](uint32_t i) -> int32_t { return r[i]; };
    __syncthreads();

    if (tid == 0) {
        if (with_header) {  // FrameHeader::build, stream.rs:242-276 (+ CRC-8, :194-197)
            const flacgpu_frame_plan fp = p.frame_plan[frame];
            uint32_t k = 0;
            hdr[k++] = 0xFF;
            hdr[k++] = 0xF8;  // sync 0b111111111111100 + blocking strategy 0
            hdr[k++] = (uint8_t)((hc.bcode << 4) | hc.rcode);
            const uint32_t acode = fp.assignment == FLACGPU_ASSIGN_INDEPENDENT ? p.channels - 1 : fp.assignment;
            const uint32_t pcode = p.bps == 8 ? 1 : p.bps == 12 ? 2 : p.bps == 16 ? 4 : p.bps == 20 ? 5
                                 : p.bps == 24 ? 6 : p.bps == 32 ? 7 : 0;
            hdr[k++] = (uint8_t)((acode << 4) | (pcode << 1));
            if (hc.fn_bytes == 1) {
                hdr[k++] = (uint8_t)fn;
            } else {  // UTF-8-like frame number, stream.rs:1264-1325
                const uint32_t nb = hc.fn_bytes;
                const uint32_t lead = (0xFFu << (8 - nb)) & 0xFF;
                hdr[k++] = (uint8_t)(lead | (uint32_t)(fn >> (6 * (nb - 1))));
                for (int b = (int)nb - 2; b >= 0; b--) hdr[k++] = (uint8_t)(0x80 | ((fn >> (6 * b)) & 0x3F));
            }
            if (hc.bextra_bits == 8) hdr[k++] = (uint8_t)(n - 1);
            else if (hc.bextra_bits == 16) { hdr[k++] = (uint8_t)((n - 1) >> 8); hdr[k++] = (uint8_t)(n - 1); }
            if (hc.rextra_bits == 8) hdr[k++] = (uint8_t)hc.rextra;
            else if (hc.rextra_bits == 16) { hdr[k++] = (uint8_t)(hc.rextra >> 8); hdr[k++] = (uint8_t)hc.rextra; }
            uint32_t crc = 0;  // CRC-8, poly 0x07 (crc.rs:99-128)
            for (uint32_t i = 0; i < k; i++) {
                crc ^= hdr[i];
                for (int b = 0; b < 8; b++) crc = (crc & 0x80) ? ((crc << 1) ^ 0x07) & 0xFF : (crc << 1) & 0xFF;
            }
            hdr[k++] = (uint8_t)crc;
            for (uint32_t i = 0; i < k; i++) lds_put(sb, 8 * i, hdr[i], 8);
        }
        // SubframeHeader, stream.rs:1390-1413
        const uint32_t tcode = type == FLACGPU_SUB_CONSTANT ? 0u : type == FLACGPU_SUB_VERBATIM ? 1u
                             : type == FLACGPU_SUB_FIXED ? 8u + order : 31u + order;
        lds_put(sb, base, tcode, 7);
        if (wasted) {
            lds_put(sb, base + 7, 1, 1);
            lds_put(sb, base + 8 + (wasted - 1), 1, 1);  // wasted-1 zeros then a one
        }
        if (type == FLACGPU_SUB_CONSTANT) lds_put(sb, base + 8 + wasted, (uint32_t)rd(0), bps);
    }
    const uint32_t body0 = base + 8 + wasted;
    if (type == FLACGPU_SUB_VERBATIM) {
        for (uint32_t i = tid; i < n; i += WG) lds_put(sb, body0 + i * bps, (uint32_t)rd(i), bps);
    } else if (type == FLACGPU_SUB_FIXED || type == FLACGPU_SUB_LPC) {
        // warm-up, precision, shift, coefficients (encode.rs:3083-3085, 3118-3133)
        if (tid < order) lds_put(sb, body0 + tid * bps, (uint32_t)rd(tid), bps);
        uint32_t pos = body0 + order * bps;
        if (type == FLACGPU_SUB_LPC) {
            const uint32_t prec = sp->precision;
            if (tid == 32) {
                lds_put(sb, pos, prec - 1, 4);
                lds_put(sb, pos + 4, sp->shift, 5);
            }
            if (tid >= 64 && tid < 64 + order) lds_put(sb, pos + 9 + (tid - 64) * prec, (uint32_t)sp->coeffs[tid - 64], prec);
            pos += 9 + order * prec;
        }
        // residual block (encode.rs:3944-3961, 3898-3907)
        const uint32_t method = sp->coding_method, hb = method ? 5u : 4u, esc_code = method ? 31u : 15u;
        if (tid == 96) {
            lds_put(sb, pos, method, 2);
            lds_put(sb, pos + 2, sp->partition_order, 4);
        }
        pos += 6;
        const uint32_t np = sp->n_partitions, plen = sp->part_len;
        const uint32_t first_j = plen ? n / plen - np : 0;  // chunks lying inside the warm-up
        const uint32_t ept = (n + WG - 1) / WG;
        const uint32_t lo = tid * ept > order ? tid * ept : order;
        const uint32_t hi = (tid + 1) * ept < n ? (tid + 1) * ept : n;
        // pass 1: code lengths of this lane's residuals (+ partition headers)
        uint32_t mybits = 0;
        if (lo < hi) {
            uint32_t pj = lo / plen;                 // partition (block-aligned index)
            uint32_t bound = (pj + 1) * plen;
            uint32_t k = sp->rice[pj - first_j], eb = sp->escape_bits[pj - first_j];
            const uint32_t pstart0 = pj * plen > order ? pj * plen : order;
            if (lo == pstart0) mybits += hb + (k == 0xFF ? 5u : 0u);
            for (uint32_t i = lo; i < hi; i++) {
                if (i == bound) {
                    pj++;
                    bound += plen;
                    k = sp->rice[pj - first_j];
                    eb = sp->escape_bits[pj - first_j];
                    mybits += hb + (k == 0xFF ? 5u : 0u);
                }
                mybits += (k != 0xFF) ? (zigzag(r[i]) >> k) + 1u + k : eb;
            }
        }
        // exclusive scan of mybits over the workgroup
        uint32_t v = mybits;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(v, off, 64);
            if ((tid & 63) >= (uint32_t)off) v += t;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = v;
        __syncthreads();
        uint32_t mypos = pos + v - mybits;
        for (uint32_t w = 0; w < (tid >> 6); w++) mypos += wave_tot[w];
        // pass 2: emit
        if (lo < hi) {
            uint32_t pj = lo / plen;
            uint32_t bound = (pj + 1) * plen;
            uint32_t k = sp->rice[pj - first_j], eb = sp->escape_bits[pj - first_j];
            const uint32_t pstart0 = pj * plen > order ? pj * plen : order;
            if (lo == pstart0) {
                if (k != 0xFF) { lds_put(sb, mypos, k, hb); mypos += hb; }
                else { lds_put(sb, mypos, esc_code, hb); lds_put(sb, mypos + hb, eb, 5); mypos += hb + 5; }
            }
            for (uint32_t i = lo; i < hi; i++) {
                if (i == bound) {
                    pj++;
                    bound += plen;
                    k = sp->rice[pj - first_j];
                    eb = sp->escape_bits[pj - first_j];
                    if (k != 0xFF) { lds_put(sb, mypos, k, hb); mypos += hb; }
                    else { lds_put(sb, mypos, esc_code, hb); lds_put(sb, mypos + hb, eb, 5); mypos += hb + 5; }
                }
                if (k != 0xFF) {
                    const uint32_t u = zigzag(r[i]);
                    const uint32_t qn = u >> k;
                    mypos += qn;  // unary zeros (buffer is pre-zeroed)
                    lds_put(sb, mypos, (1u << k) | (u & ((1u << k) - 1u)), k + 1);
                    mypos += k + 1;
                } else if (eb) {
                    lds_put(sb, mypos, (uint32_t)r[i], eb);
                    mypos += eb;
                }
            }
        }
        if (n == order && tid == 0) {  // no residuals at all: lone partition header
            const uint32_t k = sp->rice[0], eb = sp->escape_bits[0];
            if (k != 0xFF) lds_put(sb, pos, k, hb);
            else { lds_put(sb, pos, esc_code, hb); lds_put(sb, pos + hb, eb, 5); }
        }
    }
}

__global__ void __launch_bounds__(WG) k_pack(Params p, PackParams q) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    uint32_t frame, ch;
    map_block(blockIdx.x, p.channels, p.fcount, frame, ch);
    frame += p.f0;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = frame_len(p, frame);
    const uint32_t sub_bits = p.out_plan[(size_t)frame * p.channels + ch].bits;

    // position of this subframe inside the frame
    const uint64_t fn = q.first_frame_number + frame;
    const HeaderCodes hc = header_codes(n, q.sample_rate, fn);
    const uint32_t hbytes = header_bytes(hc);
    uint32_t start_bit = hbytes * 8;
    for (uint32_t c = 0; c < ch; c++) start_bit += p.out_plan[(size_t)frame * p.channels + c].bits;
    const uint32_t prefix_bits = ch == 0 ? hbytes * 8 : 0;  // subframe 0 also carries the header
    const uint32_t total_bits = prefix_bits + sub_bits;
    const uint32_t nwords = (total_bits + 31) / 32 + 1;

    uint32_t *sb = reinterpret_cast<uint32_t *>(lds);  // the subframe's bit string
    for (uint32_t i = tid; i < nwords; i += WG) sb[i] = 0;
    pack_subframe(p, q, frame, ch, n, sb, prefix_bits, ch == 0, hc, fn);
    __syncthreads();
    // copy out: absolute bit position of sb[0] in the output stream
    const uint64_t abs_bit = q.frame_off[frame] * 8ull + (ch == 0 ? 0u : start_bit);
    const uint64_t w0 = abs_bit >> 5;
    const uint32_t s = (uint32_t)(abs_bit & 31);
    const uint32_t nout = (s + total_bits + 31) / 32;
    for (uint32_t j = tid; j < nout; j += WG) {
        uint32_t val;
        if (s == 0) val = sb[j];
        else val = (j ? (sb[j - 1] << (32 - s)) : 0u) | (sb[j] >> s);
        const uint32_t be = __builtin_bswap32(val);
        if (j == 0 || j + 1 == nout) atomicOr(&q.out_words[w0 + j], be);
        else q.out_words[w0 + j] = be;
    }
}

// CRC-16 (poly 0x8005, MSB first, init 0; crc.rs:142-188) of every packed frame
__device__ __forceinline__ uint32_t gf_mulmod(uint32_t a, uint32_t b) {  // a*b mod P over GF(2)
    uint32_t r = 0;
    for (int i = 15; i >= 0; i--) {
        r = (r & 0x8000) ? ((r << 1) ^ 0x8005) & 0xFFFF : (r << 1) & 0xFFFF;
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}
constexpr uint32_t CRC_CHUNK = 16384;  // bytes staged in LDS per pass: 256 lanes x 64 B

constexpr uint32_t gf_mulmod_c(uint32_t a, uint32_t b) {  // compile-time a*b mod P over GF(2)
    uint32_t r = 0;
    for (int i = 15; i >= 0; i--) {
        r = (r & 0x8000) ? ((r << 1) ^ 0x8005) & 0xFFFF : (r << 1) & 0xFFFF;
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}
// W[k] = x^(512 k) mod P: weight of a 64-byte slice that is followed by k more slices
struct CrcWeights {
    uint16_t w[WG + 1];
    constexpr CrcWeights() : w() {
        uint32_t x512 = 0x100;
        for (int i = 0; i < 6; i++) x512 = gf_mulmod_c(x512, x512);
        uint32_t v = 1;
        for (int k = 0; k <= WG; k++) {
            w[k] = (uint16_t)v;
            v = gf_mulmod_c(v, x512);
        }
    }
};
__constant__ CrcWeights kCrcW = CrcWeights();

// W17[k] = x^(544 k) mod P: weight of a 68-byte slice that is followed by k more slices
struct CrcWeights17 {
    uint16_t w[512 + 1];
    constexpr CrcWeights17() : w() {
        uint32_t x32 = 0x100;                       // x^8
        x32 = gf_mulmod_c(x32, x32);                // x^16
        x32 = gf_mulmod_c(x32, x32);                // x^32
        uint32_t x544 = 1;
        for (int i = 0; i < 17; i++) x544 = gf_mulmod_c(x544, x32);
        uint32_t v = 1;
        for (int k = 0; k <= 512; k++) {
            w[k] = (uint16_t)v;
            v = gf_mulmod_c(v, x544);
        }
    }
};
__constant__ CrcWeights17 kCrcW17 = CrcWeights17();

// slicing-by-4 tables of the CRC-16: T[k][b] = CRC state after byte b followed by k zero bytes
struct CrcTables {
    uint16_t t[4][256];
    constexpr CrcTables() : t() {
        for (int b = 0; b < 256; b++) {
            uint32_t c = (uint32_t)b << 8;
            for (int k = 0; k < 4; k++) {
                for (int i = 0; i < 8; i++) c = (c & 0x8000) ? ((c << 1) ^ 0x8005) & 0xFFFF : (c << 1) & 0xFFFF;
                t[k][b] = (uint16_t)c;
            }
        }
    }
};
__constant__ __attribute__((aligned(16))) CrcTables kCrcT = CrcTables();

// words of LDS a whole frame of 4096-sample subframes may need (VERBATIM everywhere + header +
// CRC-16 + one guard word for the funnel shifts); multiple of 4 words
__host__ __device__ constexpr uint32_t frame_fb_words(uint32_t channels, uint32_t bps, uint32_t n = FN) {
    return (((16u + 2u) * 8u + channels * (n * (bps + 1u) + 64u) + 31u) / 32u + 2u + 3u) & ~3u;
}

// ---------------------------------------------------------------------------------
// Wave-per-subframe frame assembly (4096-sample frames, LPC order <= 16, <= 4 channels): the
// workgroup is one frame, wave c is subframe c.  Lane l owns samples [64 l, 64 l + 64) in
// registers as in k_cand64: the residual is recomputed in place (FIR with wave-uniform
// coefficients / iterated differences), the code lengths of a lane's run are summed, one DPP scan
// gives every lane its bit offset, and the lane then streams its codes through a 64-bit shift
// register into the frame's LDS bit string one finished 32-bit word at a time.  The only workgroup
// barriers are the ones around the CRC-16.
// ---------------------------------------------------------------------------------
struct BitRun {  // a lane's contiguous MSB-first bit run inside an LDS word array
    uint32_t *word;   // next word to complete
    uint64_t acc;     // pending bits, right-aligned; bits above `fill` are stale and never read
    uint32_t fill;    // pending bit count (< 32 between calls); starts at the run's bit offset
    __device__ __forceinline__ void init(uint32_t *sb, uint32_t pos) {
        word = sb + (pos >> 5);
        fill = pos & 31;  // the bits before the run belong to someone else: contribute zeros
        acc = 0;
    }
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb) {  // nb <= 32, v < 2^nb
        acc = (acc << nb) | v;
        fill += nb;
        if (fill >= 32) {
            fill -= 32;
            atomicOr(word, (uint32_t)(acc >> fill));  // words at the ends of a run are shared
            word++;
        }
    }
    // branch-free put for nb <= 32 with nb + (pending < 32) < 64: always one LDS OR (of 0 when no
    // word was completed) -- no exec-mask juggling, 10 VALU instead of 8 VALU + 7 SALU
    __device__ __forceinline__ void put_sel(uint32_t v, uint32_t nb) {
        acc = (acc << nb) | v;
        fill += nb;
        const bool full = fill >= 32;
        fill &= 31;
        const uint32_t w = (uint32_t)(acc >> fill);
        atomicOr(word, full ? w : 0u);
        word += full ? 1 : 0;
    }
    __device__ __forceinline__ void finish() {
        if (fill) atomicOr(word, (uint32_t)(acc << (32 - fill)));
    }
};

template <int SPL, int MAXO>
__device__ __forceinline__ void wave_subframe(const Params &p, uint32_t frame, uint32_t ch,
                                              uint32_t *sb, uint32_t base) {
    const uint32_t lane = threadIdx.x & 63;
    const SubPlan *sp = p.out_plan + (size_t)frame * p.channels + ch;
    // One memory round trip: the whole 280-byte plan as one dword per lane (fields, coefficients
    // and partition parameters are handed out with readlane / bpermute) and, at the same time,
    // the samples -- for stereo frames both input rows, whatever candidate the plan names.
    static_assert(sizeof(SubPlan) == 280, "plan layout");
    const uint32_t *d = reinterpret_cast<const uint32_t *>(sp);
    const uint32_t pw = d[lane];                       // dwords 0..63
    const uint32_t pw2 = lane < 6 ? d[64 + lane] : 0;  // dwords 64..69
    const int32_t *rowa, *rowb;
    if (p.stereo4) {
        rowa = p.planar + (size_t)frame * 2 * p.ldb;
        rowb = rowa + p.ldb;
    } else {
        rowa = p.planar + ((size_t)frame * p.channels + ch) * p.ldb;  // source == ch
        rowb = rowa;
    }
    int32_t x[SPL];
    load_lane<SPL>(rowa, lane, x);
    if (p.stereo4) {
        int32_t xb[SPL];
        load_lane<SPL>(rowb, lane, xb);
        const uint32_t srcid = sread(pw, 2) & 0xFF;
        // L: a, R: b, mid: a + b (>> 1 below), side: a - b
        const int32_t ca = srcid == 1 ? 0 : 1;
        const int32_t cb = srcid == 0 ? 0 : srcid == FLACGPU_SRC_SIDE ? -1 : 1;
#pragma unroll
        for (int e = 0; e < SPL; e++) x[e] = x[e] * ca + xb[e] * cb;
    }
    const uint32_t d0 = sread(pw, 0), d1 = sread(pw, 1), d2 = sread(pw, 2);
    const uint32_t type = d0 & 0xFF, wasted = (d0 >> 8) & 0xFF, bps = (d0 >> 16) & 0xFF, order = d0 >> 24;
    const uint32_t prec = d1 & 0xFF, shift = (d1 >> 8) & 0xFF, method = (d1 >> 16) & 0xFF, porder = d1 >> 24;
    {
        const uint32_t sh = wasted + ((p.stereo4 && (d2 & 0xFF) == FLACGPU_SRC_MID) ? 1u : 0u);
#pragma unroll
        for (int e = 0; e < SPL; e++) x[e] >>= sh;
    }
    // Rice parameters of the partition this lane's samples lie in (a partition is 2^(6 - porder)
    // lanes; the warm-up never swallows a whole partition since order <= 16 <= SPL):
    // rice[pj] is byte pj of dwords 38..53, escape_bits[pj] byte pj of dwords 54..69
    const uint32_t lpp = 6u - (porder < 6u ? porder : 6u);  // log2(lanes per partition)
    const uint32_t pj = lane >> lpp;
    const bool coded = type == FLACGPU_SUB_FIXED || type == FLACGPU_SUB_LPC;
    const uint32_t rw = __shfl(pw, (int)(38 + (pj >> 2)), 64);
    const uint32_t ei = 54 + (pj >> 2);
    const uint32_t ew1 = __shfl(pw, (int)(ei & 63), 64), ew2 = __shfl(pw2, (int)(ei & 63), 64);
    const uint32_t ew = ei < 64 ? ew1 : ew2;
    const uint32_t k = coded ? (rw >> (8 * (pj & 3))) & 0xFF : 0u;
    const uint32_t eb = coded ? (ew >> (8 * (pj & 3))) & 0xFF : 0u;
    const uint32_t cw = pw;  // LPC coefficient j sits on lane 6 + j
    const uint32_t smask = bps >= 32 ? 0xFFFFFFFFu : (1u << bps) - 1u;
    // SubframeHeader, stream.rs:1390-1413
    if (lane == 0) {
        const uint32_t tcode = type == FLACGPU_SUB_CONSTANT ? 0u : type == FLACGPU_SUB_VERBATIM ? 1u
                             : type == FLACGPU_SUB_FIXED ? 8u + order : 31u + order;
        lds_put(sb, base, tcode, 7);
        if (wasted) {
            lds_put(sb, base + 7, 1, 1);
            lds_put(sb, base + 8 + (wasted - 1), 1, 1);  // wasted-1 zeros then a one
        }
        if (type == FLACGPU_SUB_CONSTANT) lds_put(sb, base + 8 + wasted, (uint32_t)x[0], bps);
    }
    const uint32_t body0 = base + 8 + wasted;
    if (type == FLACGPU_SUB_CONSTANT) return;
    BitRun br;
    if (type == FLACGPU_SUB_VERBATIM) {
        br.init(sb, body0 + lane * (uint32_t)SPL * bps);
#pragma unroll
        for (int e = 0; e < SPL; e++) br.put((uint32_t)x[e] & smask, bps);
        br.finish();
        return;
    }
    // warm-up samples (encode.rs:3083-3085, 3118-3133): the first `order` samples of lane 0
    if (lane == 0) {
        br.init(sb, body0);
#pragma unroll
        for (int e = 0; e < MAXO; e++)
            if ((uint32_t)e < order) br.put((uint32_t)x[e] & smask, bps);
        br.finish();
    }
    // Code lengths are summed where each residual is produced; Rice-coded lanes keep zigzag(r)
    // in place of r.  Warm-up samples (lane 0, e < order) carry no residual.
    const uint32_t first = lane == 0 ? order : 0u;
    const bool rice = k != 0xFF;
    struct LenAcc {
        uint32_t qsum, qmax, ks, first;
        bool rice;
        __device__ __forceinline__ int32_t operator()(int e, int32_t r) {
            uint32_t u = zigzag(r);
            if (e < MAXO) u = (uint32_t)e >= first ? u : 0u;
            const uint32_t q = u >> ks;
            qsum += q;
            qmax = q > qmax ? q : qmax;
            return rice ? (int32_t)u : r;
        }
    } len{0u, 0u, rice ? k : 0u, first, rice};
    uint32_t resid_pos = body0 + order * bps;
    if (type == FLACGPU_SUB_LPC) {
        if (lane == 32) {
            lds_put(sb, resid_pos, prec - 1, 4);
            lds_put(sb, resid_pos + 4, shift, 5);
        }
        if (lane >= 6 && lane < 6 + order) lds_put(sb, resid_pos + 9 + (lane - 6) * prec, cw, prec);
        resid_pos += 9 + order * prec;
        int32_t hp[MAXO];
#pragma unroll
        for (int kk = 0; kk < MAXO; kk++) hp[kk] = lane_prev(x[SPL - MAXO + kk]);
        switch ((order + 3) >> 2) {  // the residual, in place (encode.rs:3181-3197)
        case 1: fir64<4, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
        case 2: fir64<8, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
        case 3: fir64<12, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
        case 4: fir64<16, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
        default:
            if constexpr (MAXO == 32) {
                switch ((order + 3) >> 2) {
                case 5: fir64<20, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
                case 6: fir64<24, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
                case 7: fir64<28, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
                default: fir64<32, SPL, MAXO, 6>(x, hp, cw, order, shift, len); break;
                }
            } else {
                fir64<16, SPL, MAXO, 6>(x, hp, cw, order, shift, len);
            }
            break;
        }
    } else {  // FIXED: iterated differences in place (encode.rs:3039-3060)
        int32_t h[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) h[kk] = lane_prev(x[SPL - 4 + kk]);
        int32_t q0 = h[3], q1 = h[3] - h[2], q2 = q1 - (h[2] - h[1]);
        int32_t q3 = q2 - ((h[2] - h[1]) - (h[1] - h[0]));
#pragma unroll
        for (int e = 0; e < SPL; e++) {
            const int32_t e1 = x[e] - q0, e2 = e1 - q1, e3 = e2 - q2, e4 = e3 - q3;
            q0 = x[e]; q1 = e1; q2 = e2; q3 = e3;
            x[e] = len(e, order == 0 ? x[e] : order == 1 ? e1 : order == 2 ? e2 : order == 3 ? e3 : e4);
            if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // residual block (encode.rs:3944-3961, 3898-3907)
    const uint32_t hb = method ? 5u : 4u, esc_code = method ? 31u : 15u;
    if (lane == 1) {
        lds_put(sb, resid_pos, method, 2);
        lds_put(sb, resid_pos + 2, porder, 4);
    }
    const uint32_t pos = resid_pos + 6;
    const bool head = (lane & ((1u << lpp) - 1u)) == 0;   // this lane starts a partition
    const uint32_t cnt = (uint32_t)SPL - first;
    const uint32_t mybits = (head ? hb + (rice ? 0u : 5u) : 0u) + (rice ? len.qsum + cnt * (k + 1u) : cnt * eb);
    const uint32_t incl = wave_scan_u32(mybits);
    br.init(sb, pos + incl - mybits);
    if (head) {
        if (rice) br.put(k, hb);
        else { br.put(esc_code, hb); br.put(eb, 5); }
    }
    // no code of the wave longer than 32 bits (the rule): one branch-free put per sample
    const bool short_codes = !__any(rice && len.qmax + k + 1u > 32u);
    if (rice && short_codes) {
        const uint32_t stop = 1u << k, lowmask = stop - 1u, k1 = k + 1u;
#pragma unroll
        for (int e = 0; e < SPL; e++) {
            const uint32_t u = (uint32_t)x[e];
            uint32_t nb = (u >> k) + k1;
            uint32_t v = stop | (u & lowmask);
            if (e < MAXO) {  // warm-up samples of lane 0 carry no residual
                nb = (uint32_t)e >= first ? nb : 0u;
                v = (uint32_t)e >= first ? v : 0u;
            }
            br.put_sel(v, nb);
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    } else if (rice) {
        const uint32_t stop = 1u << k, lowmask = stop - 1u;
#pragma unroll
        for (int e = 0; e < SPL; e++) {
            if (e >= MAXO || (uint32_t)e >= first) {
                const uint32_t u = (uint32_t)x[e];
                uint32_t qn = u >> k;
                const uint32_t v = stop | (u & lowmask);
                if (qn + k + 1u > 32u) {  // long unary run: zeros in pieces
                    while (qn >= 32u) { br.put(0, 32); qn -= 32u; }
                    br.put(0, qn);
                    br.put(v, k + 1u);
                } else {
                    br.put(v, qn + k + 1u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (eb) {
        const uint32_t emask = eb >= 32 ? 0xFFFFFFFFu : (1u << eb) - 1u;
#pragma unroll
        for (int e = 0; e < SPL; e++)
            if (e >= MAXO || (uint32_t)e >= first) br.put((uint32_t)x[e] & emask, eb);
    }
    br.finish();
}

template <int NT, int SPL, int MAXO>
__global__ void __launch_bounds__(NT, 2) k_frame64(Params p, PackParams q) {
    constexpr uint32_t N = 64u * SPL;
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    __shared__ __attribute__((aligned(16))) uint16_t T[4][256];  // slicing-by-4 tables
    __shared__ uint32_t part[NT / 64];
    __shared__ uint8_t hdr[16];
    const uint32_t frame = p.f0 + blockIdx.x, tid = threadIdx.x;
    const uint32_t ch = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *fb = reinterpret_cast<uint32_t *>(lds);
    const uint64_t fn = q.first_frame_number + frame;
    const HeaderCodes hc = header_codes(N, q.sample_rate, fn);
    const uint64_t begin = q.frame_off[frame];
    const uint32_t flen = (uint32_t)(q.frame_off[frame + 1] - begin);  // bytes, CRC-16 included
    const uint32_t nwords = (flen + 3) / 4 + 1;
    for (uint32_t i = tid; i < (nwords + 3) / 4; i += NT) reinterpret_cast<uint4 *>(fb)[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = tid; i < 128; i += NT)
        reinterpret_cast<uint4 *>(&T[0][0])[i] = reinterpret_cast<const uint4 *>(&kCrcT.t[0][0])[i];
    __syncthreads();
    if (tid == 0) {  // FrameHeader::build, stream.rs:242-276 (+ CRC-8, :194-197)
        const flacgpu_frame_plan fp = p.frame_plan[frame];
        uint32_t kk = 0;
        hdr[kk++] = 0xFF;
        hdr[kk++] = 0xF8;  // sync 0b111111111111100 + blocking strategy 0
        hdr[kk++] = (uint8_t)((hc.bcode << 4) | hc.rcode);
        const uint32_t acode = fp.assignment == FLACGPU_ASSIGN_INDEPENDENT ? p.channels - 1 : fp.assignment;
        const uint32_t pcode = p.bps == 8 ? 1 : p.bps == 12 ? 2 : p.bps == 16 ? 4 : p.bps == 20 ? 5
                             : p.bps == 24 ? 6 : p.bps == 32 ? 7 : 0;
        hdr[kk++] = (uint8_t)((acode << 4) | (pcode << 1));
        if (hc.fn_bytes == 1) {
            hdr[kk++] = (uint8_t)fn;
        } else {  // UTF-8-like frame number, stream.rs:1264-1325
            const uint32_t nb = hc.fn_bytes;
            const uint32_t lead = (0xFFu << (8 - nb)) & 0xFF;
            hdr[kk++] = (uint8_t)(lead | (uint32_t)(fn >> (6 * (nb - 1))));
            for (int b = (int)nb - 2; b >= 0; b--) hdr[kk++] = (uint8_t)(0x80 | ((fn >> (6 * b)) & 0x3F));
        }
        if (hc.bextra_bits == 8) hdr[kk++] = (uint8_t)(N - 1);
        else if (hc.bextra_bits == 16) { hdr[kk++] = (uint8_t)((FN - 1) >> 8); hdr[kk++] = (uint8_t)(N - 1); }
        if (hc.rextra_bits == 8) hdr[kk++] = (uint8_t)hc.rextra;
        else if (hc.rextra_bits == 16) { hdr[kk++] = (uint8_t)(hc.rextra >> 8); hdr[kk++] = (uint8_t)hc.rextra; }
        uint32_t crc = 0;  // CRC-8, poly 0x07 (crc.rs:99-128)
        for (uint32_t i = 0; i < kk; i++) {
            crc ^= hdr[i];
            for (int b = 0; b < 8; b++) crc = (crc & 0x80) ? ((crc << 1) ^ 0x07) & 0xFF : (crc << 1) & 0xFF;
        }
        hdr[kk++] = (uint8_t)crc;
        for (uint32_t i = 0; i < kk; i++) lds_put(fb, 8 * i, hdr[i], 8);
    }
    uint32_t start_bit = header_bytes(hc) * 8;
    for (uint32_t c = 0; c < ch; c++) start_bit += p.out_plan[(size_t)frame * p.channels + c].bits;
    if (!(p.dbg & 2)) wave_subframe<SPL, MAXO>(p, frame, ch, fb, start_bit);
    __syncthreads();
    // ---- CRC-16 of bytes [0, len) (crc.rs:142-188); byte i = fb[i / 4] >> (24 - 8 (i % 4)).
    // One pass: the frame, left-padded with zero bytes (they leave a zero CRC state unchanged) to
    // S * NT slices of 17 words, slice g = s * NT + tid; a lane runs its S <= 3 table-lookup chains
    // interleaved (each step is one LDS round trip), folds them with Horner in x^(544 NT) and the
    // lanes' results are combined with the weights x^(544 (NT - 1 - tid)) (GF(2) linearity).
    const uint32_t len = (p.dbg & 4) ? 0 : flen - 2;
    constexpr uint32_t CH = NT * 68;                 // bytes per slice row
    const uint32_t S = (len + CH - 1) / CH;          // 1..3 (frame_fb_words bounds the frame)
    const uint32_t padb = S * CH - len;
    uint32_t running = 0;
    auto chains = [&](auto sc) {
        constexpr int SC = decltype(sc)::value;
        uint32_t crc[SC], cur[SC], sh[SC];
        int32_t w0[SC];
#pragma unroll
        for (int c = 0; c < SC; c++) {
            const int32_t f0 = (int32_t)((c * NT + tid) * 68) - (int32_t)padb;  // first frame byte of the slice
            w0[c] = f0 >> 2;                         // floor: f0 may be negative
            sh[c] = ((uint32_t)f0 & 3u) * 8u;
            crc[c] = 0;
            cur[c] = w0[c] >= 0 ? fb[w0[c]] : 0u;
        }
#pragma unroll
        for (int kk = 0; kk < 17; kk++) {
#pragma unroll
            for (int c = 0; c < SC; c++) {
                const int32_t wn = w0[c] + kk + 1;
                const uint32_t nxt = wn >= 0 ? fb[wn] : 0u;
                const uint32_t m = sh[c] ? (cur[c] << sh[c]) | (nxt >> (32 - sh[c])) : cur[c];
                crc[c] = T[3][((crc[c] >> 8) ^ (m >> 24)) & 0xFF] ^ T[2][(crc[c] ^ (m >> 16)) & 0xFF] ^
                         T[1][(m >> 8) & 0xFF] ^ T[0][m & 0xFF];
                cur[c] = nxt;
            }
        }
        const uint32_t xrow = kCrcW17.w[NT];          // x^(544 NT)
        uint32_t acc = crc[0];
#pragma unroll
        for (int c = 1; c < SC; c++) acc = gf_mulmod(acc, xrow) ^ crc[c];
        return acc;
    };
    if (S) {
        uint32_t c = S == 1 ? chains(std::integral_constant<int, 1>{})
                   : S == 2 ? chains(std::integral_constant<int, 2>{})
                            : chains(std::integral_constant<int, 3>{});
        c = gf_mulmod(c, kCrcW17.w[NT - 1 - tid]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c ^= __shfl_xor(c, off, 64);
        if ((tid & 63) == 0) part[tid >> 6] = c;
        __syncthreads();
#pragma unroll
        for (int wv = 0; wv < NT / 64; wv++) running ^= part[wv];
    }
    if (tid == 0) {  // the two CRC bytes follow byte len - 1 (still zero there)
        atomicOr(&fb[len >> 2], ((running >> 8) & 0xFF) << (24 - 8 * (len & 3)));
        atomicOr(&fb[(len + 1) >> 2], (running & 0xFF) << (24 - 8 * ((len + 1) & 3)));
    }
    __syncthreads();
    // ---- copy out: output dword j covers frame bytes [4 j - r, 4 j - r + 4)
    uint8_t *ob = reinterpret_cast<uint8_t *>(q.out_words);
    const uint32_t r = (uint32_t)(begin & 3);
    uint32_t *og = reinterpret_cast<uint32_t *>(ob + (begin - r));
    const uint32_t nout = (p.dbg & 8) ? 0 : (r + flen + 3) / 4;
    auto emit = [&](uint32_t j, uint32_t hi, uint32_t lo) {
        const uint32_t m = r ? (hi << (8 * (4 - r))) | (lo >> (8 * r)) : hi;  // MSB-first window
        const int32_t fbyte = (int32_t)(4 * j) - (int32_t)r;
        if (fbyte >= 0 && fbyte + 4 <= (int32_t)flen) {
            og[j] = __builtin_bswap32(m);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int32_t fbk = fbyte + e;
                if (fbk >= 0 && fbk < (int32_t)flen) ob[begin + fbk] = (uint8_t)(m >> (24 - 8 * e));
            }
        }
    };
    const uint32_t back = r ? 1u : 0u;
    for (uint32_t j0 = tid; j0 < nout; j0 += 4 * NT) {  // four independent LDS round trips in flight
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + u * NT;
            const bool in = j < nout;
            hi[u] = (in && !(r && j == 0)) ? fb[j - back] : 0u;
            lo[u] = in ? fb[j] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (j0 + u * NT < nout) emit(j0 + u * NT, hi[u], lo[u]);
    }
}

// VERIFY = false: store the CRC-16 behind the frame; true: compare with the stored one and
// count mismatching frames in verify_counts[1]
template <bool VERIFY>
__global__ void __launch_bounds__(WG) k_crc(Params p, PackParams q, uint32_t *verify_counts) {
    __shared__ uint16_t T[4][256];                 // slicing-by-4 tables
    __shared__ uint32_t buf[CRC_CHUNK / 4 + WG];   // one pad dword per 64-byte slice
    __shared__ uint32_t part[4];
    const uint32_t frame = p.f0 + blockIdx.x, tid = threadIdx.x;
    {
        uint32_t c = tid << 8;
        for (int k = 0; k < 4; k++) {  // T[k][b] = CRC state after byte b followed by k zero bytes
            for (int b = 0; b < 8; b++) c = (c & 0x8000) ? ((c << 1) ^ 0x8005) & 0xFFFF : (c << 1) & 0xFFFF;
            T[k][tid] = (uint16_t)c;
        }
    }
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(q.out_words);
    const uint64_t begin = q.frame_off[frame];
    const uint32_t len = (uint32_t)(q.frame_off[frame + 1] - begin) - 2;  // all but the CRC itself
    const uint32_t my_weight = kCrcW.w[WG - 1 - tid];   // slices to the right of mine
    const uint32_t xchunk = kCrcW.w[WG];                // x^(8 * CRC_CHUNK)
    uint32_t running = 0;
    uint32_t pos = 0;
    while (pos < len) {
        // the FIRST pass takes the odd-sized head so every later pass is a full chunk; a chunk
        // is right-aligned in the LDS window, i.e. left-padded with zero bytes, which leave a
        // zero CRC state unchanged -- all 256 slices then have the same length
        const uint32_t clen = (pos == 0 && (len % CRC_CHUNK)) ? len % CRC_CHUNK : CRC_CHUNK;
        const uint32_t padb = CRC_CHUNK - clen;
        __syncthreads();
        for (uint32_t d = tid; d < CRC_CHUNK / 4; d += WG) {
            uint32_t v = 0;
            const int64_t j0 = (int64_t)4 * d - padb;  // chunk byte index of this dword's first byte
            if (j0 + 3 >= 0) {
                const uint64_t src = begin + pos;
                if (j0 >= 0 && ((src + j0) & 3) == 0) {
                    v = *reinterpret_cast<const uint32_t *>(bytes + src + j0);
                } else {
                    for (int e = 0; e < 4; e++)
                        if (j0 + e >= 0) v |= (uint32_t)bytes[src + j0 + e] << (8 * e);
                }
            }
            buf[d + (d >> 4)] = v;
        }
        __syncthreads();
        uint32_t crc = 0;
        const uint32_t *sl = buf + tid * 17;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t v = sl[w];
            crc = T[3][((crc >> 8) ^ v) & 0xFF] ^ T[2][(crc ^ (v >> 8)) & 0xFF] ^
                  T[1][(v >> 16) & 0xFF] ^ T[0][v >> 24];
        }
        // CRC (init 0) is GF(2)-linear: crc(chunk) = XOR_t crc(slice_t) * x^(512 * slices after t)
        uint32_t c = gf_mulmod(crc, my_weight);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c ^= __shfl_xor(c, off, 64);
        if ((tid & 63) == 0) part[tid >> 6] = c;
        __syncthreads();
        running = gf_mulmod(running, xchunk) ^ part[0] ^ part[1] ^ part[2] ^ part[3];
        pos += clen;
    }
    if (tid == 0) {
        uint8_t *ob = reinterpret_cast<uint8_t *>(q.out_words);
        if constexpr (VERIFY) {
            const uint32_t stored = ((uint32_t)ob[begin + len] << 8) | ob[begin + len + 1];
            if (stored != running) atomicAdd(&verify_counts[1], 1u);
        } else {
            ob[begin + len] = (uint8_t)(running >> 8);
            ob[begin + len + 1] = (uint8_t)running;
        }
    }
}


// ---------------------------------------------------------------------------------
// Frame decoder + verifier: the reference's read_frame / read_subframes /
// read_subframe / read_residuals / predict (decode.rs:1388-1436, 1494-1856) for frames whose
// byte offsets are known (everything this encoder produces).  Rice decoding and LPC synthesis
// are sequential inside a subframe, so ONE LANE decodes ONE FRAME: the parallelism is across
// the thousands of frames of a batch.  Output: planar PCM [frame][channel][ldb].
// verify_counts[0] frames with a header / structure error, [2] frames whose PCM differs from the
// reference buffer (when given), [3] differing samples.
// ---------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *base;
    uint64_t pos;  // absolute bit position
    __device__ __forceinline__ uint32_t peek32() const {  // next 32 bits, MSB first
        const uint8_t *b = base + (pos >> 3);
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) v = (v << 8) | b[i];
        return (uint32_t)(v >> (8 - (pos & 7)));
    }
    __device__ __forceinline__ uint32_t get(uint32_t n) {  // n in 0..32
        if (n == 0) return 0;
        const uint32_t v = peek32() >> (32 - n);
        pos += n;
        return v;
    }
    __device__ __forceinline__ int32_t get_signed(uint32_t n) {
        const uint32_t v = get(n);
        return n < 32 ? (int32_t)(v << (32 - n)) >> (32 - n) : (int32_t)v;
    }
    __device__ __forceinline__ uint32_t unary1() {  // zeros before the next 1 bit
        uint32_t q = 0;
        for (;;) {
            const uint32_t w = peek32();
            if (w) {
                const uint32_t z = (uint32_t)__builtin_clz(w);
                pos += z + 1;
                return q + z;
            }
            q += 32;
            pos += 32;
            if (q > (1u << 24)) return q;  // corrupt stream guard
        }
    }
};

__global__ void __launch_bounds__(64) k_decode(Params p, PackParams q, int32_t *__restrict__ out,
                                               const int32_t *__restrict__ expect,
                                               uint32_t *verify_counts) {
    const uint32_t frame = blockIdx.x * 64 + threadIdx.x;
    if (frame >= p.n_frames) return;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(q.out_words);
    BitReader r{bytes, q.frame_off[frame] * 8};
    const uint64_t end_bit = q.frame_off[frame + 1] * 8;
    bool bad = false;
    // FrameHeader::parse, stream.rs:214-240
    if (r.get(15) != 0x7FFC) bad = true;
    r.get(1);
    const uint32_t bcode = r.get(4), rcode = r.get(4), acode = r.get(4), pcode = r.get(3);
    r.get(1);
    {   // frame number (UTF-8 like), stream.rs:1244-1262
        uint32_t ones = 0;
        while (ones < 8 && r.get(1)) ones++;
        if (ones == 0) r.get(7);
        else if (ones == 1 || ones > 7) bad = true;
        else {
            r.get(7 - ones);
            for (uint32_t i = 1; i < ones; i++) {
                if (r.get(2) != 2) bad = true;
                r.get(6);
            }
        }
    }
    uint32_t n;
    switch (bcode) {
    case 1: n = 192; break;
    case 2: n = 576; break;
    case 3: n = 1152; break;
    case 4: n = 2304; break;
    case 5: n = 4608; break;
    case 6: n = r.get(8) + 1; break;
    case 7: n = r.get(16) + 1; break;
    case 0: n = 0; bad = true; break;
    default: n = 256u << (bcode - 8); break;
    }
    if (rcode == 12) r.get(8);
    else if (rcode == 13 || rcode == 14) r.get(16);
    else if (rcode == 15) bad = true;
    {   // CRC-8 over the header
        const uint64_t hb0 = q.frame_off[frame], hb1 = r.pos >> 3;
        uint32_t crc = 0;
        for (uint64_t i = hb0; i < hb1; i++) {
            crc ^= bytes[i];
            for (int b = 0; b < 8; b++) crc = (crc & 0x80) ? ((crc << 1) ^ 0x07) & 0xFF : (crc << 1) & 0xFF;
        }
        if (r.get(8) != crc) bad = true;
    }
    const uint32_t bps_tab[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    const uint32_t bps = pcode ? bps_tab[pcode] : p.bps;
    const uint32_t nch = acode < 8 ? acode + 1 : 2;
    if (n != frame_len(p, frame) || bps != p.bps || nch != p.channels || acode > 10) bad = true;
    int32_t *rows = out + (size_t)frame * p.channels * p.ldb;
    for (uint32_t c = 0; c < nch && !bad; c++) {
        int32_t *x = rows + (size_t)c * p.ldb;
        uint32_t sbps = bps;
        if ((acode == 8 && c == 1) || (acode == 9 && c == 0) || (acode == 10 && c == 1)) sbps++;
        // SubframeHeader, stream.rs:1375-1388
        if (r.get(1)) bad = true;
        const uint32_t type = r.get(6);
        uint32_t wasted = 0;
        if (r.get(1)) wasted = r.unary1() + 1;
        if (wasted >= sbps) { bad = true; break; }
        const uint32_t eb = sbps - wasted;
        uint32_t order = 0;
        int32_t coef[32];
        uint32_t shift = 0;
        bool has_res = false;
        if (type == 0) {  // CONSTANT
            const int32_t v = r.get_signed(eb);
            for (uint32_t i = 0; i < n; i++) x[i] = v;
        } else if (type == 1) {  // VERBATIM
            for (uint32_t i = 0; i < n; i++) x[i] = r.get_signed(eb);
        } else if (type >= 8 && type <= 12) {  // FIXED, decode.rs:1683-1702
            order = type - 8;
            const int32_t fc[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
            for (uint32_t j = 0; j < 4; j++) coef[j] = fc[order][j];
            has_res = true;
        } else if (type >= 32) {  // LPC, decode.rs:1704-1736
            order = type - 31;
            has_res = true;
        } else {
            bad = true;
            break;
        }
        if (has_res) {
            if (order > n) { bad = true; break; }
            for (uint32_t i = 0; i < order; i++) x[i] = r.get_signed(eb);
            if (type >= 32) {
                const uint32_t prec = r.get(4) + 1;
                if (prec == 16) bad = true;
                const int32_t sh = r.get_signed(5);
                if (sh < 0) bad = true;
                shift = (uint32_t)sh;
                for (uint32_t j = 0; j < order; j++) coef[j] = r.get_signed(prec);
            }
            // read_residuals (decode.rs:1800-1856) fused with predict (decode.rs:1738-1752)
            const uint32_t method = r.get(2);
            if (method > 1) { bad = true; break; }
            const uint32_t hb = method ? 5u : 4u, esc = method ? 31u : 15u;
            const uint32_t po = r.get(4);
            const uint32_t plen = n >> po;
            uint32_t i = order;
            for (uint32_t part = 0; part < (1u << po) && !bad; part++) {
                uint32_t cnt = plen;
                if (part == 0) {
                    if (plen < order) { bad = true; break; }
                    cnt = plen - order;
                }
                const uint32_t k = r.get(hb);
                const uint32_t ebits = k == esc ? r.get(5) : 0;
                for (uint32_t t = 0; t < cnt; t++, i++) {
                    int32_t res;
                    if (k == esc) {
                        res = ebits ? r.get_signed(ebits) : 0;
                    } else {
                        const uint32_t qn = r.unary1();
                        const uint32_t u = (qn << k) | r.get(k);
                        res = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
                    }
                    long long s = 0;
                    for (uint32_t j = 0; j < order; j++) s += (long long)x[i - 1 - j] * (long long)coef[j];
                    x[i] = res + (int32_t)(s >> shift);
                }
                if (r.pos > end_bit) bad = true;
            }
            if (i != n) bad = true;
        }
        if (wasted)
            for (uint32_t i = 0; i < n; i++) x[i] = (int32_t)((uint32_t)x[i] << wasted);
    }
    if (!bad && acode >= 8) {  // undo the stereo decorrelation, decode.rs:1520-1628
        int32_t *c0 = rows, *c1 = rows + p.ldb;
        for (uint32_t i = 0; i < n; i++) {
            const long long a = c0[i], b = c1[i];
            if (acode == 8) c1[i] = (int32_t)(a - b);
            else if (acode == 9) c0[i] = (int32_t)(a + b);
            else {
                const long long side = b;
                const long long sum = a * 2 + ((side < 0 ? -side : side) & 1);
                c0[i] = (int32_t)((sum + side) >> 1);
                c1[i] = (int32_t)((sum - side) >> 1);
            }
        }
    }
    if (!bad) {  // byte alignment + CRC-16 must end the frame exactly
        r.pos = (r.pos + 7) & ~7ull;
        if (r.pos + 16 != end_bit) bad = true;
    }
    if (bad) {
        atomicAdd(&verify_counts[0], 1u);
        return;
    }
    if (expect) {
        const int32_t *e = expect + (size_t)frame * p.channels * p.ldb;
        uint32_t diff = 0;
        for (uint32_t c = 0; c < nch; c++)
            for (uint32_t i = 0; i < n; i++) diff += rows[(size_t)c * p.ldb + i] != e[(size_t)c * p.ldb + i];
        if (diff) {
            atomicAdd(&verify_counts[2], 1u);
            atomicAdd(&verify_counts[3], diff);
        }
    }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
const char *const kKernelNames[FLACGPU_N_KERNELS] = {
    "k_deinterleave", "k_stereo_stats", "k_fixed", "k_autocorr", "k_lpc",
    "k_fir",          "k_decide",       "k_emit",  "k_layout",   "k_pack",
    "k_crc",          "k_cand64"};

}  // namespace

struct flacgpu_ctx {
    flacgpu_options opts;
    uint32_t bps, channels, max_frames, ldb, ncand, stereo4;
    int device;
    // device buffers
    int32_t *d_in = nullptr, *d_planar = nullptr, *d_resid = nullptr;
    double *d_window_full = nullptr, *d_window_last = nullptr, *d_log2_thr = nullptr, *d_ac = nullptr;
    CandInfo *d_cinfo = nullptr;
    SubPlan *d_fixed = nullptr, *d_cand = nullptr, *d_out = nullptr;
    LpcParams *d_lpc = nullptr;
    FrameInfo *d_finfo = nullptr;
    flacgpu_frame_plan *d_fplan = nullptr;
    uint32_t *d_stats = nullptr;
    uint32_t *d_orbits = nullptr;   // OR of all samples per (frame, candidate)
    int32_t *d_decoded = nullptr;   // [F][C][ldb] PCM decoded back from the packed frames (lazy)
    uint32_t *d_verify = nullptr;   // [4] verify counters
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_layout = nullptr;
    bool two_ranges = false;         // FLACGPU_TUNE_TWO_RANGES
    int lag_split = 4;               // FLACGPU_TUNE_LAG_SPLIT
    uint32_t *d_packed = nullptr;   // packed frame bytes (as 32-bit words)
    uint64_t *d_frame_off = nullptr;
    uint64_t packed_cap = 0;        // bytes
    bool packed_valid = false;
    bool resid_valid = false;       // d_resid holds the rows of the last analysed batch
    Params last_params;
    uint32_t window_last_len = 0;
    hipStream_t own_stream = nullptr;
    // last call
    uint32_t last_frames = 0, last_len = 0;
    bool timing = false;
    hipEvent_t ev[FLACGPU_N_KERNELS + 1];
    bool ev_ok = false;
    bool ev_used[FLACGPU_N_KERNELS];
    float last_ms[FLACGPU_N_KERNELS];
};

namespace {

// Window::generate (encode.rs:1725-1783) on the host: it needs cos(), whose last-ulp
// behaviour differs between libms; the table is built once per block length with the
// host libm (the one a Rust build on this box links) and uploaded.
void window_generate(int kind, float p, uint32_t n, std::vector<double> &w) {
    const double PI = 3.14159265358979323846264338327950288;
    w.assign(n, 1.0);
    if (kind == FLACGPU_WINDOW_RECTANGLE) return;
    auto hann = [&]() {
        double np = (double)(uint16_t)n - 1.0;
        for (uint32_t i = 0; i < n; i++) w[i] = 0.5 - 0.5 * cos(2.0 * PI * (double)i / np);
    };
    if (kind == FLACGPU_WINDOW_HANN) {
        hann();
        return;
    }
    if (p != p) p = 0.5f;  // NaN => Tukey(0.5)
    if (p <= 0.0f) return;
    if (p >= 1.0f) {
        hann();
        return;
    }
    double t = (double)p / 2.0 * (double)n;
    uint64_t tu = (uint64_t)t;
    if (tu == 0) return;
    uint64_t np = tu - 1;
    if (np > n || n - np < np) return;
    double npf = (double)(uint16_t)np;
    for (uint32_t i = 0; i < (uint32_t)np; i++) {
        double x = 0.5 - 0.5 * cos(PI * (double)i / npf);
        w[i] = x;
        w[n - 1 - i] = x;
    }
}

// thresholds for floor(log2(l)) (encode.rs:3360): for exponent e, the smallest double in
// [2^e, 2^(e+1)) whose host log2() already rounds up to e+1 (or 2^(e+1) if none)
void build_log2_thresholds(double *thr) {
    for (int e = -64; e < 64; e++) {
        double hi = ldexp(1.0, e + 1);
        double top = nextafter(hi, 0.0);
        if (floor(log2(top)) == (double)e) {
            thr[e + 64] = hi;
            continue;
        }
        uint64_t lo_b, hi_b;
        double lo = ldexp(1.0, e);
        memcpy(&lo_b, &lo, 8);
        memcpy(&hi_b, &top, 8);
        while (lo_b < hi_b) {  // first bit pattern with floor(log2) == e+1
            uint64_t mid = lo_b + (hi_b - lo_b) / 2;
            double m;
            memcpy(&m, &mid, 8);
            if (floor(log2(m)) == (double)e) lo_b = mid + 1;
            else hi_b = mid;
        }
        memcpy(&thr[e + 64], &lo_b, 8);
    }
}

int upload_window(flacgpu_ctx *c, uint32_t n, double *dst, hipStream_t st) {
    std::vector<double> w;
    window_generate(c->opts.window_kind, c->opts.window_param, n, w);
    w.resize((size_t)n + 64, 0.0);
    HIP_TRY(hipMemcpyAsync(dst, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // `w` is a local
    return 0;
}

// dynamic LDS of k_pack: the residual row + the subframe's bit string (a chosen subframe is
// never longer than its VERBATIM form: <= 40 + 33 n bits, plus the 16-byte frame header)
size_t pack_lds_bytes(uint32_t block_size) { return (size_t)pack_sb_words(block_size) * 4; }

// K0 for frames [f0, f0 + fcount): split the channels into planar rows and OR every candidate's
// samples.  Returns true when the orbits are already accumulated.
bool launch_k0(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames, uint32_t last_len,
               uint32_t f0, uint32_t fcount, hipStream_t st) {
    const uint32_t B = c->opts.block_size;
    const dim3 grid(std::max<uint32_t>(1u, (B + WG * 8 - 1) / (WG * 8)), fcount);  // 8 samples per lane
    if (layout == FLACGPU_LAYOUT_INTERLEAVED && c->channels == 2) {
        hipLaunchKernelGGL(k_deinterleave2, grid, dim3(WG), 0, st, (const int2 *)d_pcm, c->d_planar, B, c->ldb,
                           n_frames, last_len, c->d_orbits, c->ncand, f0);
        return true;
    }
    if (layout == FLACGPU_LAYOUT_INTERLEAVED && c->ncand == c->channels) {
        switch (c->channels) {
#define X(C) case C: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deinterleave_n<C>), grid, dim3(WG), 0, st, d_pcm, \
                                        c->d_planar, B, c->ldb, n_frames, last_len, c->d_orbits, f0); return true;
            X(1) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
        default: break;
        }
    }
    hipLaunchKernelGGL(k_deinterleave, grid, dim3(WG), 0, st, d_pcm, c->d_planar, c->channels, B, c->ldb,
                       n_frames, last_len, layout == FLACGPU_LAYOUT_PLANAR, f0);
    return false;
}

void launch_lpc(const Params &p, uint32_t blocks, hipStream_t st) {
    if (p.max_lpc_order <= 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<8>), dim3(blocks), dim3(64), 0, st, p);
    else if (p.max_lpc_order <= 12) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<12>), dim3(blocks), dim3(64), 0, st, p);
    else if (p.max_lpc_order <= 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<16>), dim3(blocks), dim3(64), 0, st, p);
    else if (getenv("FLACGPU_LPC_DYN")) hipLaunchKernelGGL(k_lpc, dim3(blocks), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lpc_u<32>), dim3(blocks), dim3(64), 0, st, p);
}

// block lengths the wave kernels are instantiated for: 64 lanes x SPL samples
#define FLACGPU_WAVE_SIZES(X) X(4096, 64) X(2304, 36) X(2048, 32) X(1152, 18) X(1024, 16)
bool wave_block_size(uint32_t B) {
    switch (B) {
#define X(n, spl) case n:
        FLACGPU_WAVE_SIZES(X)
#undef X
        return true;
    default: return false;
    }
}
void launch_cand64(const Params &p, uint32_t B, uint32_t blocks, hipStream_t st) {
    if (p.max_lpc_order > 16) {  // orders 17..32: 4096-sample blocks only
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<64, 32>), dim3(blocks), dim3(WG), 0, st, p);
        return;
    }
    switch (B) {
#define X(n, spl) case n: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cand64<spl, 16>), dim3(blocks), dim3(WG), 0, st, p); break;
        FLACGPU_WAVE_SIZES(X)
#undef X
    default: break;
    }
}
template <int NT, int SPL, int MAXO = 16>
void launch_frame64_nt(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st) {
    static bool big_lds = false;  // frames of 5..8 channels need more than the default 64 KB
    if (lds > 64 * 1024 && !big_lds) {
        (void)hipFuncSetAttribute((const void *)k_frame64<NT, SPL, MAXO>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        big_lds = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frame64<NT, SPL, MAXO>), dim3(frames), dim3(NT), lds, st, p, q);
}
template <int SPL>
void launch_frame64_spl(const Params &p, const PackParams &q, uint32_t frames, size_t lds, hipStream_t st) {
    switch (p.channels) {
    case 1: launch_frame64_nt<64, SPL>(p, q, frames, lds, st); break;
    case 2: launch_frame64_nt<128, SPL>(p, q, frames, lds, st); break;
    case 3: launch_frame64_nt<192, SPL>(p, q, frames, lds, st); break;
    case 4: launch_frame64_nt<256, SPL>(p, q, frames, lds, st); break;
    default:
        if constexpr (SPL == 64) {  // 5..8 channels: 4096-sample frames only
            switch (p.channels) {
            case 5: launch_frame64_nt<320, SPL>(p, q, frames, lds, st); break;
            case 6: launch_frame64_nt<384, SPL>(p, q, frames, lds, st); break;
            case 7: launch_frame64_nt<448, SPL>(p, q, frames, lds, st); break;
            default: launch_frame64_nt<512, SPL>(p, q, frames, lds, st); break;
            }
        }
        break;
    }
}
void launch_frame64(const Params &p, const PackParams &q, uint32_t B, uint32_t frames, size_t lds, hipStream_t st) {
    if (p.max_lpc_order > 16) {  // orders 17..32: 4096-sample blocks, <= 4 channels
        switch (p.channels) {
        case 1: launch_frame64_nt<64, 64, 32>(p, q, frames, lds, st); break;
        case 2: launch_frame64_nt<128, 64, 32>(p, q, frames, lds, st); break;
        case 3: launch_frame64_nt<192, 64, 32>(p, q, frames, lds, st); break;
        default: launch_frame64_nt<256, 64, 32>(p, q, frames, lds, st); break;
        }
        return;
    }
    switch (B) {
#define X(n, spl) case n: launch_frame64_spl<spl>(p, q, frames, lds, st); break;
        FLACGPU_WAVE_SIZES(X)
#undef X
    default: break;
    }
}

template <int NL, bool STEREO>
void launch_autocorr3(const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n,
                      const double *win, hipStream_t st) {
    const uint32_t groups = (nframes * p.ncand + 63) / 64;
    // 4 waves per 64 candidates (lags split 4 ways) by default: the f64 stream needs two waves per
    // SIMD to issue at full rate and 8192 frames are only 512 candidate groups (0.23 ms against
    // 0.31 ms split 2 ways).  When other contexts keep the SIMDs busy anyway, the 2-way split wins:
    // the int -> f64 x window conversion is replicated 2x instead of 4x (71 M instead of 92 M
    // instructions).
    if (p.ac_split == 2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3<NL, 2, STEREO>), dim3(groups), dim3(128), 0, st, p,
                           frame0, nframes, n, win);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3<NL, 4, STEREO>), dim3(groups), dim3(256), 0, st, p,
                           frame0, nframes, n, win);
}
template <bool STEREO>
void launch_autocorr3_nl(const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n, const double *win,
                         hipStream_t st) {
    const uint32_t nl = p.max_lpc_order + 1;
    if (nl <= 5) launch_autocorr3<5, STEREO>(p, frame0, nframes, n, win, st);
    else if (nl <= 9) launch_autocorr3<9, STEREO>(p, frame0, nframes, n, win, st);
    else if (nl <= 13) launch_autocorr3<13, STEREO>(p, frame0, nframes, n, win, st);
    else launch_autocorr3<17, STEREO>(p, frame0, nframes, n, win, st);
}
// frame length a multiple of 32, order <= 16, and either stereo L/R/M/S candidates of <= 24-bit
// samples (mid/side formed with one v_mad_i32_i24) or independent channels of any width
bool try_autocorr3(const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n, const double *win,
                   hipStream_t st) {
    if (n < 32 || n % 32 != 0 || getenv("FLACGPU_NO_AC3")) return false;
    const bool stereo = p.stereo4 && p.ncand == 4 && p.channels == 2 && p.bps <= 24;
    const bool indep = !p.stereo4 && p.ncand == p.channels;
    if (!stereo && !indep) return false;
    if (p.max_lpc_order > 16) {  // lags up to 32: two blocks of history, frame a multiple of 64
        if (n % 64 != 0) return false;
        const uint32_t groups = (nframes * p.ncand + 63) / 64;
        if (stereo)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3_deep<true>), dim3(groups), dim3(256), 0, st, p, frame0,
                               nframes, n, win);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr3_deep<false>), dim3(groups), dim3(256), 0, st, p, frame0,
                               nframes, n, win);
        return true;
    }
    if (stereo) launch_autocorr3_nl<true>(p, frame0, nframes, n, win, st);
    else launch_autocorr3_nl<false>(p, frame0, nframes, n, win, st);
    return true;
}

template <int H>
void launch_autocorr(const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n,
                     const double *win, hipStream_t st) {
    const uint32_t lanes = nframes * p.ncand;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_autocorr2<H, 4>), dim3((lanes + 63) / 64), dim3(WG), 0, st, p,
                       frame0, nframes, n, win);
}

void dispatch_autocorr(uint32_t H, const Params &p, uint32_t frame0, uint32_t nframes, uint32_t n,
                       const double *win, hipStream_t st) {
    if (try_autocorr3(p, frame0, nframes, n, win, st)) return;
    switch (H) {
    case 4: launch_autocorr<4>(p, frame0, nframes, n, win, st); break;
    case 8: launch_autocorr<8>(p, frame0, nframes, n, win, st); break;
    case 12: launch_autocorr<12>(p, frame0, nframes, n, win, st); break;
    case 16: launch_autocorr<16>(p, frame0, nframes, n, win, st); break;
    case 20: launch_autocorr<20>(p, frame0, nframes, n, win, st); break;
    case 24: launch_autocorr<24>(p, frame0, nframes, n, win, st); break;
    case 28: launch_autocorr<28>(p, frame0, nframes, n, win, st); break;
    case 32: launch_autocorr<32>(p, frame0, nframes, n, win, st); break;
    default: launch_autocorr<36>(p, frame0, nframes, n, win, st); break;
    }
}

}  // namespace

extern "C" {

const char *flacgpu_last_error(void) { return g_last_error.c_str(); }
const char *flacgpu_kernel_name(int i) {
    return (i >= 0 && i < FLACGPU_N_KERNELS) ? kKernelNames[i] : "";
}

int flacgpu_create(const flacgpu_options *o, uint32_t bps, uint32_t channels, int device,
                   uint32_t max_frames, flacgpu_ctx **out) {
    if (!o || !out) return FLACGPU_ERR_INVALID_ARG;
    *out = nullptr;
    // Options validation, encode.rs:1418-1455; stream validation, :495, :1904
    if (o->block_size < 16 || o->block_size > 65535 || o->max_lpc_order > 32 ||
        o->max_partition_order > 15 || bps < 1 || bps > 32 || channels < 1 || channels > 8 ||
        max_frames == 0 || max_frames > 65535) {  // frames index gridDim.y of K0
        g_last_error = "invalid option / stream parameter";
        return FLACGPU_ERR_INVALID_ARG;
    }
    if (o->block_size > FLACGPU_MAX_BLOCK_SIZE) {
        g_last_error = "block_size > FLACGPU_MAX_BLOCK_SIZE is not supported by the LDS-resident kernels";
        return FLACGPU_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_last_error = "no HIP device";
        return FLACGPU_ERR_NO_DEVICE;
    }
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDevice(&device));
    flacgpu_ctx *c = new flacgpu_ctx();
    c->opts = *o;
    c->bps = bps;
    c->channels = channels;
    c->max_frames = max_frames;
    c->device = device;
    c->ldb = (o->block_size + 3u) & ~3u;
    c->stereo4 = (channels == 2 && bps < 32) ? 1u : 0u;  // side needs bps+1 <= 32 (encode.rs:2715)
    c->ncand = c->stereo4 ? 4u : channels;
    const size_t F = max_frames, B = o->block_size, C = channels, NC = c->ncand;
    const size_t slack = 64;
#define ALLOC(ptr, count) HIP_TRY(hipMalloc((void **)&(ptr), sizeof(*(ptr)) * (count)))
    ALLOC(c->d_in, F * B * C + slack);
    ALLOC(c->d_planar, F * C * c->ldb + slack);
    ALLOC(c->d_resid, F * C * B);
    ALLOC(c->d_window_full, B + slack);
    ALLOC(c->d_window_last, B + slack);
    ALLOC(c->d_log2_thr, 128);
    ALLOC(c->d_ac, F * NC * AC_LD);
    ALLOC(c->d_cinfo, F * NC);
    ALLOC(c->d_fixed, F * NC);
    ALLOC(c->d_cand, F * NC);
    ALLOC(c->d_out, F * C);
    ALLOC(c->d_lpc, F * NC);
    ALLOC(c->d_finfo, F);
    ALLOC(c->d_fplan, F);
    ALLOC(c->d_stats, 4);
    ALLOC(c->d_orbits, F * NC);
    // worst case: every subframe VERBATIM at 32 bits + headers
    c->packed_cap = (uint64_t)F * C * B * 4 + (uint64_t)F * (C * 8 + 64) + 256;
    ALLOC(c->d_packed, c->packed_cap / 4 + 8);
    ALLOC(c->d_frame_off, F + 1);
#undef ALLOC
    HIP_TRY(hipStreamCreate(&c->own_stream));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_layout, hipEventDisableTiming));
    HIP_TRY(hipMemsetAsync(c->d_planar, 0, sizeof(int32_t) * (F * C * c->ldb + slack), c->own_stream));
    HIP_TRY(hipMemsetAsync(c->d_cinfo, 0, sizeof(CandInfo) * F * NC, c->own_stream));
    double thr[128];
    build_log2_thresholds(thr);
    HIP_TRY(hipMemcpyAsync(c->d_log2_thr, thr, sizeof thr, hipMemcpyHostToDevice, c->own_stream));
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    if (int rc = upload_window(c, o->block_size, c->d_window_full, c->own_stream)) return rc;
    // k_fixed / k_fir use 2 * block_size * 4 bytes of dynamic LDS (up to 128 KiB of the 160)
    const int dyn = (int)((2 * B + B / 16 + 16) * sizeof(int32_t));
    HIP_TRY(hipFuncSetAttribute((const void *)k_fixed, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    HIP_TRY(hipFuncSetAttribute((const void *)k_fir, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    HIP_TRY(hipFuncSetAttribute((const void *)k_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(B * sizeof(int32_t))));
    HIP_TRY(hipFuncSetAttribute((const void *)k_pack, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(pack_lds_bytes((uint32_t)B))));
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    c->ev_ok = true;
    *out = c;
    return FLACGPU_OK;
}

void flacgpu_destroy(flacgpu_ctx *c) {
    if (!c) return;
    (void)hipFree(c->d_in); (void)hipFree(c->d_planar); (void)hipFree(c->d_resid); (void)hipFree(c->d_window_full);
    (void)hipFree(c->d_window_last); (void)hipFree(c->d_log2_thr); (void)hipFree(c->d_ac); (void)hipFree(c->d_cinfo);
    (void)hipFree(c->d_fixed); (void)hipFree(c->d_cand); (void)hipFree(c->d_out); (void)hipFree(c->d_lpc);
    (void)hipFree(c->d_finfo); (void)hipFree(c->d_fplan); (void)hipFree(c->d_stats);
    (void)hipFree(c->d_packed); (void)hipFree(c->d_frame_off); (void)hipFree(c->d_orbits);
    (void)hipFree(c->d_decoded); (void)hipFree(c->d_verify);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_layout) (void)hipEventDestroy(c->ev_layout);
    if (c->ev_ok) for (auto &e : c->ev) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int flacgpu_set_timing(flacgpu_ctx *c, int enable) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    c->timing = enable != 0;
    return 0;
}

static void fill_params(const flacgpu_ctx *c, uint32_t n_frames, uint32_t last_len, Params &p) {
    const uint32_t B = c->opts.block_size;
    memset(&p, 0, sizeof p);
    p.channels = c->channels;
    p.bps = c->bps;
    p.block_size = B;
    p.ldb = c->ldb;
    p.ncand = c->ncand;
    p.stereo4 = c->stereo4;
    p.mid_side = c->opts.mid_side;
    p.exhaustive = c->opts.exhaustive_channel_correlation;
    p.max_lpc_order = c->opts.max_lpc_order;
    p.max_po = c->opts.max_partition_order;
    p.use_rice2 = c->bps > 16;  // encode.rs:1965
    p.n_frames = n_frames;
    p.last_len = last_len;
    p.f0 = 0;
    p.fcount = n_frames;
    p.ac_split = (uint32_t)c->lag_split;
    { const char *e = getenv("FLACGPU_DEBUG"); p.dbg = e ? (uint32_t)atoi(e) : 0; }
    p.planar = c->d_planar;
    p.window_full = c->d_window_full;
    p.window_last = c->d_window_last;
    p.log2_thr = c->d_log2_thr;
    p.cinfo = c->d_cinfo;
    p.fixed_plan = c->d_fixed;
    p.cand_plan = c->d_cand;
    p.out_plan = c->d_out;
    p.lpc = c->d_lpc;
    p.ac = c->d_ac;
    p.finfo = c->d_finfo;
    p.frame_plan = c->d_fplan;
    p.residuals = c->d_resid;
    p.stats = c->d_stats;

}

int flacgpu_analyze_device(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames,
                           uint32_t last_len, void *stream) {
    if (!c || !d_pcm || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size || (layout != 0 && layout != 1)) {
        g_last_error = "invalid analyze arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    const uint32_t B = c->opts.block_size;
    // the reference collects partitions into ArrayVec<_, 64> and panics beyond (encode.rs:3880)
    for (uint32_t n : {n_frames > 1 ? B : last_len, last_len}) {
        uint32_t tz = (uint32_t)__builtin_ctz(n);
        if ((tz < c->opts.max_partition_order ? tz : c->opts.max_partition_order) > (uint32_t)MAXP) {
            g_last_error = "effective partition order > 6: the reference panics (MAX_PARTITIONS = 64)";
            return FLACGPU_ERR_UNSUPPORTED;
        }
    }
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    if (last_len != B && last_len != c->window_last_len) {
        if (int rc = upload_window(c, last_len, c->d_window_last, st)) return rc;
        c->window_last_len = last_len;
    }
    Params p;
    fill_params(c, n_frames, last_len, p);

    int evi = 0;
    for (auto &u : c->ev_used) u = false;
    auto mark = [&](int k) {  // event BEFORE kernel k; the next mark closes it
        if (c->timing) {
            (void)hipEventRecord(c->ev[evi], st);
            c->ev_used[k] = true;
            evi++;
        }
    };
    int order_of_marks[FLACGPU_N_KERNELS];
    int n_marks = 0;
    auto begin = [&](int k) {
        order_of_marks[n_marks++] = k;
        mark(k);
    };

    HIP_TRY(hipMemsetAsync(c->d_stats, 0, 4 * sizeof(uint32_t), st));
    const uint32_t ncb = n_frames * c->ncand;
    HIP_TRY(hipMemsetAsync(c->d_orbits, 0, sizeof(uint32_t) * ncb, st));
    // K0 (+ OR of every candidate's samples -> wasted bits)
    const bool planar_direct = (layout == FLACGPU_LAYOUT_PLANAR) && (B % 4 == 0) && last_len == B;
    begin(0);
    bool have_orbits = false;
    if (planar_direct) {
        p.planar = d_pcm;  // [frame][ch][B] with ldb == B
    } else {
        have_orbits = launch_k0(c, d_pcm, layout, n_frames, last_len, 0, n_frames, st);
    }
    if (!have_orbits)
        hipLaunchKernelGGL(k_orbits, dim3(std::max<uint32_t>(1u, (B + WG * 8 - 1) / (WG * 8)), n_frames), dim3(WG), 0, st, p,
                           c->d_orbits);
    const size_t dyn2 = (2 * (size_t)B + B / 16 + 16) * sizeof(int32_t);
    if (c->stereo4 && !p.exhaustive) {
        begin(1);
        hipLaunchKernelGGL(k_stereo_stats, dim3(n_frames), dim3(WG), 0, st, p);
    }
    hipLaunchKernelGGL(k_candinfo, dim3((ncb + WG - 1) / WG), dim3(WG), 0, st, p, c->d_orbits);
    // blocks of exactly 4096 samples take the register-resident kernels; anything else (other
    // block sizes, a short last frame, candidates wider than 25 bits) the generic LDS ones
    const bool narrow = (c->bps + (c->stereo4 ? 1u : 0u) <= 25u) && !getenv("FLACGPU_NO_FAST");
    // wave-per-candidate kernel (FIXED + LPC analysis of a candidate in one wave, after the LPC
    // parameters are known): block lengths 64 x {16, 18, 32, 36, 64}, LPC order <= 16
    const bool w64 = narrow && wave_block_size(B) && (p.max_lpc_order <= 16 || B == FN) && p.max_po <= 6 &&
                     !getenv("FLACGPU_NO_W64");
    const uint32_t n_fast = w64 ? ((last_len == B) ? n_frames : n_frames - 1) : 0;
    Params pf = p, pg = p;
    pf.f0 = 0;
    pf.fcount = n_fast;
    pg.f0 = n_fast;
    pg.fcount = n_frames - n_fast;
    const bool lpc = p.max_lpc_order > 0;
    // Otherwise the FIXED analysis and the autocorrelation -> Levinson chain, which only share
    // their input, run concurrently on two HIP streams (fork after k_candinfo, join before
    // k_fir).  With per-kernel timing enabled everything is serialised on one stream.
    const bool fork = lpc && !c->timing && !getenv("FLACGPU_NO_FORK") && !(w64 && pg.fcount == 0);
    hipStream_t sf = fork ? c->aux_stream : st;
    if (fork) {
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(sf, c->ev_fork, 0));
    }
    begin(2);
    if (pg.fcount) hipLaunchKernelGGL(k_fixed, dim3(pg.fcount * c->ncand), dim3(WG), dyn2, sf, pg);
    if (fork) HIP_TRY(hipEventRecord(c->ev_join, sf));
    if (lpc) {
        const uint32_t H = ((p.max_lpc_order + 1) + 3u) & ~3u;
        begin(3);
        const uint32_t full = (last_len == B) ? n_frames : n_frames - 1;
        if (full) dispatch_autocorr(H, p, 0, full, B, c->d_window_full, st);
        if (full != n_frames) dispatch_autocorr(H, p, full, 1, last_len, c->d_window_last, st);
        begin(4);
        launch_lpc(p, (ncb + 63) / 64, st);
        if (fork) HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
        begin(5);
        if (pg.fcount) hipLaunchKernelGGL(k_fir, dim3(pg.fcount * c->ncand), dim3(WG), dyn2, st, pg);
    }
    if (w64 && pf.fcount) {
        begin(11);
        launch_cand64(pf, B, (pf.fcount * c->ncand + 3) / 4, st);
    }
    begin(6);
    hipLaunchKernelGGL(k_decide, dim3(n_frames), dim3(64), 0, st, p);
    // the residual rows (k_emit) are produced lazily: flacgpu_fetch(residuals) / host packing
    // need them, the device-side packer recomputes residuals in registers instead
    c->resid_valid = false;
    if (c->timing) (void)hipEventRecord(c->ev[evi], st);
    HIP_TRY(hipGetLastError());
    c->last_frames = n_frames;
    c->last_len = last_len;
    c->last_params = p;
    c->packed_valid = false;
    if (c->timing) {
        HIP_TRY(hipStreamSynchronize(st));
        for (auto &m : c->last_ms) m = 0.f;
        for (int i = 0; i < n_marks; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]);
            c->last_ms[order_of_marks[i]] = ms;
        }
    }
    return FLACGPU_OK;
}

static int ensure_residual_rows(flacgpu_ctx *c) {
    if (c->resid_valid) return FLACGPU_OK;
    HIP_TRY(hipDeviceSynchronize());
    const Params &p = c->last_params;
    hipLaunchKernelGGL(k_emit, dim3(p.n_frames * p.channels), dim3(WG),
                       (size_t)p.block_size * sizeof(int32_t), c->own_stream, p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    c->resid_valid = true;
    return FLACGPU_OK;
}

int flacgpu_fetch(flacgpu_ctx *c, flacgpu_frame_plan *plans, flacgpu_subframe_plan *subs,
                  int32_t *residuals) {
    if (!c || c->last_frames == 0) return FLACGPU_ERR_INVALID_ARG;
    const size_t F = c->last_frames;
    hipStream_t st = c->own_stream;
    HIP_TRY(hipDeviceSynchronize());
    if (residuals)
        if (int rc = ensure_residual_rows(c)) return rc;
    if (plans) HIP_TRY(hipMemcpyAsync(plans, c->d_fplan, sizeof(*plans) * F, hipMemcpyDeviceToHost, st));
    if (subs)
        HIP_TRY(hipMemcpyAsync(subs, c->d_out, sizeof(*subs) * F * c->channels, hipMemcpyDeviceToHost, st));
    if (residuals)
        HIP_TRY(hipMemcpyAsync(residuals, c->d_resid,
                               sizeof(int32_t) * F * c->channels * c->opts.block_size,
                               hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FLACGPU_OK;
}

int flacgpu_analyze(flacgpu_ctx *c, const int32_t *pcm, int layout, uint32_t n_frames,
                    uint32_t last_len, flacgpu_frame_plan *plans, flacgpu_subframe_plan *subs,
                    int32_t *residuals) {
    if (!c || !pcm || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size) {
        g_last_error = "invalid analyze arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    const size_t B = c->opts.block_size, C = c->channels;
    const size_t count = ((size_t)(n_frames - 1) * B + last_len) * C;
    HIP_TRY(hipMemcpyAsync(c->d_in, pcm, count * sizeof(int32_t), hipMemcpyHostToDevice, c->own_stream));
    // planar host input with a short last frame is laid out [frame][ch][len]; handled by K0
    int rc = flacgpu_analyze_device(c, c->d_in, layout, n_frames, last_len, c->own_stream);
    if (rc) return rc;
    return flacgpu_fetch(c, plans, subs, residuals);
}


int flacgpu_pack_device(flacgpu_ctx *c, uint64_t first_frame_number, uint32_t sample_rate,
                        void *stream) {
    if (!c || c->last_frames == 0) {
        g_last_error = "flacgpu_pack_device: no analysed batch";
        return FLACGPU_ERR_INVALID_ARG;
    }
    hipStream_t st = stream ? (hipStream_t)stream : c->own_stream;
    const Params &p = c->last_params;
    PackParams q;
    q.first_frame_number = first_frame_number;
    q.sample_rate = sample_rate;
    q.out_words = c->d_packed;
    q.frame_off = c->d_frame_off;
    q.cap_bytes = c->packed_cap;
    hipEvent_t *ev = c->ev;  // reuse the event pool: [0..3]
    if (c->timing) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, st, p, q);
    // frames of a wave block length are assembled whole in LDS by k_frame64 (residuals recomputed
    // from the PCM, CRC-16 from LDS, one write of the finished bytes); any other frame goes
    // through k_emit (residual rows) -> k_pack (one workgroup per subframe, zero-filled output,
    // atomic OR at shared words) -> k_crc
    const uint32_t B = p.block_size;
    const bool narrow = (c->bps + (c->stereo4 ? 1u : 0u) <= 25u) && !getenv("FLACGPU_NO_FAST");
    const uint32_t fbw = frame_fb_words(p.channels, c->bps, B);
    // wave per subframe: block lengths 64 x {16, 18, 32, 36, 64}; orders 17..32 and 5..8 channels
    // for 4096-sample blocks only (and not both)
    const bool f64w = narrow && wave_block_size(B) &&
                      (p.max_lpc_order <= 16 || (B == FN && p.channels <= 4)) &&
                      (p.channels <= 4 || B == FN) && p.max_po <= 6 &&
                      (size_t)fbw * sizeof(int32_t) <= 150 * 1024 &&
                      !getenv("FLACGPU_NO_FUSED_PACK") && !getenv("FLACGPU_NO_FRAME64");
    const uint32_t n_fast = f64w ? (p.last_len == B ? p.n_frames : p.n_frames - 1) : 0;
    const bool fused = n_fast != 0;
    Params pf = p, pg = p;
    pf.f0 = 0;
    pf.fcount = n_fast;
    pg.f0 = n_fast;
    pg.fcount = p.n_frames - n_fast;
    if (pg.fcount) hipLaunchKernelGGL(k_zero, dim3(2048), dim3(WG), 0, st, q, p.n_frames);
    if (c->timing) (void)hipEventRecord(ev[1], st);
    if (pf.fcount) launch_frame64(pf, q, B, pf.fcount, (size_t)fbw * sizeof(int32_t), st);
    if (pg.fcount) {
        if (!c->resid_valid) {  // residual rows of these frames
            hipLaunchKernelGGL(k_emit, dim3(pg.fcount * p.channels), dim3(WG),
                               (size_t)p.block_size * sizeof(int32_t), st, pg);
        }
        hipLaunchKernelGGL(k_pack, dim3(pg.fcount * p.channels), dim3(WG), pack_lds_bytes(p.block_size), st, pg, q);
    }
    if (c->timing) (void)hipEventRecord(ev[2], st);
    {   // CRC-16 of the frames that did not take the fused kernel
        Params pc = p;
        pc.f0 = fused ? n_fast : 0;
        pc.fcount = p.n_frames - pc.f0;
        if (pc.fcount) hipLaunchKernelGGL(k_crc<false>, dim3(pc.fcount), dim3(WG), 0, st, pc, q, (uint32_t *)nullptr);
    }
    if (c->timing) (void)hipEventRecord(ev[3], st);
    HIP_TRY(hipGetLastError());
    c->packed_valid = true;
    if (c->timing) {
        HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < 3; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            c->last_ms[8 + i] = ms;
        }
    }
    return FLACGPU_OK;
}

int flacgpu_set_tuning(flacgpu_ctx *c, int key, int value) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    switch (key) {
    case FLACGPU_TUNE_TWO_RANGES: c->two_ranges = value != 0; return FLACGPU_OK;
    case FLACGPU_TUNE_LAG_SPLIT:
        if (value != 2 && value != 4) break;
        c->lag_split = value;
        return FLACGPU_OK;
    }
    g_last_error = "flacgpu_set_tuning: unknown key / value";
    return FLACGPU_ERR_INVALID_ARG;
}

// Analysis + frame assembly of one batch in one call.  With FLACGPU_TUNE_TWO_RANGES set, when
// every frame takes the wave kernels (4096-sample blocks, order <= 16, <= 4 channels), the batch
// is cut into two frame ranges that run
// the whole kernel chain on two HIP streams: the HBM-bound (K0, copy-out), latency-bound (Levinson,
// layout, CRC) and VALU-bound (autocorrelation, k_cand64, k_frame64) kernels of the two ranges
// overlap instead of running back to back.  The second range's byte offsets continue from the first
// range's total (k_layout), so the output is the same contiguous byte string.  Anything else runs
// flacgpu_analyze_device + flacgpu_pack_device.
int flacgpu_encode_device(flacgpu_ctx *c, const int32_t *d_pcm, int layout, uint32_t n_frames,
                          uint32_t last_len, uint64_t first_frame_number, uint32_t sample_rate,
                          void *stream) {
    if (!c) return FLACGPU_ERR_INVALID_ARG;
    const uint32_t B = c->opts.block_size;
    const uint32_t fbw = frame_fb_words(c->channels, c->bps, B);
    const bool eligible =
        d_pcm && n_frames >= 256 && n_frames <= c->max_frames && last_len == B && wave_block_size(B) &&
        (c->bps + (c->stereo4 ? 1u : 0u) <= 25u) &&
        (c->opts.max_lpc_order <= 16 || (B == FN && c->channels <= 4)) &&
        c->opts.max_partition_order <= 6 && (c->channels <= 4 || B == FN) && (layout == 0 || layout == 1) &&
        (size_t)fbw * sizeof(int32_t) <= 150 * 1024 && !c->timing && !getenv("FLACGPU_NO_FAST") &&
        !getenv("FLACGPU_NO_W64") && !getenv("FLACGPU_NO_FUSED_PACK") && !getenv("FLACGPU_NO_FRAME64") &&
        c->two_ranges;
    if (!eligible) {
        if (int rc = flacgpu_analyze_device(c, d_pcm, layout, n_frames, last_len, stream)) return rc;
        return flacgpu_pack_device(c, first_frame_number, sample_rate, stream);
    }
    hipStream_t st0 = stream ? (hipStream_t)stream : c->own_stream;
    hipStream_t st1 = c->aux_stream;
    Params p;
    fill_params(c, n_frames, last_len, p);
    PackParams q;
    q.first_frame_number = first_frame_number;
    q.sample_rate = sample_rate;
    q.out_words = c->d_packed;
    q.frame_off = c->d_frame_off;
    q.cap_bytes = c->packed_cap;
    const bool planar_direct = (layout == FLACGPU_LAYOUT_PLANAR) && (B % 4 == 0);
    if (planar_direct) p.planar = d_pcm;
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, 4 * sizeof(uint32_t), st0));
    HIP_TRY(hipMemsetAsync(c->d_orbits, 0, sizeof(uint32_t) * n_frames * c->ncand, st0));
    HIP_TRY(hipEventRecord(c->ev_fork, st0));
    HIP_TRY(hipStreamWaitEvent(st1, c->ev_fork, 0));
    const uint32_t h = ((n_frames / 2) + 15u) & ~15u;  // 16 frames = one autocorrelation wave group
    const bool lpc = p.max_lpc_order > 0;
    const uint32_t H = ((p.max_lpc_order + 1) + 3u) & ~3u;
    const size_t l64 = (size_t)fbw * sizeof(int32_t);
    for (int half = 0; half < 2; half++) {
        hipStream_t st = half ? st1 : st0;
        Params r = p;
        r.f0 = half ? h : 0;
        r.fcount = half ? n_frames - h : h;
        const uint32_t ncb = r.fcount * c->ncand;
        const dim3 g0(std::max<uint32_t>(1u, (B + WG * 8 - 1) / (WG * 8)), r.fcount);
        bool have_orbits = false;
        if (!planar_direct) have_orbits = launch_k0(c, d_pcm, layout, n_frames, last_len, r.f0, r.fcount, st);
        if (!have_orbits) hipLaunchKernelGGL(k_orbits, g0, dim3(WG), 0, st, r, c->d_orbits);
        if (c->stereo4 && !p.exhaustive) hipLaunchKernelGGL(k_stereo_stats, dim3(r.fcount), dim3(WG), 0, st, r);
        hipLaunchKernelGGL(k_candinfo, dim3((ncb + WG - 1) / WG), dim3(WG), 0, st, r, c->d_orbits);
        if (lpc) {
            dispatch_autocorr(H, r, r.f0, r.fcount, B, c->d_window_full, st);
            launch_lpc(r, (ncb + 63) / 64, st);
        }
        launch_cand64(r, B, (ncb + 3) / 4, st);
        hipLaunchKernelGGL(k_decide, dim3(r.fcount), dim3(64), 0, st, r);
        // frame assembly of this range; the second range's offsets continue from the first's
        if (half) HIP_TRY(hipStreamWaitEvent(st, c->ev_layout, 0));
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, st, r, q);
        if (!half) HIP_TRY(hipEventRecord(c->ev_layout, st));
        launch_frame64(r, q, B, r.fcount, l64, st);
    }
    HIP_TRY(hipEventRecord(c->ev_join, st1));
    HIP_TRY(hipStreamWaitEvent(st0, c->ev_join, 0));
    HIP_TRY(hipGetLastError());
    c->resid_valid = false;
    c->last_frames = n_frames;
    c->last_len = last_len;
    c->last_params = p;
    c->packed_valid = true;
    return FLACGPU_OK;
}

int flacgpu_fetch_frames(flacgpu_ctx *c, uint8_t *out, size_t cap, uint64_t *offsets,
                         uint64_t *total) {
    if (!c || !c->packed_valid) {
        g_last_error = "flacgpu_fetch_frames: nothing packed";
        return FLACGPU_ERR_INVALID_ARG;
    }
    const size_t F = c->last_frames;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint64_t> off;
    uint64_t *offp = offsets;
    if (!offp) {
        off.resize(F + 1);
        offp = off.data();
    }
    HIP_TRY(hipMemcpy(offp, c->d_frame_off, sizeof(uint64_t) * (F + 1), hipMemcpyDeviceToHost));
    const uint64_t bytes = offp[F];
    if (total) *total = bytes;
    if (!out || cap < bytes) {
        g_last_error = "output buffer too small";
        return FLACGPU_ERR_BUFFER_TOO_SMALL;
    }
    HIP_TRY(hipMemcpy(out, c->d_packed, bytes, hipMemcpyDeviceToHost));
    return FLACGPU_OK;
}

int flacgpu_encode_frames(flacgpu_ctx *c, const int32_t *pcm, int layout, uint32_t n_frames,
                          uint32_t last_len, uint64_t first_frame_number, uint32_t sample_rate,
                          uint8_t *out, size_t cap, uint64_t *offsets, uint64_t *total) {
    if (!c || !pcm || n_frames == 0 || n_frames > c->max_frames || last_len == 0 ||
        last_len > c->opts.block_size) {
        g_last_error = "invalid encode arguments";
        return FLACGPU_ERR_INVALID_ARG;
    }
    const size_t B = c->opts.block_size, C = c->channels;
    const size_t count = ((size_t)(n_frames - 1) * B + last_len) * C;
    HIP_TRY(hipMemcpyAsync(c->d_in, pcm, count * sizeof(int32_t), hipMemcpyHostToDevice, c->own_stream));
    int rc = flacgpu_encode_device(c, c->d_in, layout, n_frames, last_len, first_frame_number, sample_rate,
                                   c->own_stream);
    if (rc) return rc;
    return flacgpu_fetch_frames(c, out, cap, offsets, total);
}

int flacgpu_experiment_mfma_autocorr(flacgpu_ctx *c, float *kernel_ms, uint32_t *compared,
                                     uint32_t *params_differ, double *max_rel_err) {
    if (!c || c->last_frames == 0 || c->opts.max_lpc_order == 0 || c->opts.max_lpc_order > 16 ||
        c->last_len != c->opts.block_size) {
        g_last_error = "mfma experiment: needs an analysed batch of full blocks with 1 <= max_lpc_order <= 16";
        return FLACGPU_ERR_INVALID_ARG;
    }
    hipStream_t st = c->own_stream;
    Params p = c->last_params;
    const size_t nc = (size_t)p.n_frames * p.ncand;
    std::vector<LpcParams> exact(nc), mfma(nc);
    std::vector<double> ac_exact(nc * AC_LD), ac_mfma(nc * AC_LD);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(exact.data(), c->d_lpc, sizeof(LpcParams) * nc, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ac_exact.data(), c->d_ac, sizeof(double) * nc * AC_LD, hipMemcpyDeviceToHost));
    // warm-up + timed launch of the MFMA kernel, writing into the regular ac buffer
    for (int it = 0; it < 2; it++) {
        if (it == 1) (void)hipEventRecord(c->ev[0], st);
        hipLaunchKernelGGL(k_autocorr_mfma, dim3((unsigned)((nc + 3) / 4)), dim3(WG), 0, st, p,
                           p.block_size, c->d_window_full, c->d_ac);
        if (it == 1) (void)hipEventRecord(c->ev[1], st);
    }
    hipLaunchKernelGGL(k_lpc, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, st, p);
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    HIP_TRY(hipMemcpy(mfma.data(), c->d_lpc, sizeof(LpcParams) * nc, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ac_mfma.data(), c->d_ac, sizeof(double) * nc * AC_LD, hipMemcpyDeviceToHost));
    // restore the exact results so that the context stays consistent
    HIP_TRY(hipMemcpy(c->d_lpc, exact.data(), sizeof(LpcParams) * nc, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_ac, ac_exact.data(), sizeof(double) * nc * AC_LD, hipMemcpyHostToDevice));
    uint32_t cmp = 0, diff = 0;
    double worst = 0.0;
    for (size_t i = 0; i < nc; i++) {
        if (exact[i].status != 0 && mfma[i].status != 0) continue;
        cmp++;
        bool d = exact[i].status != mfma[i].status || exact[i].order != mfma[i].order ||
                 exact[i].shift != mfma[i].shift;
        if (!d)
            for (uint32_t j = 0; j < exact[i].order; j++) d |= exact[i].qlp[j] != mfma[i].qlp[j];
        diff += d;
        for (uint32_t l = 0; l <= c->opts.max_lpc_order; l++) {
            double e = ac_exact[i * AC_LD + l], m = ac_mfma[i * AC_LD + l];
            if (e != 0.0) worst = std::max(worst, std::abs((m - e) / e));
        }
    }
    if (kernel_ms) *kernel_ms = ms;
    if (compared) *compared = cmp;
    if (params_differ) *params_differ = diff;
    if (max_rel_err) *max_rel_err = worst;
    return FLACGPU_OK;
}

int flacgpu_verify_device(flacgpu_ctx *c, uint32_t sample_rate, uint64_t first_frame_number,
                          flacgpu_verify_result *result, float *kernel_ms) {
    if (!c || !c->packed_valid || !result) {
        g_last_error = "flacgpu_verify_device: nothing packed";
        return FLACGPU_ERR_INVALID_ARG;
    }
    hipStream_t st = c->own_stream;
    Params p = c->last_params;
    const size_t F = c->max_frames, C = c->channels;
    if (!c->d_decoded) {
        HIP_TRY(hipMalloc((void **)&c->d_decoded, sizeof(int32_t) * (F * C * c->ldb + 64)));
        HIP_TRY(hipMalloc((void **)&c->d_verify, sizeof(uint32_t) * 4));
    }
    PackParams q;
    q.first_frame_number = first_frame_number;
    q.sample_rate = sample_rate;
    q.out_words = c->d_packed;
    q.frame_off = c->d_frame_off;
    q.cap_bytes = c->packed_cap;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemsetAsync(c->d_verify, 0, sizeof(uint32_t) * 4, st));
    // compare against the planar PCM the analysis consumed, when it is the context's own copy
    const int32_t *expect = (p.planar == c->d_planar && p.ldb == c->ldb) ? c->d_planar : nullptr;
    Params pd = p;
    pd.ldb = c->ldb;
    (void)hipEventRecord(c->ev[0], st);
    hipLaunchKernelGGL(k_decode, dim3((p.n_frames + 63) / 64), dim3(64), 0, st, pd, q, c->d_decoded,
                       expect, c->d_verify);
    hipLaunchKernelGGL(k_crc<true>, dim3(p.n_frames), dim3(WG), 0, st, p, q, c->d_verify);
    (void)hipEventRecord(c->ev[1], st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    uint32_t counts[4];
    HIP_TRY(hipMemcpy(counts, c->d_verify, sizeof counts, hipMemcpyDeviceToHost));
    result->frames = p.n_frames;
    result->bad_structure = counts[0];
    result->bad_crc16 = counts[1];
    result->frames_pcm_differs = counts[2];
    result->samples_differ = counts[3];
    result->compared_pcm = expect != nullptr;
    if (kernel_ms) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        *kernel_ms = ms;
    }
    return FLACGPU_OK;
}

int flacgpu_fetch_decoded(flacgpu_ctx *c, int32_t *interleaved) {
    if (!c || !c->d_decoded || !interleaved || c->last_frames == 0) return FLACGPU_ERR_INVALID_ARG;
    const size_t F = c->last_frames, C = c->channels, B = c->opts.block_size, ldb = c->ldb;
    std::vector<int32_t> planar(F * C * ldb);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(planar.data(), c->d_decoded, sizeof(int32_t) * planar.size(), hipMemcpyDeviceToHost));
    size_t o = 0;
    for (size_t f = 0; f < F; f++) {
        const size_t n = (f + 1 == F) ? c->last_len : B;
        for (size_t i = 0; i < n; i++)
            for (size_t ch = 0; ch < C; ch++) interleaved[o++] = planar[(f * C + ch) * ldb + i];
    }
    return FLACGPU_OK;
}

int flacgpu_get_stats(flacgpu_ctx *c, flacgpu_stats *out) {
    if (!c || !out) return FLACGPU_ERR_INVALID_ARG;
    uint32_t s[4];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(s, c->d_stats, sizeof s, hipMemcpyDeviceToHost));
    out->frames = c->last_frames;
    out->lpc_failed = s[0];
    out->order_ties = s[1];
    out->log2_edge = s[2];
    return FLACGPU_OK;
}

void *flacgpu_device_buffer(flacgpu_ctx *c, int which) {
    if (!c) return nullptr;
    switch (which) {
    case 0: return c->d_fplan;
    case 1: return c->d_out;
    case 2: return ensure_residual_rows(c) == FLACGPU_OK ? c->d_resid : nullptr;
    case 3: return c->d_planar;
    case 4: return c->d_packed;
    case 5: return c->d_frame_off;
    default: return nullptr;
    }
}

int flacgpu_get_kernel_ms(flacgpu_ctx *c, float ms[FLACGPU_N_KERNELS]) {
    if (!c || !ms) return FLACGPU_ERR_INVALID_ARG;
    memcpy(ms, c->last_ms, sizeof(float) * FLACGPU_N_KERNELS);
    return FLACGPU_OK;
}

}  // extern "C"
