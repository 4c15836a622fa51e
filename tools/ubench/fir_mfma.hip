// fir_mfma.hip -- VERDICT r03 item 7: the LPC residual FIR (encode.rs:3174-3203), an exact integer contraction, on the i8 matrix
// cores against the VALU form the product kernels use.  Stand-alone experiment: both kernels take the same planar 24-bit
// samples and quantised coefficients (random, precision 12, order T), compute every residual x[i] - (sum_j c_j x[i-1-j] >> shift)
// with zero history before the block, and leave sum |r| per 64-sample partition; the residuals of the first candidates are
// compared value by value (the i64 sum is exact under any association, so the two must agree bit for bit).
//
//   VALU:  one wave per candidate, lane = 64 consecutive samples in registers, T v_mad_i64_i32 per sample (kernels/wave_cand.inc
//          fir64), history from the previous lane.
//   MFMA:  one wave per candidate.  The samples go to LDS as three byte planes (24-bit samples: two unsigned low limbs, stored
//          minus 128 so that they are i8, and the signed top byte).  An MFMA v_mfma_i32_16x16x64_i8 multiplies A = 16 blocks of
//          16 outputs x their 64-sample windows (one limb plane, 16 aligned bytes per lane) by B = the banded Toeplitz matrix of
//          one coefficient limb (balanced digits c = 256 ch + cl, both i8): D[block][n] = sum_k limb(x[16 block - 48 + k])
//          limb(c[n + 47 - k]) -- all 256 outputs useful, orders up to 32.  3 x 2 limb products = 6 MFMAs per 256 outputs into
//          4 accumulators (weights 2^0, 2^8, 2^16, 2^24; the -128 offsets come back as accumulator initial values), recombined
//          in i64 with two shifts-and-adds and one v_mad_i64_i32 per output.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fir_mfma.hip -o tools/ubench/bin/fir_mfma && tools/ubench/bin/fir_mfma
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int N = 4096, PADB = 48, PLANE = PADB + N + 16;   // bytes per limb plane in LDS
struct Cand { int32_t coef[32]; int32_t shift; int32_t pad[3]; };

template <int T, bool DUMP>
__global__ void __launch_bounds__(256) k_fir_valu(const int32_t *__restrict__ x, const Cand *__restrict__ cd, uint32_t *__restrict__ sums,
                                                  int32_t *__restrict__ resid) {
    const uint32_t lane = threadIdx.x & 63, cand = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int32_t *row = x + (size_t)cand * N;
    int32_t xs[64], hist[T];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int4 v = reinterpret_cast<const int4 *>(row)[16 * lane + k];
        xs[4 * k] = v.x; xs[4 * k + 1] = v.y; xs[4 * k + 2] = v.z; xs[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < T; k++) {
        const int32_t h = __shfl_up(xs[64 - T + k], 1, 64);
        hist[k] = lane ? h : 0;
    }
    const Cand *c = cd + cand;
    int32_t co[T];
#pragma unroll
    for (int j = 0; j < T; j++) co[j] = __builtin_amdgcn_readfirstlane(c->coef[j]);
    const int32_t shift = __builtin_amdgcn_readfirstlane(c->shift);
    uint32_t acc = 0;
#pragma unroll
    for (int e = 63; e >= 0; e--) {
        long long s = 0;
#pragma unroll
        for (int j = 0; j < T; j++) {
            const int i = e - 1 - j;
            s += (long long)(i >= 0 ? xs[i >= 0 ? i : 0] : hist[i >= 0 ? 0 : T + i]) * (long long)co[j];
        }
        const int32_t r = (int32_t)((uint32_t)xs[e] - (uint32_t)(int32_t)(s >> shift));
        xs[e] = r;
        acc += (uint32_t)(r < 0 ? -r : r);
        if ((e & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    sums[(size_t)cand * 64 + lane] = acc;
    if constexpr (DUMP) {
#pragma unroll
        for (int e = 0; e < 64; e++) resid[(size_t)cand * N + 64 * lane + e] = xs[e];
    }
}

__device__ __forceinline__ uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }

template <int T, bool DUMP>
__global__ void __launch_bounds__(256) k_fir_mfma(const int32_t *__restrict__ x, const Cand *__restrict__ cd, uint32_t *__restrict__ sums,
                                                  int32_t *__restrict__ resid) {
    static_assert(T <= 32, "the 64-sample window covers 16 outputs and 32 taps");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cand = blockIdx.x * 4 + wave;
    uint8_t *base = lds + wave * (3 * PLANE + 2 * 96 + 32);
    uint8_t *pl[3] = {base, base + PLANE, base + 2 * PLANE};
    int8_t *R[2] = {reinterpret_cast<int8_t *>(base + 3 * PLANE), reinterpret_cast<int8_t *>(base + 3 * PLANE + 96)};
    const int32_t *row = x + (size_t)cand * N;
    const Cand *c = cd + cand;
    // ---- planes: the lane's 64 consecutive samples as 3 x 64 bytes; low limbs minus 128 (xor 0x80)
#pragma unroll
    for (int p = 0; p < 3; p++)
        if (lane < 12) reinterpret_cast<uint32_t *>(pl[p])[lane] = p < 2 ? 0x80808080u : 0u;   // sample "0" before the block: limb 0 -> -128 as i8
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t o0[4], o1[4], o2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int4 v = reinterpret_cast<const int4 *>(row)[16 * lane + 4 * k + u];
            const uint32_t ab01 = perm((uint32_t)v.y, (uint32_t)v.x, 0x05010400u), cd01 = perm((uint32_t)v.w, (uint32_t)v.z, 0x05010400u);
            const uint32_t ab2 = perm((uint32_t)v.y, (uint32_t)v.x, 0x00000602u), cd2 = perm((uint32_t)v.w, (uint32_t)v.z, 0x00000602u);
            o0[u] = perm(cd01, ab01, 0x05040100u) ^ 0x80808080u;
            o1[u] = perm(cd01, ab01, 0x07060302u) ^ 0x80808080u;
            o2[u] = perm(cd2, ab2, 0x05040100u);
        }
        *reinterpret_cast<uint4 *>(pl[0] + PADB + 64 * lane + 16 * k) = make_uint4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<uint4 *>(pl[1] + PADB + 64 * lane + 16 * k) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        *reinterpret_cast<uint4 *>(pl[2] + PADB + 64 * lane + 16 * k) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    }
    // ---- coefficient limbs (balanced digits), reversed and zero padded: R[i] = limb(c[62 - i]), 0 <= i < 96
    int32_t slo = 0, shi = 0;
    {
        for (uint32_t i = lane; i < 96; i += 64) {
            const int t = 62 - (int)i;
            const int32_t cv = (t >= 0 && t < T) ? c->coef[t] : 0;
            const int32_t cl = (int32_t)(int8_t)(cv & 0xFF), ch = (cv - cl) >> 8;
            R[0][i] = (int8_t)cl;
            R[1][i] = (int8_t)ch;
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int32_t cv = __builtin_amdgcn_readfirstlane(c->coef[t]);
            const int32_t cl = (int32_t)(int8_t)(cv & 0xFF);
            slo += cl;
            shi += (cv - cl) >> 8;
        }
    }
    const int32_t shift = __builtin_amdgcn_readfirstlane(c->shift);
    __syncthreads();
    const uint32_t n = lane & 15, q = lane >> 4;
    // B fragment of limb L: byte j <-> window position k = 16 q + j: limb(c[n + 47 - k]) = R[15 - n + k]
    v4i B[2];
#pragma unroll
    for (int L = 0; L < 2; L++) {
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v |= (uint32_t)(uint8_t)R[L][15 - n + 16 * q + 4 * d + b] << (8 * b);
            w[d] = v;
        }
        B[L] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    }
    // the -128 of the two low sample limbs, back as accumulator initial values
    const int32_t i0 = 128 * slo, i8 = 128 * (slo + shi), i16 = 128 * shi;
    // the samples at this lane's output positions, one group ahead (a reload per group in the loop body left every
    // iteration waiting for a global round trip: 0.24 ms per launch instead of 0.1)
    int32_t xn[4];
#pragma unroll
    for (int v = 0; v < 4; v++) xn[v] = row[16u * (4u * q + v) + n];
#pragma unroll 2
    for (int g = 0; g < 16; g++) {
        int32_t xc[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            xc[v] = xn[v];
            xn[v] = row[16u * (16u * (g < 15 ? g + 1 : 15) + 4u * q + v) + n];
        }
        // A fragments: lane (r = n, q): the 16 bytes of block 16 g + r - 3 + q of each plane
        const uint32_t off = PADB + 16 * (16 * g + n + q) - 48;
        const v4i A0 = *reinterpret_cast<const v4i *>(pl[0] + off), A1 = *reinterpret_cast<const v4i *>(pl[1] + off),
                  A2 = *reinterpret_cast<const v4i *>(pl[2] + off);
        v4i w0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[0], v4i{i0, i0, i0, i0}, 0, 0, 0);
        v4i w8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B[1], v4i{i8, i8, i8, i8}, 0, 0, 0);
        w8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[0], w8, 0, 0, 0);
        v4i w16 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B[1], v4i{i16, i16, i16, i16}, 0, 0, 0);
        w16 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2, B[0], w16, 0, 0, 0);
        const v4i w24 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A2, B[1], v4i{0, 0, 0, 0}, 0, 0, 0);
        uint32_t a = 0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int32_t lo = w0[v] + (w8[v] << 8), hi = w16[v] + (w24[v] << 8);
            const long long s = (long long)hi * 65536ll + (long long)lo;
            const uint32_t idx = 16u * (16u * g + 4u * q + v) + n;
            const int32_t xi = xc[v];
            const int32_t r = (int32_t)((uint32_t)xi - (uint32_t)(int32_t)(s >> shift));
            a += (uint32_t)(r < 0 ? -r : r);
            if constexpr (DUMP) resid[(size_t)cand * N + idx] = r;
        }
        // partition (64 samples = the 4 blocks of lane group q of this MFMA group) sum: over the 16 lanes of a DPP row
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (n == 0) sums[(size_t)cand * 64 + 4 * g + q] = a;
    }
}

template <int T>
static void run(int ncand, const int32_t *dx, const Cand *dc, uint32_t *ds0, uint32_t *ds1, int32_t *dr0, int32_t *dr1, int dump, bool first) {
    const size_t lds = 4 * (3 * PLANE + 2 * 96 + 32);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)k_fir_mfma<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k_fir_mfma<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // verification launches (residuals of the first `dump` candidates dumped by both)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fir_valu<T, true>), dim3(dump / 4), dim3(256), 0, 0, dx, dc, ds0, dr0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fir_mfma<T, true>), dim3(dump / 4), dim3(256), lds, 0, dx, dc, ds1, dr1);
    CK(hipDeviceSynchronize());
    std::vector<int32_t> r0((size_t)dump * N), r1((size_t)dump * N);
    CK(hipMemcpy(r0.data(), dr0, r0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), dr1, r1.size() * 4, hipMemcpyDeviceToHost));
    size_t rdiff = 0;
    for (size_t i = 0; i < r0.size(); i++) rdiff += r0[i] != r1[i];
    float ms[2] = {0, 0};
    for (int which = 0; which < 2; which++) {
        for (int it = 0; it < 3; it++) {
            if (which == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fir_valu<T, false>), dim3(ncand / 4), dim3(256), 0, 0, dx, dc, ds0, dr0);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fir_mfma<T, false>), dim3(ncand / 4), dim3(256), lds, 0, dx, dc, ds1, dr1);
        }
        CK(hipEventRecord(a));
        for (int it = 0; it < 20; it++) {
            if (which == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fir_valu<T, false>), dim3(ncand / 4), dim3(256), 0, 0, dx, dc, ds0, dr0);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fir_mfma<T, false>), dim3(ncand / 4), dim3(256), lds, 0, dx, dc, ds1, dr1);
        }
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[which], a, b));
        ms[which] /= 20;
    }
    std::vector<uint32_t> s0((size_t)ncand * 64), s1((size_t)ncand * 64);
    CK(hipMemcpy(s0.data(), ds0, s0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(s1.data(), ds1, s1.size() * 4, hipMemcpyDeviceToHost));
    size_t sdiff = 0;
    for (size_t i = 0; i < s0.size(); i++) sdiff += s0[i] != s1[i];
    const double mfmas = (double)ncand * 16 * 6;
    const double util = mfmas * 16.0 / (ms[1] * 1e-3 * 1024 * 2.4e9);   // 16 cycles per MFMA on one of 1024 SIMDs at the nominal 2.4 GHz
    printf("%s {\"taps\": %d, \"candidates\": %d, \"samples\": %lld, \"valu_ms\": %.4f, \"mfma_ms\": %.4f, \"residuals_compared\": %zu, "
           "\"residuals_differ\": %zu, \"partition_sums_differ\": %zu, \"mfma_instructions\": %.0f, \"mfma_pipe_utilisation_at_2.4GHz\": %.4f}",
           first ? "" : ",\n", T, ncand, (long long)ncand * N, ms[0], ms[1], r0.size(), rdiff, sdiff, mfmas, util);
}

int main() {
    const int ncand = 32768, dump = 64;
    std::vector<int32_t> hx((size_t)ncand * N);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &v : hx) v = (int32_t)(rnd() % 16777216ull) - 8388608;          // full-range 24-bit samples
    std::vector<Cand> hc(ncand);
    for (auto &c : hc) {
        for (int j = 0; j < 32; j++) c.coef[j] = (int32_t)(rnd() % 4096ull) - 2048;   // precision 12
        c.shift = (int32_t)(rnd() % 16ull);
    }
    int32_t *dx, *dr0, *dr1;
    Cand *dc;
    uint32_t *ds0, *ds1;
    CK(hipMalloc(&dx, hx.size() * 4));
    CK(hipMalloc(&dc, hc.size() * sizeof(Cand)));
    CK(hipMalloc(&ds0, (size_t)ncand * 64 * 4));
    CK(hipMalloc(&ds1, (size_t)ncand * 64 * 4));
    CK(hipMalloc(&dr0, (size_t)dump * N * 4));
    CK(hipMalloc(&dr1, (size_t)dump * N * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), hc.size() * sizeof(Cand), hipMemcpyHostToDevice));
    printf("{\"what\": \"LPC residual FIR of 32768 candidates x 4096 24-bit samples: VALU (v_mad_i64_i32, lane = 64 samples) against i8 MFMA "
           "(3 sample limbs x 2 coefficient limbs, banded Toeplitz B), both leaving sum |r| per 64-sample partition\", \"runs\": [\n");
    run<12>(ncand, dx, dc, ds0, ds1, dr0, dr1, dump, true);
    run<16>(ncand, dx, dc, ds0, ds1, dr0, dr1, dump, false);
    run<24>(ncand, dx, dc, ds0, ds1, dr0, dr1, dump, false);
    run<32>(ncand, dx, dc, ds0, ds1, dr0, dr1, dump, false);
    printf("\n]}\n");
    return 0;
}
