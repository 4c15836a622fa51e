// Does the ORDER of fast-class (v_add_u32 / v_sub_u32) and full-cost (v_sad_u32) instructions matter for issue cost?
// The same 1:1 population in runs of 1, 2, 4 and 8 instructions of a kind, independent chains, W waves per SIMD
// (r03 result, profiles/r03_issue_group_ubench.txt: at three waves per SIMD 1.74 ns alternating, 1.66 in runs of 8,
// 1.56 / 1.88 for the two kinds alone -- grouping is worth <= 4 %, and the compiler schedules the passes anyway).  hipcc --offload-arch=gfx950 -O2 -o issue_group issue_group.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITER 16384
#define REP 16
// RUN: instructions of one kind in a row (1..8 inside one unrolled body of 16)
template <int RUN, bool FAST_ONLY, bool SLOW_ONLY> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a[REP], b = (seed ^ threadIdx.x) & 0x3ff8, c = seed * 3 + 1;
    for (int i = 0; i < REP; i++) a[i] = threadIdx.x * 7 + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < REP; i++) {
            bool slow = (i / RUN) & 1;
            if (FAST_ONLY) slow = false;
            if (SLOW_ONLY) slow = true;
            if (slow) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            else asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < REP; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + b + c;
}
static int n_cu = 256;
template <int RUN, bool F, bool S> void run(const char *name, uint32_t *d, int w) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = n_cu * w;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<RUN, F, S>), dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k<RUN, F, S>), dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s W=%d  %.3f ms  %.3f ns per wave-instr per SIMD\n", name, w, ms, ms * 1e6 / ((double)ITER * REP * w));
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); n_cu = pr.multiProcessorCount;
    uint32_t *d; hipMalloc(&d, (size_t)n_cu * 8 * 256 * 4);
    for (int w : {2, 3, 4}) {
        run<1, true, false>("fast only", d, w);
        run<1, false, true>("sad only", d, w);
        run<1, false, false>("runs of 1", d, w);
        run<2, false, false>("runs of 2", d, w);
        run<4, false, false>("runs of 4", d, w);
        run<8, false, false>("runs of 8", d, w);
    }
    return 0;
}
