// sustained f64 issue rate: LG independent accumulate chains acc[k] += w * h[k] (mul then add, no FMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int LG, int MODE> __global__ void __launch_bounds__(256) k(double *out, int iters, double seed) {
    double acc[LG], h[LG];
    for (int i = 0; i < LG; i++) { acc[i] = -0.0; h[i] = seed * (threadIdx.x + i + 1); }
    double w = seed * 0.5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int k = 0; k < LG; k++) {
                if (MODE == 0) { double pr; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(pr) : "v"(w), "v"(h[k])); asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[k]) : "v"(pr)); }
                if (MODE == 1) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(w), "v"(h[k])); }
                if (MODE == 2) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[k]) : "v"(h[k])); }
                if (MODE == 3) { uint64_t *a = (uint64_t*)&acc[k]; asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(*a) : "v"(h[k])); }
            }
        }
    }
    double r = 0; for (int i = 0; i < LG; i++) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int LG, int MODE> void run(const char *name, double *d, int wps, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * wps;
    hipLaunchKernelGGL((k<LG, MODE>), dim3(blocks), dim3(256), 0, 0, d, 16, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LG, MODE>), dim3(blocks), dim3(256), 0, 0, d, iters, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ninstr = (double)iters * 16 * LG * (MODE == 0 ? 2 : 1) * wps;  // per SIMD
    printf("%-22s LG=%d waves/simd %d: %.3f ms  %.2f cycles@2.4GHz per wave-instr per SIMD\n", name, LG, wps, ms, ms * 1e-3 * 2.4e9 / ninstr);
}
int main() {
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<7, 0>("mul+add chains", d, 1, 4000); run<7, 0>("mul+add chains", d, 2, 4000); run<7, 0>("mul+add chains", d, 4, 4000);
    run<3, 0>("mul+add chains", d, 1, 8000); run<3, 0>("mul+add chains", d, 2, 8000); run<3, 0>("mul+add chains", d, 4, 8000);
    run<13, 0>("mul+add chains", d, 1, 2000); run<13, 0>("mul+add chains", d, 2, 2000);
    run<7, 1>("fma chains", d, 1, 8000); run<7, 1>("fma chains", d, 2, 8000); run<7, 1>("fma chains", d, 4, 8000);
    run<7, 2>("add chains", d, 1, 8000); run<7, 2>("add chains", d, 2, 8000);
    run<7, 3>("u64 add chains", d, 1, 8000); run<7, 3>("u64 add chains", d, 2, 8000);
    return 0;
}
