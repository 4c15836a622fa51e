// Does the double rate of the simple 32-bit VALU classes (v_add_u32 & co: 1.15 ns per wave-instruction per SIMD when a
// kernel issues nothing else, 1.6 ns "inside a mix", tools/ubench/issue_rate2.hip) come back when the simple instructions
// stand in RUNS?  Pattern: R x v_add_u32, then S x v_sad_u32, all on independent registers, repeated; W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o issue_runs issue_runs.hip && ./issue_runs out.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 16384
template <int R, int S, int KIND> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    constexpr int N = R + S;
    uint32_t a[N], b = (seed ^ threadIdx.x) & 0x3ff8, c = seed * 3 + 1;
    for (int i = 0; i < N; i++) a[i] = threadIdx.x * 7 + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (i < R) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 1) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % N]));   // two VGPR sources
                if (KIND == 2) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
            } else {
                asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < N; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int n_cu = 256;
template <int R, int S, int KIND> void run(uint32_t *d, int w, FILE *js, bool &first) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = n_cu * w;
    hipLaunchKernelGGL((k<R, S, KIND>), dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, S, KIND>), dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)ITER * (R + S) * w);
    printf("kind %d  R=%2d S=%2d  W=%d  %.4f ns per wave-instr per SIMD   (linear model 1.15 / 1.80: %.4f)\n", KIND, R, S, w, ns,
           (R * 1.15 + S * 1.80) / (R + S));
    fprintf(js, "%s\n {\"kind\": %d, \"run_simple\": %d, \"run_sad\": %d, \"waves_per_simd\": %d, \"ns_per_inst\": %.4f}", first ? "" : ",", KIND, R, S, w, ns);
    first = false;
}
int main(int argc, char **argv) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); n_cu = pr.multiProcessorCount;
    uint32_t *d; hipMalloc(&d, (size_t)n_cu * 8 * 256 * 4);
    FILE *js = fopen(argc > 1 ? argv[1] : "issue_runs.json", "w");
    fprintf(js, "{\"cus\": %d, \"iter\": %d, \"results\": [", n_cu, ITER);
    bool first = true;
#define RR(r, s, kd) run<r, s, kd>(d, 1, js, first); run<r, s, kd>(d, 2, js, first); run<r, s, kd>(d, 4, js, first);
    RR(16, 0, 0) RR(0, 16, 0) RR(1, 1, 0) RR(2, 1, 0) RR(2, 2, 0) RR(4, 2, 0) RR(4, 4, 0) RR(8, 4, 0) RR(8, 8, 0) RR(16, 8, 0) RR(16, 16, 0) RR(32, 16, 0) RR(32, 32, 0)
    RR(6, 5, 0) RR(12, 10, 0) RR(24, 20, 0) RR(48, 40, 0)
    RR(16, 0, 1) RR(8, 8, 1) RR(32, 32, 1) RR(1, 1, 1)
    RR(16, 0, 2) RR(8, 8, 2) RR(32, 32, 2) RR(1, 1, 2)
    fprintf(js, "\n]}\n");
    fclose(js);
    return 0;
}
