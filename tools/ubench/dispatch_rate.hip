// dispatch_rate.hip -- what launching many small workgroups costs: a kernel that does (almost) nothing, by workgroup size and
// LDS allocation.  k_sub64 (one 64-thread workgroup per subframe, 65 536 of them per batch) asked.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_nop(uint32_t *out, uint32_t spin) {
    extern __shared__ uint32_t lds[];
    uint32_t v = threadIdx.x;
    for (uint32_t i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x12345678u) out[0] = v + lds[0];
}
int main() {
    uint32_t *d;
    CK(hipMalloc(&d, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)k_nop, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    printf("{\"runs\": [\n");
    bool first = true;
    for (uint32_t spin : {0u, 3000u})
        for (uint32_t threads : {64u, 128u, 256u})
            for (uint32_t lds : {0u, 14336u}) {
                const uint32_t waves = 65536, grid = waves * 64 / threads;
                const uint32_t l = lds * (threads / 64);
                for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_nop, dim3(grid), dim3(threads), l, 0, d, spin);
                CK(hipEventRecord(a));
                for (int it = 0; it < 10; it++) hipLaunchKernelGGL(k_nop, dim3(grid), dim3(threads), l, 0, d, spin);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                printf("%s {\"waves\": %u, \"threads_per_workgroup\": %u, \"lds_bytes_per_workgroup\": %u, \"alu_iterations\": %u, \"ms_per_launch\": %.4f}",
                       first ? "" : ",\n", waves, threads, l, spin, ms / 10);
                first = false;
            }
    printf("\n]}\n");
    return 0;
}
