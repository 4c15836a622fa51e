// Does the register placement of a v_fma_f64's three 64-bit operands change its issue cost (VGPR banks)?  Fixed physical
// registers: sixteen accumulators v[32:63], operand pairs at chosen offsets; W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o f64_banks f64_banks.hip && ./f64_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 16384
#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
// MODE 0: fma acc, a(v[4:5]), b(v[8:9]), acc       a, b in different bank pairs (4%4=0, 8%4=0 -> same!)
// We enumerate (A, B) offsets explicitly instead.
template <int A, int B, int KIND> __global__ void __launch_bounds__(256) k(double *out, double seed) {
    double r = 0;
    asm volatile("v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0x3ff00000\n"
                 "v_mov_b32 v8, 0\n v_mov_b32 v9, 0x3ff00000\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0x3ff00000\n"
                 "v_mov_b32 v12, 0\n v_mov_b32 v13, 0x3ff00000\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0x3ff00000\n" ::: CLOB);
    for (int i = 32; i < 64; i++) { }
    asm volatile(
        "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"
        "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
        "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"
        "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n" ::: CLOB);
    for (int it = 0; it < ITER; it++) {
#define F(acc) if (KIND == 0) asm volatile("v_fma_f64 v[" #acc ":" #acc "+1], v[%0:%0+1], v[%1:%1+1], v[" #acc ":" #acc "+1]" :: "n"(A), "n"(B) : CLOB); \
               else if (KIND == 1) asm volatile("v_mul_f64 v[" #acc ":" #acc "+1], v[%0:%0+1], v[%1:%1+1]" :: "n"(A), "n"(B) : CLOB); \
               else asm volatile("v_add_f64 v[" #acc ":" #acc "+1], v[" #acc ":" #acc "+1], v[%0:%0+1]" :: "n"(A) : CLOB);
        F(32) F(34) F(36) F(38) F(40) F(42) F(44) F(46) F(48) F(50) F(52) F(54) F(56) F(58) F(60) F(62)
    }
    asm volatile("v_mov_b32 %0, v32" : "=v"(((uint32_t *)&r)[0]) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;
}
static int n_cu = 256;
template <int A, int B, int KIND> void run(double *d, int w) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = n_cu * w;
    hipLaunchKernelGGL((k<A, B, KIND>), dim3(blocks), dim3(256), 0, 0, d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B, KIND>), dim3(blocks), dim3(256), 0, 0, d, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s a=v[%d:%d] b=v[%d:%d]  W=%d  %.4f ns per wave-instr per SIMD\n", KIND == 0 ? "fma acc,a,b,acc" : KIND == 1 ? "mul acc,a,b    " : "add acc,acc,a  ", A, A + 1, B, B + 1, w,
           ms * 1e6 / ((double)ITER * 16 * w));
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); n_cu = pr.multiProcessorCount;
    double *d; hipMalloc(&d, (size_t)n_cu * 8 * 256 * 8);
#define RR(a, b, kd) run<a, b, kd>(d, 1); run<a, b, kd>(d, 2); run<a, b, kd>(d, 4);
    RR(4, 8, 0) RR(4, 6, 0) RR(4, 4, 0) RR(6, 10, 0) RR(4, 10, 0)
    RR(4, 8, 1) RR(4, 6, 1) RR(4, 8, 2) RR(6, 8, 2)
    return 0;
}
