// instruction issue-rate microbenchmark for gfx950 (wave64): cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define REP 8
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a[REP], b = (seed ^ threadIdx.x) & 0x3ff8, c = seed * 3 + 1;
    uint64_t w[REP];
    for (int i = 0; i < REP; i++) { a[i] = threadIdx.x * 7 + i; w[i] = a[i]; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < REP; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "s10", "s11");
            if (OP == 4) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 5) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == 6) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(w[i]));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 8) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "s10", "s11");
            if (OP == 9) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 10) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 11) asm volatile("v_max_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 12) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 13) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
            if (OP == 14) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == 16) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b));
            if (OP == 17) asm volatile("v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(a[i]), "+v"(b) , "+v"(c):: "vcc");
            if (OP == 18) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == 19) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 20) asm volatile("v_bfe_i32 %0, %0, 0, 12" : "+v"(a[i]));
            if (OP == 21) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 23) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == 24) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == 25) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == 26) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(w[i]) : "v"(a[i]));
            if (OP == 27) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(w[i]) : "v"(b));
            if (OP == 28) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == 22) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < REP; i++) r += a[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + b + c;
}
template <int OP> void run(const char *name, uint32_t *d, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // blocks of 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    double cycles = ms * 1e-3 * clk_khz * 1e3;
    double per = cycles / ((double)ITER * REP * waves_per_simd);
    printf("%-18s waves/simd %d: %.3f ms, %.2f cycles per wave-instr per SIMD (clk %d MHz)\n", name, waves_per_simd, ms, per, clk_khz / 1000);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(op, nm) run<op>(nm, d, 1); run<op>(nm, d, 2); run<op>(nm, d, 4);
    R(0, "v_add_u32") R(1, "v_mul_lo_u32") R(2, "v_mad_i32_i24") R(3, "v_mad_i64_i32") R(4, "v_sad_u32")
    R(5, "v_lshl_add_u64") R(6, "v_ashrrev_i64") R(7, "v_cndmask_b32") R(8, "v_mad_u64_u32") R(9, "v_mul_hi_u32")
    R(10, "v_add3_u32") R(11, "v_max_i32") R(12, "v_mul_i32_i24") R(13, "v_ffbh_u32") R(14, "v_lshrrev_b32")
    R(15, "v_mov_dpp") R(16, "ds_bpermute+wait") R(17, "add_co+addc") R(18, "v_pk_add_u16") R(19, "v_alignbit") R(20, "v_bfe_i32")
    R(23, "v_add_f64") R(24, "v_mul_f64") R(25, "v_fma_f64") R(26, "v_cvt_f64_i32") R(27, "ds_read_b64+wait") R(28, "v_pk_add_f32")
    return 0;
}
