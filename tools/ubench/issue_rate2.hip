// Sustained instruction issue cost on gfx950 (wave64), long kernels: for each instruction class the kernel runs
// ITER x REP independent-chain instructions per wave (about 1 M, >= 2 ms) with W waves resident per SIMD and reports
//   cyc/inst = shader-clock cycles (s_memtime) a SIMD spends per wave-instruction it issues = elapsed / (ITER REP W)
// together with the clock the run sustained (elapsed cycles / wall time).  Feeds tools/issue_floor.py (the attainable
// time of a kernel from its instruction histogram).   hipcc --offload-arch=gfx950 -O2 -o issue_rate2 issue_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define ITER 32768
#define REP 16
enum { ADD = 0, SUB, XOR, ASHR, LSHR_V, MOV, ADD_LIT, SAD, ADD3, OR3, BITOP3, MAD_I64, MAD_I24, LSHL_ADD_U64, ASHR_I64, ALIGNBIT,
       CNDMASK_S, MOV_DPP, ADD_DPP, READLANE, MUL_F64, ADD_F64, FMA_F64, CVT_F64_I32, MAX_I32, MUL_LO, PERM, SNOP, DS_READ_B128,
       MAD_I64_CHAIN, MIX_FIR, AND, OR, LSHL, CMP_VCC, CMP_SGPR, ADD_CO, ADDC, SUBREV, MIN_U32, MUL_I24, ADD_E64_S, BFE_U32, LSHL_ADD,
       AND_OR, MOV_B64, READFIRST, WRITELANE, LDEXP_F64, NOT, SUB_CO, MIX_AB, MIX_STATS, MIX_AAB, NOPS };
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t *out, uint64_t *cyc, uint32_t seed) {
    uint32_t a[REP], b = (seed ^ threadIdx.x) & 0x3ff8, c = seed * 3 + 1;
    uint64_t w[REP];
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 q4 = {0, 0, 0, 0};
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < REP; i++) { a[i] = threadIdx.x * 7 + i; w[i] = a[i]; }
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < REP; i++) {
            if (OP == ADD) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == SUB) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i]));
            if (OP == LSHR_V) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            if (OP == ADD_LIT) asm volatile("v_add_u32 %0, 0x40000000, %0" : "+v"(a[i]));
            if (OP == SAD) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == ADD3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == OR3) asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x48" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == MAD_I64) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "s10", "s11");
            if (OP == MAD_I24) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == ASHR_I64) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(w[i]));
            if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == CNDMASK_S) asm volatile("v_cndmask_b32 %0, %0, %1, s[12:13]" : "+v"(a[i]) : "v"(b) : "s12", "s13");
            if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == ADD_DPP) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == READLANE) asm volatile("v_readlane_b32 s14, %0, 3" : : "v"(a[i]) : "s14");
            if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(w[i]) : "v"(a[i]));
            if (OP == MAX_I32) asm volatile("v_max_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == PERM) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == SNOP) asm volatile("s_nop 0");
            if (OP == DS_READ_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(q4) : "v"(b));
            if (OP == AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OR) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == CMP_VCC) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == CMP_SGPR) asm volatile("v_cmp_gt_u32 s[12:13], %0, %1" : : "v"(a[i]), "v"(b) : "s12", "s13");
            if (OP == ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == ADDC) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == SUB_CO) asm volatile("v_sub_co_u32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == SUBREV) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == MIN_U32) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == MUL_I24) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == ADD_E64_S) asm volatile("v_add_u32_e64 %0, %0, s14" : "+v"(a[i]) : : "s14");
            if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 12" : "+v"(a[i]));
            if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(w[i]) : "v"(w[(i + 1) % REP]));
            if (OP == READFIRST) asm volatile("v_readfirstlane_b32 s14, %0" : : "v"(a[i]) : "s14");
            if (OP == WRITELANE) asm volatile("v_writelane_b32 %0, s14, 3" : "+v"(a[i]) : : "s14");
            if (OP == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(w[i]) : "v"(b));
            if (OP == NOT) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
            // mixes of a fast-class (v_add_u32) and a full-cost (v_sad_u32) instruction: alternating, 2:1, and the
            // order statistics' own pattern (7 fast + 5 sad per sample, here 16 = 9 + 7)
            if (OP == MIX_AB) {
                if (i & 1) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            }
            if (OP == MIX_AAB) {
                if (i % 3 == 2) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            }
            if (OP == MIX_STATS) {
                if (i == 1 || i == 3 || i == 5 || i == 8 || i == 10 || i == 12 || i == 14) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (i & 1) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                else asm volatile("v_add_u32 %0, 0x40000000, %0" : "+v"(a[i]));
            }
            // a dependent chain of 64-bit multiply-adds, the shape of the in-place FIR (one accumulator per output)
            if (OP == MAD_I64_CHAIN) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(w[0]) : "v"(a[i]), "v"(c) : "s10", "s11");
            // the FIR's own mix: 12 dependent mads, one 64-bit shift, one subtract (REP = 16: 12 + 1 + 1 + 2 adds)
            if (OP == MIX_FIR) {
                if (i < 12) asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0" : "+v"(w[0]) : "v"(a[i]), "v"(c) : "s10", "s11");
                else if (i == 12) asm volatile("v_ashrrev_i64 %0, 9, %0" : "+v"(w[0]));
                else asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    uint32_t r = 0;
    for (int i = 0; i < REP; i++) r += a[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + b + c + q4.x + q4.w;
}
static int n_cu = 256;
template <int OP> void run(const char *name, uint32_t *d, uint64_t *dc, int waves_per_simd, FILE *js, bool &first) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = n_cu * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, dc, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static uint64_t h[256 * 8 * 4];
    hipMemcpy(h, dc, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks * 4; i++) avg += (double)h[i];
    avg /= blocks * 4;
    // s_memtime ticks at a fixed 100 MHz on this part: convert through the wall time instead
    const double per_wall_ns = ms * 1e6 / ((double)ITER * REP * waves_per_simd);
    printf("%-16s W=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (counter %.0f ticks)\n", name, waves_per_simd, ms, per_wall_ns, avg);
    fprintf(js, "%s\n {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_inst\": %.4f}", first ? "" : ",", name, waves_per_simd, ms, per_wall_ns);
    first = false;
}
int main(int argc, char **argv) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); n_cu = pr.multiProcessorCount;
    uint32_t *d; hipMalloc(&d, (size_t)n_cu * 8 * 256 * 4);
    uint64_t *dc; hipMalloc(&dc, (size_t)n_cu * 8 * 4 * 8);
    FILE *js = fopen(argc > 1 ? argv[1] : "issue_rate2.json", "w");
    fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"iter\": %d, \"rep\": %d, \"results\": [", pr.name, n_cu, ITER, REP);
    bool first = true;
#define R(op, nm) run<op>(nm, d, dc, 1, js, first); run<op>(nm, d, dc, 2, js, first); run<op>(nm, d, dc, 4, js, first);
    R(ADD, "v_add_u32") R(SUB, "v_sub_u32") R(XOR, "v_xor_b32") R(ASHR, "v_ashrrev_i32") R(LSHR_V, "v_lshrrev_b32") R(MOV, "v_mov_b32")
    R(ADD_LIT, "v_add_u32_lit") R(SAD, "v_sad_u32") R(ADD3, "v_add3_u32") R(OR3, "v_or3_b32") R(BITOP3, "v_bitop3_b32")
    R(MAD_I64, "v_mad_i64_i32") R(MAD_I64_CHAIN, "mad_i64_chain") R(MIX_FIR, "fir_mix_16") R(MAD_I24, "v_mad_i32_i24")
    R(LSHL_ADD_U64, "v_lshl_add_u64") R(ASHR_I64, "v_ashrrev_i64") R(ALIGNBIT, "v_alignbit_b32") R(CNDMASK_S, "v_cndmask_b32")
    R(MOV_DPP, "v_mov_b32_dpp") R(ADD_DPP, "v_add_u32_dpp") R(READLANE, "v_readlane_b32") R(MUL_F64, "v_mul_f64") R(ADD_F64, "v_add_f64")
    R(FMA_F64, "v_fma_f64") R(CVT_F64_I32, "v_cvt_f64_i32") R(MAX_I32, "v_max_i32") R(MUL_LO, "v_mul_lo_u32") R(PERM, "v_perm_b32")
    R(SNOP, "s_nop") R(DS_READ_B128, "ds_read_b128")
    R(AND, "v_and_b32") R(OR, "v_or_b32") R(LSHL, "v_lshlrev_b32") R(NOT, "v_not_b32") R(CMP_VCC, "v_cmp_vcc") R(CMP_SGPR, "v_cmp_sgpr")
    R(ADD_CO, "v_add_co_u32") R(ADDC, "v_addc_co_u32") R(SUB_CO, "v_sub_co_u32") R(SUBREV, "v_subrev_u32") R(MIN_U32, "v_min_u32")
    R(MUL_I24, "v_mul_i32_i24") R(ADD_E64_S, "v_add_u32_e64_sgpr") R(BFE_U32, "v_bfe_u32") R(LSHL_ADD, "v_lshl_add_u32")
    R(AND_OR, "v_and_or_b32") R(MOV_B64, "v_mov_b64") R(READFIRST, "v_readfirstlane_b32") R(WRITELANE, "v_writelane_b32")
    R(LDEXP_F64, "v_ldexp_f64") R(MIX_AB, "mix_add_sad_1to1") R(MIX_AAB, "mix_add_sad_2to1") R(MIX_STATS, "mix_stats_9fast_7sad")
    fprintf(js, "\n]}\n");
    fclose(js);
    return 0;
}
