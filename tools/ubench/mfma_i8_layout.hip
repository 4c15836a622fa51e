// mfma_i8_layout.hip -- which (lane, byte) holds A[row][k] / B[k][col] of v_mfma_i32_16x16x64_i8 on gfx950?
// Tests the hypotheses with exact integer data (asymmetric A and B) against a CPU product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k_one(const v4i *a, const v4i *b, v4i *d) {
    const int l = threadIdx.x;
    v4i c = {0, 0, 0, 0};
    d[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], c, 0, 0, 0);
}
static int kmap(int hyp, int q, int j) {   // k index of byte j of lane group q
    if (hyp == 0) return 16 * q + j;                        // contiguous 16
    return (j < 8 ? 0 : 32) + 8 * q + (j & 7);              // two 8-byte halves (two 16x16x32 steps)
}
int main() {
    int8_t A[16][64], B[64][16];
    for (int r = 0; r < 16; r++) for (int k = 0; k < 64; k++) A[r][k] = (int8_t)(((r * 7 + k * 3) % 29) - 14);
    for (int k = 0; k < 64; k++) for (int n = 0; n < 16; n++) B[k][n] = (int8_t)(((k * 5 + n * 11) % 13) - 6);
    int ref[16][16];
    for (int r = 0; r < 16; r++) for (int n = 0; n < 16; n++) { int s = 0; for (int k = 0; k < 64; k++) s += A[r][k] * B[k][n]; ref[r][n] = s; }
    v4i *da, *db, *dd;
    hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 16); hipMalloc(&dd, 64 * 16);
    for (int hyp = 0; hyp < 2; hyp++) {
        int8_t ha[64][16], hb[64][16];
        for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            const int k = kmap(hyp, l >> 4, j);
            ha[l][j] = A[l & 15][k];
            hb[l][j] = B[k][l & 15];
        }
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
        hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dd);
        int out[64][4];
        hipMemcpy(out, dd, sizeof out, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) if (out[l][v] != ref[4 * (l >> 4) + v][l & 15]) bad++;
        printf("hypothesis %d (%s): %d of 256 outputs differ\n", hyp, hyp == 0 ? "k = 16 q + j" : "k = 32 (j >> 3) + 8 q + (j & 7)", bad);
    }
    return 0;
}
