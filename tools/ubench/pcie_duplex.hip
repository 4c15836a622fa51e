// pcie_duplex.hip -- what the host link carries in both directions at once, by WHO moves the bytes: the copy engines
// (hipMemcpyAsync on two streams) or kernels that load from / store to pinned host memory themselves.
// tools/pcie_probe.py (copy engines only) sees 57 GB/s per direction but also only 57 GB/s for both together; the
// pipelined encode path (frames stored by k_frame64 into pinned memory while the next batch is uploaded) moves more.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_duplex.hip -o tools/ubench/bin/pcie_duplex && tools/ubench/bin/pcie_duplex
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const size_t n = 256u << 20;
    void *h_up, *h_down, *d_up, *d_down;
    CK(hipHostMalloc(&h_up, n, hipHostMallocDefault));
    CK(hipHostMalloc(&h_down, n, hipHostMallocDefault));
    CK(hipMalloc(&d_up, n));
    CK(hipMalloc(&d_down, n));
    memset(h_up, 1, n);
    CK(hipMemset(d_down, 2, n));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto run = [&](int up, int down, int grid) {   // 0: off, 1: copy engine, 2: kernel
        double best = 0;
        for (int it = 0; it < 4; it++) {
            CK(hipDeviceSynchronize());
            auto t = std::chrono::steady_clock::now();
            if (up == 1) CK(hipMemcpyAsync(d_up, h_up, n, hipMemcpyHostToDevice, s0));
            if (up == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s0, (const uint4 *)h_up, (uint4 *)d_up, n / 16);
            if (down == 1) CK(hipMemcpyAsync(h_down, d_down, n, hipMemcpyDeviceToHost, s1));
            if (down == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s1, (const uint4 *)d_down, (uint4 *)h_down, n / 16);
            CK(hipDeviceSynchronize());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
            const double gbs = (double)n * ((up != 0) + (down != 0)) / dt / 1e9;
            if (gbs > best) best = gbs;
        }
        return best;
    };
    const char *who[3] = {"-", "engine", "kernel"};
    printf("{\"bytes_each_way\": %zu, \"runs\": [\n", n);
    bool first = true;
    for (int grid : {256, 1024})
        for (int up = 0; up < 3; up++)
            for (int down = 0; down < 3; down++) {
                if (!up && !down) continue;
                if (grid != 256 && up != 2 && down != 2) continue;
                printf("%s {\"up\": \"%s\", \"down\": \"%s\", \"kernel_grid\": %d, \"sum_GB/s\": %.1f}", first ? "" : ",\n", who[up], who[down], grid,
                       run(up, down, grid));
                first = false;
            }
    printf("\n]}\n");
    return 0;
}
