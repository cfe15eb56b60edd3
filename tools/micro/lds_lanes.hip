// microbenchmark (diagnostic, not shipped): what does a ds_read_b64 cost a CU's LDS when only some lanes of the wavefront are active?
// one wavefront per workgroup, WPC workgroups per CU (bounded by dynamic LDS), every wavefront streams ds_read_b64 from its own image.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(64) void k(double* out, int iters, unsigned long long mask, int stride, unsigned long long* cyc) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    for (int e = lane; e < 2048; e += 64) lds[e] = e * 0.5;
    __syncthreads();
    double acc[16];
    for (int u = 0; u < 16; u++) acc[u] = 0.0;
    const bool on = (mask >> lane) & 1ull;
    const int base = (lane * stride) & 1023;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (on) {
        for (int i = 0; i < iters; i++) {
            const int o = (base + i) & 1023;
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u] += lds[o + 8 * u + (u & 1) * 512];
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double a = 0.0;
    for (int u = 0; u < 16; u++) a += acc[u];
    if (on) out[blockIdx.x * 64 + lane] = a;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 4;
    const int grid = 256 * wpc, iters = 2000;
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, grid * 64 * 8); (void)hipMalloc(&cyc, grid * 8);
    const size_t lds = 160 * 1024 / wpc - 1024;      // exactly wpc workgroups per CU
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    struct { const char* name; unsigned long long mask; int stride; } cases[] = {
        {"64 lanes, stride 25", ~0ull, 25}, {"32 lanes (0..31)", 0xffffffffull, 25}, {"16 lanes (0..15)", 0xffffull, 25}, {"8 lanes (0..7)", 0xffull, 25},
        {"16 lanes (every 4th)", 0x1111111111111111ull, 25}, {"12 lanes (0-5, 32-37)", 0x3f0000003full, 25}, {"64 lanes same address", ~0ull, 0},
        {"6 lanes (0-5)", 0x3full, 25}, {"24 lanes (6 of each 16)", 0x003f003f003f003full, 25}};
    unsigned long long* h = (unsigned long long*)malloc(grid * 8);
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, 0, out, iters, c.mask, c.stride, cyc);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; i++) s += h[i];
        printf("%-28s wpc %d: %.2f cycles per ds_read_b64 per wavefront (%.2f per CU-instruction)\n", c.name, wpc, s / grid / (iters * 16.0), s / grid / (iters * 16.0) / wpc);
    }
    return 0;
}
