// Test helper (tests/test_gpu_lds_poison.py), not part of the product: fills the LDS of EVERY compute unit with a 64-bit pattern.  LDS is not cleared
// between kernels, so whatever the next kernel reads before writing it is this pattern (a signalling NaN, or 1e300).
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void poison_kernel(unsigned long long pattern, int n, unsigned long long* sink) {
    extern __shared__ unsigned long long lds[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // read back so that the stores cannot be dropped; a workgroup stays long enough for the dispatcher to place the others on the remaining units
    unsigned long long acc = 0;
    for (int r = 0; r < 8; r++)
        for (int i = threadIdx.x; i < n; i += blockDim.x) acc += lds[i] ^ (unsigned long long)r;
    if (acc == 0x1234567ull) sink[0] = acc;
}

// bytes: LDS per workgroup (the whole unit: 160 KB on gfx950); workgroups: several times the unit count, one resident per unit at a time
extern "C" int poison_lds(unsigned long long pattern, int bytes, int workgroups) {
    static unsigned long long* sink = nullptr;
    hipError_t e = hipSuccess;
    if (!sink && (e = hipMalloc((void**)&sink, 8)) != hipSuccess) return 1000 + (int)e;
    if ((e = hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return 2000 + (int)e;
    hipLaunchKernelGGL(poison_kernel, dim3(workgroups), dim3(256), bytes, 0, pattern, bytes / 8, sink);
    if ((e = hipGetLastError()) != hipSuccess) return 3000 + (int)e;
    return (e = hipDeviceSynchronize()) == hipSuccess ? 0 : 4000 + (int)e;
}

// what a kernel that reads LDS without writing it sees: word (17 i mod n) of every workgroup's allocation, for 256 threads -> out[workgroups][256] (host)
__global__ __launch_bounds__(256) void peek_kernel(int n, unsigned long long* out) {
    extern __shared__ unsigned long long lds[];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = lds[(17 * threadIdx.x + 3 * blockIdx.x) % n];
}
extern "C" int peek_lds(int bytes, int workgroups, unsigned long long* host_out) {
    unsigned long long* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, (size_t)workgroups * 256 * 8);
    if (e != hipSuccess) return 1000 + (int)e;
    if ((e = hipFuncSetAttribute((const void*)peek_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return 2000 + (int)e;
    hipLaunchKernelGGL(peek_kernel, dim3(workgroups), dim3(256), bytes, 0, bytes / 8, d);
    if ((e = hipGetLastError()) != hipSuccess) return 3000 + (int)e;
    e = hipMemcpy(host_out, d, (size_t)workgroups * 256 * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? 0 : 4000 + (int)e;
}
