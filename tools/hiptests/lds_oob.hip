// Dev probe: what does a ds_read_b128 return for an address beyond the block's LDS allocation?  (convwin.hip redirects the
// fragment reads of padding taps there instead of masking the data, if and only if this prints all zeros.)
// hipcc --offload-arch=gfx950 -O3 lds_oob.hip -o lds_oob && ./lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < lds_bytes / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0xdead0000u + blockIdx.x;
    __syncthreads();
    // give the second block on this CU time to fill its own allocation
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
    const unsigned offs[8] = {(unsigned)lds_bytes, 0x3C000u, 0x3C000u + 0x3E00u, 0x100000u, 0x40000u, 0x7FFFFFF0u, 0x80000u + 4096u, (unsigned)lds_bytes - 16u};
    for (int q = 0; q < 8; ++q) {
        unsigned a = offs[q] + 0 * threadIdx.x;
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if (threadIdx.x == 0) { unsigned* o = out + (blockIdx.x * 8 + q) * 4; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
    }
}
int main() {
    unsigned* out; const int nb = 1024;
    hipMalloc(&out, nb * 8 * 16); hipMemset(out, 0xff, nb * 8 * 16);
    const int lds_bytes = 81920;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), lds_bytes, 0, out, lds_bytes);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    static unsigned h[nb * 8 * 4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"size", "0x3C000", "0x3FE00", "0x100000", "0x40000", "0x7FFFFFF0", "0x81000", "size-16 (in range)"};
    for (int q = 0; q < 8; ++q) {
        int nz = 0; unsigned ex = 0;
        for (int b = 0; b < nb; ++b) for (int c = 0; c < 4; ++c) if (h[(b * 8 + q) * 4 + c]) { ++nz; ex = h[(b * 8 + q) * 4 + c]; }
        printf("offset %-20s: %d of %d dwords nonzero (example 0x%08x)\n", names[q], nz, nb * 4, ex);
    }
    return 0;
}
