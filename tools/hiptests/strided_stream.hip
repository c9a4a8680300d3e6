// Probe: HBM rate of a read + write stream by access pattern (hipcc --offload-arch=gfx950 -O3 strided_stream.hip -o strided_stream).
// A [M][K] fp32 tensor is read once and [M][K] f16x2 (4 B per element) written once, like the ResBlock entry sweep.
//   pattern 0: a block owns 128 rows and walks K in 32-float steps: every step touches 128 rows x 128 B (skipgn_kernel's pattern)
//   pattern 1: the same, 64-float steps (128 rows x 256 B)
//   pattern 2: a block owns 128 rows and reads them row by row, whole rows (K floats) at a time
//   pattern 3: flat grid-stride copy (16 B per thread, fully contiguous waves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int STEPF>
__global__ __launch_bounds__(256, 2) void tile_steps(const float* __restrict__ x, float* __restrict__ y, int M, int K) {
    const int m0 = blockIdx.x * 128, tid = threadIdx.x;
    constexpr int LPR = STEPF / 4;                  // lanes per row per step (16 B each)
    constexpr int RPP = 256 / LPR;                  // rows per pass
    const int c = tid % LPR, r0 = tid / LPR;
    for (int k = 0; k < K; k += STEPF)
#pragma unroll
        for (int r = r0; r < 128; r += RPP) {
            const long o = (long)(m0 + r) * K + k + 4 * c;
            const float4 v = *reinterpret_cast<const float4*>(x + o);
            *reinterpret_cast<float4*>(y + o) = make_float4(v.x + 1.f, v.y, v.z, v.w);
        }
}
__global__ __launch_bounds__(256, 2) void tile_rows(const float* __restrict__ x, float* __restrict__ y, int M, int K) {
    const int m0 = blockIdx.x * 128, tid = threadIdx.x;
    const int per = K / 4;                          // float4 per row
    for (int i = tid; i < 128 * per; i += 256) {
        const long o = (long)m0 * K + 4L * i;
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        *reinterpret_cast<float4*>(y + o) = make_float4(v.x + 1.f, v.y, v.z, v.w);
    }
}
__global__ void flat(const float4* __restrict__ x, float4* __restrict__ y, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        y[i] = make_float4(v.x + 1.f, v.y, v.z, v.w);
    }
}
int main() {
    const int M = 128 * 4096, K = 256;
    const size_t bytes = (size_t)M * K * 4;
    float *x, *y;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes);
    hipMemset(x, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(tile_steps<32>, dim3(M / 128), dim3(256), 0, 0, x, y, M, K);
            else if (pat == 1) hipLaunchKernelGGL(tile_steps<64>, dim3(M / 128), dim3(256), 0, 0, x, y, M, K);
            else if (pat == 2) hipLaunchKernelGGL(tile_rows, dim3(M / 128), dim3(256), 0, 0, x, y, M, K);
            else hipLaunchKernelGGL(flat, dim3(4096), dim3(256), 0, 0, (const float4*)x, (float4*)y, (long)M * K / 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("pattern %d: %.1f us  %.2f TB/s (read + write, %zu MB each)\n", pat, best * 1e3, 2.0 * bytes / (best * 1e-3) / 1e12, bytes >> 20);
    }
    return 0;
}
