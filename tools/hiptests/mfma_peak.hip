// Dev microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate on random operands (the practical ceiling of the f16x3 kernels).
// hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int CHAIN>
__global__ void k(const half8* __restrict__ in, float* out, int iters) {
    half8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], b0 = in[threadIdx.x + 128], b1 = in[threadIdx.x + 192];
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        if (CHAIN) {      // three dependent MFMAs per accumulator, like the f16x3 product
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c1, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c2, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c3, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c3, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c3, 0, 0, 0);
        } else {
            for (int r = 0; r < 3; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c3, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same flops per iteration on v_mfma_f32_16x16x32_f16 (24 MFMAs of half the size): eight accumulators, three dependent MFMAs each
__global__ void k16(const half8* __restrict__ in, float* out, int iters) {
    half8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], b0 = in[threadIdx.x + 128], b1 = in[threadIdx.x + 192];
    f32x4 c[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[q], 0, 0, 0);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[q], 0, 0, 0);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[q], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) s += c[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    half8* in; float* out;
    hipMalloc(&in, 256 * sizeof(half8)); hipMalloc(&out, 4096 * 1024 * sizeof(float));
    _Float16 h[256 * 8];
    for (int i = 0; i < 256 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int chain = 0; chain < 2; ++chain)
        for (int wps = 1; wps <= 4; wps *= 2) {        // waves per SIMD
            const int blocks = 256 * 4 * wps, iters = 4000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (chain) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * iters * 12 * 2.0 * 32 * 32 * 16;
            printf("chain=%d waves/SIMD=%d: %.3f ms  %.0f TFLOP/s f16 MFMA (%.0f TF f16x3-equivalent)\n", chain, wps, ms, fl / ms / 1e9, fl / ms / 3e9);
        }
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * 4 * wps, iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k16, dim3(blocks), dim3(64), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * iters * 24 * 2.0 * 16 * 16 * 32;
        printf("16x16x32 chain=1 waves/SIMD=%d: %.3f ms  %.0f TFLOP/s f16 MFMA (%.0f TF f16x3-equivalent)\n", wps, ms, fl / ms / 1e9, fl / ms / 3e9);
    }
    return 0;
}
