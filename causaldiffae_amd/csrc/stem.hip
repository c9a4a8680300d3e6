// stem.hip — the UNet's input convolution (reference unet.py:395-399: conv_nd(dims, in_channels, model_channels, 3, padding=1) on the
// 1-, 3- or 4-channel image), gfx950 only.
//
// As an implicit GEMM its reduction is 9 * Cin <= 36 deep: the MFMA tile kernels spend their time in the scalar gather of a 4-wide
// channel axis (230 us per DDIM step at batch 128).  The op is a pure streaming WRITE of [pixels][Cout] — 268 MB — with 36 FMAs per
// output, so it runs on the vector ALUs in exact fp32: a block owns one output image row; the three input rows (+ zero padding) and the
// weights live in LDS; a thread computes four consecutive output channels of a pixel (weights read as conflict-free float4s across
// the 32 channel groups of a wave half, the 36 inputs as LDS broadcasts) and stores one float4.
#include <hip/hip_runtime.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

namespace {

template <int CIN>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, long sn, long sy, long sx, long sc, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, long ldo, int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 9 * CIN;
    const int cg = Cout >> 2;                          // channel groups of 4
    float* const wl = smem;                             // [K][Cout]   (k = tap * CIN + ci)
    float* const in = smem + K * Cout;                  // [3][W + 2][CIN], zero padded
    const int n = blockIdx.x / H, y = blockIdx.x - n * H, tid = threadIdx.x;
    for (int i = tid; i < K * Cout; i += 256) {         // OHWI weight [co][tap][ci] -> [k][co]
        const int co = i / K, k = i - co * K;
        wl[k * Cout + co] = w[i];
    }
    for (int i = tid; i < 3 * (W + 2) * CIN; i += 256) {
        const int ci = i % CIN, xx = (i / CIN) % (W + 2) - 1, yy = y + i / (CIN * (W + 2)) - 1;
        in[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[n * sn + yy * sy + xx * sx + ci * sc] : 0.f;
    }
    __syncthreads();
    const int ppp = 256 / cg;                           // pixels per pass
    const int g = tid % cg, pl = tid / cg;
    if (pl >= ppp) return;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + 4 * g);
    for (int px = pl; px < W; px += ppp) {
        float4 acc = b4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float v = in[(ky * (W + 2) + px + kx) * CIN + ci];
                    const float4 ww = *reinterpret_cast<const float4*>(wl + ((ky * 3 + kx) * CIN + ci) * Cout + 4 * g);
                    acc.x = fmaf(v, ww.x, acc.x); acc.y = fmaf(v, ww.y, acc.y); acc.z = fmaf(v, ww.z, acc.z); acc.w = fmaf(v, ww.w, acc.w);
                }
        *reinterpret_cast<float4*>(out + ((long)(n * H + y) * W + px) * ldo + 4 * g) = acc;
    }
}

}  // namespace

extern "C" int cdae_conv3x3_stem_supported(int Cin, int Cout, int W) {
    return Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && Cout >= 4 && Cout <= 1024 && W >= 1 && (size_t)(9 * Cin * Cout + 3 * (W + 2) * Cin) * 4 <= 160 * 1024;
}

extern "C" int cdae_conv3x3_stem(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* bias, float* out, long ldo,
                                 int N, int H, int W, int Cin, int Cout, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!cdae_conv3x3_stem_supported(Cin, Cout, W) || ldo % 4 || (((size_t)out | (size_t)bias) & 15))
        return cdae_fail("conv3x3_stem: 1..4 input channels, Cout % 4 == 0, 16-byte aligned output rows required");
    const size_t smem = (size_t)(9 * Cin * Cout + 3 * (W + 2) * Cin) * sizeof(float);
    cdae_prof_begin(PROF_IGEMM, 2.0 * N * H * W * 9.0 * Cin * Cout, st);
#define STEM(C) do { \
        static size_t attr = 0; \
        if (smem > attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed"); attr = smem; } \
        hipLaunchKernelGGL(stem_conv_kernel<C>, dim3(N * H), dim3(256), smem, st, x, sn, sy, sx, sc, w, bias, out, ldo, H, W, Cout); } while (0)
    switch (Cin) { case 1: STEM(1); break; case 2: STEM(2); break; case 3: STEM(3); break; default: STEM(4); }
#undef STEM
    cdae_prof_end(PROF_IGEMM, st);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("stem_conv launch failed");
}
