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

constexpr int STEM_ROWS = 4;

template <int CIN>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, long sn, long sy, long sx, long sc, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out, long ldo, int H, int W, int Cout,
                                                        float* __restrict__ gn_part) {
    // gn_part (optional, W % 32 == 0, W <= 128): per (32-pixel chunk, channel) sum and sum of squares of the result, [N H W / 32][Cout][2] —
    // the first ResBlock's GroupNorm (and the last decoder block's, which reads this tensor as its skip input) then need no statistics pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 9 * CIN;
    const int cg = Cout >> 2;                          // channel groups of 4
    float* const wl = smem;                             // [K][Cout + 4]   (k = tap * CIN + ci)
    const int WP = Cout + 4;                            // weight row pitch: the transposing writes below go to banks 4 k + co instead of all to bank co
    float* const in = smem + K * WP;                    // [3][W + 2][CIN], zero padded
    // A block owns STEM_ROWS consecutive output rows of one image (round 4; one row before): the weight transposition into LDS — 9 CIN
    // Cout scattered 4-byte writes, as long as a row's arithmetic — is paid once per block instead of once per row.
    const int rblocks = (H + STEM_ROWS - 1) / STEM_ROWS;
    const int n = blockIdx.x / rblocks, y0 = (blockIdx.x - n * rblocks) * STEM_ROWS, tid = threadIdx.x;
    for (int i = tid; i < K * Cout; i += 256) {         // OHWI weight [co][tap][ci] -> [k][co]
        const int co = i / K, k = i - co * K;
        wl[k * WP + co] = w[i];
    }
    for (int y = y0; y < y0 + STEM_ROWS && y < H; ++y) {
    if (y > y0) __syncthreads();                        // everyone is done with the previous row's inputs (and its reduction buffer)
    for (int i = tid; i < 3 * (W + 2) * CIN; i += 256) {
        const int ci = i % CIN, xx = (i / CIN) % (W + 2) - 1, yy = y + i / (CIN * (W + 2)) - 1;
        in[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[n * sn + yy * sy + xx * sx + ci * sc] : 0.f;
    }
    __syncthreads();
    const int ppp = 256 / cg;                           // pixels per pass
    const int g = tid % cg, pl = tid / cg;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && pl < ppp) b4 = *reinterpret_cast<const float4*>(bias + 4 * g);
    float gs[4][4], gq[4][4];                           // [chunk of the row][channel of the group]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[c][e] = 0.f; gq[c][e] = 0.f; }
    for (int px = pl; px < W && pl < ppp; px += ppp) {
        float4 acc = b4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float v = in[(ky * (W + 2) + px + kx) * CIN + ci];
                    const float4 ww = *reinterpret_cast<const float4*>(wl + ((ky * 3 + kx) * CIN + ci) * WP + 4 * g);
                    acc.x = fmaf(v, ww.x, acc.x); acc.y = fmaf(v, ww.y, acc.y); acc.z = fmaf(v, ww.z, acc.z); acc.w = fmaf(v, ww.w, acc.w);
                }
        *reinterpret_cast<float4*>(out + ((long)(n * H + y) * W + px) * ldo + 4 * g) = acc;
        if (gn_part) {
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((px >> 5) == c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gs[c][e] += v[e]; gq[c][e] += v[e] * v[e]; }
                }
        }
    }
    if (!gn_part) continue;
    // the ppp pixel lanes of a channel group, added in a fixed order through LDS (behind the weights and the input rows)
    float* const red = in + 3 * (W + 2) * CIN;          // [chunk][pl][Cout][2]
    const int nch = W >> 5;
    if (pl < ppp) {
        for (int c = 0; c < nch; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s_ = 0.f, q_ = 0.f;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) if (cc == c) { s_ = gs[cc][e]; q_ = gq[cc][e]; }
                red[((c * ppp + pl) * Cout + 4 * g + e) * 2] = s_;
                red[((c * ppp + pl) * Cout + 4 * g + e) * 2 + 1] = q_;
            }
    }
    __syncthreads();
    for (int i = tid; i < nch * Cout; i += 256) {
        const int c = i / Cout, co = i - c * Cout;
        float s_ = 0.f, q_ = 0.f;
        for (int l = 0; l < ppp; ++l) { s_ += red[((c * ppp + l) * Cout + co) * 2]; q_ += red[((c * ppp + l) * Cout + co) * 2 + 1]; }
        float* o = gn_part + (((long)(n * H + y) * nch + c) * Cout + co) * 2;
        o[0] = s_; o[1] = q_;
    }
    }
}

}  // namespace

extern "C" int cdae_conv3x3_stem_supported(int Cin, int Cout, int W) {
    return Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && Cout >= 4 && Cout <= 1024 && W >= 1 && (size_t)(9 * Cin * (Cout + 4) + 3 * (W + 2) * Cin) * 4 <= 160 * 1024;
}

extern "C" int cdae_conv3x3_stem(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* bias, float* out, long ldo,
                                 int N, int H, int W, int Cin, int Cout, void* stream) {
    return cdae_conv3x3_stem_gn(x, sn, sy, sx, sc, w, bias, out, ldo, nullptr, N, H, W, Cin, Cout, stream);
}

// + gn_part (may be NULL): [N H W / 32][Cout][2] partial sums of the result for the next GroupNorm (cdae_gn_stats_from_parts); W % 32 == 0, W <= 128
extern "C" int cdae_conv3x3_stem_gn(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* bias, float* out, long ldo,
                                    float* gn_part, int N, int H, int W, int Cin, int Cout, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!cdae_conv3x3_stem_supported(Cin, Cout, W) || ldo % 4 || (((size_t)out | (size_t)bias) & 15))
        return cdae_fail("conv3x3_stem: 1..4 input channels, Cout % 4 == 0, 16-byte aligned output rows required");
    size_t smem = (size_t)(9 * Cin * (Cout + 4) + 3 * (W + 2) * Cin) * sizeof(float);
    if (gn_part) {
        if (W % 32 || W > 128) return cdae_fail("conv3x3_stem_gn: partial sums need W % 32 == 0 and W <= 128");
        smem += (size_t)(W / 32) * (256 / (Cout / 4)) * Cout * 2 * sizeof(float);
        if (smem > 160 * 1024) return cdae_fail("conv3x3_stem_gn: LDS budget exceeded");
    }
    cdae_prof_begin(PROF_IGEMM, 2.0 * N * H * W * 9.0 * Cin * Cout, st);
#define STEM(C) do { \
        static size_t attr = 0; \
        if (smem > attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed"); attr = smem; } \
        hipLaunchKernelGGL(stem_conv_kernel<C>, dim3(N * ((H + STEM_ROWS - 1) / STEM_ROWS)), dim3(256), smem, st, x, sn, sy, sx, sc, w, bias, out, ldo, H, W, Cout, gn_part); } while (0)
    switch (Cin) { case 1: STEM(1); break; case 2: STEM(2); break; case 3: STEM(3); break; default: STEM(4); }
#undef STEM
    cdae_prof_end(PROF_IGEMM, st);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("stem_conv launch failed");
}
