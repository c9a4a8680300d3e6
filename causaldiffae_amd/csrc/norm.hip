// norm.hip — GroupNorm32 (+scale-shift, +SiLU), BatchNorm(+LeakyReLU) and row softmax for NHWC fp32
// activations on gfx950.  All HBM-bound: float4 loads, per-thread register accumulation with a fixed
// channel per thread (coalesced rows), LDS only for the final cross-thread combine, wave shuffles for
// the softmax row reductions.  Statistics are accumulated relative to a pivot sample of the same
// group/channel so that E[x^2]-E[x]^2 never cancels catastrophically; partials are combined in f64.
//
// Reference semantics: GroupNorm32 nn.py:435-437,541-548 (32 groups, eps 1e-5, fp32 statistics),
// ResBlock scale-shift unet.py:190-194, SiLU nn.py:430-432, BatchNorm2d+LeakyReLU nn.py:46-53,
// softmax over keys unet.py:252.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

#ifndef GN_U16
#define GN_U16 4      // loads in flight per thread on bf16 rows (8 was measured: 61 -> 75 us on a 256-channel backward, and deeper dx pipelining 61 -> 70)
#endif

namespace {

__device__ __forceinline__ float silu_f(float v) { return cdae_silu(v); }
// On bf16 rows (the 16-bit torso) the result is rounded to 8 significand bits: the short form of the logistic function, 1 / (1 + 2^(-x log2 e)) —
// four instructions instead of cdae_sigmoid's eight (which carries the rounding error of the exponent into a 2-ulp result so that fp32
// producers agree bit for bit).  2^t overflows to inf for x < -88 and the reciprocal of inf is 0: no clamp needed for finite x.
// The bf16 GroupNorm kernels are VALU-bound (two transcendentals + ~30 fp32 operations per element against 2-byte loads).
template <typename T>
__device__ __forceinline__ float sigmoid_t(float x) {
    if constexpr (sizeof(T) == 2) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269502162933349609375f));
    else return cdae_sigmoid(x);
}
template <typename T>
__device__ __forceinline__ float silu_t(float v) {
    if constexpr (sizeof(T) == 2) return v * sigmoid_t<T>(v);
    else return cdae_silu(v);
}

// Storage type of an activation / gradient tensor: float (the parity modes) or __bf16 (the half-precision torso: the reference's
// convert_to_fp16 placement, unet.py:501-507 — 16-bit activations between the layers, fp32 statistics inside GroupNorm32, nn.py:435-437).
// Every kernel below computes in fp32 registers; only the loads and stores differ.
typedef __bf16 gn_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldv4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldv4(const __bf16* p) {
    const gn_bf16x4 v = *reinterpret_cast<const gn_bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void stv4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void stv4(__bf16* p, const float4& v) {
    gn_bf16x4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<gn_bf16x4*>(p) = o;
}
__device__ __forceinline__ float ldv1(const float* p) { return *p; }
__device__ __forceinline__ float ldv1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void stv1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv1(__bf16* p, float v) { *p = (__bf16)v; }

// ------------------------------------------------------------------ GroupNorm statistics
// grid (nchunk, N), 256 threads.  partial[((n*nchunk + chunk)*G + g)*2 + {0,1}] = (sum, sumsq) of (x - pivot_g)
// Two-source form (x2 != nullptr): channels [0, C1) come from x (pitch ldx), channels [C1, C) from x2 (pitch ld2) — the UNet's
// skip concatenation th.cat([h, hs.pop()], dim=1) (unet.py:629) read in place instead of being copied into one tensor.
template <typename T>
__device__ __forceinline__ const T* gn_src(const T* x, int ldx, const T* x2, int ld2, int C1, long npix0, int c, int& ld) {
    if (x2 && c >= C1) { ld = ld2; return x2 + npix0 * ld2 + (c - C1); }
    ld = ldx;
    return x + npix0 * ldx + c;
}
template <int VEC, typename T = float>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int HW, int C, int ldx, int cpg, int G,
                                                          int pix_per_block, float* __restrict__ partial,
                                                          const T* __restrict__ x2 = nullptr, int ld2 = 0, int C1 = 0) {
    __shared__ float sS[256], sQ[256];
    const int n = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int E = C / VEC;
    const int rows = 256 / E;                  // host guarantees E <= 256
    const int tid = threadIdx.x;
    const int r = tid / E, e = tid - r * E;
    const bool active = r < rows;
    float S = 0.f, Q = 0.f;
    if (active) {
        const int g = (e * VEC) / cpg;
        int ld, ldp;
        const T* xe = gn_src(x, ldx, x2, ld2, C1, (long)n * HW, e * VEC, ld);         // this thread's channel vector at pixel 0
        const float pivot = ldv1(gn_src(x, ldx, x2, ld2, C1, (long)n * HW, g * cpg, ldp));
        const int p0 = chunk * pix_per_block, p1 = min(HW, p0 + pix_per_block);
        int p = p0 + r;
        if (VEC == 4) {
            auto acc4 = [&](const float4& v) {
                float a = v.x - pivot, b = v.y - pivot, c = v.z - pivot, d = v.w - pivot;
                S += (a + b) + (c + d);
                Q += (a * a + b * b) + (c * c + d * d);
            };
            constexpr int U = sizeof(T) == 2 ? GN_U16 : 4;         // independent loads in flight (16 bytes each in fp32, 8 bytes in bf16: twice as many)
            for (; p + (U - 1) * rows < p1; p += U * rows) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ldv4(xe + (long)(p + u * rows) * ld);
#pragma unroll
                for (int u = 0; u < U; ++u) acc4(v[u]);
            }
            for (; p < p1; p += rows) acc4(ldv4(xe + (long)p * ld));
        } else {
            for (; p < p1; p += rows) { float a = ldv1(xe + (long)p * ld) - pivot; S += a; Q += a * a; }
        }
    }
    sS[tid] = S; sQ[tid] = Q;
    __syncthreads();
    if (tid < G) {
        const int e0 = tid * cpg / VEC, e1 = (tid + 1) * cpg / VEC;
        float s = 0.f, q = 0.f;
        for (int rr = 0; rr < rows; ++rr)
            for (int ee = e0; ee < e1; ++ee) { s += sS[rr * E + ee]; q += sQ[rr * E + ee]; }
        float* out = partial + (((long)n * nchunk + chunk) * G + tid) * 2;
        out[0] = s; out[1] = q;
    }
}

// (a, b) of y = x * a + b for one (image, channel): GroupNorm + affine (+ scale-shift) folded — THE definition, used by gn_coef_kernel and
// by the statistics kernels that emit the table themselves (one launch less per GroupNorm whose consumer streams the fp32 rows)
__device__ __forceinline__ void gn_coef_one(float mu, float rs, float gm, float bt, const float* __restrict__ ss, long ss_row, int C, int c,
                                            float* __restrict__ out) {
    float a = rs * gm;
    float b = fmaf(-mu, a, bt);
    if (ss) {
        const float sc = 1.f + ss[ss_row + c], sh = ss[ss_row + C + c];
        a *= sc;
        b = fmaf(b, sc, sh);
    }
    out[0] = a; out[1] = b;
}

// G == 32 (GroupNorm32, every norm of the model): grid N, 256 threads = 8 chunk lanes x 32 groups; the serial per-thread walk over
// up to 128 chunks below costs ~10 us of pure latency per launch, 56 launches per network pass
template <typename T = float>
__global__ __launch_bounds__(256) void gn_finalize32_kernel(const T* __restrict__ x, int HW, int ldx, int cpg, int nchunk, float eps,
                                                            const float* __restrict__ partial, float* __restrict__ mean, float* __restrict__ rstd,
                                                            const T* __restrict__ x2, int ld2, int C1,
                                                            const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                                            const float* __restrict__ ss = nullptr, int ld_ss = 0, int C = 0,
                                                            float* __restrict__ coef = nullptr) {
    __shared__ double sS[8][32], sQ[8][32];
    const int n = blockIdx.x, g = threadIdx.x & 31, cl = threadIdx.x >> 5;
    double s = 0.0, q = 0.0;
    {
        int c = cl;
        for (; c + 24 < nchunk; c += 32) {          // four chunk records in flight; same order of additions
            const float2 v0 = *reinterpret_cast<const float2*>(partial + (((long)n * nchunk + c) * 32 + g) * 2), v1 = *reinterpret_cast<const float2*>(partial + (((long)n * nchunk + c + 8) * 32 + g) * 2),
                         v2 = *reinterpret_cast<const float2*>(partial + (((long)n * nchunk + c + 16) * 32 + g) * 2), v3 = *reinterpret_cast<const float2*>(partial + (((long)n * nchunk + c + 24) * 32 + g) * 2);
            s += v0.x; q += v0.y; s += v1.x; q += v1.y; s += v2.x; q += v2.y; s += v3.x; q += v3.y;
        }
        for (; c < nchunk; c += 8) {
            const float2 v = *reinterpret_cast<const float2*>(partial + (((long)n * nchunk + c) * 32 + g) * 2);
            s += v.x; q += v.y;
        }
    }
    sS[cl][g] = s; sQ[cl][g] = q;
    __syncthreads();
    if (cl == 0) {
        s = 0.0; q = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += sS[k][g]; q += sQ[k][g]; }
        const double cnt = (double)HW * cpg;
        int ldp;
        const double pivot = ldv1(gn_src(x, ldx, x2, ld2, C1, (long)n * HW, g * cpg, ldp));
        const double m = s / cnt;
        double var = q / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float mu = (float)(pivot + m), rs = (float)(1.0 / sqrt(var + (double)eps));
        mean[n * 32 + g] = mu;
        rstd[n * 32 + g] = rs;
        if (coef)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) gn_coef_one(mu, rs, gamma[c], beta[c], ss, (long)n * ld_ss, C, c, coef + ((long)n * C + c) * 2);
    }
}

// Small images (the 8 x 8 / 16 x 16 levels, everything at the sampling script's batch 16): partial + finalize are two launches of pure
// latency (6.7 + 3.8 us on 2 MB).  One block per (image, group) instead: grid (32, N), 256 threads walk the group's HW x cpg / 4 channel
// vectors, fp32 sums of (x - pivot) per thread, f64 from the wave reduction on.  Same definition of mean / rstd / (a, b) as above.
template <typename T = float>
__global__ __launch_bounds__(256) void gn_stats_group_kernel(const T* __restrict__ x, int HW, int ldx, int cpg, float eps,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             const T* __restrict__ x2, int ld2, int C1,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ ss, int ld_ss, int C, float* __restrict__ coef) {
    __shared__ double sS[4], sQ[4];
    __shared__ float sMR[2];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int vpp = cpg >> 2, total = HW * vpp;
    int ldp;
    const float pivot = ldv1(gn_src(x, ldx, x2, ld2, C1, (long)n * HW, g * cpg, ldp));
    float S = 0.f, Q = 0.f;
    auto acc4 = [&](const float4& v) {
        float a = v.x - pivot, b = v.y - pivot, c = v.z - pivot, d = v.w - pivot;
        S += (a + b) + (c + d);
        Q += (a * a + b * b) + (c * c + d * d);
    };
    auto addr = [&](int idx) -> const T* {
        const int pix = idx / vpp, v = idx - pix * vpp;
        int ld;
        const T* b = gn_src(x, ldx, x2, ld2, C1, (long)n * HW, g * cpg + 4 * v, ld);
        return b + (long)pix * ld;
    };
    int idx = tid;
    for (; idx + 3 * 256 < total; idx += 4 * 256) {
        const float4 v0 = ldv4(addr(idx)), v1 = ldv4(addr(idx + 256)), v2 = ldv4(addr(idx + 512)), v3 = ldv4(addr(idx + 768));
        acc4(v0); acc4(v1); acc4(v2); acc4(v3);
    }
    for (; idx < total; idx += 256) acc4(ldv4(addr(idx)));
    double s = S, q = Q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); q += __shfl_down(q, o, 64); }
    if ((tid & 63) == 0) { sS[tid >> 6] = s; sQ[tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) {
        s = (sS[0] + sS[1]) + (sS[2] + sS[3]); q = (sQ[0] + sQ[1]) + (sQ[2] + sQ[3]);
        const double cnt = (double)HW * cpg;
        const double m = s / cnt;
        double var = q / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float mu = (float)((double)pivot + m), rs = (float)(1.0 / sqrt(var + (double)eps));
        mean[n * 32 + g] = mu;
        rstd[n * 32 + g] = rs;
        sMR[0] = mu; sMR[1] = rs;
    }
    if (coef) {
        __syncthreads();
        if (tid < cpg) {
            const int c = g * cpg + tid;
            gn_coef_one(sMR[0], sMR[1], gamma[c], beta[c], ss, (long)n * ld_ss, C, c, coef + ((long)n * C + c) * 2);
        }
    }
}
// the one-launch form pays where the two-launch form is pure latency (11 us): a group's walk of at most 16 vectors per thread AND a tensor
// of a few MB — a block reads cpg-channel pieces of its image's rows (32..64 bytes each), which streams at a fraction of the chunked
// kernel's rate (N = 256, 1024 pixels, 128 channels: 142 us against 25)
inline bool gn_small(int N, int HW, int C, int cpg, int groups, int elt) {
    static const int cfg = CDAE_DEV_INT("CDAE_GN_SMALL", 1);
    return cfg && groups == 32 && cpg % 4 == 0 && cpg <= 256 && (long)HW * (cpg >> 2) <= 4096 && HW <= 1024 && (long)N * HW * C * elt <= (6L << 20);
}

// grid N, G threads: stats[n*G+g] = mean, stats[N*G + n*G+g] = rstd
__global__ void gn_finalize_kernel(const float* __restrict__ x, int HW, int ldx, int cpg, int G, int nchunk, float eps,
                                   const float* __restrict__ partial, float* __restrict__ mean, float* __restrict__ rstd,
                                   const float* __restrict__ x2 = nullptr, int ld2 = 0, int C1 = 0) {
    const int n = blockIdx.x, g = threadIdx.x;
    if (g >= G) return;
    double s = 0.0, q = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        const float* pp = partial + (((long)n * nchunk + c) * G + g) * 2;
        s += pp[0]; q += pp[1];
    }
    const double cnt = (double)HW * cpg;
    int ldp;
    const double pivot = *gn_src(x, ldx, x2, ld2, C1, (long)n * HW, g * cpg, ldp);
    const double m = s / cnt;
    double var = q / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[n * G + g] = (float)(pivot + m);
    rstd[n * G + g] = (float)(1.0 / sqrt(var + (double)eps));
}

// GroupNorm statistics from per-(32-row chunk, channel) partial (sum, sumsq) left behind by the producing conv's epilogue
// (igemm.hip ps_kernel / pswin_kernel), for one or two sources (a channel concatenation).  Two small kernels, fixed summation
// order (deterministic): (1) grid (N, C/32): 8 chunk-lanes x 32 channels per block fold the image's HW/32 chunks per channel in
// f64 -> chan[n][c] = (sum, sumsq); (2) grid N: thread g folds the channels of group g.
// nseg: a tensor written by the four sub-pixel phases of an up-conv carries four segments of partials, each [N][HW/4/32][C][2].
__global__ __launch_bounds__(256) void gn_parts_channel_kernel(const float* __restrict__ part1, int C1, int nseg1, const float* __restrict__ part2,
                                                                int C2, int nseg2, int N, int HW, double* __restrict__ chan) {
    __shared__ double sS[8][32], sQ[8][32];
    const int n = blockIdx.y, C = C1 + C2;
    const int cl = threadIdx.x >> 5, c = blockIdx.x * 32 + (threadIdx.x & 31);
    double s = 0.0, q = 0.0;
    if (c < C) {
        const bool second = c >= C1;
        const int Cs = second ? C2 : C1, nseg = second ? nseg2 : nseg1, nch = (HW / nseg) >> 5;
        const float2* base = reinterpret_cast<const float2*>(second ? part2 : part1) + (second ? c - C1 : c);
        for (int sg = 0; sg < nseg; ++sg) {
            const float2* src = base + ((long)sg * N + n) * nch * Cs;
            for (int k = cl; k < nch; k += 8) { const float2 v = src[(long)k * Cs]; s += v.x; q += v.y; }
        }
    }
    sS[cl][threadIdx.x & 31] = s; sQ[cl][threadIdx.x & 31] = q;
    __syncthreads();
    if (cl == 0 && c < C) {
        for (int k = 1; k < 8; ++k) { s += sS[k][threadIdx.x]; q += sQ[k][threadIdx.x]; }
        chan[((long)n * C + c) * 2] = s; chan[((long)n * C + c) * 2 + 1] = q;
    }
}
// Both steps in one launch: a block owns gpb whole groups (gpb * cpg <= 32 channels) of one image, sums their channels exactly as
// gn_parts_channel_kernel does (8 chunk lanes per channel, double) and then the groups exactly as gn_parts_group_kernel does — the
// same additions in the same order, so mean / rstd are bit-identical to the two-kernel form.
__global__ __launch_bounds__(256) void gn_parts_stats_kernel(const float* __restrict__ part1, int C1, int nseg1, const float* __restrict__ part2,
                                                              int C2, int nseg2, int N, int HW, int cpg, int G, int gpb, float eps,
                                                              float* __restrict__ mean, float* __restrict__ rstd,
                                                              const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr,
                                                              const float* __restrict__ ss = nullptr, int ld_ss = 0, float* __restrict__ coef = nullptr) {
    __shared__ double sS[8][32], sQ[8][32];
    const int n = blockIdx.y, g0 = blockIdx.x * gpb, cw = gpb * cpg;
    const int cl = threadIdx.x >> 5, ci = threadIdx.x & 31, c = g0 * cpg + ci;
    const bool live = ci < cw && g0 + ci / cpg < G;
    double s = 0.0, q = 0.0;
    if (live) {
        const bool second = c >= C1;
        const int Cs = second ? C2 : C1, nseg = second ? nseg2 : nseg1, nch = (HW / nseg) >> 5;
        const float2* base = reinterpret_cast<const float2*>(second ? part2 : part1) + (second ? c - C1 : c);
        for (int sg = 0; sg < nseg; ++sg) {
            const float2* src = base + ((long)sg * N + n) * nch * Cs;
            int k = cl;
            for (; k + 24 < nch; k += 32) {          // four loads in flight (a 64 x 64 image: 16 chunks per lane, each a round trip when taken one by one); same order of additions
                const float2 v0 = src[(long)k * Cs], v1 = src[(long)(k + 8) * Cs], v2 = src[(long)(k + 16) * Cs], v3 = src[(long)(k + 24) * Cs];
                s += v0.x; q += v0.y; s += v1.x; q += v1.y; s += v2.x; q += v2.y; s += v3.x; q += v3.y;
            }
            for (; k < nch; k += 8) { const float2 v = src[(long)k * Cs]; s += v.x; q += v.y; }
        }
    }
    sS[cl][ci] = s; sQ[cl][ci] = q;
    __syncthreads();
    if (cl == 0) {
        for (int k = 1; k < 8; ++k) { s += sS[k][ci]; q += sQ[k][ci]; }
        sS[0][ci] = s; sQ[0][ci] = q;
    }
    __syncthreads();
    if (threadIdx.x < gpb && g0 + threadIdx.x < G) {
        const int g = threadIdx.x;
        double gs = 0.0, gq = 0.0;
        for (int j = g * cpg; j < (g + 1) * cpg; ++j) { gs += sS[0][j]; gq += sQ[0][j]; }
        const double cnt = (double)HW * cpg, m = gs / cnt;
        double var = gq / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        mean[n * G + g0 + g] = mu;
        rstd[n * G + g0 + g] = rs;
        if (coef) {
            const int C = G * cpg;
            for (int c = (g0 + g) * cpg; c < (g0 + g + 1) * cpg; ++c) gn_coef_one(mu, rs, gamma[c], beta[c], ss, (long)n * ld_ss, C, c, coef + ((long)n * C + c) * 2);
        }
    }
}
__global__ void gn_parts_group_kernel(const double* __restrict__ chan, int C, int HW, int cpg, int G, float eps,
                                      float* __restrict__ mean, float* __restrict__ rstd) {
    const int n = blockIdx.x, g = threadIdx.x;
    if (g >= G) return;
    double s = 0.0, q = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += chan[((long)n * C + c) * 2]; q += chan[((long)n * C + c) * 2 + 1]; }
    const double cnt = (double)HW * cpg, m = s / cnt;
    double var = q / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[n * G + g] = (float)m;
    rstd[n * G + g] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = silu?( gn(x) * gamma + beta  [ * (1 + scale) + shift ] )
// grid (nchunk, N), 256 threads: a thread owns one channel vector for the whole chunk, so the normalisation folds into a
// per-thread affine (y = x*A + B, the form ATen's CPU kernel uses too) computed once; the pixel loop is pure
// load / fma / SiLU / store with four independent 16-byte loads in flight and no index arithmetic.
// SPLIT: the result is written as two f16 planes (hi = f16(v), lo = f16(v - hi), pitch ldy elements each) — the operand format
// of the pre-split conv / GEMM kernel (igemm.hip ps_kernel), so the consumer's main loop has no conversion work.
typedef _Float16 gn_half4 __attribute__((ext_vector_type(4)));
typedef __bf16 gn_bf4 __attribute__((ext_vector_type(4)));
template <int VEC, bool SPLIT = false, typename T = float>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int C, int ldx, int ldy,
                                                        int cpg, int G, int pix_per_block,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ ss, int ld_ss, int do_silu,
                                                        unsigned short* __restrict__ y_hi = nullptr, unsigned short* __restrict__ y_lo = nullptr,
                                                        const T* __restrict__ x2 = nullptr, int ld2 = 0, int C1 = 0,
                                                        unsigned short* __restrict__ yb_hi = nullptr, unsigned short* __restrict__ yb_lo = nullptr,
                                                        int plane_gm = 0) {
    // plane_gm: the f16 planes are written group-major, [C / 16][N * HW][16] (the window conv kernel's contiguous half-windows)
    const int n = blockIdx.y, E = C / VEC, rows = 256 / E, tid = threadIdx.x;
    const int r = tid / E, e = tid - r * E;
    if (r >= rows) return;
    const int c = e * VEC, g = c / cpg;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    float A[VEC], B[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        A[i] = rs * gamma[c + i];
        B[i] = fmaf(-mu, A[i], beta[c + i]);
        if (ss) {
            const float sc = 1.f + ss[(long)n * ld_ss + c + i], sh = ss[(long)n * ld_ss + C + c + i];
            A[i] *= sc;
            B[i] = fmaf(B[i], sc, sh);
        }
    }
    int ldxe;
    const T* xp = gn_src(x, ldx, x2, ld2, C1, (long)n * HW, c, ldxe);
    ldx = ldxe;
    T* yp = y + (long)n * HW * ldy + c;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    auto apply = [&](float v, int i) { float h = fmaf(v, A[i], B[i]); return do_silu ? silu_t<T>(h) : h; };
    int p = p0 + r;
    if constexpr (VEC == 4) {
        const long hbase = (long)n * HW * ldy + c;
        const long gbase = plane_gm ? ((long)(c >> 4) * gridDim.y * HW + (long)n * HW) * 16 + (c & 15) : hbase;
        const long gld = plane_gm ? 16 : ldy;
        auto put = [&](int pp, const float4& v) {
            float r0 = apply(v.x, 0), r1 = apply(v.y, 1), r2 = apply(v.z, 2), r3 = apply(v.w, 3);
            if constexpr (SPLIT) {
                asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));      // opaque before the split (attention.hip split8)
                gn_half4 hi, lo;
                hi[0] = (_Float16)r0; hi[1] = (_Float16)r1; hi[2] = (_Float16)r2; hi[3] = (_Float16)r3;
                lo[0] = (_Float16)(r0 - (float)hi[0]); lo[1] = (_Float16)(r1 - (float)hi[1]);
                lo[2] = (_Float16)(r2 - (float)hi[2]); lo[3] = (_Float16)(r3 - (float)hi[3]);
                *reinterpret_cast<gn_half4*>(y_hi + gbase + (long)pp * gld) = hi;
                *reinterpret_cast<gn_half4*>(y_lo + gbase + (long)pp * gld) = lo;
                if (yb_hi) {       // training: the same values also as bf16 hi/lo planes, the wgrad kernel's operand format (wgrad.hip)
                    gn_bf4 bh, bl;
                    bh[0] = (__bf16)r0; bh[1] = (__bf16)r1; bh[2] = (__bf16)r2; bh[3] = (__bf16)r3;
                    bl[0] = (__bf16)(r0 - (float)bh[0]); bl[1] = (__bf16)(r1 - (float)bh[1]);
                    bl[2] = (__bf16)(r2 - (float)bh[2]); bl[3] = (__bf16)(r3 - (float)bh[3]);
                    *reinterpret_cast<gn_bf4*>(yb_hi + hbase + (long)pp * ldy) = bh;
                    *reinterpret_cast<gn_bf4*>(yb_lo + hbase + (long)pp * ldy) = bl;
                }
            } else stv4(yp + (long)pp * ldy, make_float4(r0, r1, r2, r3));
        };
        constexpr int U = sizeof(T) == 2 ? GN_U16 : 4;
        for (; p + (U - 1) * rows < p1; p += U * rows) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ldv4(xp + (long)(p + u * rows) * ldx);
#pragma unroll
            for (int u = 0; u < U; ++u) put(p + u * rows, v[u]);
        }
        for (; p < p1; p += rows) put(p, ldv4(xp + (long)p * ldx));
    } else {
        for (; p < p1; p += rows) stv1(yp + (long)p * ldy, apply(ldv1(xp + (long)p * ldx), 0));
    }
}

// The same pass writing GROUP-MAJOR planes ([C / 16][N * HW][16]): grid (pixel chunks, N, C / 16); a block handles ONE 16-channel plane group,
// four lanes per pixel (16 B of x each) and 64 consecutive pixels per pass, so a wave's plane store is 16 pixels x 32 B = one contiguous
// 512-byte run (with the channel-vector mapping of gn_apply_kernel the same layout is 32-byte pieces 16 KB apart: 1.34 -> 1.76 ms per
// DDIM step).  Same per-element arithmetic (folded affine, cdae_silu, opaque before the split): bit-identical values.
template <int GPB>
__global__ __launch_bounds__(256) void gn_apply_gm_kernel(const float* __restrict__ x, int HW, int C, int ldx, int cpg, int G, int pix_per_block,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ ss, int ld_ss, int do_silu,
                                                          unsigned short* __restrict__ y_hi, unsigned short* __restrict__ y_lo,
                                                          const float* __restrict__ x2, int ld2, int C1) {
    // GPB plane groups per block: with two, the eight lanes of a pixel read one whole 128-byte line (32 channels) and a wave stores two runs of
    // 256 contiguous bytes; with one (C % 32 != 0), 64-byte reads and one 512-byte run
    constexpr int LPP = 4 * GPB, PPB = 256 / LPP;                      // lanes per pixel, pixels per pass
    const int n = blockIdx.z, tid = threadIdx.x;                      // plane groups fastest: the blocks that read the same pixels' lines run together
    const int lp = tid % LPP, pl = tid / LPP;
    const int pg = blockIdx.x * GPB + lp / 4, ch4 = lp & 3;
    const int c = pg * 16 + ch4 * 4, g = c / cpg;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    float A[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A[i] = rs * gamma[c + i];
        B[i] = fmaf(-mu, A[i], beta[c + i]);
        if (ss) {
            const float sc = 1.f + ss[(long)n * ld_ss + c + i], sh = ss[(long)n * ld_ss + C + c + i];
            A[i] *= sc;
            B[i] = fmaf(B[i], sc, sh);
        }
    }
    int ld;
    const float* xp = gn_src(x, ldx, x2, ld2, C1, (long)n * HW, c, ld);
    const long P = (long)gridDim.z * HW;
    unsigned short* const oh = y_hi + ((long)pg * P + (long)n * HW) * 16 + ch4 * 4;
    unsigned short* const ol = y_lo + ((long)pg * P + (long)n * HW) * 16 + ch4 * 4;
    const int p0 = blockIdx.y * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    auto put = [&](int pp, const float4& v) {
        float r0 = fmaf(v.x, A[0], B[0]), r1 = fmaf(v.y, A[1], B[1]), r2 = fmaf(v.z, A[2], B[2]), r3 = fmaf(v.w, A[3], B[3]);
        if (do_silu) { r0 = silu_f(r0); r1 = silu_f(r1); r2 = silu_f(r2); r3 = silu_f(r3); }
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));      // opaque before the split (attention.hip split8)
        gn_half4 hi, lo;
        hi[0] = (_Float16)r0; hi[1] = (_Float16)r1; hi[2] = (_Float16)r2; hi[3] = (_Float16)r3;
        lo[0] = (_Float16)(r0 - (float)hi[0]); lo[1] = (_Float16)(r1 - (float)hi[1]);
        lo[2] = (_Float16)(r2 - (float)hi[2]); lo[3] = (_Float16)(r3 - (float)hi[3]);
        *reinterpret_cast<gn_half4*>(oh + (long)pp * 16) = hi;
        *reinterpret_cast<gn_half4*>(ol + (long)pp * 16) = lo;
    };
    int p = p0 + pl;
    for (; p + 3 * PPB < p1; p += 4 * PPB) {
        const float4 v0 = *reinterpret_cast<const float4*>(xp + (long)p * ld);
        const float4 v1 = *reinterpret_cast<const float4*>(xp + (long)(p + PPB) * ld);
        const float4 v2 = *reinterpret_cast<const float4*>(xp + (long)(p + 2 * PPB) * ld);
        const float4 v3 = *reinterpret_cast<const float4*>(xp + (long)(p + 3 * PPB) * ld);
        put(p, v0); put(p + PPB, v1); put(p + 2 * PPB, v2); put(p + 3 * PPB, v3);
    }
    for (; p < p1; p += PPB) put(p, *reinterpret_cast<const float4*>(xp + (long)p * ld));
}

// ------------------------------------------------------------------ GroupNorm backward
// With xh = (x-mean)*rstd, a = 1+scale (or 1), u = xh*gamma+beta, h = u*a+shift, out = silu?(h):
//   dh = dy * silu'(h);  du = dh*a;  dxh = du*gamma
//   dx = rstd * (dxh - mean_g(dxh) - xh * mean_g(dxh*xh))
//   dgamma[c] += sum du*xh, dbeta[c] += sum du, dscale[n][c] = sum_p dh*u, dshift[n][c] = sum_p dh
// Pass 1 (this kernel): per (n, chunk) partial sums  [G][2] (sum dxh, sum dxh*xh) and per-channel
// partials [C][4] (du*xh, du, dh*u, dh).  grid (nchunk, N).
template <int VEC, typename T = float>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              int HW, int C, int ldx, int lddy, int cpg, int G, int pix_per_block,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ ss, int ld_ss, int do_silu,
                                                              float* __restrict__ gpart /*[N][nchunk][G][2]*/,
                                                              float* __restrict__ cpart /*[N][nchunk][C][4]*/,
                                                              const T* __restrict__ x2 = nullptr, int ld2 = 0, int C1 = 0) {
    __shared__ float sA[256], sB[256];
    __shared__ float sC[256 * 4 * 4];          // per thread VEC x 4 channel sums (only when rows > 1)
    const int n = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int E = C / VEC, rows = 256 / E, tid = threadIdx.x;
    const int r = tid / E, e = tid - r * E;
    const bool active = r < rows;
    float A = 0.f, B = 0.f;
    float ch[VEC][4];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { ch[i][0] = ch[i][1] = ch[i][2] = ch[i][3] = 0.f; }
    if (active) {
        const int c = e * VEC, g = c / cpg;
        const float mu = mean[n * G + g], rs = rstd[n * G + g];
        float gm[VEC], bt[VEC], a[VEC], sh[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            gm[i] = gamma[c + i]; bt[i] = beta[c + i];
            a[i] = ss ? 1.f + ss[(long)n * ld_ss + c + i] : 1.f;
            sh[i] = ss ? ss[(long)n * ld_ss + C + c + i] : 0.f;
        }
        const int p0 = chunk * pix_per_block, p1 = min(HW, p0 + pix_per_block);
        int ldxe;                                   // this thread's channel vector lives in one of the two sources (skip concatenation)
        const T* const xb = gn_src(x, ldx, x2, ld2, C1, (long)n * HW, c, ldxe);
        auto pixel = [&](const float* xv, const float* dv) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float xh = (xv[i] - mu) * rs;
                float u = xh * gm[i] + bt[i];
                float h = u * a[i] + sh[i];
                float dh = dv[i];
                if (do_silu) { float s = sigmoid_t<T>(h); dh *= s * (1.f + h * (1.f - s)); }
                float du = dh * a[i];
                float dxh = du * gm[i];
                A += dxh; B += dxh * xh;
                ch[i][0] += du * xh; ch[i][1] += du; ch[i][2] += dh * u; ch[i][3] += dh;
            }
        };
        int p = p0 + r;
        if (VEC == 4) {
            const T* const db = dy + (long)n * HW * lddy + c;
            // four pixels (eight 16-byte loads) in flight per thread — eight pixels of 8-byte loads on bf16 rows; the pixels are
            // accumulated in the same order as one by one
            constexpr int U = sizeof(T) == 2 ? GN_U16 : 4;
            for (; p + (U - 1) * rows < p1; p += U * rows) {
                float4 t[U], d[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    t[u] = ldv4(xb + (long)(p + u * rows) * ldxe);
                    d[u] = ldv4(db + (long)(p + u * rows) * lddy);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float xv[4] = {t[u].x, t[u].y, t[u].z, t[u].w}, dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
                    pixel(xv, dv);
                }
            }
            for (; p < p1; p += rows) {
                const float4 t = ldv4(xb + (long)p * ldxe);
                const float4 d = ldv4(db + (long)p * lddy);
                const float xv[4] = {t.x, t.y, t.z, t.w}, dv[4] = {d.x, d.y, d.z, d.w};
                pixel(xv, dv);
            }
        } else {
            for (; p < p1; p += rows) {
                const long pix = (long)n * HW + p;
                const float xv[1] = {ldv1(xb + (long)p * ldxe)}, dv[1] = {ldv1(dy + pix * lddy + c)};
                pixel(xv, dv);
            }
        }
    }
    sA[tid] = A; sB[tid] = B;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) sC[(i * 4 + k) * 256 + tid] = ch[i][k];      // [sum][thread]: consecutive lanes on consecutive banks (thread-major
                                                                                 // 64-byte records were 16-way conflicts: 87 % of the kernel's LDS cycles)
    __syncthreads();
    if (tid < G) {
        const int e0 = tid * cpg / VEC, e1 = (tid + 1) * cpg / VEC;
        float s = 0.f, q = 0.f;
        for (int rr = 0; rr < rows; ++rr)
            for (int ee = e0; ee < e1; ++ee) { s += sA[rr * E + ee]; q += sB[rr * E + ee]; }
        float* out = gpart + (((long)n * nchunk + chunk) * G + tid) * 2;
        out[0] = s; out[1] = q;
    }
    // channel partials: thread (rr=0, e) folds the rows
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s = 0.f;
                for (int rr = 0; rr < rows; ++rr) s += sC[(i * 4 + k) * 256 + rr * E + e];
                cpart[(((long)n * nchunk + chunk) * C + e * VEC + i) * 4 + k] = s;
            }
    }
}

// Pass 2a: fold chunk partials.  grid (C/64 + 1, N), 256 threads = 4 chunk-lanes x 64 columns; fixed summation order.
// Blocks x < C/64 fold 64 channels each ([N][C][2] for pass 2b, and d(scale, shift)); the last block folds the G group sums.
__global__ __launch_bounds__(256) void gn_bwd_fold_kernel(int N, int HW, int C, int cpg, int G, int nchunk,
                                                          const float* __restrict__ gpart, const float* __restrict__ cpart,
                                                          float* __restrict__ gsum /*[N][G][2]*/, float* __restrict__ dss, int ld_dss,
                                                          float* __restrict__ nc_part /*[N][C][2]*/) {
    __shared__ double red[4][64][4];
    const int n = blockIdx.y, cl = threadIdx.x >> 6, col = threadIdx.x & 63;
    const int cblocks = (C + 63) / 64;
    double v[4] = {0, 0, 0, 0};
    if ((int)blockIdx.x < cblocks) {
        const int c = blockIdx.x * 64 + col;
        if (c < C)
        {
            int k = cl;
            for (; k + 12 < nchunk; k += 16) {      // four chunk records in flight; same order of additions
                const float4 p0 = *reinterpret_cast<const float4*>(cpart + (((long)n * nchunk + k) * C + c) * 4), p1 = *reinterpret_cast<const float4*>(cpart + (((long)n * nchunk + k + 4) * C + c) * 4),
                             p2 = *reinterpret_cast<const float4*>(cpart + (((long)n * nchunk + k + 8) * C + c) * 4), p3 = *reinterpret_cast<const float4*>(cpart + (((long)n * nchunk + k + 12) * C + c) * 4);
                v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
                v[0] += p1.x; v[1] += p1.y; v[2] += p1.z; v[3] += p1.w;
                v[0] += p2.x; v[1] += p2.y; v[2] += p2.z; v[3] += p2.w;
                v[0] += p3.x; v[1] += p3.y; v[2] += p3.z; v[3] += p3.w;
            }
            for (; k < nchunk; k += 4) {
                const float4 pp = *reinterpret_cast<const float4*>(cpart + (((long)n * nchunk + k) * C + c) * 4);
                v[0] += pp.x; v[1] += pp.y; v[2] += pp.z; v[3] += pp.w;
            }
        }
        for (int i = 0; i < 4; ++i) red[cl][col][i] = v[i];
        __syncthreads();
        if (cl == 0 && c < C) {
            for (int k = 1; k < 4; ++k)
                for (int i = 0; i < 4; ++i) v[i] += red[k][col][i];
            nc_part[((long)n * C + c) * 2 + 0] = (float)v[0];
            nc_part[((long)n * C + c) * 2 + 1] = (float)v[1];
            if (dss) { dss[(long)n * ld_dss + c] = (float)v[2]; dss[(long)n * ld_dss + C + c] = (float)v[3]; }
        }
    } else {
        for (int g0 = 0; g0 < G; g0 += 64) {            // G <= 256: at most four rounds
            const int g = g0 + col;
            v[0] = v[1] = 0;
            if (g < G)
                for (int k = cl; k < nchunk; k += 4) {
                    const float* pp = gpart + (((long)n * nchunk + k) * G + g) * 2;
                    v[0] += pp[0]; v[1] += pp[1];
                }
            __syncthreads();
            red[cl][col][0] = v[0]; red[cl][col][1] = v[1];
            __syncthreads();
            if (cl == 0 && g < G) {
                for (int k = 1; k < 4; ++k) { v[0] += red[k][col][0]; v[1] += red[k][col][1]; }
                const double cnt = (double)HW * cpg;
                gsum[(n * G + g) * 2 + 0] = (float)(v[0] / cnt);
                gsum[(n * G + g) * 2 + 1] = (float)(v[1] / cnt);
            }
        }
    }
}

// Pass 2b: dgamma[c] (+)= sum_n, dbeta[c] (+)= sum_n.  grid C/32, 256 threads = 8 image lanes x 32 channels (fixed fold order)
__device__ __forceinline__ void gn_bwd_param8_block(int cblock, int N, int C, const float* __restrict__ nc_part, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int accumulate, double (*sA)[32], double (*sB)[32]) {
    const int cc = threadIdx.x & 31, nl = threadIdx.x >> 5, c = cblock * 32 + cc;
    double a = 0, b = 0;
    if (c < C)
    {
        int n = nl;
        for (; n + 24 < N; n += 32) {       // four images in flight per lane (batch 256: 32 images per lane, each load a round trip when taken one by one); same order of additions
            const float2 v0 = *reinterpret_cast<const float2*>(nc_part + ((long)n * C + c) * 2), v1 = *reinterpret_cast<const float2*>(nc_part + ((long)(n + 8) * C + c) * 2),
                         v2 = *reinterpret_cast<const float2*>(nc_part + ((long)(n + 16) * C + c) * 2), v3 = *reinterpret_cast<const float2*>(nc_part + ((long)(n + 24) * C + c) * 2);
            a += v0.x; b += v0.y; a += v1.x; b += v1.y; a += v2.x; b += v2.y; a += v3.x; b += v3.y;
        }
        for (; n < N; n += 8) { const float2 v = *reinterpret_cast<const float2*>(nc_part + ((long)n * C + c) * 2); a += v.x; b += v.y; }
    }
    sA[nl][cc] = a; sB[nl][cc] = b;
    __syncthreads();
    if (nl == 0 && c < C) {
        a = 0; b = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += sA[k][cc]; b += sB[k][cc]; }
        dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)a;
        dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)b;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_param8_kernel(int N, int C, const float* __restrict__ nc_part, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate) {
    __shared__ double sA[8][32], sB[8][32];
    gn_bwd_param8_block(blockIdx.x, N, C, nc_part, dgamma, dbeta, accumulate, sA, sB);
}

__global__ void gn_bwd_param_kernel(int N, int C, const float* __restrict__ nc_part, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int n = 0; n < N; ++n) { a += nc_part[((long)n * C + c) * 2]; b += nc_part[((long)n * C + c) * 2 + 1]; }
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)a;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)b;
}

// Pass 3: dx
template <int VEC, typename T = float>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                         long total_vec, int HW, int C, int ldx, int lddy, int lddx, int cpg, int G,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ ss, int ld_ss, int do_silu,
                                                         const float* __restrict__ gsum, int accumulate,
                                                         const T* __restrict__ dx_add = nullptr, int ld_add = 0,
                                                         unsigned short* __restrict__ dxb_hi = nullptr, unsigned short* __restrict__ dxb_lo = nullptr,
                                                         const T* __restrict__ x2 = nullptr, int ld2 = 0, int C1 = 0,
                                                         T* __restrict__ dx2 = nullptr, int lddx2 = 0) {
    // x2 / dx2: channels [C1, C) of the normalised tensor live in a second source (the skip concatenation read in place); their
    // gradient goes to dx2 [pixels][C - C1]
    // dx_add: a second gradient of the same tensor (the ResBlock's residual path) added on the way out instead of by a separate
    // add kernel.  dxb_hi / dxb_lo (VEC == 4, dense [pixels][C]): the result also — or, with dx == nullptr, only — as bf16 hi/lo
    // planes, the operand format in which the preceding conv's dgrad / wgrad consume it.
    const int E = C / VEC;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total_vec; idx += (long)gridDim.x * blockDim.x) {
        const long pix = idx / E;
        const int e = (int)(idx - pix * E), n = (int)(pix / HW), c = e * VEC, g = c / cpg;
        const float mu = mean[n * G + g], rs = rstd[n * G + g];
        const float m1 = gsum[(n * G + g) * 2], m2 = gsum[(n * G + g) * 2 + 1];
        float xv[VEC], dv[VEC], o[VEC];
        const bool second = x2 != nullptr && c >= C1;
        const T* const xp = second ? x2 + pix * ld2 + (c - C1) : x + pix * ldx + c;
        T* const dxp = dx == nullptr ? nullptr : (second ? dx2 + pix * lddx2 + (c - C1) : dx + pix * lddx + c);
        if (VEC == 4) {
            float4 t = ldv4(xp);
            float4 d = ldv4(dy + pix * lddy + c);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
        } else { xv[0] = ldv1(xp); dv[0] = ldv1(dy + pix * lddy + c); }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float gm = gamma[c + i];
            float a = ss ? 1.f + ss[(long)n * ld_ss + c + i] : 1.f;
            float sh = ss ? ss[(long)n * ld_ss + C + c + i] : 0.f;
            float xh = (xv[i] - mu) * rs;
            float h = (xh * gm + beta[c + i]) * a + sh;
            float dh = dv[i];
            if (do_silu) { float s = sigmoid_t<T>(h); dh *= s * (1.f + h * (1.f - s)); }
            float dxh = dh * a * gm;
            o[i] = rs * (dxh - m1 - xh * m2);
        }
        if (accumulate) {
            if (VEC == 4) {
                float4 t = ldv4(dxp);
                o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
            } else o[0] += ldv1(dxp);
        }
        if (dx_add) {
            if (VEC == 4) {
                float4 t = ldv4(dx_add + pix * ld_add + c);
                o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w;
            } else o[0] += ldv1(dx_add + pix * ld_add + c);
        }
        if (dx) {
            if (VEC == 4) stv4(dxp, make_float4(o[0], o[1], o[2], o[3]));
            else stv1(dxp, o[0]);
        }
        if constexpr (VEC == 4) {
            if (dxb_hi) {
                gn_bf4 bh, bl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { bh[i] = (__bf16)o[i]; bl[i] = (__bf16)(o[i] - (float)bh[i]); }
                *reinterpret_cast<gn_bf4*>(dxb_hi + pix * C + c) = bh;
                *reinterpret_cast<gn_bf4*>(dxb_lo + pix * C + c) = bl;
            }
        }
    }
}

// Pass 3, streaming form (VEC == 4): grid (nchunk, N) like pass 1 — a thread owns one channel vector of one image for a range of
// pixels, so every per-(image, channel) constant (mean, rstd, the group sums, gamma, beta, scale, shift) is loaded ONCE (the
// grid-stride form above re-reads ~20 scalars per element, each a dependent round trip through L1) and the pixel loop is two pixels
// of 16-byte loads in flight, fma / sigmoid, 16-byte stores.  Same per-element arithmetic in the same order: bit-identical results.
template <typename T = float>
__global__ __launch_bounds__(256) void gn_bwd_dx_stream_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                                int HW, int C, int ldx, int lddy, int lddx, int cpg, int G, int pix_per_block,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ ss, int ld_ss, int do_silu,
                                                                const float* __restrict__ gsum, int accumulate,
                                                                const T* __restrict__ dx_add, int ld_add,
                                                                unsigned short* __restrict__ dxb_hi, unsigned short* __restrict__ dxb_lo,
                                                                const T* __restrict__ x2, int ld2, int C1,
                                                                T* __restrict__ dx2, int lddx2,
                                                                int N, const float* __restrict__ nc_part, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int accumulate_params,
                                                                const float* __restrict__ gpart = nullptr, const float* __restrict__ cpart = nullptr,
                                                                int nchunk_p = 0, float* __restrict__ dss = nullptr, int ld_dss = 0) {
    if (cpart != nullptr && (int)blockIdx.y >= N) {
        // FOLD form (gpart / cpart given: no separate fold launch): rows N.. of the grid fold pass 1's per-chunk channel sums — channel
        // block cb = (y - N) * gridDim.x + x, 8 image lanes x 32 channels: d(scale, shift)[n][c] = the chunk sums of image n, and
        // dgamma / dbeta = those of every image, all in a fixed order (chunks of an image in sequence, images of a lane in sequence, lanes 0..7)
        __shared__ double sA[8][32], sB[8][32];
        const int cb = ((int)blockIdx.y - N) * (int)gridDim.x + (int)blockIdx.x;
        if (cb * 32 >= C) return;
        const int cc = threadIdx.x & 31, nl = threadIdx.x >> 5, c = cb * 32 + cc;
        double a = 0, b = 0;
        if (c < C) {
            for (int n = nl; n < N; n += 8) {
                const float* src = cpart + ((long)n * nchunk_p * C + c) * 4;
                double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                int k = 0;
                for (; k + 4 <= nchunk_p; k += 4) {      // four chunk records in flight; additions in chunk order
                    const float4 p0 = *reinterpret_cast<const float4*>(src + (long)k * C * 4), p1 = *reinterpret_cast<const float4*>(src + (long)(k + 1) * C * 4),
                                 p2 = *reinterpret_cast<const float4*>(src + (long)(k + 2) * C * 4), p3 = *reinterpret_cast<const float4*>(src + (long)(k + 3) * C * 4);
                    v0 += p0.x; v1 += p0.y; v2 += p0.z; v3 += p0.w;
                    v0 += p1.x; v1 += p1.y; v2 += p1.z; v3 += p1.w;
                    v0 += p2.x; v1 += p2.y; v2 += p2.z; v3 += p2.w;
                    v0 += p3.x; v1 += p3.y; v2 += p3.z; v3 += p3.w;
                }
                for (; k < nchunk_p; ++k) { const float4 pp = *reinterpret_cast<const float4*>(src + (long)k * C * 4); v0 += pp.x; v1 += pp.y; v2 += pp.z; v3 += pp.w; }
                a += (double)(float)v0; b += (double)(float)v1;          // (rounded per image like the separate fold's [N][C][2] table)
                if (dss) { dss[(long)n * ld_dss + c] = (float)v2; dss[(long)n * ld_dss + C + c] = (float)v3; }
            }
        }
        sA[nl][cc] = a; sB[nl][cc] = b;
        __syncthreads();
        if (nl == 0 && c < C) {
            a = 0; b = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { a += sA[k][cc]; b += sB[k][cc]; }
            dgamma[c] = (accumulate_params ? dgamma[c] : 0.f) + (float)a;
            dbeta[c] = (accumulate_params ? dbeta[c] : 0.f) + (float)b;
        }
        return;
    }
    if (cpart == nullptr && (int)blockIdx.y == N) {       // (nc_part != nullptr) one extra row of blocks: pass 2b, dgamma / dbeta = the per-image sums folded over the batch
        __shared__ double sA[8][32], sB[8][32];
        for (int cb = blockIdx.x; cb * 32 < C; cb += gridDim.x) {
            if (cb != (int)blockIdx.x) __syncthreads();
            gn_bwd_param8_block(cb, N, C, nc_part, dgamma, dbeta, accumulate_params, sA, sB);
        }
        return;
    }
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int E = C / 4, rows = 256 / E, tid = threadIdx.x;
    const int r = tid / E, e = tid - r * E;
    if (r >= rows) return;
    const int c = e * 4, g = c / cpg;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    float m1, m2;
    if (gpart != nullptr) {
        // the group's two means from pass 1's chunk sums (at most 64 records of 8 bytes, L2-resident): every thread folds its own group's —
        // the same few hundred bytes the separate fold launch read once and every dx block then re-read as a table
        const float* src = gpart + ((long)n * nchunk_p * G + g) * 2;
        double s0 = 0, s1 = 0;
        int k = 0;
        for (; k + 4 <= nchunk_p; k += 4) {
            const float2 p0 = *reinterpret_cast<const float2*>(src + (long)k * G * 2), p1 = *reinterpret_cast<const float2*>(src + (long)(k + 1) * G * 2),
                         p2 = *reinterpret_cast<const float2*>(src + (long)(k + 2) * G * 2), p3 = *reinterpret_cast<const float2*>(src + (long)(k + 3) * G * 2);
            s0 += p0.x; s1 += p0.y; s0 += p1.x; s1 += p1.y; s0 += p2.x; s1 += p2.y; s0 += p3.x; s1 += p3.y;
        }
        for (; k < nchunk_p; ++k) { const float2 pp = *reinterpret_cast<const float2*>(src + (long)k * G * 2); s0 += pp.x; s1 += pp.y; }
        const double cnt = (double)HW * cpg;
        m1 = (float)(s0 / cnt); m2 = (float)(s1 / cnt);
    } else { m1 = gsum[(n * G + g) * 2]; m2 = gsum[(n * G + g) * 2 + 1]; }
    float gm[4], bt[4], a[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gm[i] = gamma[c + i]; bt[i] = beta[c + i];
        a[i] = ss ? 1.f + ss[(long)n * ld_ss + c + i] : 1.f;
        sh[i] = ss ? ss[(long)n * ld_ss + C + c + i] : 0.f;
    }
    const bool second = x2 != nullptr && c >= C1;
    const long pixb = (long)n * HW;
    const T* const xb = second ? x2 + pixb * ld2 + (c - C1) : x + pixb * ldx + c;
    const int xl = second ? ld2 : ldx;
    T* const ob = dx == nullptr ? nullptr : (second ? dx2 + pixb * lddx2 + (c - C1) : dx + pixb * lddx + c);
    const int ol = second ? lddx2 : lddx;
    const T* const db = dy + pixb * lddy + c;
    const T* const ab = dx_add ? dx_add + pixb * ld_add + c : nullptr;
    const int p0 = chunk * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    auto pixel = [&](int p, const float4& t, const float4& d, const float4& acc4, const float4& add4) {
        const float xv[4] = {t.x, t.y, t.z, t.w}, dv[4] = {d.x, d.y, d.z, d.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float xh = (xv[i] - mu) * rs;
            float h = (xh * gm[i] + bt[i]) * a[i] + sh[i];
            float dh = dv[i];
            if (do_silu) { float s = sigmoid_t<T>(h); dh *= s * (1.f + h * (1.f - s)); }
            float dxh = dh * a[i] * gm[i];
            o[i] = rs * (dxh - m1 - xh * m2);
        }
        if (accumulate) { o[0] += acc4.x; o[1] += acc4.y; o[2] += acc4.z; o[3] += acc4.w; }
        if (ab) { o[0] += add4.x; o[1] += add4.y; o[2] += add4.z; o[3] += add4.w; }
        if (ob) stv4(ob + (long)p * ol, make_float4(o[0], o[1], o[2], o[3]));
        if (dxb_hi) {
            gn_bf4 bh, bl;
#pragma unroll
            for (int i = 0; i < 4; ++i) { bh[i] = (__bf16)o[i]; bl[i] = (__bf16)(o[i] - (float)bh[i]); }
            *reinterpret_cast<gn_bf4*>(dxb_hi + (pixb + p) * C + c) = bh;
            *reinterpret_cast<gn_bf4*>(dxb_lo + (pixb + p) * C + c) = bl;
        }
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = p0 + r;
    for (; p + rows < p1; p += 2 * rows) {                            // two pixels: up to eight 16-byte loads in flight
        const int q = p + rows;
        const float4 t0 = ldv4(xb + (long)p * xl), t1 = ldv4(xb + (long)q * xl);
        const float4 d0 = ldv4(db + (long)p * lddy), d1 = ldv4(db + (long)q * lddy);
        float4 c0 = z4, c1 = z4, e0 = z4, e1 = z4;
        if (accumulate) { c0 = ldv4(ob + (long)p * ol); c1 = ldv4(ob + (long)q * ol); }
        if (ab) { e0 = ldv4(ab + (long)p * ld_add); e1 = ldv4(ab + (long)q * ld_add); }
        pixel(p, t0, d0, c0, e0);
        pixel(q, t1, d1, c1, e1);
    }
    for (; p < p1; p += rows) {
        const float4 t0 = ldv4(xb + (long)p * xl);
        const float4 d0 = ldv4(db + (long)p * lddy);
        float4 c0 = z4, e0 = z4;
        if (accumulate) c0 = ldv4(ob + (long)p * ol);
        if (ab) e0 = ldv4(ab + (long)p * ld_add);
        pixel(p, t0, d0, c0, e0);
    }
}

// ------------------------------------------------------------------ BatchNorm (encoder; C <= 256 channels, rows = N*H*W)
// partial[(chunk*C + c)*2] = (sum, sumsq) of x - x[0][c]
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, long rows_total, int C, int rows_per_block,
                                                          float* __restrict__ partial) {
    __shared__ float sS[256], sQ[256];
    const int tid = threadIdx.x, rows = 256 / C, r = tid / C, c = tid - r * C;
    float S = 0.f, Q = 0.f;
    if (r < rows) {
        const float pivot = x[c];
        const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(rows_total, p0 + rows_per_block);
        for (long p = p0 + r; p < p1; p += rows) { float a = x[p * C + c] - pivot; S += a; Q += a * a; }
    }
    sS[tid] = S; sQ[tid] = Q;
    __syncthreads();
    if (tid < C) {
        float s = 0.f, q = 0.f;
        for (int rr = 0; rr < rows; ++rr) { s += sS[rr * C + tid]; q += sQ[rr * C + tid]; }
        partial[((long)blockIdx.x * C + tid) * 2] = s;
        partial[((long)blockIdx.x * C + tid) * 2 + 1] = q;
    }
}

// training: batch mean / biased var -> (scale, shift) for the apply pass, save mean & rstd, update running stats
// (momentum 0.1, unbiased variance) like nn.BatchNorm2d.  eval: scale/shift from the running stats.
__global__ void bn_finalize_kernel(const float* __restrict__ x, long rows_total, int C, int nchunk, const float* __restrict__ partial,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int training, float eps, float momentum,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_rstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m, v;
    if (training) {
        double s = 0, q = 0;
        {
            const float2* pp = reinterpret_cast<const float2*>(partial) + c;
            int k = 0;
            for (; k + 8 <= nchunk; k += 8) {       // eight chunk records in flight (one block walks up to 1024 of them: 23 us one by one at batch 256); same order of additions
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = pp[(long)(k + u) * C];
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
            }
            for (; k < nchunk; ++k) { const float2 v = pp[(long)k * C]; s += v.x; q += v.y; }
        }
        double cnt = (double)rows_total, mm = s / cnt, var = q / cnt - mm * mm;
        if (var < 0) var = 0;
        m = (float)(x[c] + mm); v = (float)var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * cnt / (cnt - 1.0));
    } else { m = running_mean[c]; v = running_var[c]; }
    float rs = 1.f / sqrtf(v + eps);
    float a = gamma[c] * rs;
    scale[c] = a; shift[c] = beta[c] - m * a;
    if (save_mean) { save_mean[c] = m; save_rstd[c] = rs; }
}

__global__ void affine_lrelu_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C,
                                    const float* __restrict__ scale, const float* __restrict__ shift, float slope) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c = (int)(idx % C);
        float h = x[idx] * scale[c] + shift[c];
        y[idx] = h > 0.f ? h : h * slope;
    }
}

// BN+LeakyReLU backward (training-mode statistics): xh=(x-mean)*rstd, h = xh*gamma+beta, y = lrelu(h)
// pass 1: per-channel partial sums of dh and dh*xh; pass 2 (finalize): dgamma, dbeta, means; pass 3: dx
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, long rows_total, int C,
                                                              int rows_per_block, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                                              float* __restrict__ partial) {
    __shared__ float sS[256], sQ[256];
    const int tid = threadIdx.x, rows = 256 / C, r = tid / C, c = tid - r * C;
    float S = 0.f, Q = 0.f;
    if (r < rows) {
        const float mu = mean[c], rs = rstd[c], sc = scale[c], sf = shift[c];
        const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(rows_total, p0 + rows_per_block);
        for (long p = p0 + r; p < p1; p += rows) {
            float xv = x[p * C + c];
            float xh = (xv - mu) * rs;
            float h = xv * sc + sf;            // exactly the forward's pre-activation: same side of the LeakyReLU kink
            float dh = dy[p * C + c] * (h > 0.f ? 1.f : slope);
            S += dh; Q += dh * xh;
        }
    }
    sS[tid] = S; sQ[tid] = Q;
    __syncthreads();
    if (tid < C) {
        float s = 0.f, q = 0.f;
        for (int rr = 0; rr < rows; ++rr) { s += sS[rr * C + tid]; q += sQ[rr * C + tid]; }
        partial[((long)blockIdx.x * C + tid) * 2] = s;
        partial[((long)blockIdx.x * C + tid) * 2 + 1] = q;
    }
}

__global__ void bn_bwd_finalize_kernel(long rows_total, int C, int nchunk, const float* __restrict__ partial,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, float* __restrict__ sums /*[C][2]*/) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    {
        const float2* pp = reinterpret_cast<const float2*>(partial) + c;
        int k = 0;
        for (; k + 8 <= nchunk; k += 8) {           // eight chunk records in flight; same order of additions
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(long)(k + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
        }
        for (; k < nchunk; ++k) { const float2 v = pp[(long)k * C]; s += v.x; q += v.y; }
    }
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
    sums[c * 2] = (float)(s / (double)rows_total);
    sums[c * 2 + 1] = (float)(q / (double)rows_total);
}

__global__ void bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long total, int C,
                                 const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                 const float* __restrict__ scale, const float* __restrict__ shift, float slope, const float* __restrict__ sums) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int c = (int)(idx % C);
        float rs = rstd[c], gm = gamma[c];
        float xv = x[idx];
        float xh = (xv - mean[c]) * rs;
        float h = xv * scale[c] + shift[c];
        float dh = dy[idx] * (h > 0.f ? 1.f : slope);
        dx[idx] = gm * rs * (dh - sums[c * 2] - xh * sums[c * 2 + 1]);
    }
}

// ------------------------------------------------------------------ row softmax (in place), one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, long rows, int T) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = s + row * T;
    float v[16];                                  // T <= 1024; statically indexed (stays in registers)
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < T ? p[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < T ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < T) p[c] = v[i] * inv;
    }
}

// dS = P * (dP - rowsum(dP * P)), in place on dP
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, long rows, int T) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * T;
    float* d = dP + row * T;
    float dot = 0.f;
    for (int c = lane; c < T; c += 64) dot += p[c] * d[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int c = lane; c < T; c += 64) d[c] = p[c] * (d[c] - dot);
}

int grid_for(long total, int per_block = 256, int cap = 4096) {
    long b = (total + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int gn_chunks(int HW, int N, int E) {
    int rows = 256 / E;
    int nchunk = HW / (rows * 8);              // >= 8 pixels per thread
    if (nchunk < 1) nchunk = 1;
    static const int cfg_cap = CDAE_DEV_INT("CDAE_GN_CHUNK_CAP", 1024);      // blocks per launch: 2048 / 1024 / 512 -> 30.0 / 29.7 / 29.8 ms per C64 training step (short blocks are all epilogue)
    while (nchunk > 1 && (long)nchunk * N > cfg_cap) nchunk >>= 1;
    if (nchunk > CDAE_GN_MAX_CHUNKS) nchunk = CDAE_GN_MAX_CHUNKS;
    return nchunk;
}

}  // namespace

#define CHECK_LAUNCH(msg) do { if (hipGetLastError() != hipSuccess) return cdae_fail(msg); } while (0)

template <typename T>
static int gn_bwd_impl(const T* x, const T* x2, int ld2, int C1, const T* dy, T* dx, T* dx2, int lddx2, int N, int HW, int C, int ldx,
                       int lddy, int lddx, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* scale_shift, int ld_ss, int silu, float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift,
                       int ld_dss, int accumulate_dx, const T* dx_add, int ld_add, unsigned short* dxb_hi, unsigned short* dxb_lo, float* ws,
                       void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!dx && !dxb_hi) return cdae_fail("gn_bwd: no output for dx");
    if ((dxb_hi != nullptr) != (dxb_lo != nullptr)) return cdae_fail("gn_bwd: both bf16 planes or none");
    if (!dx && accumulate_dx) return cdae_fail("gn_bwd: accumulate_dx needs the fp32 dx");
    if (dxb_hi && !((C / groups) % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && (!dx || lddx % 4 == 0) && (!dx_add || ld_add % 4 == 0)))
        return cdae_fail("gn_bwd: plane output needs the 4-channel vector path");
    const int cpg = C / groups;
    const int VEC = (cpg % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && (!dx || lddx % 4 == 0) && (!dx_add || ld_add % 4 == 0)) ? 4 : 1;
    const int E = C / VEC;
    if (E > 256) return cdae_fail("gn_bwd: unsupported channel count");
    const int nchunk = gn_chunks(HW, N, E);
    const int ppb = (HW + nchunk - 1) / nchunk;
    float* gpart = ws;
    float* cpart = gpart + (size_t)N * nchunk * groups * 2;
    float* gsum = cpart + (size_t)N * nchunk * C * 4;
    float* ncp = gsum + (size_t)N * groups * 2;
    const long total = (long)N * HW * E;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 5.0 * sizeof(T), st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "gn_bwd_impl N=%d HW=%d C=%d B=%d", (int)N, (int)HW, (int)C, (int)sizeof(T)); cdae_prof_tag(tag); }
    if (VEC == 4) hipLaunchKernelGGL((gn_bwd_partial_kernel<4, T>), dim3(nchunk, N), dim3(256), 0, st, x, dy, HW, C, ldx, lddy, cpg, groups, ppb, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, gpart, cpart, x2, ld2, C1);
    else hipLaunchKernelGGL((gn_bwd_partial_kernel<1, T>), dim3(nchunk, N), dim3(256), 0, st, x, dy, HW, C, ldx, lddy, cpg, groups, ppb, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, gpart, cpart, x2, ld2, C1);
    static const int cfg_dxs = CDAE_DEV_INT("CDAE_GN_BWD_STREAM", 1);
    const bool dx_stream = VEC == 4 && cfg_dxs && E <= 256 && (!accumulate_dx || dx);
    // TWO launches (round 6): the streaming dx kernel folds the group sums itself (each thread its own group's <= 64 chunk records) and
    // carries the channel folds — d(scale, shift) per image, dgamma / dbeta over the batch — as extra rows of its grid, one block per 32
    // channels: the separate fold launch (56 per C64 training step, 4.8 us each + a launch boundary) is gone.  CDAE_GN_BWD_FOLD2=0
    // (dev build): the three-launch form
    static const int cfg_fold2 = CDAE_DEV_INT("CDAE_GN_BWD_FOLD2", 1);
    if (dx_stream && cfg_fold2 && cdae_tune(TUNE_GN_BWD_FOLD2)) {
        const int cblocks = (C + 31) / 32, rows_extra = (cblocks + nchunk - 1) / nchunk;
        hipLaunchKernelGGL((gn_bwd_dx_stream_kernel<T>), dim3(nchunk, N + rows_extra), dim3(256), 0, st, x, dy, dx, HW, C, ldx, lddy, lddx, cpg, groups, ppb, mean, rstd,
                           gamma, beta, scale_shift, ld_ss, silu, (const float*)nullptr, accumulate_dx, dx_add, ld_add, dxb_hi, dxb_lo, x2, ld2, C1, dx2, lddx2,
                           N, (const float*)nullptr, dgamma, dbeta, accumulate_params, (const float*)gpart, (const float*)cpart, nchunk, d_scale_shift, ld_dss);
        cdae_prof_end(PROF_GN, st);
        CHECK_LAUNCH("gn_bwd launch failed");
        return 0;
    }
    hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3((C + 63) / 64 + 1, N), dim3(256), 0, st, N, HW, C, cpg, groups, nchunk, gpart, cpart, gsum, d_scale_shift, ld_dss, ncp);
    // pass 2b (dgamma / dbeta over the batch) rides along as one extra row of blocks of the streaming dx launch (it only needs pass 2a's
    // per-image sums, like dx): one launch less per GroupNorm, 56 per C64 training step; CDAE_GN_BWD_PARAM_ROW=0: its own launch
    static const int cfg_prow = CDAE_DEV_INT("CDAE_GN_BWD_PARAM_ROW", 1);
    // (the row has only nchunk blocks, each walking ceil(C / 32 / nchunk) channel blocks x N / 8 images one after the other: at N = 256
    // and one chunk that serial walk was 80 of the 101 us of a 16-pixel GroupNorm backward — then the fold gets its own C / 32 blocks)
    const long prow_serial = (long)(((C + 31) / 32 + nchunk - 1) / nchunk) * ((N + 7) / 8);
    const bool param_row = dx_stream && cfg_prow && N >= 8 && prow_serial <= 64;
    if (param_row) {}
    else if (N >= 8) hipLaunchKernelGGL(gn_bwd_param8_kernel, dim3((C + 31) / 32), dim3(256), 0, st, N, C, ncp, dgamma, dbeta, accumulate_params);
    else hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, st, N, C, ncp, dgamma, dbeta, accumulate_params);
    if (dx_stream)
        hipLaunchKernelGGL((gn_bwd_dx_stream_kernel<T>), dim3(nchunk, param_row ? N + 1 : N), dim3(256), 0, st, x, dy, dx, HW, C, ldx, lddy, lddx, cpg, groups, ppb, mean, rstd,
                           gamma, beta, scale_shift, ld_ss, silu, gsum, accumulate_dx, dx_add, ld_add, dxb_hi, dxb_lo, x2, ld2, C1, dx2, lddx2,
                           param_row ? N : -1, ncp, dgamma, dbeta, accumulate_params);
    else if (VEC == 4) hipLaunchKernelGGL((gn_bwd_dx_kernel<4, T>), dim3(grid_for(total)), dim3(256), 0, st, x, dy, dx, total, HW, C, ldx, lddy, lddx, cpg, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, gsum, accumulate_dx, dx_add, ld_add, dxb_hi, dxb_lo, x2, ld2, C1, dx2, lddx2);
    else hipLaunchKernelGGL((gn_bwd_dx_kernel<1, T>), dim3(grid_for(total)), dim3(256), 0, st, x, dy, dx, total, HW, C, ldx, lddy, lddx, cpg, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, gsum, accumulate_dx, dx_add, ld_add, (unsigned short*)nullptr, (unsigned short*)nullptr, x2, ld2, C1, dx2, lddx2);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_bwd launch failed");
    return 0;
}

extern "C" {

// workspace floats needed by cdae_gn_stats / cdae_gn_bwd
size_t cdae_gn_workspace_floats(int N, int C) {
    return (size_t)N * CDAE_GN_MAX_CHUNKS * ((size_t)32 * 2 + (size_t)C * 4) + (size_t)N * 32 * 2 + (size_t)N * C * 2;
}

int cdae_gn_stats(const float* x, int N, int HW, int C, int ldx, int groups, float eps, float* mean, float* rstd,
                  float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (C % groups) return cdae_fail("gn: C % groups != 0");
    const int cpg = C / groups;
    const int VEC = (cpg % 4 == 0 && ldx % 4 == 0) ? 4 : 1;
    const int E = C / VEC;
    if (E > 256 || groups > 256) return cdae_fail("gn: unsupported channel count for this vector width");
    const int nchunk = gn_chunks(HW, N, E);
    const int ppb = (HW + nchunk - 1) / nchunk;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 4.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_stats N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    if (VEC == 4 && gn_small(N, HW, C, cpg, groups, 4)) {
        hipLaunchKernelGGL((gn_stats_group_kernel<float>), dim3(32, N), dim3(256), 0, st, x, HW, ldx, cpg, eps, mean, rstd, (const float*)nullptr, 0, 0,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, C, (float*)nullptr);
        cdae_prof_end(PROF_GN, st);
        CHECK_LAUNCH("gn_stats launch failed");
        return 0;
    }
    if (VEC == 4) hipLaunchKernelGGL(gn_partial_kernel<4>, dim3(nchunk, N), dim3(256), 0, st, x, HW, C, ldx, cpg, groups, ppb, ws);
    else hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(nchunk, N), dim3(256), 0, st, x, HW, C, ldx, cpg, groups, ppb, ws);
    if (groups == 32) hipLaunchKernelGGL((gn_finalize32_kernel<float>), dim3(N), dim3(256), 0, st, x, HW, ldx, cpg, nchunk, eps, ws, mean, rstd, (const float*)nullptr, 0, 0);
    else hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(groups < 64 ? 64 : groups), 0, st, x, HW, ldx, cpg, groups, nchunk, eps, ws, mean, rstd);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_stats launch failed");
    return 0;
}

int cdae_gn_apply(const float* x, float* y, int N, int HW, int C, int ldx, int ldy, int groups, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int cpg = C / groups;
    const int VEC = (cpg % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0) ? 4 : 1;
    const int E = C / VEC;
    if (E > 256) return cdae_fail("gn_apply: unsupported channel count for this vector width");
    const int rows = 256 / E;
    int nchunk = HW / (rows * 16);             // >= 16 pixels per thread
    if (nchunk < 1) nchunk = 1;
    while (nchunk > 1 && (long)nchunk * N > 4096) nchunk >>= 1;
    const int ppb = (HW + nchunk - 1) / nchunk;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 8.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_apply N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    if (VEC == 4) hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(nchunk, N), dim3(256), 0, st, x, y, HW, C, ldx, ldy, cpg, groups, ppb, mean, rstd, gamma, beta, scale_shift, ld_ss, silu);
    else hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(nchunk, N), dim3(256), 0, st, x, y, HW, C, ldx, ldy, cpg, groups, ppb, mean, rstd, gamma, beta, scale_shift, ld_ss, silu);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_apply launch failed");
    return 0;
}

int cdae_gn_bwd(const float* x, const float* dy, float* dx, int N, int HW, int C, int ldx, int lddy, int lddx, int groups,
                const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu,
                float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift, int ld_dss, int accumulate_dx,
                float* ws, void* stream) {
    return cdae_gn_bwd_ex(x, dy, dx, N, HW, C, ldx, lddy, lddx, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, dgamma, dbeta,
                          accumulate_params, d_scale_shift, ld_dss, accumulate_dx, nullptr, 0, nullptr, nullptr, ws, stream);
}


int cdae_gn_bwd_ex(const float* x, const float* dy, float* dx, int N, int HW, int C, int ldx, int lddy, int lddx, int groups,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu,
                   float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift, int ld_dss, int accumulate_dx,
                   const float* dx_add, int ld_add, unsigned short* dxb_hi, unsigned short* dxb_lo, float* ws, void* stream) {
    return gn_bwd_impl<float>(x, nullptr, 0, C, dy, dx, nullptr, 0, N, HW, C, ldx, lddy, lddx, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, dgamma,
                              dbeta, accumulate_params, d_scale_shift, ld_dss, accumulate_dx, dx_add, ld_add, dxb_hi, dxb_lo, ws, stream);
}

int cdae_gn_bwd_cat(const float* x1, int ld1, const float* x2, int ld2, int C1, const float* dy, int lddy, float* dx1, int lddx1, float* dx2, int lddx2,
                    int N, int HW, int C, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* scale_shift, int ld_ss, int silu, float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift,
                    int ld_dss, int accumulate_dx, float* ws, void* stream) {
    if (!x2 || !dx1 || !dx2 || C1 <= 0 || C1 >= C || C1 % 4 || ld1 % 4 || ld2 % 4 || lddx1 % 4 || lddx2 % 4 || (C / groups) % 4)
        return cdae_fail("gn_bwd_cat: two sources with 4-channel aligned widths and pitches required");
    return gn_bwd_impl<float>(x1, x2, ld2, C1, dy, dx1, dx2, lddx2, N, HW, C, ld1, lddy, lddx1, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, dgamma,
                              dbeta, accumulate_params, d_scale_shift, ld_dss, accumulate_dx, nullptr, 0, nullptr, nullptr, ws, stream);
}

// GroupNorm32 backward on a 16-bit torso (bf16 rows in, bf16 rows out; statistics, folds and parameter gradients fp32): x / x2 the (one
// or two) sources of the normalised tensor, dy the gradient of the norm's output, dx / dx2 its input gradient(s), dx_add a second
// gradient of the same tensor added on the way out (the ResBlock's residual path).
int cdae_gn_bwd16(const void* x, int ldx, const void* x2, int ld2, int C1, const void* dy, int lddy, void* dx, int lddx, void* dx2, int lddx2,
                  int N, int HW, int C, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  const float* scale_shift, int ld_ss, int silu, float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift,
                  int ld_dss, int accumulate_dx, const void* dx_add, int ld_add, float* ws, void* stream) {
    if ((C / groups) % 4 || ldx % 4 || lddy % 4 || lddx % 4 || (x2 && (ld2 % 4 || C1 % 4 || !dx2 || lddx2 % 4)) || (dx_add && ld_add % 4))
        return cdae_fail("gn_bwd16: 4-channel aligned widths and pitches required");
    typedef __bf16 B;
    return gn_bwd_impl<B>((const B*)x, (const B*)x2, ld2, x2 ? C1 : C, (const B*)dy, (B*)dx, (B*)dx2, lddx2, N, HW, C, ldx, lddy, lddx, groups, mean, rstd, gamma,
                          beta, scale_shift, ld_ss, silu, dgamma, dbeta, accumulate_params, d_scale_shift, ld_dss, accumulate_dx, (const B*)dx_add, ld_add,
                          nullptr, nullptr, ws, stream);
}


// GroupNorm32 statistics of a bf16 tensor (one or two sources: a channel concatenation read in place), fp32 sums — what the reference's
// GroupNorm32 computes after its .float() (nn.py:435-437); optionally the folded (a, b) table like cdae_gn_stats2_coef
int cdae_gn_stats16(const void* x1, int ld1, const void* x2, int ld2, int C1, int N, int HW, int C, int groups, float eps, float* mean, float* rstd,
                    const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    typedef __bf16 B;
    if (groups != 32 || C % 32 || (coef && (!gamma || !beta))) return cdae_fail("gn_stats16: GroupNorm32 only (coef needs gamma and beta)");
    const int cpg = C / groups;
    if (!(cpg % 4 == 0 && ld1 % 4 == 0 && (!x2 || (ld2 % 4 == 0 && C1 % 4 == 0))) || C / 4 > 256) return cdae_fail("gn_stats16: needs 4-channel vectors, C <= 1024");
    const int E = C / 4;
    const int nchunk = gn_chunks(HW, N, E);
    const int ppb = (HW + nchunk - 1) / nchunk;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 2.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_stats16 N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    if (gn_small(N, HW, C, cpg, groups, 2)) {
        hipLaunchKernelGGL((gn_stats_group_kernel<B>), dim3(32, N), dim3(256), 0, st, (const B*)x1, HW, ld1, cpg, eps, mean, rstd, (const B*)x2, ld2, x2 ? C1 : C,
                           gamma, beta, scale_shift, ld_ss, C, coef);
        cdae_prof_end(PROF_GN, st);
        CHECK_LAUNCH("gn_stats16 launch failed");
        return 0;
    }
    hipLaunchKernelGGL((gn_partial_kernel<4, B>), dim3(nchunk, N), dim3(256), 0, st, (const B*)x1, HW, C, ld1, cpg, groups, ppb, ws, (const B*)x2, ld2, x2 ? C1 : C);
    hipLaunchKernelGGL((gn_finalize32_kernel<B>), dim3(N), dim3(256), 0, st, (const B*)x1, HW, ld1, cpg, nchunk, eps, ws, mean, rstd, (const B*)x2, ld2, x2 ? C1 : C,
                       gamma, beta, scale_shift, ld_ss, C, coef);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_stats16 launch failed");
    return 0;
}

// y = silu?(GroupNorm(x) * (1 + scale) + shift), bf16 rows in (one or two sources), bf16 rows out: in the 16-bit torso the result IS the
// next conv's operand plane (and the tensor the weight gradient reads back)
int cdae_gn_apply16(const void* x1, int ld1, const void* x2, int ld2, int C1, void* y, int ldy, int N, int HW, int C, int groups, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    typedef __bf16 B;
    const int cpg = C / groups;
    if (!(cpg % 4 == 0 && ld1 % 4 == 0 && ldy % 4 == 0 && (!x2 || (ld2 % 4 == 0 && C1 % 4 == 0))) || C / 4 > 256) return cdae_fail("gn_apply16: needs 4-channel vectors, C <= 1024");
    const int E = C / 4, rows = 256 / E;
    int nchunk = HW / (rows * 16);
    if (nchunk < 1) nchunk = 1;
    while (nchunk > 1 && (long)nchunk * N > 4096) nchunk >>= 1;
    const int ppb = (HW + nchunk - 1) / nchunk;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 4.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_apply16 N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    hipLaunchKernelGGL((gn_apply_kernel<4, false, B>), dim3(nchunk, N), dim3(256), 0, st, (const B*)x1, (B*)y, HW, C, ld1, ldy, cpg, groups, ppb, mean, rstd, gamma, beta,
                       scale_shift, ld_ss, silu, (unsigned short*)nullptr, (unsigned short*)nullptr, (const B*)x2, ld2, x2 ? C1 : C);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_apply16 launch failed");
    return 0;
}

int cdae_gn_apply_split(const float* x, unsigned short* y_hi, unsigned short* y_lo, int N, int HW, int C, int ldx, int ldy, int groups,
                        const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss,
                        int silu, void* stream) {
    return cdae_gn_apply_split2(x, ldx, nullptr, 0, C, y_hi, y_lo, N, HW, C, ldy, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, stream);
}

// GroupNorm (+scale-shift) folded to one affine per (image, channel): coef[n][c] = (a, b) with y = x * a + b — the same arithmetic,
// in the same order, as gn_apply_kernel's per-thread fold, so a consumer that applies it reproduces that kernel bit for bit.
__global__ void gn_coef_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ ss, int ld_ss, int C, int cpg, int G,
                               float* __restrict__ coef, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), c = (int)(i - (long)n * C), g = c / cpg;
        gn_coef_one(mean[n * G + g], rstd[n * G + g], gamma[c], beta[c], ss, (long)n * ld_ss, C, c, coef + i * 2);
    }
}

int cdae_gn_coef(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef,
                 int N, int C, int groups, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (C % groups || (((size_t)coef) & 15)) return cdae_fail("gn_coef: C % groups == 0 and a 16-byte aligned coefficient buffer required");
    const long total = (long)N * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    cdae_prof_begin(PROF_GN, (double)total * 16.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_coef N=%d HW=%d C=%d", (int)N, (int)0, (int)0); cdae_prof_tag(tag); }
    hipLaunchKernelGGL(gn_coef_kernel, dim3(blocks), dim3(256), 0, st, mean, rstd, gamma, beta, scale_shift, ld_ss, C, C / groups, groups, coef, total);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_coef launch failed");
    return 0;
}

int cdae_gn_stats_from_parts(const float* part1, int C1, int nseg1, const float* part2, int C2, int nseg2, int N, int HW, int groups, float eps,
                             float* mean, float* rstd, float* ws /* >= N * C * 4 floats */, void* stream) {
    return cdae_gn_stats_from_parts_coef(part1, C1, nseg1, part2, C2, nseg2, N, HW, groups, eps, mean, rstd, nullptr, nullptr, nullptr, 0, nullptr, ws, stream);
}

int cdae_gn_stats_from_parts_coef(const float* part1, int C1, int nseg1, const float* part2, int C2, int nseg2, int N, int HW, int groups, float eps,
                                  float* mean, float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef,
                                  float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int C = C1 + C2;
    if (nseg1 < 1 || HW % (32 * nseg1) || (part2 && (nseg2 < 1 || HW % (32 * nseg2)))) return cdae_fail("gn_stats_from_parts: HW must be a multiple of 32 per segment");
    if (HW % 32 || C % groups || groups > 256 || (part2 == nullptr) != (C2 == 0) || (((size_t)ws) & 7))
        return cdae_fail("gn_stats_from_parts: HW % 32 == 0, groups <= 256 and an 8-byte aligned workspace required");
    double* chan = reinterpret_cast<double*>(ws);
    cdae_prof_begin(PROF_GN, (double)N * (HW / 32) * C * 8.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_stats_from_parts N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    const int cpg = C / groups;
    static const int cfg_fused = CDAE_DEV_INT("CDAE_GN_PARTS_FUSED", 1);
    if (cfg_fused && cpg <= 32) {
        const int gpb = 32 / cpg;
        hipLaunchKernelGGL(gn_parts_stats_kernel, dim3((groups + gpb - 1) / gpb, N), dim3(256), 0, st, part1, C1, nseg1, part2, C2, part2 ? nseg2 : 1, N, HW,
                           cpg, groups, gpb, eps, mean, rstd, gamma, beta, scale_shift, ld_ss, coef);
    } else {
        if (coef) return cdae_fail("gn_stats_from_parts_coef: the coefficient table needs at most 32 channels per group");
        hipLaunchKernelGGL(gn_parts_channel_kernel, dim3((C + 31) / 32, N), dim3(256), 0, st, part1, C1, nseg1, part2, C2, part2 ? nseg2 : 1, N, HW, chan);
        hipLaunchKernelGGL(gn_parts_group_kernel, dim3(N), dim3(groups < 64 ? 64 : groups), 0, st, chan, C, HW, C / groups, groups, eps, mean, rstd);
    }
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_stats_from_parts launch failed");
    return 0;
}

int cdae_gn_stats2(const float* x1, int ld1, const float* x2, int ld2, int C1, int N, int HW, int C, int groups, float eps, float* mean,
                   float* rstd, float* ws, void* stream) {
    return cdae_gn_stats2_coef(x1, ld1, x2, ld2, C1, N, HW, C, groups, eps, mean, rstd, nullptr, nullptr, nullptr, 0, nullptr, ws, stream);
}

int cdae_gn_stats2_coef(const float* x1, int ld1, const float* x2, int ld2, int C1, int N, int HW, int C, int groups, float eps, float* mean,
                        float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (coef && (groups != 32 || !gamma || !beta)) return cdae_fail("gn_stats2_coef: the coefficient table needs 32 groups, gamma and beta");
    const int cpg = C / groups;
    if (!(cpg % 4 == 0 && ld1 % 4 == 0 && ld2 % 4 == 0 && C1 % 4 == 0) || C / 4 > 256) return cdae_fail("gn_stats2: needs 4-channel vectors, C <= 1024");
    const int E = C / 4;
    const int nchunk = gn_chunks(HW, N, E);
    const int ppb = (HW + nchunk - 1) / nchunk;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 4.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "cdae_gn_stats2 N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    if (gn_small(N, HW, C, cpg, groups, 4)) {
        hipLaunchKernelGGL((gn_stats_group_kernel<float>), dim3(32, N), dim3(256), 0, st, x1, HW, ld1, cpg, eps, mean, rstd, x2, ld2, C1, gamma, beta,
                           scale_shift, ld_ss, C, coef);
        cdae_prof_end(PROF_GN, st);
        CHECK_LAUNCH("gn_stats2 launch failed");
        return 0;
    }
    hipLaunchKernelGGL(gn_partial_kernel<4>, dim3(nchunk, N), dim3(256), 0, st, x1, HW, C, ld1, cpg, groups, ppb, ws, x2, ld2, C1);
    if (groups == 32) hipLaunchKernelGGL((gn_finalize32_kernel<float>), dim3(N), dim3(256), 0, st, x1, HW, ld1, cpg, nchunk, eps, ws, mean, rstd, x2, ld2, C1, gamma, beta,
                                         scale_shift, ld_ss, C, coef);
    else hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(groups < 64 ? 64 : groups), 0, st, x1, HW, ld1, cpg, groups, nchunk, eps, ws, mean, rstd, x2, ld2, C1);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_stats2 launch failed");
    return 0;
}

static int gn_apply_split_impl(const float* x, int ldx, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo,
                               unsigned short* yb_hi, unsigned short* yb_lo, int N, int HW, int C, int ldy, int groups, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream,
                               int plane_gm = 0);

int cdae_gn_apply_split2(const float* x, int ldx, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo, int N, int HW,
                         int C, int ldy, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         const float* scale_shift, int ld_ss, int silu, void* stream) {
    return gn_apply_split_impl(x, ldx, x2, ld2, C1, y_hi, y_lo, nullptr, nullptr, N, HW, C, ldy, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, stream);
}

// cdae_gn_apply_split2 writing the planes GROUP-MAJOR: [C / 16][N * HW][16] instead of [N * HW][C] (operand layout of cdae_conv3x3_fwd_psg, x_gm = 1)
int cdae_gn_apply_split2g(const float* x, int ldx, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo, int N, int HW,
                          int C, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* scale_shift, int ld_ss, int silu, void* stream) {
    return gn_apply_split_impl(x, ldx, x2, ld2, C1, y_hi, y_lo, nullptr, nullptr, N, HW, C, C, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, stream, 1);
}

int cdae_gn_apply_split_train(const float* x, unsigned short* y_hi, unsigned short* y_lo, unsigned short* yb_hi, unsigned short* yb_lo, int N,
                              int HW, int C, int ldx, int ldy, int groups, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream) {
    if (!yb_hi || !yb_lo) return cdae_fail("gn_apply_split_train: both bf16 planes required");
    return gn_apply_split_impl(x, ldx, nullptr, 0, C, y_hi, y_lo, yb_hi, yb_lo, N, HW, C, ldy, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, stream);
}

int cdae_gn_apply_split_train2(const float* x1, int ld1, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo,
                               unsigned short* yb_hi, unsigned short* yb_lo, int N, int HW, int C, int ldy, int groups, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream) {
    if (!yb_hi || !yb_lo) return cdae_fail("gn_apply_split_train2: both bf16 planes required");
    return gn_apply_split_impl(x1, ld1, x2, ld2, C1, y_hi, y_lo, yb_hi, yb_lo, N, HW, C, ldy, groups, mean, rstd, gamma, beta, scale_shift, ld_ss, silu, stream);
}

static int gn_apply_split_impl(const float* x, int ldx, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo,
                               unsigned short* yb_hi, unsigned short* yb_lo, int N, int HW, int C, int ldy, int groups, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream,
                               int plane_gm) {
    hipStream_t st = (hipStream_t)stream;
    if (plane_gm && (C % 16 || ldy != C)) return cdae_fail("gn_apply_split: group-major planes need C % 16 == 0 and dense planes");
    const int cpg = C / groups;
    if (x2 && (ld2 % 4 || C1 % 4)) return cdae_fail("gn_apply_split2: second source needs 4-channel alignment");
    if (!(cpg % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0) || C / 4 > 256) return cdae_fail("gn_apply_split: needs channels-per-group % 4 == 0, C <= 1024");
    const int E = C / 4, rows = 256 / E;
    int nchunk = HW / (rows * 16);
    if (nchunk < 1) nchunk = 1;
    while (nchunk > 1 && (long)nchunk * N > 4096) nchunk >>= 1;
    const int ppb = (HW + nchunk - 1) / nchunk;
    cdae_prof_begin(PROF_GN, (double)N * HW * C * 8.0, st);
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "gn_apply_split_impl N=%d HW=%d C=%d", (int)N, (int)HW, (int)C); cdae_prof_tag(tag); }
    static const int cfg_gmk = CDAE_DEV_INT("CDAE_GN_APPLY_GM", 1);      // 0: group-major planes from the channel-vector kernel
    if (plane_gm && cfg_gmk && !yb_hi && (!x2 || C1 % 16 == 0) && N <= 65535) {
        const int gpb = (C % 32 == 0 && (!x2 || C1 % 32 == 0)) ? 2 : 1;
        int nch = HW / (gpb == 2 ? 128 : 256);       // >= 4 pixels per thread
        if (nch < 1) nch = 1;
        while (nch > 1 && (long)nch * N * (C / 16 / gpb) > 16384) nch >>= 1;
        const int pp = (HW + nch - 1) / nch;
        if (gpb == 2) hipLaunchKernelGGL(gn_apply_gm_kernel<2>, dim3(C / 32, nch, N), dim3(256), 0, st, x, HW, C, ldx, cpg, groups, pp, mean, rstd, gamma, beta,
                                         scale_shift, ld_ss, silu, y_hi, y_lo, x2, ld2, C1);
        else hipLaunchKernelGGL(gn_apply_gm_kernel<1>, dim3(C / 16, nch, N), dim3(256), 0, st, x, HW, C, ldx, cpg, groups, pp, mean, rstd, gamma, beta, scale_shift,
                                ld_ss, silu, y_hi, y_lo, x2, ld2, C1);
    } else
    hipLaunchKernelGGL((gn_apply_kernel<4, true>), dim3(nchunk, N), dim3(256), 0, st, x, (float*)nullptr, HW, C, ldx, ldy, cpg, groups, ppb, mean, rstd,
                       gamma, beta, scale_shift, ld_ss, silu, y_hi, y_lo, x2, ld2, C1, yb_hi, yb_lo, plane_gm);
    cdae_prof_end(PROF_GN, st);
    CHECK_LAUNCH("gn_apply_split launch failed");
    return 0;
}

size_t cdae_bn_workspace_floats(int C) { return (size_t)CDAE_BN_MAX_CHUNKS * C * 2 + (size_t)C * 2; }

int cdae_bn_lrelu_fwd(const float* x, float* y, long rows, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int training, float eps, float momentum, float slope, float* scale, float* shift,
                      float* save_mean, float* save_rstd, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (C > 256) return cdae_fail("bn: C > 256 unsupported");
    int nchunk = 1;
    if (training) {
        nchunk = (int)((rows + 255) / 256)      /* (1024 rows per block left the encoder's first layers on 16-64 blocks: 107 us for an 8 MB pass at batch 256) */;
        if (nchunk > CDAE_BN_MAX_CHUNKS) nchunk = CDAE_BN_MAX_CHUNKS;
        if (nchunk < 1) nchunk = 1;
        int rpb = (int)((rows + nchunk - 1) / nchunk);
        hipLaunchKernelGGL(bn_partial_kernel, dim3(nchunk), dim3(256), 0, st, x, rows, C, rpb, ws);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, st, x, rows, C, nchunk, ws, gamma, beta, running_mean, running_var,
                       training, eps, momentum, scale, shift, save_mean, save_rstd);
    const long total = rows * C;
    hipLaunchKernelGGL(affine_lrelu_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, y, total, C, scale, shift, slope);
    CHECK_LAUNCH("bn_lrelu_fwd launch failed");
    return 0;
}

int cdae_bn_lrelu_bwd(const float* x, const float* dy, float* dx, long rows, int C, const float* gamma, const float* scale,
                      const float* shift, const float* save_mean, const float* save_rstd, float slope, float* dgamma, float* dbeta,
                      int accumulate, float* ws, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (C > 256) return cdae_fail("bn: C > 256 unsupported");
    int nchunk = (int)((rows + 255) / 256)      /* (1024 rows per block left the encoder's first layers on 16-64 blocks: 107 us for an 8 MB pass at batch 256) */;
    if (nchunk > CDAE_BN_MAX_CHUNKS) nchunk = CDAE_BN_MAX_CHUNKS;
    if (nchunk < 1) nchunk = 1;
    int rpb = (int)((rows + nchunk - 1) / nchunk);
    float* sums = ws + (size_t)CDAE_BN_MAX_CHUNKS * C * 2;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nchunk), dim3(256), 0, st, x, dy, rows, C, rpb, save_mean, save_rstd, scale, shift, slope, ws);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, rows, C, nchunk, ws, dgamma, dbeta, accumulate, sums);
    const long total = rows * C;
    hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, dy, dx, total, C, save_mean, save_rstd, gamma, scale, shift, slope, sums);
    CHECK_LAUNCH("bn_lrelu_bwd launch failed");
    return 0;
}

int cdae_softmax_rows(float* s, long rows, int T, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (T > 1024) return cdae_fail("softmax: T > 1024 unsupported");
    cdae_prof_begin(PROF_SOFTMAX, (double)rows * T * 8.0, st);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, s, rows, T);
    cdae_prof_end(PROF_SOFTMAX, st);
    CHECK_LAUNCH("softmax launch failed");
    return 0;
}

int cdae_softmax_rows_bwd(const float* P, float* dP, long rows, int T, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, rows, T);
    CHECK_LAUNCH("softmax_bwd launch failed");
    return 0;
}

}  // extern "C"
