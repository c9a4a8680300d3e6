// elementwise.hip — HBM-bound pointwise kernels of the diffusion hot path (gfx950).
// Built with -ffp-contract=off: the sampler updates must round exactly like the reference's
// separate ATen ops (a*x - b*eps is two roundings + a subtraction, not an FMA), because
// sqrt(1/abar-1) ~ 150 at the first DDIM steps amplifies any difference.
//
// Reference lines: q_sample gaussian_diffusion.py:201-222; p_mean_variance :336-341,355-360;
// ddim_sample :533-558; p_sample :409-414; timestep_embedding nn.py:551-569; reparameterize
// nn.py:460-467; softplus nn.py:108; causal_masking nn.py:290-295; update_ema nn.py:503-513.
#include <hip/hip_runtime.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

namespace {

#define GRID_STRIDE(idx, total) \
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < (total); idx += (long)gridDim.x * blockDim.x)

int grid_for(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    GRID_STRIDE(i, n) { float v = x[i]; y[i] = cdae_silu(v); }
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    GRID_STRIDE(i, n) { float v = x[i], s = cdae_sigmoid(v); dx[i] = dy[i] * s * (1.f + v * (1.f - s)); }
}

// fp32 -> two f16 planes (hi = f16(x), lo = f16(x - hi)): the operand format of the pre-split GEMM kernel (weights, once per version)
typedef _Float16 ew_half4 __attribute__((ext_vector_type(4)));
__global__ void split_f16_kernel(const float4* __restrict__ src, ew_half4* __restrict__ hi, ew_half4* __restrict__ lo, long n4) {
    GRID_STRIDE(i, n4) {
        const float4 v = src[i];
        ew_half4 h, l;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
        l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
        hi[i] = h; lo[i] = l;
    }
}

// ---- per-tensor power-of-two scale of WEIGHT planes.  hi = f16(w), lo = f16(w - hi) loses its second half as soon as |w| < 2^-3: lo is then an
// f16 SUBNORMAL and the pair is fixed point with an LSB of 2^-24 (|w| ~ 0.01: 1e-5 relative, 100x the fp32 epsilon) — and network weights live
// exactly there (+-sqrt(3 / fan_in) <= 0.03 for every 3x3 conv).  So weight planes hold w * 2^k with k = 14 - floor(log2(max |w|)) per tensor
// (scaled maximum in [2^14, 2^15): below the f16 maximum with room for the rounding), which keeps lo normal for every element above
// 2^-18 of the tensor's maximum and bounds the error of ANY element by 2^-39 of that maximum; the contraction's epilogue multiplies by the
// exact 2^-k.  A scale record is two floats {2^k, 2^-k}; max |w| == 0 or a non-finite maximum gives {1, 1} (non-finite weights then surface
// through the range flag as before).
__device__ __forceinline__ float wscale_from_maxbits(unsigned bits) {       // bits of max |w| (sign cleared) -> 2^k
    const int ex = (int)(bits >> 23);
    if (ex == 0 || ex == 255) return 1.f;                                    // zero / subnormal maximum, inf, NaN
    int k = 14 - (ex - 127);
    k = k > 126 ? 126 : (k < -126 ? -126 : k);
    return __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
}

struct WScaleDesc { long off; long n; int chunk0; int pad; };                // chunk0: running sum of ceil(n / WS_CHUNK)
constexpr int WS_CHUNK = 8192;
__global__ __launch_bounds__(256) void wscale_max_kernel(const float* __restrict__ flat, const WScaleDesc* __restrict__ desc, int nw, unsigned* __restrict__ maxbits) {
    int lo = 0, hi = nw - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].chunk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const WScaleDesc d = desc[lo];
    const long c0 = (long)(blockIdx.x - d.chunk0) * WS_CHUNK;
    const long c1 = c0 + WS_CHUNK < d.n ? c0 + WS_CHUNK : d.n;
    const float* w = flat + d.off;
    unsigned m = 0;
    // |w| as an integer: monotone in the magnitude, NaN sorts above inf.  16-byte loads over the aligned middle of the chunk (tensor offsets
    // in the flat buffer are arbitrary), scalar loads for the ends
    const long a0 = c0 + ((4 - ((d.off + c0) & 3)) & 3);                          // first element whose address is 16-byte aligned
    const long a1 = a0 <= c1 ? a0 + ((c1 - a0) & ~3L) : c1;
    for (long i = c0 + threadIdx.x; i < (a0 < c1 ? a0 : c1); i += 256) { const unsigned b = __builtin_bit_cast(unsigned, w[i]) & 0x7FFFFFFFu; m = b > m ? b : m; }
    for (long i = a0 + 4L * threadIdx.x; i + 3 < a1 + 0 && i < a1; i += 1024) {
        const uint4 v = *reinterpret_cast<const uint4*>(w + i);
        const unsigned b0 = v.x & 0x7FFFFFFFu, b1 = v.y & 0x7FFFFFFFu, b2 = v.z & 0x7FFFFFFFu, b3 = v.w & 0x7FFFFFFFu;
        const unsigned p0 = b0 > b1 ? b0 : b1, p1 = b2 > b3 ? b2 : b3, pm = p0 > p1 ? p0 : p1;
        m = pm > m ? pm : m;
    }
    for (long i = (a1 > c0 ? a1 : c0) + threadIdx.x; i < c1; i += 256) { const unsigned b = __builtin_bit_cast(unsigned, w[i]) & 0x7FFFFFFFu; m = b > m ? b : m; }
    for (int o = 32; o; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = wm[i] > m ? wm[i] : m;
        atomicMax(maxbits + lo, m);
    }
}
__global__ __launch_bounds__(256) void wscale_max1_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ maxbits) {      // one tensor
    unsigned m = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned b = __builtin_bit_cast(unsigned, w[i]) & 0x7FFFFFFFu;
        m = b > m ? b : m;
    }
    for (int o = 32; o; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, m);
}
__global__ void wscale_finish_kernel(const unsigned* __restrict__ maxbits, float* __restrict__ rec, int nw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const float s = wscale_from_maxbits(maxbits[i]);
    rec[2 * i] = s; rec[2 * i + 1] = 1.f / s;                                // exact: a power of two
}
// hi = f16(x * 2^k), lo = f16(x * 2^k - hi), the scale read from a record (the multiplication is exact)
__global__ void split_f16w_kernel(const float4* __restrict__ src, const float* __restrict__ rec, ew_half4* __restrict__ hi, ew_half4* __restrict__ lo, long n4) {
    const float sc = rec[0];
    GRID_STRIDE(i, n4) {
        float4 v = src[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        ew_half4 h, l;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
        l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
        hi[i] = h; lo[i] = l;
    }
}

// fp32 -> two bf16 planes (hi = bf16(x), lo = bf16(x - hi)): gradients on the pre-split path (full fp32 range, 16 significand bits)
typedef __bf16 ew_bf4 __attribute__((ext_vector_type(4)));
__global__ void split_bf16_kernel(const float4* __restrict__ src, ew_bf4* __restrict__ hi, ew_bf4* __restrict__ lo, long n4) {
    GRID_STRIDE(i, n4) {
        const float4 v = src[i];
        ew_bf4 h, l;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
        l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
        hi[i] = h; lo[i] = l;
    }
}

// nearest-2x upsample of an NHWC activation written straight as operand planes of the training convs: f16 hi/lo (forward conv) and
// bf16 hi/lo (kept for wgrad), [N][2H][2W][C] dense.  One thread = 4 channels of one input pixel = 4 output pixels.
__global__ void upsample2_split_kernel(const float4* __restrict__ x, ew_half4* __restrict__ fh, ew_half4* __restrict__ fl, ew_bf4* __restrict__ bh,
                                       ew_bf4* __restrict__ bl, int H, int W, int C4, long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        long pix = i / C4;
        const int xx = (int)(pix % W); pix /= W;
        const int yy = (int)(pix % H);
        const long n = pix / H;
        const float4 v = x[i];
        ew_half4 h, l; ew_bf4 b, m;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        l[0] = (_Float16)(v.x - (float)h[0]); l[1] = (_Float16)(v.y - (float)h[1]);
        l[2] = (_Float16)(v.z - (float)h[2]); l[3] = (_Float16)(v.w - (float)h[3]);
        b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
        m[0] = (__bf16)(v.x - (float)b[0]); m[1] = (__bf16)(v.y - (float)b[1]);
        m[2] = (__bf16)(v.z - (float)b[2]); m[3] = (__bf16)(v.w - (float)b[3]);
        const long o = ((n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C4 + c, row = 2L * W * C4;
        fh[o] = h; fh[o + C4] = h; fh[o + row] = h; fh[o + row + C4] = h;
        fl[o] = l; fl[o + C4] = l; fl[o + row] = l; fl[o + row + C4] = l;
        bh[o] = b; bh[o + C4] = b; bh[o + row] = b; bh[o + row + C4] = b;
        bl[o] = m; bl[o + C4] = m; bl[o + row] = m; bl[o + row + C4] = m;
    }
}

// conv3x3 weight (OHWI [Cout][9][Cin]) -> the weight of the dgrad convolution, [Cin][9][Cout] with the taps flipped, as bf16 hi/lo
// planes: dx = conv3x3(dy, this).  One 32x32 (co, ci) tile of one tap per block, transposed through LDS.
__global__ __launch_bounds__(256) void wdgrad_planes_kernel(const float* __restrict__ w, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int Cout, int Cin) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((long)co * 9 + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) {
            const float v = tile[tx][r];
            const __bf16 h = (__bf16)v;
            const long o = ((long)ci * 9 + (8 - tap)) * Cout + co;
            hi[o] = h; lo[o] = (__bf16)(v - (float)h);
        }
    }
}

// All registered conv3x3 weights in ONE launch: f16 hi/lo planes in the weight's own OHWI order (forward operand) and bf16 hi/lo
// planes of the dgrad weight ([Cin][9][Cout], taps flipped).  desc[i] = {weight offset in `flat` (elements), Cout, Cin, first tile};
// a block = one 32 x 32 (co, ci) tile of one tap of one weight; output planes use the same element offsets, relative to the first
// registered weight's offset `base`.
struct WPrepDesc { long off; int Cout, Cin, tile0, flags; };      // flags bit 0: also write the K-group-major planes (Cout % 16 == 0 and Cin % 16 == 0)
__global__ __launch_bounds__(256) void wprep_all_kernel(const float* __restrict__ flat, const WPrepDesc* __restrict__ desc, int nw, long base,
                                                        _Float16* __restrict__ fh, _Float16* __restrict__ fl, __bf16* __restrict__ bh,
                                                        __bf16* __restrict__ bl, _Float16* __restrict__ kfh, _Float16* __restrict__ kfl,
                                                        __bf16* __restrict__ kbh, __bf16* __restrict__ kbl, const float* __restrict__ wscale,
                                                        int lo_as_fwd16 = 0) {
    // lo_as_fwd16 (the 16-bit torso, where the lo planes are never read): bl / kbl receive the bf16 FORWARD weights instead — OHWI order at
    // bl, K-group-major [Cin / 16][9][Cout][16] at kbl — the operand of the bf16 forward conv (activations are bf16 there, so the f16 planes
    // cannot pair with them on the matrix cores)
    // wscale (optional): scale records {2^k, 2^-k} per weight (cdae_weight_scales over the same descriptors' tensors, same order): the f16
    // planes then hold w * 2^k; the bf16 dgrad planes (fp32 exponent range: no subnormal problem) stay unscaled
    // kf* / kb* (optional): the same planes in K-group-major order [K / 16][9][rows][16] (cdae_conv_wpack's layout) at the same offsets, for the
    // weights whose descriptor asks for them (a packed index of a weight with Cin % 16 != 0 would leave the weight's own region)
    __shared__ float tile[32][33];
    int lo = 0, hi = nw - 1;                       // last descriptor whose first tile <= blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const WPrepDesc d = desc[lo];
    const float fsc = wscale ? wscale[2 * (d.flags >> 8)] : 1.f;          // flags: bit 0 packable, bits 8.. the weight's record index
    int t = blockIdx.x - d.tile0;
    const int nci = (d.Cin + 31) >> 5, nco = (d.Cout + 31) >> 5;
    const int cit = t % nci; t /= nci;
    const int cot = t % nco; const int tap = t / nco;
    const float* w = flat + d.off;
    const long o0 = d.off - base;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, ci0 = cit * 32, co0 = cot * 32;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < d.Cout && ci < d.Cin) {
            const long i = ((long)co * 9 + tap) * d.Cin + ci;
            v = w[i];
            const float vs = v * fsc;
            const _Float16 h = (_Float16)vs;
            const _Float16 l = (_Float16)(vs - (float)h);
            fh[o0 + i] = h; fl[o0 + i] = l;
            if (lo_as_fwd16) bl[o0 + i] = (__bf16)v;
            if (kfh && (d.flags & 1)) {
                const long k = o0 + (((long)(ci >> 4) * 9 + tap) * d.Cout + co) * 16 + (ci & 15);
                kfh[k] = h; kfl[k] = l;
                if (lo_as_fwd16) kbl[k] = (__bf16)v;
            }
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < d.Cin && co < d.Cout) {
            const float v = tile[tx][r];
            const __bf16 h = (__bf16)v;
            const long o = o0 + ((long)ci * 9 + (8 - tap)) * d.Cout + co;
            const __bf16 l = (__bf16)(v - (float)h);
            bh[o] = h; if (!lo_as_fwd16) bl[o] = l;
            if (kbh && (d.flags & 1)) {
                const long k = o0 + (((long)(co >> 4) * 9 + (8 - tap)) * d.Cin + ci) * 16 + (co & 15);
                kbh[k] = h; if (!lo_as_fwd16) kbl[k] = l;
            }
        }
    }
}

// W [N][K] (row pitch ldw) -> bf16 hi / lo planes of W^T, [K][N] dense: the weight operand of the streaming dgrad GEMM (skipgn.hip,
// cdae_linear_dgrad_stream: dx = dy @ W reads W^T rows K-contiguous).  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void wt_planes_bf16_kernel(const float* __restrict__ w, long ldw, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int N, int K) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (n0 + r < N && k0 + tx < K) ? w[(long)(n0 + r) * ldw + k0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        if (k < K && n < N) {
            const float v = tile[tx][r];
            const __bf16 h = (__bf16)v;
            hi[(long)k * N + n] = h; lo[(long)k * N + n] = (__bf16)(v - (float)h);
        }
    }
}

// dgrad of a STRIDE-2 conv3x3 (pad 1; the UNet's Downsample, reference unet.py:82-105) as four sub-pixel phases: input pixel (2y + py, 2x + px)
// receives dy[(2y + py + 1 - ky) / 2][...] w[ky] for the ky of matching parity — py = 0: ky = 1 (dy row y); py = 1: ky = 2 (row y) and ky = 0
// (row y + 1).  That is the 2 x 2 window of convwin_kernel's 4-tap form (tap t reads low-res row y - 1 + t / 2 + py): folded weights
// w4[phase][ci][ty][tx][co] = w[co][ky(py, ty)][kx(px, tx)][ci] with ky(0, 0) = none (zero), ky(0, 1) = 1, ky(1, 0) = 2, ky(1, 1) = 0 — as bf16
// hi / lo planes (gradient operand path).  9 of the 16 taps carry weights: 1.78x the multiply-adds of the exact form, against the 4x of the
// masked 9-tap gather the fp32-operand kernel runs.
__global__ void s2dgrad_wfold_kernel(const float* __restrict__ w, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int Cout, int Cin, long total) {
    GRID_STRIDE(i, total) {
        long r = i;
        const int co = (int)(r % Cout); r /= Cout;
        const int t = (int)(r & 3); r >>= 2;
        const int ci = (int)(r % Cin);
        const int ph = (int)(r / Cin);
        const int py = ph >> 1, px = ph & 1, ty = t >> 1, tx = t & 1;
        const int ky = py ? (ty ? 0 : 2) : (ty ? 1 : -1), kx = px ? (tx ? 0 : 2) : (tx ? 1 : -1);
        const float v = (ky >= 0 && kx >= 0) ? w[(((long)co * 3 + ky) * 3 + kx) * Cin + ci] : 0.f;
        const __bf16 h = (__bf16)v;
        hi[i] = h; lo[i] = (__bf16)(v - (float)h);
    }
}

// generic pointwise activations on a stored pre-activation (codes = the GEMM epilogue's: 1 SiLU, 2 LeakyReLU(0.01), 3 ReLU, 4 sigmoid)
__device__ inline float act_apply(float v, int kind) {
    if (kind == 1) return cdae_silu(v);
    if (kind == 2) return v > 0.f ? v : 0.01f * v;
    if (kind == 3) return fmaxf(v, 0.f);
    if (kind == 4) return cdae_sigmoid(v);
    return v;
}
__global__ void act_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int kind) {
    GRID_STRIDE(i, n) { y[i] = act_apply(x[i], kind); }
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n, int kind) {
    GRID_STRIDE(i, n) {
        float v = x[i], g = dy[i], d = 1.f;
        if (kind == 1) { float s = cdae_sigmoid(v); d = s * (1.f + v * (1.f - s)); }
        else if (kind == 2) d = v > 0.f ? 1.f : 0.01f;
        else if (kind == 3) d = v > 0.f ? 1.f : 0.f;
        else if (kind == 4) { float s = cdae_sigmoid(v); d = s * (1.f - s); }
        dx[i] = g * d;
    }
}

// out[n][k] = cos(t[n]*f[k]), out[n][half+k] = sin(t[n]*f[k]); odd dim gets a trailing zero
__global__ void temb_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int N, int dim) {
    const int half = dim / 2;
    GRID_STRIDE(i, (long)N * dim) {
        int n = (int)(i / dim), k = (int)(i - (long)n * dim);
        float v = 0.f;
        if (k < half) v = cosf(t[n] * freqs[k]);
        else if (k < 2 * half) v = sinf(t[n] * freqs[k - half]);
        out[i] = v;
    }
}

// spaced step index -> what the network sees: float(map[t]) * scale, or the integer map itself (as float storage)
__global__ void model_t_kernel(const long long* __restrict__ t, const long long* __restrict__ map, float scale, int rescale,
                               float* __restrict__ out_f, long long* __restrict__ out_i, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    long long m = map[t[i]];
    if (out_i) out_i[i] = m;
    out_f[i] = rescale ? (float)m * scale : (float)m;
}

__global__ void embedding_add_kernel(float* __restrict__ emb, const float* __restrict__ table, const long long* __restrict__ idx, int N, int D) {
    GRID_STRIDE(i, (long)N * D) { int n = (int)(i / D), d = (int)(i - (long)n * D); emb[i] += table[idx[n] * D + d]; }
}
__global__ void embedding_bwd_kernel(const float* __restrict__ demb, float* __restrict__ dtable, const long long* __restrict__ idx, int N, int D) {
    GRID_STRIDE(i, (long)N * D) { int n = (int)(i / D), d = (int)(i - (long)n * D); atomicAdd(&dtable[idx[n] * D + d], demb[i]); }
}

// out = x * m * scale: nn.Dropout in training (reference unet.py:153 `nn.Dropout(p=dropout)`; m = the Bernoulli(1 - p) keep mask as 0 / 1
// floats, scale = 1 / (1 - p)), and its backward with dy in place of x
__global__ void mul_scale_kernel(const float* __restrict__ x, const float* __restrict__ m, float scale, float* __restrict__ out, long n) {
    GRID_STRIDE(i, n) { out[i] = x[i] * m[i] * scale; }
}
__global__ void axpby_kernel(float a, const float* __restrict__ x, float b, const float* __restrict__ y, float* __restrict__ out, long n) {
    GRID_STRIDE(i, n) {
        float r = a * x[i];
        if (y) r = r + b * y[i];
        out[i] = r;
    }
}

__global__ void mul_rows_kernel(float* __restrict__ x, const float* __restrict__ m, int N, int D) {
    GRID_STRIDE(i, (long)N * D) { x[i] *= m[i / D]; }
}

// dst[row*ldd + off + c] = src[row*lds + c]  (channel concat / slice of NHWC tensors)
__global__ void copy2d_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int cols, long lds, long ldd, int accumulate) {
    if ((cols & 3) == 0 && (lds & 3) == 0 && (ldd & 3) == 0 && (((size_t)src | (size_t)dst) & 15) == 0) {
        const int c4n = cols >> 2;
        GRID_STRIDE(i, rows * c4n) {
            long r = i / c4n; int c = (int)(i - r * c4n) * 4;
            float4 v = *reinterpret_cast<const float4*>(src + r * lds + c);
            float4* d = reinterpret_cast<float4*>(dst + r * ldd + c);
            if (accumulate) { float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *d = v;
        }
    } else {
        GRID_STRIDE(i, rows * cols) {
            long r = i / cols; int c = (int)(i - r * cols);
            float v = src[r * lds + c];
            if (accumulate) v += dst[r * ldd + c];
            dst[r * ldd + c] = v;
        }
    }
}

// NCHW <-> NHWC (small channel counts at the model boundary)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
    GRID_STRIDE(i, (long)N * C * HW) {
        int c = (int)(i % C); long r = i / C; int p = (int)(r % HW); int n = (int)(r / HW);
        dst[i] = src[((long)n * C + c) * HW + p];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
    GRID_STRIDE(i, (long)N * C * HW) {
        int p = (int)(i % HW); long r = i / HW; int c = (int)(r % C); int n = (int)(r / C);
        dst[i] = src[((long)n * HW + p) * C + c];
    }
}

// Batch assembly from an HBM-resident u8 image pool (HWC per image): out[b] = pool[idx[b]] / div + shift, NHWC fp32
// (a true IEEE division: ToTensor's x/255 and the image-folder x/127.5-1 are reproduced bit for bit).
// per4 = bytes per image / 4.  One thread expands 4 bytes to one float4; both sides are fully coalesced.
__global__ void gather_u8_kernel(const unsigned* __restrict__ pool, const long long* __restrict__ idx, float4* __restrict__ out,
                                 long per4, long total4, float div, float shift) {
    GRID_STRIDE(i, total4) {
        long b = i / per4, r = i - b * per4;
        unsigned w = pool[idx[b] * per4 + r];
        out[i] = make_float4((float)(w & 255u) / div + shift, (float)((w >> 8) & 255u) / div + shift,
                             (float)((w >> 16) & 255u) / div + shift, (float)(w >> 24) / div + shift);
    }
}

// 2x2 sum pool of an NHWC tensor [N,2H,2W,C] -> [N,H,W,C]  (dgrad of the fused nearest-2x upsample)
__global__ void sumpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C) {
    GRID_STRIDE(i, (long)N * H * W * C) {
        int c = (int)(i % C); long r = i / C; int x = (int)(r % W); r /= W; int y = (int)(r % H); int n = (int)(r / H);
        const float* s = src + (((long)n * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
        dst[i] = (s[0] + s[C]) + (s[(long)2 * W * C] + s[(long)2 * W * C + C]);
    }
}

// the same, four channels per thread (C % 4 == 0, 16-byte aligned tensors): one index decomposition per float4 and 16-byte accesses —
// the scalar form above spends three 64-bit divisions on every element (79 -> ~40 us for the 64 x 64 -> 32 x 32 gradient at batch 32)
// ---- 16-bit torso glue (bf16 NHWC rows; ew_bf4 = 4 channels = 8 bytes)
__global__ void cast_f32_bf16_kernel(const float4* __restrict__ x, ew_bf4* __restrict__ y, long n4) {
    GRID_STRIDE(i, n4) { const float4 v = x[i]; ew_bf4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w; y[i] = o; }
}
__global__ void cast_bf16_f32_kernel(const ew_bf4* __restrict__ x, float4* __restrict__ y, long n4) {
    GRID_STRIDE(i, n4) { const ew_bf4 v = x[i]; y[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
}
__global__ void upsample2_16_kernel(const ew_bf4* __restrict__ x, ew_bf4* __restrict__ y, int H, int W, int C4, long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        long pix = i / C4;
        const int xx = (int)(pix % W); pix /= W;
        const int yy = (int)(pix % H);
        const long n = pix / H;
        const ew_bf4 v = x[i];
        const long o = ((n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * C4 + c, row = 2L * W * C4;
        y[o] = v; y[o + C4] = v; y[o + row] = v; y[o + row + C4] = v;
    }
}
// rows of the 3 x 3 patch matrix of a bf16 NHWC tensor: y[(n, yy, xx)][tap][c] = x[n][yy + ky - 1][xx + kx - 1][c] (zero outside), tap = 3 ky + kx —
// the B operand of a conv's weight gradient as a plain [pixels][9 C] matrix (the 4 x 4 level of the 16-bit torso, where the window wgrad
// kernel's rows are too short: 19 MB at batch 256 instead of two fp32 casts and the scalar-gather implicit GEMM)
typedef unsigned short ew_u16x8 __attribute__((ext_vector_type(8)));
__global__ void im2col3x3_16_kernel(const ew_u16x8* __restrict__ x, ew_u16x8* __restrict__ y, int H, int W, int C8, long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C8);
        long r = i / C8;
        const int tap = (int)(r % 9); r /= 9;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const long n = r / H;
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        ew_u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[((n * H + sy) * W + sx) * C8 + c];
        y[i] = v;
    }
}
__global__ void sumpool2_16_kernel(const ew_bf4* __restrict__ src, ew_bf4* __restrict__ dst, int N, int H, int W, int C4) {
    const long total = (long)N * H * W * C4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4); long r = i / C4; const int x = (int)(r % W); r /= W; const int y = (int)(r % H); const int n = (int)(r / H);
        const ew_bf4* s = src + (((long)n * 2 * H + 2 * y) * 2 * W + 2 * x) * C4 + c;
        const ew_bf4 a = s[0], b = s[C4], d = s[(long)2 * W * C4], e = s[(long)2 * W * C4 + C4];
        ew_bf4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (__bf16)(((float)a[k] + (float)b[k]) + ((float)d[k] + (float)e[k]));
        dst[i] = o;
    }
}
// per (32-row chunk, channel) sums (sum, sum of squares) of a bf16 [M][C] tensor: the format the conv epilogues leave behind
// (GemmParams::gn_part) for producers that cannot (ATen adds of the skip stack, casts), consumed by cdae_gn_stats_from_parts
__global__ __launch_bounds__(256) void gn_parts16_kernel(const __bf16* __restrict__ x, long ldx, float* __restrict__ parts, long M, int C) {
    const int c4n = C >> 2;
    const long total = ((M + 31) >> 5) * c4n;
    GRID_STRIDE(i, total) {
        const long chunk = i / c4n; const int c4 = (int)(i - chunk * c4n);
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
        const long r0 = chunk * 32, r1 = r0 + 32 < M ? r0 + 32 : M;
        for (long r = r0; r < r1; ++r) {
            const ew_bf4 v = *reinterpret_cast<const ew_bf4*>(x + r * ldx + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float f = (float)v[k]; s[k] += f; q[k] += f * f; }
        }
        float* o = parts + (chunk * C + 4 * c4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[2 * k] = s[k]; o[2 * k + 1] = q[k]; }
    }
}

__global__ void sumpool2_v4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int N, int H, int W, int C4) {
    const long total = (long)N * H * W * C4;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4); long r = i / C4; const int x = (int)(r % W); r /= W; const int y = (int)(r % H); const int n = (int)(r / H);
        const float4* s = src + (((long)n * 2 * H + 2 * y) * 2 * W + 2 * x) * C4 + c;
        const float4 a = s[0], b = s[C4], d = s[(long)2 * W * C4], e = s[(long)2 * W * C4 + C4];
        dst[i] = make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
    }
}

// nearest-2x of an fp32 NHWC tensor [N,H,W,C] -> [N,2H,2W,C], times `scale` (1: Upsample without a conv, unet.py:76-78; 0.25: the
// gradient of the 2 x 2 average pool), and the 2 x 2 pool [N,2H,2W,C] -> [N,H,W,C] times `scale` (0.25: Downsample without a conv,
// unet.py:101-103 avg_pool_nd; 1: the gradient of the upsample).  C % 4 == 0 vector forms, scalar otherwise.
__global__ void upsample2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int V, float scale, long total) {
    GRID_STRIDE(i, total) {
        const int cv = C / V, c = (int)(i % cv);
        long pix = i / cv;
        const int xx = (int)(pix % W); pix /= W;
        const int yy = (int)(pix % H);
        const long n = pix / H;
        const long o = (((n * 2 * H + 2 * yy) * 2 * W + 2 * xx) * cv + c) * V, row = 2L * W * C;
        if (V == 4) {
            float4 v = *reinterpret_cast<const float4*>(x + i * 4);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            *reinterpret_cast<float4*>(y + o) = v; *reinterpret_cast<float4*>(y + o + C) = v;
            *reinterpret_cast<float4*>(y + o + row) = v; *reinterpret_cast<float4*>(y + o + row + C) = v;
        } else {
            const float v = x[i] * scale;
            y[o] = v; y[o + C] = v; y[o + row] = v; y[o + row + C] = v;
        }
    }
}
__global__ void pool2_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C, int V, float scale, long total) {
    GRID_STRIDE(i, total) {
        const int cv = C / V, c = (int)(i % cv);
        long r = i / cv;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long n = r / H;
        const float* s = src + (((n * 2 * H + 2 * y) * 2 * W + 2 * x) * cv + c) * V;
        const long row = 2L * W * C;
        if (V == 4) {
            const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + C), d = *reinterpret_cast<const float4*>(s + row),
                         e = *reinterpret_cast<const float4*>(s + row + C);
            *reinterpret_cast<float4*>(dst + i * 4) = make_float4(((a.x + b.x) + (d.x + e.x)) * scale, ((a.y + b.y) + (d.y + e.y)) * scale,
                                                                  ((a.z + b.z) + (d.z + e.z)) * scale, ((a.w + b.w) + (d.w + e.w)) * scale);
        } else dst[i] = ((s[0] + s[C]) + (s[row] + s[row + C])) * scale;
    }
}

// ---- sampler math.  tab = fp32 copies of the f64 tables (rounded exactly like `.float()`), [NTAB][T]
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const long long* __restrict__ t,
                                const float* __restrict__ tab, int T, float* __restrict__ out, long per_sample, long total) {
    GRID_STRIDE(i, total) {
        long long tt = t[i / per_sample];
        float a = tab[CDAE_TAB_SQRT_AC * T + tt], b = tab[CDAE_TAB_SQRT_1MAC * T + tt];
        out[i] = a * x0[i] + b * noise[i];
    }
}

__global__ void ddim_update_kernel(const float* __restrict__ x, const float* __restrict__ eps, const long long* __restrict__ t,
                                   const float* __restrict__ tab, int T, float eta, const float* __restrict__ noise, int clip,
                                   float* __restrict__ sample, float* __restrict__ pred_xstart, long per_sample, long total) {
    GRID_STRIDE(i, total) {
        long long tt = t[i / per_sample];
        float c1 = tab[CDAE_TAB_SQRT_RECIP_AC * T + tt], c2 = tab[CDAE_TAB_SQRT_RECIPM1_AC * T + tt];
        float ab = tab[CDAE_TAB_AC * T + tt], abp = tab[CDAE_TAB_AC_PREV * T + tt];
        float xv = x[i];
        float x0 = (clip & 2) ? eps[i] : c1 * xv - c2 * eps[i];       // bit 1: `eps` already holds the processed pred_xstart
        if (clip & 1) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float e2 = (c1 * xv - x0) / c2;
        float sigma = eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
        float mean = x0 * sqrtf(abp) + sqrtf(1.f - abp - sigma * sigma) * e2;
        float nz = tt != 0 ? 1.f : 0.f;
        float nv = noise ? noise[i] : 0.f;
        sample[i] = mean + nz * sigma * nv;
        if (pred_xstart) pred_xstart[i] = x0;
    }
}

__global__ void ddpm_update_kernel(const float* __restrict__ x, const float* __restrict__ eps, const long long* __restrict__ t,
                                   const float* __restrict__ tab, int T, const float* __restrict__ noise, int clip,
                                   float* __restrict__ sample, float* __restrict__ pred_xstart, long per_sample, long total) {
    GRID_STRIDE(i, total) {
        long long tt = t[i / per_sample];
        float c1 = tab[CDAE_TAB_SQRT_RECIP_AC * T + tt], c2 = tab[CDAE_TAB_SQRT_RECIPM1_AC * T + tt];
        float k1 = tab[CDAE_TAB_POST_COEF1 * T + tt], k2 = tab[CDAE_TAB_POST_COEF2 * T + tt];
        float lv = tab[CDAE_TAB_MODEL_LOGVAR * T + tt];
        float xv = x[i];
        float x0 = c1 * xv - c2 * eps[i];
        if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float mean = k1 * x0 + k2 * xv;
        float nz = tt != 0 ? 1.f : 0.f;
        sample[i] = mean + nz * expf(0.5f * lv) * noise[i];
        if (pred_xstart) pred_xstart[i] = x0;
    }
}

// ---- causal encoder glue
// mu/var head: var = softplus(v) + 1e-8 (threshold 20 like F.softplus)
__global__ void softplus_eps_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float add) {
    GRID_STRIDE(i, n) { float v = x[i]; y[i] = (v > 20.f ? v : log1pf(expf(v))) + add; }
}
__global__ void softplus_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
    GRID_STRIDE(i, n) { float v = x[i]; dx[i] = dy[i] * (v > 20.f ? 1.f : 1.f / (1.f + expf(-v))); }
}
// z = m + sqrt(v*vscale) * eps
__global__ void reparam_kernel(const float* __restrict__ m, const float* __restrict__ v, float vscale, const float* __restrict__ eps,
                               float* __restrict__ z, long n) {
    GRID_STRIDE(i, n) { z[i] = m[i] + sqrtf(v[i] * vscale) * eps[i]; }
}
// z_pre[n][i][:] = sum_j A[j][i] * u[n][j][:]   (A^T u);  transpose=1 gives A u (the backward)
__global__ void causal_mask_kernel(const float* __restrict__ u, const float* __restrict__ A, float* __restrict__ out, int N, int nv, int d, int transpose) {
    GRID_STRIDE(idx, (long)N * nv * d) {
        int k = (int)(idx % d); long r = idx / d; int i = (int)(r % nv); int n = (int)(r / nv);
        float s = 0.f;
        for (int j = 0; j < nv; ++j) {
            float a = transpose ? A[i * nv + j] : A[j * nv + i];
            s += a * u[((long)n * nv + j) * d + k];
        }
        out[idx] = s;
    }
}

// ---- optimizer: AdamW (decoupled decay, bias correction) + EMA over one flat fp32 buffer (train_util.py:292-297, nn.py:503-513)
__global__ void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 float* __restrict__ ema, long n, float lr, float b1, float b2, float eps, float wd,
                                 float bc1, float sqrt_bc2, float ema_rate, float grad_scale) {
    GRID_STRIDE(i, n) {
        float gi = g[i] * grad_scale;
        float pi = p[i] * (1.f - lr * wd);
        float mi = m[i] * b1 + (1.f - b1) * gi;
        float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) / sqrt_bc2 + eps;
        pi = pi - (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema) ema[i] = ema[i] * ema_rate + (1.f - ema_rate) * pi;
    }
}

// the same update with up to four EMA buffers (TrainLoop's "--ema_rate 0.999,0.9999"): every further rate was an ATen mul_ + add_ pass
// over the flat buffer (16 B / parameter each) behind the optimizer kernel
struct EmaSet { float* buf[4]; float rate[4]; int n; };
__global__ void adamw_ema_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                       EmaSet es, long n, float lr, float b1, float b2, float eps, float wd,
                                       float bc1, float sqrt_bc2, float grad_scale) {
    GRID_STRIDE(i, n) {
        float gi = g[i] * grad_scale;
        float pi = p[i] * (1.f - lr * wd);
        float mi = m[i] * b1 + (1.f - b1) * gi;
        float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) / sqrt_bc2 + eps;
        pi = pi - (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < es.n) es.buf[k][i] = es.buf[k][i] * es.rate[k] + (1.f - es.rate[k]) * pi;
    }
}

__global__ void sqsum_kernel(const float* __restrict__ x, long n, double* __restrict__ out) {
    double s = 0.0;
    GRID_STRIDE(i, n) { double v = x[i]; s += v * v; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}


// ---- learned-variance / variational-bound branches (gaussian_diffusion.py:289-303, 682-715; losses.py) -------------------
// mean_type: 0 = model predicts eps, 1 = model predicts x0.  var_type: 0 = fixed (table rows 8/9), 1 = LEARNED (output is the
// log-variance), 2 = LEARNED_RANGE (output in [-1,1] interpolates log beta_t .. clipped posterior log-variance).
struct PmvCoef { float c1, c2, k1, k2, lv_fixed, v_fixed, min_log, max_log; };
__device__ inline PmvCoef pmv_coef(const float* __restrict__ tab, int T, long long tt) {
    PmvCoef c;
    c.c1 = tab[CDAE_TAB_SQRT_RECIP_AC * T + tt]; c.c2 = tab[CDAE_TAB_SQRT_RECIPM1_AC * T + tt];
    c.k1 = tab[CDAE_TAB_POST_COEF1 * T + tt];    c.k2 = tab[CDAE_TAB_POST_COEF2 * T + tt];
    c.lv_fixed = tab[CDAE_TAB_MODEL_LOGVAR * T + tt]; c.v_fixed = tab[CDAE_TAB_MODEL_VAR * T + tt];
    c.min_log = tab[CDAE_TAB_POST_LOGVAR_CLIPPED * T + tt]; c.max_log = tab[CDAE_TAB_LOG_BETAS * T + tt];
    return c;
}
// returns: x0 (processed), mean, lv; `inside` = 1 when the clamp passed the value through (gradient gate)
__device__ inline void pmv_elem(float xv, float mo, float mv, const PmvCoef& c, int mean_type, int var_type, int clip,
                                float& x0, float& mean, float& lv, float& inside) {
    if (var_type == 1) lv = mv;
    else if (var_type == 2) { float frac = (mv + 1.f) / 2.f; lv = frac * c.max_log + (1.f - frac) * c.min_log; }
    else lv = c.lv_fixed;
    x0 = mean_type == 1 ? mo : c.c1 * xv - c.c2 * mo;
    inside = 1.f;
    if (clip) { inside = (x0 >= -1.f && x0 <= 1.f) ? 1.f : 0.f; x0 = fminf(fmaxf(x0, -1.f), 1.f); }
    mean = c.k1 * x0 + c.k2 * xv;
}

// p_mean_variance (+ optional ancestral sample).  out2: model output, per sample `ostride` floats: [mean part | var part].
__global__ void pmv_kernel(const float* __restrict__ x, const float* __restrict__ out2, long ostride, const long long* __restrict__ t,
                           const float* __restrict__ tab, int T, int mean_type, int var_type, int clip, const float* __restrict__ noise,
                           float* __restrict__ mean_o, float* __restrict__ var_o, float* __restrict__ lv_o, float* __restrict__ x0_o,
                           float* __restrict__ sample_o, long per, long total) {
    GRID_STRIDE(i, total) {
        long n = i / per, e = i - n * per;
        long long tt = t[n];
        PmvCoef c = pmv_coef(tab, T, tt);
        float mo = out2[n * ostride + e], mv = var_type ? out2[n * ostride + per + e] : 0.f;
        float x0, mean, lv, inside;
        pmv_elem(x[i], mo, mv, c, mean_type, var_type, clip, x0, mean, lv, inside);
        if (mean_o) mean_o[i] = mean;
        if (var_o) var_o[i] = var_type ? expf(lv) : c.v_fixed;
        if (lv_o) lv_o[i] = lv;
        if (x0_o) x0_o[i] = x0;
        if (sample_o) sample_o[i] = mean + (tt != 0 ? 1.f : 0.f) * expf(0.5f * lv) * noise[i];
    }
}

#define CDF_K 0.7978845608028654f          /* sqrt(2/pi) */
__device__ inline float approx_cdf(float v) { return 0.5f * (1.f + tanhf(CDF_K * (v + 0.044715f * (v * v * v)))); }
// d/dv of approx_cdf
__device__ inline float approx_pdf(float v) {
    float th = tanhf(CDF_K * (v + 0.044715f * (v * v * v)));
    return 0.5f * (1.f - th * th) * CDF_K * (1.f + 3.f * 0.044715f * v * v);
}

// one element of the bound: KL(q(x_{t-1}|x_t,x_0) || p) for t > 0, -log p(x_0 | x_1) (discretised Gaussian) at t == 0, in nats;
// optionally its derivatives with respect to the model mean and log-variance.
__device__ inline float vb_elem(float x0v, float xv, float mean, float lv, const PmvCoef& c, bool first, bool want_grad, float& dmean, float& dlv) {
    if (!first) {
        float m1 = c.k1 * x0v + c.k2 * xv, lv1 = c.min_log;
        float d = m1 - mean, e1 = expf(lv1 - lv), e2 = expf(-lv);
        if (want_grad) { dmean = -d * e2; dlv = 0.5f * (1.f - e1 - d * d * e2); }
        return 0.5f * (-1.f + lv - lv1 + e1 + d * d * e2);
    }
    float cen = x0v - mean, inv = expf(-(0.5f * lv));
    float pin = inv * (cen + 1.f / 255.f), nin = inv * (cen - 1.f / 255.f);
    float cp = approx_cdf(pin), cm = approx_cdf(nin);
    float lp, dp = 0.f, dn = 0.f;             // d lp / d pin, d lp / d nin
    if (x0v < -0.999f)      { lp = logf(fmaxf(cp, 1e-12f));        if (want_grad && cp >= 1e-12f) dp = approx_pdf(pin) / cp; }
    else if (x0v > 0.999f)  { float q = 1.f - cm; lp = logf(fmaxf(q, 1e-12f)); if (want_grad && q >= 1e-12f) dn = -approx_pdf(nin) / q; }
    else { float dl = cp - cm; lp = logf(fmaxf(dl, 1e-12f)); if (want_grad && dl >= 1e-12f) { dp = approx_pdf(pin) / dl; dn = -approx_pdf(nin) / dl; } }
    if (want_grad) {                         // nll = -lp; d pin/d mean = d nin/d mean = -inv; d pin/d lv = -pin/2, d nin/d lv = -nin/2
        dmean = (dp + dn) * inv;
        dlv = 0.5f * (dp * pin + dn * nin);
    }
    return -lp;
}

// vb[n] = mean over the sample of vb_elem / ln 2 (bits per dim).  One block per sample; also writes pred_xstart if asked.
__global__ void vb_terms_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ out2, long ostride,
                                const long long* __restrict__ t, const float* __restrict__ tab, int T, int mean_type, int var_type, int clip,
                                float* __restrict__ vb, float* __restrict__ x0_o, long per) {
    const long n = blockIdx.x;
    long long tt = t[n];
    PmvCoef c = pmv_coef(tab, T, tt);
    float s = 0.f;
    for (long e = threadIdx.x; e < per; e += blockDim.x) {
        float mo = out2[n * ostride + e], mv = var_type ? out2[n * ostride + per + e] : 0.f;
        float p0, mean, lv, inside, dm, dl;
        pmv_elem(x[n * per + e], mo, mv, c, mean_type, var_type, clip, p0, mean, lv, inside);
        s += vb_elem(x0[n * per + e], x[n * per + e], mean, lv, c, tt == 0, false, dm, dl);
        if (x0_o) x0_o[n * per + e] = p0;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) vb[n] = (((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)per) / 0.6931471805599453f;
}

// d vb / d model output (both halves; the mean half is zero-filled when the mean is frozen, gaussian_diffusion.py:822-825)
__global__ void vb_terms_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ out2, long ostride,
                                    const long long* __restrict__ t, const float* __restrict__ tab, int T, int mean_type, int var_type, int clip,
                                    int freeze_mean, const float* __restrict__ gout, float* __restrict__ dout2, long per, long total) {
    GRID_STRIDE(i, total) {
        long n = i / per, e = i - n * per;
        long long tt = t[n];
        PmvCoef c = pmv_coef(tab, T, tt);
        float mo = out2[n * ostride + e], mv = var_type ? out2[n * ostride + per + e] : 0.f;
        float p0, mean, lv, inside, dm, dl;
        pmv_elem(x[i], mo, mv, c, mean_type, var_type, clip, p0, mean, lv, inside);
        vb_elem(x0[i], x[i], mean, lv, c, tt == 0, true, dm, dl);
        float g = gout[n] / ((float)per * 0.6931471805599453f);
        float dmo = 0.f;
        if (!freeze_mean) dmo = g * dm * c.k1 * inside * (mean_type == 1 ? 1.f : -c.c2);
        dout2[n * ostride + e] = dmo;
        if (var_type) dout2[n * ostride + per + e] = g * dl * (var_type == 2 ? 0.5f * (c.max_log - c.min_log) : 1.f);
    }
}

// per-sample mean of (a-b)^2 over `per` elements: one block per sample
__global__ void mse_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long per) {
    const long n = blockIdx.x;
    float s = 0.f;
    for (long i = threadIdx.x; i < per; i += blockDim.x) { float d = a[n * per + i] - b[n * per + i]; s += d * d; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[n] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)per;
}
// d/d b of mean((a-b)^2) * gout[n]:  -2 (a-b) gout[n] / per
__global__ void mse_rows_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout,
                                    float* __restrict__ db, long per, long total) {
    GRID_STRIDE(i, total) { db[i] = -2.f * (a[i] - b[i]) * gout[i / per] / (float)per; }
}

// representation loss (reference gaussian_diffusion.py:727-766 with nn.py:440-457): per sample
//   kld[n] = sum_j 0.5 (-log var + var + mu^2 - 1)  [KL(N(mu, var) || N(0, I))]  (+ sum_i sum_{j in slice i} 0.5 (z_post - c_i)^2
//   [KL(N(z_post_i, I) || N(c_i, I)): the prior mean of variable i is its label c_i, reference :718-725 with scale [[0, 1]]])
// as ONE kernel instead of ~50 ATen launches (and as many autograd nodes) per direction.  One wave per sample; j ascending per lane.
__global__ void rep_loss_kernel(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ zp, const float* __restrict__ c,
                                float* __restrict__ out, int D, int nv) {
    const int n = blockIdx.x, d = nv > 0 ? D / nv : D;
    float s = 0.f;
    for (int j = threadIdx.x; j < D; j += 64) {
        const float m = mu[(long)n * D + j], v = var[(long)n * D + j];
        float t = 0.5f * (((0.f - logf(v)) + v + m * m) - 1.f);
        if (zp) { const float df = zp[(long)n * D + j] - c[n * nv + j / d]; t += 0.5f * (((0.f - 0.f) + 1.f + df * df) - 1.f); }
        s += t;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[n] = s;
}
__global__ void rep_loss_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ zp, const float* __restrict__ c,
                                    const float* __restrict__ g, float* __restrict__ dmu, float* __restrict__ dvar, float* __restrict__ dzp, int D, int nv, long total) {
    const int d = nv > 0 ? D / nv : D;
    GRID_STRIDE(i, total) {
        const int n = (int)(i / D), j = (int)(i - (long)n * D);
        const float gn = g[n], v = var[i];
        dmu[i] = gn * mu[i];
        dvar[i] = gn * 0.5f * (1.f - 1.f / v);
        if (dzp) dzp[i] = gn * (zp[i] - c[n * nv + j / d]);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)
#define LAUNCH1D(kernel, total, ...) do { \
    cdae_prof_begin(PROF_ELEMWISE, 0.0, ST); \
    hipLaunchKernelGGL(kernel, dim3(grid_for(total)), dim3(256), 0, ST, __VA_ARGS__); \
    cdae_prof_end(PROF_ELEMWISE, ST); \
    if (hipGetLastError() != hipSuccess) return cdae_fail(#kernel " launch failed"); \
    return 0; } while (0)

extern "C" {

int cdae_split_f16(const float* src, unsigned short* hi, unsigned short* lo, long n, void* stream) {
    if (n % 4 || (((size_t)src | (size_t)hi | (size_t)lo) & 7)) return cdae_fail("split_f16: n % 4 == 0 and 8-byte aligned planes required");
    LAUNCH1D(split_f16_kernel, n / 4, (const float4*)src, (ew_half4*)hi, (ew_half4*)lo, n / 4);
}
int cdae_split_f16w(const float* src, const float* w_scale, unsigned short* hi, unsigned short* lo, long n, void* stream) {
    if (!w_scale) return cdae_split_f16(src, hi, lo, n, stream);
    if (n % 4 || (((size_t)src | (size_t)hi | (size_t)lo) & 7)) return cdae_fail("split_f16w: n % 4 == 0 and 8-byte aligned planes required");
    LAUNCH1D(split_f16w_kernel, n / 4, (const float4*)src, w_scale, (ew_half4*)hi, (ew_half4*)lo, n / 4);
}
int cdae_weight_scales(const float* flat, const void* desc, int nw, int total_chunks, float* records, unsigned* scratch, void* stream) {
    if (nw <= 0) return 0;
    if (!flat || !desc || !records || !scratch || total_chunks <= 0) return cdae_fail("weight_scales: flat, desc, records and scratch (nw uint32) required");
    if (hipMemsetAsync(scratch, 0, sizeof(unsigned) * nw, ST) != hipSuccess) return cdae_fail("weight_scales: memset failed");
    cdae_prof_begin(PROF_ELEMWISE, 0.0, ST);
    hipLaunchKernelGGL(wscale_max_kernel, dim3(total_chunks), dim3(256), 0, ST, flat, (const WScaleDesc*)desc, nw, scratch);
    hipLaunchKernelGGL(wscale_finish_kernel, dim3((nw + 255) / 256), dim3(256), 0, ST, scratch, records, nw);
    cdae_prof_end(PROF_ELEMWISE, ST);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("weight_scales launch failed");
}
int cdae_weight_scales_chunk(void) { return WS_CHUNK; }
int cdae_weight_scale1(const float* w, long n, float* record, unsigned* scratch, void* stream) {
    if (!w || n <= 0 || !record || !scratch) return cdae_fail("weight_scale1: w, record (2 floats) and scratch (1 uint32) required");
    if (hipMemsetAsync(scratch, 0, sizeof(unsigned), ST) != hipSuccess) return cdae_fail("weight_scale1: memset failed");
    hipLaunchKernelGGL(wscale_max1_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, ST, w, n, scratch);
    hipLaunchKernelGGL(wscale_finish_kernel, dim3(1), dim3(64), 0, ST, scratch, record, 1);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("weight_scale1 launch failed");
}
int cdae_split_bf16(const float* src, unsigned short* hi, unsigned short* lo, long n, void* stream) {
    if (n % 4 || (((size_t)src & 15) | (((size_t)hi | (size_t)lo) & 7))) return cdae_fail("split_bf16: n % 4 == 0, 16-byte aligned source and 8-byte aligned planes required");
    LAUNCH1D(split_bf16_kernel, n / 4, (const float4*)src, (ew_bf4*)hi, (ew_bf4*)lo, n / 4);
}
int cdae_upsample2_split(const float* x, unsigned short* f_hi, unsigned short* f_lo, unsigned short* b_hi, unsigned short* b_lo, int N, int H,
                         int W, int C, void* stream) {
    if (C % 4 || (((size_t)x) & 15) || (((size_t)f_hi | (size_t)f_lo | (size_t)b_hi | (size_t)b_lo) & 7))
        return cdae_fail("upsample2_split: C % 4 == 0, 16-byte aligned input and 8-byte aligned planes required");
    const long total = (long)N * H * W * (C / 4);
    LAUNCH1D(upsample2_split_kernel, total, (const float4*)x, (ew_half4*)f_hi, (ew_half4*)f_lo, (ew_bf4*)b_hi, (ew_bf4*)b_lo, H, W, C / 4, total);
}
int cdae_wprep_all(const float* flat, const void* desc, int nw, int total_tiles, long base, unsigned short* f_hi, unsigned short* f_lo,
                   unsigned short* b_hi, unsigned short* b_lo, void* stream) {
    if (nw <= 0 || total_tiles <= 0) return 0;
    return cdae_wprep_all_k(flat, desc, nw, total_tiles, base, f_hi, f_lo, b_hi, b_lo, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
// + the K-group-major copies for the second-generation window kernel (all four NULL, or all four given; Cin, Cout % 16 == 0 then)
int cdae_wprep_all_k(const float* flat, const void* desc, int nw, int total_tiles, long base, unsigned short* f_hi, unsigned short* f_lo,
                     unsigned short* b_hi, unsigned short* b_lo, unsigned short* kf_hi, unsigned short* kf_lo, unsigned short* kb_hi,
                     unsigned short* kb_lo, const float* w_scales, void* stream) {
    if (nw <= 0 || total_tiles <= 0) return 0;
    if ((kf_hi || kf_lo || kb_hi || kb_lo) && !(kf_hi && kf_lo && kb_hi && kb_lo)) return cdae_fail("wprep_all_k: give all four packed planes or none");
    hipLaunchKernelGGL(wprep_all_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, flat, (const WPrepDesc*)desc, nw, base, (_Float16*)f_hi,
                       (_Float16*)f_lo, (__bf16*)b_hi, (__bf16*)b_lo, (_Float16*)kf_hi, (_Float16*)kf_lo, (__bf16*)kb_hi, (__bf16*)kb_lo, w_scales);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("wprep_all launch failed");
}
// cdae_wprep_all_k for the 16-bit torso: the lo planes of the bf16 dgrad weights (never read by the one-plane kernels) receive the bf16
// FORWARD weights instead — OHWI order at b_lo, K-group-major at kb_lo
int cdae_wprep_all_m16(const float* flat, const void* desc, int nw, int total_tiles, long base, unsigned short* f_hi, unsigned short* f_lo,
                       unsigned short* b_hi, unsigned short* b_lo, unsigned short* kf_hi, unsigned short* kf_lo, unsigned short* kb_hi,
                       unsigned short* kb_lo, const float* w_scales, void* stream) {
    if (nw <= 0 || total_tiles <= 0) return 0;
    if (!(kf_hi && kf_lo && kb_hi && kb_lo)) return cdae_fail("wprep_all_m16: all packed planes required");
    hipLaunchKernelGGL(wprep_all_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, flat, (const WPrepDesc*)desc, nw, base, (_Float16*)f_hi,
                       (_Float16*)f_lo, (__bf16*)b_hi, (__bf16*)b_lo, (_Float16*)kf_hi, (_Float16*)kf_lo, (__bf16*)kb_hi, (__bf16*)kb_lo, w_scales, 1);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("wprep_all_m16 launch failed");
}
int cdae_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
    if (n % 4 || (((size_t)x) & 15) || (((size_t)y) & 7)) return cdae_fail("cast_f32_bf16: n % 4 == 0 and aligned buffers required");
    LAUNCH1D(cast_f32_bf16_kernel, n / 4, (const float4*)x, (ew_bf4*)y, n / 4);
}
int cdae_cast_bf16_f32(const void* x, float* y, long n, void* stream) {
    if (n % 4 || (((size_t)y) & 15) || (((size_t)x) & 7)) return cdae_fail("cast_bf16_f32: n % 4 == 0 and aligned buffers required");
    LAUNCH1D(cast_bf16_f32_kernel, n / 4, (const ew_bf4*)x, (float4*)y, n / 4);
}
int cdae_upsample2_16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
    if (C % 4) return cdae_fail("upsample2_16: C % 4 == 0 required");
    const long total = (long)N * H * W * (C / 4);
    LAUNCH1D(upsample2_16_kernel, total, (const ew_bf4*)x, (ew_bf4*)y, H, W, C / 4, total);
}
int cdae_im2col3x3_16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
    if (C % 8 || (((size_t)x | (size_t)y) & 15)) return cdae_fail("im2col3x3_16: C % 8 == 0 and 16-byte aligned buffers required");
    const long total = (long)N * H * W * 9 * (C / 8);
    LAUNCH1D(im2col3x3_16_kernel, total, (const ew_u16x8*)x, (ew_u16x8*)y, H, W, C / 8, total);
}
int cdae_sumpool2_16(const void* src, void* dst, int N, int H, int W, int C, void* stream) {
    if (C % 4) return cdae_fail("sumpool2_16: C % 4 == 0 required");
    LAUNCH1D(sumpool2_16_kernel, (long)N * H * W * (C / 4), (const ew_bf4*)src, (ew_bf4*)dst, N, H, W, C / 4);
}
int cdae_gn_parts16(const void* x, long ldx, float* parts, long M, int C, void* stream) {
    if (C % 4 || ldx % 4) return cdae_fail("gn_parts16: C % 4 == 0 required");
    LAUNCH1D(gn_parts16_kernel, ((M + 31) / 32) * (C / 4), (const __bf16*)x, ldx, parts, M, C);
}
int cdae_wt_planes_bf16(const float* w, long ldw, unsigned short* hi, unsigned short* lo, int N, int K, void* stream) {
    if (N <= 0 || K <= 0 || !w || !hi || !lo) return cdae_fail("wt_planes_bf16: empty weight");
    hipLaunchKernelGGL(wt_planes_bf16_kernel, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, ST, w, ldw, (__bf16*)hi, (__bf16*)lo, N, K);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("wt_planes_bf16 launch failed");
}
int cdae_s2dgrad_wfold(const float* w, unsigned short* hi, unsigned short* lo, int Cout, int Cin, void* stream) {
    if (Cout <= 0 || Cin <= 0 || !w || !hi || !lo) return cdae_fail("s2dgrad_wfold: empty weight");
    const long total = 16L * Cin * Cout;
    LAUNCH1D(s2dgrad_wfold_kernel, total, w, (__bf16*)hi, (__bf16*)lo, Cout, Cin, total);
}
int cdae_wdgrad_planes(const float* w, unsigned short* hi, unsigned short* lo, int Cout, int Cin, void* stream) {
    if (Cout <= 0 || Cin <= 0) return cdae_fail("wdgrad_planes: empty weight");
    hipLaunchKernelGGL(wdgrad_planes_kernel, dim3((Cin + 31) / 32, (Cout + 31) / 32, 9), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)hi, (__bf16*)lo, Cout, Cin);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("wdgrad_planes launch failed");
}
int cdae_act_fwd(const float* x, float* y, long n, int kind, void* stream) { LAUNCH1D(act_kernel, n, x, y, n, kind); }
int cdae_act_bwd(const float* x, const float* dy, float* dx, long n, int kind, void* stream) { LAUNCH1D(act_bwd_kernel, n, x, dy, dx, n, kind); }
int cdae_silu_fwd(const float* x, float* y, long n, void* stream) { LAUNCH1D(silu_kernel, n, x, y, n); }
int cdae_silu_bwd(const float* x, const float* dy, float* dx, long n, void* stream) { LAUNCH1D(silu_bwd_kernel, n, x, dy, dx, n); }
int cdae_timestep_embed_fwd(const float* t, const float* freqs, float* out, int N, int dim, void* stream) {
    LAUNCH1D(temb_kernel, (long)N * dim, t, freqs, out, N, dim);
}
int cdae_model_timesteps(const long long* t, const long long* map, float scale, int rescale, float* out_f, long long* out_i, int N, void* stream) {
    LAUNCH1D(model_t_kernel, N, t, map, scale, rescale, out_f, out_i, N);
}
int cdae_embedding_add(float* emb, const float* table, const long long* idx, int N, int D, void* stream) {
    LAUNCH1D(embedding_add_kernel, (long)N * D, emb, table, idx, N, D);
}
int cdae_embedding_bwd(const float* demb, float* dtable, const long long* idx, int N, int D, void* stream) {
    LAUNCH1D(embedding_bwd_kernel, (long)N * D, demb, dtable, idx, N, D);
}
int cdae_axpby(float a, const float* x, float b, const float* y, float* out, long n, void* stream) { LAUNCH1D(axpby_kernel, n, a, x, b, y, out, n); }
// group-major planes [C / 16][P][16] -> pixel-major [P][C] (16-byte pieces; both planes in one launch): for the consumers that do not read
// the window conv kernel's layout (small shapes that fall back to the first-generation kernels)
__global__ void planes_gm_to_pc_kernel(const uint4* __restrict__ a_hi, const uint4* __restrict__ a_lo, uint4* __restrict__ o_hi, uint4* __restrict__ o_lo,
                                       long P, int C) {
    const int G2 = C >> 3;                         // 16-byte pieces per pixel
    const long total = P * G2;
    GRID_STRIDE(i, total) {
        const long pix = i / G2;
        const int piece = (int)(i - pix * G2);
        const long src = ((long)(piece >> 1) * P + pix) * 2 + (piece & 1);
        o_hi[i] = a_hi[src]; o_lo[i] = a_lo[src];
    }
}
int cdae_planes_gm_to_pc(const unsigned short* a_hi, const unsigned short* a_lo, unsigned short* o_hi, unsigned short* o_lo, long P, int C, void* stream) {
    if (C % 16) return cdae_fail("planes_gm_to_pc: C % 16 == 0 required");
    LAUNCH1D(planes_gm_to_pc_kernel, P * (C >> 3), reinterpret_cast<const uint4*>(a_hi), reinterpret_cast<const uint4*>(a_lo), reinterpret_cast<uint4*>(o_hi),
             reinterpret_cast<uint4*>(o_lo), P, C);
}
int cdae_mul_scale(const float* x, const float* m, float scale, float* out, long n, void* stream) { LAUNCH1D(mul_scale_kernel, n, x, m, scale, out, n); }
int cdae_mul_rows(float* x, const float* m, int N, int D, void* stream) { LAUNCH1D(mul_rows_kernel, (long)N * D, x, m, N, D); }
int cdae_copy2d(const float* src, float* dst, long rows, int cols, long lds, long ldd, int accumulate, void* stream) {
    LAUNCH1D(copy2d_kernel, rows * cols / 4 + 1, src, dst, rows, cols, lds, ldd, accumulate);
}
int cdae_gather_u8(const unsigned char* pool, const long long* idx, float* out, int B, long per_sample, float div, float shift, void* stream) {
    if (per_sample % 4 != 0) return cdae_fail("cdae_gather_u8: bytes per image must be a multiple of 4");
    LAUNCH1D(gather_u8_kernel, (long)B * (per_sample / 4), (const unsigned*)pool, idx, (float4*)out, per_sample / 4, (long)B * (per_sample / 4), div, shift);
}
int cdae_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, void* stream) { LAUNCH1D(nchw_to_nhwc_kernel, (long)N * C * HW, src, dst, N, C, HW); }
int cdae_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, void* stream) { LAUNCH1D(nhwc_to_nchw_kernel, (long)N * C * HW, src, dst, N, C, HW); }
int cdae_sumpool2(const float* src, float* dst, int N, int H, int W, int C, void* stream) {
    if (C % 4 == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0)
        LAUNCH1D(sumpool2_v4_kernel, (long)N * H * W * (C / 4), reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), N, H, W, C / 4);
    LAUNCH1D(sumpool2_kernel, (long)N * H * W * C, src, dst, N, H, W, C);
}

int cdae_upsample2(const float* x, float* y, int N, int H, int W, int C, float scale, void* stream) {
    const int V = (C % 4 == 0 && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0) ? 4 : 1;
    LAUNCH1D(upsample2_f32_kernel, (long)N * H * W * (C / V), x, y, H, W, C, V, scale, (long)N * H * W * (C / V));
}
int cdae_pool2(const float* src, float* dst, int N, int H, int W, int C, float scale, void* stream) {
    const int V = (C % 4 == 0 && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0) ? 4 : 1;
    LAUNCH1D(pool2_f32_kernel, (long)N * H * W * (C / V), src, dst, H, W, C, V, scale, (long)N * H * W * (C / V));
}

int cdae_q_sample(const float* x0, const float* noise, const long long* t, const float* tab, int T, float* out, int N, long per_sample, void* stream) {
    LAUNCH1D(q_sample_kernel, N * per_sample, x0, noise, t, tab, T, out, per_sample, N * per_sample);
}
int cdae_ddim_update(const float* x, const float* eps, const long long* t, const float* tab, int T, float eta, const float* noise, int clip,
                     float* sample, float* pred_xstart, int N, long per_sample, void* stream) {
    LAUNCH1D(ddim_update_kernel, N * per_sample, x, eps, t, tab, T, eta, noise, clip, sample, pred_xstart, per_sample, N * per_sample);
}
int cdae_ddpm_update(const float* x, const float* eps, const long long* t, const float* tab, int T, const float* noise, int clip,
                     float* sample, float* pred_xstart, int N, long per_sample, void* stream) {
    if (!noise) return cdae_fail("ddpm_update needs a noise tensor");
    LAUNCH1D(ddpm_update_kernel, N * per_sample, x, eps, t, tab, T, noise, clip, sample, pred_xstart, per_sample, N * per_sample);
}

int cdae_p_mean_variance(const float* x, const float* model_out, const long long* t, const float* tab, int T, int mean_type, int var_type,
                         int clip, const float* noise, float* mean, float* variance, float* log_variance, float* pred_xstart, float* sample,
                         int N, long per_sample, void* stream) {
    if (mean_type < 0 || mean_type > 1 || var_type < 0 || var_type > 2) return cdae_fail("p_mean_variance: bad mean/var type");
    if (sample && !noise) return cdae_fail("p_mean_variance: sample requested without noise");
    long ostride = var_type ? 2 * per_sample : per_sample;
    LAUNCH1D(pmv_kernel, N * per_sample, x, model_out, ostride, t, tab, T, mean_type, var_type, clip, noise, mean, variance, log_variance,
             pred_xstart, sample, per_sample, N * per_sample);
}
int cdae_vb_terms(const float* x_start, const float* x_t, const float* model_out, const long long* t, const float* tab, int T, int mean_type,
                  int var_type, int clip, float* vb, float* pred_xstart, int N, long per_sample, void* stream) {
    if (mean_type < 0 || mean_type > 1 || var_type < 0 || var_type > 2) return cdae_fail("vb_terms: bad mean/var type");
    long ostride = var_type ? 2 * per_sample : per_sample;
    cdae_prof_begin(PROF_ELEMWISE, 0.0, ST);
    hipLaunchKernelGGL(vb_terms_kernel, dim3(N), dim3(256), 0, ST, x_start, x_t, model_out, ostride, t, tab, T, mean_type, var_type, clip, vb,
                       pred_xstart, per_sample);
    cdae_prof_end(PROF_ELEMWISE, ST);
    if (hipGetLastError() != hipSuccess) return cdae_fail("vb_terms launch failed");
    return 0;
}
int cdae_vb_terms_bwd(const float* x_start, const float* x_t, const float* model_out, const long long* t, const float* tab, int T, int mean_type,
                      int var_type, int clip, int freeze_mean, const float* gout, float* dmodel_out, int N, long per_sample, void* stream) {
    if (mean_type < 0 || mean_type > 1 || var_type < 0 || var_type > 2) return cdae_fail("vb_terms_bwd: bad mean/var type");
    long ostride = var_type ? 2 * per_sample : per_sample;
    LAUNCH1D(vb_terms_bwd_kernel, N * per_sample, x_start, x_t, model_out, ostride, t, tab, T, mean_type, var_type, clip, freeze_mean, gout,
             dmodel_out, per_sample, N * per_sample);
}

int cdae_softplus_fwd(const float* x, float* y, long n, float add, void* stream) { LAUNCH1D(softplus_eps_kernel, n, x, y, n, add); }
int cdae_softplus_bwd(const float* x, const float* dy, float* dx, long n, void* stream) { LAUNCH1D(softplus_bwd_kernel, n, x, dy, dx, n); }
int cdae_reparam(const float* m, const float* v, float vscale, const float* eps, float* z, long n, void* stream) {
    LAUNCH1D(reparam_kernel, n, m, v, vscale, eps, z, n);
}
int cdae_causal_mask(const float* u, const float* A, float* out, int N, int nv, int d, int transpose, void* stream) {
    LAUNCH1D(causal_mask_kernel, (long)N * nv * d, u, A, out, N, nv, d, transpose);
}

int cdae_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int step, double ema_rate, double grad_scale, void* stream) {
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float sqrt_bc2 = (float)sqrt(1.0 - pow(beta2, (double)step));
    cdae_prof_begin(PROF_OPT, (double)n * 32.0, ST);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, ST, p, g, m, v, ema, n, (float)lr, (float)beta1, (float)beta2, (float)eps,
                       (float)weight_decay, bc1, sqrt_bc2, (float)ema_rate, (float)grad_scale);
    cdae_prof_end(PROF_OPT, ST);
    if (hipGetLastError() != hipSuccess) return cdae_fail("adamw_ema launch failed");
    return 0;
}
int cdae_adamw_ema_multi(float* p, const float* g, float* m, float* v, float* const* emas, const double* ema_rates, int n_ema, long n, double lr,
                         double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
    if (n_ema < 0 || n_ema > 4 || (n_ema && (!emas || !ema_rates))) return cdae_fail("adamw_ema_multi: 0..4 EMA buffers (host arrays of pointers and rates)");
    EmaSet es{};
    es.n = n_ema;
    for (int k = 0; k < n_ema; ++k) { if (!emas[k]) return cdae_fail("adamw_ema_multi: NULL EMA buffer"); es.buf[k] = emas[k]; es.rate[k] = (float)ema_rates[k]; }
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float sqrt_bc2 = (float)sqrt(1.0 - pow(beta2, (double)step));
    cdae_prof_begin(PROF_OPT, (double)n * (24.0 + 8.0 * n_ema), ST);
    hipLaunchKernelGGL(adamw_ema_multi_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, ST, p, g, m, v, es, n, (float)lr, (float)beta1, (float)beta2, (float)eps,
                       (float)weight_decay, bc1, sqrt_bc2, (float)grad_scale);
    cdae_prof_end(PROF_OPT, ST);
    if (hipGetLastError() != hipSuccess) return cdae_fail("adamw_ema_multi launch failed");
    return 0;
}
int cdae_sqsum(const float* x, long n, double* out, void* stream) {
    if (hipMemsetAsync(out, 0, sizeof(double), ST) != hipSuccess) return cdae_fail("memset failed");
    LAUNCH1D(sqsum_kernel, n, x, n, out);
}
int cdae_mse_rows(const float* a, const float* b, float* out, int N, long per, void* stream) {
    hipLaunchKernelGGL(mse_rows_kernel, dim3(N), dim3(256), 0, ST, a, b, out, per);
    if (hipGetLastError() != hipSuccess) return cdae_fail("mse_rows launch failed");
    return 0;
}
int cdae_rep_loss(const float* mu, const float* var, const float* z_post, const float* c, float* out, int N, int D, int nv, void* stream) {
    if (N <= 0 || D <= 0 || !mu || !var || !out || (z_post && (!c || nv <= 0 || D % nv))) return cdae_fail("rep_loss: mu, var, out (and c, nv | D with z_post) required");
    hipLaunchKernelGGL(rep_loss_kernel, dim3(N), dim3(64), 0, ST, mu, var, z_post, c, out, D, z_post ? nv : 0);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("rep_loss launch failed");
}
int cdae_rep_loss_bwd(const float* mu, const float* var, const float* z_post, const float* c, const float* gout, float* dmu, float* dvar, float* dz_post,
                      int N, int D, int nv, void* stream) {
    if (N <= 0 || D <= 0 || !mu || !var || !gout || !dmu || !dvar || (z_post && (!c || !dz_post || nv <= 0 || D % nv))) return cdae_fail("rep_loss_bwd: bad arguments");
    LAUNCH1D(rep_loss_bwd_kernel, (long)N * D, mu, var, z_post, c, gout, dmu, dvar, z_post ? dz_post : nullptr, D, z_post ? nv : 0, (long)N * D);
}
int cdae_mse_rows_bwd(const float* a, const float* b, const float* gout, float* db, int N, long per, void* stream) {
    LAUNCH1D(mse_rows_bwd_kernel, N * per, a, b, gout, db, per, N * per);
}

}  // extern "C"
