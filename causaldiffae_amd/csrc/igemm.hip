// igemm.hip — fp32 implicit-GEMM on CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// One kernel family does every dense contraction of the CausalDiffAE hot path:
//   C[i][j] = alpha * sum_k A(i,k) * B(j,k)  (+ bias[j]) (+ res[i][j]) (-> SiLU)
// with pluggable operand loaders:
//   A_PLAIN_KC   A[i][k], k contiguous                       linear / conv1x1 / QK^T / dgrad
//   A_CONV_VEC   im2col gather of an NHWC tensor, k=(tap,cin)  conv3x3 fwd (stride 1|2, fused nearest-2x
//                upsample, "transposed" stride-2 gather for the Downsample dgrad); needs Cin % 32 == 0
//   A_CONV_GEN   same gather, any Cin / any element strides    stem (NCHW input), encoder convs
//   A_PLAIN_MC   A[i][k] stored k-major (i contiguous)         wgrad (dY^T), attention backward
//   B_PLAIN_KC   B[j][k], k contiguous                         weights [Cout][9*Cin] (OHWI), K rows of K^T
//   B_PLAIN_MC   B[j][k] stored k-major (j contiguous)         V in P@V, W in linear dgrad, X in wgrad
//   B_CONV_MC    k = output pixel, j=(tap,cin): shifted NHWC rows   conv3x3 wgrad
//   B_WDGRAD_MC  k=(tap,cout), j=cin of OHWI weights, tap flipped   conv3x3 dgrad
//
// Numerics: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fmaf chain (no reduced
// precision), so results stay within fp32 rounding of the reference's ATen path.
//
// Tile: BM x BN x 32, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 32x32 MFMA tiles.
// LDS: K-contiguous operands as [rows][32+4] (conflict-free ds_read_b128: one read feeds 4 MFMAs),
// k-major operands as [32][rows+4] with interleaved MFMA row slots (ds_read_b64 along rows).  Register-staged double
// buffering: tile t+1's global loads are issued before the MFMAs of tile t and written to the
// other LDS buffer afterwards (one barrier per K-step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "cdae_internal.h"
#include "../../include/cdae.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// dev ablations / stamps of the plane kernels (CDAE_PS_DBG bits) exist only in a -DCW_DEV=1 build (EXTRA_HIPCC_FLAGS=-DCW_DEV=1 build.sh):
// as run-time tests they sat in every K step of the production kernels
#ifndef CW_DEV
#define CW_DEV 0
#endif
#define PDBG(P) (CW_DEV ? (P).dbg : 0)

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;     // row pitch (floats) of a K-contiguous LDS tile

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- conv gather geometry for one GEMM row (an output pixel, or for wgrad a reduction pixel)
struct PixRow {
    long base;     // element offset of image n
    int iy0, ix0;  // top-left input coordinate of the 3x3 window (already * stride - pad)
    int ok;        // row < M
};

// exact floor(n / d) for n < 2^31 with a host-computed (magic, shift): (mulhi(n, magic) + n) >> shift
__device__ __forceinline__ int fdiv(int n, unsigned magic, int shift) {
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

// Everything below is straight-line (no branches, no early returns): any control flow in the gather geometry ends
// up between the tile loads and makes hipcc serialise them with vmcnt waits.  The mode switches (stride / fused
// upsample / transposed stride-2 gather) are folded into host-computed constants g_*.
__device__ __forceinline__ PixRow make_pixrow(const GemmParams& p, int m) {
    PixRow r;
    r.ok = m < p.conv_M;
    const int mm = r.ok ? m : 0;
    const int n = fdiv(mm, p.hw_magic, p.hw_shift);
    const int rem = mm - n * p.hw;
    const int oy = fdiv(rem, p.wo_magic, p.wo_shift), ox = rem - oy * p.Wo;
    r.base = (long)n * p.sn;
    r.iy0 = oy * p.g_mul + p.g_add;         // conv: o*stride - 1;  transposed gather: o + 1
    r.ix0 = ox * p.g_mul + p.g_add;
    return r;
}

// offset (elements) of the input pixel under window tap (ky,kx); false when the tap reads padding
__device__ __forceinline__ bool tap_offset(const GemmParams& p, const PixRow& r, int ky, int kx, long& off) {
    const int ty = r.iy0 + p.g_sign * ky, tx = r.ix0 + p.g_sign * kx;      // transposed gather walks the taps backwards
    const bool ok = ((ty | tx) >= 0) & (((ty | tx) & p.g_pm) == 0) & (ty < (p.H << p.g_sh)) & (tx < (p.W << p.g_sh));
    const int iy = ty >> p.g_sh, ix = tx >> p.g_sh;                       // >>1: fused upsample source / stride-2 transpose
    off = ok ? r.base + (long)iy * p.sy + (long)ix * p.sx : 0;
    return ok;
}

// Branch-free guarded loads.  A conditional `ok ? load : 0` makes hipcc branch around every load and wait
// vmcnt(0) before the next one (all tile loads of a K-step serialised), and a value select after the load drags
// the vmcnt wait in front of the MFMAs.  Selecting the ADDRESS instead (a zero-filled device constant when !ok)
// keeps the loads unconditional, back-to-back and un-waited until the LDS store after the MFMAs.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // non-const: stays in the global address space (a constant-space pointer turns the select into flat loads)

__device__ __forceinline__ float4 ld4_if(const float* /*unused*/, const float* p, bool ok) {
    return ld4(ok ? p : g_zero16);
}
__device__ __forceinline__ float ld1_if(const float* /*unused*/, const float* p, bool ok) {
    return *(ok ? p : g_zero16);
}

// second output of an epilogue: v as f16 hi/lo planes
__device__ __forceinline__ void store_planes(const GemmParams& p, long addr, float v) {
    asm volatile("" : "+v"(v));        // opaque: no second, differently rounded f16 conversion folded into the producing fma (see attention.hip split8)
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    p.C_hi[addr] = __builtin_bit_cast(unsigned short, h);
    p.C_lo[addr] = __builtin_bit_cast(unsigned short, l);
}

// SCALAR = element-wise operand loads (odd K / pitch / alignment, tiny channel counts); only built for 64x64 tiles.
//
// PREC = 1: "f16x3" split precision for K-contiguous operand pairs.  Each fp32 operand value x is split on its way
// into LDS into two f16 planes, x ~ hi + lo (hi = f16(x), lo = f16(x - hi): 22 significand bits), and every product
// is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dropped lo*lo term
// is 2^-22 relative).  3 MFMAs at 16x the fp32-MFMA rate = 5.3x less matrix-pipe time than v_mfma_f32_32x32x2_f32 at
// ~fp32 accuracy, all into one fp32 accumulator (a two-stage register prefetch was tried: +30 VGPRs, one wave per
// SIMD less, 17 % slower).  Range: |x| < 65504 (the
// reference network is fp16-safe by construction: it ships a use_fp16 mode, unet.py:501-507).
// PREC = 3 / 4: ONE f16 / bf16 plane and one MFMA per product — the reduced-precision torso ("mixed16" mode, the
// counterpart of the reference's use_fp16 torso / BASELINE's bf16 training config; not used for the fp32-parity path).
// PREC = 2: the same as PREC 1 with bf16 planes ("bf16x3", 16 significand bits, full fp32 range) — used for every GEMM that
// has a GRADIENT operand (dgrad / wgrad / attention backward): gradients underflow f16, and 2^-16 is far below what
// the optimizer can see.
// K-contiguous operands: planes of [rows][32 x 16-bit] (64-B rows, 16-B chunks XOR-swizzled by (row>>2)&3), fragments
// by one ds_read_b128.  k-major operands: planes of [32 k][rows+32] (coalesced 8-B stores), fragments by two
// ds_read_b64_tr_b16 (hardware transpose read; the +32 pad makes both conflict-free).
// GNS (one instantiation: the 1x1 skip conv of a ResBlock, K-contiguous fp32 A, f16x3): the A operand is the RAW block input, which the
// block's first GroupNorm reads as well — so while an A tile sits in registers on its way into LDS, the n-tile-0 blocks also apply that
// GroupNorm (+SiLU; folded per (image, channel) coefficients p.gn_coef, bit-identical to gn_apply_kernel) and write the result as
// f16 hi/lo planes (p.S_hi / p.S_lo, dense [M][K]): the separate normalisation pass over the tensor disappears.  The GEMM is
// HBM-bound, the extra VALU work is free.
template <int BM, int BN, int AMODE, int BMODE, bool SCALAR, int WAVES_N = 2, int PREC = 0, bool GNS = false, bool DEEP = false>
__global__ __launch_bounds__(128 * WAVES_N, (GNS && DEEP) ? 4 : 1) void igemm_kernel(const GemmParams p) {     // GNS + DEEP: keep four waves per SIMD (128 VGPRs)
    static_assert(PREC == 0 || !SCALAR, "split precision is only built for vectorised loaders");
    static_assert(!GNS || (AMODE == A_PLAIN_KC && PREC == 1 && !SCALAR), "the GroupNorm side output rides on the K-contiguous f16x3 loader");
    constexpr int NPL = (PREC == 1 || PREC == 2) ? 2 : 1;       // 16-bit planes per operand
    constexpr bool BF = (PREC == 2 || PREC == 4);               // bf16 (else f16) planes
    // 2 x WAVES_N waves; WAVES_N = 4 (512 threads, 64x32 per wave at 128x128) doubles the waves per SIMD that can
    // cover each other's barrier / LDS waits at the same LDS footprint
    constexpr int THREADS = 128 * WAVES_N, RPP = THREADS / 8;     // RPP = tile rows covered per loader pass
    constexpr int WM = BM / 2, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    constexpr bool A_MC = (AMODE == A_PLAIN_MC);
    constexpr bool B_MC = (BMODE != B_PLAIN_KC);
    // K-contiguous operands: LDS [rows][32+4], one ds_read_b128 per lane feeds 4 MFMAs.  k-major operands keep their
    // global order in LDS, [32][rows+4] (coalesced float4 stores, no bank conflicts); the wave's MFMA row slots are
    // INTERLEAVED over its tiles (slot s of tile t <-> row T*s+t) so one ds_read_b64/b128 along the rows serves all
    // tiles of a k.  (A register-transposed variant measured 8-way ds_write conflicts and MFMA busy 0.39.)
    constexpr int LDAM = BM + 4, LDBM = BN + 4;
    // PREC 1: two f16 planes of [rows][32 halfs] (64 B rows, 16-B chunks XOR-swizzled by (row>>2)&3: conflict-free b128)
    constexpr int PAM = BM + 32, PBM = BN + 32;                                  // k-major 16-bit plane pitch (elements)
    constexpr int A_PLANE = A_MC ? BK * PAM * 2 : BM * 64;                       // bytes per 16-bit plane
    constexpr int B_PLANE = B_MC ? BK * PBM * 2 : BN * 64;
    constexpr int A_TILE = PREC ? NPL * A_PLANE / 4 : (A_MC ? BK * LDAM : BM * LDK);   // in floats
    constexpr int B_TILE = PREC ? NPL * B_PLANE / 4 : (B_MC ? BK * LDBM : BN * LDK);
    constexpr int A_V4 = BM * BK / 4 / THREADS;     // float4 loads per thread per tile
    constexpr int B_V4 = BN * BK / 4 / THREADS;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // block -> tile.  1-D grid; blocks b and b+8 share an XCD (and its L2), so first give each XCD a CONTIGUOUS run
    // of virtual ids, then decode n-tile fastest, m-tile, K-split, batch: the n-tiles of one m-tile (same A rows),
    // neighbouring m-tiles (shared 3x3 halo rows) and — for wgrad — all (tap, cin-block) tiles of one pixel chunk
    // run on the same XCD at about the same time and hit in its L2 instead of re-fetching from HBM/MALL.
    const int nmt = (p.M + BM - 1) / BM, nnt = (p.N + BN - 1) / BN;
    int mt, nt, ks, bz;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned q = G >> 3, r = G & 7, x = b & 7;
        unsigned v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        nt = v % nnt; v /= nnt;
        mt = v % nmt; v /= nmt;
        ks = v % p.ksplit; bz = v / p.ksplit;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
    const float* __restrict__ Ag = p.A + bo * p.a_bs0 + bi * p.a_bs1;
    const float* __restrict__ Bg = p.B + bo * p.b_bs0 + bi * p.b_bs1;

    const int nk_total = (p.K + BK - 1) / BK;
    const int nk_per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * nk_per;
    const int kt_end = min(nk_total, kt_begin + nk_per);

    // GNS: GroupNorm coefficients of this tile's images from LDS.  Read from global memory inside store_tiles they sit BEHIND the
    // register prefetch of the next two K-steps in the (in-order) vmcnt queue, so waiting for them drained the prefetch every
    // step — the "two steps ahead" pipeline of this HBM-bound kernel never ran ahead at all.
    float* const ctab = smem + 2 * (A_TILE + B_TILE);
    int gn_img0 = 0;
    const bool gn_tab = GNS && p.gn_tab && nt == 0;
    if constexpr (GNS) {
        if (gn_tab) {
            gn_img0 = fdiv(m0, p.hw_magic, p.hw_shift);
            const int img_last = fdiv(p.M - 1, p.hw_magic, p.hw_shift);
            const int per = p.K >> 1;                                  // float4s per image: K x (a, b)
            for (int t = tid; t < 2 * per; t += THREADS) {
                const int sel = t >= per ? 1 : 0;
                const int img = min(gn_img0 + sel, img_last);
                reinterpret_cast<float4*>(ctab)[t] = *reinterpret_cast<const float4*>(p.gn_coef + ((long)img * p.K) * 2 + (long)(t - sel * per) * 4);
            }
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- per-thread loader state
    PixRow arow[A_V4];
    long atap_off[A_V4];          // A_CONV_VEC: element offset of the current tap's pixel, refreshed when the tap changes
    bool atap_ok[A_V4];
    int a_cur_tap = -1;
    if constexpr (AMODE == A_CONV_VEC || AMODE == A_CONV_GEN) {
#pragma unroll
        for (int q = 0; q < A_V4; ++q) { arow[q] = make_pixrow(p, m0 + (tid >> 3) + RPP * q); atap_off[q] = 0; atap_ok[q] = false; }
    }
    // K-contiguous plain operands: row pointers are loop invariant
    const float* arowp[A_V4];
    long adelta[A_V4];            // two-source A: what to add to arowp[q] + k0 once k0 reaches the second source (else unused)
    bool arow_ok[A_V4];
    if constexpr (AMODE == A_PLAIN_KC) {
#pragma unroll
        for (int q = 0; q < A_V4; ++q) {
            int m = m0 + (tid >> 3) + RPP * q;
            arow_ok[q] = m < p.M;
            const long mm = arow_ok[q] ? m : 0;
            arowp[q] = Ag + mm * p.lda + (tid & 7) * 4;
            adelta[q] = p.A2 ? (p.A2 + mm * p.lda2 + (tid & 7) * 4 - p.K1) - arowp[q] : 0;
        }
    }
    const float* browp[B_V4];
    bool brow_ok[B_V4];
    if constexpr (BMODE == B_PLAIN_KC) {
#pragma unroll
        for (int q = 0; q < B_V4; ++q) {
            int n = n0 + (tid >> 3) + RPP * q;
            brow_ok[q] = n < p.N;
            browp[q] = Bg + (long)(brow_ok[q] ? n : 0) * p.ldb + (tid & 7) * 4;
        }
    }

    float4 areg[A_V4], breg[B_V4];
    float4 areg2[DEEP ? A_V4 : 1], breg2[DEEP ? B_V4 : 1];          // DEEP: second register set of the two-step load pipeline

    // K-step -> first k of the step: tap major, channel minor (the per-row tap geometry is refreshed once per tap;
    // a channel-chunk-major order was measured 5 % slower because it recomputes it every step).
    auto kstart = [&](int kt) -> int { return kt * BK; };

    auto load_A = [&](int kt, float4* ra) {
        const int k0 = kstart(kt);
        if constexpr (AMODE == A_PLAIN_KC) {
            const int k = k0 + (tid & 7) * 4;
#pragma unroll
            for (int q = 0; q < A_V4; ++q) {
                if constexpr (!SCALAR) ra[q] = ld4_if(Ag, arowp[q] + k0 + ((p.A2 && k0 >= p.K1) ? adelta[q] : 0L), arow_ok[q] && k < p.K);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ld1_if(Ag, arowp[q] + k0 + e, arow_ok[q] && k + e < p.K);
                    ra[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        } else if constexpr (AMODE == A_CONV_VEC) {
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
            if (tap != a_cur_tap) {                 // block-uniform: refresh the per-row pixel offsets once per tap
                a_cur_tap = tap;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int q = 0; q < A_V4; ++q) {
                    long off = 0;
                    atap_ok[q] = arow[q].ok && tap_offset(p, arow[q], ky, kx, off);
                    atap_off[q] = off + (tid & 7) * 4;
                }
            }
#pragma unroll
            for (int q = 0; q < A_V4; ++q) ra[q] = ld4_if(Ag, Ag + atap_off[q] + c0, atap_ok[q]);
        } else if constexpr (AMODE == A_CONV_GEN) {
#pragma unroll
            for (int q = 0; q < A_V4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + (tid & 7) * 4 + e;
                    const int kk = k < p.K ? k : 0;
                    const int tap = kk / p.Cin, c = kk - tap * p.Cin;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    long off = 0;
                    const bool ok = arow[q].ok && k < p.K && tap_offset(p, arow[q], ky, kx, off);
                    v[e] = ld1_if(Ag, Ag + off + (long)c * p.sc, ok);
                }
                ra[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {   // A_PLAIN_MC: element (i,k) at Ag + k*lda + i
#pragma unroll
            for (int q = 0; q < A_V4; ++q) {
                const int idx = tid + THREADS * q;
                const int kk = idx / (BM / 4), i4 = idx - kk * (BM / 4);
                const int k = k0 + kk, i = m0 + i4 * 4;
                const float* src = Ag + (long)(k < p.K ? k : 0) * p.lda + i;
                if constexpr (!SCALAR) ra[q] = ld4_if(Ag, src, k < p.K && i < p.M);      // host guarantees M % 4 == 0
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ld1_if(Ag, src + e, k < p.K && i + e < p.M);
                    ra[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };

    auto load_B = [&](int kt, float4* rb) {
        const int k0 = kstart(kt);
        if constexpr (BMODE == B_PLAIN_KC) {
            const int k = k0 + (tid & 7) * 4;
#pragma unroll
            for (int q = 0; q < B_V4; ++q) {
                if constexpr (!SCALAR) rb[q] = ld4_if(Bg, browp[q] + k0, brow_ok[q] && k < p.K);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ld1_if(Bg, browp[q] + k0 + e, brow_ok[q] && k + e < p.K);
                    rb[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < B_V4; ++q) {
                const int idx = tid + THREADS * q;
                const int kk = idx / (BN / 4), j4 = idx - kk * (BN / 4);
                const int k = k0 + kk, j = n0 + j4 * 4;
                if constexpr (BMODE == B_CONV_MC && SCALAR) {
                    // k = pixel of the dY grid, j = tap*Cin + cin decoded per element
                    const PixRow r = make_pixrow(p, k);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int jj = j + e < p.N ? j + e : 0;
                        const int tap = jj / p.Cin, c = jj - tap * p.Cin;
                        const int ky = tap / 3, kx = tap - ky * 3;
                        long off = 0;
                        const bool ok = r.ok && j + e < p.N && tap_offset(p, r, ky, kx, off);
                        v[e] = ld1_if(Bg, Bg + off + (long)c * p.sc, ok);
                    }
                    rb[q] = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    const float* src;
                    bool kok = k < p.K;
                    if constexpr (BMODE == B_CONV_MC) {
                        const PixRow r = make_pixrow(p, k);
                        const int tap = n0 / p.Cin;                 // whole block inside one tap (Cin % BN == 0)
                        const int ky = tap / 3, kx = tap - ky * 3;
                        long off = 0;
                        kok = r.ok && tap_offset(p, r, ky, kx, off);
                        src = Bg + off + (j - tap * p.Cin);
                    } else if constexpr (BMODE == B_PLAIN_MC) {
                        src = Bg + (long)(kok ? k : 0) * p.ldb + j;
                    } else {   // B_WDGRAD_MC: k = tap*Cout + co (tap of the dY gather), j = cin; OHWI weights, tap flipped
                        const int kk2 = kok ? k : 0;
                        const int tap = kk2 / p.wCout, co = kk2 - tap * p.wCout;
                        const int ft = p.wflip ? 8 - tap : tap;
                        src = Bg + ((long)co * 9 + ft) * p.wCin + j;
                    }
                    if constexpr (!SCALAR) rb[q] = ld4_if(Bg, src, kok && j < p.N);      // host guarantees N % 4 == 0
                    else {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ld1_if(Bg, src + e, kok && j + e < p.N);
                        rb[q] = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
    };

    // split a float4 into hi/lo 16-bit planes (f16 for PREC 1, bf16 for PREC 2) and store 8 B into each plane
    auto store_split = [&](char* tile, int plane_bytes, int off, const float4& v) {
        if constexpr (BF) {
            bf4 hi, lo;
            hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
            *reinterpret_cast<bf4*>(tile + off) = hi;
            if constexpr (NPL == 2) {
                lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
                lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
                *reinterpret_cast<bf4*>(tile + plane_bytes + off) = lo;
            }
        } else {
            half4 hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            *reinterpret_cast<half4*>(tile + off) = hi;
            if constexpr (NPL == 2) {
                lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
                lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
                *reinterpret_cast<half4*>(tile + plane_bytes + off) = lo;
            }
        }
    };
    auto kc_off = [&](int row, int c4) { return row * 64 + 16 * ((c4 >> 1) ^ ((row >> 2) & 3)) + 8 * (c4 & 1); };

    // wgrad: the A operand is dY^T, so its column sums over the pixels (= the bias gradient, the reference gets it from
    // autograd's conv backward) fall out of the tiles that pass through anyway: the n-tile-0 blocks add up what they stage
    float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = A_MC && p.colsum_out != nullptr && nt == 0;

    auto store_tiles = [&](int buf, int kt_of, const float4* ra, const float4* rb) {
        float* a = As + buf * A_TILE;
        float* b = Bs + buf * B_TILE;
        if constexpr (GNS) {
            if (nt == 0) {                                    // block-uniform: one column block per row tile writes the planes
                const int k = kstart(kt_of) + (tid & 7) * 4;
#pragma unroll
                for (int q = 0; q < A_V4; ++q) {
                    const int m = m0 + (tid >> 3) + RPP * q;
                    if (m < p.M && k < p.K) {
                        const int img = fdiv(m, p.hw_magic, p.hw_shift);
                        const float4* cf = gn_tab ? reinterpret_cast<const float4*>(ctab + ((img - gn_img0) * p.K + k) * 2)
                                                  : reinterpret_cast<const float4*>(p.gn_coef + ((long)img * p.K + k) * 2);
                        const float4 c01 = cf[0], c23 = cf[1];                // a0 b0 a1 b1 | a2 b2 a3 b3
                        float y0 = fmaf(ra[q].x, c01.x, c01.y), y1 = fmaf(ra[q].y, c01.z, c01.w);
                        float y2 = fmaf(ra[q].z, c23.x, c23.y), y3 = fmaf(ra[q].w, c23.z, c23.w);
                        if (p.gn_silu) { y0 = cdae_silu(y0); y1 = cdae_silu(y1); y2 = cdae_silu(y2); y3 = cdae_silu(y3); }
                        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));      // opaque before the split (attention.hip split8)
                        half4 hi, lo;
                        hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
                        lo[0] = (_Float16)(y0 - (float)hi[0]); lo[1] = (_Float16)(y1 - (float)hi[1]);
                        lo[2] = (_Float16)(y2 - (float)hi[2]); lo[3] = (_Float16)(y3 - (float)hi[3]);
                        const long o = (long)m * p.K + k;
                        *reinterpret_cast<half4*>(p.S_hi + o) = hi;
                        *reinterpret_cast<half4*>(p.S_lo + o) = lo;
                    }
                }
            }
        }
        if constexpr (A_MC) {
            if (do_colsum) {
#pragma unroll
                for (int q = 0; q < A_V4; ++q) { colsum.x += ra[q].x; colsum.y += ra[q].y; colsum.z += ra[q].z; colsum.w += ra[q].w; }
            }
        }
        if constexpr (PREC != 0) {
            char* ac = reinterpret_cast<char*>(a);
            char* bc = reinterpret_cast<char*>(b);
#pragma unroll
            for (int q = 0; q < A_V4; ++q) {
                if constexpr (!A_MC) store_split(ac, A_PLANE, kc_off((tid >> 3) + RPP * q, tid & 7), ra[q]);
                else {
                    const int idx = tid + THREADS * q;
                    const int kk = idx / (BM / 4), i4 = idx - kk * (BM / 4);
                    store_split(ac, A_PLANE, (kk * PAM + i4 * 4) * 2, ra[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < B_V4; ++q) {
                if constexpr (!B_MC) store_split(bc, B_PLANE, kc_off((tid >> 3) + RPP * q, tid & 7), rb[q]);
                else {
                    const int idx = tid + THREADS * q;
                    const int kk = idx / (BN / 4), j4 = idx - kk * (BN / 4);
                    store_split(bc, B_PLANE, (kk * PBM + j4 * 4) * 2, rb[q]);
                }
            }
            return;
        }
        if constexpr (!A_MC) {
#pragma unroll
            for (int q = 0; q < A_V4; ++q)
                *reinterpret_cast<float4*>(a + ((tid >> 3) + RPP * q) * LDK + (tid & 7) * 4) = ra[q];
        } else {
#pragma unroll
            for (int q = 0; q < A_V4; ++q) {
                const int idx = tid + THREADS * q;
                const int kk = idx / (BM / 4), i4 = idx - kk * (BM / 4);
                *reinterpret_cast<float4*>(a + kk * LDAM + i4 * 4) = ra[q];
            }
        }
        if constexpr (!B_MC) {
#pragma unroll
            for (int q = 0; q < B_V4; ++q)
                *reinterpret_cast<float4*>(b + ((tid >> 3) + RPP * q) * LDK + (tid & 7) * 4) = rb[q];
        } else {
#pragma unroll
            for (int q = 0; q < B_V4; ++q) {
                const int idx = tid + THREADS * q;
                const int kk = idx / (BN / 4), j4 = idx - kk * (BN / 4);
                *reinterpret_cast<float4*>(b + kk * LDBM + j4 * 4) = rb[q];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
            }

    if (kt_begin < kt_end) {
        load_A(kt_begin, areg);
        load_B(kt_begin, breg);
        store_tiles(0, kt_begin, areg, breg);
        if constexpr (DEEP) { if (kt_begin + 1 < kt_end) { load_A(kt_begin + 1, areg); load_B(kt_begin + 1, breg); } }
    }
    __syncthreads();

    int cur = 0;
    // one 32-deep K step.  DEEP: the global loads run TWO steps ahead of the MFMAs in two alternating register sets (the shallow, tall
    // GEMMs of the 1x1 convs are latency-bound with one step in flight: each step waits a full HBM round trip for 32 KB per block)
    auto kstep = [&](int kt, float4* ra_c, float4* rb_c, float4* ra_n, float4* rb_n) {
        const bool more = kt + 1 < kt_end;
        if constexpr (DEEP) { if (kt + 2 < kt_end) { load_A(kt + 2, ra_n); load_B(kt + 2, rb_n); } }
        else if (more) { load_A(kt + 1, ra_c); load_B(kt + 1, rb_c); }

        const float* a = As + cur * A_TILE;
        const float* b = Bs + cur * B_TILE;
        if constexpr (PREC != 0) {
            const char* ac = reinterpret_cast<const char*>(a);
            const char* bc = reinterpret_cast<const char*>(b);
            // one 16-bit MFMA operand (8 consecutive k of this lane's row) from a plane
            auto frag = [&](const char* plane, int row0, int sk, auto mc_tag, int pitch) -> u16x8 {
                if constexpr (!decltype(mc_tag)::value) {
                    const int row = row0 + l31;
                    return *reinterpret_cast<const u16x8*>(plane + row * 64 + 16 * ((2 * sk + hh) ^ ((row >> 2) & 3)));
                } else {
                    // k-major plane [k][pitch]: ds_read_b64_tr_b16 — lane 4q+p of a 16-lane group addresses row k0+q,
                    // columns c0+4p..+3 and receives column (lane&15) of the 4 rows: 4 consecutive k of its own row slot
                    const int q4 = (lane & 15) >> 2, p4 = lane & 3;
                    const int c0 = row0 + 16 * ((lane >> 4) & 1), k0 = 16 * sk + 8 * hh;
                    const char* src = plane + ((k0 + q4) * pitch + c0 + 4 * p4) * 2;
                    fp16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
                    fp16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + 4 * pitch * 2));
                    u16x4 a4 = __builtin_bit_cast(u16x4, lo4), b4 = __builtin_bit_cast(u16x4, hi4);
                    u16x8 r;
                    r[0] = a4[0]; r[1] = a4[1]; r[2] = a4[2]; r[3] = a4[3]; r[4] = b4[0]; r[5] = b4[1]; r[6] = b4[2]; r[7] = b4[3];
                    return r;
                }
            };
            auto mma = [&](const u16x8& x, const u16x8& y, const f32x16& c) -> f32x16 {
                if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0);
                else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
            };
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {                 // two 16-deep MFMA steps per 32-k tile; lane half hh holds k = 8hh..8hh+7
                u16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = frag(ac, wm * WM + i * 32, sk, std::integral_constant<bool, A_MC>{}, PAM);
                    if constexpr (NPL == 2) al[i] = frag(ac + A_PLANE, wm * WM + i * 32, sk, std::integral_constant<bool, A_MC>{}, PAM);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = frag(bc, wn * WN + j * 32, sk, std::integral_constant<bool, B_MC>{}, PBM);
                    if constexpr (NPL == 2) bl[j] = frag(bc + B_PLANE, wn * WN + j * 32, sk, std::integral_constant<bool, B_MC>{}, PBM);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (NPL == 2) {
                            acc[i][j] = mma(al[i], bh[j], acc[i][j]);        // cross terms first, then hi*hi, one fp32 accumulator
                            acc[i][j] = mma(ah[i], bl[j], acc[i][j]);
                        }
                        acc[i][j] = mma(ah[i], bh[j], acc[i][j]);
                    }
            }
        } else
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float af[TM][4], bf[TN][4];
            if constexpr (!A_MC) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float4 v = *reinterpret_cast<const float4*>(a + (wm * WM + i * 32 + l31) * LDK + kg * 8 + 4 * hh);
                    af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* src = a + (kg * 8 + 4 * hh + e) * LDAM + wm * WM + TM * l31;      // rows TM*slot + t
                    if constexpr (TM == 2) { float2 v = *reinterpret_cast<const float2*>(src); af[0][e] = v.x; af[1][e] = v.y; }
                    else if constexpr (TM == 4) { float4 v = *reinterpret_cast<const float4*>(src); af[0][e] = v.x; af[1][e] = v.y; af[2][e] = v.z; af[3][e] = v.w; }
                    else {
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[i][e] = src[i];
                    }
                }
            }
            if constexpr (!B_MC) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float4 v = *reinterpret_cast<const float4*>(b + (wn * WN + j * 32 + l31) * LDK + kg * 8 + 4 * hh);
                    bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* src = b + (kg * 8 + 4 * hh + e) * LDBM + wn * WN + TN * l31;
                    if constexpr (TN == 2) { float2 v = *reinterpret_cast<const float2*>(src); bf[0][e] = v.x; bf[1][e] = v.y; }
                    else if constexpr (TN == 4) { float4 v = *reinterpret_cast<const float4*>(src); bf[0][e] = v.x; bf[1][e] = v.y; bf[2][e] = v.z; bf[3][e] = v.w; }
                    else {
#pragma unroll
                        for (int j = 0; j < TN; ++j) bf[j][e] = src[j];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }

        if (more) store_tiles(cur ^ 1, kt + 1, ra_c, rb_c);
        __syncthreads();
        cur ^= 1;
    };
    if constexpr (DEEP) {
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            kstep(kt, areg, breg, areg2, breg2);
            if (kt + 1 < kt_end) kstep(kt + 1, areg2, breg2, areg, breg);
        }
    } else {
        for (int kt = kt_begin; kt < kt_end; ++kt) kstep(kt, areg, breg, areg, breg);
    }

    if constexpr (A_MC) {
        if (do_colsum) {          // block-uniform
            // threads with equal tid % (BM/4) hold partial sums of the same 4 columns (rows of the GEMM): fold through LDS
            float4* red = reinterpret_cast<float4*>(smem);          // the tile buffers are dead after the last barrier
            red[tid] = colsum;
            __syncthreads();
            if (tid < BM / 4) {
                float4 s4 = red[tid];
                for (int t = tid + BM / 4; t < THREADS; t += BM / 4) { float4 v = red[t]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
                const int i = m0 + tid * 4;
                if (i < p.M) atomicAdd(p.colsum_out + i, s4.x);
                if (i + 1 < p.M) atomicAdd(p.colsum_out + i + 1, s4.y);
                if (i + 2 < p.M) atomicAdd(p.colsum_out + i + 2, s4.z);
                if (i + 3 < p.M) atomicAdd(p.colsum_out + i + 3, s4.w);
            }
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- epilogue
    // C layout of v_mfma_f32_32x32xK (dtype independent): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* __restrict__ Cg;
    const float* __restrict__ Rg = nullptr;
    if (p.ksplit > 1) {
        Cg = p.splitk_ws + ((long)ks * p.batch + bz) * (long)p.M * p.N;      // dense [M][N] slab
    } else {
        Cg = p.C + bo * p.c_bs0 + bi * p.c_bs1;
        if (p.res) Rg = p.res + bo * p.c_bs0 + bi * p.c_bs1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + ((B_MC && PREC == 0) ? TN * l31 + j : j * 32 + l31);      // fp32 k-major B: interleaved column slots
            if (col >= p.N) continue;
            const float bv = (p.ksplit == 1 && p.bias) ? p.bias[col] : 0.f;
            // residual / accumulate operands of the sub-tile's 16 rows are requested together, ahead of the loop (a test + load + use per
            // element is one exposed memory round trip each: 16 per sub-tile); rows beyond M read element 0
            float rv[16], cv[16];
            if (p.ksplit == 1 && (Rg || p.accumulate)) {
                long ad[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int slot = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const int row = m0 + wm * WM + ((A_MC && PREC == 0) ? TM * slot + i : i * 32 + slot);
                    if (p.out_mode == OUT_NCHW) {
                        int img = row / p.out_hw, pix = row - img * p.out_hw;
                        ad[r] = ((long)img * p.N + col) * p.out_hw + pix;
                    } else ad[r] = (long)row * p.ldc + col;
                    if (row >= p.M) ad[r] = 0;
                }
                if (Rg) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = Rg[ad[r]];
                }
                if (p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cv[r] = Cg[ad[r]];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int slot = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int row = m0 + wm * WM + ((A_MC && PREC == 0) ? TM * slot + i : i * 32 + slot);              // fp32 k-major A: interleaved row slots
                if (row >= p.M) continue;
                if (p.ksplit > 1) { Cg[(long)row * p.N + col] = acc[i][j][r]; continue; }
                long addr;
                if (p.out_mode == OUT_NCHW) {
                    int img = row / p.out_hw, pix = row - img * p.out_hw;
                    addr = ((long)img * p.N + col) * p.out_hw + pix;
                } else addr = (long)row * p.ldc + col;
                float v = acc[i][j][r] * p.alpha + bv;
                if (Rg) v += rv[r];
                if (p.act == ACT_SILU) v = cdae_silu(v);
                else if (p.act == ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (p.accumulate) v += cv[r];
                Cg[addr] = v;
                if (!__builtin_isfinite(v) && p.range_flag) *p.range_flag = 1;
                if (p.C_hi) store_planes(p, addr, v);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// ps_kernel: the same f16x3 product on operands that are ALREADY split into f16 planes in HBM (activations by the
// producing GroupNorm kernel, weights once per weight version).  The main loop then has no conversion VALU and no
// register staging: every 16-byte piece of a tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4), the planes
// keep the [rows][64 B] image of the in-kernel-split path (16-B chunks XOR-swizzled by (row>>2)&3 — applied on the
// per-lane SOURCE address, the LDS destination of a DMA is lane-linear), and the fragment reads / MFMAs are unchanged.
// One barrier per 32-deep step: wait own DMAs -> barrier -> issue the next stage's DMAs -> 12 MFMAs per wave.
// The conv gather's per-row tap offsets (9 per output pixel, -1 = padding) are computed once into an LDS table.
__device__ __attribute__((aligned(16))) unsigned g_zero_ps[4] = {0u, 0u, 0u, 0u};

// LOADERS > 0: that many extra waves do nothing but issue the DMAs (an LDS-DMA costs its issuing wave 60-180 cycles, four per
// step in a wave that also has 12 MFMAs to issue); the compute waves then only read fragments and issue MFMAs.
// BF: the planes hold bf16 (gradient operands: full fp32 range, 16 significand bits over the two planes) instead of f16.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NPL, int STAGES, int LOADERS = 0, bool BF = false>
__global__ __launch_bounds__(64 * (WAVES_M * WAVES_N + LOADERS)) void ps_kernel(const GemmParams p) {
    constexpr int NCOMP = WAVES_M * WAVES_N, THREADS = 64 * (NCOMP + LOADERS);
    constexpr int LT = LOADERS ? 64 * LOADERS : THREADS;               // threads that issue DMAs
    constexpr int RPP = LT / 4;                                        // tile rows covered by one DMA pass
    static_assert(LOADERS == 0 || STAGES == 2, "loader waves are built for the 2-stage loop");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
    static_assert(STAGES == 2 || STAGES == 3, "2 stages: one __syncthreads per step; 3 stages: DMAs stay in flight across a raw barrier");
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;                // bytes per 16-bit plane
    constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
    constexpr int A_P = BM / RPP, B_P = BN / RPP;                      // 16-byte pieces per thread per plane
    static_assert(A_P >= 1 && B_P >= 1, "tile smaller than one DMA pass");
    typedef const unsigned short* hp;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    int* const taptab = reinterpret_cast<int*>(lds + STAGES * STAGE);  // [taps][BM] element offsets, -1 = zero row

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool is_loader = LOADERS > 0 && wave >= NCOMP;               // wave-uniform role
    const bool loads_here = LOADERS == 0 || is_loader;
    const int ltid = LOADERS ? tid - 64 * NCOMP : tid, lwave = LOADERS ? wave - NCOMP : wave;   // index among the DMA-issuing threads

    const int nmt = (p.M + BM - 1) / BM, nnt = (p.N + BN - 1) / BN;
    int mt, nt, ks;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned q = G >> 3, r = G & 7, x = b & 7;
        unsigned v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        nt = v % nnt; v /= nnt;
        mt = v % nmt; ks = v / nmt;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const hp a_hi = reinterpret_cast<hp>(p.A), a_lo = p.A_lo;
    const hp b_hi = reinterpret_cast<hp>(p.B), b_lo = p.B_lo;
    // padding rows read a zero page.  The select is done on the element OFFSET (zero page expressed relative to each plane):
    // selecting between two pointers makes hipcc branch around two different load forms.
    const hp zero = reinterpret_cast<hp>(g_zero_ps);
    const long za_hi = zero - a_hi, za_lo = NPL == 2 ? zero - a_lo : 0, zb_hi = zero - b_hi, zb_lo = NPL == 2 ? zero - b_lo : 0;

    const int taps = p.amode == A_CONV_VEC ? (p.ps_taps == 4 ? 4 : 9) : 1;
    const int cpt = (taps == 1 ? p.K : p.Cin) / BK;                    // 32-deep steps per tap
    const int nk_total = taps * cpt;
    const int nk_per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * nk_per;
    const int kt_end = min(nk_total, kt_begin + nk_per);

    for (int idx = tid; idx < taps * BM; idx += THREADS) {
        const int tap = idx / BM, row = idx - tap * BM, m = m0 + row;
        int off = -1;
        if (taps == 1) { if (m < p.M) off = m * (int)p.lda; }
        else {
            const PixRow r = make_pixrow(p, m);
            long o;
            // 9 taps: the 3x3 window; 4 taps: the 2x2 window of one sub-pixel phase, shifted by (ph_y, ph_x)
            const int ky = taps == 9 ? tap / 3 : (tap >> 1) + p.ph_y, kx = taps == 9 ? tap - 3 * (tap / 3) : (tap & 1) + p.ph_x;
            const bool ok = tap_offset(p, r, ky, kx, o);
            if (ok && r.ok) off = (int)o;
        }
        if ((PDBG(p) & 1) && off >= 0) off = (row & 15) * 64;
        taptab[idx] = off;
    }

    // B rows are loop invariant: per piece a running pointer (or the zero page for rows >= N)
    long boff[B_P];                // element offset of this thread's piece in the weight planes
    bool bok[B_P];
#pragma unroll
    for (int q = 0; q < B_P; ++q) {
        const int row = ((ltid >> 2) + RPP * q) & (BN - 1), c = (ltid & 3) ^ ((row >> 2) & 3);      // (& mask: compute waves of a loader build carry a dummy index)
        bok[q] = n0 + row < p.N;
        boff[q] = (long)(bok[q] ? n0 + row : 0) * p.ldb + (long)kt_begin * BK + c * 8;
    }
    int acol[A_P];                 // this thread's 16-byte chunk inside the 32-deep k slice, in elements
    int arow[A_P];
#pragma unroll
    for (int q = 0; q < A_P; ++q) { arow[q] = ((ltid >> 2) + RPP * q) & (BM - 1); acol[q] = 8 * ((ltid & 3) ^ ((arow[q] >> 2) & 3)); }

    // LDS-DMA through inline asm (cdae_lds_dma16): with the builtin, hipcc drains the DMAs (vmcnt(0)) in front of the next LDS read that
    // might alias their destination — i.e. right after they were issued, before the MFMAs they were meant to overlap
    auto dma = [&](hp src, char* dst_wave_base) {
        cdae_lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst_wave_base - lds)));
    };
    // issue the DMAs of one 32-deep step (tap, channel offset kc) into `stage`; B pointers advance by one step
    auto issue = [&](int stage, int tap, int kc) {
        char* const sa = lds + stage * STAGE;
        char* const sb = sa + NPL * A_PLANE;
#pragma unroll
        for (int q = 0; q < A_P; ++q) {
            const int off = taptab[tap * BM + arow[q]];
            const long e = (long)off + kc + acol[q];
            const bool ok = off >= 0;
            char* const dst = sa + (q * LT + lwave * 64) * 16;
            dma(a_hi + (ok ? e : za_hi), dst);
            if constexpr (NPL == 2) dma(a_lo + (ok ? e : za_lo), dst + A_PLANE);
        }
#pragma unroll
        for (int q = 0; q < B_P; ++q) {
            char* const dst = sb + (q * LT + lwave * 64) * 16;
            dma(b_hi + (bok[q] ? boff[q] : zb_hi), dst);
            if constexpr (NPL == 2) dma(b_lo + (bok[q] ? boff[q] : zb_lo), dst + B_PLANE);
            boff[q] += (PDBG(p) & 2) ? 0 : BK;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int tap = kt_begin / cpt, chunk = kt_begin - tap * cpt;            // position of the NEXT step to issue
    auto advance = [&]() { ++chunk; const bool wrap = chunk == cpt; chunk = wrap ? 0 : chunk; tap += wrap ? 1 : 0; };

    __syncthreads();                                                   // tap table visible
    constexpr int G = NPL * (A_P + B_P);                               // DMAs per wave per step
    const bool nodma = (PDBG(p) & 4) != 0;
    auto issue_next = [&](int stage) { issue(stage, (PDBG(p) & 1) ? 0 : tap, (PDBG(p) & 1) ? 0 : chunk * BK); advance(); };
    if (loads_here && kt_begin < kt_end) issue_next(0);
    if constexpr (STAGES == 3) { if (kt_begin + 1 < kt_end) issue_next(1); }

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if constexpr (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's DMAs of stage `cur` have landed
            __syncthreads();                                           // ... everyone's have, and stage cur^1 is no longer read
            if (loads_here && kt + 1 < kt_end && !nodma) issue_next(cur ^ 1);
            if (is_loader) { cur ^= 1; continue; }                     // loader waves: back to the barrier
        } else {
            // stage kt landed when at most the G DMAs of stage kt+1 are still outstanding; the barrier also tells that every
            // wave is done reading stage kt-1 == (kt+2) % 3, which the next DMAs overwrite.  No vmcnt(0) in the loop.
            if (kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(PDBG(p) & 8)) __builtin_amdgcn_s_barrier();
            if (kt + 2 < kt_end && !nodma) issue_next(cur == 0 ? 2 : cur - 1);
        }

        const char* ac = lds + cur * STAGE;
        const char* bc = ac + NPL * A_PLANE;
        auto frag = [&](const char* plane, int row0, int sk) -> u16x8 {
            const int row = (PDBG(p) & 16) ? 0 : row0 + l31;             // dbg 16: every lane reads the same 16 bytes (LDS broadcast, no bandwidth)
            return *reinterpret_cast<const u16x8*>(plane + row * 64 + 16 * ((2 * sk + hh) ^ ((row >> 2) & 3)));
        };
        auto mma = [&](const u16x8& x, const u16x8& y, const f32x16& c) -> f32x16 {
            if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
        };
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
            u16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = frag(ac, wm * WM + i * 32, sk);
                if constexpr (NPL == 2) al[i] = frag(ac + A_PLANE, wm * WM + i * 32, sk);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = frag(bc, wn * WN + j * 32, sk);
                if constexpr (NPL == 2) bl[j] = frag(bc + B_PLANE, wn * WN + j * 32, sk);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (NPL == 2) {
                        acc[i][j] = mma(al[i], bh[j], acc[i][j]);        // same order as the in-kernel-split path: bit-identical sums
                        acc[i][j] = mma(ah[i], bl[j], acc[i][j]);
                    }
                    acc[i][j] = mma(ah[i], bh[j], acc[i][j]);
                }
        }
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }

    // ---------------------------------------------------------------- epilogue (as igemm_kernel's, K-contiguous case)
    if (is_loader) return;
    float* __restrict__ Cg;
    const float* __restrict__ Rg = nullptr;
    if (p.ksplit > 1) Cg = p.splitk_ws + (long)ks * (long)p.M * p.N;
    else { Cg = p.C; Rg = p.res; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (p.ksplit == 1 && p.bias) ? p.bias[col] : 0.f;
            float gs = 0.f, gq = 0.f;
            auto out_addr = [&](int row) -> long {
                if (p.out_mode == OUT_NCHW) {
                    int img = row / p.out_hw, pix = row - img * p.out_hw;
                    return ((long)img * p.N + col) * p.out_hw + pix;
                } else if (p.out_mode == OUT_UP2) {
                    const int x = row - fdiv(row, p.wo_magic, p.wo_shift) * p.Wo;          // (n, y, x) -> (n, 2y + ph_y, 2x + ph_x)
                    return (4L * row - 2 * x + p.ph_y * 2 * p.Wo + p.ph_x) * p.ldc + col;
                }
                return (long)row * p.ldc + col;
            };
            // residual / accumulate operands of the sub-tile's 16 rows requested together (as in igemm_kernel's epilogue)
            float rv[16], cv[16];
            if (p.ksplit == 1 && (Rg || p.accumulate)) {
                long ad[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    ad[r] = row < p.M ? out_addr(row) : 0;
                }
                if (Rg) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = Rg[ad[r]];
                }
                if (p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cv[r] = Cg[ad[r]];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row >= p.M) continue;
                if (p.ksplit > 1) { Cg[(long)row * p.N + col] = acc[i][j][r]; continue; }
                const long addr = out_addr(row);
                float v = acc[i][j][r] * p.alpha + bv;
                if (Rg) v += rv[r];
                if (p.act == ACT_SILU) v = cdae_silu(v);
                else if (p.act == ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (p.accumulate) v += cv[r];
                Cg[addr] = v;
                if (!__builtin_isfinite(v) && p.range_flag) *p.range_flag = 1;
                if (p.C_hi) store_planes(p, addr, v);
                gs += v; gq += v * v;
            }
            if (p.gn_part && p.ksplit == 1) {           // this wave owns the whole 32 x 32 sub-tile: one deterministic write per (chunk, column)
                gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32);
                if (hh == 0) {
                    float* o = p.gn_part + ((long)((m0 + wm * WM + i * 32) >> 5) * p.N + col) * 2;
                    o[0] = gs; o[1] = gq;
                }
            }
        }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NPL, int STAGES, int LOADERS = 0, bool BF = false>
int launch_ps(const GemmParams& p, hipStream_t st) {
    const int taps = p.amode == A_CONV_VEC ? (p.ps_taps == 4 ? 4 : 9) : 1;
    constexpr size_t tiles = (size_t)STAGES * NPL * (BM + BN) * 64;
    const size_t smem = tiles + (size_t)taps * BM * sizeof(int);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ps_kernel<BM, BN, WAVES_M, WAVES_N, NPL, STAGES, LOADERS, BF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(tiles + 9 * BM * sizeof(int))) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    dim3 grid((unsigned)((long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.ksplit));
    hipLaunchKernelGGL((ps_kernel<BM, BN, WAVES_M, WAVES_N, NPL, STAGES, LOADERS, BF>), grid, dim3(64 * (WAVES_M * WAVES_N + LOADERS)), smem, st, p);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("ps_kernel launch failed");
}

// ---------------------------------------------------------------------------------------------------------------
// pswin_kernel: stride-1 conv3x3 on pre-split planes with the activation WINDOW resident in LDS.  The 9 taps of a tile of
// 128 consecutive output pixels read the same input pixels shifted by (ky-1)*W + (kx-1), so per 32-channel chunk the block
// loads one window of 128 + 2W + 2 pixel rows ONCE (instead of nine 128-row tiles) and every tap reads its A fragments from
// the window at a row offset; pixels that fall outside the image (or into the neighbouring image of the batch) are masked
// to zero in the fragment registers by a per-lane 9-bit tap mask.  Per step (chunk, tap) only the 128 x 32 weight tile is
// staged (double-buffered).  A-operand traffic drops 4.5x (W = 64) .. 7.9x (W = 8), bytes per step from 32 KB to ~20 KB.
// K order is (chunk, tap, channel) — the sums differ from ps_kernel's (tap, channel) order by fp32 rounding only.
// BST = 3 weight stages (rows up to 32 pixels: the window is small enough for two blocks per CU): the DMAs of step s+2 are in
// flight while step s computes, and the loop waits with a counted vmcnt instead of draining.
// GNA: the window is not copied from pre-split planes but PRODUCED in the kernel from the GroupNorm's fp32 input: every thread
// keeps the next chunk's float4s in registers (loaded one chunk ahead), folds y = silu?(x * a + b) with the per-(image, channel)
// coefficients (a 1-KB LDS-DMA per chunk), splits to f16 hi/lo and writes the same swizzled window image.  This is the
// reference's ResBlock prologue (GroupNorm -> SiLU -> conv3x3, unet.py:187-197) without the normalised tensor ever existing in HBM.
// BM = 256 (4 x 2 waves of 64 x 64): twice the MFMAs per step and per staged weight byte.  Its window is TIGHT — exactly BM + 2W
// rows (384 at W = 64, so that window + two weight stages are 80 KB and two blocks still share a CU): the two corner rows of the
// loose window are only ever read by masked taps when tiles start on an image-row boundary (BM % W == 0), so their reads clamp.
template <int BN, int NPL, int BST, int MAXWIN, int WAVES_N = 4, bool GNA = false, int BM = 128, bool TIGHT = (BM == 256), int BLOCKS = 2, bool BF = false, int WM_ = 64>
__global__ __launch_bounds__(BM / WM_ * WAVES_N * 64, BM / WM_ * WAVES_N * BLOCKS / 4) void pswin_kernel(const GemmParams p) {     // BLOCKS blocks per CU; WM_ = rows of a wave tile
    constexpr int WAVES_M = BM / WM_, NW = WAVES_M * WAVES_N, THREADS = 64 * NW;
    static_assert(!GNA || (BST == 2 && BM == 128), "the fused-GroupNorm window is built for the 2-stage weight ring and 128-row tiles");
    constexpr int G_SLOTS = (MAXWIN * 8 + THREADS - 1) / THREADS;        // float4s of a window chunk per thread
    constexpr int WM = WM_, WN = BN / WAVES_N, TM = WM_ / 32, TN = WN / 32;
    static_assert(MAXWIN % 16 == 0 && (BST == 2 || BST == 3), "window rows come in 16-row DMA blocks");
    constexpr int A_PLANE = MAXWIN * 64, B_PLANE = BN * 64;
    constexpr int A_SLOTS = (2 * (MAXWIN / 16) + NW - 1) / NW;          // 2 planes x 16-row blocks over the block's waves
    constexpr int B_RB = (BN / 16) / NW;                                // 16-row weight blocks per wave
    typedef const unsigned short* hp;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    char* const awin = lds;                                             // [NPL][MAXWIN][64 B]
    char* const bst = lds + NPL * A_PLANE;                              // [BST stages][NPL][BN][64 B]
    char* const coefb = bst + BST * NPL * B_PLANE;                      // GNA: [2][4 images][32 channels][a, b] floats

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int nmt = (p.M + BM - 1) / BM, nnt = (p.N + BN - 1) / BN;
    int mt, nt, ks;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned q = G >> 3, r = G & 7, x = b & 7;
        unsigned v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        nt = v % nnt; v /= nnt;
        mt = v % nmt; ks = v / nmt;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const hp a_hi = reinterpret_cast<hp>(p.A), a_lo = p.A_lo;
    const hp b_hi = reinterpret_cast<hp>(p.B), b_lo = p.B_lo;
    const hp zero = reinterpret_cast<hp>(g_zero_ps);
    const long za_hi = zero - a_hi, za_lo = NPL == 2 ? zero - a_lo : 0, zb_hi = zero - b_hi, zb_lo = NPL == 2 ? zero - b_lo : 0;

    const int W = p.W, win = BM + 2 * W + (TIGHT ? 0 : 2), NB = (win + 15) >> 4;      // window rows, 16-row DMA blocks per plane
    const int pix0 = m0 - W - (TIGHT ? 0 : 1);                          // flattened input pixel of window row 0
    const int nchunk = p.Cin / BK;
    const int c_per = (nchunk + p.ksplit - 1) / p.ksplit;
    const int c_begin = ks * c_per, c_end = min(nchunk, c_begin + c_per);

    // ---- GNA: this thread's float4 slots of a window chunk (item = tid + THREADS q: window row item / 8, channels 4 (item % 8) .. +3)
    int gpix[G_SLOTS], gimg[G_SLOTS];          // flattened pixel (or -1) and image index relative to the window's first image
    float4 xv[G_SLOTS];
    const int pix_first = max(pix0, 0), img_first = fdiv(pix_first, p.hw_magic, p.hw_shift), img_last = fdiv(p.M - 1, p.hw_magic, p.hw_shift);
    if constexpr (GNA) {
#pragma unroll
        for (int q = 0; q < G_SLOTS; ++q) {
            const int item = tid + THREADS * q, j = item >> 3, pix = pix0 + j;
            const bool ok = j < win && pix >= 0 && pix < p.M;
            gpix[q] = ok ? pix : -1;
            gimg[q] = ok ? min(fdiv(pix, p.hw_magic, p.hw_shift) - img_first, 3) : 0;
        }
    }
    const float* const gzero = reinterpret_cast<const float*>(g_zero_ps);
    auto gna_load = [&](int chunk) {           // issue the global loads of one chunk of the window into registers
        const int c0 = chunk * BK + (tid & 7) * 4;
        const bool second = p.A2 && c0 >= p.K1;
        const float* src = second ? p.A2 : p.A;
        const long pitch = second ? p.lda2 : p.lda;
        const int cc = second ? c0 - p.K1 : c0;
#pragma unroll
        for (int q = 0; q < G_SLOTS; ++q)
            xv[q] = ld4(gpix[q] >= 0 ? src + (long)gpix[q] * pitch + cc : gzero);
    };
    auto gna_coef_dma = [&](int chunk, int buf) {      // wave 0: (a, b) of 4 images x 32 channels -> 1 KB of LDS
        if (wave == 0) {
            const int img = min(img_first + (lane >> 4), img_last);
            const float* src = p.gn_coef + ((long)img * p.Cin + chunk * BK) * 2 + (lane & 15) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(coefb + buf * 1024), 16, 0, 0);
        }
    };
    auto gna_stage = [&](int buf) {            // registers -> normalise (+SiLU) -> f16 hi/lo -> swizzled window image
#pragma unroll
        for (int q = 0; q < G_SLOTS; ++q) {
            const int item = tid + THREADS * q, j = item >> 3, f4 = item & 7;
            if (j < MAXWIN) {
                const float4* cf = reinterpret_cast<const float4*>(coefb + buf * 1024 + gimg[q] * 256 + f4 * 32);
                const float4 c01 = cf[0], c23 = cf[1];              // a0 b0 a1 b1 | a2 b2 a3 b3
                float y0 = fmaf(xv[q].x, c01.x, c01.y), y1 = fmaf(xv[q].y, c01.z, c01.w);
                float y2 = fmaf(xv[q].z, c23.x, c23.y), y3 = fmaf(xv[q].w, c23.z, c23.w);
                if (p.gn_silu) { y0 = cdae_silu(y0); y1 = cdae_silu(y1); y2 = cdae_silu(y2); y3 = cdae_silu(y3); }
                asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));      // opaque before the split (attention.hip split8)
                half4 hi, lo;
                hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
                const int off = j * 64 + 16 * ((f4 >> 1) ^ ((j >> 2) & 3)) + 8 * (f4 & 1);
                *reinterpret_cast<half4*>(awin + off) = hi;
                if constexpr (NPL == 2) {
                    lo[0] = (_Float16)(y0 - (float)hi[0]); lo[1] = (_Float16)(y1 - (float)hi[1]);
                    lo[2] = (_Float16)(y2 - (float)hi[2]); lo[3] = (_Float16)(y3 - (float)hi[3]);
                    *reinterpret_cast<half4*>(awin + A_PLANE + off) = lo;
                }
            }
        }
    };

    // ---- A window DMA slots of this wave: piece pc = wave + 8q covers plane pc / NB, rows 16 (pc % NB) .. +15
    int aoff[A_SLOTS];             // element offset of this lane's 16 bytes at chunk 0, or -1 (outside the tensor)
#pragma unroll
    for (int q = 0; q < A_SLOTS; ++q) {
        const int pc = wave + NW * q, pl = pc >= NB ? 1 : 0, rb = pc - pl * NB;
        const int j = rb * 16 + (lane >> 2);                             // window row
        const int pix = pix0 + j;                                        // flattened input pixel (n, y, x)
        const int c = (lane & 3) ^ ((j >> 2) & 3);
        aoff[q] = (pix >= 0 && pix < p.M && j < win) ? pix * (int)p.sx + c * 8 : -1;
    }
    // ---- B tile pieces: wave w stages the 16-row blocks w, w + NW, ... (hi and lo)
    long boff[B_RB];
    bool bok[B_RB];
#pragma unroll
    for (int q = 0; q < B_RB; ++q) {
        const int row = (wave + NW * q) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        bok[q] = n0 + row < p.N;
        boff[q] = (long)(bok[q] ? n0 + row : 0) * p.ldb + c * 8;
    }
    // LDS-DMA through inline asm (cdae_lds_dma16): with the builtin, hipcc drains the DMAs (vmcnt(0)) in front of the next LDS read that
    // might alias their destination — i.e. right after they were issued, before the MFMAs they were meant to overlap
    auto dma = [&](hp src, char* dst_wave_base) {
        cdae_lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst_wave_base - lds)));
    };
    auto issue_A = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < A_SLOTS; ++q) {
            const int pc = wave + NW * q;
            if (pc < NPL * NB) {                                         // wave-uniform
                const int pl = pc >= NB ? 1 : 0, rb = pc - pl * NB;
                const bool ok = aoff[q] >= 0;
                const long e = (long)aoff[q] + chunk * BK;
                char* const dst = awin + pl * A_PLANE + rb * 1024;
                if (pl == 0) dma(a_hi + (ok ? e : za_hi), dst);
                else dma(a_lo + (ok ? e : za_lo), dst);
            }
        }
    };
    auto issue_B = [&](int stage, int chunk, int tap) {
#pragma unroll
        for (int q = 0; q < B_RB; ++q) {
            char* const dst = bst + stage * (NPL * B_PLANE) + (wave + NW * q) * 1024;
            const long e = boff[q] + (long)tap * p.Cin + chunk * BK;
            dma(b_hi + (bok[q] ? e : zb_hi), dst);
            if constexpr (NPL == 2) dma(b_lo + (bok[q] ? e : zb_lo), dst + B_PLANE);
        }
    };

    // ---- per-lane tap masks of the wave's two 32-row sub-tiles: bit (3 ky + kx) set when tap (ky, kx) reads a real pixel
    int tapmask[TM], jrow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31, m = m0 + r;
        jrow[i] = r;                                                     // window row of tap (0, 0); tap (ky, kx) adds ky*W + kx
        const PixRow pr = make_pixrow(p, m);                            // iy0 = y - 1, ix0 = x - 1 (stride 1)
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ty = pr.iy0 + t / 3, tx = pr.ix0 + t % 3;
            mk |= (pr.ok && ty >= 0 && ty < p.H && tx >= 0 && tx < W) ? (1 << t) : 0;
        }
        tapmask[i] = mk;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mma = [&](const u16x8& x, const u16x8& y, const f32x16& c) -> f32x16 {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    if (c_begin < c_end) {
        if constexpr (GNA) {
            gna_load(c_begin); gna_coef_dma(c_begin, 0); issue_B(0, c_begin, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                             // coefficients visible
            gna_stage(0);
            if (c_begin + 1 < c_end) { gna_load(c_begin + 1); gna_coef_dma(c_begin + 1, 1); }
        } else {
            issue_A(c_begin); issue_B(0, c_begin, 0);
            if constexpr (BST == 3) issue_B(1, c_begin, 1);
        }
    }
    // diagnostic build only (CDAE_PS_DBG & 32): s_memtime stamps around the three segments of a step, summed per wave and
    // written to the split-K workspace by lane 0 of every wave of the first 64 blocks.  The stamps' lgkmcnt(0) serialises what the
    // real kernel overlaps: read the SHARES, never the run time of this mode.
    const bool stamps = (PDBG(p) & 32) != 0;
    unsigned long long t_wait = 0, t_issue = 0, t_comp = 0, t_prev = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    if (stamps) t_prev = stamp();
    int stage = 0;
    // ntaps = 9: the 3x3 window; 4: the 2x2 window of one sub-pixel phase (ph_y, ph_x) of an upsample + conv — tap t reads window
    // position (t / 2 + ph_y, t % 2 + ph_x) of the same 3x3 neighbourhood, so window, masks and row shifts are shared.
    const int ntaps = p.ps_taps == 4 ? 4 : 9, lasttap = ntaps - 1;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
#pragma unroll 1
        for (int tap = 0; tap < ntaps; ++tap) {       // not unrolled: nine copies keep every tap's addresses and masks live (197 VGPRs)
            if constexpr (BST == 2) {
                // GNA, tap 0: the weight DMAs are older than the next chunk's register prefetch (<= G_SLOTS loads + one coefficient
                // DMA issued after them), so a counted wait retires the weights and leaves the prefetch in flight
                if (GNA && tap == 0 && chunk + 1 < c_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GNA ? G_SLOTS : 0) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (GNA) {         // raw barrier: __syncthreads() would drain the register prefetch with its own vmcnt(0)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else __syncthreads();
                if (stamps) { const unsigned long long t = stamp(); t_wait += t - t_prev; t_prev = t; }
                const int ntap = tap == lasttap ? 0 : tap + 1, nchk = tap == lasttap ? chunk + 1 : chunk;     // stage the next step's weight tile
                if (nchk < c_end) issue_B(stage ^ 1, nchk, ntap);
                if (stamps) { const unsigned long long t = stamp(); t_issue += t - t_prev; t_prev = t; }
            } else {
                // weights of this step landed when only the next step's NPL pieces may still be in flight; at tap 0 the window
                // (issued last) must be complete too, and on the very last step nothing younger exists: drain.
                const bool last = chunk + 1 == c_end && tap == lasttap;
                if (tap == 0 || last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL * B_RB) : "memory");
                __builtin_amdgcn_s_barrier();
                if (!(PDBG(p) & 64)) {
                    const int t2 = tap + 2, ntap = t2 >= ntaps ? t2 - ntaps : t2, nchk = t2 >= ntaps ? chunk + 1 : chunk;
                    if (nchk < c_end) issue_B(stage >= 1 ? stage - 1 : 2, nchk, ntap);            // (stage + 2) % 3
                }
            }
            const int ky = ntaps == 9 ? tap / 3 : (tap >> 1) + p.ph_y, kx = ntaps == 9 ? tap - 3 * ky : (tap & 1) + p.ph_x;
            const int wtap = ky * 3 + kx, shift = ky * W + kx - (TIGHT ? 1 : 0);          // wtap: position in the 3x3 neighbourhood (mask bit)
            const char* bc = bst + stage * (NPL * B_PLANE);
            unsigned amask[TM];
            int abase[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                amask[i] = (tapmask[i] >> wtap) & 1 ? 0xffffffffu : 0u;
                int j = jrow[i] + shift;
                if constexpr (TIGHT) j = min(max(j, 0), win - 1);            // the clamped reads belong to masked taps
                abase[i] = j * 64 + 16 * (hh ^ ((j >> 2) & 3));           // sk = 0 chunk; sk = 1 flips chunk bit 1 (+-32 bytes)
            }
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                u16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int a = abase[i] ^ (sk * 32);
                    u32x4 h4 = *reinterpret_cast<const u32x4*>(awin + a);
                    h4 &= amask[i];
                    ah[i] = __builtin_bit_cast(u16x8, h4);
                    if constexpr (NPL == 2) {
                        u32x4 l4 = *reinterpret_cast<const u32x4*>(awin + A_PLANE + a);
                        l4 &= amask[i];
                        al[i] = __builtin_bit_cast(u16x8, l4);
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WN + j * 32 + l31;
                    const int b = row * 64 + 16 * ((2 * sk + hh) ^ ((row >> 2) & 3));
                    bh[j] = *reinterpret_cast<const u16x8*>(bc + b);
                    if constexpr (NPL == 2) bl[j] = *reinterpret_cast<const u16x8*>(bc + B_PLANE + b);
                }
                if (!(PDBG(p) & 128)) __builtin_amdgcn_s_setprio(1);     // MFMA bursts win issue arbitration over other waves' staging work (+1-2 %)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (NPL == 2) {
                            acc[i][j] = mma(al[i], bh[j], acc[i][j]);
                            acc[i][j] = mma(ah[i], bl[j], acc[i][j]);
                        }
                        acc[i][j] = mma(ah[i], bh[j], acc[i][j]);
                    }
                if (!(PDBG(p) & 128)) __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (BST == 3) {
                if (PDBG(p) & 64) {     // late issue: the step's own MFMAs are queued before this wave joins the block's DMA burst
                    const int t2 = tap + 2, ntap = t2 >= ntaps ? t2 - ntaps : t2, nchk = t2 >= ntaps ? chunk + 1 : chunk;
                    if (nchk < c_end) issue_B(stage >= 1 ? stage - 1 : 2, nchk, ntap);
                }
            }
            stage = stage + 1 == BST ? 0 : stage + 1;
            if (stamps) { const unsigned long long t = stamp(); t_comp += t - t_prev; t_prev = t; }
        }
        if (chunk + 1 < c_end) {
            if constexpr (GNA) {
                __syncthreads();                                         // every wave is done with this chunk's window
                gna_stage((chunk + 1 - c_begin) & 1);                    // its registers and coefficients landed steps ago
                if (chunk + 2 < c_end) { gna_load(chunk + 2); gna_coef_dma(chunk + 2, (chunk + 2 - c_begin) & 1); }
            } else {
                __builtin_amdgcn_s_barrier();                            // every wave is done with this chunk's window
                issue_A(chunk + 1);                                      // lands before the vmcnt(0) + barrier of the next step
            }
        }
    }

    if (stamps && blockIdx.x < 64 && lane == 0 && p.splitk_ws && p.ksplit == 1) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.splitk_ws) + (blockIdx.x * NW + wave) * 4;
        o[0] = t_wait; o[1] = t_issue; o[2] = t_comp; o[3] = stamp() - t_prev;
    }
    // ---------------------------------------------------------------- epilogue (row-major result)
    float* __restrict__ Cg;
    const float* __restrict__ Rg = nullptr;
    if (p.ksplit > 1) Cg = p.splitk_ws + (long)ks * (long)p.M * p.N;
    else { Cg = p.C; Rg = p.res; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = (p.ksplit == 1 && p.bias) ? p.bias[col] : 0.f;
            float gs = 0.f, gq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row >= p.M) continue;
                if (p.ksplit > 1) { Cg[(long)row * p.N + col] = acc[i][j][r]; continue; }
                long addr;
                if (p.out_mode == OUT_UP2) {
                    const int x = row - fdiv(row, p.wo_magic, p.wo_shift) * p.Wo;          // (n, y, x) -> (n, 2y + ph_y, 2x + ph_x)
                    addr = (4L * row - 2 * x + p.ph_y * 2 * p.Wo + p.ph_x) * p.ldc + col;
                } else addr = (long)row * p.ldc + col;
                float v = acc[i][j][r] * p.alpha + bv;
                if (Rg) v += Rg[addr];
                if (p.act == ACT_SILU) v = cdae_silu(v);
                else if (p.act == ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (p.accumulate) v += Cg[addr];
                Cg[addr] = v;
                if (!__builtin_isfinite(v) && p.range_flag) *p.range_flag = 1;
                if (p.C_hi) store_planes(p, addr, v);
                gs += v; gq += v * v;
            }
            if (p.gn_part && p.ksplit == 1) {
                gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32);
                if (hh == 0) {
                    float* o = p.gn_part + ((long)((m0 + wm * WM + i * 32) >> 5) * p.N + col) * 2;
                    o[0] = gs; o[1] = gq;
                }
            }
        }
}

template <int BN, int NPL, int BST, int MAXWIN, int WAVES_N = 4, bool GNA = false, int BM = 128, bool TIGHT = (BM == 256), int BLOCKS = 2, bool BF = false, int WM_ = 64>
int launch_pswin(const GemmParams& p, hipStream_t st) {
    constexpr size_t smem = (size_t)NPL * MAXWIN * 64 + (size_t)BST * NPL * BN * 64 + (GNA ? 2048 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pswin_kernel<BN, NPL, BST, MAXWIN, WAVES_N, GNA, BM, TIGHT, BLOCKS, BF, WM_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    dim3 grid((unsigned)((long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.ksplit));
    static const size_t pad = getenv("CDAE_PS_PAD_LDS") ? (size_t)atoi(getenv("CDAE_PS_PAD_LDS")) : 0;       // dev: force fewer blocks per CU
    if (pad) hipFuncSetAttribute(reinterpret_cast<const void*>(&pswin_kernel<BN, NPL, BST, MAXWIN, WAVES_N, GNA, BM, TIGHT, BLOCKS, BF, WM_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem + pad));
    hipLaunchKernelGGL((pswin_kernel<BN, NPL, BST, MAXWIN, WAVES_N, GNA, BM, TIGHT, BLOCKS, BF, WM_>), grid, dim3(BM / WM_ * WAVES_N * 64), smem + pad, st, p);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("pswin_kernel launch failed");
}

// split-K finish: C = alpha * sum_s slab[s] + bias (+res) (-> act), deterministic order
__global__ void splitk_reduce_kernel(const GemmParams p) {
    long total = (long)p.batch * p.M * p.N;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long bz = idx / ((long)p.M * p.N), rem = idx - bz * (long)p.M * p.N;
        int row = rem / p.N, col = rem - (long)row * p.N;
        float s = 0.f;
        const float* slab = p.splitk_ws + bz * (long)p.M * p.N + rem;
        const long kstride = (long)p.batch * p.M * p.N;
        int k = 0;
        for (; k + 4 <= p.ksplit; k += 4) {          // four slab loads in flight (the dependent-add loop alone waits a round trip per slab); same order of additions
            const float v0 = slab[k * kstride], v1 = slab[(k + 1) * kstride], v2 = slab[(k + 2) * kstride], v3 = slab[(k + 3) * kstride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < p.ksplit; ++k) s += slab[k * kstride];
        int bo = bz / p.batch_inner, bi = bz - bo * p.batch_inner;
        float* Cg = p.C + bo * p.c_bs0 + bi * p.c_bs1;
        long addr;
        if (p.out_mode == OUT_NCHW) {
            int img = row / p.out_hw, pix = row - img * p.out_hw;
            addr = ((long)img * p.N + col) * p.out_hw + pix;
        } else if (p.out_mode == OUT_UP2) {
            const int x = row - fdiv(row, p.wo_magic, p.wo_shift) * p.Wo;
            addr = (4L * row - 2 * x + p.ph_y * 2 * p.Wo + p.ph_x) * p.ldc + col;
        } else addr = (long)row * p.ldc + col;
        float v = s * p.alpha + (p.bias ? p.bias[col] : 0.f);
        if (p.res) v += (p.res + bo * p.c_bs0 + bi * p.c_bs1)[addr];
        if (p.act == ACT_SILU) v = cdae_silu(v);
        else if (p.act == ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
        if (p.accumulate) v += Cg[addr];
        Cg[addr] = v;
        if (!__builtin_isfinite(v) && p.range_flag) *p.range_flag = 1;
        if (p.C_hi) store_planes(p, addr, v);
    }
}

// The same finish, four columns per thread (row-major output, one batch, N and every pitch a multiple of 4, 16-byte aligned
// pointers): one integer division per float4 instead of two per element, 16-byte slab loads.  Additions in splitk_reduce_kernel's
// order, so the two kernels are interchangeable bit for bit.
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const GemmParams p) {
    const int n4 = p.N >> 2;
    const long total4 = (long)p.M * n4, kstride4 = total4;
    const float4* __restrict__ ws = reinterpret_cast<const float4*>(p.splitk_ws);
    bool bad = false;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / n4), c4 = (int)(idx - (long)row * n4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* slab = ws + idx;
        int k = 0;
        for (; k + 4 <= p.ksplit; k += 4) {
            const float4 v0 = slab[k * kstride4], v1 = slab[(k + 1) * kstride4], v2 = slab[(k + 2) * kstride4], v3 = slab[(k + 3) * kstride4];
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; k < p.ksplit; ++k) { const float4 v = slab[k * kstride4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        const long addr = (long)row * p.ldc + 4 * c4;
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float v[4] = {s.x * p.alpha + b.x, s.y * p.alpha + b.y, s.z * p.alpha + b.z, s.w * p.alpha + b.w};
        if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + addr); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        if (p.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = cdae_silu(v[e]);
        } else if (p.act == ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
        }
        if (p.accumulate) { const float4 c = *reinterpret_cast<const float4*>(p.C + addr); v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; }
        *reinterpret_cast<float4*>(p.C + addr) = make_float4(v[0], v[1], v[2], v[3]);
        bad |= !__builtin_isfinite(v[0] + v[1] + v[2] + v[3]);
        if (p.C_hi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_planes(p, addr + e, v[e]);
        }
    }
    if (bad && p.range_flag) *p.range_flag = 1;
}

template <int BM, int BN, int AMODE, int BMODE, bool SCALAR, int WAVES_N = 2, int PREC = 0, bool GNS = false, bool DEEP = false>
int launch(const GemmParams& p, hipStream_t st) {
    constexpr bool A_MC = (AMODE == A_PLAIN_MC);
    constexpr bool B_MC = (BMODE != B_PLAIN_KC);
    constexpr int NPL = (PREC == 1 || PREC == 2) ? 2 : 1;
    constexpr int A_TILE = PREC ? NPL * (A_MC ? BK * (BM + 32) : BM * 32) / 2 : (A_MC ? BK * (BM + 4) : BM * LDK);
    constexpr int B_TILE = PREC ? NPL * (B_MC ? BK * (BN + 32) : BN * 32) / 2 : (B_MC ? BK * (BN + 4) : BN * LDK);
    constexpr size_t tiles = 2 * (A_TILE + B_TILE) * sizeof(float);
    constexpr size_t tab_max = GNS ? 16 * 1024 : 0;          // coefficient table: 16 bytes x K, K <= 1024 (two blocks of 80 KB still share a CU)
    static bool attr_done = false;      // per-instantiation; value is idempotent so a race is benign
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, AMODE, BMODE, SCALAR, WAVES_N, PREC, GNS, DEEP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(tiles + tab_max)) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    GemmParams q = p;
    size_t smem = tiles;
    if constexpr (GNS) {
        // a 128-row tile touches at most two images when an image has >= 64 rows... in general ceil(128 / hw) + 1: table only for hw >= 128 or hw == 64
        static const int cfg_tab = getenv("CDAE_GNS_TAB") ? atoi(getenv("CDAE_GNS_TAB")) : 1;
        const bool two = p.hw >= BM || (p.hw * 2 == BM);
        q.gn_tab = cfg_tab && two && (size_t)16 * p.K <= tab_max && p.K % 4 == 0 && p.ksplit == 1 && p.batch == 1;
        if (q.gn_tab) smem += (size_t)16 * p.K;
    }
    dim3 grid((unsigned)((long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batch * p.ksplit));
    hipLaunchKernelGGL((igemm_kernel<BM, BN, AMODE, BMODE, SCALAR, WAVES_N, PREC, GNS, DEEP>), grid, dim3(128 * WAVES_N), smem, st, q);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("igemm launch failed");
}

template <int AMODE, int BMODE>
int launch_tiles(const GemmParams& p, int big, bool scalar, hipStream_t st) {
    if (scalar) return launch<64, 64, AMODE, BMODE, true>(p, st);
    if constexpr (AMODE == A_CONV_GEN) return cdae_fail("A_CONV_GEN is a scalar-only loader");
    else {
        if constexpr (AMODE == A_PLAIN_KC && BMODE == B_PLAIN_KC) {
            static const int cfg_deep = getenv("CDAE_IGEMM_DEEP") ? atoi(getenv("CDAE_IGEMM_DEEP")) : 15;      // bit 0 plain fwd GEMM, 1 the GroupNorm-carrying skip GEMM, 2 linear dgrad / wgrad, 3 the same two on 64x64 tiles (2048x512x512: 19.7 -> 16.5 us)
            if (p.S_hi) {
                if (p.prec != 1 || !big) return cdae_fail("GroupNorm side output: f16x3 mode and a grid of 128x128 tiles required");
                return (cfg_deep & 2) ? launch<128, 128, AMODE, BMODE, false, 4, 1, true, true>(p, st) : launch<128, 128, AMODE, BMODE, false, 4, 1, true>(p, st);
            }
            if (p.prec == 1 && big && (cfg_deep & 1)) return launch<128, 128, AMODE, BMODE, false, 4, 1, false, true>(p, st);
            if (p.prec == 1 && !big && (cfg_deep & 8)) return launch<64, 64, AMODE, BMODE, false, 2, 1, false, true>(p, st);
        }
        if constexpr ((AMODE == A_PLAIN_KC || AMODE == A_PLAIN_MC) && BMODE == B_PLAIN_MC) {     // linear / 1x1 dgrad and wgrad: the same latency-bound shape
            static const int cfg_deep2 = getenv("CDAE_IGEMM_DEEP") ? atoi(getenv("CDAE_IGEMM_DEEP")) : 15;
            if (p.prec == 2 && big && (cfg_deep2 & 4)) return launch<128, 128, AMODE, BMODE, false, 4, 2, false, true>(p, st);
            if (p.prec == 2 && !big && (cfg_deep2 & 8)) return launch<64, 64, AMODE, BMODE, false, 2, 2, false, true>(p, st);
        }
        if (p.prec == 1) return big ? launch<128, 128, AMODE, BMODE, false, 4, 1>(p, st) : launch<64, 64, AMODE, BMODE, false, 2, 1>(p, st);
        if (p.prec == 2) return big ? launch<128, 128, AMODE, BMODE, false, 4, 2>(p, st) : launch<64, 64, AMODE, BMODE, false, 2, 2>(p, st);
        if (p.prec == 3) return big ? launch<128, 128, AMODE, BMODE, false, 4, 3>(p, st) : launch<64, 64, AMODE, BMODE, false, 2, 3>(p, st);
        if (p.prec == 4) return big ? launch<128, 128, AMODE, BMODE, false, 4, 4>(p, st) : launch<64, 64, AMODE, BMODE, false, 2, 4>(p, st);
        if (big && p.waves8) return launch<128, 128, AMODE, BMODE, false, 4>(p, st);
        return big ? launch<128, 128, AMODE, BMODE, false>(p, st) : launch<64, 64, AMODE, BMODE, false>(p, st);
    }
}

}  // namespace

namespace {
int g_default_prec = -1;      // -1: not yet initialised (env CDAE_IGEMM_PREC, else f16x3)
}
extern "C" int cdae_get_default_precision(void) {
    if (g_default_prec < 0) g_default_prec = getenv("CDAE_IGEMM_PREC") ? atoi(getenv("CDAE_IGEMM_PREC")) : CDAE_PREC_F16X3;
    return g_default_prec;
}
extern "C" int cdae_set_default_precision(int prec) {
    if (prec != CDAE_PREC_FP32 && prec != CDAE_PREC_F16X3 && prec != CDAE_PREC_MIXED16) return cdae_fail("unknown precision mode");
    g_default_prec = prec;
    return 0;
}

// Heuristics: 128x128 tiles when they fill the chip (>= ~1 block per CU), else 64x64; split-K when even
// 64x64 tiles leave CUs idle and K is deep (low-resolution levels at small batch, wgrad).
int cdae_gemm_dispatch(GemmParams p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return 0;
    if (p.batch_inner <= 0) p.batch_inner = 1;
    const long tiles_big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.batch;
    const long tiles_small = (long)((p.M + 63) / 64) * ((p.N + 63) / 64) * p.batch;
    const int nk0 = (p.K + BK - 1) / BK;
    // 128x128 tiles when they fill the chip, counting the K-splits a deep reduction allows (wgrad: K = all pixels)
    const long can_split = (p.ksplit_auto && p.splitk_ws) ? (nk0 / 8 > 64 ? 64 : (nk0 / 8 < 1 ? 1 : nk0 / 8)) : 1;
    int big = tiles_big * can_split >= 192 && p.M >= 96 && p.N >= 96;
    // conv gather constants (see make_pixrow / tap_offset)
    if (p.Ho > 0 && p.Wo > 0) {
        auto magic = [](unsigned d, unsigned& m, int& sh) {
            sh = 0;
            while ((1u << sh) < d) ++sh;
            m = (unsigned)(((unsigned long long)((1ull << sh) - d) << 32) / d) + 1u;
        };
        p.hw = p.Ho * p.Wo;
        magic((unsigned)p.hw, p.hw_magic, p.hw_shift);
        magic((unsigned)p.Wo, p.wo_magic, p.wo_shift);
        if (p.tconv) { p.g_mul = 1; p.g_add = 1; p.g_sign = -1; p.g_pm = 1; p.g_sh = 1; }
        else { p.g_mul = p.stride; p.g_add = -1; p.g_sign = 1; p.g_pm = 0; p.g_sh = p.up ? 1 : 0; }
    }
    if (p.force_tile == 64) big = 0;
    if (p.force_tile == 128) big = 1;
    static const bool cfg_dev = getenv("CDAE_GEMM_DEV") != nullptr;          // dev sweeps (tools/gemm_sweep.py): tile and split from the environment, per call
    if (cfg_dev && !p.presplit) {
        const char* e = getenv("CDAE_TILE_FORCE");
        if (e && atoi(e) == 64) big = 0;
        if (e && atoi(e) == 128 && p.M >= 96 && p.N >= 96) big = 1;
        e = getenv("CDAE_KS_FORCE");
        if (e && atoi(e) > 0 && p.ksplit_auto && p.splitk_ws) p.ksplit_force = atoi(e);
    }
    static const int cfg_waves8 = getenv("CDAE_IGEMM_WAVES8") ? atoi(getenv("CDAE_IGEMM_WAVES8")) : 1;   // 8-wave 128x128 tiles by default
    p.waves8 = cfg_waves8;
    // precision: fp32 mode -> fp32 MFMA everywhere; split mode -> f16x3 for activation x weight GEMMs, bf16x3 when an
    // operand is a gradient (api.hip marks those with grad_operand)
    if (p.prec < 0) {
        const int mode = cdae_get_default_precision();
        p.prec = mode == CDAE_PREC_FP32 ? 0 : mode == CDAE_PREC_MIXED16 ? (p.grad_operand ? 4 : 3) : (p.grad_operand ? 2 : 1);
    }
    if (p.A2 && !p.gn_coef && (p.amode != A_PLAIN_KC || p.a_scalar || p.K1 % BK || p.batch != 1)) return cdae_fail("two-source A: vectorised A_PLAIN_KC only, K1 % 32 == 0");
    p.range_flag = cdae_range_flag_ptr();
    if (p.presplit) {
        static const int cfg_dbg = getenv("CDAE_PS_DBG") ? atoi(getenv("CDAE_PS_DBG")) : 0;
        p.dbg = cfg_dbg;
        if (!((p.amode == A_CONV_VEC || p.amode == A_PLAIN_KC) && p.bmode == B_PLAIN_KC) || p.batch != 1)
            return cdae_fail("pre-split operands: only K-contiguous conv / plain GEMMs without batch");
        const int kin = p.amode == A_CONV_VEC ? p.Cin : p.K;
        if (kin % BK || p.K % BK || p.ldb % 8 || (p.amode == A_PLAIN_KC && p.lda % 8) || (p.prec != 1 && p.prec != 2 && p.prec != 3))
            return cdae_fail("pre-split operands: need K (Cin) % 32 == 0, 16-byte aligned rows and a 16-bit split precision mode");
        if (p.prec == 2 && (p.gn_coef || p.ps_taps == 4)) return cdae_fail("pre-split bf16 planes: plain conv3x3 / GEMM only");
    }
    if (p.amode == A_PLAIN_MC && p.M % 4) p.a_scalar = 1;          // vector k-major loaders read 4 rows / columns at once
    if (p.bmode != B_PLAIN_KC && p.N % 4) p.b_scalar = 1;
    const bool scalar = p.a_scalar || p.b_scalar || p.amode == A_CONV_GEN;
    if (scalar) big = 0;
    if (p.bmode == B_CONV_MC && !scalar) {           // a vectorised wgrad block must sit inside one tap
        if (p.Cin % 128 != 0) big = 0;
        if (p.Cin % 64 != 0) return cdae_fail("B_CONV_MC needs Cin % 64 == 0");
    }
    const long tiles = big ? tiles_big : tiles_small;
    const int nk = (p.K + BK - 1) / BK;
    int ks = 1;
    static const int cfg_mintiles = getenv("CDAE_KS_MINTILES") ? atoi(getenv("CDAE_KS_MINTILES")) : 256;
    static const int cfg_minnk = getenv("CDAE_KS_MINNK") ? atoi(getenv("CDAE_KS_MINNK")) : 8;
    if (p.ksplit_auto && p.splitk_ws && tiles < cfg_mintiles && nk >= cfg_minnk) {
        ks = (int)((512 + tiles - 1) / tiles);
        if (ks > nk / 4) ks = nk / 4;
        if (ks > 128) ks = 128;             // (a 128 x 128 weight gradient over 131072 pixels: 64 splits 77 us, 128 splits 60 us)
        // 128 x 128 tiles run two blocks per CU (512 slots): 96 tiles x 6 splits = 576 blocks need a second, nearly empty round where
        // x 5 = 480 do not.  Smallest rounds x (K-steps per split + ~8 steps of prologue / epilogue) + finish; CDAE_KS_ROUNDS=0: plain ceil
        static const int cfg_rounds = getenv("CDAE_KS_ROUNDS") ? atoi(getenv("CDAE_KS_ROUNDS")) : 1;
        if (cfg_rounds && big && ks > 1) {
            long best_cost = -1; int best = 1;
            for (int k = 1; k <= ks; ++k) {
                const int per = (nk + k - 1) / k, kk = (nk + per - 1) / per;
                // a split also pays for the finish launch and the slab round trip: measured ~7 us = a dozen K-steps (8192 x 384 x 384: 22 us unsplit, 29 us in two)
                const long cost = ((tiles * kk + 511) / 512) * (per + 8) + (kk > 1 ? 12 + kk / 4 : 0);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = kk; }      // kk: no empty splits
            }
            ks = best;
        }
        size_t need = (size_t)ks * p.batch * p.M * p.N * sizeof(float);
        while (ks > 1 && need > p.splitk_ws_bytes) { --ks; need = (size_t)ks * p.batch * p.M * p.N * sizeof(float); }
        if (ks < 1) ks = 1;
    }
    if (p.ksplit_force > 0) ks = p.ksplit_force;
    if (ks > 1 && (!p.splitk_ws || (size_t)ks * p.batch * p.M * p.N * sizeof(float) > p.splitk_ws_bytes))
        return cdae_fail("split-K workspace too small");
    p.ksplit = ks;
    if (p.gn_part && (ks > 1 || !p.presplit || (p.out_mode != OUT_ROWMAJOR && p.out_mode != OUT_UP2) || p.accumulate))
        return cdae_fail("GroupNorm partial sums from the epilogue need a pre-split, unsplit-K, row-major, non-accumulating launch");

    cdae_prof_begin(PROF_IGEMM, 2.0 * p.M * p.N * (double)p.K * p.batch * (p.nphase > 1 ? p.nphase : 1), st);
    if (cdae_prof_on()) {
        char tag[128];
        snprintf(tag, sizeof(tag), "gemm M=%d N=%d K=%d b=%d a%d b%d %s ks=%d prec=%d ps=%d taps=%d res=%d gn=%d", p.M, p.N, p.K, p.batch, p.amode, p.bmode, big ? "128" : "64", ks, p.prec,
                 p.presplit, p.ps_taps, p.res != nullptr, p.gn_part != nullptr);
        cdae_prof_tag(tag);
    }
    int rc = -1;
#define CASE(AM, BM_) if (p.amode == AM && p.bmode == BM_) rc = launch_tiles<AM, BM_>(p, big, scalar, st)
    if (p.presplit) {
        static const int cfg_tile = getenv("CDAE_PS_TILE") ? atoi(getenv("CDAE_PS_TILE")) : 128;   // 256: measured 6 % slower end to end (1 block per CU)
        const long tiles_256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
        const bool huge = cfg_tile == 256 && big && tiles_256 * ks >= 200;            // 256x128 tiles, 1 block / CU, 3-stage DMA ring
        static const int cfg_win = getenv("CDAE_PS_WIN") ? atoi(getenv("CDAE_PS_WIN")) : 1;
        static const int cfg_subpix = getenv("CDAE_PS_WIN_SUBPIX") ? atoi(getenv("CDAE_PS_WIN_SUBPIX")) : 1;      // sub-pixel phases on the window kernel
        // cdae_tune_set(CDAE_TUNE_CONVWIN_MIN_TILES, <= 1): every shape convwin_kernel can take runs on it, whatever the grid size (the
        // parity tests push the small golden cases through the kernel the benchmark shapes dispatch)
        if (cdae_tune(TUNE_CONVWIN_MIN_TILES) <= 1 && p.amode == A_CONV_VEC && p.stride == 1 && !p.up && cdae_convwin_ok(p)) big = 1;
        // window-resident form: stride-1 3x3 convs on a dense NHWC tensor, rows up to 64 pixels, row-major result
        const bool win_ok = cfg_win && p.amode == A_CONV_VEC && p.stride == 1 && !p.up && p.W <= 64 && big &&
                            p.sy == (long)p.W * p.sx && p.sn == (long)p.H * p.W * p.sx &&
                            (p.ps_taps == 4 ? p.out_mode == OUT_UP2 && !p.gn_coef && cfg_subpix : p.out_mode == OUT_ROWMAJOR);
        if (p.a_gm && (!win_ok || p.gn_coef)) { cdae_prof_end(PROF_IGEMM, st); return 3; }
        if (p.nphase > 1 && (!win_ok || p.gn_coef)) { cdae_prof_end(PROF_IGEMM, st); return 2; }      // (only convwin_kernel walks the four phases itself)
        if (p.gn_coef && (!win_ok || (p.A2 && p.K1 % BK))) return cdae_fail("fused GroupNorm prologue: only on the window-resident conv path");
        if (win_ok && p.gn_coef) {
            const int nchunk = p.Cin / BK;
            if (p.ksplit > nchunk) p.ksplit = nchunk;
            ks = p.ksplit;
            rc = p.prec == 1 ? launch_pswin<128, 2, 2, 272, 4, true>(p, st) : launch_pswin<128, 1, 2, 272, 4, true>(p, st);
        } else
        if (win_ok) {
            const int nchunk = p.Cin / BK;
            if (p.ksplit > nchunk) p.ksplit = nchunk;              // K is split by whole channel chunks here
            ks = p.ksplit;
            const bool deep = p.W <= 32 && cfg_win == 3;           // CDAE_PS_WIN=3: 3-stage weight ring where it fits (measured: no gain over 2 stages)
            static const int cfg_bm = getenv("CDAE_PS_WIN_BM") ? atoi(getenv("CDAE_PS_WIN_BM")) : 256;
            // 256-row tiles: rows must divide the tile (tight window) and the larger grid must still fill two blocks per CU
            const bool tall2 = cfg_bm == 256 && 256 % p.W == 0 && (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * ks >= 512;
            const bool tall = tall2 && p.prec == 1;
            static const int cfg_wm = getenv("CDAE_PS_WIN_WM") ? atoi(getenv("CDAE_PS_WIN_WM")) : 64;
            // rows of 16 or 8 pixels: the tight window is 160 rows, 52.5 KB with the two weight stages: THREE blocks (24 waves) per CU
            static const int cfg_b3 = getenv("CDAE_PS_WIN_B3") ? atoi(getenv("CDAE_PS_WIN_B3")) : 0;     // measured: +2 % at 16x16, -8 % at 8x8 -> off
            const bool small_rows = cfg_b3 && p.prec == 1 && p.W <= 16 && 128 % p.W == 0 && cfg_win == 1;
            // second-generation window kernel (convwin.hip): 4 waves of 128 x 64, 16x16x32 MFMA, staggered half-window reloads
            static const int cfg_cw = getenv("CDAE_CONVWIN") ? atoi(getenv("CDAE_CONVWIN")) : 1;
            const int cfg_cw_min = cdae_tune(TUNE_CONVWIN_MIN_TILES);
            const int cfg_cw_ks = cdae_tune(TUNE_CONVWIN_SPLITK);
            const long cw_tiles = (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * (p.nphase > 1 ? p.nphase : 1);
            bool cw = cfg_cw && cdae_convwin_ok(p);
            if (cw) {
                // fewer 256 x 128 tiles than block slots (two per CU): split K by whole 32-channel chunks (the low-resolution levels, and
                // everything below 64 x 64 at training batch sizes).  floor, not ceil: 96 tiles x 6 = 576 would need a second, nearly
                // empty round of blocks; x 5 = 480 runs in one
                int kbest = ks;
                if (cw_tiles * ks < 512 && cfg_cw_ks && p.ksplit_auto && p.splitk_ws && !p.gn_part) {
                    int k2 = (int)(512 / cw_tiles);
                    static const int cfg_minchunk = getenv("CDAE_CONVWIN_MINCHUNK") ? atoi(getenv("CDAE_CONVWIN_MINCHUNK")) : 3;
                    if (k2 > nchunk / cfg_minchunk) k2 = nchunk / cfg_minchunk;          // at least three chunks = 27 K-steps per split (1 / 2 / 3 / 4 / unsplit: 30.37 / 30.26 / 30.14 / 30.25 / 30.62 ms per C64 training step)
                    while (k2 > 1 && (size_t)k2 * p.M * p.N * sizeof(float) > p.splitk_ws_bytes) --k2;
                    if (k2 > 1) { const int c_per = (nchunk + k2 - 1) / k2; k2 = (nchunk + c_per - 1) / c_per; }      // 12 chunks over 5 splits are 3 + 3 + 3 + 3 + 0: no empty slabs
                    if (k2 > 1) kbest = k2;
                }
                if (cw_tiles * kbest >= cfg_cw_min) p.ksplit = ks = kbest;
                else cw = false;
            }
            if (p.a_gm && !cw) { cdae_prof_end(PROF_IGEMM, st); return 3; }                      // group-major planes: only convwin_kernel reads them (the caller converts)
            if (p.nphase > 1 && (!cw || ks > 1)) { cdae_prof_end(PROF_IGEMM, st); return 2; }      // fused phases only on the window kernel: the caller launches them one by one
            if (cw) {
                // algorithmic bytes: both activation planes, both weight planes, the fp32 result (+ the residual read)
                const double nph = p.nphase > 1 ? p.nphase : 1;      // (the phases of an up-conv share the input planes)
                cdae_prof_note(p.ps_taps == 4 ? PROF_CONVWIN_UP : p.prec == 2 ? PROF_CONVWIN_DGRAD : PROF_CONVWIN, 4.0 * p.M * p.Cin + nph * (4.0 * p.K * p.N + 4.0 * p.M * p.N * (p.res ? 2 : 1)));
                rc = cdae_convwin_launch(p, st);
            }
            else
            if (p.prec == 2) rc = tall2 ? launch_pswin<128, 2, 2, 384, 2, false, 256, true, 2, true>(p, st) : launch_pswin<128, 2, 2, 272, 4, false, 128, false, 2, true>(p, st);
            else if (tall && cfg_wm == 128) rc = launch_pswin<128, 2, 2, 384, 2, false, 256, true, 2, false, 128>(p, st);      // 4 waves of 128 x 64
            else if (tall) rc = launch_pswin<128, 2, 2, 384, 2, false, 256>(p, st);
            else if (small_rows) rc = launch_pswin<128, 2, 2, 160, 4, false, 128, true, 3>(p, st);
            else
            if (p.prec == 1 && cfg_win == 4) rc = launch_pswin<128, 2, 2, 272, 2>(p, st);       // CDAE_PS_WIN=4: 4 waves of 64x64 per block
            else if (p.prec == 1) rc = deep ? launch_pswin<128, 2, 3, 208>(p, st) : launch_pswin<128, 2, 2, 272>(p, st);
            else rc = deep ? launch_pswin<128, 1, 3, 208>(p, st) : launch_pswin<128, 1, 2, 272>(p, st);
        }
        static const int cfg_loaders = getenv("CDAE_PS_LOADERS") ? atoi(getenv("CDAE_PS_LOADERS")) : 0;
        if (win_ok) {}
        else if (p.prec == 2) rc = big ? launch_ps<128, 128, 2, 4, 2, 2, 0, true>(p, st) : launch_ps<64, 64, 2, 2, 2, 2, 0, true>(p, st);
        else
        if (p.prec == 1 && big && !huge && cfg_loaders == 4) rc = launch_ps<128, 128, 2, 4, 2, 2, 4>(p, st);
        else if (p.prec == 1 && big && !huge && cfg_loaders == 2) rc = launch_ps<128, 128, 2, 4, 2, 2, 2>(p, st);
        else if (p.prec == 1) rc = huge ? launch_ps<256, 128, 4, 2, 2, 3>(p, st) : big ? launch_ps<128, 128, 2, 4, 2, 2>(p, st) : launch_ps<64, 64, 2, 2, 2, 2>(p, st);
        else rc = huge ? launch_ps<256, 128, 4, 2, 1, 3>(p, st) : big ? launch_ps<128, 128, 2, 4, 1, 2>(p, st) : launch_ps<64, 64, 2, 2, 1, 2>(p, st);
    }
    else CASE(A_PLAIN_KC, B_PLAIN_KC);
    else CASE(A_CONV_VEC, B_PLAIN_KC);
    else CASE(A_CONV_GEN, B_PLAIN_KC);
    else CASE(A_PLAIN_KC, B_PLAIN_MC);
    else CASE(A_PLAIN_MC, B_PLAIN_MC);
    else CASE(A_PLAIN_MC, B_CONV_MC);
    else CASE(A_CONV_VEC, B_WDGRAD_MC);
    else CASE(A_CONV_GEN, B_WDGRAD_MC);
    else rc = cdae_fail("unsupported igemm operand mode combination");
#undef CASE
    if (rc == 0 && ks > 1) {
        long total = (long)p.batch * p.M * p.N;
        static const int cfg_red4 = getenv("CDAE_SPLITK_REDUCE4") ? atoi(getenv("CDAE_SPLITK_REDUCE4")) : 1;
        auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
        const bool vec4 = cfg_red4 && p.batch == 1 && p.out_mode == OUT_ROWMAJOR && p.N % 4 == 0 && p.ldc % 4 == 0 && al16(p.C) && al16(p.res) && al16(p.bias) &&
                          al16(p.splitk_ws) && ((long)p.M * p.N) % 4 == 0;
        if (vec4) total >>= 2;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (vec4) hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
        if (hipGetLastError() != hipSuccess) rc = cdae_fail("splitk reduce launch failed");
    }
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
