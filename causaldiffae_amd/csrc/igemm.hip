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
#include "gemm_common.h"

namespace {

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
// GROUP: the kernel argument is a GemmGroupArg — several GEMMs of one operand-mode family in one launch (the linear / 1x1 weight
// gradients of a resolution level: each alone has 9 - 48 tiles and split K 8 - 26 ways into slabs + a finish launch; together they
// fill the chip unsplit).  A block finds its member by a scalar scan and assembles that member's GemmParams in registers.
__device__ __forceinline__ const GemmParams& igemm_pick(const GemmParams& a, const GemmParams&) { return a; }
__device__ __forceinline__ const GemmParams& igemm_pick(const GemmGroupArg&, const GemmParams& l) { return l; }

template <int BM, int BN, int AMODE, int BMODE, bool SCALAR, int WAVES_N = 2, int PREC = 0, bool GNS = false, bool DEEP = false, bool GROUP = false>
__global__ __launch_bounds__(128 * WAVES_N, (GNS && DEEP) ? 4 : 1) void igemm_kernel(const std::conditional_t<GROUP, GemmGroupArg, GemmParams> karg) {     // GNS + DEEP: keep four waves per SIMD (128 VGPRs)
    GemmParams pl_;
    unsigned grid_ = gridDim.x, blk_ = blockIdx.x;
    if constexpr (GROUP) {
        int g = 0;
        for (int i = 1; i < karg.n; ++i) g = (int)blockIdx.x >= karg.first[i] ? i : g;
        g = __builtin_amdgcn_readfirstlane(g);
        pl_ = karg.common;
        const GemmGroupItem& it = karg.items[g];
        pl_.A = it.A; pl_.B = it.B; pl_.C = it.C; pl_.colsum_out = it.colsum_out;
        pl_.M = it.M; pl_.N = it.N; pl_.K = it.K; pl_.accumulate = it.accumulate;
        pl_.lda = it.lda; pl_.ldb = it.ldb; pl_.ldc = it.ldc;
        grid_ = (unsigned)(karg.first[g + 1] - karg.first[g]); blk_ = blockIdx.x - (unsigned)karg.first[g];
    }
    const GemmParams& p = igemm_pick(karg, pl_);
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
        const unsigned G = grid_, b = blk_;
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
    // the zero line of the guarded loads, held in a scalar register pair: as a plain global its address was re-loaded from the GOT inside
    // the K loop (s_load_dwordx2 + s_waitcnt lgkmcnt(0), twice per iteration — the wait also drains the LDS fragment reads in flight)
    const float* zp = g_zero16;
    asm volatile("" : "+s"(zp));

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
                if constexpr (!SCALAR) ra[q] = ld4_if(zp, arowp[q] + k0 + ((p.A2 && k0 >= p.K1) ? adelta[q] : 0L), arow_ok[q] && k < p.K);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ld1_if(zp, arowp[q] + k0 + e, arow_ok[q] && k + e < p.K);
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
            for (int q = 0; q < A_V4; ++q) ra[q] = ld4_if(zp, Ag + atap_off[q] + c0, atap_ok[q]);
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
                    v[e] = ld1_if(zp, Ag + off + (long)c * p.sc, ok);
                }
                ra[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {   // A_PLAIN_MC: element (i,k) at Ag + k*lda + i
            if constexpr (PREC == 4 && !SCALAR) {
                if (p.io16 & 4) {          // (block-uniform) the operand is a bf16 tensor: four elements = 8 bytes, kept RAW in ra[q].x/.y until store_tiles
                    const unsigned short* A16 = reinterpret_cast<const unsigned short*>(p.A) + (Ag - p.A);
#pragma unroll
                    for (int q = 0; q < A_V4; ++q) {
                        const int idx = tid + THREADS * q;
                        const int kk = idx / (BM / 4), i4 = idx - kk * (BM / 4);
                        const int k = k0 + kk, i = m0 + i4 * 4;
                        const bool ok = k < p.K && i < p.M;
                        const float2 raw = *reinterpret_cast<const float2*>(ok ? reinterpret_cast<const float*>(A16 + (long)k * p.lda + i) : zp);
                        ra[q] = make_float4(raw.x, raw.y, 0.f, 0.f);
                    }
                    return;
                }
            }
#pragma unroll
            for (int q = 0; q < A_V4; ++q) {
                const int idx = tid + THREADS * q;
                const int kk = idx / (BM / 4), i4 = idx - kk * (BM / 4);
                const int k = k0 + kk, i = m0 + i4 * 4;
                const float* src = Ag + (long)(k < p.K ? k : 0) * p.lda + i;
                if constexpr (!SCALAR) ra[q] = ld4_if(zp, src, k < p.K && i < p.M);      // host guarantees M % 4 == 0
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ld1_if(zp, src + e, k < p.K && i + e < p.M);
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
                if constexpr (!SCALAR) rb[q] = ld4_if(zp, browp[q] + k0, brow_ok[q] && k < p.K);
                else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ld1_if(zp, browp[q] + k0 + e, brow_ok[q] && k + e < p.K);
                    rb[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        } else {
            if constexpr (PREC == 4 && !SCALAR && BMODE == B_PLAIN_MC) {
                if (p.io16 & 8) {          // (block-uniform) bf16 B operand, raw as above
                    const unsigned short* B16 = reinterpret_cast<const unsigned short*>(p.B) + (Bg - p.B);
#pragma unroll
                    for (int q = 0; q < B_V4; ++q) {
                        const int idx = tid + THREADS * q;
                        const int kk = idx / (BN / 4), j4 = idx - kk * (BN / 4);
                        const int k = k0 + kk, j = n0 + j4 * 4;
                        const bool ok = k < p.K && j < p.N;
                        const float2 raw = *reinterpret_cast<const float2*>(ok ? reinterpret_cast<const float*>(B16 + (long)k * p.ldb + j) : zp);
                        rb[q] = make_float4(raw.x, raw.y, 0.f, 0.f);
                    }
                    return;
                }
            }
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
                        v[e] = ld1_if(zp, Bg + off + (long)c * p.sc, ok);
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
                    if constexpr (!SCALAR) rb[q] = ld4_if(zp, src, kok && j < p.N);      // host guarantees N % 4 == 0
                    else {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ld1_if(zp, src + e, kok && j + e < p.N);
                        rb[q] = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
    };

    const float bsc = (PREC != 0 && p.w_scale) ? p.w_scale[0] : 1.f;
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
        const bool a_raw16 = PREC == 4 && A_MC && (p.io16 & 4), b_raw16 = PREC == 4 && BMODE == B_PLAIN_MC && (p.io16 & 8);
        if constexpr (A_MC) {
            if (do_colsum) {
                if (a_raw16) {
#pragma unroll
                    for (int q = 0; q < A_V4; ++q) {
                        const unsigned u0 = __builtin_bit_cast(unsigned, ra[q].x), u1 = __builtin_bit_cast(unsigned, ra[q].y);
                        colsum.x += __builtin_bit_cast(float, u0 << 16); colsum.y += __builtin_bit_cast(float, u0 & 0xffff0000u);
                        colsum.z += __builtin_bit_cast(float, u1 << 16); colsum.w += __builtin_bit_cast(float, u1 & 0xffff0000u);
                    }
                } else {
#pragma unroll
                for (int q = 0; q < A_V4; ++q) { colsum.x += ra[q].x; colsum.y += ra[q].y; colsum.z += ra[q].z; colsum.w += ra[q].w; }
                }
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
                    if (a_raw16) *reinterpret_cast<float2*>(ac + (kk * PAM + i4 * 4) * 2) = make_float2(ra[q].x, ra[q].y);      // already bf16
                    else store_split(ac, A_PLANE, (kk * PAM + i4 * 4) * 2, ra[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < B_V4; ++q) {
                // f16 planes: the weight operand goes in as w * 2^k (p.w_scale, see cdae_internal.h) so that its lo plane stays normal
                const float4 rbs = (!BF && p.w_scale) ? make_float4(rb[q].x * bsc, rb[q].y * bsc, rb[q].z * bsc, rb[q].w * bsc) : rb[q];
                if constexpr (!B_MC) store_split(bc, B_PLANE, kc_off((tid >> 3) + RPP * q, tid & 7), rbs);
                else {
                    const int idx = tid + THREADS * q;
                    const int kk = idx / (BN / 4), j4 = idx - kk * (BN / 4);
                    if (b_raw16) *reinterpret_cast<float2*>(bc + (kk * PBM + j4 * 4) * 2) = make_float2(rb[q].x, rb[q].y);
                    else store_split(bc, B_PLANE, (kk * PBM + j4 * 4) * 2, rbs);
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
    const float alpha_ = p.w_scale ? p.alpha * p.w_scale[1] : p.alpha;      // scaled weight planes / B tile: the exact 2^-k rides on alpha
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
                    for (int r = 0; r < 16; ++r) rv[r] = (PREC == 3 || PREC == 4) ? cdae_load_res(p, Rg, ad[r]) : Rg[ad[r]];
                }
                if (p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cv[r] = (PREC == 3 || PREC == 4) ? cdae_load_c(p, Cg, ad[r]) : Cg[ad[r]];
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
                float v = acc[i][j][r] * alpha_ + bv;
                if (Rg) v += rv[r];
                if (p.act == ACT_SILU) v = cdae_silu(v);
                else if (p.act == ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (p.accumulate) v += cv[r];
                if constexpr (PREC == 3 || PREC == 4) cdae_store_c(p, Cg, addr, v);
                else Cg[addr] = v;
                if (!__builtin_isfinite(v) && p.range_flag) *p.range_flag = 1;
                if (p.C_hi) store_planes(p, addr, v);
            }
        }
}

// split-K finish: C = alpha * sum_s slab[s] + bias (+res) (-> act), deterministic order
__global__ void splitk_reduce_kernel(const GemmParams p) {
    long total = (long)p.batch * p.M * p.N;
    const float alpha_ = p.w_scale ? p.alpha * p.w_scale[1] : p.alpha;      // scaled weight planes / B tile: the exact 2^-k rides on alpha

    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long bz = idx / ((long)p.M * p.N), rem = idx - bz * (long)p.M * p.N;
        int row = rem / p.N, col = rem - (long)row * p.N;
        float s = 0.f;
        const float* slab = p.splitk_ws + bz * (long)p.M * p.N + rem;
        const long kstride = (long)p.batch * p.M * p.N;
        int k = 0;
        for (; k + 8 <= p.ksplit; k += 8) {          // eight slab loads in flight; same order of additions
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(k + u) * kstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
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
        float v = s * alpha_ + (p.bias ? p.bias[col] : 0.f);
        if (p.res) v += cdae_load_res(p, p.res, bo * p.c_bs0 + bi * p.c_bs1 + addr);
        if (p.act == ACT_SILU) v = cdae_silu(v);
        else if (p.act == ACT_LRELU) v = v > 0.f ? v : 0.01f * v;
        if (p.accumulate) v += cdae_load_c(p, Cg, addr);
        cdae_store_c(p, Cg, addr, v);
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
    const float alpha_ = p.w_scale ? p.alpha * p.w_scale[1] : p.alpha;      // scaled weight planes / B tile: the exact 2^-k rides on alpha
    bool bad = false;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / n4), c4 = (int)(idx - (long)row * n4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* slab = ws + idx;
        int k = 0;
        // (many slabs over a small result — wg16.hip's row splits — leave few threads, each with a long chain of loads: eight in flight;
        //  the additions stay in slab order, so the result does not depend on which loop ran)
        for (; k + 8 <= p.ksplit; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(k + u) * kstride4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
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
        float v[4] = {s.x * alpha_ + b.x, s.y * alpha_ + b.y, s.z * alpha_ + b.z, s.w * alpha_ + b.w};
        if (p.res) {
            if (p.io16 & 2) { for (int e = 0; e < 4; ++e) v[e] += cdae_load_res(p, p.res, addr + e); }
            else { const float4 r = *reinterpret_cast<const float4*>(p.res + addr); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        }
        if (p.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = cdae_silu(v[e]);
        } else if (p.act == ACT_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
        }
        if (p.io16 & 1) {
            for (int e = 0; e < 4; ++e) { if (p.accumulate) v[e] += cdae_load_c(p, p.C, addr + e); cdae_store_c(p, p.C, addr + e, v[e]); }
        } else {
        if (p.accumulate) { const float4 c = *reinterpret_cast<const float4*>(p.C + addr); v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; }
        *reinterpret_cast<float4*>(p.C + addr) = make_float4(v[0], v[1], v[2], v[3]);
        }
        bad |= !__builtin_isfinite(v[0] + v[1] + v[2] + v[3]);
        if (p.C_hi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_planes(p, addr + e, v[e]);
        }
    }
    if (bad && p.range_flag) *p.range_flag = 1;
}

// The split-K finish that ALSO leaves the next GroupNorm's partial sums (p.gn_part, [ceil(M / 32)][N][2]: per 32-row chunk and column the
// sum and the sum of squares of the final values) — what the unsplit epilogues of the plane kernels do, for the convs whose K was split
// (the 8 x 8 level at batch 128, everything below 64 x 64 at training batch sizes): the statistics pass over the tensor (gn_partial +
// finalize) goes away.  Row-major fp32 result, N % 4 == 0, no activation / accumulation / plane output.  One block = 32 rows x 128 columns:
// thread (rg = t / 32, c4 = t % 32) finishes rows rg, rg + 8, rg + 16, rg + 24 of one float4 column group — the slab additions in
// splitk_reduce4_kernel's order — and the eight row groups of a column are added in a fixed order through LDS (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(const GemmParams p) {
    __shared__ float sh[2][8][128];
    const float alpha_ = p.w_scale ? p.alpha * p.w_scale[1] : p.alpha;
    const int ncb = (p.N + 127) >> 7;
    const int chunk = blockIdx.x / ncb, cb = blockIdx.x - chunk * ncb;
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = cb * 128 + c4 * 4;
    const long slab = (long)p.M * p.N;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
    bool bad = false;
    if (col < p.N) {
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = chunk * 32 + rg + 8 * i;
            if (row >= p.M) continue;
            const float* src = p.splitk_ws + (long)row * p.N + col;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= p.ksplit; k += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + k * slab), v1 = *reinterpret_cast<const float4*>(src + (k + 1) * slab),
                             v2 = *reinterpret_cast<const float4*>(src + (k + 2) * slab), v3 = *reinterpret_cast<const float4*>(src + (k + 3) * slab);
                s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
                s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
                s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
                s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
            }
            for (; k < p.ksplit; ++k) { const float4 v = *reinterpret_cast<const float4*>(src + k * slab); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            const long addr = (long)row * p.ldc + col;
            float v[4] = {s.x * alpha_ + b.x, s.y * alpha_ + b.y, s.z * alpha_ + b.z, s.w * alpha_ + b.w};
            if (p.res) {
                if (p.io16 & 2) { for (int e = 0; e < 4; ++e) v[e] += cdae_load_res(p, p.res, addr + e); }
                else { const float4 r = *reinterpret_cast<const float4*>(p.res + addr); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
            }
            if (p.io16 & 1) { for (int e = 0; e < 4; ++e) { v[e] = cdae_round_c(p, v[e]); cdae_store_c(p, p.C, addr + e, v[e]); } }
            else *reinterpret_cast<float4*>(p.C + addr) = make_float4(v[0], v[1], v[2], v[3]);
            bad |= !__builtin_isfinite(v[0] + v[1] + v[2] + v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s4[e] += v[e]; q4[e] += v[e] * v[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[0][rg][c4 * 4 + e] = s4[e]; sh[1][rg][c4 * 4 + e] = q4[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = cb * 128 + threadIdx.x;
        if (c < p.N) {
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) { s_ += sh[0][g][threadIdx.x]; q_ += sh[1][g][threadIdx.x]; }
            float* o = p.gn_part + ((long)chunk * p.N + c) * 2;
            o[0] = s_; o[1] = q_;
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
        static const int cfg_tab = CDAE_DEV_INT("CDAE_GNS_TAB", 1);
        const bool two = p.hw >= BM || (p.hw * 2 == BM);
        q.gn_tab = cfg_tab && two && (size_t)16 * p.K <= tab_max && p.K % 4 == 0 && p.ksplit == 1 && p.batch == 1;
        if (q.gn_tab) smem += (size_t)16 * p.K;
    }
    dim3 grid((unsigned)((long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batch * p.ksplit));
    hipLaunchKernelGGL((igemm_kernel<BM, BN, AMODE, BMODE, SCALAR, WAVES_N, PREC, GNS, DEEP>), grid, dim3(128 * WAVES_N), smem, st, q);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("igemm launch failed");
}

static_assert(sizeof(GemmGroupArg) <= 4096, "a group travels as kernel arguments");
template <int BM, int BN, int AMODE, int BMODE, int WAVES_N, int PREC, bool DEEP>
int launch_group(const GemmGroupArg& g, hipStream_t st) {
    constexpr bool A_MC = (AMODE == A_PLAIN_MC);
    constexpr bool B_MC = (BMODE != B_PLAIN_KC);
    constexpr int NPL = (PREC == 1 || PREC == 2) ? 2 : 1;
    constexpr int A_TILE = PREC ? NPL * (A_MC ? BK * (BM + 32) : BM * 32) / 2 : (A_MC ? BK * (BM + 4) : BM * LDK);
    constexpr int B_TILE = PREC ? NPL * (B_MC ? BK * (BN + 32) : BN * 32) / 2 : (B_MC ? BK * (BN + 4) : BN * LDK);
    constexpr size_t tiles = 2 * (A_TILE + B_TILE) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, AMODE, BMODE, false, WAVES_N, PREC, false, DEEP, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)tiles) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    GemmGroupArg q = g;
    int blocks = 0;
    for (int i = 0; i < q.n; ++i) {
        q.first[i] = blocks;
        blocks += ((q.items[i].M + BM - 1) / BM) * ((q.items[i].N + BN - 1) / BN);
    }
    q.first[q.n] = blocks;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, AMODE, BMODE, false, WAVES_N, PREC, false, DEEP, true>), dim3((unsigned)blocks), dim3(128 * WAVES_N), tiles, st, q);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("igemm group launch failed");
}

template <int AMODE, int BMODE>
int launch_tiles(const GemmParams& p, int big, bool scalar, hipStream_t st) {
    if (scalar) return launch<64, 64, AMODE, BMODE, true>(p, st);
    if constexpr (AMODE == A_CONV_GEN) return cdae_fail("A_CONV_GEN is a scalar-only loader");
    else {
        if constexpr (AMODE == A_PLAIN_KC && BMODE == B_PLAIN_KC) {
            static const int cfg_deep = CDAE_DEV_INT("CDAE_IGEMM_DEEP", 15);      // bit 0 plain fwd GEMM, 1 the GroupNorm-carrying skip GEMM, 2 linear dgrad / wgrad, 3 the same two on 64x64 tiles (2048x512x512: 19.7 -> 16.5 us)
            if (p.S_hi) {
                if (p.prec != 1 || !big) return cdae_fail("GroupNorm side output: f16x3 mode and a grid of 128x128 tiles required");
                return (cfg_deep & 2) ? launch<128, 128, AMODE, BMODE, false, 4, 1, true, true>(p, st) : launch<128, 128, AMODE, BMODE, false, 4, 1, true>(p, st);
            }
            if (p.prec == 1 && big && (cfg_deep & 1)) return launch<128, 128, AMODE, BMODE, false, 4, 1, false, true>(p, st);
            if (p.prec == 1 && !big && (cfg_deep & 8)) return launch<64, 64, AMODE, BMODE, false, 2, 1, false, true>(p, st);
        }
        if constexpr ((AMODE == A_PLAIN_KC || AMODE == A_PLAIN_MC) && BMODE == B_PLAIN_MC) {     // linear / 1x1 dgrad and wgrad: the same latency-bound shape
            static const int cfg_deep2 = CDAE_DEV_INT("CDAE_IGEMM_DEEP", 15);
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

// The finish of a K-split launch: p.ksplit dense [batch][M][N] slabs at p.splitk_ws -> C (alpha, bias, residual, activation, accumulate,
// plane output), optionally leaving the next GroupNorm's partial sums.  Also what wg16.hip's weight-gradient kernel ends with.
int cdae_splitk_finish(const GemmParams& p, bool gn_finish_ok, hipStream_t st) {
    int rc = 0;
    long total = (long)p.batch * p.M * p.N;
    static const int cfg_red4 = CDAE_DEV_INT("CDAE_SPLITK_REDUCE4", 1);
    auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
    const bool vec4 = cfg_red4 && p.batch == 1 && p.out_mode == OUT_ROWMAJOR && p.N % 4 == 0 && p.ldc % 4 == 0 && al16(p.C) && al16(p.res) && al16(p.bias) &&
                      al16(p.splitk_ws) && ((long)p.M * p.N) % 4 == 0;
    if (vec4) total >>= 2;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (p.gn_part && !gn_finish_ok) rc = cdae_fail("K was split but this launch cannot leave GroupNorm partial sums from the finish");
    else if (p.gn_part) {
        if (!(al16(p.C) && al16(p.res) && al16(p.bias) && al16(p.splitk_ws))) rc = cdae_fail("split-K finish with GroupNorm sums: 16-byte aligned result, residual, bias and workspace required");
        else hipLaunchKernelGGL(splitk_reduce_gn_kernel, dim3((unsigned)(((p.M + 31) / 32) * ((p.N + 127) / 128))), dim3(256), 0, st, p);
    }
    else if (vec4) hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) rc = cdae_fail("splitk reduce launch failed");
    return rc;
}

// A group of k-major x k-major GEMMs (the weight gradients of linears / 1x1 convs: C[N_out][K_in] (+)= dy^T x over the rows) as one
// unsplit launch.  Taken when the members together fill the chip (>= 1.5 blocks of 128 x 128 per CU, else >= 0.75 blocks of 64 x 64 per
// CU); otherwise the caller launches them one by one, each with its own K split.
int cdae_gemm_group_dispatch(GemmGroupArg& g, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GemmParams& c = g.common;
    if (g.n < 2 || g.n > GEMM_GROUP_MAX || c.amode != A_PLAIN_MC || c.bmode != B_PLAIN_MC || c.batch != 1 || c.presplit) return 1;
    if (c.prec < 0) {
        const int mode = cdae_get_default_precision();
        c.prec = mode == CDAE_PREC_FP32 ? 0 : mode == CDAE_PREC_MIXED16 ? (c.grad_operand ? 4 : 3) : (c.grad_operand ? 2 : 1);
    }
    const bool rows16 = c.prec == 4 && (c.io16 & 12) == 12;       // the 16-bit torso: both operands are bf16 rows (8-byte vector loads)
    if (c.prec != 0 && c.prec != 2 && !rows16) return 1;
    long tiles_big = 0, tiles_small = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < g.n; ++i) {
        const GemmGroupItem& it = g.items[i];
        if (it.M <= 0 || it.N <= 0 || it.K <= 0 || it.M % 4 || it.N % 4 || it.lda % 4 || it.ldb % 4 ||
            (reinterpret_cast<size_t>(it.A) & (rows16 ? 7 : 15)) || (reinterpret_cast<size_t>(it.B) & (rows16 ? 7 : 15))) return 1;      // the vector k-major loaders
        tiles_big += (long)((it.M + 127) / 128) * ((it.N + 127) / 128);
        tiles_small += (long)((it.M + 63) / 64) * ((it.N + 63) / 64);
        flops += 2.0 * it.M * it.N * (double)it.K;
        bytes += (rows16 ? 2.0 : 4.0) * ((double)it.M * it.K + (double)it.N * it.K) + 4.0 * (double)it.M * it.N;
    }
    static const int cfg_small = CDAE_DEV_INT("CDAE_GROUP_SMALL_TILES", 192);
    const bool big = tiles_big >= cdae_tune(TUNE_GROUP_BIG_TILES);
    if (!big && tiles_small < cfg_small) return 1;
    c.ksplit = 1; c.batch_inner = 1; c.a_scalar = c.b_scalar = 0; c.w_scale = nullptr; c.range_flag = cdae_range_flag_ptr();
    static const int cfg_waves8 = CDAE_DEV_INT("CDAE_IGEMM_WAVES8", 1);
    c.waves8 = cfg_waves8;
    cdae_prof_begin(PROF_IGEMM, flops, st);
    cdae_prof_note(PROF_IGEMM, bytes);
    if (cdae_prof_on()) {
        char tag[128];
        snprintf(tag, sizeof(tag), "gemm group x%d a3 b1 %s tiles=%ld prec=%d (first: M=%d N=%d K=%d)", g.n, big ? "128" : "64", big ? tiles_big : tiles_small, c.prec,
                 g.items[0].M, g.items[0].N, g.items[0].K);
        cdae_prof_tag(tag);
    }
    int rc;
    if (c.prec == 4) rc = big ? launch_group<128, 128, A_PLAIN_MC, B_PLAIN_MC, 4, 4, false>(g, st) : launch_group<64, 64, A_PLAIN_MC, B_PLAIN_MC, 2, 4, false>(g, st);
    else if (c.prec == 2) rc = big ? launch_group<128, 128, A_PLAIN_MC, B_PLAIN_MC, 4, 2, true>(g, st) : launch_group<64, 64, A_PLAIN_MC, B_PLAIN_MC, 2, 2, true>(g, st);
    else rc = big ? launch_group<128, 128, A_PLAIN_MC, B_PLAIN_MC, 4, 0, false>(g, st) : launch_group<64, 64, A_PLAIN_MC, B_PLAIN_MC, 2, 0, false>(g, st);
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
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
#if CW_DEV
    if (getenv("CDAE_GEMM_DEV") && !p.presplit) {          // dev sweeps (tools/gemm_sweep.py): tile and split from the environment, per call
        const char* e = getenv("CDAE_TILE_FORCE");
        if (e && atoi(e) == 64) big = 0;
        if (e && atoi(e) == 128 && p.M >= 96 && p.N >= 96) big = 1;
        e = getenv("CDAE_KS_FORCE");
        if (e && atoi(e) > 0 && p.ksplit_auto && p.splitk_ws) p.ksplit_force = atoi(e);
    }
#endif
    static const int cfg_waves8 = CDAE_DEV_INT("CDAE_IGEMM_WAVES8", 1);   // 8-wave 128x128 tiles by default
    p.waves8 = cfg_waves8;
    // precision: fp32 mode -> fp32 MFMA everywhere; split mode -> f16x3 for activation x weight GEMMs, bf16x3 when an
    // operand is a gradient (api.hip marks those with grad_operand)
    if (p.prec < 0) {
        const int mode = cdae_get_default_precision();
        p.prec = mode == CDAE_PREC_FP32 ? 0 : mode == CDAE_PREC_MIXED16 ? (p.grad_operand ? 4 : 3) : (p.grad_operand ? 2 : 1);
    }
    if (p.A2 && !p.gn_coef && (p.amode != A_PLAIN_KC || p.a_scalar || p.K1 % BK || p.batch != 1)) return cdae_fail("two-source A: vectorised A_PLAIN_KC only, K1 % 32 == 0");
    p.range_flag = cdae_range_flag_ptr();
    if (p.amode == A_PLAIN_MC && p.M % 4) p.a_scalar = 1;          // vector k-major loaders read 4 rows / columns at once
    if (p.bmode != B_PLAIN_KC && p.N % 4) p.b_scalar = 1;
    const bool scalar = p.a_scalar || p.b_scalar || p.amode == A_CONV_GEN;
    if (scalar) big = 0;
    // the weight-operand scale applies where the weight goes through F16 planes: pre-split planes carry it already (their producer applied
    // it), the fp32-operand kernels apply it to the B tile in their f16 modes (vectorised loaders); everywhere else (fp32 MFMA, bf16 planes,
    // the scalar fp32 loaders) the operand is used as it is and the epilogue must not unscale
    if (!p.presplit && ((p.prec != 1 && p.prec != 3) || scalar)) p.w_scale = nullptr;
    if (p.bmode == B_CONV_MC && !scalar) {           // a vectorised wgrad block must sit inside one tap
        if (p.Cin % 128 != 0) big = 0;
        if (p.Cin % 64 != 0) return cdae_fail("B_CONV_MC needs Cin % 64 == 0");
    }
    const long tiles = big ? tiles_big : tiles_small;
    const int nk = (p.K + BK - 1) / BK;
    int ks = 1;
    static const int cfg_mintiles = CDAE_DEV_INT("CDAE_KS_MINTILES", 256);
    static const int cfg_minnk = CDAE_DEV_INT("CDAE_KS_MINNK", 8);
    if (p.ksplit_auto && p.splitk_ws && tiles < cfg_mintiles && nk >= cfg_minnk) {
        ks = (int)((512 + tiles - 1) / tiles);
        if (ks > nk / 4) ks = nk / 4;
        if (ks > 128) ks = 128;             // (a 128 x 128 weight gradient over 131072 pixels: 64 splits 77 us, 128 splits 60 us)
        // 128 x 128 tiles run two blocks per CU (512 slots): 96 tiles x 6 splits = 576 blocks need a second, nearly empty round where
        // x 5 = 480 do not.  Smallest rounds x (K-steps per split + ~8 steps of prologue / epilogue) + finish; CDAE_KS_ROUNDS=0: plain ceil
        static const int cfg_rounds = CDAE_DEV_INT("CDAE_KS_ROUNDS", 1);
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
    // GroupNorm partial sums: from the epilogue of an unsplit plane kernel, or — K split — from the finish kernel (row-major fp32 result,
    // N % 4 == 0, no activation / accumulation / plane output)
    const bool gn_finish_ok = p.gn_part && p.presplit && p.out_mode == OUT_ROWMAJOR && !p.accumulate && p.act == ACT_NONE && !p.C_hi && p.batch == 1 &&
                              p.N % 4 == 0 && p.ldc % 4 == 0;
    if (p.gn_part && ((ks > 1 && !gn_finish_ok) || !p.presplit || (p.out_mode != OUT_ROWMAJOR && p.out_mode != OUT_UP2) || p.accumulate))
        return cdae_fail("GroupNorm partial sums need a pre-split, row-major, non-accumulating launch (K split: fp32 result with N % 4 == 0 and no activation)");

    cdae_prof_begin(PROF_IGEMM, 2.0 * p.M * p.N * (double)p.K * p.batch * (p.nphase > 1 ? p.nphase : 1), st);
    if (!p.presplit) {
        // algorithmic bytes of an fp32-operand contraction: each operand once (a conv gather: the gathered TENSOR, not its im2col rows),
        // the result once (+ the residual read)
        const bool conv = p.amode == A_CONV_VEC || p.amode == A_CONV_GEN;
        const double a_el = conv ? (double)(p.conv_M / (p.Ho * p.Wo > 0 ? p.Ho * p.Wo : 1)) * p.H * p.W * p.Cin : (double)p.M * p.K;
        cdae_prof_note(PROF_IGEMM, 4.0 * p.batch * (a_el + (double)p.N * p.K + (double)p.M * p.N * (p.res ? 2 : 1)));
    }
    if (cdae_prof_on()) {
        char tag[128];
        snprintf(tag, sizeof(tag), "gemm M=%d N=%d K=%d b=%d a%d b%d %s ks=%d prec=%d ps=%d taps=%d res=%d gn=%d", p.M, p.N, p.K, p.batch, p.amode, p.bmode, big ? "128" : "64", ks, p.prec,
                 p.presplit, p.ps_taps, p.res != nullptr, p.gn_part != nullptr);
        cdae_prof_tag(tag);
    }
    int rc = -1;
#define CASE(AM, BM_) if (p.amode == AM && p.bmode == BM_) rc = launch_tiles<AM, BM_>(p, big, scalar, st)
    if (p.presplit) {
        rc = cdae_planes_dispatch(p, big, ks, st);          // planes.hip: convwin / pswin / ps kernels
        if (rc == 2 || rc == 3) { cdae_prof_end(PROF_IGEMM, st); return rc; }
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
    if (rc == 0 && ks > 1) rc = cdae_splitk_finish(p, gn_finish_ok, st);
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
