// skipgn_kernel: the entry sweep of a ResBlock whose channel count changes (reference unet.py:165-171 skip_connection = conv 1x1,
// unet.py:143-147 in_layers = GroupNorm32 -> SiLU -> conv3x3) in ONE pass over the block input x = [x1 | x2] (fp32 NHWC rows,
// the decoder's skip concatenation read in place):
//     y      = x @ W^T + bias                       f16x3 split-precision GEMM, W as pre-split f16 hi/lo planes [Cout][C]
//     planes = split_f16(silu?(x * a + b))          the block's first GroupNorm folded to per-(image, channel) (a, b), written as
//                                                   the hi/lo f16 planes the window conv kernel DMAs
// The pass is HBM bound (read 4 B, write 4 B of planes per input element, + 4 B per output element; the MFMA work is 0.25-0.75 of
// that time), so the kernel is built around the memory stream rather than the MFMA stream:
//   * one block = 128 rows x 128 output columns, 4 waves (2 x 2 of 64 x 64), two blocks per CU
//   * x goes global -> registers (2 x 2 dwordx4 per thread and 32-channel step, issued one step ahead and left in flight across
//     the MFMA phase), is split raw into the [row][64 B] swizzled LDS image ps_kernel uses (ds_write_b128), and the same registers
//     are normalised and stored as 16-byte pieces of the planes (by ONE of the row tile's n-tile blocks: the steps are dealt out round robin)
//   * the weight tile of the step arrives by LDS-DMA from the pre-split planes (L2 resident), one step ahead
//   * the (a, b) table of the images the row tile touches sits in LDS
//   * one barrier per step; n-tiles of one row tile are neighbours inside an XCD so the second..fourth read of x hits its L2
// Sums: k ascending in 32-deep steps, per 16-deep MFMA the order lo*hi, hi*lo, hi*hi — ps_kernel's order on the same planes.
#include <stdio.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

// SG_NT (compile-time experiment): bit 0 = nt on the streamed x loads, bit 1 = nontemporal plane stores
#ifndef SG_NT
#define SG_NT 0
#endif
// SG_ABL (dev ablations, timing only): 1 no output stores, 2 no plane stores, 4 no MFMAs, 8 no normalisation arithmetic, 16 no weight DMAs after step 0
#ifndef SG_ABL
#define SG_ABL 0
#endif
// SG_DEPTH: how many 32-channel steps ahead the x rows are requested (2 or 3 register sets of 16 VGPRs)
#ifndef SG_DEPTH
#define SG_DEPTH 2
#endif
// SG_BALANCE: 1 = the plane-writing steps of a row tile are dealt out over its n-tile blocks, 0 = n-tile 0 writes all planes,
// 2 = as 1 but with the turn visible to the optimiser: the build in which hipcc copied the in-flight x registers in front of the counted
// wait (tests/test_host_cpu.py::test_entry_sweep_never_copies_registers_of_loads_in_flight keeps it as its known-bad sample — never ship it)
#ifndef SG_BALANCE
#define SG_BALANCE 1
#endif
#if SG_NT & 1
#define SG_NT_LD " nt"
#else
#define SG_NT_LD ""
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct SkipGnParams {
    const float* x1; const float* x2; long ld1, ld2; int K1;          // columns [0, K1) from x1, [K1, K) from x2 (x2 == nullptr: K1 == K)
    const unsigned short* w_hi; const unsigned short* w_lo; long ldw;
    const float* w_scale;                                              // scale record of the weight planes ({2^k, 2^-k}, nullptr: unscaled): y = acc * 2^-k + bias
    const float* bias; float* y; long ldy;
    const float* res; long ldres;                                      // optional residual rows added to y
    unsigned short* c_hi; unsigned short* c_lo;                        // optional: y also as f16 hi / lo planes (row pitch ldy), the next conv's operand
    float* gn_part;                                                    // optional: per (32-row chunk, column) sum / sum of squares of y, [M / 32][N][2] (M % 32 == 0)
    int planes_gm;                                                     // s_hi / s_lo group-major: [K / 16][M][16] (convwin_kernel's contiguous half-windows)
    const float* coef; int silu; unsigned short* s_hi; unsigned short* s_lo; int norm_a;
    int M, N, K, HW, nimg_tab;                                         // HW = rows per image; nimg_tab = images the LDS table holds per block
    int* range_flag;
};

constexpr int SG_BM = 128, SG_BN = 128, SG_BK = 32;
constexpr int SG_A_PLANE = SG_BM * 64, SG_B_PLANE = SG_BN * 64;       // bytes per 16-bit plane of one stage
constexpr int SG_STAGE = 2 * (SG_A_PLANE + SG_B_PLANE);               // 32 KB
constexpr int SG_TILES = 2 * SG_STAGE;                                 // two stages

__device__ __attribute__((aligned(16))) unsigned g_zero_sg[4] = {0u, 0u, 0u, 0u};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// BF: bf16 hi / lo (16 significand bits, fp32's exponent range) — the operand is a GRADIENT (the streaming dgrad GEMM)
template <bool BF = false>
__device__ __forceinline__ void split8(const f32x4& u, const f32x4& v, u16x8& hi, u16x8& lo) {
    float f[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    if constexpr (BF) {
        bf16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = (__bf16)f[i]; l[i] = (__bf16)(f[i] - (float)h[i]); }
        hi = __builtin_bit_cast(u16x8, h); lo = __builtin_bit_cast(u16x8, l);
    } else {
        half8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) { h[i] = (_Float16)f[i]; l[i] = (_Float16)(f[i] - (float)h[i]); }
        hi = __builtin_bit_cast(u16x8, h); lo = __builtin_bit_cast(u16x8, l);
    }
}

__device__ __forceinline__ void store_planes_sg(const SkipGnParams& p, long addr, float v) {      // as igemm.hip store_planes
    asm volatile("" : "+v"(v));
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    p.c_hi[addr] = __builtin_bit_cast(unsigned short, h);
    p.c_lo[addr] = __builtin_bit_cast(unsigned short, l);
}

// NORMA: the GEMM operand itself is the normalised row, silu?(x * a + b) (a GroupNorm in front of a 1x1 conv: the attention block's qkv) —
// every block folds the coefficients into the rows it stages; no plane side output in that mode.
// BF: bf16 planes and MFMAs — x is a gradient (dy) and the weight planes are the bf16 hi / lo planes of W^T: dx = dy @ W as a streaming GEMM
// (no GroupNorm side, no plane outputs in that form).
template <bool NORMA, bool BF = false>
__global__ __launch_bounds__(256, 2) void skipgn_kernel(const SkipGnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const ctab = reinterpret_cast<float*>(lds + SG_TILES);     // [nimg_tab][K][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nmt = (p.M + SG_BM - 1) / SG_BM, nnt = (p.N + SG_BN - 1) / SG_BN;
    int mt, nt;
    {   // blocks b, b + 8, b + 16, ... run on one XCD: consecutive n-tiles of a row tile stay inside it
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned q = G >> 3, r = G & 7, x = b & 7;
        const unsigned v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        nt = v % nnt; mt = v / nnt;
    }
    (void)nmt;
    const int m0 = mt * SG_BM, n0 = nt * SG_BN;
    // Who writes the normalised planes of a row tile: every n-tile block holds all of x, so the 32-channel steps are dealt out round
    // robin (step kt belongs to the block with nt == kt % nnt) — the SiLU + split arithmetic and the plane stores, the larger part of
    // a step's vector work, are spread over the row tile's blocks instead of making the nt == 0 block the slow one (SG_BALANCE=0:
    // that block writes everything).  The values do not depend on who computes them.
    const bool planes_out = p.s_hi != nullptr;                         // (no planes wanted: the plain streaming GEMM)
    auto writes_planes_at = [&](int kt) -> bool {
        int turn = SG_BALANCE ? kt % nnt : 0;
#if SG_BALANCE != 2
        asm volatile("" : "+s"(turn));          // opaque: nothing about the step loop may be specialised on whose turn it is
#endif
        return planes_out && turn == nt;
    };

    // ---- this thread's slice of the x tile: rows r and r + 64, channels 8 c8 .. 8 c8 + 7 of every 32-deep step
    const int r0 = tid >> 2, c8 = tid & 3;
    const int img0 = m0 / p.HW;
    bool rok[2]; long xoff1[2], xoff2[2], soff[2]; int ioff[2], aoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = r0 + 64 * q, m = m0 + row;
        rok[q] = m < p.M;
        const int mm = rok[q] ? m : 0;
        xoff1[q] = (long)mm * p.ld1 + c8 * 8;
        xoff2[q] = (long)mm * p.ld2 + c8 * 8 - p.K1;
        soff[q] = p.planes_gm ? (long)mm * 16 + (c8 & 1) * 8 : (long)mm * p.K + c8 * 8;      // (+ k: the step's channel offset, see the stores)
        ioff[q] = (mm / p.HW - img0) * p.K * 2 + c8 * 16;             // float index of (image, channel 8 c8) in the table
        aoff[q] = row * 64 + 16 * (c8 ^ ((row >> 2) & 3));
    }
    const bool one_image = ioff[0] == ioff[1];

    // ---- weight tile: two 16-byte pieces per plane per thread (rows t >> 2 and + 64 of the n-tile), by LDS-DMA
    const unsigned short* const zero = reinterpret_cast<const unsigned short*>(g_zero_sg);
    long boff[2]; bool bok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (tid >> 2) + 64 * q, c = (tid & 3) ^ ((row >> 2) & 3);
        bok[q] = n0 + row < p.N;
        boff[q] = (long)(bok[q] ? n0 + row : 0) * p.ldw + c * 8;
    }
    auto issue_b = [&](int stage, int k) {
        const unsigned sb = stage * SG_STAGE + 2 * SG_A_PLANE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (q * 256 + wave * 64) * 16));
            cdae_lds_dma16(bok[q] ? p.w_hi + boff[q] + k : zero, dst);
            cdae_lds_dma16(bok[q] ? p.w_lo + boff[q] + k : zero, dst + SG_B_PLANE);
        }
    };
    // rows beyond M read row 0 (finite data, results never stored).  The x loads are inline assembly: hipcc's waitcnt pass merges the
    // "maybe pending" states of a software-pipelined loop with conditional issues into vmcnt(0) at the first use, which would drain
    // the prefetch of the step after next and the plane stores at the top of every step.  All waiting on them is the counted
    // s_waitcnt at the top of a step, which names the registers so that no use can move above it.
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 c00, c01, c10, c11, n00, n01, n10, n11;                         // [row pass][half]: x of an even / an odd step
#if SG_DEPTH == 3
    f4 m00, m01, m10, m11;                                             // third set: x is requested three steps ahead
#endif
#define SG_LOAD_X(A0, A1, B0, B1, KK)                                                                                      \
    {                                                                                                                      \
        const bool first = (KK) < p.K1; /* block-uniform: a step never straddles the two sources */                        \
        const float* s0 = first ? p.x1 + xoff1[0] + (KK) : p.x2 + xoff2[0] + (KK);                                         \
        const float* s1 = first ? p.x1 + xoff1[1] + (KK) : p.x2 + xoff2[1] + (KK);                                         \
        asm volatile("global_load_dwordx4 %0, %4, off" SG_NT_LD "\n\tglobal_load_dwordx4 %1, %4, off offset:16" SG_NT_LD "\n\t"  \
                     "global_load_dwordx4 %2, %5, off" SG_NT_LD "\n\tglobal_load_dwordx4 %3, %5, off offset:16" SG_NT_LD       \
                     : "=&v"(A0), "=&v"(A1), "=&v"(B0), "=&v"(B1) : "v"(s0), "v"(s1) : "memory");                          \
    }
// one statement for every count (different statements in an if / else chain would meet in phi nodes, i.e. register copies)
#define SG_WAIT_X(CNT, A0, A1, B0, B1)                                                                                     \
    asm volatile("s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(0)\n\ts_branch 3f\n"                           \
                 "1:\n\ts_cmp_lg_u32 %4, 4\n\ts_cbranch_scc1 2f\n\ts_waitcnt vmcnt(4)\n\ts_branch 3f\n"                      \
                 "2:\n\ts_cmp_lg_u32 %4, 8\n\ts_cbranch_scc1 4f\n\ts_waitcnt vmcnt(8)\n\ts_branch 3f\n"                      \
                 "4:\n\ts_waitcnt vmcnt(12)\n3:"                                                                          \
                 : "+v"(A0), "+v"(A1), "+v"(B0), "+v"(B1) : "s"(CNT) : "memory", "scc")

    // ---- prologue: coefficient table, the first two x steps, first weight step
    if (p.coef) {
        const int nfl = p.nimg_tab * p.K * 2;                          // floats; K % 32 == 0 so a multiple of 4
        const float* src = p.coef + (long)img0 * p.K * 2;
        const long lim = ((long)p.M / p.HW) * p.K * 2 - (long)img0 * p.K * 2;     // floats that exist behind img0
        for (int t = tid * 4; t < nfl; t += 1024)
            *reinterpret_cast<float4*>(ctab + t) = t < lim ? *reinterpret_cast<const float4*>(src + t) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();                                               // step 0 reads entries other waves wrote, before its own barrier
    }
    const int nk = p.K / SG_BK;
    const bool full_rows = m0 + SG_BM <= p.M;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One 32-deep step on the register set (x0a, x0b | x1a, x1b) = rows r0 and r0 + 64 of step kt, LDS stage st (compile-time).
    // Vector-memory operations of a step, in issue order: [before the barrier: nothing] weight DMAs of step kt + 1 (4), x loads of
    // step kt + 2 (4, into the registers this step just consumed), plane stores of step kt (4, in the block the step belongs to).  The counter is in
    // order, so the wait at the top of step kt + 1 ("weights kt + 1 landed", which also covers x of step kt + 1 from two steps ago)
    // leaves the younger x loads and stores in flight.
    auto step = [&](f4& x0a, f4& x0b, f4& x1a, f4& x1b, const int kt, const int st) {
        const int k = kt * SG_BK;
        char* const sa = lds + st * SG_STAGE;
        const bool more1 = kt + 1 < nk;                                // (at the top of step kt: x loads of step kt + 1 are the younger ones)
        const bool live = kt >= 0;                                     // the two lead-in steps (kt = -2, -1) only issue: x of steps 0 and 1, weights of step 0
        const bool writes_planes = live && writes_planes_at(kt);
        u16x8 nh[2], nl[2];                                            // the normalised rows, split
        if (live) {
        // (a partial row tile skips some plane stores: no fixed count)
        // younger than the weight DMAs of this step (issued in step kt - 1): the x loads that step issued behind them — those of step
        // kt - 1 + SG_DEPTH — and its plane stores
        const int cnt = __builtin_amdgcn_readfirstlane((kt == 0 || !full_rows) ? 0 : (kt - 1 + SG_DEPTH < nk ? 4 : 0) + (writes_planes_at(kt - 1) ? 4 : 0));
        SG_WAIT_X(cnt, x0a, x0b, x1a, x1b);
        if constexpr (!NORMA) {
            u16x8 hi, lo;
            split8<BF>(x0a, x0b, hi, lo);
            *reinterpret_cast<u16x8*>(sa + aoff[0]) = hi;
            *reinterpret_cast<u16x8*>(sa + SG_A_PLANE + aoff[0]) = lo;
            split8<BF>(x1a, x1b, hi, lo);
            *reinterpret_cast<u16x8*>(sa + aoff[1]) = hi;
            *reinterpret_cast<u16x8*>(sa + SG_A_PLANE + aoff[1]) = lo;
        }
        if (NORMA || writes_planes) {
            float4 cf[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 1 && one_image) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cf[1][e] = cf[0][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cf[q][e] = *reinterpret_cast<const float4*>(ctab + ioff[q] + 2 * k + 4 * e);      // a0 b0 a1 b1
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f4 u = q ? x1a : x0a, v = q ? x1b : x0b;
                const float xs[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                float ys[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ys[2 * e] = fmaf(xs[2 * e], cf[q][e].x, cf[q][e].y);
                    ys[2 * e + 1] = fmaf(xs[2 * e + 1], cf[q][e].z, cf[q][e].w);
                }
                if (p.silu && !(SG_ABL & 8)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ys[e] = cdae_silu(ys[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(ys[e]));      // opaque before the split: same rounding sequence as gn_apply_kernel
                split8(f4{ys[0], ys[1], ys[2], ys[3]}, f4{ys[4], ys[5], ys[6], ys[7]}, nh[q], nl[q]);
            }
        }
        if constexpr (NORMA) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *reinterpret_cast<u16x8*>(sa + aoff[q]) = nh[q];
                *reinterpret_cast<u16x8*>(sa + SG_A_PLANE + aoff[q]) = nl[q];
            }
        }
        __syncthreads();                                               // stage st complete; nobody reads stage st ^ 1 any more
        if (more1 && !(SG_ABL & 16)) issue_b(st ^ 1, k + SG_BK);        // (dev ablation 16: the weight tile of step 0 serves every step — no weight traffic)
        } else if (kt == -1) issue_b(0, 0);
        // (distance 3: at the top of step 1 the loads of step 2 AND 3 are younger than its weights; the lead-in issues x 0, 1, 2 then)
        // ONE load statement per register set for lead-in and steady state alike: a second definition would meet this one in a phi node,
        // i.e. possibly in v_mov copies of registers whose data has not landed (seen when the allocation changed: silent corruption)
        if (kt + SG_DEPTH < nk) SG_LOAD_X(x0a, x0b, x1a, x1b, k + SG_DEPTH * SG_BK);
        if (!live) return;
        if (writes_planes && !(SG_ABL & 2)) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (rok[q]) {
                    // group-major: channels k + 8 c8 .. + 7 are half of group (k + 8 c8) / 16, at [group][row][8 (c8 & 1)]
                    const long so = soff[q] + (p.planes_gm ? (long)((k >> 4) + (c8 >> 1)) * p.M * 16 : (long)k);
#if SG_NT & 2
                    __builtin_nontemporal_store(nh[q], reinterpret_cast<u16x8*>(p.s_hi + so));
                    __builtin_nontemporal_store(nl[q], reinterpret_cast<u16x8*>(p.s_lo + so));
#else
                    *reinterpret_cast<u16x8*>(p.s_hi + so) = nh[q];
                    *reinterpret_cast<u16x8*>(p.s_lo + so) = nl[q];
#endif
                }
        }

        const char* ac = sa;
        const char* bc = sa + 2 * SG_A_PLANE;
        auto frag = [&](const char* plane, int row0, int sk) -> u16x8 {
            const int row = row0 + l31;
            return *reinterpret_cast<const u16x8*>(plane + row * 64 + 16 * ((2 * sk + hh) ^ ((row >> 2) & 3)));
        };
        auto mma = [&](const u16x8& x, const u16x8& y, const f32x16& c) -> f32x16 {
            if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
        };
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
            u16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { ah[i] = frag(ac, wm * 64 + i * 32, sk); al[i] = frag(ac + SG_A_PLANE, wm * 64 + i * 32, sk); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { bh[j] = frag(bc, wn * 64 + j * 32, sk); bl[j] = frag(bc + SG_B_PLANE, wn * 64 + j * 32, sk); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SG_ABL & 4) { acc[i][j][0] += __builtin_bit_cast(float, (unsigned)(al[i][0] ^ bh[j][0] ^ ah[i][1] ^ bl[j][1])); continue; }
                    acc[i][j] = mma(al[i], bh[j], acc[i][j]);
                    acc[i][j] = mma(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = mma(ah[i], bh[j], acc[i][j]);
                }
        }
    };
    // every x load is issued by the two statements inside this loop — and the loop must stay ONE copy: when hipcc peels the lead-in
    // iterations, the peeled loads get registers of their own and the values reach the loop through v_mov copies of data still in flight
    // (the start value is made opaque: with a visible -2 the "kt < 0" iterations are peeled)
    int kt_first = -SG_DEPTH;
    asm volatile("" : "+s"(kt_first));
#if SG_DEPTH == 3
#pragma clang loop unroll(disable)
    for (int kt = kt_first; kt < nk; kt += 3) {                        // (stage = step parity: a run-time value here)
        step(c00, c01, c10, c11, kt, kt & 1);
        if (kt + 1 < nk) step(n00, n01, n10, n11, kt + 1, (kt + 1) & 1);
        if (kt + 2 < nk) step(m00, m01, m10, m11, kt + 2, kt & 1);
    }
#else
#pragma clang loop unroll(disable)
    for (int kt = kt_first; kt < nk; kt += 2) {
        step(c00, c01, c10, c11, kt, 0);
        if (kt + 1 < nk) step(n00, n01, n10, n11, kt + 1, 1);
    }
#endif

    // ---- epilogue: y = acc (* 2^-k of scaled weight planes) + bias (row-major fp32)
    const float unsc = p.w_scale ? p.w_scale[1] : 1.f;
    bool bad = false;
    if (m0 + SG_BM <= p.M && n0 + SG_BN <= p.N) {                      // interior tile: straight-line stores (a branch per row makes hipcc wait for each store)
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = p.bias ? p.bias[n0 + wn * 64 + j * 32 + l31] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* dst = p.y + (long)(m0 + wm * 64 + i * 32 + 4 * hh) * p.ldy + n0 + wn * 64 + j * 32 + l31;
                // the sub-tile's 16 residual values are requested together (a test + load + use per element is one exposed round trip each)
                float rv[16];
                if (p.res) {
                    const float* rsrc = p.res + (long)(m0 + wm * 64 + i * 32 + 4 * hh) * p.ldres + n0 + wn * 64 + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = rsrc[(long)((r & 3) + 8 * (r >> 2)) * p.ldres];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = 0.f;
                }
                float vv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    vv[r] = (acc[i][j][r] * unsc + bv[j]) + rv[r];
                    if (!(SG_ABL & 1) || vv[r] == 1.2345e30f) dst[(long)((r & 3) + 8 * (r >> 2)) * p.ldy] = vv[r];
                    bad |= !__builtin_isfinite(vv[r]);
                }
                if (p.c_hi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) store_planes_sg(p, (dst - p.y) + (long)((r & 3) + 8 * (r >> 2)) * p.ldy, vv[r]);
                }
                if (p.gn_part) {        // this wave owns the whole 32 x 32 sub-tile: the two lane halves hold its 16 + 16 rows of a column
                    float gs = 0.f, gq = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { gs += vv[r]; gq += vv[r] * vv[r]; }
                    gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32);
                    if (hh == 0) {
                        float* o = p.gn_part + ((long)((m0 + wm * 64 + i * 32) >> 5) * p.N + n0 + wn * 64 + j * 32 + l31) * 2;
                        o[0] = gs; o[1] = gq;
                    }
                }
            }
        if (bad && p.range_flag) *p.range_flag = 1;
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
            float gs = 0.f, gq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] * unsc + bv;
                if (p.res) v += p.res[(long)row * p.ldres + col];
                p.y[(long)row * p.ldy + col] = v;
                if (p.c_hi) store_planes_sg(p, (long)row * p.ldy + col, v);
                bad |= !__builtin_isfinite(v);
                gs += v; gq += v * v;
            }
            if (p.gn_part) {            // (M % 32 == 0: a chunk is inside the tensor or not at all)
                gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32);
                if (hh == 0 && m0 + wm * 64 + i * 32 < p.M) {
                    float* o = p.gn_part + ((long)((m0 + wm * 64 + i * 32) >> 5) * p.N + col) * 2;
                    o[0] = gs; o[1] = gq;
                }
            }
        }
    if (bad && p.range_flag) *p.range_flag = 1;
}

bool aligned16(const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; }

}  // namespace

// images (coefficient sets) a 128-row tile can touch
static int skipgn_tab_images(int HW) {      // (row tiles start at multiples of 128)
    if (HW >= SG_BM) return HW % SG_BM == 0 ? 1 : 2;
    return SG_BM % HW == 0 ? SG_BM / HW : SG_BM / HW + 2;
}

extern "C" int cdae_skip_gn_ok(int M, int N, int K, int K1, int HW) {
    if (M <= 0 || N <= 0 || K <= 0 || HW <= 0 || M % HW || K % 32 || K1 % 32 || K1 > K) return 0;
    return (size_t)skipgn_tab_images(HW) * K * 8 <= 16384;            // tiles 64 KB + table <= 80 KB: two blocks per CU
}

static int skipgn_launch(SkipGnParams& p, void* stream, bool bf = false) {
    p.range_flag = cdae_range_flag_ptr();
    const size_t smem = SG_TILES + (p.coef ? (size_t)p.nimg_tab * p.K * 8 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&skipgn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SG_TILES + 16384) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&skipgn_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SG_TILES + 16384) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&skipgn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SG_TILES + 16384) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    const long blocks = (long)((p.M + SG_BM - 1) / SG_BM) * ((p.N + SG_BN - 1) / SG_BN);
    hipStream_t st = (hipStream_t)stream;
    cdae_prof_begin(PROF_IGEMM, 2.0 * p.M * p.N * p.K, st);
    cdae_prof_note(PROF_IGEMM, 4.0 * p.M * ((p.s_hi ? 2.0 : 1.0) * p.K + (p.res ? 2.0 : 1.0) * p.N));
    if (cdae_prof_on()) { char tag[96]; snprintf(tag, sizeof(tag), "skipgn M=%d N=%d K=%d norm_a=%d planes=%d res=%d", p.M, p.N, p.K, p.norm_a, p.s_hi ? 1 : 0, p.res ? 1 : 0); cdae_prof_tag(tag); }
    if (bf) hipLaunchKernelGGL((skipgn_kernel<false, true>), dim3((unsigned)blocks), dim3(256), smem, st, p);
    else if (p.norm_a) hipLaunchKernelGGL(skipgn_kernel<true>, dim3((unsigned)blocks), dim3(256), smem, st, p);
    else hipLaunchKernelGGL(skipgn_kernel<false>, dim3((unsigned)blocks), dim3(256), smem, st, p);
    cdae_prof_end(PROF_IGEMM, st);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("skipgn_kernel launch failed");
}

extern "C" int cdae_skip_gn_fwd(const float* x1, long ld1, int K1, const float* x2, long ld2, const unsigned short* w_hi, const unsigned short* w_lo,
                                long ldw, const float* w_scale, const float* bias, float* y, long ldy, const float* coef, int silu, unsigned short* s_hi,
                                unsigned short* s_lo, int planes_gm, int M, int N, int K, int HW, void* stream) {
    if (!x2) K1 = K;
    if (!cdae_skip_gn_ok(M, N, K, K1, HW)) return cdae_fail("skip_gn_fwd: K, K1 % 32 == 0, M a multiple of HW, coefficient table of a row tile <= 16 KB required");
    if (ld1 % 4 || (x2 && ld2 % 4) || ldw % 8 || !aligned16(x1) || !aligned16(x2) || !aligned16(w_hi) || !aligned16(w_lo) || !aligned16(coef) ||
        !aligned16(s_hi) || !aligned16(s_lo) || !x1 || !w_hi || !w_lo || !coef || !s_hi || !s_lo || !y)
        return cdae_fail("skip_gn_fwd: 16-byte aligned rows, weight planes, coefficients and planes required");
    SkipGnParams p;
    p.x1 = x1; p.x2 = x2; p.ld1 = ld1; p.ld2 = ld2; p.K1 = K1;
    p.w_hi = w_hi; p.w_lo = w_lo; p.ldw = ldw; p.w_scale = w_scale; p.bias = bias; p.y = y; p.ldy = ldy; p.res = nullptr; p.ldres = 0; p.c_hi = nullptr; p.c_lo = nullptr;
    p.coef = coef; p.silu = silu; p.s_hi = s_hi; p.s_lo = s_lo; p.planes_gm = planes_gm; p.norm_a = 0; p.gn_part = nullptr;
    p.M = M; p.N = N; p.K = K; p.HW = HW; p.nimg_tab = skipgn_tab_images(HW);
    return skipgn_launch(p, stream);
}

// The same kernel as a plain streaming GEMM: y = [x1 | x2] @ W^T + bias (+ res), fp32 rows in, pre-split weight planes, f16x3 products
extern "C" int cdae_linear_fwd_stream(const float* x1, long ld1, int K1, const float* x2, long ld2, const unsigned short* w_hi,
                                      const unsigned short* w_lo, long ldw, const float* w_scale, const float* bias, const float* res, long ldres, float* y, long ldy,
                                      unsigned short* c_hi, unsigned short* c_lo, int M, int N, int K, void* stream) {
    return cdae_linear_fwd_stream_gn_part(x1, ld1, K1, x2, ld2, w_hi, w_lo, ldw, w_scale, bias, res, ldres, y, ldy, c_hi, c_lo, nullptr, M, N, K, stream);
}

// + gn_part (may be NULL; M % 32 == 0): [M / 32][N][2] per (32-row chunk, column) sum and sum of squares of y — the statistics of a GroupNorm
// that reads y (the attention block's proj_out + residual feeds the next ResBlock's first norm) without a pass over the tensor
extern "C" int cdae_linear_fwd_stream_gn_part(const float* x1, long ld1, int K1, const float* x2, long ld2, const unsigned short* w_hi,
                                              const unsigned short* w_lo, long ldw, const float* w_scale, const float* bias, const float* res, long ldres,
                                              float* y, long ldy, unsigned short* c_hi, unsigned short* c_lo, float* gn_part, int M, int N, int K, void* stream) {
    if (!x2) K1 = K;
    if (gn_part && M % 32) return cdae_fail("linear_fwd_stream: partial sums need M % 32 == 0");
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 || K1 % 32 || K1 > K) return cdae_fail("linear_fwd_stream: K (and K1) % 32 == 0 required");
    if (ld1 % 4 || (x2 && ld2 % 4) || ldw % 8 || !aligned16(x1) || !aligned16(x2) || !aligned16(w_hi) || !aligned16(w_lo) || !x1 || !w_hi || !w_lo || !y)
        return cdae_fail("linear_fwd_stream: 16-byte aligned rows and weight planes required");
    SkipGnParams p;
    p.x1 = x1; p.x2 = x2; p.ld1 = ld1; p.ld2 = ld2; p.K1 = K1;
    p.w_hi = w_hi; p.w_lo = w_lo; p.ldw = ldw; p.w_scale = w_scale; p.bias = bias; p.y = y; p.ldy = ldy; p.res = res; p.ldres = ldres; p.c_hi = c_hi; p.c_lo = c_lo;
    p.coef = nullptr; p.silu = 0; p.s_hi = nullptr; p.s_lo = nullptr; p.planes_gm = 0; p.norm_a = 0; p.gn_part = gn_part;
    p.M = M; p.N = N; p.K = K; p.HW = M; p.nimg_tab = 0;
    return skipgn_launch(p, stream);
}

// y = silu?(GroupNorm(x)) @ W^T + bias with the GroupNorm folded to per-(image, channel) (a, b) (cdae_gn_coef) and applied to the rows
// as they are staged: GroupNorm -> 1x1 conv (the attention block's norm -> qkv, reference unet.py:213-228) in ONE pass over the fp32
// input, no normalised tensor or planes in HBM.  Shapes as cdae_skip_gn_ok(M, N, K, K, HW).
extern "C" int cdae_linear_fwd_stream_gn(const float* x, long ldx, const unsigned short* w_hi, const unsigned short* w_lo, long ldw, const float* w_scale, const float* bias,
                                         float* y, long ldy, const float* coef, int silu, int M, int N, int K, int HW, void* stream) {
    if (!cdae_skip_gn_ok(M, N, K, K, HW)) return cdae_fail("linear_fwd_stream_gn: K % 32 == 0, M a multiple of HW, coefficient table of a row tile <= 16 KB required");
    if (ldx % 4 || ldw % 8 || !aligned16(x) || !aligned16(w_hi) || !aligned16(w_lo) || !aligned16(coef) || !x || !w_hi || !w_lo || !coef || !y)
        return cdae_fail("linear_fwd_stream_gn: 16-byte aligned rows, weight planes and coefficients required");
    SkipGnParams p;
    p.x1 = x; p.x2 = nullptr; p.ld1 = ldx; p.ld2 = 0; p.K1 = K;
    p.w_hi = w_hi; p.w_lo = w_lo; p.ldw = ldw; p.w_scale = w_scale; p.bias = bias; p.y = y; p.ldy = ldy; p.res = nullptr; p.ldres = 0; p.c_hi = nullptr; p.c_lo = nullptr;
    p.coef = coef; p.silu = silu; p.s_hi = nullptr; p.s_lo = nullptr; p.planes_gm = 0; p.norm_a = 1; p.gn_part = nullptr;
    p.M = M; p.N = N; p.K = K; p.HW = HW; p.nimg_tab = skipgn_tab_images(HW);
    return skipgn_launch(p, stream);
}

// dx[M][K] = dy[M][N] @ W[N][K] (the data gradient of a linear layer / 1x1 conv, reference autograd of nn.Linear / conv1x1) as the same
// streaming kernel on bf16: dy fp32 rows in (split to bf16 hi / lo on the way, full fp32 range), wt_hi / wt_lo the bf16 planes of W^T
// ([K][N], pitch ldwt: cdae_wt_planes_bf16), bf16x3 products, fp32 accumulate.  N % 32 == 0.
extern "C" int cdae_linear_dgrad_stream(const float* dy, long lddy, const unsigned short* wt_hi, const unsigned short* wt_lo, long ldwt, float* dx, long lddx,
                                        int M, int N, int K, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 32) return cdae_fail("linear_dgrad_stream: N % 32 == 0 required");
    if (lddy % 4 || ldwt % 8 || !aligned16(dy) || !aligned16(wt_hi) || !aligned16(wt_lo) || !dy || !wt_hi || !wt_lo || !dx)
        return cdae_fail("linear_dgrad_stream: 16-byte aligned rows and weight planes required");
    SkipGnParams p;
    p.x1 = dy; p.x2 = nullptr; p.ld1 = lddy; p.ld2 = 0; p.K1 = N;
    p.w_hi = wt_hi; p.w_lo = wt_lo; p.ldw = ldwt; p.w_scale = nullptr; p.bias = nullptr; p.y = dx; p.ldy = lddx; p.res = nullptr; p.ldres = 0; p.c_hi = nullptr; p.c_lo = nullptr;
    p.coef = nullptr; p.silu = 0; p.s_hi = nullptr; p.s_lo = nullptr; p.planes_gm = 0; p.norm_a = 0; p.gn_part = nullptr;
    p.M = M; p.N = K; p.K = N; p.HW = M; p.nimg_tab = 0;
    return skipgn_launch(p, stream, true);
}

