// wgrad.hip — weight gradient of a stride-1 conv3x3 with the activation WINDOW resident in LDS (gfx950 only).
//
//   dW[co][ky][kx][ci] = sum over pixels (n, y, x) of dy[n, y, x, co] * a[n, y + ky - 1, x + kx - 1, ci]      (zero padding)
//
// the backward the reference gets from autograd through F.conv2d (nn.py:470-480; ResBlock convs unet.py:187-197).  As a GEMM the
// reduction runs over PIXELS, so both operands are "k-major" in memory ([pixel][channel] rows): they go global -> LDS by LDS-DMA
// exactly as they lie, and the MFMA fragments come out of LDS through the hardware transpose read (ds_read_b64_tr_b16).  Both
// operands are bf16 hi/lo planes (x = hi + lo, 16 significand bits, full fp32 range — gradients underflow f16): dy from
// cdae_split_bf16, a from the GroupNorm that produced the conv input (cdae_gn_apply_split_train).  Each product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
//
// Tiling: a block owns 64 output x 64 input channels, ALL 9 taps, and a contiguous range of 64-pixel K steps; a wave owns
// 64 x 16 x 9 taps (144 accumulator registers).  The 9 taps of a step read the same activation pixels shifted by
// (ky-1)*W + (kx-1), so the activations live in a RING of pixel rows in LDS: every pixel row is fetched once per block (not 9
// times) and a step only loads the 64 new rows.  The images of the batch are laid out in a virtual pixel stream with G >= W+1
// zero rows between them (served from a zero line), which makes the vertical padding and the image boundaries ordinary rows;
// the horizontal padding (x-1 at x == 0, x+1 at x == W-1) is a mask on the activation fragments (element 0 / element 7 of a
// lane's 8 pixels, because steps start on multiples of 64 and W divides 64).
// Ring rows come in 16-row DMA blocks; a tap's 32 rows may start anywhere, so slots 0 and 1 are mirrored behind the last slot and
// reads never wrap inside a fragment.  Split-K over the pixel range: each block writes its partial [Cout][9*Cin] slab, reduced in a
// fixed order by wg_reduce_kernel (deterministic), or stores directly when one block covers all pixels.
// The bias gradient (column sums of dy) rides along as extra MFMAs against a ones operand in one wave per K group of the blocks of
// the first input-channel tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WgParams {
    const unsigned short* a_hi; const unsigned short* a_lo;     // [N*HW][Cin] bf16 planes
    const unsigned short* d_hi; const unsigned short* d_lo;     // [N*HW][Cout] bf16 planes
    float* out;              // ksplit == 1: dW (OHWI [Cout][9*Cin]); else slabs [ksplit][Cout][9*Cin]
    float* colsum;           // += column sums of dy, or nullptr
    int N, HW, W, Cin, Cout;
    int steps, steps_per, ksplit, accumulate;
    int period, period_shift; unsigned period_magic;            // HW + G rows per image in the virtual stream
    int U0, RB;              // rows before image 0 (16 * halo blocks); ring size in 16-row blocks
};

// A GROUP of weight gradients as one launch (round 5): at the 8 x 8 / 16 x 16 levels of a batch-32 step one conv has 36-128 (Cout, Cin) tiles for
// 256 CUs, so each launch split its pixel range 2-7 ways, wrote [ksplit][Cout][9 Cin] slabs (37 MB per launch on average) and paid a finish
// launch; the six to ten convs of a level together fill the chip unsplit.  The descriptors travel in the kernel arguments (scalar loads),
// block b belongs to the descriptor d with start[d] <= b < start[d + 1].
constexpr int WG_MAXD = 12;
struct WgGroup { WgParams d[WG_MAXD]; int start[WG_MAXD + 1]; int nd; };

__device__ __attribute__((aligned(16))) unsigned g_zero_wg[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int fdiv(int n, unsigned magic, int shift) {
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

// LDS image: activations [plane 2][channel half 2][(RB + 2) * 16 rows][64 B], dy [stage 2][plane 2][channel half 2][64 rows][64 B].
// 8 waves = two K groups of 4 (pixels 0..31 / 32..63 of every 64-pixel step; two waves per SIMD that cover each other's LDS latency);
// the second group's accumulators are added to the first's through LDS at the end.  A wave owns ALL 64 output channels x 16 input
// channels, and its group's 32 pixels of a step are ONE 32-deep step of v_mfma_f32_16x16x32_bf16 (the shape that holds its clock under
// load, see convwin.hip).  The dy fragments (4 channel sub-tiles x hi / lo) are read once per step and serve all 9 taps; a tap reads
// only its 16-channel activation fragment pair: 2.9 KB of LDS reads per tap and wave (a 32 x 32 wave tile on 32x32x16 MFMAs needs
// 4.4 KB and kept the LDS pipe ~70 % busy beside the MFMAs: 5.02 -> 4.74 ms over the step's 47 wgrads).  The horizontal padding mask
// sits on the tap's activation fragment (masked copies of the 8 dy fragments would not fit the registers).  A fragment spans 32 ring
// rows, so slots 0 AND 1 are mirrored behind the ring.
// NPL = 1 (the `mixed16` torso): one bf16 plane per operand, one MFMA per product; the lo sub-planes are neither staged nor read.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NPL = 2>
__global__ __launch_bounds__(512, 2) void wgwin_kernel(const WgGroup grp) {
    int di = 0;
    while (di + 1 < grp.nd && (int)blockIdx.x >= grp.start[di + 1]) ++di;           // (uniform: scalar compares on kernel arguments)
    const WgParams& p = grp.d[di];
    const unsigned gb0 = (unsigned)grp.start[di], gG = (unsigned)(grp.start[di + 1] - grp.start[di]);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int RROWS = (p.RB + 2) * 16;                 // ring rows incl. the mirrors of slots 0 and 1
    const int A_SUB = RROWS * 64;                      // bytes per (plane, half) sub-plane
    char* const dyb = lds + 4 * A_SUB;                 // dy stages
    constexpr int D_SUB = 64 * 64, D_STAGE = 4 * D_SUB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, k4 = lane >> 4;
    const int wn = wave & 3;                           // wave tile: input channels 16 wn .. + 15, all 64 output channels
    const int kg = wave >> 2;                          // K group: pixels 32 kg .. + 31 of every step
    const int nci = p.Cin >> 6, nco = p.Cout >> 6;
    int b;
    {
        const unsigned G = gG, bb = blockIdx.x - gb0, q = G >> 3, r = G & 7, x = bb & 7;
        b = (int)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bb >> 3));
    }
    const int cit = b % nci; b /= nci;
    const int cot = b % nco; b /= nco;
    const int ks = b;
    const int ci0 = cit * 64, co0 = cot * 64;
    const int s_begin = ks * p.steps_per, s_end = min(p.steps, s_begin + p.steps_per);

    // ---- DMA roles: wave w stages sub-plane (P = w >> 1, half = w & 1); group 0 the activations, group 1 dy
    const int dP = (wave >> 1) & 1, dH = wave & 1;
    const bool dma_a = kg == 0 && (NPL == 2 || dP == 0), dma_d = kg == 1 && (NPL == 2 || dP == 0);
    const unsigned short* const a_src = (dP ? p.a_lo : p.a_hi) + ci0 + dH * 32 + (lane & 3) * 8;
    const unsigned short* const d_src = (dP ? p.d_lo : p.d_hi) + co0 + dH * 32 + (lane & 3) * 8;
    char* const a_dst = lds + (dP * 2 + dH) * A_SUB;
    auto dma = [&](const void* src, char* dst_wave_base) {
        cdae_lds_dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst_wave_base - lds)));
    };
    // U0, HW and the period are multiples of 16, so a 16-row block lies entirely inside one image or entirely in a gap: which one is
    // decided on the scalar unit (the block index is uniform), and a lane only adds the block's offset to its own row pointer
    const unsigned short* const a_lane = a_src + (long)(lane >> 2) * p.Cin;
    const unsigned short* const d_lane = d_src + (long)(lane >> 2) * p.Cout;
    auto issue_a = [&](int blk, int slot) {
        const int v = __builtin_amdgcn_readfirstlane(blk * 16 - p.U0);
        const int vv = v < 0 ? 0 : v;
        const int img = fdiv(vv, p.period_magic, p.period_shift);
        const int q = vv - img * p.period;
        const bool ok = v >= 0 && q < p.HW && img < p.N;
        const long off = ((long)img * p.HW + q) * p.Cin;
        const void* src = ok ? (const void*)(a_lane + off) : (const void*)g_zero_wg;
        dma(src, a_dst + slot * 1024);
        if (slot < 2) dma(src, a_dst + (p.RB + slot) * 1024);
    };
    auto issue_d = [&](int pix0, int stage) {
        char* const dst = dyb + stage * D_STAGE + (dP * 2 + dH) * D_SUB;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
            dma(d_lane + (long)(pix0 + blk * 16) * p.Cout, dst + blk * 1024);
    };

    // ---- fragments: 16 channels x 32 pixels; lane (column l15, k block k4) gets pixels 8 k4 .. + 7 through two transpose reads 4 rows apart
    const int q4 = l15 >> 2, p4 = lane & 3;
    const int lane_off = (8 * k4 + q4) * 64 + 8 * p4;
    auto trread = [&](const char* src) -> u32x2 {
        const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
        return __builtin_bit_cast(u32x2, v);
    };
    auto frag = [&](const char* src) -> u32x4 {
        const u32x2 a = trread(src), c = trread(src + 256);
        u32x4 r; r[0] = a[0]; r[1] = a[1]; r[2] = c[0]; r[3] = c[1];
        return r;
    };
    auto mma = [&](const u32x4& x, const u32x4& y, const f32x4& c) -> f32x4 {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0);
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][c][r] = 0.f;
    f32x4 accb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) accb[c][r] = 0.f;
    const bool do_colsum = p.colsum != nullptr && cit == 0 && wn == 0;
    u32x4 ones; ones[0] = ones[1] = ones[2] = ones[3] = 0x3F803F80u;       // bf16 1.0 pairs

    // horizontal padding: this lane's 8 pixels start at x0 = (32 kg + 8 k4) mod W (steps start on multiples of 64 and W divides 64)
    const int x0 = (32 * kg + 8 * k4) & (p.W - 1);
    const unsigned mL = x0 == 0 ? 0xFFFF0000u : 0xFFFFFFFFu;           // tap kx = 0 reads x - 1: the pixel at x == 0 (element 0) contributes nothing
    const unsigned mR = x0 == p.W - 8 ? 0x0000FFFFu : 0xFFFFFFFFu;     // tap kx = 2 reads x + 1: the pixel at x == W - 1 (element 7)

    if (s_begin < s_end) {
        const int hb = p.U0 >> 4;
        int img = (s_begin * 64) / p.HW, q = s_begin * 64 - img * p.HW;
        int B0 = (img * p.period + q + p.U0) >> 4;
        int slot0 = 0;
        int next_blk = B0 - hb, next_slot = 0;
        auto load_upto = [&](int blk_end) {
            for (; next_blk < blk_end; ++next_blk) {
                if (dma_a) issue_a(next_blk, next_slot);
                next_slot = next_slot + 1 == p.RB ? 0 : next_slot + 1;
            }
        };
        load_upto(B0 + 4 + hb);
        if (dma_d) issue_d(s_begin * 64, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const int kg_s = __builtin_amdgcn_readfirstlane(kg);
        for (int s = s_begin; s < s_end; ++s) {
            const int st = (s - s_begin) & 1;
            int img_n = img, q_n = q + 64;
            if (q_n == p.HW) { q_n = 0; ++img_n; }
            const int B0n = (img_n * p.period + q_n + p.U0) >> 4;
            if (s + 1 < s_end) {
                load_upto(B0n + 4 + hb);
                if (dma_d) issue_d((s + 1) * 64, st ^ 1);
            }
            // ---- compute: this K group's 32 pixels of step s
            const char* const dy_hi = dyb + st * D_STAGE + kg_s * (32 * 64) + lane_off;
            const char* const a_hi = lds + (wn >> 1) * A_SUB + (wn & 1) * 32 + lane_off;
            const int row_own = __builtin_amdgcn_readfirstlane(slot0 * 16 + p.U0 + 32 * kg_s);
            const int ring = p.RB * 16;
            u32x4 dh[4], dl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dh[c] = frag(dy_hi + (c >> 1) * D_SUB + (c & 1) * 32);
                if constexpr (NPL == 2) dl[c] = frag(dy_hi + (2 + (c >> 1)) * D_SUB + (c & 1) * 32);
            }
            if (do_colsum) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { accb[c] = mma(dh[c], ones, accb[c]); if constexpr (NPL == 2) accb[c] = mma(dl[c], ones, accb[c]); }
            }
            auto tap_src = [&](int t) -> const char* {
                int rt = row_own + (t / 3 - 1) * p.W + (t % 3 - 1);
                rt = rt < 0 ? rt + ring : rt;
                rt = rt >= ring ? rt - ring : rt;
                return a_hi + __builtin_amdgcn_readfirstlane(rt * 64);
            };
            u32x4 fh[3], fl[3];                        // fragment sets of taps t, t + 1, t + 2
            fh[0] = frag(tap_src(0)); if constexpr (NPL == 2) fl[0] = frag(tap_src(0) + 2 * A_SUB);
            fh[1] = frag(tap_src(1)); if constexpr (NPL == 2) fl[1] = frag(tap_src(1) + 2 * A_SUB);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kx = t % 3, cur = t % 3;
                if (t + 2 < 9) {
                    const char* src = tap_src(t + 2);
                    fh[(t + 2) % 3] = frag(src); if constexpr (NPL == 2) fl[(t + 2) % 3] = frag(src + 2 * A_SUB);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kx == 0) { fh[cur][0] &= mL; if constexpr (NPL == 2) fl[cur][0] &= mL; }
                if (kx == 2) { fh[cur][3] &= mR; if constexpr (NPL == 2) fl[cur][3] &= mR; }
                if constexpr (NPL == 2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[t][c] = mma(dl[c], fh[cur], acc[t][c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[t][c] = mma(dh[c], fl[cur], acc[t][c]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][c] = mma(dh[c], fh[cur], acc[t][c]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int adv = B0n - B0;
            slot0 += adv; slot0 = slot0 >= p.RB ? slot0 - p.RB : slot0;
            B0 = B0n; img = img_n; q = q_n;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // fold the second K group into the first through LDS (the ring is dead now), three taps at a time: [wave & 3][48 + 16][64] floats
    {
        float* const xch = reinterpret_cast<float*>(lds) + (wave & 3) * (64 * 64) + lane;
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            __syncthreads();
            if (kg == 1) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xch[((t * 4 + c) * 4 + r) * 64] = acc[3 * round + t][c][r];
                if (round == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xch[(48 + c * 4 + r) * 64] = accb[c][r];
                }
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[3 * round + t][c][r] += xch[((t * 4 + c) * 4 + r) * 64];
                if (round == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) accb[c][r] += xch[(48 + c * 4 + r) * 64];
                }
            }
        }
        if (kg == 1) return;
    }

    // ---- epilogue: D[row = co][col = ci] of (tap t, output sub-tile c): row 16 c + 4 k4 + r, column 16 wn + l15
    const long ldo = 9L * p.Cin;
    float* const ob = p.out + (p.ksplit > 1 ? (long)ks * p.Cout * ldo : 0L) + (long)(co0 + 4 * k4) * ldo + ci0 + wn * 16 + l15;
    const bool accum = p.ksplit == 1 && p.accumulate;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* o = ob + (long)(16 * c + r) * ldo + t * p.Cin;
                *o = accum ? *o + acc[t][c][r] : acc[t][c][r];
            }
    if (do_colsum && l15 == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + co0 + 16 * c + 4 * k4 + r, accb[c][r]);
    }
}

// dW (+)= sum over the K-split slabs, in slab order
__global__ void wg_reduce_kernel(const float4* __restrict__ ws, float4* __restrict__ out, long n4, int ksplit, int accumulate) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 s = ws[i];
        int k = 1;
        for (; k + 8 <= ksplit; k += 8) {            // eight slab loads in flight (64 slabs: 8 dependent round trips instead of 16); the additions keep their order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(long)(k + u) * n4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k + 4 <= ksplit; k += 4) {            // four slab loads in flight
            const float4 v0 = ws[(long)k * n4 + i], v1 = ws[(long)(k + 1) * n4 + i], v2 = ws[(long)(k + 2) * n4 + i], v3 = ws[(long)(k + 3) * n4 + i];
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; k < ksplit; ++k) {
            const float4 v = ws[(long)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (accumulate) { const float4 o = out[i]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad of a stride-1 conv3x3 with a HANDFUL of output channels (the UNet's `out` conv, 128 -> 3 or 6 channels, reference
// unet.py:474-478): as an implicit GEMM it is a 6 x 1152 output over 131072 pixels — the MFMA tile kernels run it at a few percent of
// anything (0.49 ms per training step).  Here a thread owns one input channel and sweeps image rows with a sliding 3x3 register
// window of the activation (three new loads per pixel), accumulating dW[co][tap][ci] for all co in registers: plain fp32 FMAs, like
// the reference.  grid (row groups, Cin / 128); per-block partials [Cout][9][Cin] reduced in block order by wg_reduce_kernel.
template <int CO>
__global__ __launch_bounds__(128) void wgrad_fewout_kernel(const float* __restrict__ x, const float* __restrict__ dy, long lddy, float* __restrict__ part,
                                                           float* __restrict__ colsum_part, int N, int H, int W, int Cin, int rows_per_block) {
    const int ci = blockIdx.y * 128 + threadIdx.x;
    const bool live = ci < Cin;
    const int row0 = blockIdx.x * rows_per_block, row1 = min(N * H, row0 + rows_per_block);
    float acc[CO][9];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    float bs[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) bs[c] = 0.f;
    for (int row = row0; row < row1; ++row) {
        const int n = row / H, y = row - n * H;
        const float* xr[3];
        bool ok[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int yy = y + k - 1;
            ok[k] = live && yy >= 0 && yy < H;
            xr[k] = x + ((long)(n * H + (ok[k] ? yy : y)) * W) * Cin + (live ? ci : 0);
        }
        float v[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k][0] = 0.f; v[k][1] = 0.f; v[k][2] = ok[k] ? xr[k][0] : 0.f; }
        const float* d = dy + (long)row * W * lddy;
        for (int xx = 0; xx < W; ++xx) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[k][0] = v[k][1]; v[k][1] = v[k][2];
                v[k][2] = (ok[k] && xx + 1 < W) ? xr[k][(long)(xx + 1) * Cin] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const float g = d[(long)xx * lddy + c];
                bs[c] += g;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(g, v[t / 3][t % 3], acc[c][t]);
            }
        }
    }
    if (live) {
        float* o = part + (long)blockIdx.x * CO * 9 * Cin + ci;
#pragma unroll
        for (int c = 0; c < CO; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) o[(long)(c * 9 + t) * Cin] = acc[c][t];
    }
    if (colsum_part && blockIdx.y == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) colsum_part[(long)blockIdx.x * 8 + c] = c < CO ? bs[c < CO ? c : 0] : 0.f;
    }
}

__global__ __launch_bounds__(256) void fewout_bias_kernel(const float* __restrict__ part, float* __restrict__ db, int nblk, int Cout, int accumulate) {
    __shared__ float sm[256][8];
    float s[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float4 lo = *reinterpret_cast<const float4*>(part + (long)b * 8), hi = *reinterpret_cast<const float4*>(part + (long)b * 8 + 4);
        s[0] += lo.x; s[1] += lo.y; s[2] += lo.z; s[3] += lo.w; s[4] += hi.x; s[5] += hi.y; s[6] += hi.z; s[7] += hi.w;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) sm[threadIdx.x][c] = s[c];
    __syncthreads();
    if (threadIdx.x < Cout) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += sm[k][threadIdx.x];          // fixed order
        db[threadIdx.x] = accumulate ? db[threadIdx.x] + t : t;
    }
}

}  // namespace

extern "C" int cdae_conv3x3_wgrad_fewout(const float* x, const float* dy, long lddy, float* dw, float* dbias, int N, int H, int W, int Cin, int Cout,
                                         int accumulate, float* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (Cout < 1 || Cout > 8 || Cin % 4) return cdae_fail("conv3x3_wgrad_fewout: 1..8 output channels, Cin % 4 == 0");
    int rpb = 2;
    int nblk = (N * H + rpb - 1) / rpb;
    const size_t per = (size_t)Cout * 9 * Cin * sizeof(float);
    while (nblk > 1 && (size_t)nblk * per + (size_t)nblk * 8 * sizeof(float) > ws_bytes) { rpb *= 2; nblk = (N * H + rpb - 1) / rpb; }
    if (!ws || (size_t)nblk * per + (size_t)nblk * 8 * sizeof(float) > ws_bytes) return cdae_fail("conv3x3_wgrad_fewout: workspace too small");
    float* colpart = dbias ? ws + (size_t)nblk * Cout * 9 * Cin : nullptr;
    dim3 grid(nblk, (Cin + 127) / 128);
    cdae_prof_begin(PROF_IGEMM, 2.0 * Cout * 9.0 * Cin * (double)N * H * W, st);
#define FEW(C) hipLaunchKernelGGL(wgrad_fewout_kernel<C>, grid, dim3(128), 0, st, x, dy, lddy, ws, colpart, N, H, W, Cin, rpb)
    switch (Cout) { case 1: FEW(1); break; case 2: FEW(2); break; case 3: FEW(3); break; case 4: FEW(4); break;
                    case 5: FEW(5); break; case 6: FEW(6); break; case 7: FEW(7); break; default: FEW(8); }
#undef FEW
    int rc = hipGetLastError() == hipSuccess ? 0 : cdae_fail("wgrad_fewout launch failed");
    if (rc == 0) {
        const long n4 = (long)Cout * 9 * Cin / 4;
        int blocks = (int)((n4 + 255) / 256);
        hipLaunchKernelGGL(wg_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)ws, (float4*)dw, n4, nblk, accumulate);
        if (dbias) hipLaunchKernelGGL(fewout_bias_kernel, dim3(1), dim3(256), 0, st, colpart, dbias, nblk, Cout, accumulate);
        if (hipGetLastError() != hipSuccess) rc = cdae_fail("wgrad_fewout reduce launch failed");
    }
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}

extern "C" int cdae_conv3x3_wgrad_win_supported(int N, int H, int W, int Cin, int Cout) {
    return N > 0 && W >= 8 && W <= 64 && (W & (W - 1)) == 0 && (H * W) % 64 == 0 && Cin % 64 == 0 && Cout % 64 == 0 &&
           (long)N * H * W * (Cin > Cout ? Cin : Cout) < (1L << 31);
}

// One launch for up to WG_MAXD weight gradients (items[i]: the arguments of cdae_conv3x3_wgrad_win).  The pixel ranges are split so that
// every block of the launch walks about the same number of 64-pixel steps and the blocks come in whole rounds of 256 (one per CU): with
// enough (Cout, Cin) tiles in the group nothing is split at all — no slabs, no finish launch.
extern "C" int cdae_conv3x3_wgrad_win_group(const cdae_wg_item* items, int n, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0) return 0;
    if (!items) return cdae_fail("conv3x3_wgrad_win_group: no items");
    const bool single = cdae_get_default_precision() == CDAE_PREC_MIXED16;      // one bf16 plane per operand in the reduced-precision mode
    for (int i0 = 0; i0 < n; i0 += WG_MAXD) {
        const int nd = n - i0 < WG_MAXD ? n - i0 : WG_MAXD;
        WgGroup g;
        long tiles[WG_MAXD];
        size_t slab[WG_MAXD], smem = 0;
        long tiles_all = 0;
        double flops = 0;
        for (int d = 0; d < nd; ++d) {
            const cdae_wg_item& it = items[i0 + d];
            if (!cdae_conv3x3_wgrad_win_supported(it.N, it.H, it.W, it.Cin, it.Cout))
                return cdae_fail("conv3x3_wgrad_win: needs W a power of two in [8, 64], H*W % 64 == 0, Cin % 64 == 0, Cout % 64 == 0");
            if ((((size_t)it.a_hi | (size_t)it.a_lo | (size_t)it.dy_hi | (size_t)it.dy_lo | (size_t)it.dw) & 15)) return cdae_fail("conv3x3_wgrad_win: 16-byte aligned operands required");
            WgParams& p = g.d[d];
            p.a_hi = it.a_hi; p.a_lo = it.a_lo; p.d_hi = it.dy_hi; p.d_lo = it.dy_lo;
            p.N = it.N; p.HW = it.H * it.W; p.W = it.W; p.Cin = it.Cin; p.Cout = it.Cout;
            p.steps = it.N * p.HW / 64;
            const int hb = it.W / 16 + 1;                      // 16 hb >= W + 1
            const int G = 16 * hb;                             // zero rows between images
            p.U0 = 16 * hb; p.period = p.HW + G;
            p.RB = 8 + G / 16 + 2 * hb + 1;                    // live window (4 + 2 hb) + the largest prefetch (4 + G/16) + 1 spare
            int sh = 0;
            while ((1u << sh) < (unsigned)p.period) ++sh;
            p.period_magic = (unsigned)(((unsigned long long)((1ull << sh) - (unsigned)p.period) << 32) / (unsigned)p.period) + 1u;
            p.period_shift = sh;
            p.accumulate = it.accumulate; p.colsum = it.dbias;
            tiles[d] = (long)(it.Cin / 64) * (it.Cout / 64);
            slab[d] = (size_t)it.Cout * 9 * it.Cin * sizeof(float);
            tiles_all += tiles[d];
            const size_t sm = (size_t)4 * (p.RB + 2) * 16 * 64 + 2 * 4 * 64 * 64;
            if (sm > smem) smem = sm;
            flops += 2.0 * it.Cout * 9.0 * it.Cin * (double)it.N * p.HW;
            if (it.dbias && !it.accumulate && hipMemsetAsync(it.dbias, 0, sizeof(float) * it.Cout, st) != hipSuccess) return cdae_fail("dbias memset failed");
        }
        // S = steps per block.  Candidates: every steps_d / k; cost = rounds of 256 blocks x (S + ~12 steps of prologue / epilogue / fold) + what
        // the splits cost (slab round trip + finish launch, about a dozen steps' worth, once per launch that has any)
        static const int cfg_blocks = CDAE_DEV_INT("CDAE_WG_BLOCKS", 256);      // one block per CU fits (84-134 KB of LDS)
        int bestS = 1 << 30; long best_cost = -1;
        for (int d = 0; d < nd; ++d)
            for (int k = 1; k <= 256; ++k) {
                const int S = (g.d[d].steps + k - 1) / k;
                if (S < 1) break;
                long blocks = 0; bool any = false; size_t need = 0;
                for (int e = 0; e < nd; ++e) {
                    const int ks = (g.d[e].steps + S - 1) / S;
                    blocks += tiles[e] * ks;
                    if (ks > 1) { any = true; need += (size_t)ks * slab[e]; }
                }
                if (need > splitk_ws_bytes || (need && !splitk_ws)) continue;
                const long rounds = (blocks + cfg_blocks - 1) / cfg_blocks;
                const long cost = rounds * (S + 12) + (any ? 12 : 0);
                if (best_cost < 0 || cost < best_cost || (cost == best_cost && S > bestS)) { best_cost = cost; bestS = S; }
                if (blocks > 4 * cfg_blocks) break;
            }
        if (best_cost < 0) return cdae_fail("conv3x3_wgrad_win_group: split-K workspace too small");
        int nblk = 0;
        float* wsp = splitk_ws;
        for (int d = 0; d < nd; ++d) {
            WgParams& p = g.d[d];
            int ks = (p.steps + bestS - 1) / bestS;
            p.steps_per = (p.steps + ks - 1) / ks;
            ks = (p.steps + p.steps_per - 1) / p.steps_per;    // no empty blocks
            p.ksplit = ks;
            p.out = ks > 1 ? wsp : items[i0 + d].dw;
            if (ks > 1) wsp += (size_t)ks * slab[d] / sizeof(float);
            g.start[d] = nblk;
            nblk += (int)(tiles[d] * ks);
        }
        for (int d = nd; d <= WG_MAXD; ++d) g.start[d] = nblk;
        g.nd = nd;
        static size_t attr_bytes = 0;
        if (smem > attr_bytes) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgwin_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgwin_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
            attr_bytes = smem;
        }
        cdae_prof_begin(PROF_IGEMM, flops, st);
        if (cdae_prof_on()) {
            char tag[128];
            const WgParams& p = g.d[0];
            snprintf(tag, sizeof(tag), "wgwin x%d %d->%d @%dx%d n=%d tiles=%ld blocks=%d S=%d ks0=%d planes=%d", nd, p.Cin, p.Cout, p.HW / p.W, p.W, p.N, tiles_all, nblk, bestS,
                     p.ksplit, single ? 1 : 2);
            cdae_prof_tag(tag);
        }
        if (single) hipLaunchKernelGGL((wgwin_kernel<1>), dim3((unsigned)nblk), dim3(512), smem, st, g);
        else hipLaunchKernelGGL((wgwin_kernel<2>), dim3((unsigned)nblk), dim3(512), smem, st, g);
        int rc = hipGetLastError() == hipSuccess ? 0 : cdae_fail("wgwin_kernel launch failed");
        for (int d = 0; d < nd && rc == 0; ++d) {
            const WgParams& p = g.d[d];
            if (p.ksplit <= 1) continue;
            const long n4 = (long)p.Cout * 9 * p.Cin / 4;
            int blocks = (int)((n4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(wg_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)p.out, (float4*)items[i0 + d].dw, n4, p.ksplit, p.accumulate);
            if (hipGetLastError() != hipSuccess) rc = cdae_fail("wg_reduce launch failed");
        }
        cdae_prof_end(PROF_IGEMM, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cdae_conv3x3_wgrad_win(const unsigned short* a_hi, const unsigned short* a_lo, const unsigned short* dy_hi, const unsigned short* dy_lo,
                                      float* dw, float* dbias, int N, int H, int W, int Cin, int Cout, int accumulate, float* splitk_ws,
                                      size_t splitk_ws_bytes, void* stream) {
    cdae_wg_item it;
    it.a_hi = a_hi; it.a_lo = a_lo; it.dy_hi = dy_hi; it.dy_lo = dy_lo; it.dw = dw; it.dbias = dbias;
    it.N = N; it.H = H; it.W = W; it.Cin = Cin; it.Cout = Cout; it.accumulate = accumulate;
    return cdae_conv3x3_wgrad_win_group(&it, 1, splitk_ws, splitk_ws_bytes, stream);
}
