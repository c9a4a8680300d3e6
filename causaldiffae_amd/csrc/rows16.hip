// rows16.hip — the 1 x 1 convolutions / linears of the 16-bit torso on LARGE row counts (gfx950 only):
//
//     C[M][N] = A[M][K] . B[N][K]^T + bias (+ res),      A, B bf16 rows with K contiguous, C / res bf16 rows,  K in {64 .. 768}
//
// (reference improved_diffusion/unet.py:165-171 skip_connection = conv_nd(dims, channels, out_channels, 1), :218-236 AttentionBlock qkv /
// proj_out as conv_nd(1, ...), and their data gradients with B = W^T.)  With M = batch x pixels in the 10^4..10^5 and K, N of 128..768 these
// are HBM streams — 2 (K + N) bytes per row against 2 K N flops — that the 128 x 128 plane GEMM tile (ps_kernel) runs at 0.3 of the
// HBM rate: every block pays a global -> LDS prologue for both operands and a 2-byte-store epilogue around eight K-steps.  Here:
//   * the weight does not go through LDS at all: a wave keeps the MFMA fragments of ITS output channels over the whole K in registers
//     (96 - 128 VGPRs: 128 channels for K <= 128, 64 for K = 256, 32 for K = 384 / 512, 16 for K = 768) and streams 16-row steps;
//   * the activation rows: a lane's fragment IS 16 contiguous bytes of its row.  rows16_reg_kernel loads them global -> registers one step
//     ahead (K <= 256 with few channel groups); rows16_ring_kernel fetches a step once per block by LDS-DMA, two to five steps ahead
//     (the four waves of a block hold four channel groups of the same rows: without the ring a CU has two 8 KB steps in flight);
//   * v_mfma_f32_16x16x32_bf16 with the operands SWAPPED (weight fragment in the A slot): the accumulator registers of a lane are then
//     four consecutive COLUMNS of one row, and with the weight columns permuted inside each group a lane owns 16 (8, 4) consecutive
//     columns — residual and result move as 16-byte pieces (the ps_kernel epilogue: 2-byte pieces);
//   * the waves of a block take neighbouring channel groups of the same rows or, when the result has fewer than four groups,
//     neighbouring row steps; the blocks of one row range run on one XCD (N = 768: three blocks read a row).
// Arithmetic: single-plane bf16 products, fp32 accumulation — the mixed16 mode's; K order differs from ps_kernel's, nothing else.
// K = 1152 / 1536 (the qkv data gradient of the 384 / 512-channel levels) stay on the plane GEMM: 192 VGPRs of weight at 16 channels per
// wave, one ring step per slot.  (An LDS-PANEL form — weight chunk resident in LDS, rows global -> registers — was the first version:
// 24 us on 65536 x 256 x 256 where the ring form takes 20.5, and 57 us at K = 768 with the panel reloaded per row tile.)
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "cdae.h"
#include "cdae_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct RowsParams {
    const unsigned short* A; long lda;
    const unsigned short* B; long ldb;
    const float* bias; const unsigned short* res; unsigned short* C; long ldc;
    int M, N, K, tiles, panels, lanes, rsub;
};

constexpr int R16_RING_LDS = 76 * 1024;      // ring bytes per block of the LDS-ring form (two blocks per CU with the bias table beside it)

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// The loops below are written WITHOUT data-dependent branches around their memory operations: rows past M are clamped to M - 1 (their
// loads AND their stores: the clamped lanes recompute and rewrite row M - 1's own values), the prefetch past the last step reloads the
// last step.  gfx950 counts loads and stores in one in-order counter (vmcnt); with a conditional store or prefetch in the loop the compiler
// can only wait for "everything", which serialises the next step's row fetch behind this step's epilogue (measured: 31.7 us where the
// branch-free loop takes less than 20).  For the same reason the bias sits in LDS (lgkmcnt) and the residual of a step is requested BEFORE
// the next step's rows.

// one 16-row x CPL-column piece of the epilogue: v[] += bias (LDS) + residual (already in registers), rounded to bf16, 16-byte stores
// (CPL = 4: one 8-byte store, no residual form)
template <int CPL, bool RES>
__device__ __forceinline__ void r16_store(float (&v)[CPL], const float* __restrict__ bias_l, const u16x8 (&rv)[CPL / 8 > 0 ? CPL / 8 : 1], unsigned short* __restrict__ cp) {
#pragma unroll
    for (int j = 0; j < CPL / 4; ++j) {
        const float4 bb = *reinterpret_cast<const float4*>(bias_l + 4 * j);
        v[4 * j] += bb.x; v[4 * j + 1] += bb.y; v[4 * j + 2] += bb.z; v[4 * j + 3] += bb.w;
    }
    if constexpr (CPL == 4) {
        typedef __bf16 bf4s __attribute__((ext_vector_type(4)));
        bf4s o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf4s*>(cp) = o;
    } else {
#pragma unroll
        for (int h = 0; h < CPL / 8; ++h) {
            bf8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[8 * h + e];
                if (RES) x += bf2f(rv[h][e]);
                o[e] = (__bf16)x;
            }
            *reinterpret_cast<bf8*>(cp + 8 * h) = o;
        }
    }
}

// KB = K / 32 fragment blocks, NSW 16-column subtiles per wave (KB x NSW x 4 VGPRs of weight); per 16-row step: KB 16-byte fragment loads
// per lane (the next step's in flight), KB x NSW MFMAs, the 16-byte-piece epilogue.  No LDS traffic in the loop but the bias.
template <int KB, int NSW, bool RES>
__global__ __launch_bounds__(256, 2) void rows16_reg_kernel(const RowsParams p) {
    __shared__ __attribute__((aligned(16))) float bias_s[4 * NSW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = slot % p.panels, tl = slot / p.panels;
    const int wpc = 4 / p.rsub;                                    // column groups per block (4, 2 or 1); rsub row steps per block step
    const int n0 = (cb * wpc + wave % wpc) * NSW * 16;             // (the launch picks wpc so that the groups divide evenly: no idle waves)
    const int step = 8 * p.lanes * p.rsub;
    int rs = (xcd + 8 * tl) * p.rsub + wave / wpc;                 // 16-row step
    for (int i = tid; i < 4 * NSW * 16; i += 256) {
        const int w = i / (NSW * 16), c = (cb * wpc + w % wpc) * NSW * 16 + i % (NSW * 16);
        bias_s[i] = (p.bias && c < p.N) ? p.bias[c] : 0.f;
    }
    __syncthreads();
    if (rs >= p.tiles) return;

    // weight fragments: subtile s = 4 t + j, index i -> column n0 + 64 t + 16 (i >> 2) + 4 j + (i & 3): a lane then owns 16 consecutive
    // result columns per group of four subtiles
    bf8 b[KB][NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        const int col = n0 + 64 * (s >> 2) + 16 * (i16 >> 2) + 4 * (s & 3) + (i16 & 3);
        const unsigned short* src = p.B + (long)col * p.ldb + q * 8;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[kb][s] = *reinterpret_cast<const bf8*>(src + kb * 32);
    }
    auto row_of = [&](int r) { const int row = r * 16 + i16; return row < p.M ? row : p.M - 1; };
    auto load_a = [&](bf8 (&a)[KB], int r) {
        const unsigned short* src = p.A + (long)row_of(r) * p.lda + q * 8;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a[kb] = *reinterpret_cast<const bf8*>(src + kb * 32);
    };
    constexpr int NG = NSW / 4;
    const float* bias_l = bias_s + (wave % wpc) * NSW * 16 + 16 * q;
    const int last = rs + ((p.tiles - 1 - rs) / step) * step;      // this wave's last step
    // one step: request the residual of step r and the rows of the step after it, multiply the rows that are here, finish.  Two row
    // buffers that swap roles (a copy between them at the end of a step would wait for the rows just requested)
    auto do_step = [&](const bf8 (&a_use)[KB], bf8 (&a_load)[KB], int r) {
        const long off = (long)row_of(r) * p.ldc + n0 + 16 * q;
        u16x8 rv[NG][2];
        if (RES) {
#pragma unroll
            for (int t = 0; t < NG; ++t) { rv[t][0] = *reinterpret_cast<const u16x8*>(p.res + off + 64 * t); rv[t][1] = *reinterpret_cast<const u16x8*>(p.res + off + 64 * t + 8); }
        }
        const int r2 = r + step;
        load_a(a_load, r2 <= last ? r2 : last);
        f32x4 acc[NSW];
#pragma unroll
        for (int s = 0; s < NSW; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int s = 0; s < NSW; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kb][s], a_use[kb], acc[s], 0, 0, 0);
        unsigned short* cp = p.C + off;
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[4 * t + j][e];
            r16_store<16, RES>(v, bias_l + 64 * t, rv[t], cp + 64 * t);
        }
    };
    bf8 a0[KB], a1[KB];
    load_a(a0, rs);
    while (true) {
        do_step(a0, a1, rs);
        if (rs + step > last) break;
        rs += step;
        do_step(a1, a0, rs);
        if (rs + step > last) break;
        rs += step;
    }
}

// ------------------------------------------------------------------ K = 256, N % 256 == 0: rows through an LDS ring
// The four waves of a block above hold four channel groups of the SAME rows: what a CU has in flight towards HBM is then 2 blocks x one
// 8 KB step — 1.1 TB/s of row reads on 65536 x 256 x 256 (30.7 us; Little's law at ~3 us of loaded latency), and the registers have
// no room for a second step.  Here the rows of a step are fetched ONCE per block by LDS-DMA (global_load_lds_dwordx4: no VGPRs), D steps
// ahead, into a ring of D + 1 slots; each wave fetches a quarter of a step (and, RES, its own 16 x 64 residual piece), waits for its own
// share with a counted vmcnt, a block barrier makes the step visible, and the fragments are 16-byte LDS reads that each feed four MFMAs.
//   slot layout: row r (512 bytes) holds its 16-byte pieces XOR-swizzled in the low four index bits with r — a fragment read (16 rows,
//   one piece each) then covers the 64 banks once; the DMA lands linearly, so the swizzle is applied to the GLOBAL address of a lane.
//   vmcnt: per step a wave issues 2 (+ 2) DMAs and 2 stores; at the wait of step s the operations younger than DMA(s) are D store pairs
//   and D - 1 DMA groups, all counted statically (rows past the end are clamped like above, the ring keeps fetching the last step).
template <int KB, int NSW, bool RES>
__global__ __launch_bounds__(256, 2) void rows16_ring_kernel(const RowsParams p) {
    // KB = K / 32 (a multiple of 4: the four waves fetch a quarter of a step each), NSW = 4 (64 channels per wave: K = 256), 2 (32
    // channels per wave: K = 384, 512 — the weight fragments stay at 96 / 128 VGPRs) or 1 (16 channels per wave, no residual: K = 768 —
    // the qkv data gradient; a fragment read then feeds ONE MFMA, which makes this form LDS-read-bound near the HBM floor of those shapes)
    constexpr int ROW_B = KB * 64, STEP_B = 16 * ROW_B;                   // bytes of a row, of a step's rows
    constexpr int CW = NSW * 16, RROW_B = CW * 2, RES_B = 16 * RROW_B;    // a wave's channels; bytes of a residual row, of a wave's residual piece
    constexpr int NA = KB / 4, NR = RES ? RES_B / 1024 : 0;               // DMAs per wave and step: rows, residual
    constexpr int STORES = CW / 32 > 0 ? CW / 32 : 1;                     // (16 consecutive columns per lane = two stores; 8 or 4 = one)
    constexpr int D = (R16_RING_LDS / (STEP_B + (RES ? 4 * RES_B : 0))) - 1 > 5 ? 5 : (R16_RING_LDS / (STEP_B + (RES ? 4 * RES_B : 0))) - 1, R = D + 1;
    constexpr int WAITN = D * STORES + (D - 1) * (NA + NR);
    static_assert(KB % 4 == 0 && (NSW == 4 || NSW == 2 || (NSW == 1 && !RES)) && D >= 2, "rows16 ring: unsupported shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* a_ring = lds;
    unsigned char* r_ring = lds + R * STEP_B;                            // [slot][wave][RES_B]
    float* bias_s = reinterpret_cast<float*>(lds + R * STEP_B + (RES ? R * 4 * RES_B : 0));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int cb = slot0 % p.panels, tl = slot0 / p.panels;
    const int n0 = (cb * 4 + wave) * CW;
    const int step = 8 * p.lanes;
    int rs = xcd + 8 * tl;
    if (tid < 4 * CW) bias_s[tid] = p.bias ? p.bias[cb * 4 * CW + tid] : 0.f;
    if (rs >= p.tiles) return;                                            // (block-uniform)
    const int last = rs + ((p.tiles - 1 - rs) / step) * step;

    // weight fragments; subtile s, index i -> column n0 + (CW / 4) (i >> 2) + 4 s + (i & 3): a lane then owns CW / 4 consecutive columns
    bf8 b[KB][NSW];
#pragma unroll
    for (int s = 0; s < NSW; ++s) {
        const int col = n0 + (CW / 4) * (i16 >> 2) + 4 * s + (i16 & 3);
        const unsigned short* src = p.B + (long)col * p.ldb + q * 8;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[kb][s] = *reinterpret_cast<const bf8*>(src + kb * 32);
    }
    const unsigned a_base = (unsigned)(size_t)a_ring, r_base = (unsigned)(size_t)r_ring;        // LDS byte addresses (the low 32 bits of a shared pointer)
    // this wave's share of a step: the 16 KB 16-byte pieces [wave 16 KB, (wave + 1) 16 KB) of the step's linear image (row = piece / 4 KB);
    // the LDS piece <- the global piece with its low four index bits XORed with the row
    // (a fetch past the block's last step keeps the operation count of the loop static — the vmcnt below is a constant — but must not
    //  cost bandwidth: every lane then reads the same 16 bytes of the weight, one hot cache line, into a slot nobody reads again)
    auto fetch = [&](int r, int slot) {
        const bool live = r <= last;
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int g = wave * 16 * KB + 64 * e + lane, rl = g / (4 * KB), pl = g - rl * 4 * KB;
            int row = r * 16 + rl;
            if (row >= p.M) row = p.M - 1;
            const int pc = (pl & ~15) | ((pl ^ rl) & 15);
            cdae_lds_dma16(live ? p.A + (long)row * p.lda + pc * 8 : p.B, a_base + slot * STEP_B + (wave * 16 * KB + 64 * e) * 16);
        }
        if (RES) {
            constexpr int PPR = RROW_B / 16;                              // residual pieces per row: 8 (64 channels) or 4 (32)
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int g = 64 * e + lane, rl = g / PPR, pl = g - rl * PPR;
                int row = r * 16 + rl;
                if (row >= p.M) row = p.M - 1;
                const int pc = pl ^ ((rl >> (PPR == 8 ? 1 : 2)) & (PPR - 1));
                cdae_lds_dma16(live ? p.res + (long)row * p.ldc + n0 + pc * 8 : p.B, r_base + (slot * 4 + wave) * RES_B + e * 1024);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(rs + d * step, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int CPL = CW / 4;                                           // consecutive columns per lane: 16 or 8
    const float* bias_l = bias_s + wave * CW + CPL * q;
    int slot = 0;
    while (true) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        __syncthreads();
        fetch(rs + D * step, slot == 0 ? R - 1 : slot - 1);
        const unsigned char* al = a_ring + slot * STEP_B + i16 * ROW_B;
        f32x4 acc[NSW];
#pragma unroll
        for (int s = 0; s < NSW; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int pl = kb * 4 + q;
            const bf8 av = *reinterpret_cast<const bf8*>(al + (((pl & ~15) | ((pl ^ i16) & 15)) << 4));
#pragma unroll
            for (int s = 0; s < NSW; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kb][s], av, acc[s], 0, 0, 0);
        }
        int row = rs * 16 + i16;
        if (row >= p.M) row = p.M - 1;
        float v[CPL];
#pragma unroll
        for (int j = 0; j < NSW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * j + e] = acc[j][e];
        u16x8 rv[CPL / 8 > 0 ? CPL / 8 : 1];
        if constexpr (RES) {
            constexpr int PPR = RROW_B / 16;
            const unsigned char* rl = r_ring + (slot * 4 + wave) * RES_B + i16 * RROW_B;
            const int sw = (i16 >> (PPR == 8 ? 1 : 2)) & (PPR - 1);
#pragma unroll
            for (int h = 0; h < CPL / 8; ++h) rv[h] = *reinterpret_cast<const u16x8*>(rl + ((((CPL / 8) * q + h) ^ sw) << 4));
        }
        r16_store<CPL, RES>(v, bias_l, rv, p.C + (long)row * p.ldc + n0 + CPL * q);
        if (rs + step > last) break;
        rs += step;
        slot = slot == R - 1 ? 0 : slot + 1;
    }
}

template <int KB, int NSW, bool RES>
int launch_rows16_ring(RowsParams& p, hipStream_t st) {
    constexpr int STEP_B = 16 * KB * 64, RES_B = 16 * NSW * 32;
    constexpr int D0 = (R16_RING_LDS / (STEP_B + (RES ? 4 * RES_B : 0))) - 1, D = D0 > 5 ? 5 : D0, R = D + 1;
    constexpr int smem = R * STEP_B + (RES ? R * 4 * RES_B : 0) + 4 * NSW * 16 * 4;
    static_assert(2 * smem <= 160 * 1024, "two blocks per CU");
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rows16_ring_kernel<KB, NSW, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    p.rsub = 1;
    p.panels = p.N / (4 * NSW * 16);
    p.tiles = (p.M + 15) / 16;
    int lanes = 64 / p.panels;
    if (lanes < 1) lanes = 1;
    const int need = (p.tiles + 7) / 8;
    if (lanes > need) lanes = need;
    p.lanes = lanes;
    hipLaunchKernelGGL((rows16_ring_kernel<KB, NSW, RES>), dim3(8 * p.panels * lanes), dim3(256), smem, st, p);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("rows16 (LDS ring) launch failed");
}

template <int KB, int NSW>
int launch_rows16(RowsParams& p, hipStream_t st) {
    const int ncg = p.N / (NSW * 16);                   // column groups (one per wave)
    const int wpc = ncg % 4 == 0 ? 4 : ncg % 2 == 0 ? 2 : 1;
    p.rsub = 4 / wpc;
    p.panels = ncg / wpc;
    p.tiles = (p.M + 15) / 16;                          // 16-row steps
    int lanes = 64 / p.panels;                          // per XCD: 32 CUs x 2 blocks of 4 waves
    if (lanes < 1) lanes = 1;
    const int need = (p.tiles + 8 * p.rsub - 1) / (8 * p.rsub);
    if (lanes > need) lanes = need;
    p.lanes = lanes;
    if (p.res) hipLaunchKernelGGL((rows16_reg_kernel<KB, NSW, true>), dim3(8 * p.panels * lanes), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((rows16_reg_kernel<KB, NSW, false>), dim3(8 * p.panels * lanes), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("rows16 launch failed");
}

}  // namespace

// 1: this GEMM runs on rows16_reg_kernel (cdae_gemm16_ps asks; M below the threshold stays on the plane GEMM, whose split-K covers
// short row counts, and so does everything the kernel has no form for)
int cdae_rows16_ok(const void* a16, long lda, const void* b16, long ldb, const float* bias, const void* res, const void* c, long ldc, const float* gn_part,
                   int M, int N, int K, int io, int accumulate) {
    auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
    if (gn_part || accumulate || M < cdae_tune(TUNE_ROWS16_MIN_M)) return 0;
    if (!(io & 1) || (res && !(io & 2))) return 0;                              // bf16 result, bf16 residual
    if (!(K == 64 || K == 128 || K == 192 || K == 256 || ((K == 384 || K == 512) && N % 128 == 0) || (K == 768 && !res && N % 64 == 0))) return 0;
    if ((long)M * (lda > ldc ? lda : ldc) >= (1L << 31)) return 0;
    if (N % (K <= 128 ? 128 : 64) || N > 8192) return 0;
    if (lda % 8 || ldb % 8 || ldc % 8 || !al16(a16) || !al16(b16) || !al16(c) || !al16(res) || (reinterpret_cast<size_t>(bias) & 3)) return 0;
    return 1;
}

extern "C" int cdae_rows16_supported(int M, int N, int K, int io, int has_res) {
    alignas(16) static const char dummy[16] = {0};
    return cdae_rows16_ok(dummy, K, dummy, K, nullptr, has_res ? dummy : nullptr, dummy, N, nullptr, M, N, K, io, 0);
}

int cdae_rows16_gemm(const void* a16, long lda, const void* b16, long ldb, const float* bias, const void* res, void* c, long ldc, int M, int N, int K, int io,
                     void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RowsParams p;
    p.A = reinterpret_cast<const unsigned short*>(a16); p.lda = lda;
    p.B = reinterpret_cast<const unsigned short*>(b16); p.ldb = ldb;
    p.bias = bias; p.res = reinterpret_cast<const unsigned short*>(res); p.C = reinterpret_cast<unsigned short*>(c); p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    cdae_prof_begin(PROF_IGEMM, 2.0 * M * N * (double)K, st);
    cdae_prof_note(PROF_IGEMM, 2.0 * M * K + 2.0 * N * K + 2.0 * M * N * (res ? 2 : 1));
    if (cdae_prof_on()) {
        char tag[128];
        snprintf(tag, sizeof(tag), "rows16 M=%d N=%d K=%d bias=%d res=%d", M, N, K, bias != nullptr, res != nullptr);
        cdae_prof_tag(tag);
    }
    int rc;
    const bool ring = cdae_tune(TUNE_ROWS16_RING) != 0;
    if (K == 256 && N % 256 == 0 && ring) rc = p.res ? launch_rows16_ring<8, 4, true>(p, st) : launch_rows16_ring<8, 4, false>(p, st);
    else if (K == 384) rc = p.res ? launch_rows16_ring<12, 2, true>(p, st) : launch_rows16_ring<12, 2, false>(p, st);
    else if (K == 512) rc = p.res ? launch_rows16_ring<16, 2, true>(p, st) : launch_rows16_ring<16, 2, false>(p, st);
    else if (K == 768) rc = launch_rows16_ring<24, 1, false>(p, st);
    else if (K == 256) rc = launch_rows16<8, 4>(p, st);
    else if (K == 192) rc = launch_rows16<6, 4>(p, st);
    else if (K == 128) rc = launch_rows16<4, 8>(p, st);
    else rc = launch_rows16<2, 8>(p, st);
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
