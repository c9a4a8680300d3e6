// wg16.hip — weight gradient of the 16-bit torso's 1 x 1 convolutions / linears (gfx950 only):
//
//     dW[N][K] (+)= sum over rows m of dY[m][N]^T . X[m][K],   dbias[N] (+)= sum over rows of dY        dY, X bf16 rows, dW fp32
//
// (reference improved_diffusion/unet.py:165-171 skip_connection, :218-236 AttentionBlock qkv / proj_out: the .grad of their weights as
// autograd computes it.)  The reduction runs over M = batch x pixels = 10^4..10^5 rows while N, K are 128..768: the kernel is an HBM
// stream of the two operands (2 (N + K) bytes per row) into a few hundred KB of sums.  On the general GEMM (igemm_kernel, both operands
// k-major through register-staged LDS tiles, two stages) these launches ran at 0.25 of the HBM rate plus a finish each.  Here:
//   * a block owns one 128 x 128 tile of dW and a contiguous range of rows; four waves as 2 x 2 of 64 x 64 (64 accumulator VGPRs);
//   * the rows arrive by LDS-DMA exactly as they lie in HBM (32 rows x 128 channels of each operand per step, 16 KB), three steps ahead,
//     in a ring of four slots (64 KB: two blocks per CU), counted vmcnt + one barrier per step — rows16.hip's scheme;
//   * the MFMA fragments (8 consecutive ROWS of one channel per lane) come out of LDS through the hardware transpose read
//     ds_read_b64_tr_b16; the 16-byte pieces of an LDS row are XOR-swizzled with row bits 0, 1 and 3 (applied to the global address of the
//     DMA lane), which spreads the eight rows a half-wave touches per read over all 64 banks;
//   * operands swapped in the MFMA (X fragment in the A slot): a lane's four accumulator registers are four consecutive K of one output
//     channel — the partial tile leaves as 16-byte stores;
//   * dbias: one more MFMA per 16 output channels against a fragment of ones, in the blocks of the first K tile (atomic adds of the
//     per-block column sums, like the general kernel's);
//   * the blocks that share a row range (the tiles of dW) sit on one XCD, so an operand's second read hits that XCD's L2;
//   * the row splits leave dense [split][N][K] partial slabs and the general split-K finish (igemm.hip) adds them in a fixed order.
// Arithmetic: single-plane bf16 products, fp32 accumulation — the mixed16 mode's backward.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "cdae.h"
#include "cdae_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Wg16Params {
    const unsigned short* X; long ldx;
    const unsigned short* DY; long lddy;
    float* slab; float* dbias;
    int M, N, K, tn, tk, splits, rows_per_split;
};

__device__ __attribute__((aligned(16))) unsigned g_zero_wg16[4] = {0u, 0u, 0u, 0u};

constexpr int WG_D = 3, WG_R = 4;                  // steps in flight, ring slots
constexpr int WG_OP_B = 32 * 256;                  // bytes of one operand's 32 x 128 step
constexpr int WG_SLOT_B = 2 * WG_OP_B;

__global__ __launch_bounds__(256, 2) void wg16_kernel(const Wg16Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int T = p.tn * p.tk;
    const int tile = slot0 % T, sl = slot0 / T;
    const int split = xcd + 8 * sl;
    if (split >= p.splits) return;                                       // (block-uniform)
    const int tnb = tile / p.tk, tkb = tile - tnb * p.tk;
    const int n0 = tnb * 128, k0 = tkb * 128;
    const long row0 = (long)split * p.rows_per_split;
    long row1 = row0 + p.rows_per_split;
    if (row1 > p.M) row1 = p.M;
    const int nsteps = row1 > row0 ? (int)((row1 - row0 + 31) >> 5) : 0;
    const unsigned base = (unsigned)(size_t)lds;

    // this wave's share of a step: rows 8 wave .. + 7 of both operands, two DMAs each (4 rows x 16 pieces); LDS piece j of row r <- global
    // piece j ^ f(r), f(r) = 2 ((r & 3) | ((r >> 3 & 1) << 2))
    // (the zero line's address in a scalar register pair: as a plain global it was re-loaded from the GOT twice per step, each load with an
    //  lgkmcnt(0) wait that also drains the fragment reads in flight)
    const void* zero_line = (const void*)g_zero_wg16;
    asm volatile("" : "+s"(zero_line));
    auto fetch = [&](int s, int slot) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 8 * wave + 4 * e + (lane >> 4);
            const long row = row0 + (long)s * 32 + r;
            const bool live = s < nsteps && row < row1;
            const int pc = (lane & 15) ^ (2 * ((r & 3) | (((r >> 3) & 1) << 2)));
            const void* sy = live ? (const void*)(p.DY + row * p.lddy + n0 + pc * 8) : zero_line;
            const void* sx = live ? (const void*)(p.X + row * p.ldx + k0 + pc * 8) : zero_line;
            cdae_lds_dma16(sy, base + slot * WG_SLOT_B + (8 * wave + 4 * e) * 256);
            cdae_lds_dma16(sx, base + slot * WG_SLOT_B + WG_OP_B + (8 * wave + 4 * e) * 256);
        }
    };
#pragma unroll
    for (int d = 0; d < WG_D; ++d) fetch(d, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int wn = wave >> 1, wk = wave & 1;                             // this wave's 64 x 64 quarter of the tile
    const bool do_bias = p.dbias && tkb == 0 && wk == 0;
    f32x4 acc[4][4], bacc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        bacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // transpose read: lane 4 q' + p' of a 16-lane group addresses row 8 q + q' (+ 4), channels c0 + 4 p' .. + 3, and receives channel
    // c0 + (lane & 15) of those four rows
    const int qq = i16 >> 2, pp = i16 & 3;
    const int rr = 8 * q + qq;
    const int fsw = 2 * ((rr & 3) | (((rr >> 3) & 1) << 2));
    auto frag = [&](const unsigned char* img, int c0) -> bf8 {          // 8 rows (8 q .. + 7) of channel c0 + i16
        const int piece = ((c0 >> 3) + (pp >> 1)) ^ fsw;
        const unsigned char* src = img + rr * 256 + piece * 16 + (pp & 1) * 8;
        const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
        const fp16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + 4 * 256));
        const u32x2 a2 = __builtin_bit_cast(u32x2, a), c2 = __builtin_bit_cast(u32x2, c);
        u32x4 r4; r4[0] = a2[0]; r4[1] = a2[1]; r4[2] = c2[0]; r4[3] = c2[1];
        return __builtin_bit_cast(bf8, r4);
    };
    bf8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    int slot = 0;
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WG_D - 1) * 4) : "memory");
        __syncthreads();
        fetch(s + WG_D, slot == 0 ? WG_R - 1 : slot - 1);
        const unsigned char* ys = lds + slot * WG_SLOT_B;
        const unsigned char* xs = ys + WG_OP_B;
        bf8 fy[4], fx[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { fy[a] = frag(ys, wn * 64 + a * 16); fx[a] = frag(xs, wk * 64 + a * 16); }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[b], fy[a], acc[a][b], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int a = 0; a < 4; ++a) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fy[a], bacc[a], 0, 0, 0);
        }
        slot = slot == WG_R - 1 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the ring's trailing fetches land before the block's LDS is released
    // acc[a][b][r]: output channel n0 + 64 wn + 16 a + i16, input channel k0 + 64 wk + 16 b + 4 q + r
    float* out = p.slab + ((long)split * p.N + n0 + wn * 64 + i16) * p.K + k0 + wk * 64 + 4 * q;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *reinterpret_cast<float4*>(out + (long)a * 16 * p.K + b * 16) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    if (do_bias && q == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) atomicAdd(p.dbias + n0 + wn * 64 + a * 16 + i16, bacc[a][0]);
    }
}

}  // namespace

int cdae_wg16_ok(const void* x, long ldx, const void* dy, long lddy, const float* dw, long lddw, int M, int N, int K, int io, size_t ws_bytes) {
    static const int cfg = CDAE_DEV_INT("CDAE_WG16", 1);
    auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
    if (!cfg || (io & 12) != 12 || (long)M < 2L * cdae_tune(TUNE_ROWS16_MIN_M)) return 0;
    if (N % 128 || K % 128 || N > 4096 || K > 4096) return 0;
    if (ldx % 8 || lddy % 8 || lddw % 4 || !al16(x) || !al16(dy) || !al16(dw)) return 0;
    if (ws_bytes < (size_t)8 * N * K * sizeof(float)) return 0;
    return 1;
}

int cdae_wg16(const void* x, long ldx, const void* dy, long lddy, float* dw, long lddw, float* dbias, int M, int N, int K, int accumulate, float* ws,
              size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    static bool attr_done = false;
    constexpr int smem = WG_R * WG_SLOT_B;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wg16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    Wg16Params p;
    p.X = reinterpret_cast<const unsigned short*>(x); p.ldx = ldx;
    p.DY = reinterpret_cast<const unsigned short*>(dy); p.lddy = lddy;
    p.slab = ws; p.dbias = dbias; p.M = M; p.N = N; p.K = K;
    p.tn = N / 128; p.tk = K / 128;
    const int T = p.tn * p.tk;
    // row splits: fill the 512 block slots (two blocks per CU), but keep the slabs (written once, read once by the finish) below half of
    // the operand bytes, a split at least four steps long, and inside the workspace
    // (measured, tools/wg16_slots.py: with at most four tiles one block per CU is enough to stream at the rate two reach — the slabs halve: 256 x 256
    //  over 65 536 rows 31.0 -> 28.5 us, over 16 384 rows 25.3 -> 19.4; 768 x 256 — twelve tiles — needs both slots: 43.5 -> 60.6 us with one)
    const int slots = T <= 4 ? cdae_tune(TUNE_WG16_SLOTS) / 2 : cdae_tune(TUNE_WG16_SLOTS);
    long splits = 8 * (slots / (8 * T) > 0 ? slots / (8 * T) : 1);
    if (splits > 256) splits = 256;            // (one tile: the finish walks every slab per result element — 512 slabs cost it more than the second block per CU gains)
    const double in_bytes = 2.0 * M * (N + K), slab_bytes = 4.0 * N * K;
    while (splits > 8 && (2.0 * splits * slab_bytes > in_bytes || (long)M / splits < 128 || (size_t)(splits * slab_bytes) > ws_bytes)) splits -= 8;
    p.rows_per_split = (int)((((long)M + splits - 1) / splits + 31) / 32 * 32);
    p.splits = (int)(((long)M + p.rows_per_split - 1) / p.rows_per_split);
    const int sl = (p.splits + 7) / 8;
    cdae_prof_begin(PROF_IGEMM, 2.0 * M * N * (double)K, st);
    cdae_prof_note(PROF_IGEMM, in_bytes + slab_bytes);
    if (cdae_prof_on()) {
        char tag[128];
        snprintf(tag, sizeof(tag), "wg16 N=%d K=%d rows=%d splits=%d bias=%d", N, K, M, p.splits, dbias != nullptr);
        cdae_prof_tag(tag);
    }
    if (dbias && !accumulate && hipMemsetAsync(dbias, 0, sizeof(float) * N, st) != hipSuccess) return cdae_fail("dbias memset failed");
    hipLaunchKernelGGL(wg16_kernel, dim3(8 * T * sl), dim3(256), smem, st, p);
    int rc = hipGetLastError() == hipSuccess ? 0 : cdae_fail("wg16 launch failed");
    if (rc == 0) {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.batch = 1; g.batch_inner = 1; g.alpha = 1.f; g.M = N; g.N = K; g.ldc = lddw; g.C = dw; g.accumulate = accumulate;
        g.ksplit = p.splits; g.splitk_ws = ws; g.splitk_ws_bytes = ws_bytes; g.out_mode = OUT_ROWMAJOR; g.act = ACT_NONE;
        g.range_flag = cdae_range_flag_ptr();
        rc = cdae_splitk_finish(g, false, st);
    }
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
