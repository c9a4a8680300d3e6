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
// Ring rows come in 16-row DMA blocks; a tap's 32 rows may start anywhere, so slots 0, 1 and 2 are mirrored behind the last slot and
// reads never wrap inside a fragment.  Split-K over the pixel range: each block writes its partial [Cout][9*Cin] slab, reduced in a
// fixed order by wg_reduce_kernel (deterministic), or stores directly when one block covers all pixels.
// The bias gradient (column sums of dy) rides along as extra MFMAs against a ones operand in one wave per K group of the blocks of
// the first input-channel tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

// WG_ABL (dev ablations, timing only — results are wrong): 1 no operand DMAs after the prologue, 2 no MFMAs, 4 no activation fragment reads
// after tap 1, 8 no end-of-step wait / barrier, 32 ONE step per block (what a block costs outside its step loop),
// 512 cycle stamps (s_memtime) around the prefetch code, the compute part and the end-of-step wait + barrier, printed per wave by blocks 0 and 37
#ifndef WG_ABL
#define WG_ABL 0
#endif
// WG_NF (compile-time experiment): activation fragment sets in flight (3 = two taps ahead)
#ifndef WG_NF
#define WG_NF 3
#endif

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
    int D;                   // prefetch distance in 64-pixel steps (1 or 2): dy has D + 1 stages, the ring holds D prefetch groups
    int swz;                 // W >= 16: rows with bit 3 set keep their two 32-byte halves swapped in LDS (conflict-free transpose reads)
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

// LDS image: activations [plane NPL][channel half 2][(RB + 3) * 16 rows][64 B], dy [stage D + 1][plane NPL][channel half 2][64 rows][64 B].
// 8 waves = two K groups of 4 (pixels 0..31 / 32..63 of every 64-pixel step; two waves per SIMD that cover each other's LDS latency);
// the second group's accumulators are added to the first's through LDS at the end.  A wave owns ALL 64 output channels x 16 input
// channels, and its group's 32 pixels of a step are ONE 32-deep step of v_mfma_f32_16x16x32_bf16 (the shape that holds its clock under
// load, see convwin.hip).  The dy fragments (4 channel sub-tiles x hi / lo) are read once per step and serve all 9 taps; a tap reads
// only its 16-channel activation fragment pair: 2.9 KB of LDS reads per tap and wave (a 32 x 32 wave tile on 32x32x16 MFMAs needs
// 4.4 KB and kept the LDS pipe ~70 % busy beside the MFMAs: 5.02 -> 4.74 ms over the step's 47 wgrads).  The horizontal padding mask
// sits on the tap's activation fragment (masked copies of the 8 dy fragments would not fit the registers).  A fragment spans 32 ring
// rows starting up to one row behind the ring's end, so slots 0, 1 AND 2 are mirrored behind the ring.
// NPL = 1 (the `mixed16` torso): one bf16 plane per operand, one MFMA per product; the lo sub-planes do not exist in LDS (the image is
// [NPL][half 2] sub-planes per operand), which is what pays for the deeper ring there.
//
// Round 6 (docs/NOTES.md has the ablation table and the cycle stamps behind each point):
// (0) SCALAR SIDE.  The step loop re-loaded descriptor fields from the kernel-argument segment (s_load + lgkmcnt(0), which also drains the
// wave's LDS reads) — a third of the one-plane kernel.  Fields are pinned in SGPRs at entry, DMA addresses are scalar base + constant lane
// offset with an incremental cursor, the regular four-block group is issued straight-line, the nine tap bases are three scalar wraps.
// (a) PREFETCH DISTANCE.  The operands of step s + 2 are requested at the top of step s (ring deeper by one prefetch group, dy in three
// stages) and the wait at the end of a step is COUNTED: only the youngest group (>= WG_NMIN DMAs per issuing wave) may stay in flight.
// D = 1 (the old schedule) where the LDS does not hold the deeper ring (two planes at W = 64).  Worth 1-3 %: latency was not the limiter.
// (b) BANK CONFLICTS.  ds_read_b64_tr_b16 is banked per 32-lane half over 256 bytes; a half reads rows r .. r + 3 and r + 8 .. r + 11 of a
// 64-byte-pitch sub-plane, 32 bytes each: rows r and r + 8 are 512 bytes apart, i.e. on the same banks (2-way on every fragment read,
// 1.65e9 conflict cycles per config [1] step).  Rows with bit 3 set now keep their two 32-byte halves SWAPPED — done by the DMA's
// per-lane SOURCE pointer (the LDS side of an LDS-DMA is fixed at 16 bytes per lane), so a half-wave's eight 32-byte pieces cover all
// 64 banks for every start row.  The read side pays no per-read arithmetic for it: bit 3 of a lane's row depends only on the tap's
// horizontal shift (the vertical one is a multiple of 16 rows for W >= 16), so each lane keeps six offsets (3 kx x 2 reads).
typedef float f32x4 __attribute__((ext_vector_type(4)));
// CO2 (one plane only): the block owns 128 output channels — the two wave groups split the OUTPUT CHANNELS (64 each) instead of the step's
// pixels, every wave walks both 32-pixel halves of a step.  Per MFMA: half the activation DMAs, half the barriers and cursor work, no fold
// of the second group through LDS at the end; dy comes in four 32-channel sub-planes per stage.
template <int NPL = 2, bool CO2 = false>
__global__ __launch_bounds__(512, 2) void wgwin_kernel(const WgGroup grp) {
    static_assert(!CO2 || NPL == 1, "the 128-channel block tile exists for one operand plane");
    constexpr int NDH = CO2 ? 4 : 2;                   // 32-channel dy sub-planes per plane
    constexpr int NPH = CO2 ? 2 : 1;                   // 32-pixel halves of a step a wave walks
    int di = 0;
    while (di + 1 < grp.nd && (int)blockIdx.x >= grp.start[di + 1]) ++di;           // (uniform: scalar compares on kernel arguments)
    const WgParams& pd = grp.d[di];
    // The descriptor is reached through a run-time index: left alone, hipcc re-loads its fields from the kernel-argument segment INSIDE the
    // step loop (s_load_dword + s_waitcnt lgkmcnt(0) — which also drains the wave's LDS fragment reads; a dozen dependent round trips per
    // step in the DMA-issuing waves: a third of the single-plane kernel's time, `WG_ABL=1`).  Every field a loop uses is copied once into
    // a register the optimiser cannot re-derive.
    struct { int N, HW, W, Cin, Cout, steps, period, U0, RB, D, swz; } p;
#define WG_PIN(F) { int v_ = pd.F; asm volatile("" : "+s"(v_)); p.F = v_; }
    WG_PIN(N) WG_PIN(HW) WG_PIN(W) WG_PIN(Cin) WG_PIN(Cout) WG_PIN(steps) WG_PIN(period) WG_PIN(U0) WG_PIN(RB) WG_PIN(D) WG_PIN(swz)
#undef WG_PIN
    const unsigned gb0 = (unsigned)grp.start[di], gG = (unsigned)(grp.start[di + 1] - grp.start[di]);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int RROWS = (p.RB + 3) * 16;                 // ring rows incl. the mirrors of slots 0, 1 and 2
    const int A_SUB = RROWS * 64;                      // bytes per (plane, half) sub-plane
    char* const dyb = lds + 2 * NPL * A_SUB;           // dy stages
    constexpr int D_SUB = 64 * 64, D_STAGE = NDH * NPL * D_SUB;
    const int NST = p.D + 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, k4 = lane >> 4;
    const int wn = wave & 3;                           // wave tile: input channels 16 wn .. + 15, all 64 output channels
    const int kg = wave >> 2;                          // wave group: pixels 32 kg .. + 31 of every step (CO2: output channels 64 kg .. + 63 of the block's 128)
    const int nci = p.Cin >> 6, nco = CO2 ? p.Cout >> 7 : p.Cout >> 6;
    int b;
    {
        const unsigned G = gG, bb = blockIdx.x - gb0, q = G >> 3, r = G & 7, x = bb & 7;
        b = (int)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bb >> 3));
    }
    const int cit = b % nci; b /= nci;
    const int cot = b % nco; b /= nco;
    const int ks = b;
    const int ci0 = cit * 64, co0 = cot * (CO2 ? 128 : 64);
    const int s_begin = ks * pd.steps_per, s_end = min(p.steps, s_begin + pd.steps_per);

    // ---- DMA roles: the waves of K group 0 issue everything.  Cycle stamps (`WG_ABL=512`) show the two waves of a SIMD far apart: the
    // K group 0 wave gets the matrix cores first and is through its taps in 1 150 (one plane) / 2 700 (two planes) cycles, its partner needs
    // 1 650 / 4 300 — group 0 waited 500 / 1 500 cycles per step at the barrier, and what the dy-issuing waves of group 1 spent on DMAs was
    // pure critical path.  Two planes: wave w < 4 stages sub-plane (P = w >> 1, half = w & 1) of BOTH operands; one plane: waves 0, 1 the
    // activation halves, waves 2, 3 the dy halves.
    const int dH = wave & 1;
    const int dP = NPL == 2 ? (wave >> 1) & 1 : 0;
    const bool dma_a = kg == 0 && (NPL == 2 || (wave & 2) == 0), dma_d = kg == 0 && (NPL == 2 || CO2 || (wave & 2) != 0);
    const int dQ = CO2 ? (wave & 3) : dH;                // dy sub-plane this wave stages (CO2: the four 32-channel quarters of the block's 128, one per wave of group 0)
    constexpr int WG_NMIN = NPL == 2 ? 8 : 4;          // DMAs of a prefetch group in an issuing wave, at least
    // Addresses are formed on the SCALAR unit: a DMA reads from (64-bit scalar base) + (32-bit lane offset); the lane offset — row
    // (lane >> 2) of the 16-row block, 16-byte chunk lane & 3 of the sub-plane's 64-byte row piece — never changes, the base walks.
    // (lanes 32..63 fill rows 8..15 of the block: their chunk is the one of the other 32-byte half, see (b) above)
    const int swz_a = (p.swz && (lane & 32)) ? 2 : 0, swz_d = (lane & 32) ? 2 : 0;
    const unsigned a_voff = (unsigned)(((lane >> 2) * p.Cin + ci0 + dH * 32 + ((lane & 3) ^ swz_a) * 8) * 2);
    const unsigned d_voff = (unsigned)(((lane >> 2) * p.Cout + co0 + dQ * 32 + ((lane & 3) ^ swz_d) * 8) * 2);
    unsigned v_zero = 0;
    asm volatile("" : "+v"(v_zero));
    const char* const a_base = reinterpret_cast<const char*>(dP ? pd.a_lo : pd.a_hi);
    const char* const d_base = reinterpret_cast<const char*>(dP ? pd.d_lo : pd.d_hi);
    const char* const zero_base = reinterpret_cast<const char*>(g_zero_wg);
    const unsigned a_dst = (unsigned)((dP * 2 + dH) * A_SUB);
    auto dma = [&](const char* sbase, unsigned voff, unsigned dst) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory", "m0");
    };
    // The activation cursor: the next 16-row block of the virtual pixel stream is rows a_q .. a_q + 15 of image a_img's period (rows
    // HW .. period - 1 are the zero gap; image -1 is the lead-in in front of image 0), a_off the element offset of the next REAL row
    // (images are contiguous in memory, so it simply advances by 16 rows per real block).  U0, HW and the period are multiples of 16.
    int a_img = 0, a_q = 0, a_off = 0;
    auto issue_a = [&](int slot) {
        const bool ok = (unsigned)a_img < (unsigned)p.N && a_q < p.HW;
        const unsigned dst = a_dst + (unsigned)slot * 1024u;
        if (ok) {
            const char* const src = a_base + 2 * (long)a_off;
            dma(src, a_voff, dst);
            if (slot < 3) dma(src, a_voff, a_dst + (unsigned)(p.RB + slot) * 1024u);
            a_off += 16 * p.Cin;
        } else {
            dma(zero_base, v_zero, dst);
            if (slot < 3) dma(zero_base, v_zero, a_dst + (unsigned)(p.RB + slot) * 1024u);
        }
        a_q += 16;
        if (a_q == p.period) { a_q = 0; ++a_img; }
    };
    auto issue_d = [&](int pix0, int stage) {
        const unsigned dst = (unsigned)((dyb - lds) + stage * D_STAGE + (dP * NDH + dQ) * D_SUB);
        const char* src = d_base + 2 * (long)pix0 * p.Cout;
        const long pitch16 = 32L * p.Cout;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) { dma(src, d_voff, dst + blk * 1024u); src += pitch16; }
    };

    // ---- fragments: 16 channels x 32 pixels; lane (column l15, k block k4) gets pixels 8 k4 .. + 7 through two transpose reads 4 rows apart
    const int q4 = l15 >> 2, p4 = lane & 3;
    const int lane_off = (8 * k4 + q4) * 64 + 8 * p4;
    // dy rows are 32 kg + 8 k4 + q4 (+ 4): bit 3 = k4 & 1 for both reads; the sub-tile's 32-byte half (c & 1) is the other one in those rows
    const int dy_off[2] = {lane_off + (k4 & 1) * 32, lane_off + ((k4 & 1) ^ 1) * 32};
    // activation rows are rt + 8 k4 + q4 (+ 4) with rt = 16 m + (kx - 1) (+ 8 (ky - 1) at W = 8: no swap there): per kx and read
    int a_rd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int L = 8 * k4 + q4 + 4 * rd;
            const int sb = p.swz ? (((kx + 15 + L) & 15) >> 3) : 0;
            a_rd[kx][rd] = (L + kx - 1) * 64 + 8 * p4 + (((wn & 1) ^ sb) * 32);      // (the tap's own +-1 row rides in the lane offset)
        }
    auto trread = [&](const char* src) -> u32x2 {
        const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
        return __builtin_bit_cast(u32x2, v);
    };
    auto frag2 = [&](const char* s0, const char* s1) -> u32x4 {
        const u32x2 a = trread(s0), c = trread(s1);
        u32x4 r; r[0] = a[0]; r[1] = a[1]; r[2] = c[0]; r[3] = c[1];
        return r;
    };
    auto mma = [&](const u32x4& x, const u32x4& y, const f32x4& c) -> f32x4 {
        if (WG_ABL & 2) { f32x4 r = c; r[0] += __builtin_bit_cast(float, x[0] ^ y[1]); return r; }
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
    const bool do_colsum = pd.colsum != nullptr && cit == 0 && wn == 0;
    u32x4 ones; ones[0] = ones[1] = ones[2] = ones[3] = 0x3F803F80u;       // bf16 1.0 pairs

    // horizontal padding: this lane's 8 pixels start at x0 = (32 kg + 8 k4) mod W (steps start on multiples of 64 and W divides 64)
    // (CO2: a wave walks both 32-pixel halves, one mask pair per half)
    unsigned mLp[NPH], mRp[NPH];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        const int x0 = (32 * (CO2 ? ph : kg) + 8 * k4) & (p.W - 1);
        mLp[ph] = x0 == 0 ? 0xFFFF0000u : 0xFFFFFFFFu;             // tap kx = 0 reads x - 1: the pixel at x == 0 (element 0) contributes nothing
        mRp[ph] = x0 == p.W - 8 ? 0x0000FFFFu : 0xFFFFFFFFu;       // tap kx = 2 reads x + 1: the pixel at x == W - 1 (element 7)
    }

    unsigned long long st_pre = 0, st_cmp = 0, st_syn = 0;
    if (s_begin < s_end) {
        const int hb = p.U0 >> 4;
        int img = (s_begin * 64) / p.HW, q = s_begin * 64 - img * p.HW;
        int B0 = (img * p.period + q + p.U0) >> 4;
        int next_blk = B0 - hb, next_slot = 0;
        {   // the first block is virtual row img * period + q - U0: inside image img, or in the gap behind image img - 1 (q < U0)
            const int qq = q - p.U0;
            a_img = qq >= 0 ? img : img - 1;
            a_q = qq >= 0 ? qq : qq + p.period;
            a_off = (img * p.HW + (qq >= 0 ? qq : 0)) * p.Cin;
        }
        auto load_upto = [&](int blk_end) {
            if (dma_a) {
                if (blk_end - next_blk == 4 && a_q + 64 <= p.HW && (unsigned)a_img < (unsigned)p.N && next_slot >= 3 && next_slot + 4 <= p.RB) {
                    // the common group — four real blocks inside one image, no ring wrap, no mirrored slot — straight-line: the cursor walk
                    // below was 1 030 cycles per step in the issuing wave (stamps), this is the four DMAs and their scalar adds
                    const char* src = a_base + ((WG_ABL & 64) ? 0L : 2 * (long)a_off);
                    const unsigned dst = a_dst + (unsigned)next_slot * 1024u;
                    const long pitch16 = 32L * p.Cin;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { dma(src, a_voff, dst + j * 1024u); src += pitch16; }
                    a_off += 64 * p.Cin; a_q += 64; next_blk += 4;
                    next_slot = next_slot + 4 == p.RB ? 0 : next_slot + 4;
                } else {
                    for (; next_blk < blk_end; ++next_blk) {
                        issue_a(next_slot);
                        next_slot = next_slot + 1 == p.RB ? 0 : next_slot + 1;
                    }
                }
            }
        };
        // the prefetch cursor: step sP (pixel qP of its image) is the next one whose operands are requested, into dy stage stP
        int sP = s_begin, qP = q, stP = 0;
        int endP = B0 + 4 + hb;                                            // one past the last block the step at the cursor needs
        auto prefetch_step = [&]() {
            load_upto(endP);
            if (dma_d) issue_d(sP * 64, stP);
            ++sP; qP += 64; endP += 4;
            if (qP == p.HW) { qP = 0; endP += (p.period - p.HW) >> 4; }
            stP = stP + 1 == NST ? 0 : stP + 1;
        };
        prefetch_step();
        if (p.D == 2 && sP < s_end) {
            prefetch_step();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_NMIN) : "memory");      // (a group is >= WG_NMIN DMAs in every issuing wave: step s_begin has landed)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();

        const int kg_s = __builtin_amdgcn_readfirstlane(kg);
        const int ring_rows = p.RB * 16, row_base = p.U0 + (CO2 ? 0 : 32 * kg_s);      // this K group's 32 pixel rows of the block's first step start at ring row row_base
        int row_cur = row_base;
        int st = 0;
        for (int s = s_begin; s < ((WG_ABL & 32) ? s_begin + 1 : s_end); ++s) {
            // (the consumer's position advances by 64 rows per step, + the G zero rows behind an image's last step: kept as a row count)
            q += 64;
            int adv_rows = 64;
            if (q == p.HW) { q = 0; adv_rows += p.period - p.HW; }
            const bool pre = sP < s_end && !(WG_ABL & 1);              // (uniform) a group younger than step s + 1's goes out now
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
            if (WG_ABL & 512) ts0 = __builtin_readcyclecounter();
            if (pre) prefetch_step();
            if (WG_ABL & 512) ts1 = __builtin_readcyclecounter();
            // ---- compute: this wave group's 32 pixels of step s (CO2: both halves, one after the other, into the same accumulators)
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
            const char* const dy_hi = dyb + st * D_STAGE + (CO2 ? ph : kg_s) * (32 * 64) + (CO2 ? 2 * kg_s : 0) * D_SUB;
            const char* const a_hi = lds + (wn >> 1) * A_SUB;
            const int row_own = row_cur + 32 * ph;
            const unsigned mL = mLp[ph], mR = mRp[ph];
            const int ring = ring_rows;
            u32x4 dh[4], dl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const char* const src = dy_hi + (c >> 1) * D_SUB + dy_off[c & 1];
                dh[c] = frag2(src, src + 256);
                if constexpr (NPL == 2) dl[c] = frag2(src + NDH * D_SUB, src + NDH * D_SUB + 256);
            }
            if (do_colsum) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { accb[c] = mma(dh[c], ones, accb[c]); if constexpr (NPL == 2) accb[c] = mma(dl[c], ones, accb[c]); }
            }
            // the centre-column taps of the three kernel rows start at ring rows c_ky = row_own + (ky - 1) W, taken into (0, ring]: with the
            // +-1 of kx a fragment then spans rows c - 1 .. c + 32 <= ring + 32, inside the ring or its three mirrored slots (row_own >= U0 > W,
            // so c is positive before the wrap) — three scalar wraps per step instead of nine
            const char* cky[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int c = row_own + (ky - 1) * p.W;
                c = c > ring ? c - ring : c;
                cky[ky] = a_hi + __builtin_amdgcn_readfirstlane(c * 64);
            }
            constexpr int NF = WG_NF;                  // fragment sets in flight: taps t .. t + NF - 1
            u32x4 fh[NF], fl[NF];
            auto load_tap = [&](int t) {
                const char* const b = cky[t / 3];
                const char* const s0 = b + a_rd[t % 3][0];
                const char* const s1 = b + a_rd[t % 3][1];
                fh[t % NF] = frag2(s0, s1);
                if constexpr (NPL == 2) fl[t % NF] = frag2(s0 + 2 * A_SUB, s1 + 2 * A_SUB);
            };
#pragma unroll
            for (int t = 0; t < NF - 1; ++t) load_tap(t);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kx = t % 3, cur = t % NF;
                if (t + NF - 1 < 9 && !(WG_ABL & 4)) load_tap(t + NF - 1);
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
            }
            row_cur += adv_rows; row_cur = row_cur >= ring_rows + row_base ? row_cur - ring_rows : row_cur;
            st = st + 1 == NST ? 0 : st + 1;
            // step s + 1 must have landed; with D = 2 the group issued at the top of THIS step (>= 4 DMAs per issuing wave) may stay in flight
            if (WG_ABL & 512) ts2 = __builtin_readcyclecounter();
            if (!(WG_ABL & 8)) {
            if (pre && p.D == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_NMIN) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            }
            if (WG_ABL & 512) { const unsigned long long ts3 = __builtin_readcyclecounter(); st_pre += ts1 - ts0; st_cmp += ts2 - ts1; st_syn += ts3 - ts2; }
        }
        if ((WG_ABL & 512) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 37))
            printf("wgwin stamps block %d wave %d steps %d: prefetch %llu compute %llu wait+barrier %llu cycles per step\n", (int)blockIdx.x, wave, s_end - s_begin,
                   st_pre / (s_end - s_begin), st_cmp / (s_end - s_begin), st_syn / (s_end - s_begin));
    }

    // fold the second K group into the first through LDS (the ring is dead now), three taps at a time: [wave & 3][48 + 16][64] floats
    // (CO2: the groups own different output channels — nothing to fold, all eight waves store)
    if constexpr (!CO2) {
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
    const int cow = co0 + (CO2 ? 64 * kg : 0);            // first output channel of this wave
    float* const ob = pd.out + (pd.ksplit > 1 ? (long)ks * p.Cout * ldo : 0L) + (long)(cow + 4 * k4) * ldo + ci0 + wn * 16 + l15;
    const bool accum = pd.ksplit == 1 && pd.accumulate;
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
            for (int r = 0; r < 4; ++r) atomicAdd(pd.colsum + cow + 16 * c + 4 * k4 + r, accb[c][r]);
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
    // 128-output-channel block tiles: every member of a launch must allow them.  Measured (tools/wgwin_co2.py): the grouped launches of a level
    // gain 5-19 % (x7 128 -> 128 @ 32 x 32, batch 256: 574 -> 500 us; x7 256 -> 256 @ 16 x 16: 608 -> 493), a lone conv mostly loses (half the tiles:
    // twice the splits) — tune key 1 = launches of two or more members, 2 = every launch (tests), 0 = never
    const int co2_mode = cdae_tune(TUNE_WGWIN_CO2);
    bool co2_all = single && (co2_mode >= 2 || (co2_mode == 1 && n >= 2));
    for (int i = 0; i < n && co2_all; ++i) co2_all = items[i].Cout % 128 == 0;
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
            // ring = live window (4 + 2 hb) + D prefetch groups (the largest: 4 + G/16 blocks) + 1 spare; D = 2 where the image fits the 160 KB
            const int npl = single ? 1 : 2;
            auto lds_bytes = [&](int D) { return (size_t)2 * npl * ((4 + 2 * hb) + D * (4 + G / 16) + 1 + 3) * 1024 + (size_t)(D + 1) * (co2_all ? 4 : 2) * npl * 4096; };
            p.D = (cdae_tune(TUNE_WGWIN_DIST) >= 2 && lds_bytes(2) <= 160 * 1024) ? 2 : 1;
            p.RB = (4 + 2 * hb) + p.D * (4 + G / 16) + 1;
            p.swz = (cdae_tune(TUNE_WGWIN_SWZ) && it.W >= 16) ? 1 : 0;
            int sh = 0;
            while ((1u << sh) < (unsigned)p.period) ++sh;
            p.period_magic = (unsigned)(((unsigned long long)((1ull << sh) - (unsigned)p.period) << 32) / (unsigned)p.period) + 1u;
            p.period_shift = sh;
            p.accumulate = it.accumulate; p.colsum = it.dbias;
            tiles[d] = (long)(it.Cin / 64) * (it.Cout / (co2_all ? 128 : 64));
            slab[d] = (size_t)it.Cout * 9 * it.Cin * sizeof(float);
            tiles_all += tiles[d];
            size_t sm = lds_bytes(p.D);
            if (sm < 65536) sm = 65536;                        // (the fold of the two K groups uses [4][64][64] floats of it)
            if (sm > smem) smem = sm;
            flops += 2.0 * it.Cout * 9.0 * it.Cin * (double)it.N * p.HW;
            if (it.dbias && !it.accumulate && hipMemsetAsync(it.dbias, 0, sizeof(float) * it.Cout, st) != hipSuccess) return cdae_fail("dbias memset failed");
        }
        // S = steps per block.  Candidates: every steps_d / k; cost = rounds of 256 blocks x (S + ~12 steps of prologue / epilogue / fold) + what
        // the splits cost (slab round trip + finish launch, about a dozen steps' worth, once per launch that has any)
        static const int cfg_blocks = CDAE_DEV_INT("CDAE_WG_BLOCKS", 256);      // one block per CU fits (84-134 KB of LDS)
        const int fixed = cdae_tune(TUNE_WGWIN_FIXED);
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
                const long cost = rounds * (S + fixed) + (any ? fixed : 0);
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
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgwin_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wgwin_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
            attr_bytes = smem;
        }
        cdae_prof_begin(PROF_IGEMM, flops, st);
        if (cdae_prof_on()) {
            char tag[128];
            const WgParams& p = g.d[0];
            snprintf(tag, sizeof(tag), "wgwin x%d %d->%d @%dx%d n=%d tiles=%ld blocks=%d S=%d ks0=%d planes=%d", nd, p.Cin, p.Cout, p.HW / p.W, p.W, p.N, tiles_all, nblk, bestS,
                     p.ksplit, co2_all ? 128 : single ? 1 : 2);
            cdae_prof_tag(tag);
        }
        if (co2_all) hipLaunchKernelGGL((wgwin_kernel<1, true>), dim3((unsigned)nblk), dim3(512), smem, st, g);
        else if (single) hipLaunchKernelGGL((wgwin_kernel<1>), dim3((unsigned)nblk), dim3(512), smem, st, g);
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
