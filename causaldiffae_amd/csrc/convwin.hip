// convwin.hip — stride-1 conv3x3 (and the 2x2 sub-pixel phases of upsample+conv) on pre-split 16-bit planes, gfx950 only.
// Second generation of the window-resident kernel (igemm.hip pswin_kernel); reference op: the two conv3x3 of a ResBlock,
// /root/reference/improved_diffusion/unet.py:143-162,185-198 (plain fp32 ATen conv2d there).
//
// What changed against pswin_kernel and why (round-1 profile: MFMA busy 0.40, waves parked 0.44 of the time):
//  * 4 waves per block, each owning a 128 x 64 tile of the 256 x 128 block tile, two blocks per CU (two waves per SIMD from
//    DIFFERENT barrier domains): 96 MFMAs per wave per barrier instead of 24, 0.75x the LDS fragment bytes per MFMA.
//  * v_mfma_f32_16x16x32_{f16,bf16}: the same flops per cycle as 32x32x16 at visibly lower power — a register-resident loop
//    sustains 1980 vs 1650 TFLOP/s on this part (tools/hiptests/mfma_peak.hip), and K = 32 per instruction lets ONE MFMA
//    span two (tap, 16-channel) units: lanes 0-31 supply unit 2s, lanes 32-63 unit 2s+1 — every lane reads its own 16 bytes
//    of LDS anyway, so the two halves simply read different taps (row shifts) / different window halves.
//  * the activation window is kept as two 16-channel halves; K order is (16-channel group, tap, channel).  A half is dead after
//    the last tap of its group and is reloaded (LDS-DMA) four K-steps before the group after next needs it — the window load
//    latency, which pswin_kernel exposed once per 32-channel chunk, is hidden behind MFMA work.
//  * LDS rows are 32 bytes, unswizzled: the hardware's ds_read_b128 lane groups ({0-3,12-15,20-27}, ...) see 16 distinct rows
//    with the pieces alternating exactly so that 2*row + piece covers all sixteen 16-byte bank groups for ANY row shift.
//
// Product per element (same as every split-precision kernel here): lo*hi + hi*lo + hi*hi, fp32 accumulate.
// LDS map (80 KB, two blocks per CU): window [half 2][plane 2][384 rows][32 B] = 48 KB, weights [stage 2][plane 2][khalf 2]
// [128 rows][32 B] = 32 KB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "cdae_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (Compile-time variants of the K loop that were measured and removed — next-tile fragment reads in front of the tile's MFMAs, the mid-step
// wait behind six of tile 4's MFMAs, the next step's geometry inside tile 5's MFMA burst, de-phasing the two blocks of a CU by half a
// tile, a channel-major result tile with 16-byte stores: docs/NOTES.md.)

namespace {

__device__ __forceinline__ int fdiv_cw(int n, unsigned magic, int shift) {             // exact floor(n / d), see igemm.hip fdiv
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

__device__ __forceinline__ void store_planes_cw(const GemmParams& p, long addr, float v) {
    asm volatile("" : "+v"(v));        // opaque: one rounding of hi (see attention.hip split8)
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    p.C_hi[addr] = __builtin_bit_cast(unsigned short, h);
    p.C_lo[addr] = __builtin_bit_cast(unsigned short, l);
}

__device__ __attribute__((aligned(64))) unsigned g_zero_cw[16] = {0u};      // a zero line: the bias / residual of launches that have none

constexpr int CW_BM = 256, CW_BN = 128;
constexpr int CW_A_PLANE = 12288, CW_A_HALF = 2 * CW_A_PLANE;               // 384 rows x 32 B
constexpr int CW_B_BASE = 2 * CW_A_HALF, CW_B_KH = 4096, CW_B_PLANE = 2 * CW_B_KH, CW_B_STAGE = 2 * CW_B_PLANE;
constexpr int CW_LDS = CW_B_BASE + 2 * CW_B_STAGE;                           // 81920
constexpr int CW_WIN_ROWBLOCKS = 3;                                          // window DMAs per wave per reload: 3 row blocks x NPL planes
// Fragment reads of taps that fall on padding go to an LDS address beyond every allocation: the hardware returns zeros for
// out-of-range DS reads (tools/hiptests/lds_oob.hip; tests/test_gpu_kernels.py::test_lds_out_of_range_reads_return_zero), which
// replaces eight v_and per 16-row tile and K-step — the loop's vector-issue budget belongs to the MFMAs.
constexpr unsigned CW_OOB = 0x0003C000u;       // + the largest read offset stays below 2^18 (0x100000 reads in-range data: the DS address wraps)

// LDS traffic of the K loop goes through inline assembly: the kernel counts its own lgkmcnt (hipcc waits with lgkmcnt(0) whenever
// anything is outstanding, which would expose the latency of the prefetches issued a few instructions earlier).
template <int IMM>
__device__ __forceinline__ u32x4 cw_lds_read(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
// lane l's 16 bytes at (base + voff) -> LDS byte lds_dst + 16 l; base and lds_dst wave-uniform, voff a 32-bit byte offset
// (sbase a kernel-argument pointer, soff a uniform byte offset: readfirstlane pins the sum to the scalar unit — hipcc is free to
// compute uniform values on the vector ALU and would then hand the "s" operand a VGPR pair)
__device__ __forceinline__ void cw_dma(const void* sbase, int soff, unsigned voff, unsigned lds_dst) {
    const char* base = reinterpret_cast<const char*>(sbase) + __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "v"(voff), "s"(base) : "memory", "m0");
}

// sum of a value over the 16 lanes of its DPP row (lanes 16 k .. 16 k + 15), left in every lane: quad butterfly, then half-row and row mirrors
__device__ __forceinline__ float cw_row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));      // quad_perm [2, 3, 0, 1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));     // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));     // row_mirror
    return x;
}

// NT = 9: the 3x3 window.  NT = 4: the 2x2 window of one sub-pixel phase (ph_y, ph_x) of nearest-2x-upsample + conv3x3 (tap t reads window
// position (t / 2 + ph_y, t % 2 + ph_x) of the same 3x3 neighbourhood; weights [rows][4][K]; result scattered to (2y + ph_y, 2x + ph_x)).
// NPL = 2: hi / lo plane pairs, three MFMAs per product (the fp32-parity modes).  NPL = 1: ONE plane per operand and one MFMA per product —
// the reduced-precision torso (`mixed16`: f16 forward, bf16 dgrad); the lo halves of the LDS image stay unused.
// NJ: 16-column MFMA tiles per wave = the n-tile is 32 NJ columns wide (two waves side by side).  4: 256 x 128 tiles.  3: 256 x 96 — for
// Cout = 384 at the 16 x 16 level of a batch-128 step, where 128 m-tiles x 3 n-tiles of 128 are 384 blocks for 512 block slots (a quarter of
// the CUs run one block instead of two) and 128 x 4 n-tiles of 96 are exactly 512.  The weight stage keeps its 128-row geometry (rows beyond
// the n-tile are loaded and never read).
// IO16 (one-plane instantiations of the 16-bit torso): the result and the residual are bf16 rows (GemmParams::io16 bits 0 and 1 both set)
// PAIR (NPL = 2, the 16-bit torso): the two plane slots carry the FIRST and the SECOND HALF of the input channels of a one-plane operand
// (the launch points A_lo / Bk_lo half the channels further and halves Cin): two MFMAs per product pair — hi.hi + lo.lo — for the fragment
// reads, barrier and DMAs of one K step, i.e. half the K steps of the NPL = 1 form, whose step is bound by exactly those fixed costs.
// IO16 instantiations are CHANNEL-MAJOR (round 6): the weight fragment sits in the MFMA's first operand, so a lane's accumulator registers are
// four consecutive output CHANNELS of one pixel, and with the weight rows permuted inside each 64-row group at DMA time (LDS row 16 j + l holds
// channel 16 (l >> 2) + 4 j + (l & 3): the fragment reads keep their conflict-free addresses) a lane owns 16 consecutive channels of a pixel per
// 16-row tile — the bf16 result and the residual move as two 16-byte pieces per lane and tile row instead of 16 two-byte ones (the row-major
// epilogue was 17 us of a 101 us conv without and 47 of 131 us with a residual: 128 -> 128 at 32 x 32, batch 256).  The GroupNorm partial sums
// then need the 16 lanes of a DPP row added up (cw_row16_sum).  (On fp32 rows this form lost in round 3 — the residual's tag look-ups; fp32
// stores are dwords either way — and stays row-major.)
template <bool BF, int NT, int NPL = 2, int NJ = 4, bool IO16 = false, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void convwin_kernel(const GemmParams p, const int ntiles) {
    typedef const unsigned short* hp;
    constexpr int BN = 32 * NJ, WNC = 16 * NJ;                         // n-tile width, columns per wave
    constexpr bool SWAP = IO16;
    static_assert(!IO16 || NJ == 4, "bf16-row instantiations: 64 channels per wave");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kg = lane >> 4, pc = kg & 1;
    const bool selb = kg >= 2;                                       // this lane feeds K 16..31 of every MFMA: the step's SECOND unit
    const int wm = wave >> 1, wn = wave & 1;

    const int nmt = (p.M + CW_BM - 1) / CW_BM, nnt = (p.N + BN - 1) / BN;
    const hp a_hi = reinterpret_cast<hp>(p.A), a_lo = p.A_lo;
    const bool packed = p.Bk_hi != nullptr;                            // weights [K / 16][9][rows][16]: a (group, tap) unit is one contiguous run
    const hp b_hi = packed ? p.Bk_hi : reinterpret_cast<hp>(p.B), b_lo = packed ? p.Bk_lo : p.B_lo;
    const int W = p.W;                                                 // tight window of 256 + 2 W rows: row j <-> flattened pixel pix0 + j
    const int nchunk = p.Cin >> 5;
    const int c_per = (nchunk + p.ksplit - 1) / p.ksplit;

    // ---- state of the tile being computed (a persistent block walks tiles blockIdx.x, + gridDim.x, ...)
    int m0, n0, ks, g_begin, g_end, nsteps;
    int cph = 0, cph_y = p.ph_y, cph_x = p.ph_x;   // sub-pixel phase of the tile (NT = 4; all four phases of an up-conv can share one launch)
    unsigned aoffb[3];             // window DMA: byte offset (group 0) of this lane's 16 bytes of rows 32 (wave + 4 q) + lane / 2
    unsigned woffb;                // weight DMA: byte offset of row n0 + 32 wave + lane / 2 (tap 0, group 0)
    int tapmask[8];                // per 16-row tile: bit (3 ky + kx) set when tap (ky, kx) of this lane's pixel reads a real pixel
    auto setup = [&](int tile) {
        // lane coordinates through an opaque asm: hipcc otherwise hoists every lane-dependent sub-expression of this function (8 row
        // indices, the DMA row offsets ...) out of the persistent loop and keeps them live across the K loop — the kernel sits at the
        // 256-VGPR limit of two waves per SIMD and the allocator answers with spills whose reloads land INSIDE the K loop
        int lane_s = lane;
        asm volatile("" : "+v"(lane_s));
        const int lr_s = lane_s & 15;
        int mt, nt;
        {       // XCD-aware decode: consecutive tiles of one XCD share the activation window / the weight tile in that XCD's L2
            const unsigned G = (unsigned)ntiles, b = (unsigned)tile;
            const unsigned q = G >> 3, r = G & 7, x = b & 7;
            unsigned v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
            if (NT == 4 && p.nphase > 1) { cph = v & 3; v >>= 2; cph_y = cph >> 1; cph_x = cph & 1; }      // phase fastest: the four phases of a tile read the same window
            nt = v % nnt; v /= nnt;
            mt = v % nmt; ks = v / nmt;
        }
        // the tile coordinates are wave-uniform, but the divisions above run on the vector ALU: without the readfirstlane everything
        // derived from them (K-loop bounds, the weight / window DMA offsets) stays in VGPRs and the K loop carries v_mul_lo_u32 (quarter
        // rate) and v_readfirstlane where s_mul_i32 does the job beside the MFMAs
        mt = __builtin_amdgcn_readfirstlane(mt); nt = __builtin_amdgcn_readfirstlane(nt); ks = __builtin_amdgcn_readfirstlane(ks);
        m0 = mt * CW_BM; n0 = nt * BN;
        const int c_begin = ks * c_per, c_end = max(min(nchunk, c_begin + c_per), c_begin);       // the last splits of an uneven division are empty
        g_begin = 2 * c_begin; g_end = 2 * c_end;                     // 16-channel groups of this K split
        nsteps = ((g_end - g_begin) * NT) >> 1;                        // K-steps of two (group, tap) units each (0: the slab is zeros)
        const int pix0 = m0 - W;
        // rows outside the tensor (and weight rows beyond Cout) are CLAMPED, not zero-filled: whatever lands there is only ever
        // addressed by padding taps (redirected out of range) or feeds output columns that are never stored
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int pix = min(max(pix0 + (wave + 4 * q) * 32 + (lane_s >> 1), 0), p.M - 1);
            aoffb[q] = (p.a_gm ? (unsigned)pix * 32u : (unsigned)pix * (unsigned)p.sx * 2u) + (lane_s & 1) * 16u;
        }
        int wr_ = wave * 32 + (lane_s >> 1);                          // LDS row of the 128-row weight stage this lane fetches
        // ... holds this channel.  (K-split launches keep the natural order: their fp32 slab pieces — register j = channels 16 j + 4 kg .. + 3 —
        //  then lie 64 bytes contiguous per row and store instruction; permuted, the pieces of a row are 64 bytes apart: measured +17 % on those convs)
        if constexpr (SWAP) { if (p.ksplit == 1) { const int l_ = wr_ & 15, j_ = (wr_ >> 4) & 3; wr_ = (wr_ & 64) + 16 * (l_ >> 2) + 4 * j_ + (l_ & 3); } }
        const int wrow = min(n0 + wr_, p.N - 1);
        woffb = packed ? (unsigned)wrow * 32u + (lane_s & 1) * 16u : (unsigned)wrow * (unsigned)p.ldb * 2u + (lane_s & 1) * 16u;
        if (NT == 4 && p.nphase > 1) woffb += (unsigned)cph * (unsigned)p.phase_w * 2u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm * 128 + 16 * i + lr_s;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int n = fdiv_cw(mm, p.hw_magic, p.hw_shift), rem = mm - n * p.hw;
            const int y = fdiv_cw(rem, p.wo_magic, p.wo_shift), x = rem - y * p.Wo;
            int mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = y - 1 + t / 3, tx = x - 1 + t % 3;
                mk |= (ok && ty >= 0 && ty < p.H && tx >= 0 && tx < W) ? (1 << t) : 0;
            }
            tapmask[i] = mk;
        }
    };

    const int gstride = p.a_gm ? p.M * 32 : 32;                        // bytes between 16-channel groups of the activation planes
    auto issue_window = [&](int g) {                                  // group g -> half g & 1
        const unsigned dst = (g & 1) * CW_A_HALF + wave * 1024;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            cw_dma(a_hi, g * gstride, aoffb[q], dst + q * 4096);
            if constexpr (NPL == 2) cw_dma(a_lo, g * gstride, aoffb[q], dst + q * 4096 + CW_A_PLANE);
        }
    };
    // weights of the step whose first unit is (g, t) -> stage
    auto issue_weights = [&](int stage, int g, int t) {
        int gb = g, tb = t + 1;
        if (tb == NT) { tb = 0; ++gb; }
        const unsigned dst = CW_B_BASE + stage * CW_B_STAGE + wave * 1024;
        int ea, eb;
        if (packed) { ea = (g * NT + t) * p.N * 32; eb = (gb * NT + tb) * p.N * 32; }
        else { ea = (t * p.Cin + g * 16) * 2; eb = (tb * p.Cin + gb * 16) * 2; }
        cw_dma(b_hi, ea, woffb, dst);
        cw_dma(b_hi, eb, woffb, dst + CW_B_KH);
        if constexpr (NPL == 2) {
            cw_dma(b_lo, ea, woffb, dst + CW_B_PLANE);
            cw_dma(b_lo, eb, woffb, dst + CW_B_PLANE + CW_B_KH);
        }
    };
    auto issue_prologue = [&]() {
        if (nsteps == 0) return;
        issue_window(g_begin);
        issue_window(g_begin + 1);
        issue_weights(0, g_begin, 0);
        if (NT == 9) issue_weights(1, g_begin, 2);
        else issue_weights(1, g_begin, 2);
    };

    unsigned a_lane = (wm * 128 + lr) * 32 + pc * 16;            // byte offset of (tile 0 row, piece) in a window plane (re-pinned in front of every K loop)
    unsigned b_lane = CW_B_BASE + (selb ? CW_B_KH : 0) + (wn * WNC + lr) * 32 + pc * 16;

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const u32x4& x_, const u32x4& y_, const f32x4& c) -> f32x4 {
        const u32x4& x = SWAP ? y_ : x_;                               // (activation fragment, weight fragment): channel-major tiles take the weights first
        const u32x4& y = SWAP ? x_ : y_;
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
    };
    // geometry of the step whose first unit is (g, t): this lane's mask bit and byte address of its tile-0 fragment (half + row shift)
    auto geom = [&](int g, int t, int& wtap, unsigned& addr) {
        int gb = g, tb = t + 1;
        if (tb == NT) { tb = 0; ++gb; }
        const int my_t = selb ? tb : t, my_g = selb ? gb : g;
        const int ky = NT == 9 ? (my_t * 11) >> 5 : (my_t >> 1) + cph_y;  // tap / 3 for tap < 9
        const int kx = NT == 9 ? my_t - 3 * ky : (my_t & 1) + cph_x;
        wtap = 3 * ky + kx;
        addr = a_lane + (my_g & 1) * CW_A_HALF + (ky * W + kx - 1) * 32;
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    // dev (CDAE_PS_DBG & 32): s_memtime around the waits, summed per wave, written to the split-K workspace by lane 0 of the first 64 blocks
    // dev ablations / stamps (CDAE_PS_DBG bits) exist only in a -DCW_DEV=1 build: as run-time tests they cost the K loop 15 scalar branches
    // per step (and, with the de-phasing state, one around every MFMA triple: 4-5 % of the kernel)
    const int dbg_ = CW_DEV ? p.dbg : 0;
    const bool stamps = (dbg_ & 32) != 0;
    unsigned long long t_top = 0, t_vm = 0, t_bar = 0, t_epi = 0;
    auto now = [&]() -> unsigned long long { return stamps ? (unsigned long long)__builtin_readcyclecounter() : 0ull; };
    const unsigned long long t_begin = now();
    setup(tile);
    issue_prologue();
    while (true) {
        const unsigned long long t0_ = now();
        // the tile's first window halves and weight stages (and the previous tile's stores).  The BUILTIN wait, not asm: hipcc's waitcnt
        // pass must see that nothing it knows of (epilogue loads) is pending when the K loop starts, or it plants a vmcnt(0) on the
        // first register redefinition inside the loop — a full drain of the DMAs the loop has just issued, every step
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        t_top += now() - t0_;
        int ga = g_begin, ta = 0;              // first unit of the current step
        int gi = g_begin + (NT == 4 ? 1 : 0), ti = NT == 4 ? 0 : 4;      // first unit of the next step whose weights are to be staged (step 2 = unit 4)
        int g_old = g_begin;                   // oldest group whose window half is still live
        bool reload_prev = false;
        int wtap_c; unsigned a_c;
        geom(ga, ta, wtap_c, a_c);
        unsigned b_c = b_lane;                 // this step's weight stage
        u32x4 bh[NJ], ah[2], bl_[NPL == 2 ? NJ : 1], al_[NPL == 2 ? 2 : 1];      // (one plane: the lo names alias the hi registers)
#define al(C) (NPL == 2 ? al_[(NPL == 2) * (C)] : ah[C])
#define bl(J) (NPL == 2 ? bl_[(NPL == 2) * (J)] : bh[J])
        // fragment address of padding taps -> out of range -> zeros
#define CW_READ_A(I, BUF, WTAP, ADDR) { const unsigned m_ = (unsigned)__builtin_amdgcn_sbfe(tapmask[I], WTAP, 1); \
            const unsigned ad_ = (ADDR & m_) | (CW_OOB & ~m_); \
            ah[BUF] = cw_lds_read<(I) * 512>(ad_); if constexpr (NPL == 2) al(BUF) = cw_lds_read<(I) * 512 + CW_A_PLANE>(ad_); }
#define CW_READ_B(J, ADDR) { bh[J] = cw_lds_read<(J) * 512>(ADDR); if constexpr (NPL == 2) bl(J) = cw_lds_read<(J) * 512 + CW_B_PLANE>(ADDR); }
#define CW_WAIT(...) asm volatile(__VA_ARGS__)
        if (nsteps > 0) {
            CW_READ_A(0, 0, wtap_c, a_c);
            CW_READ_B(0, b_c); CW_READ_B(1, b_c);
            if constexpr (NJ >= 3) CW_READ_B(2, b_c);
            if constexpr (NJ == 4) CW_READ_B(3, b_c);
        }
        // Per-lane loop state that the register allocator spills around the epilogue must be back in registers HERE: a scratch reload is a
        // vector-memory load, and when its first use sits inside the K loop hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of that
        // use — a full drain of the weight / window DMAs just issued, every K step (round 2's kernel had exactly that at the top of its
        // loop; tools/isa_lint.py finds it and tests/test_host_cpu.py::test_window_conv_k_loop_has_no_compiler_drain guards it).
        // The empty asm pins the values; the builtin wait (vmcnt only: the fragment reads above stay in flight) tells the pass that every
        // reload up to here has landed — nothing is outstanding in vmcnt at this point anyway (the tile's operands were waited for above).
        asm volatile("" : "+v"(wtap_c), "+v"(a_c), "+v"(b_c), "+v"(woffb), "+v"(a_lane), "+v"(b_lane), "+v"(aoffb[0]), "+v"(aoffb[1]), "+v"(aoffb[2]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(tapmask[i]));
        __builtin_amdgcn_s_waitcnt(0x0F70);

        for (int s = 0; s < nsteps; ++s) {
            int gn = ga, tn = ta + 2;          // first unit of the next step
            if (tn >= NT) { tn -= NT; ++gn; }
            int wtap_n; unsigned a_n;
            geom(gn, tn, wtap_n, a_n);
            const unsigned b_n = b_lane + ((s + 1) & 1) * CW_B_STAGE;
            // the two waves of a SIMD belong to different blocks: alternating the issue priority by step parity lets one of them run
            // its MFMA burst unbroken while the other is at its mid-step wait (measured +2..3 %; CDAE_PS_DBG & 64 turns it off)
            if (!(dbg_ & 64)) { if (s & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            // outstanding LDS reads here, oldest first: A(0) [2], B(0) [2], B(1) [2], B(2) [2], B(3) [2]      (NJ = 4, two planes)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cur = i & 1;
                auto midstep = [&]() __attribute__((always_inline)) {
                    // ---- mid-step: every wave holds this step's weight fragments in registers, so the stage is free for step s + 2;
                    // the weights of step s + 1 (issued one step ago) must have landed; a window reload issued after them may stay in flight
                    const unsigned long long t1_ = now();
                    if (reload_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW_WIN_ROWBLOCKS * NPL) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long t2_ = now();
                    if (!(dbg_ & 8)) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (stamps) { const unsigned long long t3_ = now(); t_vm += t2_ - t1_; t_bar += t3_ - t2_; }
                    // NT = 4: a group lasts two steps only, so a reloaded half is read (look-ahead of the step after next) one step after
                    // it was requested: the window goes out BEFORE the weights and the next mid-step wait drains everything
                    bool reload = false;
                    if (ga > g_old) {              // every unit of g_old lies in finished steps: its half is free
                        reload = g_old + 2 < g_end && !(dbg_ & 20);
                        if (NT == 4 && reload) issue_window(g_old + 2);
                    }
                    if (s + 2 < nsteps && !(dbg_ & 4)) issue_weights(s & 1, gi, ti);
                    ti += 2;
                    if (ti >= NT) { ti -= NT; ++gi; }
                    reload_prev = false;
                    if (ga > g_old) {
                        if (NT == 9 && reload) { issue_window(g_old + 2); reload_prev = true; }
                        ++g_old;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                if (i == 4) midstep();
                // tile 0 needs A(0) and B(0) of the ten reads in flight; every other tile's A pair is the only thing outstanding
                // (one plane: never list a register twice — the aliased lo names would make hipcc copy a fragment whose read is still in flight)
                if constexpr (NPL == 2) {
                    if (i == 0) CW_WAIT("s_waitcnt lgkmcnt(%4)" : "+v"(ah[0]), "+v"(al(0)), "+v"(bh[0]), "+v"(bl(0)) : "n"(2 * (NJ - 1)));
                    else CW_WAIT("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]), "+v"(al(cur)));
                } else {
                    if (i == 0) CW_WAIT("s_waitcnt lgkmcnt(%2)" : "+v"(ah[0]), "+v"(bh[0]) : "n"(NJ - 1));
                    else CW_WAIT("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]));
                }
                if constexpr (NPL == 2 && PAIR) acc[i][0] = mma(al(cur), bl(0), acc[i][0]);
                else if constexpr (NPL == 2) {
                    acc[i][0] = mma(al(cur), bh[0], acc[i][0]);
                    acc[i][0] = mma(ah[cur], bl(0), acc[i][0]);
                }
                acc[i][0] = mma(ah[cur], bh[0], acc[i][0]);
                __builtin_amdgcn_sched_barrier(0);
                // the next tile's fragment reads go out behind the first three MFMAs and have nine MFMAs to land
                if (i == 0) { CW_READ_A(1, 1, wtap_c, a_c); }
                else if (i == 1) { CW_READ_A(2, 0, wtap_c, a_c); }
                else if (i == 2) { CW_READ_A(3, 1, wtap_c, a_c); }
                else if (i == 3) { CW_READ_A(4, 0, wtap_c, a_c); }
                else if (i == 4) { CW_READ_A(5, 1, wtap_c, a_c); }
                else if (i == 5) { CW_READ_A(6, 0, wtap_c, a_c); }
                else if (i == 6) { CW_READ_A(7, 1, wtap_c, a_c); }
                else { CW_READ_A(0, 0, wtap_n, a_n); CW_READ_B(0, b_n); }      // next step: its first tile and the first dead weight fragments
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 1; j < NJ; ++j) {
                    // tile 0: B(j) is the oldest of the reads in flight [B(j) .. B(NJ - 1), A(1)]
                    if constexpr (NPL == 2) { if (i == 0) CW_WAIT("s_waitcnt lgkmcnt(%2)" : "+v"(bh[j]), "+v"(bl(j)) : "n"(2 * (NJ - j))); }
                    else { if (i == 0) CW_WAIT("s_waitcnt lgkmcnt(%1)" : "+v"(bh[j]) : "n"(NJ - j)); }
                    if constexpr (NPL == 2 && PAIR) acc[i][j] = mma(al(cur), bl(j), acc[i][j]);
                    else if constexpr (NPL == 2) {
                        acc[i][j] = mma(al(cur), bh[j], acc[i][j]);
                        acc[i][j] = mma(ah[cur], bl(j), acc[i][j]);
                    }
                    acc[i][j] = mma(ah[cur], bh[j], acc[i][j]);
                    if (i == 7) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (j == 1) { CW_READ_B(1, b_n); } else if (j == 2) { if constexpr (NJ >= 3) CW_READ_B(2, b_n); } else { if constexpr (NJ == 4) CW_READ_B(3, b_n); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ta = tn; ga = gn; wtap_c = wtap_n; a_c = a_n; b_c = b_n;
        }
#undef al
#undef bl
#undef CW_READ_A
#undef CW_READ_B
#undef CW_WAIT

        __builtin_amdgcn_s_setprio(0);
        // ---- tile done: stage the next tile's operands, then write this tile's result (the stores drain behind the next K loop)
        const int em0 = m0, en0 = n0, eks = ks, eph = cph, eph_y = cph_y, eph_x = cph_x;
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the look-ahead reads of the step that does not exist
        __builtin_amdgcn_s_barrier();                                    // every wave is done reading the window and the weight stages
        asm volatile("" ::: "memory");
        if (has_next) { setup(next); issue_prologue(); }
        const unsigned long long t4_ = now();

        // accumulator tile (i, j): this lane holds rows 4 kg + r (r = 0..3) of column 16 j + lr; the four column tiles of a row are
        // stored back to back so that the 256 bytes a wave owns of each output row reach L2 together
        const bool interior = em0 + CW_BM <= p.M && en0 + BN <= p.N;
        // lane coordinates of the epilogue through an opaque asm (as in setup): the row / column offsets and pointers derived from them are
        // recomputed here, once per tile, instead of being hoisted out of the persistent loop and kept live across the K loop
        int lr_ = lr, kg_ = kg;
        asm volatile("" : "+v"(lr_), "+v"(kg_));
        const float alpha_ = p.w_scale ? p.alpha * p.w_scale[1] : p.alpha;      // the weight planes hold w * 2^k: the exact 2^-k rides on alpha
        if (dbg_ & 256) {}                                                // dev ablation: no epilogue
        else if constexpr (SWAP) {
            // channel-major tiles (bf16 rows; interior tiles only, cdae_convwin_ok): this lane holds pixel 16 i + lr of the wave's 128 rows and
            // channels 16 kg .. 16 kg + 15 of its 64 (register j, element r <-> channel 4 j + r)
            typedef __bf16 cw_bf8 __attribute__((ext_vector_type(8)));
            const int colb = en0 + wn * WNC + 16 * kg_;
            const long row_l = (long)(em0 + wm * 128 + lr_);
            if (p.ksplit > 1) {
                float* __restrict__ cb = p.splitk_ws + (long)eks * (long)p.M * p.N + row_l * p.N + en0 + wn * WNC + 4 * kg_;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(cb + (long)(16 * i) * p.N + 16 * j) = acc[i][j];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // 32-bit byte offsets from the (scalar) base pointers: a tile's addresses cost one VALU add each instead of 64-bit pointer
                // arithmetic per access (cdae_convwin_ok: M * ldc < 2^30 for these instantiations).  No branch around a memory operation: a missing
                // bias / residual reads a zero line instead (an `if (p.res)` per row is a scalar branch per row with a full vmcnt drain at its join).
                // Few values live beside the 128 accumulator registers: a row pair is finished eight channels at a time — its rounded results stay
                // packed (8 registers) for the GroupNorm sums.
                char* const cbase = reinterpret_cast<char*>(p.C);
                const bool has_res = p.res != nullptr;
                const char* const rbase = has_res ? reinterpret_cast<const char*>(p.res) : reinterpret_cast<const char*>(g_zero_cw);
                const unsigned ldc2 = (unsigned)p.ldc * 2u;                                   // row pitch, bytes
                const unsigned ob = ((unsigned)(em0 + wm * 128 + lr_) * (unsigned)p.ldc + (unsigned)colb) * 2u;
                const unsigned rmask = has_res ? 0xffffffffu : 0u;                            // (no residual: every lane reads byte 0 of the zero line)
                const float* __restrict__ bp = p.bias ? p.bias + colb : reinterpret_cast<const float*>(g_zero_cw);
                const unsigned gb = (((unsigned)(em0 + wm * 128) >> 5) * (unsigned)p.N + (unsigned)colb) * 8u;      // byte offset of (chunk, channel) in gn_part
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    float ssel = 0.f, qsel = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                        // channels 8 h .. 8 h + 7: accumulator registers j = 2 h, 2 h + 1
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + 8 * h), b1 = *reinterpret_cast<const f32x4*>(bp + 8 * h + 4);
                        cw_bf8 ov[2];
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const unsigned ro = ob + (unsigned)(32 * i2 + 16 * ii) * ldc2 + 16u * h;
                            const cw_bf8 rr = *reinterpret_cast<const cw_bf8*>(rbase + (ro & rmask));
                            float fin = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float v = acc[2 * i2 + ii][2 * h + (e >> 2)][e & 3] * alpha_ + (e < 4 ? b0[e & 3] : b1[e & 3]) + (float)rr[e];
                                const __bf16 h_ = (__bf16)v;
                                ov[ii][e] = h_;
                                fin += (float)h_;
                            }
                            *reinterpret_cast<cw_bf8*>(cbase + ro) = ov[ii];
                            // (the branch per row also keeps hipcc's register allocation in check — see the row-major epilogue below; accumulated
                            //  branch-free, the whole epilogue became one block with 299 spilled registers)
                            if (!__builtin_isfinite(fin) && p.range_flag) *p.range_flag = 1;
                        }
                        if (p.gn_part) {
                            // per (32-row chunk, channel): this lane's two rows, then the 16 lanes of the DPP row; every lane of the row ends with
                            // every channel's sums and keeps those of channel lr
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float a0 = (float)ov[0][e], a1 = (float)ov[1][e];
                                const float s_ = cw_row16_sum(a0 + a1), q_ = cw_row16_sum(a0 * a0 + a1 * a1);
                                ssel = lr_ == 8 * h + e ? s_ : ssel;
                                qsel = lr_ == 8 * h + e ? q_ : qsel;
                            }
                        }
                    }
                    if (p.gn_part) {                                     // ONE 8-byte store per lane and chunk (a row's 16 lanes: 128 contiguous bytes)
                        typedef float cw_f2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<cw_f2*>(reinterpret_cast<char*>(p.gn_part) + (gb + (unsigned)i2 * (unsigned)p.N * 8u + (unsigned)lr_ * 8u)) = cw_f2{ssel, qsel};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        else if (interior && p.ksplit == 1 && !p.accumulate && !p.C_hi) {
            // the common case, kept lean (the general path below spends ~20 instructions per element on bounds and mode tests):
            // one row pointer per (tile, r), the four column tiles at immediate offsets
            const bool up2 = p.out_mode == OUT_UP2;       // sub-pixel phase: GEMM row (n, y, x) -> output pixel (n, 2y + ph_y, 2x + ph_x)
            const int row0 = em0 + wm * 128 + 4 * kg_;
            const long lane_off = (up2 ? 0 : (long)row0 * p.ldc) + en0 + wn * WNC + lr_;
            float* __restrict__ cbase = p.C + lane_off;
            const float* __restrict__ rbase = p.res ? p.res + lane_off : nullptr;
            __bf16* __restrict__ cbase16 = reinterpret_cast<__bf16*>(p.C) + lane_off;
            const __bf16* __restrict__ rbase16 = reinterpret_cast<const __bf16*>(p.res) + lane_off;
            float bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = p.bias ? p.bias[en0 + wn * WNC + lr_ + 16 * j] : 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                float gs[NJ], gq[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    long ro[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (up2) {
                            const int row = row0 + 32 * i2 + 16 * ii + r, x = row & (p.Wo - 1);             // Wo is a power of two here
                            ro[r] = (4L * row - 2 * x + eph_y * 2 * p.Wo + eph_x) * p.ldc;
                        } else ro[r] = (long)(32 * i2 + 16 * ii + r) * p.ldc;
                    }
                    // the residual values of the row tile are requested together: row by row, each pair of loads waited for vmcnt(0) — 64
                    // exposed round trips per tile (and a drain of the next tile's operand DMAs each time)
                    float rv[4][NJ];
                    if (rbase) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) { if constexpr (IO16) rv[r][j] = (float)rbase16[ro[r] + 16 * j]; else rv[r][j] = rbase[ro[r] + 16 * j]; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) rv[r][j] = 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v[NJ];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) v[j] = (acc[2 * i2 + ii][j][r] * alpha_ + bv[j]) + rv[r][j];
                        if constexpr (IO16) {
#pragma unroll
                            for (int j = 0; j < NJ; ++j) { const __bf16 h_ = (__bf16)v[j]; cbase16[ro[r] + 16 * j] = h_; v[j] = (float)h_; gs[j] += v[j]; gq[j] += v[j] * v[j]; }
                        } else {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) { cbase[ro[r] + 16 * j] = v[j]; gs[j] += v[j]; gq[j] += v[j] * v[j]; }      // (nontemporal stores: measured +-0)
                        }
                        // f16 plane overflow surfaces as NaN / inf.  (The branch per row also keeps hipcc's register allocation in check: with a
                        // branch-free accumulated check the epilogue becomes one block and 230 VGPRs of accumulators are spilled.)
                        float vs_ = v[0];
#pragma unroll
                        for (int j = 1; j < NJ; ++j) vs_ += v[j];
                        if (!__builtin_isfinite(vs_) && p.range_flag) *p.range_flag = 1;
                    }
                }
                if (p.gn_part) {                                         // per (32-row chunk, column) partial sums of the final values
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float s_ = gs[j], q_ = gq[j];
                        s_ += __shfl_xor(s_, 16); q_ += __shfl_xor(q_, 16);
                        s_ += __shfl_xor(s_, 32); q_ += __shfl_xor(q_, 32);
                        if (kg_ == 0) {
                            float* o = p.gn_part + (long)eph * p.phase_gn + ((long)((em0 + wm * 128 + 32 * i2) >> 5) * p.N + en0 + wn * WNC + lr_ + 16 * j) * 2;
                            o[0] = s_; o[1] = q_;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (interior && p.ksplit > 1) {
            // split-K partial tile, interior: plain stores into slab eks (the general path spends ~8 instructions per element on tests)
            float* __restrict__ cb = p.splitk_ws + (long)eks * (long)p.M * p.N + (long)(em0 + wm * 128 + 4 * kg_) * p.N + en0 + wn * WNC + lr_;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* __restrict__ o = cb + (long)(16 * i + r) * p.N;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) o[16 * j] = acc[i][j][r];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float* __restrict__ Cg;
            const float* __restrict__ Rg = nullptr;
            if (p.ksplit > 1) Cg = p.splitk_ws + (long)eks * (long)p.M * p.N;
            else { Cg = p.C; Rg = p.res; }
            const int col0 = en0 + wn * WNC + lr_;
            float bv[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = (col0 + 16 * j < p.N && p.ksplit == 1 && p.bias) ? p.bias[col0 + 16 * j] : 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                float gs[NJ], gq[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = em0 + wm * 128 + 32 * i2 + 16 * ii + 4 * kg_ + r;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const int col = col0 + 16 * j;
                            if (row < p.M && col < p.N) {
                                const float a = acc[2 * i2 + ii][j][r];
                                if (p.ksplit > 1) Cg[(long)row * p.N + col] = a;
                                else {
                                    long addr;
                                    if (p.out_mode == OUT_UP2) {
                                        const int x = row & (p.Wo - 1);
                                        addr = (4L * row - 2 * x + eph_y * 2 * p.Wo + eph_x) * p.ldc + col;
                                    } else addr = (long)row * p.ldc + col;
                                    float v = a * alpha_ + bv[j];
                                    if (Rg) v += Rg[addr];
                                    if (p.accumulate) v += Cg[addr];
                                    Cg[addr] = v;
                                    if (!__builtin_isfinite(v) && p.range_flag) *p.range_flag = 1;
                                    if (p.C_hi) store_planes_cw(p, addr, v);
                                    gs[j] += v; gq[j] += v * v;
                                }
                            }
                        }
                    }
                }
                if (p.gn_part && p.ksplit == 1) {                        // per (32-row chunk, column) partial sums of the final values
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float s_ = gs[j], q_ = gq[j];
                        s_ += __shfl_xor(s_, 16); q_ += __shfl_xor(q_, 16);
                        s_ += __shfl_xor(s_, 32); q_ += __shfl_xor(q_, 32);
                        if (kg_ == 0 && col0 + 16 * j < p.N && em0 + wm * 128 + 32 * i2 < p.M) {      // chunks beyond the last row have no slot
                            float* o = p.gn_part + (long)eph * p.phase_gn + ((long)((em0 + wm * 128 + 32 * i2) >> 5) * p.N + col0 + 16 * j) * 2;
                            o[0] = s_; o[1] = q_;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the stores' address arithmetic from being hoisted in front of the first one (spills)
            }
        }
        t_epi += now() - t4_;
        if (!has_next) break;
        tile = next;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (stamps && blockIdx.x < 64 && lane == 0 && p.splitk_ws && p.ksplit == 1) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.splitk_ws) + (blockIdx.x * 4 + wave) * 8;
        o[0] = t_top; o[1] = t_vm; o[2] = t_bar; o[3] = t_epi; o[4] = now() - t_begin;
    }
}

template <bool BF, int NT, int NPL = 2, int NJ = 4, bool IO16 = false, bool PAIR = false>
int launch_convwin(const GemmParams& p, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&convwin_kernel<BF, NT, NPL, NJ, IO16, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    const long ntiles = (long)((p.M + CW_BM - 1) / CW_BM) * ((p.N + 32 * NJ - 1) / (32 * NJ)) * p.ksplit * (p.nphase > 1 ? p.nphase : 1);
    static const int cfg_persist = CDAE_DEV_INT("CDAE_CONVWIN_GRID", 512);      // persistent blocks: two per CU
    dim3 grid((unsigned)(ntiles < cfg_persist ? ntiles : cfg_persist));
    hipLaunchKernelGGL((convwin_kernel<BF, NT, NPL, NJ, IO16, PAIR>), grid, dim3(256), CW_LDS, st, p, (int)ntiles);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("convwin_kernel launch failed");
}

// [rows][taps][K] 16-bit planes (OHWI conv weights, or the [Cin][9][Cout] dgrad weights) -> [K / 16][taps][rows][16]
__global__ void wpack_kernel(const uint4* __restrict__ s_hi, const uint4* __restrict__ s_lo, uint4* __restrict__ d_hi, uint4* __restrict__ d_lo,
                             int rows, int taps, int K) {
    const long total = (long)rows * taps * (K >> 3);                 // 16-byte pieces per plane
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pc = (int)(i & 1);                                  // destination order: piece, row, tap, group
        long r = i >> 1;
        const int row = (int)(r % rows); r /= rows;
        const int t = (int)(r % taps);
        const int g = (int)(r / taps);
        const long src = ((long)row * taps + t) * (K >> 3) + g * 2 + pc;
        d_hi[i] = s_hi[src];
        d_lo[i] = s_lo[src];
    }
}

}  // namespace

extern "C" int cdae_conv_wpack(const unsigned short* w_hi, const unsigned short* w_lo, unsigned short* k_hi, unsigned short* k_lo, int rows, int taps,
                               int K, void* stream) {
    if (K % 16 || rows <= 0 || taps <= 0 || (((size_t)w_hi | (size_t)w_lo | (size_t)k_hi | (size_t)k_lo) & 15))
        return cdae_fail("conv_wpack: K % 16 == 0 and 16-byte aligned planes required");
    const long total = (long)rows * taps * (K >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(w_hi), reinterpret_cast<const uint4*>(w_lo),
                       reinterpret_cast<uint4*>(k_hi), reinterpret_cast<uint4*>(k_lo), rows, taps, K);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("wpack_kernel launch failed");
}

// Shapes this kernel takes (the dispatcher has already checked: stride 1, dense NHWC planes, row-major output).
bool cdae_convwin_ok(const GemmParams& p) {
    if (p.prec < 1 || p.prec > 4) return false;                           // 1 / 2: f16 / bf16 plane pairs; 3 / 4: one f16 / bf16 plane (3x3 only)
    if (p.prec > 2 && p.ps_taps == 4) return false;
    if (p.gn_coef || p.A2 || p.act != ACT_NONE) return false;
    // 16-bit rows (IO16 instantiation): interior tiles only — its edge path stores fp32 (a runtime choice of the element type there cost 350
    // spilled registers) — and both the result and the residual bf16
    if ((p.io16 & 3) && (p.M % CW_BM || p.N % CW_BN || p.prec != 4 || p.accumulate || p.C_hi || (p.io16 & 3) != ((p.res ? 2 : 0) | 1))) return false;
    // (channel-major bf16 tiles: 16-byte pieces of the result / residual rows, float4 pieces of the GroupNorm sums and the split-K slabs)
    if ((p.io16 & 3) && (p.ldc % 8 || (reinterpret_cast<size_t>(p.C) & 15) || (reinterpret_cast<size_t>(p.res) & 15) || (reinterpret_cast<size_t>(p.gn_part) & 15) ||
                         (reinterpret_cast<size_t>(p.bias) & 15) || p.ps_taps == 4 || p.nphase > 1 || (long)p.M * p.ldc >= (1L << 30) || (long)p.M * p.N >= (1L << 31))) return false;
    if (p.ps_taps == 4 ? (p.out_mode != OUT_UP2 || (p.prec != 1 && p.prec != 2) || p.Bk_hi) : p.out_mode != OUT_ROWMAJOR) return false;      // (4 taps, bf16: the stride-2 conv's dgrad as sub-pixel phases)
    if (p.W != 8 && p.W != 16 && p.W != 32 && p.W != 64) return false;            // tight window: tiles start on image-row boundaries
    if (p.Cin % 32 || p.ldb % 8 || p.sx % 8) return false;
    if ((long)p.M * p.sx * 2 >= (1L << 32) || (long)p.N * p.ldb * 2 >= (1L << 32)) return false;      // 32-bit byte offsets in the DMAs
    if (p.a_gm && (long)p.M * p.Cin * 2 >= (1L << 31)) return false;      // group-major planes: the group base g * M * 32 is a signed 32-bit scalar offset
    return true;
}

// The padding taps of convwin_kernel depend on a hardware behaviour no manual states: a DS read beyond every LDS allocation of the CU
// returns zeros (CW_OOB above).  If a part or a driver ever changed that, image borders would be silently wrong — so the library
// checks it itself, once per device, before the first launch: 512 blocks (two per CU, neighbours in LDS) fill their 80 KB with a
// non-zero pattern and read the addresses the kernel uses; any non-zero dword makes every window-conv launch on that device fail loudly.
__global__ __launch_bounds__(256, 2) void cw_oob_probe_kernel(unsigned* __restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < CW_LDS / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0xdead0000u + blockIdx.x;
    __syncthreads();
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10);          // let the CU's other block fill its allocation too
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        constexpr unsigned offs[6] = {0u, 0x1000u, 0x2000u, 0x3000u, 0x3E00u, 0x3FF0u};
        const unsigned a = CW_OOB + offs[q];
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        acc |= v[0] | v[1] | v[2] | v[3];
    }
    u32x4 c;                                                              // control: the block's own last 16 bytes hold the pattern
    const unsigned in_range = CW_LDS - 16;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(in_range) : "memory");
    if (acc != 0) atomicOr(bad, 1u);
    if (c[0] != 0xdead0000u + blockIdx.x) atomicOr(bad, 2u);
}

static int cw_oob_checked(hipStream_t st) {          // 1 = zeros confirmed, 0 = not checkable now (stream is capturing), -1 = failed
    // (the binding runs the probe when it first creates a workspace on a device — causaldiffae_amd/_lib.py — so launches normally find the
    //  answer here and never allocate or synchronise; the lazy path below remains for callers of the bare C-ABI)
    static std::atomic<int> state[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return cdae_fail("convwin: hipGetDevice failed"), -1;
    if (const int s = state[dev].load(std::memory_order_acquire)) return s;
    std::lock_guard<std::mutex> lk(mu);
    if (const int s = state[dev].load(std::memory_order_acquire)) return s;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;      // a warm-up launch outside the capture has normally decided already
    unsigned* d = nullptr;
    unsigned h = 0xffffffffu;
    bool ok = hipMalloc(&d, sizeof(unsigned)) == hipSuccess && hipMemsetAsync(d, 0, sizeof(unsigned), st) == hipSuccess &&
              hipFuncSetAttribute(reinterpret_cast<const void*>(&cw_oob_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(cw_oob_probe_kernel, dim3(512), dim3(256), CW_LDS, st, d);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&h, d, sizeof(unsigned), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (d) (void)hipFree(d);
    if (!ok) { cdae_fail("convwin: the out-of-range LDS read probe could not run"); return -1; }
    state[dev].store(h == 0 ? 1 : -1, std::memory_order_release);
    return state[dev].load();
}

extern "C" int cdae_convwin_lds_probe(void* stream) {
    const int r = cw_oob_checked((hipStream_t)stream);
    if (r < 0) return cdae_fail("convwin: out-of-range LDS reads do not return zeros on this device (or the in-range control failed): the window conv kernel's padding taps would be wrong");
    return r == 1 ? 0 : 2;
}

int cdae_convwin_launch(const GemmParams& p, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (cw_oob_checked(st) < 0)
        return cdae_fail("convwin: out-of-range LDS reads do not return zeros on this device (or the in-range control failed): the window conv kernel's padding taps would be wrong");
    if (p.ps_taps == 4) return p.prec == 2 ? launch_convwin<true, 4>(p, st) : launch_convwin<false, 4>(p, st);
    if (p.prec == 3) return launch_convwin<false, 9, 1>(p, st);          // mixed16: one f16 plane
    if (p.io16 & 3) {                                                     // the 16-bit torso: bf16 result (and residual) rows
        if (p.prec != 4 || (p.io16 & 3) != ((p.res ? 2 : 0) | 1)) return cdae_fail("convwin: 16-bit rows need one bf16 plane per operand, a bf16 result and (if any) a bf16 residual");
        // channel halves in the two plane slots (see PAIR): K-group-major weights, Cin a multiple of 64
        if (p.Bk_hi && p.Cin % 64 == 0 && !p.a_gm && cdae_tune(TUNE_CONVWIN_PAIR16)) {
            GemmParams q = p;
            q.A_lo = reinterpret_cast<const unsigned short*>(p.A) + p.Cin / 2;
            q.Bk_lo = p.Bk_hi + (long)(p.Cin / 32) * 9 * p.N * 16;
            q.Cin = p.Cin / 2;
            // (a K split chosen over the 32-channel chunks of the whole Cin stays as it is: splits beyond the halved chunk count write zero slabs)
            return launch_convwin<true, 9, 2, 4, true, true>(q, st);
        }
        return launch_convwin<true, 9, 1, 4, true>(p, st);
    }
    if (p.prec == 4) return launch_convwin<true, 9, 1>(p, st);           // mixed16, gradient operand: one bf16 plane
    if (p.cw_nj == 3 && p.prec == 1) return launch_convwin<false, 9, 2, 3>(p, st);      // planes.hip chose the tile width
    if (p.cw_nj == 2 && p.prec == 1) return launch_convwin<false, 9, 2, 2>(p, st);
    return p.prec == 2 ? launch_convwin<true, 9>(p, st) : launch_convwin<false, 9>(p, st);
}
