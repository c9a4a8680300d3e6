// Internal declarations shared by the HIP translation units of libcdae.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

enum { A_PLAIN_KC = 0, A_CONV_VEC = 1, A_CONV_GEN = 2, A_PLAIN_MC = 3 };
enum { B_PLAIN_KC = 0, B_PLAIN_MC = 1, B_CONV_MC = 2, B_WDGRAD_MC = 3 };
enum { OUT_ROWMAJOR = 0, OUT_NCHW = 1, OUT_UP2 = 2 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_LRELU = 2 };

struct GemmParams {
    const float* A; const float* B; float* C;
    const float* bias; const float* res;
    float* colsum_out;     // A_PLAIN_MC only: += column sums of A over k (conv / linear bias gradient), [M] floats
    int M, N, K;
    long lda, ldb, ldc;
    int batch, batch_inner;
    long a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    float alpha;
    // scale record {2^k, 2^-k} of the WEIGHT operand (cdae_weight_scales; device memory, nullptr: unscaled).  Pre-split weight planes
    // hold w * 2^k (elementwise.hip: keeps the lo plane of small weights a normal f16) and the fp32-operand kernels multiply the B tile by
    // [0] before they split it; every epilogue / split-K finish multiplies the sum by [1] on top of alpha.
    const float* w_scale;
    int act, out_mode, out_hw, accumulate;
    int amode, bmode;
    int a_scalar, b_scalar;   // operand not 16-byte vectorisable (odd K / pitch / alignment): element-wise loads
    // conv gather geometry (A_CONV_* rows / B_CONV_MC reduction pixels)
    int conv_M;            // number of pixels enumerated by the gather (N*Ho*Wo)
    int H, W, Cin;         // gathered tensor: spatial dims (before the fused 2x upsample) and channels
    int Ho, Wo;            // pixel grid the gather rows enumerate
    int stride, up, tconv;
    // dispatcher-computed gather constants: magic division by Ho*Wo and Wo, mode folding (igemm.hip make_pixrow/tap_offset)
    int hw, hw_shift, wo_shift; unsigned hw_magic, wo_magic;
    int g_mul, g_add, g_sign, g_pm, g_sh;
    long sn, sy, sx, sc;   // element strides of the gathered tensor
    // OHWI weight access for B_WDGRAD_MC
    int wCout, wCin, wflip;
    // split-K
    int ksplit, ksplit_auto, ksplit_force, force_tile, waves8;
    int grad_operand;       // one operand is a gradient tensor (range unsafe for f16): split mode uses bf16 planes
    int prec;               // -1: library default (env CDAE_IGEMM_PREC), 0: fp32 MFMA, 1: f16x3 split precision (K-contiguous operand pairs only)
    float* splitk_ws; size_t splitk_ws_bytes;
    // pre-split operands (ps_kernel): A / B point at the hi f16 planes, A_lo / B_lo at the lo planes (same strides, in elements)
    // A_PLAIN_KC with the K range split over two row-major sources: columns [0, K1) from A (pitch lda), [K1, K) from A2 (pitch
    // lda2) — a channel concatenation consumed in place.  A2 == nullptr: one source.  K1 % 32 == 0.
    const float* A2; long lda2; int K1;
    int presplit;
    // optional second output: the result also as f16 hi/lo planes with row pitch ldc (row-major outputs only), so a following
    // conv can take the pre-split path without a conversion pass
    unsigned short* C_hi; unsigned short* C_lo;
    // ps_kernel sub-pixel phase of a nearest-2x-upsample + conv3x3 (ps_taps == 4): 2x2 taps at offset (ph_y, ph_x), and
    // out_mode OUT_UP2 scatters GEMM row (n, y, x) to output pixel (n, 2y + ph_y, 2x + ph_x)
    int ps_taps, ph_y, ph_x;
    // ps_kernel / pswin_kernel epilogues: per (32-row chunk, column) partial (sum, sum of squares) of the FINAL output values,
    // [ceil(M/32)][N][2] floats — the GroupNorm statistics of the next layer without another pass over the tensor
    float* gn_part;
    // pswin_kernel with GNA: the A operand is the fp32 INPUT of a GroupNorm (+scale-shift, +SiLU) — A / A2 as for the two-source
    // GEMM (K1 = channels of the first source) — and gn_coef [N][Cin][2] holds that norm folded to y = silu?(x * a + b) per
    // (image, channel); the kernel normalises, activates and splits each window chunk on its way into LDS
    const float* gn_coef; int gn_silu;
    unsigned short* S_hi; unsigned short* S_lo;       // igemm_kernel GNS: side output, the GroupNorm (gn_coef, gn_silu) of the A operand as f16 planes [M][K]
    int dbg;                // dev ablations of ps_kernel (env CDAE_PS_DBG): 1 A rows -> one hot line, 2 B rows -> hot, 4 no loads after the first
    const unsigned short* A_lo; const unsigned short* B_lo;
    // conv3x3 weights additionally in K-GROUP-MAJOR order [K / 16][taps][rows][16] (cdae_conv_wpack), hi / lo planes: one (group, tap)
    // unit of an n-tile is one contiguous run, which is how convwin_kernel's weight DMA wants it (nullptr: OHWI planes only)
    const unsigned short* Bk_hi; const unsigned short* Bk_lo;
    // range guard: every contraction epilogue raises *range_flag when a final value is not finite.  An operand beyond the f16 range
    // (|x| >= 65520) becomes hi = inf, lo = -inf in the split and the products NaN, so this is where an overflow of the 16-bit
    // planes surfaces; the host reads the flag with cdae_range_status() (pinned host memory, written straight from the kernels)
    int* range_flag;
    // igemm_kernel GNS: the (a, b) coefficients of the (at most two) images a 128-row tile touches are preloaded into LDS behind the
    // tiles (16 bytes x K), set by the launcher when that fits; 0 = read them from global memory inside the loop
    int gn_tab;
    // convwin_kernel, sub-pixel up-conv: the four phases (ph_y, ph_x) of one conv as ONE launch.  nphase = 4: phase ph adds ph * phase_w
    // elements to the weight planes and ph * phase_gn floats to gn_part, (ph_y, ph_x) = (ph >> 1, ph & 1); 0 / 1: p.ph_y, p.ph_x as given
    int nphase; long phase_w, phase_gn;
    // convwin_kernel: the activation planes are GROUP-MAJOR, [Cin / 16][pixels][16] (written so by the GroupNorm apply kernel / the entry sweep
    // on request): a 16-channel half-window is then one contiguous run of 32-byte rows — 8 cache lines per DMA instead of 32
    int a_gm;
    // convwin_kernel: 16-column MFMA tiles per wave (n-tile = 32 cw_nj columns): 4 (256 x 128 tiles), or 3 / 2 where the narrower tile
    // fills the 512 block slots clearly better (planes.hip decides, before it sizes a K split)
    int cw_nj;
    // 16-bit torso (the single-plane modes only, prec 3 / 4): bit 0 — C (and the accumulate read) is a bf16 tensor, bit 1 — res is; bit 2 — the
    // A operand of an fp32-operand kernel is a bf16 tensor, bit 3 — the B operand is (igemm.hip k-major loaders).  Values are rounded to
    // bf16 BEFORE they enter GroupNorm partial sums, so the statistics describe the tensor that was stored.
    int io16;
};

int cdae_gemm_dispatch(GemmParams p, void* stream);

// A GROUP of fp32-operand GEMMs in ONE launch (igemm.hip, the linear / 1x1 weight gradients of a resolution level): `common` carries
// everything the members share (operand modes, precision, flags), an item what differs.  Block -> member by a scalar scan of first[].
// The whole argument travels as kernel arguments (no device memory for descriptors).
struct GemmGroupItem {
    const float* A; const float* B; float* C; float* colsum_out;
    int M, N, K, accumulate;
    long lda, ldb, ldc;
};
constexpr int GEMM_GROUP_MAX = 24;
struct GemmGroupArg {
    GemmParams common;
    int n;
    int first[GEMM_GROUP_MAX + 1];          // first block of member i; first[n] = grid size
    GemmGroupItem items[GEMM_GROUP_MAX];
};
// 0: launched; 1: not taken as a group (the caller launches the members one by one); -1: error
int cdae_gemm_group_dispatch(GemmGroupArg& g, void* stream);
#ifdef __HIPCC__
int cdae_splitk_finish(const GemmParams& p, bool gn_finish_ok, hipStream_t st);      // igemm.hip
#endif
// wg16.hip: weight gradient of a 1 x 1 conv / linear from bf16 rows (cdae_linear_wgrad_io dispatches)
int cdae_wg16_ok(const void* x, long ldx, const void* dy, long lddy, const float* dw, long lddw, int M, int N, int K, int io, size_t ws_bytes);
int cdae_wg16(const void* x, long ldx, const void* dy, long lddy, float* dw, long lddw, float* dbias, int M, int N, int K, int accumulate, float* ws,
              size_t ws_bytes, void* stream);
// convwin.hip: second-generation window-resident conv3x3 on pre-split planes
bool cdae_convwin_ok(const GemmParams& p);
int cdae_convwin_launch(const GemmParams& p, void* stream);
// planes.hip: dispatch of a contraction on pre-split planes (convwin / pswin / ps kernels); 0 ok, -1 error, 2 / 3 see planes.hip
int cdae_planes_dispatch(GemmParams& p, int big, int& ks, hipStream_t st);

// Developer switches.  The shipped library reads NO environment variable for its dispatch (only CDAE_IGEMM_PREC, the documented default
// precision, and CDAE_PROF_DUMP, the profiler's per-launch dump file): CDAE_DEV_INT(name, default) is the constant `default` unless the
// library is built with -DCW_DEV=1 (EXTRA_HIPCC_FLAGS), in which case it reads the environment once — ablation / sweep builds only.
#ifndef CW_DEV
#define CW_DEV 0
#endif
#if CW_DEV
#include <stdlib.h>
#define CDAE_DEV_INT(NAME, DFLT) ([]() -> int { static const int v_ = getenv(NAME) ? atoi(getenv(NAME)) : (int)(DFLT); return v_; }())
#else
#define CDAE_DEV_INT(NAME, DFLT) (DFLT)
#endif

// run-time dispatch thresholds (cdae_tune_set / cdae_tune_get in include/cdae.h; prof.hip holds the values)
enum { TUNE_CONVWIN_MIN_TILES = 0, TUNE_CONVWIN_SPLITK = 1, TUNE_CONVWIN_NJ3 = 2, TUNE_HEAD_MFMA = 3, TUNE_ROWS16_MIN_M = 4, TUNE_ROWS16_RING = 5, TUNE_CONVWIN_PAIR16 = 6, TUNE_GN_BWD_FOLD2 = 7, TUNE_GROUP_BIG_TILES = 8, TUNE_CONVWIN_NJ2 = 9, TUNE_WGWIN_DIST = 10, TUNE_WGWIN_SWZ = 11, TUNE_WG16_SLOTS = 12, TUNE_WGWIN_FIXED = 13, TUNE_WGWIN_CO2 = 14, TUNE_N = 15 };
int cdae_tune(int key);

// rows16.hip: the streaming GEMM of the 16-bit torso's 1 x 1 convs / linears (cdae_gemm16_ps dispatches)
int cdae_rows16_ok(const void* a16, long lda, const void* b16, long ldb, const float* bias, const void* res, const void* c, long ldc, const float* gn_part,
                   int M, int N, int K, int io, int accumulate);
int cdae_rows16_gemm(const void* a16, long lda, const void* b16, long ldb, const float* bias, const void* res, void* c, long ldc, int M, int N, int K, int io,
                     void* stream);

// error reporting: sets the thread-local message returned by cdae_last_error(), returns -1
int cdae_fail(const char* msg);

// lightweight per-family profiling (HIP events on the launch stream), see prof.hip
enum { PROF_IGEMM = 0, PROF_GN = 1, PROF_SOFTMAX = 2, PROF_ELEMWISE = 3, PROF_OPT = 4, PROF_CONVWIN = 5, PROF_CONVWIN_DGRAD = 6, PROF_CONVWIN_UP = 7, PROF_NFAM = 8 };
void cdae_prof_begin(int family, double work, hipStream_t st);
void cdae_prof_end(int family, hipStream_t st);
void cdae_prof_note(int family, double bytes);
void cdae_prof_tag(const char* tag);
bool cdae_prof_on();
int* cdae_range_flag_ptr();      // device-visible int[1], see GemmParams::range_flag

// LDS-DMA (global_load_lds_dwordx4) issued through inline assembly: lane l's 16 bytes at `src` land at LDS byte address
// lds_dst + 16 l (lds_dst wave-uniform).  Why not __builtin_amdgcn_global_load_lds: hipcc's waitcnt pass keeps the ADDRESS
// registers of an in-flight LDS-DMA on its scoreboard and puts `s_waitcnt vmcnt(0)` in front of the next instruction that
// redefines one of them — in a software-pipelined loop that is a full drain of the DMAs just issued, in the middle of the
// MFMA stream (round-2 finding; visible in the ISA of every round-1 plane kernel).  Opaque asm leaves all waiting to the
// kernel's own counted s_waitcnt.
#ifdef __HIPCC__
// sigmoid / SiLU (reference nn.py:13-15 SiLU = x * sigmoid(x)) on the two transcendental instructions: 1 / (1 + 2^t), t = -x log2(e),
// with the rounding error of the product t carried into the result ((1 + r ln 2) correction), so the value is within ~2 ulp of the
// exact one.  8 VALU instructions instead of the ~25 of `1 / (1 + expf(-x))` (range reduction, denormal scaling and an IEEE divide the
// sigmoid does not need) — the normalisation kernels are VALU-bound on that sequence.  EVERY kernel uses this one definition, so the
// activation planes of the different producers stay bit-identical.
__device__ __forceinline__ float cdae_sigmoid(float x) {
    const float c_hi = -1.44269502162933349609375f, c_lo = -1.925963033500011e-8f;        // -log2(e) = c_hi + c_lo
    // outside [-87, 100] the sigmoid is 0 / 1 to fp32 anyway; clamped there, 2^t stays finite (t <= 125.5) and a huge |x| cannot turn
    // t or its residual into inf - inf
    const float xc = __builtin_amdgcn_fmed3f(x, -87.f, 100.f);
    const float t = xc * c_hi;
    const float r = __builtin_fmaf(xc, c_lo, __builtin_fmaf(xc, c_hi, -t));               // exact product minus t
    const float e = __builtin_amdgcn_exp2f(t);
    const float ec = __builtin_fmaf(e * 0.693147182464599609375f, r, e);                 // 2^(t + r)
    return __builtin_amdgcn_rcpf(1.f + ec);
}
__device__ __forceinline__ float cdae_silu(float x) {
    float y = x * cdae_sigmoid(x);
    asm("" : "+v"(y));              // the rounded product, never contracted into a following add (bit-identical across translation units)
    return y;
}

// result / residual element access of the epilogues that serve the 16-bit torso (GemmParams::io16)
__device__ __forceinline__ float cdae_round_c(const GemmParams& p, float v) { return (p.io16 & 1) ? (float)(__bf16)v : v; }
__device__ __forceinline__ void cdae_store_c(const GemmParams& p, float* C, long addr, float v) {
    if (p.io16 & 1) reinterpret_cast<__bf16*>(C)[addr] = (__bf16)v; else C[addr] = v;
}
__device__ __forceinline__ float cdae_load_c(const GemmParams& p, const float* C, long addr) {
    return (p.io16 & 1) ? (float)reinterpret_cast<const __bf16*>(C)[addr] : C[addr];
}
__device__ __forceinline__ float cdae_load_res(const GemmParams& p, const float* R, long addr) {
    return (p.io16 & 2) ? (float)reinterpret_cast<const __bf16*>(R)[addr] : R[addr];
}

__device__ __forceinline__ void cdae_lds_dma16(const void* src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_dst), "v"(src) : "memory", "m0");
}
#endif
