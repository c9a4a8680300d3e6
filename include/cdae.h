/* cdae.h — C-ABI of libcdae.so: the MI355X (gfx950) kernels behind the CausalDiffAE diffusion hot path.
 *
 * The reference (Akomand/CausalDiffAE) has no FFI / plugin interface: its hot path is pure Python on
 * ATen.  These entry points are what a binding for that path calls instead of the ATen
 * ops; each one cites the reference lines it replaces.  Conventions:
 *   - raw device pointers, fp32, activations NHWC ("channels_last") unless stated, weights of a 3x3
 *     conv in OHWI order ([Cout][ky][kx][Cin] = the channels_last storage of the reference's
 *     [Cout,Cin,3,3] parameter), linear / 1x1 weights [out][in];
 *   - no allocation, no host sync; every kernel is enqueued on `stream` (a hipStream_t; 0 = null stream);
 *     workspaces are caller-provided, sizes from the cdae_*_workspace_* queries;
 *   - return 0 on success, -1 on error (message via cdae_last_error(), thread-local);
 *   - re-entrant; the only global state is the opt-in profiler (cdae_prof_*).
 */
#ifndef CDAE_H
#define CDAE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDAE_VERSION 1
#define CDAE_GN_MAX_CHUNKS 64
#define CDAE_BN_MAX_CHUNKS 256
#define CDAE_PROF_FAMILIES 8      /* 0 igemm (every contraction but 5-7), 1 groupnorm, 2 softmax, 3 elementwise, 4 optimizer, 5 convwin (convwin_kernel<f16, 9 taps>: the forward
                                   * stride-1 conv3x3), 6 convwin_dgrad (convwin_kernel<bf16, 9 taps>), 7 convwin_up (convwin_kernel<f16, 4 taps>: sub-pixel up-conv phases) */

/* rows of the fp32 coefficient table passed to the sampler kernels: tab[row * T + t]
 * (float32 roundings of the float64 tables of gaussian_diffusion.py:137-179, i.e. what
 * _extract_into_tensor's `.float()` yields, gaussian_diffusion.py:948) */
enum {
    CDAE_TAB_SQRT_AC = 0,          /* sqrt_alphas_cumprod */
    CDAE_TAB_SQRT_1MAC = 1,        /* sqrt_one_minus_alphas_cumprod */
    CDAE_TAB_SQRT_RECIP_AC = 2,    /* sqrt_recip_alphas_cumprod */
    CDAE_TAB_SQRT_RECIPM1_AC = 3,  /* sqrt_recipm1_alphas_cumprod */
    CDAE_TAB_AC = 4,               /* alphas_cumprod */
    CDAE_TAB_AC_PREV = 5,          /* alphas_cumprod_prev */
    CDAE_TAB_POST_COEF1 = 6,       /* posterior_mean_coef1 */
    CDAE_TAB_POST_COEF2 = 7,       /* posterior_mean_coef2 */
    CDAE_TAB_MODEL_LOGVAR = 8,     /* model log-variance (FIXED_LARGE / FIXED_SMALL), gaussian_diffusion.py:305-318 */
    CDAE_TAB_MODEL_VAR = 9,
    CDAE_TAB_POST_LOGVAR_CLIPPED = 10, /* posterior_log_variance_clipped (LEARNED_RANGE lower end; true log-variance of the bound) */
    CDAE_TAB_LOG_BETAS = 11,           /* log(betas) (LEARNED_RANGE upper end) */
    CDAE_TAB_ROWS = 12
};

int cdae_version(void);
const char* cdae_last_error(void);

/* Scratch a caller must provide (the library never allocates).  op selects the family, dims its problem size:
 *   CDAE_WS_SPLITK      dims = {M, N, K} of the contraction (conv3x3: M = N*Ho*Wo, N = Cout, K = 9*Cin; wgrad: M = Cout, N = 9*Cin,
 *                       K = pixels): bytes of split-K slabs that let every dispatcher heuristic take its first choice.  A smaller
 *                       (or NULL) workspace is legal — the dispatcher then splits less.  dims == NULL: the size that covers every
 *                       layer of the reference's UNets up to batch 256 (256 MiB).
 *   CDAE_WS_GROUPNORM   dims = {N, C}: partial sums of cdae_gn_* (bytes of cdae_gn_workspace_floats(N, C) floats)
 *   CDAE_WS_GN_PARTS    dims = {N, C}: per-channel sums of cdae_gn_stats_from_parts (16 * N * C bytes)
 *   CDAE_WS_BATCHNORM   dims = {C}:   partial sums of cdae_bn_* (bytes of cdae_bn_workspace_floats(C) floats)
 * Returns 0 for an unknown op or missing dims. */
enum { CDAE_WS_SPLITK = 0, CDAE_WS_GROUPNORM = 1, CDAE_WS_GN_PARTS = 2, CDAE_WS_BATCHNORM = 3 };
size_t cdae_workspace_bytes(int op, const long* dims, int ndims);

/* Range guard of the split-precision modes.  f16 planes hold |x| < 65520 only; a larger operand becomes inf in the split and
 * NaN in the products.  Every contraction epilogue raises a flag when a FINAL value is not finite; cdae_range_status reports and
 * clears it (synchronise the stream first).  The Python API raises CdaeRangeError instead of returning such tensors and points at
 * the IEEE `fp32` mode, which has fp32's range.  Small magnitudes: below 2^-3 the lo plane is an f16 subnormal, i.e. the pair is
 * fixed-point with an LSB of 2^-24 (absolute error <= 2^-25 per operand element) instead of 2^-22 relative. */
int cdae_range_status(int* nonfinite);

/* Arithmetic of the dense contractions whose operands are both K-contiguous (conv3x3 / linear / 1x1 forward, QK^T).
 * Inputs, outputs and accumulation are fp32 in both modes.
 *   CDAE_PREC_FP32  : v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain.
 *   CDAE_PREC_F16X3 : every fp32 operand is split into two f16 planes (hi + lo, 22 significand bits) and each product is
 *                     hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (3 MFMAs at 16x the fp32 rate; the dropped lo*lo
 *                     term is 2^-22 relative).  Requires |operand| < 65504 (true for this network: the reference ships
 *                     an fp16 mode).  Default; also selectable with env CDAE_IGEMM_PREC=0|1|2 before the first call.
 *   CDAE_PREC_MIXED16 : reduced-precision torso — one f16 plane per operand (one bf16 plane when an operand is a
 *                     gradient), fp32 accumulation; GroupNorm / softmax / embeddings / optimizer stay fp32.  The
 *                     counterpart of the reference's use_fp16 torso (unet.py:501-507) and of BASELINE's bf16 training
 *                     config; NOT within the 1e-4 parity bar (loss agreement ~1e-2). */
#define CDAE_PREC_FP32 0
#define CDAE_PREC_F16X3 1
#define CDAE_PREC_MIXED16 2
int cdae_set_default_precision(int prec);
int cdae_get_default_precision(void);

/* ---- dense contractions on the matrix cores (igemm.hip) -------------------------------------------------- */

/* conv3x3, pad 1 — replaces nn.Conv2d in ResBlock.in_layers[2]/out_layers[3] (unet.py:143-162), the stem and
 * output head (unet.py:392,498), Downsample.op (stride 2, unet.py:99) and Upsample (F.interpolate nearest 2x
 * fused into the gather, unet.py:76-78), encoder convs (nn.py:49-50).
 *   x   : [N,H,W,Cin] with element strides (sn,sy,sx,sc); sc==1 && Cin%32==0 takes the vector path
 *   w   : OHWI [Cout][3][3][Cin];  bias [Cout] or NULL;  res: optional residual, same layout as out
 *   out : [N,Ho,Wo,Cout] row pitch ldo (out_nchw=0) or NCHW contiguous (out_nchw=1)
 *   Ho = up ? 2H : (H-1)/stride+1 (same for W).  splitk_ws may be NULL (disables split-K). */
int cdae_conv3x3_fwd(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* w_scale, const float* bias, const float* res,
                     float* out, long ldo, int out_nchw, int N, int H, int W, int Cin, int Cout, int stride, int up,
                     float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* dgrad: dx[N,H,W,Cin] (pitch lddx) from dy[N,Ho,Wo,Cout] (pitch lddy).  For up=1 dx is the gradient w.r.t.
 * the UPSAMPLED input [N,2H,2W,Cin]; follow with cdae_sumpool2. accumulate: dx += . */
/* Pre-split operand path (inference): the same f16x3 products on operands already stored as two f16 planes (hi = f16(x),
   lo = f16(x - hi)) — activations by cdae_gn_apply_split, weights by cdae_split_f16 once per weight version.  Tiles go
   global -> LDS by LDS-DMA with no conversion work in the main loop; results are bit-identical to cdae_conv3x3_fwd /
   cdae_linear_fwd in f16x3 mode.  Strides in elements of a plane; Cin (K) % 32 == 0, pixel pitch % 8 == 0. */
int cdae_conv3x3_fwd_ps(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, const unsigned short* w_hi,
                        const unsigned short* w_lo, const float* w_scale, const float* bias, const float* res, float* out, long ldo, int out_nchw,
                        unsigned short* out_hi, unsigned short* out_lo,
                        float* gn_part /* optional [ceil(M/32)][Cout][2]: per (32-pixel chunk, channel) sum and sum of squares of the
                                          result, consumed by cdae_gn_stats_from_parts; disables split-K */,
                        int N, int H, int W,
                        int Cin, int Cout, int stride, int up, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* Second-generation window kernel (convwin.hip; stride-1 3x3 convs on rows of 8..64 pixels): it wants the weights ALSO in
   K-group-major order [K / 16][taps][rows][16] (one (16-channel group, tap) unit of an n-tile = one contiguous run for its LDS-DMA).
   cdae_conv_wpack: [rows][taps][K] planes -> that order (rows = Cout, K = Cin for the forward; rows = Cin, K = Cout for the dgrad
   weights of cdae_wdgrad_planes).  cdae_conv3x3_fwd_psk / cdae_conv3x3_dgrad_psk = the _ps entry points with the packed planes
   passed along (NULL: identical to _ps; the library falls back to the first-generation kernels).
   Hardware dependency of that kernel: the zero padding of a tap that falls outside the image is an LDS read at an address beyond the
   block's allocation, which gfx950 returns as zeros (no fault, no wrap) — measured, not documented.  The library CHECKS it itself,
   once per device, before the kernel's first launch (a 512-block probe that fills both LDS allocations of every CU and reads the
   addresses the kernel uses): on a part that answers anything else every window-conv launch FAILS with a message instead of
   computing wrong borders; such a part takes the first-generation kernels with cdae_tune_set(CDAE_TUNE_CONVWIN_MIN_TILES, 1 << 30).
   cdae_convwin_lds_probe runs (or re-reads) that check: 0 = zeros confirmed, 2 = not checkable now (the stream is capturing),
   1 = the device fails it.  tests/test_gpu_kernels.py test_lds_out_of_range_reads_return_zero covers the stand-alone probe too. */
int cdae_convwin_lds_probe(void* stream);
int cdae_conv_wpack(const unsigned short* w_hi, const unsigned short* w_lo, unsigned short* k_hi, unsigned short* k_lo, int rows, int taps,
                    int K, void* stream);
int cdae_conv3x3_fwd_psk(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, const unsigned short* w_hi,
                         const unsigned short* w_lo, const unsigned short* wk_hi, const unsigned short* wk_lo, const float* w_scale, const float* bias, const float* res,
                         float* out, long ldo, int out_nchw, unsigned short* out_hi, unsigned short* out_lo, float* gn_part, int N, int H, int W,
                         int Cin, int Cout, int stride, int up, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_conv3x3_dgrad_psk(const unsigned short* dy_hi, const unsigned short* dy_lo, const unsigned short* wt_hi, const unsigned short* wt_lo,
                           const unsigned short* wtk_hi, const unsigned short* wtk_lo, float* dx, long lddx, int N, int H, int W, int Cin, int Cout,
                           float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* GroupNorm mean / rstd of a tensor (or of the channel concatenation of two) from the partial sums its producing conv(s) left
   behind — replaces the statistics pass of cdae_gn_stats(2).  HW % 32 == 0. */
int cdae_gn_stats_from_parts(const float* part1, int C1, int nseg1 /* 1, or 4 for a sub-pixel up-conv result */, const float* part2, int C2,
                             int nseg2, int N, int HW, int groups, float eps, float* mean, float* rstd,
                             float* ws /* N * (C1 + C2) * 4 floats, 8-byte aligned */, void* stream);
int cdae_gn_coef(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef,
                 int N, int C, int groups, void* stream);
/* The same statistics with cdae_gn_coef's table [N][C][2] written by the statistics kernel itself (coef != NULL: gamma, beta and the
   optional scale-shift rows are then required; 32 groups / at most 32 channels per group): one launch less per GroupNorm whose
   consumer streams the fp32 rows (cdae_skip_gn_fwd, cdae_linear_fwd_stream_gn, cdae_head_conv).  Bit-identical to cdae_gn_coef. */
int cdae_gn_stats_from_parts_coef(const float* part1, int C1, int nseg1, const float* part2, int C2, int nseg2, int N, int HW, int groups, float eps,
                                  float* mean, float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef,
                                  float* ws, void* stream);
int cdae_gn_stats2_coef(const float* x1, int ld1, const float* x2, int ld2, int C1, int N, int HW, int C, int groups, float eps, float* mean,
                        float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef, float* ws, void* stream);
/* nearest-2x upsample + conv3x3 (unet.py:67-76) as four 2x2 sub-pixel convolutions of the low-resolution input: 2.25x fewer
   multiply-adds than convolving the upsampled image.  w4 = [4][Cout][2][2][Cin] folded weights as hi / lo planes. */
int cdae_upconv3x3_fwd_ps(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, const unsigned short* w4_hi,
                          const unsigned short* w4_lo, const float* w_scale, const float* bias, float* out, long ldo,
                          float* gn_part /* optional [4][N*H*W/32][Cout][2] partial sums, one segment per phase */,
                          int N, int H, int W, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_linear_fwd_ps(const unsigned short* x_hi, const unsigned short* x_lo, long ldx, const unsigned short* w_hi, const unsigned short* w_lo,
                       long ldw, const float* w_scale, const float* bias, const float* res, float* y, long ldy, int M, int N, int K, float alpha, int act,
                       float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_split_f16(const float* src, unsigned short* hi, unsigned short* lo, long n, void* stream);
/* WEIGHT operands of the f16 modes are scaled by an exact power of two per tensor.  hi = f16(w), lo = f16(w - hi) degenerates for
   |w| < 2^-3: lo is then an f16 subnormal and the pair is fixed point with a 2^-24 LSB (|w| ~ 0.01: 1e-5 relative) — and conv / linear
   weights live there.  A scale record is two floats {2^k, 2^-k}, k = 14 - floor(log2 max|w|) (scaled maximum in [2^14, 2^15); {1, 1} for an
   all-zero or non-finite tensor): weight planes hold w * 2^k (cdae_split_f16w, cdae_wprep_all_k), the fp32-operand kernels multiply the
   weight tile by 2^k before they split it, and every epilogue multiplies the sum by the exact 2^-k.  The pair then represents every
   element to 2^-22 relative or 2^-39 of the tensor's largest magnitude, whichever is larger, for ANY weight scale.  Every entry point
   below that takes a weight takes the record as `w_scale` (device pointer; NULL = the operand is unscaled).  bf16 planes (gradient
   operands, dgrad weights) have fp32's exponent range and are never scaled.
   cdae_weight_scales: records of MANY tensors of one fp32 buffer in one pass.  desc = nw records {long offset, long n, int chunk0, int
   pad} (elements; chunk0 = running sum of ceil(n / cdae_weight_scales_chunk())), total_chunks = that sum; records [nw][2] floats,
   scratch nw uint32.  cdae_weight_scale1: one tensor. */
int cdae_weight_scales(const float* flat, const void* desc, int nw, int total_chunks, float* records, unsigned* scratch, void* stream);
int cdae_weight_scales_chunk(void);
int cdae_weight_scale1(const float* w, long n, float* record, unsigned* scratch, void* stream);
int cdae_split_f16w(const float* src, const float* w_scale, unsigned short* hi, unsigned short* lo, long n, void* stream);
/* y = [x1 | x2] @ w^T + bias with the K range split over two row-major sources (the 1x1 skip conv of a ResBlock fed by the
   skip concatenation, unet.py:629,198); K1 % 32 == 0 */
int cdae_linear_fwd_cat(const float* x1, long ld1, int K1, const float* x2, long ld2, const float* w, long ldw, const float* w_scale, const float* bias, float* y,
                        long ldy, int M, int N, int K, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* cdae_linear_fwd_cat + the block's first GroupNorm in the same sweep: besides y, the n-tile-0 blocks write
   silu?(x * a + b) (coef [N][K][2] from cdae_gn_coef: bit-identical to cdae_gn_apply_split2) as f16 hi/lo planes s_hi / s_lo [M][K],
   i.e. the operand of the block's first conv3x3 — ResBlock skip_connection (unet.py:171,198) and in_layers[0:2] (unet.py:135-137,187)
   over ONE read of the (concatenated) block input.  HW = pixels per image; x2 may be NULL.  f16x3 mode, M >= 96, N >= 96. */
int cdae_linear_fwd_cat_gn(const float* x1, long ld1, int K1, const float* x2, long ld2, const float* w, long ldw, const float* w_scale, const float* bias, float* y,
                           long ldy, const float* coef, int silu, unsigned short* s_hi, unsigned short* s_lo, int M, int N, int K, int HW,
                           float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* The same sweep as its own HBM-stream kernel (skipgn.hip), on PRE-SPLIT weight planes w_hi / w_lo [N][K] (cdae_split_f16 of the 1x1
   weight, row pitch ldw elements): x is read once through registers, split raw into LDS for the f16x3 GEMM and normalised into the
   planes; bit-identical planes, y equal to cdae_linear_fwd_cat_gn's up to fp32 summation order.  cdae_skip_gn_ok: 1 when the shape
   is supported (K, K1 % 32 == 0, M % HW == 0, the (a, b) table of the images a 128-row tile touches <= 16 KB). */
int cdae_skip_gn_ok(int M, int N, int K, int K1, int HW);
/* the same kernel without the GroupNorm side output: y = [x1 | x2] @ W^T + bias (+ res, row pitch ldres) for large row counts — the
   attention proj_out with its residual (unet.py:231) and the other 1x1 convs / linears on fp32 rows (f16x3 products, K % 32 == 0) */
int cdae_linear_fwd_stream(const float* x1, long ld1, int K1, const float* x2, long ld2, const unsigned short* w_hi, const unsigned short* w_lo,
                           long ldw, const float* w_scale, const float* bias, const float* res, long ldres, float* y, long ldy, unsigned short* c_hi, unsigned short* c_lo,
                           int M, int N, int K, void* stream);      /* c_hi / c_lo (may be NULL): y also as f16 planes, row pitch ldy */
/* the same + gn_part (may be NULL; M % 32 == 0): [M / 32][N][2] per (32-row chunk, column) sum / sum of squares of y for cdae_gn_stats_from_parts */
int cdae_linear_fwd_stream_gn_part(const float* x1, long ld1, int K1, const float* x2, long ld2, const unsigned short* w_hi, const unsigned short* w_lo,
                                   long ldw, const float* w_scale, const float* bias, const float* res, long ldres, float* y, long ldy, unsigned short* c_hi,
                                   unsigned short* c_lo, float* gn_part, int M, int N, int K, void* stream);
/* y[M][N] = silu?(GroupNorm(x))[M][K] @ W^T + bias with the GroupNorm folded to per-(image, channel) coefficients (cdae_gn_coef) and
   applied to the fp32 rows as they are staged — GroupNorm -> 1x1 conv in one pass (AttentionBlock norm -> qkv, unet.py:213-228).
   f16x3 products on pre-split weight planes; shapes as cdae_skip_gn_ok(M, N, K, K, HW). */
int cdae_linear_fwd_stream_gn(const float* x, long ldx, const unsigned short* w_hi, const unsigned short* w_lo, long ldw, const float* w_scale, const float* bias,
                              float* y, long ldy, const float* coef, int silu, int M, int N, int K, int HW, void* stream);
int cdae_skip_gn_fwd(const float* x1, long ld1, int K1, const float* x2, long ld2, const unsigned short* w_hi, const unsigned short* w_lo, long ldw,
                     const float* w_scale, const float* bias, float* y, long ldy, const float* coef, int silu, unsigned short* s_hi, unsigned short* s_lo, int planes_gm,
                     int M, int N, int K, int HW, void* stream);
/* two-source forms: channels [0, C1) are read from x1 (pixel pitch ld1), channels [C1, C) from x2 (pitch ld2) — the skip
   concatenation th.cat([h, hs.pop()], dim=1) (unet.py:629) consumed in place instead of being copied; x2 == NULL: one source */
int cdae_gn_stats2(const float* x1, int ld1, const float* x2, int ld2, int C1, int N, int HW, int C, int groups, float eps, float* mean,
                   float* rstd, float* ws, void* stream);
int cdae_gn_apply_split2(const float* x, int ldx, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo, int N, int HW,
                         int C, int ldy, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         const float* scale_shift, int ld_ss, int silu, void* stream);
/* GROUP-MAJOR activation planes, [C / 16][N H W][16] instead of [N H W][C]: every 16-channel half-window of the window conv kernel is then
   one contiguous run (8 cache lines per DMA instead of 32).  cdae_gn_apply_split2g writes them (also cdae_skip_gn_fwd with planes_gm = 1),
   cdae_conv3x3_fwd_psg reads them with x_gm = 1 — or returns 3, with no error set, when that shape does not run on the window kernel: the
   caller then converts with cdae_planes_gm_to_pc and calls again with x_gm = 0. */
int cdae_gn_apply_split2g(const float* x, int ldx, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo, int N, int HW,
                          int C, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* scale_shift, int ld_ss, int silu, void* stream);
int cdae_planes_gm_to_pc(const unsigned short* a_hi, const unsigned short* a_lo, unsigned short* o_hi, unsigned short* o_lo, long P, int C, void* stream);
int cdae_conv3x3_fwd_psg(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, int x_gm, const unsigned short* w_hi,
                         const unsigned short* w_lo, const unsigned short* wk_hi, const unsigned short* wk_lo, const float* w_scale, const float* bias, const float* res,
                         float* out, long ldo, int out_nchw, unsigned short* out_hi, unsigned short* out_lo, float* gn_part, int N, int H, int W,
                         int Cin, int Cout, int stride, int up, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_gn_apply_split(const float* x, unsigned short* y_hi, unsigned short* y_lo, int N, int HW, int C, int ldx, int ldy, int groups,
                        const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss,
                        int silu, void* stream);
int cdae_conv3x3_dgrad(const float* dy, long lddy, const float* w, float* dx, long lddx, int N, int H, int W, int Cin, int Cout,
                       int stride, int up, int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* Training on the pre-split kernels (stride-1 3x3 convs; same reference lines as cdae_conv3x3_fwd — the backward the reference
   gets from autograd through F.conv2d, nn.py:470-480 / unet.py:187-197).  The GroupNorm in front of the conv writes its result
   as f16 planes (forward operand, 2^-22) AND as bf16 planes (kept for the backward); the gradient is split into bf16 planes
   (full fp32 range, 2^-16) once and feeds both dgrad and wgrad.
   cdae_gn_apply_split_train: cdae_gn_apply_split plus the bf16 plane pair yb_hi / yb_lo (same pitch ldy).
   cdae_split_bf16: hi = bf16(x), lo = bf16(x - hi).
   cdae_wdgrad_planes: w OHWI [Cout][9][Cin] -> bf16 planes of the dgrad weight [Cin][9][Cout], taps flipped.
   cdae_conv3x3_dgrad_ps: dx[N,H,W,Cin] (row pitch lddx) = conv3x3(dy planes [N,H,W,Cout], wt planes) on the window kernel.
   cdae_conv3x3_wgrad_win: dw (OHWI) (+)= dy^T * im2col(a), a / dy as dense NHWC bf16 planes; dbias (+)= column sums of dy (may be
   NULL).  The activation window stays in an LDS ring (each pixel row fetched once per block, not once per tap), fragments by
   transpose reads, split-K over pixel ranges with a fixed-order reduction (wgrad.hip).  _supported: W a power of two in [8, 64],
   H*W % 64 == 0, Cin % 64 == 0, Cout % 64 == 0. */
/* two-source forms for the skip concatenation in training (channels [0, C1) from x1, [C1, C) from x2; no concatenated tensor exists):
   cdae_gn_apply_split_train2 = cdae_gn_apply_split_train over the two sources; cdae_gn_bwd_cat = cdae_gn_bwd reading x from the two
   sources and writing (or accumulating onto) dx1 [pixels][C1] and dx2 [pixels][C - C1]. */
int cdae_gn_apply_split_train2(const float* x1, int ld1, const float* x2, int ld2, int C1, unsigned short* y_hi, unsigned short* y_lo,
                               unsigned short* yb_hi, unsigned short* yb_lo, int N, int HW, int C, int ldy, int groups, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream);
int cdae_gn_bwd_cat(const float* x1, int ld1, const float* x2, int ld2, int C1, const float* dy, int lddy, float* dx1, int lddx1, float* dx2, int lddx2,
                    int N, int HW, int C, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* scale_shift, int ld_ss, int silu, float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift,
                    int ld_dss, int accumulate_dx, float* ws, void* stream);
int cdae_gn_apply_split_train(const float* x, unsigned short* y_hi, unsigned short* y_lo, unsigned short* yb_hi, unsigned short* yb_lo, int N,
                              int HW, int C, int ldx, int ldy, int groups, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream);
/* nearest-2x upsample (Upsample, unet.py:86-104: F.interpolate(scale_factor=2, mode="nearest") before the conv) of x[N,H,W,C] written
   as the f16 (f_*) and bf16 (b_*) operand planes [N,2H,2W,C] of the training convs */
int cdae_upsample2_split(const float* x, unsigned short* f_hi, unsigned short* f_lo, unsigned short* b_hi, unsigned short* b_lo, int N, int H,
                         int W, int C, void* stream);
int cdae_split_bf16(const float* src, unsigned short* hi, unsigned short* lo, long n, void* stream);
int cdae_wdgrad_planes(const float* w, unsigned short* hi, unsigned short* lo, int Cout, int Cin, void* stream);
/* dgrad of a STRIDE-2 conv3x3 (Downsample, unet.py:82-105) as four 2x2 sub-pixel convolutions of dy on the plane kernels (2.25x fewer
   multiply-adds than the masked 9-tap gather of cdae_conv3x3_dgrad): cdae_s2dgrad_wfold writes the folded bf16 hi / lo weight planes
   [4 phases][Cin][2][2][Cout] of an OHWI weight; cdae_conv3x3_s2_dgrad_ps takes dy as bf16 planes [N, Ho, Wo, Cout] and writes
   dx [N, 2 Ho, 2 Wo, Cin] (rows of pitch lddx).  Returns 2 (no error set) for a shape it does not take. */
int cdae_s2dgrad_wfold(const float* w, unsigned short* hi, unsigned short* lo, int Cout, int Cin, void* stream);
/* The data gradient of a linear layer / 1x1 conv on the streaming kernel (skipgn.hip), for the large-M layers where the tiled fp32-operand
   GEMM is latency-bound: dx[M][K] = dy[M][N] @ W[N][K] with dy fp32 rows (split to bf16 hi / lo on the way into LDS) and the bf16 planes
   of W^T ([K][N], pitch ldwt) from cdae_wt_planes_bf16 (once per weight version); bf16x3 products, N % 32 == 0. */
int cdae_wt_planes_bf16(const float* w, long ldw, unsigned short* hi, unsigned short* lo, int N, int K, void* stream);
int cdae_linear_dgrad_stream(const float* dy, long lddy, const unsigned short* wt_hi, const unsigned short* wt_lo, long ldwt, float* dx, long lddx,
                             int M, int N, int K, void* stream);
int cdae_conv3x3_s2_dgrad_ps(const unsigned short* dy_hi, const unsigned short* dy_lo, const unsigned short* w4_hi, const unsigned short* w4_lo, float* dx,
                             long lddx, int N, int Ho, int Wo, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* cdae_split_f16 + cdae_wdgrad_planes for MANY conv3x3 weights in one launch (once per optimizer step).  The weights live in one
   fp32 buffer `flat`; desc = nw records {long offset (elements); int Cout, Cin, first_tile, pad} with first_tile the running sum of
   9 * ceil(Cout/32) * ceil(Cin/32); every output plane is addressed by (weight offset - base). */
int cdae_wprep_all(const float* flat, const void* desc, int nw, int total_tiles, long base, unsigned short* f_hi, unsigned short* f_lo,
                   unsigned short* b_hi, unsigned short* b_lo, void* stream);
int cdae_wprep_all_k(const float* flat, const void* desc, int nw, int total_tiles, long base, unsigned short* f_hi, unsigned short* f_lo,
                     unsigned short* b_hi, unsigned short* b_lo, unsigned short* kf_hi, unsigned short* kf_lo, unsigned short* kb_hi,
                     unsigned short* kb_lo /* K-group-major copies (cdae_conv_wpack's order), all NULL or all given */,
                     const float* w_scales /* scale records (cdae_weight_scales) applied to the f16 planes: record index = desc.flags >> 8; NULL: unscaled */,
                     void* stream);
int cdae_conv3x3_dgrad_ps(const unsigned short* dy_hi, const unsigned short* dy_lo, const unsigned short* wt_hi, const unsigned short* wt_lo,
                          float* dx, long lddx, int N, int H, int W, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* wgrad of a stride-1 conv3x3 with 1..8 output channels over a dense NHWC fp32 input (the `out` conv, unet.py:474-478): sliding-window
   fp32 FMAs, one thread per input channel; cdae_conv3x3_wgrad routes such shapes here.  ws: >= ceil(N*H/2) * (Cout*9*Cin + 8) floats
   (fewer row groups are used if it is smaller). */
/* the UNet's input conv (unet.py:395-399; 1..4 input channels, stride 1, NHWC rows out): exact fp32 on the vector ALUs, one output image
   row per block; cdae_conv3x3_fwd routes such shapes here.  x may have any element strides (the model input is NCHW). */
int cdae_conv3x3_stem_supported(int Cin, int Cout, int W);
/* cdae_conv3x3_stem that also leaves the next GroupNorm's partial sums: gn_part [N H W / 32][Cout][2] (may be NULL; W % 32 == 0, W <= 128) */
int cdae_conv3x3_stem_gn(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* bias, float* out, long ldo,
                         float* gn_part, int N, int H, int W, int Cin, int Cout, void* stream);
int cdae_conv3x3_stem(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* bias, float* out, long ldo,
                      int N, int H, int W, int Cin, int Cout, void* stream);
/* the UNet's output head (unet.py:495-499 self.out: GroupNorm32 -> SiLU -> conv3x3 to out_channels) as one kernel in exact fp32 on the
   vector ALUs: y[N][Cout][H][W] = bias + conv3x3(silu?(x * a + b)), x NHWC rows (pixel pitch ldx), coef [N][Cin][2] from cdae_gn_coef,
   w OHWI [Cout][3][3][Cin].  Cin % 32 == 0, W <= 64, Cout in {1, 2, 3, 4, 6, 8}; no-grad path only. */
int cdae_head_conv_supported(int Cin, int Cout, int W);
int cdae_head_conv_fwd(const float* x, long ldx, const float* coef, int silu, const float* w, const float* bias, float* y,
                       int N, int H, int W, int Cin, int Cout, void* stream);
int cdae_conv3x3_wgrad_fewout(const float* x, const float* dy, long lddy, float* dw, float* dbias, int N, int H, int W, int Cin, int Cout,
                              int accumulate, float* ws, size_t ws_bytes, void* stream);
int cdae_conv3x3_wgrad_win_supported(int N, int H, int W, int Cin, int Cout);
/* one weight gradient of a GROUP launch: the arguments of cdae_conv3x3_wgrad_win */
typedef struct cdae_wg_item {
    const unsigned short* a_hi; const unsigned short* a_lo; const unsigned short* dy_hi; const unsigned short* dy_lo;
    float* dw; float* dbias;
    int N, H, W, Cin, Cout, accumulate;
} cdae_wg_item;
/* n weight gradients (any mix of shapes) in ONE launch per 12 items: the convs of a resolution level together fill the chip without
   splitting their pixel ranges — no [ksplit][Cout][9 Cin] slabs, no finish launches (autograd's conv backward, nn.py:470-480) */
int cdae_conv3x3_wgrad_win_group(const cdae_wg_item* items, int n, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_conv3x3_wgrad_win(const unsigned short* a_hi, const unsigned short* a_lo, const unsigned short* dy_hi, const unsigned short* dy_lo,
                           float* dw, float* dbias, int N, int H, int W, int Cin, int Cout, int accumulate, float* splitk_ws,
                           size_t splitk_ws_bytes, void* stream);
/* wgrad: dw (OHWI) (+)= dy^T * im2col(x);  dbias (+)= column sums of dy (may be NULL). */
int cdae_conv3x3_wgrad(const float* x, long sn, long sy, long sx, long sc, const float* dy, long lddy, float* dw, float* dbias,
                       int N, int H, int W, int Cin, int Cout, int stride, int up, int accumulate,
                       float* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* y[M][N] = act(alpha * x[M][K] @ w[N][K]^T + bias + res) — replaces nn.Linear (unet.py:354-379, emb_layers
 * :148-154, nn.py:57-58,233-237) and the 1x1 convs (skip_connection unet.py:171, qkv/proj_out unet.py:216-218)
 * on NHWC rows.  act: 0 none, 1 SiLU, 2 LeakyReLU(0.01). */
int cdae_linear_fwd(const float* x, long ldx, const float* w, long ldw, const float* w_scale, const float* bias, const float* res, float* y, long ldy,
                    unsigned short* y_hi, unsigned short* y_lo /* optional: the result also as f16 hi/lo planes, pitch ldy */,
                    int M, int N, int K, float alpha, int act, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* dx[M][K] (+)= dy[M][N] @ w[N][K] */
int cdae_linear_dgrad(const float* dy, long lddy, const float* w, long ldw, float* dx, long lddx, int M, int N, int K, int accumulate,
                      float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* dw[N][K] (+)= dy[M][N]^T @ x[M][K];  dbias[N] (+)= column sums of dy (may be NULL) */
int cdae_linear_wgrad(const float* x, long ldx, const float* dy, long lddy, float* dw, long lddw, float* dbias, int M, int N, int K,
                      int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* one weight gradient of a GROUP launch: the arguments of cdae_linear_wgrad (x [M][K] rows of pitch ldx, dy [M][N] rows of pitch lddy,
   dw [N][K] rows of pitch lddw, dbias [N] or NULL) */
typedef struct cdae_lw_item {
    const float* x; const float* dy; float* dw; float* dbias;
    long ldx, lddy, lddw;
    int M, N, K, accumulate;
} cdae_lw_item;
/* n such weight gradients (the 1x1 convs / linears of a resolution level: skip_connection, qkv, proj_out — unet.py:165-171, 216-236) in
   ONE unsplit launch where together they fill the chip (each alone has 9 - 48 tiles and would split its rows 8 - 26 ways into slabs + a
   finish launch); otherwise one launch each.  Members with dbias must have accumulate = 1. */
int cdae_linear_wgrad_group(const cdae_lw_item* items, int n, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* the same for the 16-bit torso: io = 12 — x and dy of every member are bf16 rows (the pointers of cdae_lw_item are reinterpreted, pitches in
   elements); members the streaming kernel takes (cdae_linear_wgrad_io's wg16 form) keep their own launch, the rest go out together.  io = 0: as above. */
int cdae_linear_wgrad_group_io(const cdae_lw_item* items, int n, int io, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_colsum(const float* x, long ldx, float* out, long rows, int cols, int accumulate, void* stream);

/* QKVAttention (unet.py:239-253) on the NHWC output of the qkv 1x1 conv: qkv[B][T][heads*3*ch] with the
 * reference's per-head channel order q|k|v.  out[B][T][heads*ch].  probs: workspace [B*heads][T][T] floats
 * (kept for the backward).  Two batched MFMA GEMMs + a wave-shuffle softmax. */
int cdae_qkv_attention_fwd(const float* qkv, float* out, float* probs, int B, int T, int heads, int ch, void* stream);
/* dqkv from dout; dprobs: scratch [B*heads][T][T] */
int cdae_qkv_attention_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, float* dprobs,
                           int B, int T, int heads, int ch, void* stream);
/* QKVAttention forward for inference as ONE kernel (unet.py:239-253): q k^T, fp32 softmax and the product with v per (batch, head)
   with the [T, T] probabilities kept in registers (no `probs` output, so no backward).  T in {64, 256}, ch in {64, 96, 128}:
   cdae_qkv_attention_fused_supported tells; other shapes use cdae_qkv_attention_fwd. */
int cdae_qkv_attention_fused_supported(int T, int ch);
int cdae_qkv_attention_fwd_fused(const float* qkv, float* out, int B, int T, int heads, int ch, void* stream);
/* The same kernel for the TRAINING forward: additionally writes the normalised probabilities [B*heads][T][T] that
   cdae_qkv_attention_bwd reads (probs == NULL: as cdae_qkv_attention_fwd_fused).  One launch instead of GEMM, softmax, GEMM. */
int cdae_qkv_attention_fwd_fused_p(const float* qkv, float* out, float* probs, int B, int T, int heads, int ch, void* stream);
/* The query side of the attention backward in one launch (same shapes): dP = dO V^T, dS = P o (dP - rowsum(P o dP)) written to ds
   ([B*heads][T][T], for the dK GEMM), dQ = dS K / sqrt(ch) written to the q slices of dqkv.  cdae_qkv_attention_bwd uses it. */
int cdae_qkv_attention_bwd_q_fused(const float* qkv, const float* probs, const float* dout, float* dqkv, float* ds, int B, int T, int heads, int ch,
                                   void* stream);

/* ---- normalisation / softmax (norm.hip) ------------------------------------------------------------------- */
size_t cdae_gn_workspace_floats(int N, int C);
/* GroupNorm32 statistics (nn.py:435-437): mean/rstd [N*groups] over (HW x C/groups), fp32 in, f64 combine */
int cdae_gn_stats(const float* x, int N, int HW, int C, int ldx, int groups, float eps, float* mean, float* rstd, float* ws, void* stream);
/* y = silu?( ((x-mean)*rstd*gamma+beta) [* (1+scale[n][c]) + shift[n][c]] ), scale_shift = emb_out [N][2C] (unet.py:190-194) */
int cdae_gn_apply(const float* x, float* y, int N, int HW, int C, int ldx, int ldy, int groups, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream);
int cdae_gn_bwd(const float* x, const float* dy, float* dx, int N, int HW, int C, int ldx, int lddy, int lddx, int groups,
                const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu,
                float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift, int ld_dss, int accumulate_dx,
                float* ws, void* stream);
/* cdae_gn_bwd with two fusions for the ResBlock backward: dx_add (nullable, pitch ld_add) — another gradient of the same tensor, the
   residual path, added on the way out instead of by a separate add; dxb_hi / dxb_lo (nullable, dense [N*HW][C]) — dx also written
   as bf16 hi/lo planes, the operand of the preceding conv's dgrad / wgrad (dx may then be NULL: planes only). */
int cdae_gn_bwd_ex(const float* x, const float* dy, float* dx, int N, int HW, int C, int ldx, int lddy, int lddx, int groups,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu,
                   float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift, int ld_dss, int accumulate_dx,
                   const float* dx_add, int ld_add, unsigned short* dxb_hi, unsigned short* dxb_lo, float* ws, void* stream);
size_t cdae_bn_workspace_floats(int C);
/* BatchNorm2d (batch stats if training, running stats otherwise) + LeakyReLU on NHWC rows (nn.py:46-53) */
int cdae_bn_lrelu_fwd(const float* x, float* y, long rows, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int training, float eps, float momentum, float slope, float* scale, float* shift,
                      float* save_mean, float* save_rstd, float* ws, void* stream);
/* scale/shift are the forward's folded affine (y = lrelu(x*scale+shift)): the backward evaluates the kink on exactly that value */
int cdae_bn_lrelu_bwd(const float* x, const float* dy, float* dx, long rows, int C, const float* gamma, const float* scale,
                      const float* shift, const float* save_mean, const float* save_rstd, float slope, float* dgamma, float* dbeta,
                      int accumulate, float* ws, void* stream);
int cdae_softmax_rows(float* s, long rows, int T, void* stream);
int cdae_softmax_rows_bwd(const float* P, float* dP, long rows, int T, void* stream);

/* ---- pointwise (elementwise.hip) -------------------------------------------------------------------------- */
/* pointwise activation of a stored pre-activation and its gradient; kind: 1 SiLU, 2 LeakyReLU(0.01) (nn.py:232), 3 ReLU, 4 sigmoid
   (the conditioner nets of MultivariateCausalFlow, nn.py:350-366).  In place (y == x, dx == dy) is allowed. */
int cdae_act_fwd(const float* x, float* y, long n, int kind, void* stream);
int cdae_act_bwd(const float* x, const float* dy, float* dx, long n, int kind, void* stream);
int cdae_silu_fwd(const float* x, float* y, long n, void* stream);
int cdae_silu_bwd(const float* x, const float* dy, float* dx, long n, void* stream);
/* timestep_embedding (nn.py:551-569); freqs[dim/2] is the host-computed exp(-ln(P) k/half) table */
int cdae_timestep_embed_fwd(const float* t, const float* freqs, float* out, int N, int dim, void* stream);
/* _WrappedModel (respace.py:119-124): out_i = map[t] (int64, exact), out_f = rescale ? float(map[t])*scale : float(map[t]) */
int cdae_model_timesteps(const long long* t, const long long* map, float scale, int rescale, float* out_f, long long* out_i, int N, void* stream);
int cdae_embedding_add(float* emb, const float* table, const long long* idx, int N, int D, void* stream);     /* unet.py:550 */
int cdae_embedding_bwd(const float* demb, float* dtable, const long long* idx, int N, int D, void* stream);
int cdae_axpby(float a, const float* x, float b, const float* y, float* out, long n, void* stream);          /* y may be NULL */
int cdae_mul_scale(const float* x, const float* m, float scale, float* out, long n, void* stream);          /* nn.Dropout (unet.py:153): x * keep-mask / (1 - p), and its backward */
int cdae_mul_rows(float* x, const float* m, int N, int D, void* stream);                                       /* unet.py:608-611 */
int cdae_copy2d(const float* src, float* dst, long rows, int cols, long lds, long ldd, int accumulate, void* stream);  /* th.cat unet.py:628 */
int cdae_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, void* stream);
int cdae_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, void* stream);
/* batch assembly from an HBM-resident u8 HWC image pool: out[b,:] = pool[idx[b],:] / div + shift (NHWC fp32, IEEE division).  Replaces the
   reference's host DataLoader + ToTensor/`/255` per item (image_datasets.py:266-297, 352-372, 451-467); per_sample % 4 == 0 */
int cdae_gather_u8(const unsigned char* pool, const long long* idx, float* out, int B, long per_sample, float div, float shift, void* stream);
int cdae_sumpool2(const float* src, float* dst, int N, int H, int W, int C, void* stream);
/* resampling WITHOUT a convolution (conv_resample=False; unet.py:76-78 F.interpolate(scale_factor=2, "nearest"), :101-103 avg_pool_nd(2)),
   fp32 NHWC: y[N,2H,2W,C] = scale * nearest2x(x[N,H,W,C]);  dst[N,H,W,C] = scale * (sum of the 2 x 2 blocks of src[N,2H,2W,C]).
   Upsample: (upsample2, 1) forward, (pool2, 1) backward; average pool: (pool2, 0.25) forward, (upsample2, 0.25) backward. */
int cdae_upsample2(const float* x, float* y, int N, int H, int W, int C, float scale, void* stream);
int cdae_pool2(const float* src, float* dst, int N, int H, int W, int C, float scale, void* stream);
/* q_sample (gaussian_diffusion.py:201-222) */
int cdae_q_sample(const float* x0, const float* noise, const long long* t, const float* tab, int T, float* out, int N, long per_sample, void* stream);
/* one fused DDIM update (p_mean_variance eps branch + ddim_sample, gaussian_diffusion.py:336-338,533-558); noise may be NULL iff eta==0.
   clip: bit 0 = clamp pred_xstart to [-1,1]; bit 1 = `eps` already holds the processed pred_xstart (learned-variance / x0-prediction callers) */
int cdae_ddim_update(const float* x, const float* eps, const long long* t, const float* tab, int T, float eta, const float* noise, int clip,
                     float* sample, float* pred_xstart, int N, long per_sample, void* stream);
/* one fused ancestral update (p_sample, gaussian_diffusion.py:383-414) */
int cdae_ddpm_update(const float* x, const float* eps, const long long* t, const float* tab, int T, const float* noise, int clip,
                     float* sample, float* pred_xstart, int N, long per_sample, void* stream);
/* p_mean_variance for every (mean, variance) parameterisation the factory can build (gaussian_diffusion.py:248-353):
   mean_type 0 = eps-prediction, 1 = x0-prediction; var_type 0 = fixed (table), 1 = LEARNED, 2 = LEARNED_RANGE.  model_out is NCHW
   [N, C or 2C, H, W] (per sample: mean part then variance part).  Any output may be NULL; `sample` (needs `noise`) is the fused
   ancestral draw mean + (t != 0) exp(lv / 2) noise (gaussian_diffusion.py:383-414). */
int cdae_p_mean_variance(const float* x, const float* model_out, const long long* t, const float* tab, int T, int mean_type, int var_type,
                         int clip, const float* noise, float* mean, float* variance, float* log_variance, float* pred_xstart, float* sample,
                         int N, long per_sample, void* stream);
/* one term of the variational bound per sample, in bits/dim (_vb_terms_bpd, gaussian_diffusion.py:682-715; normal_kl and
   discretized_gaussian_log_likelihood, losses.py:12-77), and its gradient with respect to model_out (freeze_mean: the mean half
   gets zeros, gaussian_diffusion.py:822-825). */
int cdae_vb_terms(const float* x_start, const float* x_t, const float* model_out, const long long* t, const float* tab, int T, int mean_type,
                  int var_type, int clip, float* vb, float* pred_xstart, int N, long per_sample, void* stream);
int cdae_vb_terms_bwd(const float* x_start, const float* x_t, const float* model_out, const long long* t, const float* tab, int T, int mean_type,
                      int var_type, int clip, int freeze_mean, const float* gout, float* dmodel_out, int N, long per_sample, void* stream);
int cdae_softplus_fwd(const float* x, float* y, long n, float add, void* stream);                             /* nn.py:108 */
int cdae_softplus_bwd(const float* x, const float* dy, float* dx, long n, void* stream);
int cdae_reparam(const float* m, const float* v, float vscale, const float* eps, float* z, long n, void* stream);   /* nn.py:460-467 */
int cdae_causal_mask(const float* u, const float* A, float* out, int N, int nv, int d, int transpose, void* stream); /* nn.py:290-295 */
/* AdamW + EMA over flat fp32 buffers (train_util.py:292-297, nn.py:503-513); ema may be NULL */
int cdae_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int step, double ema_rate, double grad_scale, void* stream);
/* the same step with 0..4 EMA buffers in ONE pass (train_util.py:292-297 loops `update_ema` over every rate of "--ema_rate a,b");
   emas / ema_rates are HOST arrays of n_ema device pointers / rates; grad_scale multiplies the gradient as it is read (1 / world: the
   mean of the all-reduced sum, train_util.py:111-118, without a pass of its own) */
int cdae_adamw_ema_multi(float* p, const float* g, float* m, float* v, float* const* emas, const double* ema_rates, int n_ema, long n, double lr,
                         double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);
int cdae_sqsum(const float* x, long n, double* out, void* stream);                                            /* grad-norm, train_util.py:299-303 */
int cdae_mse_rows(const float* a, const float* b, float* out, int N, long per, void* stream);               /* mean_flat((a-b)^2), gaussian_diffusion.py:847 */
int cdae_mse_rows_bwd(const float* a, const float* b, const float* gout, float* db, int N, long per, void* stream);
/* representation loss per sample (gaussian_diffusion.py:727-766, nn.py:440-457): out[n] = KL(N(mu, var) || N(0, I)) summed over D
   (+ sum over the nv latent slices of KL(N(z_post_i, I) || N(c[n][i], I)) when z_post is given: the causal prior whose mean is the
   label, :718-725).  _bwd: gradients with respect to mu, var and z_post for an upstream gout[N]. */
int cdae_rep_loss(const float* mu, const float* var, const float* z_post, const float* c, float* out, int N, int D, int nv, void* stream);
int cdae_rep_loss_bwd(const float* mu, const float* var, const float* z_post, const float* c, const float* gout, float* dmu, float* dvar, float* dz_post,
                      int N, int D, int nv, void* stream);

/* ---- dispatch thresholds.  The dispatcher picks a kernel per shape from measured thresholds; these two can be moved at run time
 * (process-wide, not thread-safe against concurrent launches).  The parity tests use them to run small golden cases through the
 * kernels the large benchmark shapes dispatch; nothing else changes numerically relevant behaviour.
 *   CDAE_TUNE_CONVWIN_MIN_TILES  (256) stride-1 conv3x3 on planes runs on convwin_kernel when 256x128 tiles x K splits >= this,
 *                                else on the first-generation 128-row window kernel (smaller tiles fill the chip at small batch)
 *   CDAE_TUNE_CONVWIN_SPLITK     (1)   0: convwin_kernel never splits K
 *   CDAE_TUNE_CONVWIN_NJ3        (0)   256 x 96 tiles of the forward window kernel (Cout % 96 == 0, unsplit K): 0 = where they fill the
 *                                      block slots and 256 x 128 tiles do not, 1 = wherever they apply (tests), -1 = never
 *   CDAE_TUNE_HEAD_MFMA          (1)   the output head (cdae_head_conv_fwd) on v_mfma_f32_4x4x1 (exact fp32 products, like the scalar
 *                                      form it replaces: 0 = that form; the two differ in summation order only)
 *   CDAE_TUNE_ROWS16_MIN_M       (2048) cdae_gemm16_ps with at least this many rows (K in {64, 128, 192, 256, 384, 512; 768 without residual}, bf16 result / residual, no
 *                                      GroupNorm sums / accumulation) runs on the streaming kernel of rows16.hip (weight fragments in
 *                                      registers, activation rows global -> registers); below it on the plane GEMM, whose split-K
 *                                      covers short row counts (cdae_linear_wgrad_io takes wg16.hip from twice this many rows).
 *                                      1: wherever they apply (tests)
 *   CDAE_TUNE_ROWS16_RING        (1)   K = 256, N % 256 == 0 of that kernel (K = 384 / 512 always): the rows of a step reach the four waves of a block through
 *                                      an LDS ring filled by LDS-DMA several steps ahead; 0: every wave loads them into registers itself
 *   CDAE_TUNE_CONVWIN_PAIR16     (1)   the window conv on bf16 rows (cdae_conv3x3_fwd16 / _dgrad16, Cin % 64 == 0, K-group-major weights):
 *                                      the two plane slots of the kernel carry the two halves of the input channels (two MFMAs per K step);
 *                                      0: the one-plane instantiation
 *   CDAE_TUNE_GN_BWD_FOLD2       (1)   GroupNorm backward as TWO launches (partial sums; dx with the group fold in its prologue and the
 *                                      channel folds as extra rows of its grid); 0: the separate fold launch between them (A/B, tests)
 *   CDAE_TUNE_GROUP_BIG_TILES    (384) cdae_linear_wgrad_group runs 128 x 128 tiles when the members together have at least this many of them,
 *                                      else 64 x 64 tiles (at least 192 of those, or one launch per member)
 *   CDAE_TUNE_CONVWIN_NJ2        (0)   256 x 64 tiles of the forward window kernel (Cout % 64 == 0, unsplit K): 0 = where 256 x 128 tiles fill less
 *                                      than 0.6 of the block slots and the narrow ones clearly more (small-batch sampling at 64 x 64 / 32 x 32),
 *                                      1 = wherever they apply (tests), -1 = never
 *   CDAE_TUNE_WGWIN_DIST         (2)   the window weight gradient (cdae_conv3x3_wgrad_win*) requests its operands this many 64-pixel steps ahead
 *                                      (2 where the deeper ring fits the 160 KB of LDS — everything but two-plane operands at W = 64 — else 1);
 *                                      1: one step ahead everywhere (the round-3..5 schedule; A/B, tests)
 *   CDAE_TUNE_WGWIN_SWZ          (1)   its LDS image keeps the 32-byte halves of rows with bit 3 set swapped (W >= 16: conflict-free transpose
 *                                      reads); 0: plain rows (A/B, tests).  Neither key changes a result bit.
 *   CDAE_TUNE_WGWIN_FIXED        (12)  what a block of the window weight gradient costs outside its step loop, in 64-pixel steps (prologue, fold, epilogue):
 *                                      the split search of a grouped launch minimises rounds x (steps per block + this)
 *   CDAE_TUNE_WGWIN_CO2          (1)   one-plane window weight gradient on 128-output-channel block tiles (the wave groups split the output channels instead of
 *                                      the step's pixels) where every member of a launch has Cout % 128 == 0: 1 = launches of two or more members,
 *                                      2 = every launch (tests), 0 = never
 *   CDAE_TUNE_WG16_SLOTS         (512) cdae_linear_wgrad_io's streaming kernel (wg16.hip) splits its rows until tiles x splits reach this many blocks
 *                                      (two block slots per CU = 512; half of it when dW has at most four tiles); fewer blocks = fewer, larger slabs for the finish */
enum { CDAE_TUNE_CONVWIN_MIN_TILES = 0, CDAE_TUNE_CONVWIN_SPLITK = 1, CDAE_TUNE_CONVWIN_NJ3 = 2, CDAE_TUNE_HEAD_MFMA = 3, CDAE_TUNE_ROWS16_MIN_M = 4,
       CDAE_TUNE_ROWS16_RING = 5, CDAE_TUNE_CONVWIN_PAIR16 = 6, CDAE_TUNE_GN_BWD_FOLD2 = 7, CDAE_TUNE_GROUP_BIG_TILES = 8, CDAE_TUNE_CONVWIN_NJ2 = 9,
       CDAE_TUNE_WGWIN_DIST = 10, CDAE_TUNE_WGWIN_SWZ = 11, CDAE_TUNE_WG16_SLOTS = 12, CDAE_TUNE_WGWIN_FIXED = 13, CDAE_TUNE_WGWIN_CO2 = 14 };
int cdae_tune_set(int key, int value);
int cdae_tune_get(int key);       /* -1: unknown key */

/* ---- the 16-bit torso (reference unet.py:501-507 convert_to_fp16, fp16_util.py:9-15; GroupNorm32 computes in fp32, nn.py:435-437):
   activations and gradients are bf16 NHWC rows (void*), accumulation and statistics fp32.  `io` bits: 1 the result is bf16, 2 the
   residual is bf16, 4 / 8 the A / B operand of an fp32-operand entry point is bf16. */
int cdae_conv3x3_fwd16(const void* x16, long sn, long sy, long sx, const void* w16, const void* wk16, const float* bias, const void* res16, void* out16,
                       long ldo, float* gn_part, int N, int H, int W, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_conv3x3_s2_fwd16(const void* x16, long sn, long sy, long sx, const void* w16, const float* bias, void* out16, long ldo, int N, int H, int W, int Cin,
                          int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream);      /* the Downsample conv (stride 2) on bf16 rows */
int cdae_conv3x3_dgrad16(const void* dy16, const void* wt16, const void* wtk16, void* dx16, long lddx, int N, int H, int W, int Cin, int Cout,
                         float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_gemm16_ps(const void* a16, long lda, const void* b16, long ldb, const float* bias, const void* res, void* c, long ldc, float* gn_part, int M, int N,
                   int K, int io, int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_linear_fwd_io(const float* x, long ldx, const float* w, long ldw, const float* w_scale, const float* bias, const void* res, void* y, long ldy,
                       int M, int N, int K, int io, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_linear_dgrad_io(const float* dy, long lddy, const float* w, long ldw, void* dx, long lddx, int M, int N, int K, int io,
                         float* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cdae_linear_wgrad_io(const void* x, long ldx, const void* dy, long lddy, float* dw, long lddw, float* dbias, int M, int N, int K, int io,
                         int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* GroupNorm32 on bf16 rows (one or two sources = a channel concatenation read in place): statistics (+ the folded (a, b) table), apply
   (+ scale-shift, + SiLU) to bf16 rows, backward (dx / dx2 bf16, parameter and scale-shift gradients fp32) */
int cdae_gn_stats16(const void* x1, int ld1, const void* x2, int ld2, int C1, int N, int HW, int C, int groups, float eps, float* mean, float* rstd,
                    const float* gamma, const float* beta, const float* scale_shift, int ld_ss, float* coef, float* ws, void* stream);
int cdae_gn_apply16(const void* x1, int ld1, const void* x2, int ld2, int C1, void* y, int ldy, int N, int HW, int C, int groups, const float* mean,
                    const float* rstd, const float* gamma, const float* beta, const float* scale_shift, int ld_ss, int silu, void* stream);
int cdae_gn_bwd16(const void* x, int ldx, const void* x2, int ld2, int C1, const void* dy, int lddy, void* dx, int lddx, void* dx2, int lddx2,
                  int N, int HW, int C, int groups, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  const float* scale_shift, int ld_ss, int silu, float* dgamma, float* dbeta, int accumulate_params, float* d_scale_shift,
                  int ld_dss, int accumulate_dx, const void* dx_add, int ld_add, float* ws, void* stream);
/* QKVAttention on bf16 rows (unet.py:239-253), forward and backward without a [T, T] tensor in memory: lse [B * heads][T] = log-sum-exp
   of the scaled scores (kept for the backward, which recomputes the probabilities), dsum [B * heads][T] scratch.  T in {64, 256},
   ch in {64, 96, 128}. */
int cdae_attn16_supported(int T, int ch);
int cdae_attn16_fwd(const void* qkv16, void* out16, float* lse, int B, int T, int heads, int ch, void* stream);
int cdae_attn16_bwd(const void* qkv16, const void* out16, const void* dout16, const float* lse, float* dsum, void* dqkv16, int B, int T,
                    int heads, int ch, void* stream);
int cdae_gn_parts16(const void* x, long ldx, float* parts, long M, int C, void* stream);      /* [ceil(M / 32)][C][2] sums of a bf16 tensor */
/* 1: cdae_gemm16_ps with these dimensions (16-byte aligned operands, no GroupNorm sums, no accumulation) runs on the streaming
   kernel of rows16.hip; 0: on the plane GEMM */
int cdae_rows16_supported(int M, int N, int K, int io, int has_res);
int cdae_cast_f32_bf16(const float* x, void* y, long n, void* stream);
int cdae_cast_bf16_f32(const void* x, float* y, long n, void* stream);
int cdae_upsample2_16(const void* x, void* y, int N, int H, int W, int C, void* stream);       /* nearest 2x of bf16 NHWC rows (unet.py:76-78) */
int cdae_sumpool2_16(const void* src, void* dst, int N, int H, int W, int C, void* stream);
/* y[N H W][9][C] = the 3 x 3 patches (zero padded) of the bf16 NHWC tensor x: a conv's weight gradient then is cdae_linear_wgrad_io over
   [pixels][9 C] rows — used where the rows are too short for the window wgrad kernel (the 4 x 4 level) */
int cdae_im2col3x3_16(const void* x, void* y, int N, int H, int W, int C, void* stream);    /* its gradient: 2x2 sum pool */
int cdae_wprep_all_m16(const float* flat, const void* desc, int nw, int total_tiles, long base, unsigned short* f_hi, unsigned short* f_lo,
                       unsigned short* b_hi, unsigned short* b_lo, unsigned short* kf_hi, unsigned short* kf_lo, unsigned short* kb_hi,
                       unsigned short* kb_lo, const float* w_scales, void* stream);

/* ---- opt-in profiler (prof.hip): HIP events on the launch stream around every launch of a kernel family */
int cdae_prof_enable(int on);
/* per family since the last read: milliseconds, flops (2MNK as executed), algorithmic bytes (convwin: each operand and the result once), launches */
int cdae_prof_read(double* ms, double* work, double* bytes, long long* launches);
/* box calibration (bench.py reports it beside every roofline fraction: the pool's boxes differ by +-3 %): the f16 MFMA rate a
   register-resident v_mfma_f32_16x16x32_f16 loop sustains on random operands (TFLOP/s, two waves per SIMD on every CU) with the shader
   clock the part holds under that load (GHz, s_memtime over s_memrealtime), and the read + write rate of a flat HBM copy (TB/s).
   Both synchronise the stream; scratch >= 4 MiB + 8 KiB of device memory. */
int cdae_calib_mfma(void* scratch, size_t scratch_bytes, int iters, double* tflops, double* sclk_ghz, void* stream);
int cdae_calib_copy(const void* src, void* dst, size_t bytes, int reps, double* tbps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
