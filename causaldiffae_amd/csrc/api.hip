// api.hip — op-level C-ABI over the igemm kernel family: conv3x3 / linear / QKV attention (fwd, dgrad, wgrad).
// See include/cdae.h for the contract of every entry point.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "cdae_internal.h"
#include "../../include/cdae.h"

namespace {

GemmParams base_params() {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.batch = 1; p.batch_inner = 1; p.alpha = 1.f; p.ksplit = 1; p.stride = 1; p.prec = -1;
    return p;
}

inline bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

void set_splitk(GemmParams& p, float* ws, size_t bytes) {
    p.splitk_ws = ws; p.splitk_ws_bytes = bytes; p.ksplit_auto = ws != nullptr;
}

__global__ void colsum_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, long rows, int cols, long rows_per_block) {
    __shared__ float sh[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (long r = r0 + rl; r < r1; r += 4) s += x[r * ldx + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < cols) atomicAdd(&out[c], (sh[threadIdx.x] + sh[threadIdx.x + 64]) + (sh[threadIdx.x + 128] + sh[threadIdx.x + 192]));
}

}  // namespace

extern "C" {

int cdae_colsum(const float* x, long ldx, float* out, long rows, int cols, int accumulate, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * cols, st) != hipSuccess) return cdae_fail("colsum memset failed");
    long chunks = (rows + 511) / 512;
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    long rpb = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, (unsigned)chunks), dim3(256), 0, st, x, ldx, out, rows, cols, rpb);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("colsum launch failed");
}

int cdae_conv3x3_fwd(const float* x, long sn, long sy, long sx, long sc, const float* w, const float* w_scale, const float* bias, const float* res,
                     float* out, long ldo, int out_nchw, int N, int H, int W, int Cin, int Cout, int stride, int up,
                     float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (stride != 1 && stride != 2) return cdae_fail("conv3x3: stride must be 1 or 2");
    if (up && stride != 1) return cdae_fail("conv3x3: fused upsample needs stride 1");
    const int Ho = up ? 2 * H : (H - 1) / stride + 1, Wo = up ? 2 * W : (W - 1) / stride + 1;
    // the network's input conv (1..4 channels): exact-fp32 streaming kernel of stem.hip instead of a 36-deep implicit GEMM
    if (stride == 1 && !up && !out_nchw && !res && cdae_conv3x3_stem_supported(Cin, Cout, W) && ldo % 4 == 0 && aligned16(out) && aligned16(bias))
        return cdae_conv3x3_stem(x, sn, sy, sx, sc, w, bias, out, ldo, N, H, W, Cin, Cout, stream);
    GemmParams p = base_params();
    p.A = x; p.B = w; p.w_scale = w_scale; p.C = out; p.bias = bias; p.res = res;
    p.M = N * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
    p.ldb = 9L * Cin; p.ldc = ldo;
    p.out_mode = out_nchw ? OUT_NCHW : OUT_ROWMAJOR; p.out_hw = Ho * Wo;
    const bool vec = sc == 1 && Cin % 32 == 0 && sx % 4 == 0 && sy % 4 == 0 && sn % 4 == 0 && aligned16(x);
    p.amode = vec ? A_CONV_VEC : A_CONV_GEN;
    p.bmode = B_PLAIN_KC;
    p.b_scalar = !(p.K % 4 == 0 && aligned16(w));
    p.conv_M = p.M; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.up = up;
    p.sn = sn; p.sy = sy; p.sx = sx; p.sc = sc;
    set_splitk(p, out_nchw ? nullptr : splitk_ws, splitk_ws_bytes);
    if (out_nchw && res) return cdae_fail("conv3x3: residual with NCHW output unsupported");
    return cdae_gemm_dispatch(p, stream);
}

size_t cdae_workspace_bytes(int op, const long* dims, int ndims) {
    switch (op) {
    case CDAE_WS_SPLITK: {
        const size_t cap = (size_t)256 << 20;
        if (!dims || ndims < 3) return cap;
        // the dispatcher wants up to ceil(512 / tiles) slabs of M x N floats (at most 64, at most K / 128): see cdae_gemm_dispatch
        const long M = dims[0], N = dims[1], K = dims[2];
        if (M <= 0 || N <= 0 || K <= 0) return 0;
        const long tiles = ((M + 127) / 128) * ((N + 127) / 128);
        long ks = tiles >= 256 ? 1 : (512 + tiles - 1) / tiles;
        const long kmax = K / 128 > 0 ? K / 128 : 1;
        if (ks > kmax) ks = kmax;
        if (ks > 64) ks = 64;
        // the window conv kernel counts 256 x 128 tiles and splits by 32-channel chunks of a 3x3 conv (K = 9 Cin), two chunks at least
        const long tiles2 = ((M + 255) / 256) * ((N + 127) / 128);
        long ks2 = tiles2 >= 256 ? 1 : (512 + tiles2 - 1) / tiles2;
        if (ks2 > K / 576) ks2 = K / 576;
        if (ks2 > ks) ks = ks2;
        const size_t need = ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
        return need > cap ? cap : need;
    }
    case CDAE_WS_GROUPNORM: return dims && ndims >= 2 ? cdae_gn_workspace_floats((int)dims[0], (int)dims[1]) * sizeof(float) : 0;
    case CDAE_WS_GN_PARTS: return dims && ndims >= 2 ? (size_t)16 * dims[0] * dims[1] : 0;
    case CDAE_WS_BATCHNORM: return dims && ndims >= 1 ? cdae_bn_workspace_floats((int)dims[0]) * sizeof(float) : 0;
    default: return 0;
    }
}

int cdae_conv3x3_fwd_ps(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, const unsigned short* w_hi,
                        const unsigned short* w_lo, const float* w_scale, const float* bias, const float* res, float* out, long ldo, int out_nchw,
                        unsigned short* out_hi, unsigned short* out_lo, float* gn_part, int N, int H, int W,
                        int Cin, int Cout, int stride, int up, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    return cdae_conv3x3_fwd_psk(x_hi, x_lo, sn, sy, sx, w_hi, w_lo, nullptr, nullptr, w_scale, bias, res, out, ldo, out_nchw, out_hi, out_lo, gn_part, N, H, W, Cin,
                                Cout, stride, up, splitk_ws, splitk_ws_bytes, stream);
}

// the same with the weights ALSO in K-group-major order (wk_hi / wk_lo from cdae_conv_wpack, may be null): the window kernel's layout
int cdae_conv3x3_fwd_psk(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, const unsigned short* w_hi,
                         const unsigned short* w_lo, const unsigned short* wk_hi, const unsigned short* wk_lo, const float* w_scale, const float* bias, const float* res,
                         float* out, long ldo, int out_nchw, unsigned short* out_hi, unsigned short* out_lo, float* gn_part, int N, int H, int W,
                         int Cin, int Cout, int stride, int up, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    return cdae_conv3x3_fwd_psg(x_hi, x_lo, sn, sy, sx, 0, w_hi, w_lo, wk_hi, wk_lo, w_scale, bias, res, out, ldo, out_nchw, out_hi, out_lo, gn_part, N, H, W, Cin, Cout,
                                stride, up, splitk_ws, splitk_ws_bytes, stream);
}

// x_gm = 1: the activation planes are group-major, [Cin / 16][N H W][16] (cdae_gn_apply_split2g / cdae_skip_gn_fwd with planes_gm).  Only the
// window kernel reads that layout: returns 3 (no error set) when the shape would run on another kernel — the caller converts with
// cdae_planes_gm_to_pc and calls again with x_gm = 0.
int cdae_conv3x3_fwd_psg(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, int x_gm, const unsigned short* w_hi,
                         const unsigned short* w_lo, const unsigned short* wk_hi, const unsigned short* wk_lo, const float* w_scale, const float* bias, const float* res,
                         float* out, long ldo, int out_nchw, unsigned short* out_hi, unsigned short* out_lo, float* gn_part, int N, int H, int W,
                         int Cin, int Cout, int stride, int up, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (x_gm && (stride != 1 || up || out_nchw || Cin % 16)) return 3;
    if ((wk_hi != nullptr) != (wk_lo != nullptr) || !aligned16(wk_hi) || !aligned16(wk_lo)) return cdae_fail("conv3x3_fwd_psk: packed weights need both planes, 16-byte aligned");
    if (stride != 1 && stride != 2) return cdae_fail("conv3x3: stride must be 1 or 2");
    if (out_hi && (out_nchw || !out_lo)) return cdae_fail("conv3x3_fwd_ps: plane output needs both planes and a row-major result");
    if (up && stride != 1) return cdae_fail("conv3x3: fused upsample needs stride 1");
    const int Ho = up ? 2 * H : (H - 1) / stride + 1, Wo = up ? 2 * W : (W - 1) / stride + 1;
    if ((long)N * H * W * sx >= (1L << 31)) return cdae_fail("conv3x3_fwd_ps: activation larger than 2^31 elements");
    if (sx % 8 || sy % 8 || sn % 8 || !aligned16(x_hi) || !aligned16(x_lo) || !aligned16(w_hi) || !aligned16(w_lo))
        return cdae_fail("conv3x3_fwd_ps: planes must be 16-byte aligned with pixel pitch % 8 == 0");
    GemmParams p = base_params();
    p.presplit = 1;
    p.A = reinterpret_cast<const float*>(x_hi); p.A_lo = x_lo; p.B = reinterpret_cast<const float*>(w_hi); p.B_lo = w_lo;
    p.Bk_hi = wk_hi; p.Bk_lo = wk_lo; p.a_gm = x_gm; p.w_scale = w_scale;
    p.C = out; p.bias = bias; p.res = res; p.C_hi = out_hi; p.C_lo = out_lo; p.gn_part = gn_part;
    p.M = N * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
    p.ldb = 9L * Cin; p.ldc = ldo;
    p.out_mode = out_nchw ? OUT_NCHW : OUT_ROWMAJOR; p.out_hw = Ho * Wo;
    p.amode = A_CONV_VEC; p.bmode = B_PLAIN_KC;
    p.conv_M = p.M; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.up = up;
    p.sn = sn; p.sy = sy; p.sx = sx; p.sc = 1;
    // (GroupNorm partial sums with a K split come from the finish kernel; not with plane output or a strided / up-sampling conv)
    set_splitk(p, (out_nchw || (gn_part && (out_hi || stride != 1 || up))) ? nullptr : splitk_ws, splitk_ws_bytes);
    if (out_nchw && res) return cdae_fail("conv3x3: residual with NCHW output unsupported");
    return cdae_gemm_dispatch(p, stream);
}

// nearest-2x upsample + conv3x3 as four 2x2 convolutions of the LOW-resolution input, one per output parity (ph_y, ph_x):
// rows 2y+ph_y-1 .. 2y+ph_y+1 of the upsampled image are input rows {y-1+ph_y, y+ph_y} with the 3 kernel rows folded 1+2 or
// 2+1, likewise for columns — 16 instead of 36 multiply-adds per (pixel, channel pair).  w4 = [4 phases][Cout][2][2][Cin]
// folded weights (hi / lo planes); the result lands in out[N, 2H, 2W, Cout] rows of pitch ldo.
int cdae_upconv3x3_fwd_ps(const unsigned short* x_hi, const unsigned short* x_lo, long sn, long sy, long sx, const unsigned short* w4_hi,
                          const unsigned short* w4_lo, const float* w_scale, const float* bias, float* out, long ldo, float* gn_part, int N, int H, int W, int Cin,
                          int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if ((long)N * H * W * sx >= (1L << 31)) return cdae_fail("upconv3x3_fwd_ps: activation larger than 2^31 elements");
    if (sx % 8 || sy % 8 || sn % 8 || !aligned16(x_hi) || !aligned16(x_lo) || !aligned16(w4_hi) || !aligned16(w4_lo))
        return cdae_fail("upconv3x3_fwd_ps: planes must be 16-byte aligned with pixel pitch % 8 == 0");
    static const int cfg_fused = CDAE_DEV_INT("CDAE_UPCONV_FUSED", 1);
    for (int ph = cfg_fused ? -1 : 0; ph < 4; ++ph) {      // ph = -1: all four phases as one launch of the window kernel (4x the tiles: the low levels fill the chip)
        GemmParams p = base_params();
        const bool all = ph < 0;
        if (all) { ph = 0; p.nphase = 4; p.phase_w = (long)Cout * 4 * Cin; p.phase_gn = gn_part ? (long)(((long)N * H * W + 31) / 32) * Cout * 2 : 0; }
        p.presplit = 1; p.ps_taps = 4; p.ph_y = ph >> 1; p.ph_x = ph & 1;
        const long woff = (long)ph * Cout * 4 * Cin;
        p.A = reinterpret_cast<const float*>(x_hi); p.A_lo = x_lo;
        p.B = reinterpret_cast<const float*>(w4_hi + woff); p.B_lo = w4_lo + woff;
        p.C = out; p.bias = bias; p.w_scale = w_scale;
        p.M = N * H * W; p.N = Cout; p.K = 4 * Cin;
        p.ldb = 4L * Cin; p.ldc = ldo;
        p.out_mode = OUT_UP2;
        if (gn_part) p.gn_part = gn_part + (long)ph * ((p.M + 31) / 32) * Cout * 2;
        p.amode = A_CONV_VEC; p.bmode = B_PLAIN_KC;
        p.conv_M = p.M; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.stride = 1; p.up = 0;
        p.sn = sn; p.sy = sy; p.sx = sx; p.sc = 1;
        set_splitk(p, (gn_part || all) ? nullptr : splitk_ws, splitk_ws_bytes);
        const int rc = cdae_gemm_dispatch(p, stream);
        if (all) {
            if (rc == 0) return 0;
            if (rc != 2) return rc;
            ph = -1;                           // not taken as one launch: phase by phase
            continue;
        }
        if (rc) return rc;
    }
    return 0;
}

int cdae_linear_fwd_ps(const unsigned short* x_hi, const unsigned short* x_lo, long ldx, const unsigned short* w_hi, const unsigned short* w_lo,
                       long ldw, const float* w_scale, const float* bias, const float* res, float* y, long ldy, int M, int N, int K, float alpha, int act,
                       float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if ((long)M * ldx >= (1L << 31)) return cdae_fail("linear_fwd_ps: activation larger than 2^31 elements");
    if (!aligned16(x_hi) || !aligned16(x_lo) || !aligned16(w_hi) || !aligned16(w_lo)) return cdae_fail("linear_fwd_ps: planes must be 16-byte aligned");
    GemmParams p = base_params();
    p.presplit = 1;
    p.A = reinterpret_cast<const float*>(x_hi); p.A_lo = x_lo; p.B = reinterpret_cast<const float*>(w_hi); p.B_lo = w_lo;
    p.C = y; p.bias = bias; p.res = res; p.w_scale = w_scale;
    p.M = M; p.N = N; p.K = K; p.lda = ldx; p.ldb = ldw; p.ldc = ldy; p.alpha = alpha; p.act = act;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

int cdae_conv3x3_dgrad(const float* dy, long lddy, const float* w, float* dx, long lddx, int N, int H, int W, int Cin, int Cout,
                       int stride, int up, int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    const int Ho = up ? 2 * H : (H - 1) / stride + 1, Wo = up ? 2 * W : (W - 1) / stride + 1;   // dy grid
    GemmParams p = base_params();
    p.A = dy; p.B = w; p.C = dx;
    p.N = Cin; p.K = 9 * Cout; p.ldc = lddx; p.accumulate = accumulate; p.grad_operand = 1;
    // gathered tensor = dy
    p.H = Ho; p.W = Wo; p.Cin = Cout;
    p.sn = (long)Ho * Wo * lddy; p.sy = (long)Wo * lddy; p.sx = lddy; p.sc = 1;
    if (stride == 1) {          // rows enumerate the dy-sized grid (the upsampled grid when up=1)
        p.Ho = Ho; p.Wo = Wo; p.stride = 1; p.tconv = 0; p.wflip = 1;
    } else {                    // rows enumerate dx pixels, transposed stride-2 gather
        p.Ho = H; p.Wo = W; p.tconv = 1; p.wflip = 0;
    }
    p.M = N * p.Ho * p.Wo; p.conv_M = p.M;
    const bool vec = Cout % 32 == 0 && lddy % 4 == 0 && aligned16(dy);
    p.amode = vec ? A_CONV_VEC : A_CONV_GEN;
    p.bmode = B_WDGRAD_MC; p.wCout = Cout; p.wCin = Cin;
    p.b_scalar = !(Cin % 4 == 0 && aligned16(w));
    set_splitk(p, accumulate ? nullptr : splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

int cdae_conv3x3_wgrad(const float* x, long sn, long sy, long sx, long sc, const float* dy, long lddy, float* dw, float* dbias,
                       int N, int H, int W, int Cin, int Cout, int stride, int up, int accumulate,
                       float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    const int Ho = up ? 2 * H : (H - 1) / stride + 1, Wo = up ? 2 * W : (W - 1) / stride + 1;
    // a handful of output channels over a dense NHWC input (the UNet's `out` conv): the sliding-window FMA kernel of wgrad.hip
    if (Cout <= 8 && stride == 1 && !up && sc == 1 && Cin % 4 == 0 && Cin >= 32 && sx == Cin && sy == (long)W * Cin && sn == (long)H * W * Cin &&
        splitk_ws)       // (exact fp32 in every precision mode: a 1..8-row GEMM has nothing for the matrix cores — as igemm it took 969 us on the M32 model)
        return cdae_conv3x3_wgrad_fewout(x, dy, lddy, dw, dbias, N, H, W, Cin, Cout, accumulate, splitk_ws, splitk_ws_bytes, stream);
    GemmParams p = base_params();
    p.A = dy; p.B = x; p.C = dw;
    p.M = Cout; p.N = 9 * Cin; p.K = N * Ho * Wo;
    p.lda = lddy; p.ldc = 9L * Cin; p.accumulate = accumulate; p.grad_operand = 1;
    p.amode = A_PLAIN_MC; p.a_scalar = !(lddy % 4 == 0 && aligned16(dy));
    p.bmode = B_CONV_MC;
    p.b_scalar = !(sc == 1 && Cin % 64 == 0 && sx % 4 == 0 && sy % 4 == 0 && sn % 4 == 0 && aligned16(x));
    p.conv_M = p.K; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.up = up;
    p.sn = sn; p.sy = sy; p.sx = sx; p.sc = sc;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    if (dbias) {      // fused: the n-tile-0 blocks add dY's column sums while they stage it
        if (!accumulate && hipMemsetAsync(dbias, 0, sizeof(float) * Cout, (hipStream_t)stream) != hipSuccess) return cdae_fail("dbias memset failed");
        p.colsum_out = dbias;
        return cdae_gemm_dispatch(p, stream);
    }
    int rc = cdae_gemm_dispatch(p, stream);
    if (rc == 0 && dbias) rc = cdae_colsum(dy, lddy, dbias, (long)N * Ho * Wo, Cout, accumulate, stream);
    return rc;
}

// dgrad of a STRIDE-2 conv3x3 (pad 1) on the pre-split kernels, as four 2 x 2 sub-pixel convolutions of dy (elementwise.hip s2dgrad_wfold_kernel):
// dy bf16 hi / lo planes [N, Ho, Wo, Cout] dense, w4 the folded bf16 planes [4][Cin][2][2][Cout] (cdae_s2dgrad_wfold), dx fp32 [N, 2 Ho, 2 Wo, Cin]
// rows of pitch lddx.  Returns 0, -1 (error) or 2 (shape not taken: the caller runs cdae_conv3x3_dgrad).
int cdae_conv3x3_s2_dgrad_ps(const unsigned short* dy_hi, const unsigned short* dy_lo, const unsigned short* w4_hi, const unsigned short* w4_lo, float* dx,
                             long lddx, int N, int Ho, int Wo, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (Cout % 32 || Cin % 4 || Wo < 2 || (Wo & (Wo - 1)) || (long)N * Ho * Wo * Cout >= (1L << 31)) return 2;
    if (!aligned16(dy_hi) || !aligned16(dy_lo) || !aligned16(w4_hi) || !aligned16(w4_lo)) return cdae_fail("conv3x3_s2_dgrad_ps: planes must be 16-byte aligned");
    for (int ph = -1; ph < 4; ++ph) {      // ph = -1: all four phases as one launch of the window kernel
        GemmParams p = base_params();
        const bool all = ph < 0;
        if (all) { ph = 0; p.nphase = 4; p.phase_w = (long)Cin * 4 * Cout; }
        p.presplit = 1; p.ps_taps = 4; p.ph_y = ph >> 1; p.ph_x = ph & 1; p.grad_operand = 1;
        p.prec = 2;            // bf16 hi / lo pairs in every split mode (the one-plane window kernel has no 4-tap form: mixed16 ran the masked 9-tap gather, 176 us per Downsample)
        const long woff = (long)ph * Cin * 4 * Cout;
        p.A = reinterpret_cast<const float*>(dy_hi); p.A_lo = dy_lo;
        p.B = reinterpret_cast<const float*>(w4_hi + woff); p.B_lo = w4_lo + woff;
        p.C = dx;
        p.M = N * Ho * Wo; p.N = Cin; p.K = 4 * Cout;
        p.ldb = 4L * Cout; p.ldc = lddx;
        p.out_mode = OUT_UP2;
        p.amode = A_CONV_VEC; p.bmode = B_PLAIN_KC;
        p.conv_M = p.M; p.H = Ho; p.W = Wo; p.Cin = Cout; p.Ho = Ho; p.Wo = Wo; p.stride = 1; p.up = 0;
        p.sn = (long)Ho * Wo * Cout; p.sy = (long)Wo * Cout; p.sx = Cout; p.sc = 1;
        set_splitk(p, all ? nullptr : splitk_ws, splitk_ws_bytes);
        const int rc = cdae_gemm_dispatch(p, stream);
        if (all) {
            if (rc == 0) return 0;
            if (rc != 2) return rc;
            ph = -1;
            continue;
        }
        if (rc) return rc;
    }
    return 0;
}

// dgrad of a stride-1 conv3x3 on the pre-split kernels: dx = conv3x3(dy, wt), dy as bf16 hi/lo planes (cdae_split_bf16, dense NHWC,
// pixel pitch Cout) and wt = the flipped / transposed weight planes of cdae_wdgrad_planes ([Cin][9][Cout])
int cdae_conv3x3_dgrad_ps(const unsigned short* dy_hi, const unsigned short* dy_lo, const unsigned short* wt_hi, const unsigned short* wt_lo,
                          float* dx, long lddx, int N, int H, int W, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    return cdae_conv3x3_dgrad_psk(dy_hi, dy_lo, wt_hi, wt_lo, nullptr, nullptr, dx, lddx, N, H, W, Cin, Cout, splitk_ws, splitk_ws_bytes, stream);
}

int cdae_conv3x3_dgrad_psk(const unsigned short* dy_hi, const unsigned short* dy_lo, const unsigned short* wt_hi, const unsigned short* wt_lo,
                           const unsigned short* wtk_hi, const unsigned short* wtk_lo, float* dx, long lddx, int N, int H, int W, int Cin, int Cout,
                           float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if ((wtk_hi != nullptr) != (wtk_lo != nullptr) || !aligned16(wtk_hi) || !aligned16(wtk_lo)) return cdae_fail("conv3x3_dgrad_psk: packed weights need both planes, 16-byte aligned");
    if ((long)N * H * W * Cout >= (1L << 31)) return cdae_fail("conv3x3_dgrad_ps: gradient larger than 2^31 elements");
    if (Cout % 32 || !aligned16(dy_hi) || !aligned16(dy_lo) || !aligned16(wt_hi) || !aligned16(wt_lo))
        return cdae_fail("conv3x3_dgrad_ps: Cout % 32 == 0 and 16-byte aligned planes required");
    GemmParams p = base_params();
    p.presplit = 1; p.grad_operand = 1;
    p.prec = cdae_get_default_precision() == CDAE_PREC_MIXED16 ? 4 : 2;      // bf16 planes: hi / lo pairs (bf16x3) or the hi plane alone (mixed16)
    p.A = reinterpret_cast<const float*>(dy_hi); p.A_lo = dy_lo; p.B = reinterpret_cast<const float*>(wt_hi); p.B_lo = wt_lo;
    p.Bk_hi = wtk_hi; p.Bk_lo = wtk_lo;
    p.C = dx;
    p.M = N * H * W; p.N = Cin; p.K = 9 * Cout;
    p.ldb = 9L * Cout; p.ldc = lddx;
    p.out_mode = OUT_ROWMAJOR; p.out_hw = H * W;
    p.amode = A_CONV_VEC; p.bmode = B_PLAIN_KC;
    p.conv_M = p.M; p.H = H; p.W = W; p.Cin = Cout; p.Ho = H; p.Wo = W; p.stride = 1; p.up = 0;
    p.sx = Cout; p.sy = (long)W * Cout; p.sn = (long)H * W * Cout; p.sc = 1;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

int cdae_linear_fwd(const float* x, long ldx, const float* w, long ldw, const float* w_scale, const float* bias, const float* res, float* y, long ldy,
                    unsigned short* y_hi, unsigned short* y_lo, int M, int N, int K, float alpha, int act, float* splitk_ws,
                    size_t splitk_ws_bytes, void* stream) {
    GemmParams p = base_params();
    p.A = x; p.B = w; p.w_scale = w_scale; p.C = y; p.bias = bias; p.res = res; p.C_hi = y_hi; p.C_lo = y_lo;
    if (y_hi && !y_lo) return cdae_fail("linear_fwd: plane output needs both planes");
    p.M = M; p.N = N; p.K = K; p.lda = ldx; p.ldb = ldw; p.ldc = ldy; p.alpha = alpha; p.act = act;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    p.a_scalar = !(K % 4 == 0 && ldx % 4 == 0 && aligned16(x));
    p.b_scalar = !(K % 4 == 0 && ldw % 4 == 0 && aligned16(w));
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

// y = [x1 | x2] @ w^T + bias: the K range split over two row-major sources (a channel concatenation read in place)
int cdae_linear_fwd_cat(const float* x1, long ld1, int K1, const float* x2, long ld2, const float* w, long ldw, const float* w_scale, const float* bias, float* y,
                        long ldy, int M, int N, int K, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (K1 % 32 || K % 4 || ld1 % 4 || ld2 % 4 || ldw % 4 || !aligned16(x1) || !aligned16(x2) || !aligned16(w))
        return cdae_fail("linear_fwd_cat: K1 % 32 == 0 and 16-byte aligned rows required");
    GemmParams p = base_params();
    p.A = x1; p.A2 = x2; p.lda2 = ld2; p.K1 = K1; p.B = w; p.w_scale = w_scale; p.C = y; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = ld1; p.ldb = ldw; p.ldc = ldy;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

// cdae_linear_fwd_cat that ALSO writes GroupNorm(+SiLU)([x1 | x2]) as f16 hi/lo planes [M][K] while the rows pass through its loader:
// the ResBlock's 1x1 skip conv and the normalisation pass of its first GroupNorm in one sweep over the block input.  coef [N][K][2]
// from cdae_gn_coef; HW = pixels per image (rows per coefficient set).  x2 may be NULL (one source).
int cdae_linear_fwd_cat_gn(const float* x1, long ld1, int K1, const float* x2, long ld2, const float* w, long ldw, const float* w_scale, const float* bias, float* y,
                           long ldy, const float* coef, int silu, unsigned short* s_hi, unsigned short* s_lo, int M, int N, int K, int HW,
                           float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if ((x2 && K1 % 32) || K % 32 || ld1 % 4 || (x2 && ld2 % 4) || ldw % 4 || !aligned16(x1) || !aligned16(x2) || !aligned16(w) || !aligned16(coef) ||
        !aligned16(s_hi) || !aligned16(s_lo) || !s_hi || !s_lo || !coef || HW <= 0 || M % HW)
        return cdae_fail("linear_fwd_cat_gn: K (and K1) % 32 == 0, 16-byte aligned rows / planes / coefficients, M a multiple of HW required");
    GemmParams p = base_params();
    p.A = x1; p.A2 = x2; p.lda2 = ld2; p.K1 = x2 ? K1 : K; p.B = w; p.w_scale = w_scale; p.C = y; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.lda = ld1; p.ldb = ldw; p.ldc = ldy;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    p.gn_coef = coef; p.gn_silu = silu; p.S_hi = s_hi; p.S_lo = s_lo;
    p.Ho = HW; p.Wo = 1;                       // image index of a row: fdiv(m, hw_magic) with hw = Ho * Wo
    p.force_tile = 128;
    set_splitk(p, nullptr, 0);                 // every (row, k) piece must pass exactly one n-tile-0 block; the grids that use this are full
    return cdae_gemm_dispatch(p, stream);
}

int cdae_linear_dgrad(const float* dy, long lddy, const float* w, long ldw, float* dx, long lddx, int M, int N, int K, int accumulate,
                      float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p = base_params();
    p.A = dy; p.B = w; p.C = dx;
    p.M = M; p.N = K; p.K = N; p.lda = lddy; p.ldb = ldw; p.ldc = lddx; p.accumulate = accumulate; p.grad_operand = 1;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_MC;
    p.a_scalar = !(N % 4 == 0 && lddy % 4 == 0 && aligned16(dy));
    p.b_scalar = !(ldw % 4 == 0 && aligned16(w));
    set_splitk(p, accumulate ? nullptr : splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

int cdae_linear_wgrad(const float* x, long ldx, const float* dy, long lddy, float* dw, long lddw, float* dbias, int M, int N, int K,
                      int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p = base_params();
    p.A = dy; p.B = x; p.C = dw;
    p.M = N; p.N = K; p.K = M; p.lda = lddy; p.ldb = ldx; p.ldc = lddw; p.accumulate = accumulate; p.grad_operand = 1;
    p.amode = A_PLAIN_MC; p.bmode = B_PLAIN_MC;
    p.a_scalar = !(lddy % 4 == 0 && aligned16(dy));
    p.b_scalar = !(ldx % 4 == 0 && aligned16(x));
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    if (dbias) {      // fused column sums of dy (see cdae_conv3x3_wgrad)
        if (!accumulate && hipMemsetAsync(dbias, 0, sizeof(float) * N, (hipStream_t)stream) != hipSuccess) return cdae_fail("dbias memset failed");
        p.colsum_out = dbias;
    }
    return cdae_gemm_dispatch(p, stream);
}

// n linear / 1x1-conv weight gradients in ONE launch where together they fill the chip unsplit (igemm.hip cdae_gemm_group_dispatch);
// else, and for members the grouped loaders do not take, one launch each exactly as cdae_linear_wgrad.  Members with a bias gradient
// must accumulate (the fused column sums add into dbias; the trainer's flat gradient buffer is zeroed once per step).
static int lw_one(const cdae_lw_item& it, int io, float* ws, size_t wsb, void* stream) {
    return io ? cdae_linear_wgrad_io(it.x, it.ldx, it.dy, it.lddy, it.dw, it.lddw, it.dbias, it.M, it.N, it.K, io, it.accumulate, ws, wsb, stream)
              : cdae_linear_wgrad(it.x, it.ldx, it.dy, it.lddy, it.dw, it.lddw, it.dbias, it.M, it.N, it.K, it.accumulate, ws, wsb, stream);
}

// io = 0: fp32 rows (cdae_linear_wgrad_group); io = 12: both operands bf16 rows (the 16-bit torso) — members the streaming kernel takes
// (wg16.hip: >= 4096 rows, channel counts that are multiples of 128) keep their own launch, the others are grouped
int cdae_linear_wgrad_group_io(const cdae_lw_item* items, int n, int io, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (n <= 0) return 0;
    if (!items) return cdae_fail("linear_wgrad_group: null items");
    if (io != 0 && (io & 12) != 12) return cdae_fail("linear_wgrad_group_io: io = 0 (fp32 rows) or 12 (bf16 rows of both operands)");
    cdae_lw_item rest[GEMM_GROUP_MAX];
    int nr = 0;
    auto flush = [&]() -> int {
        int rc = 1;
        if (nr >= 2) {
            GemmGroupArg g;
            memset(&g, 0, sizeof(g));
            g.common = base_params();
            g.common.amode = A_PLAIN_MC; g.common.bmode = B_PLAIN_MC; g.common.grad_operand = 1;
            if (io) { g.common.prec = 4; g.common.io16 = io & 12; }
            g.n = nr;
            bool ok = true;
            for (int i = 0; i < nr; ++i) {
                const cdae_lw_item& it = rest[i];
                ok = ok && (it.accumulate || !it.dbias) && it.dw && it.x && it.dy;
                g.items[i] = GemmGroupItem{it.dy, it.x, it.dw, it.dbias, it.N, it.K, it.M, it.accumulate, it.lddy, it.ldx, it.lddw};
            }
            rc = ok ? cdae_gemm_group_dispatch(g, stream) : 1;
            if (rc < 0) return rc;
        }
        if (rc == 1)
            for (int i = 0; i < nr; ++i)
                if (lw_one(rest[i], io, splitk_ws, splitk_ws_bytes, stream)) return -1;
        nr = 0;
        return 0;
    };
    for (int i = 0; i < n; ++i) {
        const cdae_lw_item& it = items[i];
        if (io && splitk_ws && cdae_wg16_ok(it.x, it.ldx, it.dy, it.lddy, it.dw, it.lddw, it.M, it.N, it.K, io, splitk_ws_bytes)) {
            if (lw_one(it, io, splitk_ws, splitk_ws_bytes, stream)) return -1;          // the HBM-stream kernel
            continue;
        }
        rest[nr++] = it;
        if (nr == GEMM_GROUP_MAX && flush()) return -1;
    }
    return flush();
}

int cdae_linear_wgrad_group(const cdae_lw_item* items, int n, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    return cdae_linear_wgrad_group_io(items, n, 0, splitk_ws, splitk_ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The 16-bit torso (reference unet.py:501-507 convert_to_fp16 + fp16_util.py:9-15: half activations between the layers; here bf16, the
// dtype BASELINE config [1] names): activations and gradients are bf16 NHWC rows, which ARE the one-plane operands of the matrix-core
// kernels; accumulation fp32; results rounded to bf16 once, in the epilogue.  `io` bits: 1 = the result (and an accumulate read) is bf16,
// 2 = the residual is bf16, 4 / 8 = the A / B operand of an fp32-operand kernel is a bf16 tensor.
static int conv16_params(GemmParams& p, const void* a16, long sn, long sy, long sx, const void* w16, const void* wk16, int ldb, const float* bias,
                         const void* res, void* out, long ldo, float* gn_part, int N, int H, int W, int Cin, int Cout, int io) {
    if ((long)N * H * W * sx >= (1L << 31)) return cdae_fail("conv3x3 (16-bit): activation larger than 2^31 elements");
    if (sx % 8 || sy % 8 || sn % 8 || !aligned16(a16) || !aligned16(w16) || !aligned16(wk16) || Cin % 32) return cdae_fail("conv3x3 (16-bit): Cin % 32 == 0, 16-byte aligned rows required");
    p = base_params();
    p.presplit = 1; p.prec = 4; p.io16 = io;
    p.A = reinterpret_cast<const float*>(a16); p.A_lo = reinterpret_cast<const unsigned short*>(a16);
    p.B = reinterpret_cast<const float*>(w16); p.B_lo = reinterpret_cast<const unsigned short*>(w16);
    p.Bk_hi = p.Bk_lo = reinterpret_cast<const unsigned short*>(wk16);
    p.C = reinterpret_cast<float*>(out); p.bias = bias; p.res = reinterpret_cast<const float*>(res); p.gn_part = gn_part;
    p.M = N * H * W; p.N = Cout; p.K = 9 * Cin; p.ldb = ldb; p.ldc = ldo;
    p.out_mode = OUT_ROWMAJOR; p.out_hw = H * W;
    p.amode = A_CONV_VEC; p.bmode = B_PLAIN_KC;
    p.conv_M = p.M; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.stride = 1; p.up = 0;
    p.sn = sn; p.sy = sy; p.sx = sx; p.sc = 1;
    return 0;
}

// out16 = conv3x3(x16, w) + bias (+ res16), stride 1: x16 bf16 [N, H, W, Cin] rows of pitch sx, w16 the bf16 OHWI weights, wk16 the same
// K-group-major (or NULL), out16 / res16 bf16 rows of pitch ldo; gn_part as in cdae_conv3x3_fwd_ps (sums of the ROUNDED values)
int cdae_conv3x3_fwd16(const void* x16, long sn, long sy, long sx, const void* w16, const void* wk16, const float* bias, const void* res16, void* out16,
                       long ldo, float* gn_part, int N, int H, int W, int Cin, int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p;
    if (conv16_params(p, x16, sn, sy, sx, w16, wk16, 9 * Cin, bias, res16, out16, ldo, gn_part, N, H, W, Cin, Cout, 1 | (res16 ? 2 : 0))) return -1;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

// out16 = conv3x3(x16, w, stride 2, pad 1) + bias (reference unet.py:92-105 Downsample with use_conv): bf16 rows in and out, the gather of
// the plane GEMM (the window kernels are stride 1); out16 [N, (H - 1) / 2 + 1, (W - 1) / 2 + 1, Cout] rows of pitch ldo
int cdae_conv3x3_s2_fwd16(const void* x16, long sn, long sy, long sx, const void* w16, const float* bias, void* out16, long ldo, int N, int H, int W, int Cin,
                          int Cout, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p;
    if (conv16_params(p, x16, sn, sy, sx, w16, nullptr, 9 * Cin, bias, nullptr, out16, ldo, nullptr, N, H, W, Cin, Cout, 1)) return -1;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    p.stride = 2; p.Ho = Ho; p.Wo = Wo; p.M = N * Ho * Wo; p.conv_M = p.M; p.out_hw = Ho * Wo;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

// dx16 = conv3x3(dy16, wt): dy16 bf16 [N, H, W, Cout] dense, wt16 / wtk16 the bf16 dgrad weights ([Cin][9][Cout] flipped; K-group-major)
int cdae_conv3x3_dgrad16(const void* dy16, const void* wt16, const void* wtk16, void* dx16, long lddx, int N, int H, int W, int Cin, int Cout,
                         float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p;
    if (conv16_params(p, dy16, (long)H * W * Cout, (long)W * Cout, Cout, wt16, wtk16, 9 * Cout, nullptr, nullptr, dx16, lddx, nullptr, N, H, W, Cout, Cin, 1)) return -1;
    p.grad_operand = 1;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

// c[M][N] (+)= a16[M][K] . b16[N][K]^T + bias (+ res) on the plane GEMM: both operands bf16 rows, K contiguous (a 1x1 conv / linear
// forward with b16 = the bf16 weight, its data gradient with b16 = the bf16 W^T); c / res fp32 or bf16 by `io`
int cdae_gemm16_ps(const void* a16, long lda, const void* b16, long ldb, const float* bias, const void* res, void* c, long ldc, float* gn_part, int M, int N,
                   int K, int io, int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if ((long)M * lda >= (1L << 31)) return cdae_fail("gemm16_ps: operand larger than 2^31 elements");
    if (!aligned16(a16) || !aligned16(b16) || K % 32 || lda % 8 || ldb % 8) return cdae_fail("gemm16_ps: K % 32 == 0 and 16-byte aligned rows required");
    if (cdae_rows16_ok(a16, lda, b16, ldb, bias, res, c, ldc, gn_part, M, N, K, io, accumulate))
        return cdae_rows16_gemm(a16, lda, b16, ldb, bias, res, c, ldc, M, N, K, io, stream);
    GemmParams p = base_params();
    p.presplit = 1; p.prec = 4; p.io16 = io & 3;
    p.A = reinterpret_cast<const float*>(a16); p.A_lo = reinterpret_cast<const unsigned short*>(a16);
    p.B = reinterpret_cast<const float*>(b16); p.B_lo = reinterpret_cast<const unsigned short*>(b16);
    p.C = reinterpret_cast<float*>(c); p.bias = bias; p.res = reinterpret_cast<const float*>(res); p.gn_part = gn_part; p.accumulate = accumulate;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    set_splitk(p, (accumulate || gn_part) ? nullptr : splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}

// the fp32-operand linear entry points with 16-bit tensors on some sides (io bits above; single-plane bf16 products)
int cdae_linear_fwd_io(const float* x, long ldx, const float* w, long ldw, const float* w_scale, const float* bias, const void* res, void* y, long ldy,
                       int M, int N, int K, int io, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p = base_params();
    p.A = x; p.B = w; p.w_scale = w_scale; p.C = reinterpret_cast<float*>(y); p.bias = bias; p.res = reinterpret_cast<const float*>(res);
    p.M = M; p.N = N; p.K = K; p.lda = ldx; p.ldb = ldw; p.ldc = ldy; p.io16 = io & 3;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    if (!(K % 4 == 0 && ldx % 4 == 0 && aligned16(x) && ldw % 4 == 0 && aligned16(w))) return cdae_fail("linear_fwd_io: 16-byte aligned fp32 rows required");
    p.prec = 3;
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}
int cdae_linear_dgrad_io(const float* dy, long lddy, const float* w, long ldw, void* dx, long lddx, int M, int N, int K, int io,
                         float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    GemmParams p = base_params();
    p.A = dy; p.B = w; p.C = reinterpret_cast<float*>(dx);
    p.M = M; p.N = K; p.K = N; p.lda = lddy; p.ldb = ldw; p.ldc = lddx; p.grad_operand = 1; p.io16 = io & 1; p.prec = 4;
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_MC;
    if (!(N % 4 == 0 && lddy % 4 == 0 && aligned16(dy) && ldw % 4 == 0 && aligned16(w) && K % 4 == 0)) return cdae_fail("linear_dgrad_io: 16-byte aligned fp32 rows required");
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    return cdae_gemm_dispatch(p, stream);
}
// dw[N][K] (+)= dy^T . x over M rows, dbias (+)= column sums of dy; dy / x fp32 or bf16 rows by io bits 4 / 8
int cdae_linear_wgrad_io(const void* x, long ldx, const void* dy, long lddy, float* dw, long lddw, float* dbias, int M, int N, int K, int io,
                         int accumulate, float* splitk_ws, size_t splitk_ws_bytes, void* stream) {
    if (splitk_ws && cdae_wg16_ok(x, ldx, dy, lddy, dw, lddw, M, N, K, io, splitk_ws_bytes))
        return cdae_wg16(x, ldx, dy, lddy, dw, lddw, dbias, M, N, K, accumulate, splitk_ws, splitk_ws_bytes, stream);
    GemmParams p = base_params();
    p.A = reinterpret_cast<const float*>(dy); p.B = reinterpret_cast<const float*>(x); p.C = dw;
    p.M = N; p.N = K; p.K = M; p.lda = lddy; p.ldb = ldx; p.ldc = lddw; p.accumulate = accumulate; p.grad_operand = 1; p.prec = 4;
    p.io16 = io & 12;
    p.amode = A_PLAIN_MC; p.bmode = B_PLAIN_MC;
    if (!(lddy % 4 == 0 && ldx % 4 == 0 && N % 4 == 0 && K % 4 == 0 && ((size_t)dy & 7) == 0 && ((size_t)x & 7) == 0 &&
          ((io & 4) || aligned16(dy)) && ((io & 8) || aligned16(x))))
        return cdae_fail("linear_wgrad_io: 4-element aligned rows required");
    set_splitk(p, splitk_ws, splitk_ws_bytes);
    if (dbias) {
        if (!accumulate && hipMemsetAsync(dbias, 0, sizeof(float) * N, (hipStream_t)stream) != hipSuccess) return cdae_fail("dbias memset failed");
        p.colsum_out = dbias;
    }
    return cdae_gemm_dispatch(p, stream);
}

int cdae_qkv_attention_fwd(const float* qkv, float* out, float* probs, int B, int T, int heads, int ch, void* stream) {
    const long C = (long)heads * ch, C3 = 3 * C;
    if (ch % 4) return cdae_fail("attention: head dim must be a multiple of 4");
    GemmParams p = base_params();
    // S = (q k^T) / sqrt(ch)   [the reference scales q and k by ch^-1/4 each, unet.py:248-251]
    p.A = qkv; p.B = qkv + ch; p.C = probs;
    p.M = T; p.N = T; p.K = ch; p.lda = C3; p.ldb = C3; p.ldc = T;
    p.batch = B * heads; p.batch_inner = heads;
    p.a_bs0 = T * C3; p.a_bs1 = 3L * ch; p.b_bs0 = T * C3; p.b_bs1 = 3L * ch;
    p.c_bs0 = (long)heads * T * T; p.c_bs1 = (long)T * T;
    p.alpha = 1.f / sqrtf((float)ch);
    p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC;
    p.a_scalar = p.b_scalar = !aligned16(qkv);
    int rc = cdae_gemm_dispatch(p, stream);
    if (rc) return rc;
    rc = cdae_softmax_rows(probs, (long)B * heads * T, T, stream);
    if (rc) return rc;
    // O = P V
    GemmParams q = base_params();
    q.A = probs; q.B = qkv + 2 * ch; q.C = out;
    q.M = T; q.N = ch; q.K = T; q.lda = T; q.ldb = C3; q.ldc = C;
    q.batch = B * heads; q.batch_inner = heads;
    q.a_bs0 = (long)heads * T * T; q.a_bs1 = (long)T * T; q.b_bs0 = T * C3; q.b_bs1 = 3L * ch;
    q.c_bs0 = T * C; q.c_bs1 = ch;
    q.amode = A_PLAIN_KC; q.bmode = B_PLAIN_MC;
    q.a_scalar = !(T % 4 == 0 && aligned16(probs)); q.b_scalar = !aligned16(qkv);
    return cdae_gemm_dispatch(q, stream);
}

int cdae_qkv_attention_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, float* dprobs,
                           int B, int T, int heads, int ch, void* stream) {
    const long C = (long)heads * ch, C3 = 3 * C;
    const float alpha = 1.f / sqrtf((float)ch);
    const bool t4 = T % 4 == 0;
    int rc;
    {   // dV[s][c] = sum_t P[t][s] dO[t][c]
        GemmParams p = base_params();
        p.A = probs; p.B = dout; p.C = dqkv + 2 * ch;
        p.M = T; p.N = ch; p.K = T; p.lda = T; p.ldb = C; p.ldc = C3;
        p.batch = B * heads; p.batch_inner = heads;
        p.a_bs0 = (long)heads * T * T; p.a_bs1 = (long)T * T; p.b_bs0 = T * C; p.b_bs1 = ch; p.c_bs0 = T * C3; p.c_bs1 = 3L * ch;
        p.amode = A_PLAIN_MC; p.bmode = B_PLAIN_MC; p.a_scalar = !(t4 && aligned16(probs)); p.b_scalar = !aligned16(dout); p.grad_operand = 1;
        if ((rc = cdae_gemm_dispatch(p, stream))) return rc;
    }
    // query side in one launch (attention.hip): dP = dO V^T, the softmax backward and dQ = alpha dS K; dS lands in dprobs for the dK GEMM
    // below.  Measured at batch 32: T = 64 58 vs 65 us for the whole backward, T = 256 135 vs 124 us (the kernel reads the probabilities
    // twice and writes dS between two unoverlapped staging phases) -> by default only for T = 64.  CDAE_ATTN_BWD_FUSED=0 never, 2 wherever built
    static const int cfg_fused = CDAE_DEV_INT("CDAE_ATTN_BWD_FUSED", 1);
    const bool fused_q = (cfg_fused >= 2 || (cfg_fused == 1 && T <= 64)) && cdae_get_default_precision() != CDAE_PREC_FP32 && cdae_qkv_attention_fused_supported(T, ch) && aligned16(qkv) &&
                         aligned16(probs) && aligned16(dout) && aligned16(dqkv) && aligned16(dprobs);
    if (fused_q) {
        if ((rc = cdae_qkv_attention_bwd_q_fused(qkv, probs, dout, dqkv, dprobs, B, T, heads, ch, stream))) return rc;
    } else {
    {   // dP[t][s] = sum_c dO[t][c] V[s][c]
        GemmParams p = base_params();
        p.A = dout; p.B = qkv + 2 * ch; p.C = dprobs;
        p.M = T; p.N = T; p.K = ch; p.lda = C; p.ldb = C3; p.ldc = T;
        p.batch = B * heads; p.batch_inner = heads;
        p.a_bs0 = T * C; p.a_bs1 = ch; p.b_bs0 = T * C3; p.b_bs1 = 3L * ch; p.c_bs0 = (long)heads * T * T; p.c_bs1 = (long)T * T;
        p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_KC; p.a_scalar = !aligned16(dout); p.b_scalar = !aligned16(qkv); p.grad_operand = 1;
        if ((rc = cdae_gemm_dispatch(p, stream))) return rc;
    }
    if ((rc = cdae_softmax_rows_bwd(probs, dprobs, (long)B * heads * T, T, stream))) return rc;
    {   // dQ[t][c] = alpha sum_s dS[t][s] K[s][c]
        GemmParams p = base_params();
        p.A = dprobs; p.B = qkv + ch; p.C = dqkv;
        p.M = T; p.N = ch; p.K = T; p.lda = T; p.ldb = C3; p.ldc = C3; p.alpha = alpha;
        p.batch = B * heads; p.batch_inner = heads;
        p.a_bs0 = (long)heads * T * T; p.a_bs1 = (long)T * T; p.b_bs0 = T * C3; p.b_bs1 = 3L * ch; p.c_bs0 = T * C3; p.c_bs1 = 3L * ch;
        p.amode = A_PLAIN_KC; p.bmode = B_PLAIN_MC; p.a_scalar = !(t4 && aligned16(dprobs)); p.b_scalar = !aligned16(qkv); p.grad_operand = 1;
        if ((rc = cdae_gemm_dispatch(p, stream))) return rc;
    }
    }
    {   // dK[s][c] = alpha sum_t dS[t][s] Q[t][c]
        GemmParams p = base_params();
        p.A = dprobs; p.B = qkv; p.C = dqkv + ch;
        p.M = T; p.N = ch; p.K = T; p.lda = T; p.ldb = C3; p.ldc = C3; p.alpha = alpha;
        p.batch = B * heads; p.batch_inner = heads;
        p.a_bs0 = (long)heads * T * T; p.a_bs1 = (long)T * T; p.b_bs0 = T * C3; p.b_bs1 = 3L * ch; p.c_bs0 = T * C3; p.c_bs1 = 3L * ch;
        p.amode = A_PLAIN_MC; p.bmode = B_PLAIN_MC; p.a_scalar = !(t4 && aligned16(dprobs)); p.b_scalar = !aligned16(qkv); p.grad_operand = 1;
        if ((rc = cdae_gemm_dispatch(p, stream))) return rc;
    }
    return 0;
}

}  // extern "C"
