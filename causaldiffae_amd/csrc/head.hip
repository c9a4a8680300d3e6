// head_mfma_kernel (default) / head_conv_kernel (the scalar form, CDAE_TUNE_HEAD_MFMA = 0): the UNet output head (reference unet.py:495-499: self.out = GroupNorm32 -> SiLU -> zero-initialised conv3x3 to
// out_channels, 1..8 channels) in ONE kernel and in exact fp32 on the vector ALUs:
//     y[n][co][p] = bias[co] + sum_{tap, c} silu(x[n][p + tap][c] * a[n][c] + b[n][c]) * w[co][tap][c]        (NCHW out, zero padding)
// Why not the MFMA path: with 4 output channels a 64-wide n-tile is 94 % padding and the plane GEMM streams every activation nine
// times through L2 (300 us at batch 128, plus the GroupNorm pass that writes the planes: 70 us).  Here the fp32 input is read once
// (HBM floor: 4 B per input element), normalised and activated on its way into an LDS window of 256 + 2 W + 2 pixel rows x 32
// channels, and every thread owns one output pixel: 9 taps x 32 channels of ds_read_b128 against weights that sit in scalar
// registers (the weight index is wave-uniform).  Image borders: rows of the window outside the image are zero, and a tap that would
// wrap around a row end reads a zero row instead (the address is chosen once per tile, nothing is masked in the loop).
// The result is the reference's fp32 arithmetic up to summation order (no split-precision products in the model's last layer).
#include "cdae_internal.h"
#include "../../include/cdae.h"

namespace {

constexpr int HD_TP = 256;            // output pixels per block
constexpr int HD_KC = 32;             // channels per LDS window
constexpr int HD_PITCH = 36;          // floats per window row: 144 B rows keep 8 consecutive lanes' 16-byte reads on distinct banks

struct HeadParams {
    const float* x; long ldx;          // NHWC rows, pixel pitch ldx floats
    const float* coef;                 // [N][Cin][2]: y = silu?(x * a + b)
    const float* w;                    // OHWI [Cout][3][3][Cin]
    const float* bias; float* y;       // y NCHW [N][Cout][H*W]
    int N, H, W, Cin, silu, tiles_per_image;
    int* range_flag;
};

template <int CO>
__global__ __launch_bounds__(256, 2) void head_conv_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float win[];      // [rows + 1][HD_PITCH]; the last row stays zero
    const int tid = threadIdx.x;
    const int HW = p.H * p.W, W = p.W;
    const int n = blockIdx.x / p.tiles_per_image, p0 = (blockIdx.x - n * p.tiles_per_image) * HD_TP;
    const int rows = HD_TP + 2 * W + 2;                                // window row r <-> pixel p0 - W - 1 + r
    const int zrow = rows;

    // ---- this thread's output pixel and the LDS float offsets of its nine taps
    const int pix = p0 + tid;
    const int py = pix / W, px = pix - py * W;
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const bool wraps = (dx < 0 && px == 0) || (dx > 0 && px == W - 1);
        toff[t] = (wraps ? zrow : tid + W + 1 + dy * W + dx) * HD_PITCH;
    }
    for (int i = tid; i < HD_PITCH; i += 256) win[zrow * HD_PITCH + i] = 0.f;

    // ---- loader geometry: thread -> (row = tid / 8 + 32 i, channel quad c4 = tid % 8), the same quad for all its rows
    const int c4 = tid & 7, lrow = tid >> 3;
    const float* xin = p.x + (long)n * HW * p.ldx;
    const float* cf = p.coef + (long)n * p.Cin * 2;

    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;

    for (int k0 = 0; k0 < p.Cin; k0 += HD_KC) {
        const float4 ab0 = *reinterpret_cast<const float4*>(cf + (k0 + 4 * c4) * 2);          // a0 b0 a1 b1
        const float4 ab1 = *reinterpret_cast<const float4*>(cf + (k0 + 4 * c4) * 2 + 4);      // a2 b2 a3 b3
        // all of this thread's window rows in flight at once (at most 13: 386 rows / 32), then normalise and stage them
        constexpr int LR = (HD_TP + 2 * 64 + 2 + 31) / 32;
        float4 xv[LR];
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const int r = lrow + 32 * i, q = p0 - W - 1 + r;
            const bool ok = r < rows && q >= 0 && q < HW;
            xv[i] = *reinterpret_cast<const float4*>(xin + (ok ? (long)q * p.ldx + k0 + 4 * c4 : 0));      // (a dummy in-range address when masked)
        }
        if (k0) __syncthreads();                                       // everyone is done reading the previous window
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const int r = lrow + 32 * i, q = p0 - W - 1 + r;
            if (r < rows) {
                float4 v;
                v.x = fmaf(xv[i].x, ab0.x, ab0.y); v.y = fmaf(xv[i].y, ab0.z, ab0.w);
                v.z = fmaf(xv[i].z, ab1.x, ab1.y); v.w = fmaf(xv[i].w, ab1.z, ab1.w);
                if (p.silu) { v.x = cdae_silu(v.x); v.y = cdae_silu(v.y); v.z = cdae_silu(v.z); v.w = cdae_silu(v.w); }
                if (!(q >= 0 && q < HW)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(win + r * HD_PITCH + 4 * c4) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* src = win + toff[t];
            // wave-uniform addresses read through the constant address space: scalar loads (s_load_dwordx4), weights in SGPRs
            typedef const float __attribute__((address_space(4))) * cfp;
            const cfp wt = (cfp)(p.w + (long)t * p.Cin + k0);                // + co * 9 * Cin
#pragma unroll
            for (int j = 0; j < HD_KC / 4; ++j) {
                const float4 xv = *reinterpret_cast<const float4*>(src + 4 * j);
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const cfp wc = wt + (long)co * 9 * p.Cin + 4 * j;
                    acc[co] = fmaf(xv.x, wc[0], acc[co]);
                    acc[co] = fmaf(xv.y, wc[1], acc[co]);
                    acc[co] = fmaf(xv.z, wc[2], acc[co]);
                    acc[co] = fmaf(xv.w, wc[3], acc[co]);
                }
            }
        }
    }
    if (pix < HW) {
        bool bad = false;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float v = acc[co] + (p.bias ? p.bias[co] : 0.f);
            p.y[((long)n * CO + co) * HW + pix] = v;
            bad |= !__builtin_isfinite(v);
        }
        if (bad && p.range_flag) *p.range_flag = 1;
    }
}

// The same head on the matrix cores, still in exact fp32 products: v_mfma_f32_4x4x1_16b_f32 is sixteen independent 4 x 4 x 1 outer
// products per instruction — lane l supplies A = its pixel's activation for ONE input channel and B = the weight of output channel
// l % 4 for that channel, and the sixteen blocks are sixteen groups of four neighbouring pixels: one instruction = 64 pixels x 4
// output channels x 1 input channel (tools/hiptests/mfma4x4.hip pins the lane mapping: D[lane 4 b + j][reg i] = A[lane 4 b + i] *
// B[lane 4 b + j]).  Per 16 bytes of window and 16 bytes of weights read from LDS a lane issues four of them, where the scalar form
// above needs 4 CO fused multiply-adds and as many weight operands through the scalar cache (its limiter: 277 us for 4 channels at
// batch 128; this form is bound by the two LDS reads per four instructions).  Staging, window geometry and border handling are the
// scalar kernel's; the weights of the chunk sit in LDS as [tap][4 G output channels][32 + 4 pad] (rows of absent channels zero).
// Sums: per output, input channels 4 j + e with e even / odd go to two accumulators (independent MFMA chains), added at the end.
// G: groups of four output channels (1: Cout <= 4, 2: Cout 6 or 8).
template <int G>
__global__ __launch_bounds__(256, 2) void head_mfma_kernel(const HeadParams p, int CO) {
    extern __shared__ __attribute__((aligned(16))) float win[];      // [rows + 1][HD_PITCH] window, then [9][4 G][HD_PITCH] weights
    typedef float hd_f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int HW = p.H * p.W, W = p.W;
    const int n = blockIdx.x / p.tiles_per_image, p0 = (blockIdx.x - n * p.tiles_per_image) * HD_TP;
    const int rows = HD_TP + 2 * W + 2;
    const int zrow = rows;
    float* const wl = win + (rows + 1) * HD_PITCH;

    const int pix = p0 + tid;
    const int py = pix / W, px = pix - py * W;
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const bool wraps = (dx < 0 && px == 0) || (dx > 0 && px == W - 1);
        toff[t] = (wraps ? zrow : tid + W + 1 + dy * W + dx) * HD_PITCH;
    }
    for (int i = tid; i < HD_PITCH; i += 256) win[zrow * HD_PITCH + i] = 0.f;

    const int c4 = tid & 7, lrow = tid >> 3;
    const float* xin = p.x + (long)n * HW * p.ldx;
    const float* cf = p.coef + (long)n * p.Cin * 2;

    hd_f4 acc[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc[g][0] = hd_f4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = hd_f4{0.f, 0.f, 0.f, 0.f}; }

    for (int k0 = 0; k0 < p.Cin; k0 += HD_KC) {
        const float4 ab0 = *reinterpret_cast<const float4*>(cf + (k0 + 4 * c4) * 2);
        const float4 ab1 = *reinterpret_cast<const float4*>(cf + (k0 + 4 * c4) * 2 + 4);
        constexpr int LR = (HD_TP + 2 * 64 + 2 + 31) / 32;
        float4 xv[LR];
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const int r = lrow + 32 * i, q = p0 - W - 1 + r;
            const bool ok = r < rows && q >= 0 && q < HW;
            xv[i] = *reinterpret_cast<const float4*>(xin + (ok ? (long)q * p.ldx + k0 + 4 * c4 : 0));
        }
        // this chunk's weights: float4 pieces (tap t, output channel co, channels 4 c .. 4 c + 3), 9 x 4 G x 8 of them
        float4 wv[(9 * 4 * G * 8 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (9 * 4 * G * 8 + 255) / 256; ++i) {
            const int idx = tid + 256 * i, c = idx & 7, co = (idx >> 3) % (4 * G), t = idx / (32 * G);
            wv[i] = (idx < 9 * 4 * G * 8 && co < CO) ? *reinterpret_cast<const float4*>(p.w + ((long)co * 9 + t) * p.Cin + k0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (k0) __syncthreads();                                       // everyone is done reading the previous window and weights
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const int r = lrow + 32 * i, q = p0 - W - 1 + r;
            if (r < rows) {
                float4 v;
                v.x = fmaf(xv[i].x, ab0.x, ab0.y); v.y = fmaf(xv[i].y, ab0.z, ab0.w);
                v.z = fmaf(xv[i].z, ab1.x, ab1.y); v.w = fmaf(xv[i].w, ab1.z, ab1.w);
                if (p.silu) { v.x = cdae_silu(v.x); v.y = cdae_silu(v.y); v.z = cdae_silu(v.z); v.w = cdae_silu(v.w); }
                if (!(q >= 0 && q < HW)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(win + r * HD_PITCH + 4 * c4) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < (9 * 4 * G * 8 + 255) / 256; ++i) {
            const int idx = tid + 256 * i, c = idx & 7, row = idx >> 3;          // row = t * 4 G + co
            if (idx < 9 * 4 * G * 8) *reinterpret_cast<float4*>(wl + row * HD_PITCH + 4 * c) = wv[i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* src = win + toff[t];
            const float* wsrc = wl + (t * 4 * G + (lane & 3)) * HD_PITCH;
#pragma unroll
            for (int j = 0; j < HD_KC / 4; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(src + 4 * j);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float4 b = *reinterpret_cast<const float4*>(wsrc + g * 4 * HD_PITCH + 4 * j);
                    acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b.x, acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b.y, acc[g][1], 0, 0, 0);
                    acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b.z, acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b.w, acc[g][1], 0, 0, 0);
                }
            }
        }
    }
    // lane 4 b + j of a wave holds output channel 4 g + j of the wave's pixels 4 b .. 4 b + 3
    const int pb = p0 + (tid & ~63) + (lane & ~3);
    bool bad = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int co = 4 * g + (lane & 3);
        if (co >= CO) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
        float* dst = p.y + ((long)n * CO + co) * HW + pb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = (acc[g][0][i] + acc[g][1][i]) + bv;
            if (pb + i < HW) dst[i] = v;
            bad |= !__builtin_isfinite(v);
        }
    }
    if (bad && p.range_flag) *p.range_flag = 1;
}

template <int G>
int launch_head_mfma(const HeadParams& p, int CO, hipStream_t st) {
    const size_t smem = ((size_t)(HD_TP + 2 * p.W + 3) + 9 * 4 * G) * HD_PITCH * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&head_mfma_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(((HD_TP + 2 * 64 + 3) + 9 * 4 * G) * HD_PITCH * sizeof(float))) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    hipLaunchKernelGGL(head_mfma_kernel<G>, dim3((unsigned)(p.N * p.tiles_per_image)), dim3(256), smem, st, p, CO);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("head_mfma_kernel launch failed");
}

template <int CO>
int launch_head(const HeadParams& p, hipStream_t st) {
    const size_t smem = (size_t)(HD_TP + 2 * p.W + 3) * HD_PITCH * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&head_conv_kernel<CO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((HD_TP + 2 * 64 + 3) * HD_PITCH * sizeof(float))) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    hipLaunchKernelGGL(head_conv_kernel<CO>, dim3((unsigned)(p.N * p.tiles_per_image)), dim3(256), smem, st, p);
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("head_conv_kernel launch failed");
}

}  // namespace

extern "C" int cdae_head_conv_supported(int Cin, int Cout, int W) {
    return Cin > 0 && Cin % 32 == 0 && W >= 2 && W <= 64 && (Cout == 1 || Cout == 2 || Cout == 3 || Cout == 4 || Cout == 6 || Cout == 8);
}

extern "C" int cdae_head_conv_fwd(const float* x, long ldx, const float* coef, int silu, const float* w, const float* bias, float* y,
                                  int N, int H, int W, int Cin, int Cout, void* stream) {
    if (!cdae_head_conv_supported(Cin, Cout, W)) return cdae_fail("head_conv: Cin % 32 == 0, 2 <= W <= 64 and 1, 2, 3, 4, 6 or 8 output channels required");
    if (ldx % 4 || (reinterpret_cast<size_t>(x) & 15) || (reinterpret_cast<size_t>(coef) & 15) || (reinterpret_cast<size_t>(w) & 15) || !x || !coef || !w || !y)
        return cdae_fail("head_conv: 16-byte aligned rows, coefficients and weights required");
    HeadParams p;
    p.x = x; p.ldx = ldx; p.coef = coef; p.w = w; p.bias = bias; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.silu = silu; p.tiles_per_image = (H * W + HD_TP - 1) / HD_TP;
    p.range_flag = cdae_range_flag_ptr();
    hipStream_t st = (hipStream_t)stream;
    cdae_prof_begin(PROF_IGEMM, 2.0 * N * H * W * 9.0 * Cin * Cout, st);
    cdae_prof_note(PROF_IGEMM, 4.0 * N * H * W * (Cin + Cout));
    int rc;
    if (cdae_tune(TUNE_HEAD_MFMA)) {
        rc = Cout <= 4 ? launch_head_mfma<1>(p, Cout, st) : launch_head_mfma<2>(p, Cout, st);
        cdae_prof_end(PROF_IGEMM, st);
        return rc;
    }
    switch (Cout) {
        case 1: rc = launch_head<1>(p, st); break;
        case 2: rc = launch_head<2>(p, st); break;
        case 3: rc = launch_head<3>(p, st); break;
        case 4: rc = launch_head<4>(p, st); break;
        case 6: rc = launch_head<6>(p, st); break;
        default: rc = launch_head<8>(p, st); break;
    }
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
