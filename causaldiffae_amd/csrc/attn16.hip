// attn16.hip — QKVAttention (reference improved_diffusion/unet.py:239-253) of the 16-bit torso: bf16 qkv rows in, bf16 rows out,
// forward and backward without the [T, T] probabilities ever touching HBM (gfx950 only).
//
//   qkv16 [B][T][heads * 3 ch]  (per head: q | k | v, the reference's reshape to [B * heads, 3 ch, T]),  out16 [B][T][heads * ch]
//   forward : S^T = K Q^T per 32-key tile (lane = query, registers = keys), fp32 softmax in registers (unet.py:250-252 keeps it in
//             fp32 too), O = P V; besides O it leaves lse[b * heads + h][t] = log sum_s exp(alpha q_t . k_s), 4 bytes per query
//   backward: P is RECOMPUTED from q, k and lse (one exp per score), D_t = sum_c dO[t][c] O[t][c] replaces the row sum of P o dP:
//       query side  (a wave owns 32 queries, keys / values staged in LDS):   dQ = alpha dS K,         dS = P o (dP - D),  dP = dO V^T
//       key side    (a wave owns 32 keys, queries / dO staged in LDS):       dV = P^T dO,  dK = alpha dS^T Q
//     — the same kernel with the two operand pairs swapped (SIDE).  What the fp32-storage path moves per (batch, head) and T = 256:
//     probabilities written once and read three times, dS written once and read twice (1.5 MB each); here nothing of size T x T.
// Products are single-plane bf16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the mixed16 mode's arithmetic); operands are
// already 16-bit in HBM, so staging is a plain 16-byte copy global -> LDS and the fragments of a wave's own rows are 16-byte loads.
#include <hip/hip_runtime.h>
#include <math.h>

#include "cdae.h"
#include "cdae_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mma16(const u16x8& a, const u16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u16x8 pack8(const float* v) {
    bf8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (__bf16)v[i];
    return __builtin_bit_cast(u16x8, h);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// rows row0 .. row0 + ROWS of a [rows][CH] bf16 slice (row pitch `pitch_el` elements) -> LDS image [ROWS][P bytes]
template <int CH, int ROWS, int P, int THREADS>
__device__ __forceinline__ void stage16(const unsigned short* __restrict__ src, long pitch_el, char* __restrict__ dst, int tid) {
    constexpr int PIECES = CH / 8;
    for (int item = tid; item < ROWS * PIECES; item += THREADS) {
        const int row = item / PIECES, pc = item - row * PIECES;
        *reinterpret_cast<uint4*>(dst + row * P + pc * 16) = *reinterpret_cast<const uint4*>(src + (long)row * pitch_el + pc * 8);
    }
}
// B operand of a k-major [row][CH] image: 8 rows (the MFMA's k) of column (32 j + l31) for this lane, rows krow + {0..3} and + 8 + {0..3}
template <int P>
__device__ __forceinline__ u16x8 tr_frag(const char* img, int krow, int col_bytes) {
    const char* src = img + krow * P + col_bytes;
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
    const fp16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + 8 * P));
    const u16x4 a4 = __builtin_bit_cast(u16x4, a), c4 = __builtin_bit_cast(u16x4, c);
    u16x8 r;
    r[0] = a4[0]; r[1] = a4[1]; r[2] = a4[2]; r[3] = a4[3]; r[4] = c4[0]; r[5] = c4[1]; r[6] = c4[2]; r[7] = c4[3];
    return r;
}

template <int T> struct AttCfg {
    static constexpr int NKT = T / 32, WAVES = NKT > 8 ? 8 : NKT, THREADS = 64 * WAVES;
};

// ------------------------------------------------------------------------------------------------------------------- forward
template <int CH, int T>
__global__ __launch_bounds__(AttCfg<T>::THREADS) void attn16_fwd_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                                      float* __restrict__ lse, int heads, float alpha) {
    constexpr int NKT = AttCfg<T>::NKT, THREADS = AttCfg<T>::THREADS, KSTEPS = CH / 16, CT = CH / 32;
    constexpr int P = CH * 2 + 16;                                   // LDS row pitch in bytes (b128 row reads and transpose reads both conflict-light)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const kimg = reinterpret_cast<char*>(smem);               // [T][P] keys
    char* const vimg = kimg + T * P;                                 // [T][P] values
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const long C3 = (long)heads * 3 * CH, C = (long)heads * CH;
    const unsigned short* const base = qkv + (long)b * T * C3 + (long)h * 3 * CH;
    stage16<CH, T, P, THREADS>(base + CH, C3, kimg, tid);
    stage16<CH, T, P, THREADS>(base + 2 * CH, C3, vimg, tid);
    const int q0 = wave * 32;
    u16x8 qf[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qf[s] = *reinterpret_cast<const u16x8*>(base + (long)(q0 + l31) * C3 + s * 16 + 8 * hh);
    __syncthreads();
    f32x16 acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
        const char* arow = kimg + (kt * 32 + l31) * P + 16 * hh;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) acc[kt] = mma16(*reinterpret_cast<const u16x8*>(arow + s * 32), qf[s], acc[kt]);
    }
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[kt][r] *= alpha; m = fmaxf(m, acc[kt][r]); }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __expf(acc[kt][r] - m); acc[kt][r] = e; sum += e; }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    if (hh == 0) lse[(long)bh * T + q0 + l31] = m + __logf(sum);
    f32x16 o[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, chalf = 16 * ((lane >> 4) & 1);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
            float pv[8];
#pragma unroll
            for (int jx = 0; jx < 8; ++jx) pv[jx] = acc[kt][8 * sk + jx] * inv;
            const u16x8 pf = pack8(pv);
            const int krow = kt * 32 + 16 * sk + 4 * hh + q4;
#pragma unroll
            for (int j = 0; j < CT; ++j) o[j] = mma16(pf, tr_frag<P>(vimg, krow, (j * 32 + chalf + 4 * p4) * 2), o[j]);
        }
    unsigned short* const obase = out + ((long)b * T + q0) * C + (long)h * CH;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            obase[(long)row * C + j * 32 + l31] = __builtin_bit_cast(unsigned short, (__bf16)o[j][r]);
        }
}

// ------------------------------------------------------------------------------------------------------------------- backward
// SIDE 0 (queries): own rows = queries (fragments of q and dO), staged X = k, Y = v;  writes dQ and D_t.
// SIDE 1 (keys):    own rows = keys    (fragments of k and v),  staged X = q, Y = dO; reads lse, D;  writes dK and dV.
// With S'[other][own] = X own^T and dP'[other][own] = Y ownB^T (lane = own row, registers = the other side's rows):
//   SIDE 0:  acc tile = S^T (keys x queries) ... P, dS per (query lane, key register);  dQ += dS K   (A = dS, B = X k-major)
//   SIDE 1:  acc tile = S (queries x keys)   ... P, dS per (key lane, query register);  dV += P^T dO (B = Y), dK += dS^T Q (B = X)
template <int CH, int T, int SIDE>
__global__ __launch_bounds__(AttCfg<T>::THREADS) void attn16_bwd_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out,
                                                                      const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                                      float* __restrict__ Dsum, unsigned short* __restrict__ dqkv, int heads, float alpha) {
    constexpr int NKT = AttCfg<T>::NKT, THREADS = AttCfg<T>::THREADS, KSTEPS = CH / 16, CT = CH / 32;
    constexpr int P = CH * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const ximg = reinterpret_cast<char*>(smem);
    char* const yimg = ximg + T * P;
    float* const lse_s = reinterpret_cast<float*>(yimg + T * P);     // SIDE 1: lse and D of every query
    float* const d_s = lse_s + T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const long C3 = (long)heads * 3 * CH, C = (long)heads * CH;
    const unsigned short* const base = qkv + (long)b * T * C3 + (long)h * 3 * CH;         // q at +0, k at +CH, v at +2 CH
    const unsigned short* const dob = dout + (long)b * T * C + (long)h * CH;
    const int r0 = wave * 32;                                                             // this wave's first own row
    u16x8 fa[KSTEPS], fb[KSTEPS];                                                         // own rows: (q, dO) or (k, v)
    float lse_own = 0.f, d_own = 0.f;
    if constexpr (SIDE == 0) {
        stage16<CH, T, P, THREADS>(base + CH, C3, ximg, tid);
        stage16<CH, T, P, THREADS>(base + 2 * CH, C3, yimg, tid);
        const unsigned short* orow = out + ((long)b * T + r0 + l31) * C + (long)h * CH;
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            fa[s] = *reinterpret_cast<const u16x8*>(base + (long)(r0 + l31) * C3 + s * 16 + 8 * hh);
            fb[s] = *reinterpret_cast<const u16x8*>(dob + (long)(r0 + l31) * C + s * 16 + 8 * hh);
            const u16x8 ov = *reinterpret_cast<const u16x8*>(orow + s * 16 + 8 * hh);
#pragma unroll
            for (int i = 0; i < 8; ++i) dsum = fmaf(bf2f(fb[s][i]), bf2f(ov[i]), dsum);
        }
        dsum += __shfl_xor(dsum, 32);
        d_own = dsum;
        lse_own = lse[(long)bh * T + r0 + l31];
        if (hh == 0) Dsum[(long)bh * T + r0 + l31] = dsum;
    } else {
        stage16<CH, T, P, THREADS>(base, C3, ximg, tid);
        stage16<CH, T, P, THREADS>(dob, C, yimg, tid);
        for (int i = tid; i < T; i += THREADS) { lse_s[i] = lse[(long)bh * T + i]; d_s[i] = Dsum[(long)bh * T + i]; }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            fa[s] = *reinterpret_cast<const u16x8*>(base + CH + (long)(r0 + l31) * C3 + s * 16 + 8 * hh);
            fb[s] = *reinterpret_cast<const u16x8*>(base + 2 * CH + (long)(r0 + l31) * C3 + s * 16 + 8 * hh);
        }
    }
    __syncthreads();
    f32x16 g0[CT], g1[SIDE == 1 ? CT : 1];                                                // SIDE 0: dQ;  SIDE 1: dK (g0) and dV (g1)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { g0[j][r] = 0.f; if constexpr (SIDE == 1) g1[j][r] = 0.f; }
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, chalf = 16 * ((lane >> 4) & 1);
#pragma unroll 1
    for (int ot = 0; ot < NKT; ++ot) {                                                     // tiles of 32 rows of the OTHER side
        f32x16 sa, pa;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; pa[r] = 0.f; }
        const char* xrow = ximg + (ot * 32 + l31) * P + 16 * hh;
        const char* yrow = yimg + (ot * 32 + l31) * P + 16 * hh;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            sa = mma16(*reinterpret_cast<const u16x8*>(xrow + s * 32), fa[s], sa);       // scores
            pa = mma16(*reinterpret_cast<const u16x8*>(yrow + s * 32), fb[s], pa);       // dP
        }
        float pr[16], dsr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float l_, d_;
            if constexpr (SIDE == 0) { l_ = lse_own; d_ = d_own; }
            else { const int t = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh; l_ = lse_s[t]; d_ = d_s[t]; }
            const float p_ = __expf(fmaf(sa[r], alpha, -l_));
            pr[r] = p_;
            dsr[r] = p_ * (pa[r] - d_);
        }
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
            const u16x8 dsf = pack8(dsr + 8 * sk);
            const int krow = ot * 32 + 16 * sk + 4 * hh + q4;
            if constexpr (SIDE == 0) {
#pragma unroll
                for (int j = 0; j < CT; ++j) g0[j] = mma16(dsf, tr_frag<P>(ximg, krow, (j * 32 + chalf + 4 * p4) * 2), g0[j]);
            } else {
                const u16x8 pf = pack8(pr + 8 * sk);
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    g0[j] = mma16(dsf, tr_frag<P>(ximg, krow, (j * 32 + chalf + 4 * p4) * 2), g0[j]);
                    g1[j] = mma16(pf, tr_frag<P>(yimg, krow, (j * 32 + chalf + 4 * p4) * 2), g1[j]);
                }
            }
        }
    }
    // own rows x ch: lane = ch column, registers = rows
    unsigned short* const ob = dqkv + ((long)b * T + r0) * C3 + (long)h * 3 * CH + (SIDE == 0 ? 0 : CH);
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            ob[(long)row * C3 + j * 32 + l31] = __builtin_bit_cast(unsigned short, (__bf16)(g0[j][r] * alpha));
            if constexpr (SIDE == 1) ob[(long)row * C3 + CH + j * 32 + l31] = __builtin_bit_cast(unsigned short, (__bf16)g1[j][r]);
        }
}

template <int CH, int T>
int launch_fwd(const unsigned short* qkv, unsigned short* out, float* lse, int B, int heads, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * T * (CH * 2 + 16);
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_fwd_kernel<CH, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        done = true;
    }
    hipLaunchKernelGGL((attn16_fwd_kernel<CH, T>), dim3(B * heads), dim3(AttCfg<T>::THREADS), smem, st, qkv, out, lse, heads, 1.f / sqrtf((float)CH));
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("attn16_fwd launch failed");
}
template <int CH, int T, int SIDE>
int launch_bwd(const unsigned short* qkv, const unsigned short* out, const unsigned short* dout, const float* lse, float* D, unsigned short* dqkv, int B,
               int heads, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * T * (CH * 2 + 16) + 2 * T * sizeof(float);
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn16_bwd_kernel<CH, T, SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        done = true;
    }
    hipLaunchKernelGGL((attn16_bwd_kernel<CH, T, SIDE>), dim3(B * heads), dim3(AttCfg<T>::THREADS), smem, st, qkv, out, dout, lse, D, dqkv, heads,
                       1.f / sqrtf((float)CH));
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("attn16_bwd launch failed");
}

}  // namespace

extern "C" int cdae_attn16_supported(int T, int ch) { return (T == 64 || T == 256) && (ch == 64 || ch == 96 || ch == 128); }

#define ATT16_DISPATCH(CALL) \
    if (T == 256) { if (ch == 64) { CALL(64, 256); } else if (ch == 96) { CALL(96, 256); } else { CALL(128, 256); } } \
    else { if (ch == 64) { CALL(64, 64); } else if (ch == 96) { CALL(96, 64); } else { CALL(128, 64); } }

// out16 = softmax(q k^T / sqrt(ch)) v per (batch, head), lse = the log-sum-exp of the scaled scores per query (kept for the backward)
extern "C" int cdae_attn16_fwd(const void* qkv16, void* out16, float* lse, int B, int T, int heads, int ch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!cdae_attn16_supported(T, ch) || (((size_t)qkv16 | (size_t)out16) & 15) || !lse) return cdae_fail("attn16_fwd: T in {64, 256}, ch in {64, 96, 128}, 16-byte aligned rows");
    cdae_prof_begin(PROF_IGEMM, 4.0 * B * heads * (double)T * T * ch, st);
    int rc = -1;
#define CALL(CHV, TV) rc = launch_fwd<CHV, TV>((const unsigned short*)qkv16, (unsigned short*)out16, lse, B, heads, st)
    ATT16_DISPATCH(CALL)
#undef CALL
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}

// dqkv16 (all three slices) from qkv16, the forward's out16 / lse and dout16; dsum: [B * heads][T] floats of scratch
extern "C" int cdae_attn16_bwd(const void* qkv16, const void* out16, const void* dout16, const float* lse, float* dsum, void* dqkv16, int B, int T,
                               int heads, int ch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!cdae_attn16_supported(T, ch) || (((size_t)qkv16 | (size_t)out16 | (size_t)dout16 | (size_t)dqkv16) & 15) || !lse || !dsum)
        return cdae_fail("attn16_bwd: T in {64, 256}, ch in {64, 96, 128}, 16-byte aligned rows");
    cdae_prof_begin(PROF_IGEMM, 10.0 * B * heads * (double)T * T * ch, st);
    int rc = -1;
#define CALL(CHV, TV) { rc = launch_bwd<CHV, TV, 0>((const unsigned short*)qkv16, (const unsigned short*)out16, (const unsigned short*)dout16, lse, dsum, (unsigned short*)dqkv16, B, heads, st); \
                        if (rc == 0) rc = launch_bwd<CHV, TV, 1>((const unsigned short*)qkv16, (const unsigned short*)out16, (const unsigned short*)dout16, lse, dsum, (unsigned short*)dqkv16, B, heads, st); }
    ATT16_DISPATCH(CALL)
#undef CALL
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
