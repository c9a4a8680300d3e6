// Fused QKVAttention forward for inference (reference improved_diffusion/unet.py:239-253): softmax(q k^T / sqrt(ch)) v per
// (batch, head) in ONE kernel — the [T, T] probabilities never leave the registers.  Split precision like the GEMMs: every
// operand is split into f16 hi/lo on its way into LDS / registers and each product is hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation; the softmax itself is fp32.
//
// qkv: rows [B][T][heads * 3ch], per head the channel order is q | k | v (the reference's reshape to [B*heads, 3ch, T]).
// One block = one (batch, head) and 32 * WAVES queries; each wave owns 32 queries:
//   phase 0  the wave's Q rows -> registers as MFMA B-fragments (hi / lo)
//   phase 1  S^T = K Q^T per 32-key tile (K staged through LDS in passes of <= 128 keys): lane = query, registers = keys,
//            so the softmax over keys is an in-lane reduction plus one cross-half shuffle
//   phase 2  softmax in registers (exact: T <= 256 keys are all resident, no online rescaling)
//   phase 3  O = P V: P is already laid out as the MFMA A operand (lane = query row); its register order fixes a permutation
//            of the 16 keys of an MFMA step, and V (staged k-major, read with ds_read_b64_tr_b16) is fetched in that same order
//   phase 4  O -> out[B][T][heads * ch]
#include <hip/hip_runtime.h>

#include "cdae.h"
#include "cdae_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#ifndef ATT_ABL
#define ATT_ABL 0         // dev ablation (timing only): 1 = no MFMAs in the fused forward
#endif
__device__ __forceinline__ f32x16 mma(const u16x8& a, const u16x8& b, const f32x16& c) {
#if ATT_ABL & 1
    f32x16 r = c; r[0] += __builtin_bit_cast(float, (unsigned)(a[0] ^ b[1])); return r;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
#endif
}
// 8 fp32 -> f16 hi / lo fragments.  The empty asm makes each value opaque: when x is the result of a multiply, hipcc otherwise
// converts it to f16 twice — v_cvt_pk_f16_f32 of the fp32 product for the fragment, but v_fma_mixlo_f16 of the UNROUNDED
// product for the subtraction — and in the rare double-rounding cases where the two disagree the hi / lo pair is off by one f16
// ulp (seen as ~1 query in 1000 with an error of 2^-12 of a probability; -ffp-contract=off does not stop that fold).
__device__ __forceinline__ void split8(const float* v, u16x8& hi, u16x8& lo) {
    half8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x = v[i];
        asm volatile("" : "+v"(x));
        h[i] = (_Float16)x;
        l[i] = (_Float16)(x - (float)h[i]);
    }
    hi = __builtin_bit_cast(u16x8, h);
    lo = __builtin_bit_cast(u16x8, l);
}

// Row pitch in bytes of a k-major plane that is read with ds_read_b64_tr_b16 (V in the forward, K in the query-side backward).  ch = 96
// would make (ch + 32) * 2 = 256 bytes = all 64 banks: the four key rows a 16-lane group of the transpose read touches then sit on the
// same banks (measured on the forward: LDS bank-conflict cycles = 56 % of the LDS-active cycles of that instantiation); 288 bytes puts
// consecutive rows eight banks apart.
constexpr int att_vp(int ch) { return (ch + 32) * 2 + ((ch + 32) * 2 % 256 == 0 ? 32 : 0); }

#ifndef ATT_WAVES8
#define ATT_WAVES8 1      // T = 256: eight waves (all 256 queries of a (batch, head)) per block, K and V staged once instead of twice
#endif
template <int CH, int NKT, bool PROBS>
__global__ __launch_bounds__((ATT_WAVES8 && NKT >= 8 && CH <= 96) ? 512 : NKT >= 4 ? 256 : 64 * NKT) void attn_fused_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                                float* __restrict__ probs, int heads, float alpha,
                                                                                int* __restrict__ range_flag) {
    constexpr int T = 32 * NKT, WAVES = (ATT_WAVES8 && NKT >= 8 && CH <= 96) ? 8 : NKT >= 4 ? 4 : NKT, THREADS = 64 * WAVES;
    constexpr int PASS = T < 128 ? T : 128, NPASS = T / PASS, TPP = PASS / 32;       // keys per staging pass, passes, key tiles per pass
    constexpr int KSTEPS = CH / 16, CT = CH / 32;                                      // 16-deep MFMA steps over ch; 32-wide output tiles
    constexpr int KP = CH * 2 + 16;                                                    // K plane row pitch in bytes (conflict-free b128 reads)
    constexpr int VP = att_vp(CH);                                                     // V plane row pitch in bytes (k-major, transpose reads)
    constexpr int PLANE = PASS * (KP > VP ? KP : VP);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);                                   // [2 planes][PASS rows][pitch]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int q0 = (blockIdx.x * WAVES + wave) * 32;                                   // this wave's first query
    const long C3 = (long)heads * 3 * CH, C = (long)heads * CH;
    const float* const base = qkv + (long)b * T * C3 + (long)h * 3 * CH;               // q at +0, k at +CH, v at +2CH

    // ---- phase 0: Q fragments (B operand of S^T = K Q^T: lane = query column, 8 consecutive ch per lane and step)
    u16x8 qh[KSTEPS], ql[KSTEPS];
    {
        const float* qrow = base + (long)(q0 + l31) * C3;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + s * 16 + 8 * hh);
            const float4 c = *reinterpret_cast<const float4*>(qrow + s * 16 + 8 * hh + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            split8(v, qh[s], ql[s]);
        }
    }

    // ---- phase 1: S^T tiles.  acc[kt][r] = S[query = q0 + l31][key = 32 kt + (r & 3) + 8 (r >> 2) + 4 hh]
    f32x16 acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;

    // Staging of a pass of keys (K rows, then V rows) is split into its two halves so that the global loads of the NEXT pass are in
    // flight while the matrix cores work on the current one (round 4: the kernel runs one 512-thread block per CU at T = 256, so nothing
    // else hides them — without MFMAs the T = 256, ch = 96 launch takes 51 of its 83 us): stage_load requests this thread's 16-byte
    // pieces into registers, stage_commit (after the pass's MFMAs and a barrier) splits them into the LDS planes.
    constexpr int ITEMS = (PASS * (CH / 4) + THREADS - 1) / THREADS;
    constexpr bool EXACT = PASS * (CH / 4) % THREADS == 0;
    auto stage_load = [&](float4 (&v)[ITEMS], int key0, int which /*1 = K (KC layout), 2 = V (k-major)*/) {
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int item = tid + u * THREADS, row = item / (CH / 4), f4 = item - row * (CH / 4);
            if (EXACT || item < PASS * (CH / 4)) v[u] = *reinterpret_cast<const float4*>(base + (long)(key0 + row) * C3 + which * CH + f4 * 4);
        }
        __builtin_amdgcn_sched_barrier(0);             // the requests go out HERE, in front of the MFMA phase that follows
    };
    auto stage_commit = [&](const float4 (&v)[ITEMS], int which) {
        const int pitch = which == 1 ? KP : VP;
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int item = tid + u * THREADS, row = item / (CH / 4), f4 = item - row * (CH / 4);
            if (EXACT || item < PASS * (CH / 4)) {
                half4 hi, lo;
                hi[0] = (_Float16)v[u].x; hi[1] = (_Float16)v[u].y; hi[2] = (_Float16)v[u].z; hi[3] = (_Float16)v[u].w;
                lo[0] = (_Float16)(v[u].x - (float)hi[0]); lo[1] = (_Float16)(v[u].y - (float)hi[1]);
                lo[2] = (_Float16)(v[u].z - (float)hi[2]); lo[3] = (_Float16)(v[u].w - (float)hi[3]);
                *reinterpret_cast<half4*>(lds + row * pitch + f4 * 8) = hi;
                *reinterpret_cast<half4*>(lds + PLANE + row * pitch + f4 * 8) = lo;
            }
        }
    };
    float4 stg[ITEMS];
    // PIPE: the next pass is requested in front of the current pass's MFMAs.  Measured per shape (same box, tools/prof_attn.py): T = 256,
    // ch = 96: 81 -> 72 us, T = 64, ch = 128: 19.6 -> 18.3; ch = 64 (the 32 x 32 models): 103 -> 104-113 and 17.6 -> 18.8 — the 24 live
    // staging registers cost those instantiations more than the overlap returns, so they keep the load - commit - compute order.
    constexpr bool PIPE = CH >= 96;

    if constexpr (PIPE) { stage_load(stg, 0, 1); stage_commit(stg, 1); }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if constexpr (!PIPE) {
            if (ps) __syncthreads();
            stage_load(stg, ps * PASS, 1);
            stage_commit(stg, 1);
        }
        __syncthreads();                                                                // pass ps of K is in LDS
        if constexpr (PIPE) {
            if (ps + 1 < NPASS) stage_load(stg, (ps + 1) * PASS, 1);                    // next: the following K pass, or the first V pass
            else stage_load(stg, 0, 2);
        }
#pragma unroll
        for (int t = 0; t < TPP; ++t) {
            const int kt = ps * TPP + t;
            const char* arow = lds + (t * 32 + l31) * KP + 16 * hh;                     // A operand: lane = key row, 8 ch at 8 hh
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const u16x8 kh = *reinterpret_cast<const u16x8*>(arow + s * 32);
                const u16x8 kl = *reinterpret_cast<const u16x8*>(arow + PLANE + s * 32);
                acc[kt] = mma(kl, qh[s], acc[kt]);
                acc[kt] = mma(kh, ql[s], acc[kt]);
                acc[kt] = mma(kh, qh[s], acc[kt]);
            }
        }
        if constexpr (PIPE) {
            __syncthreads();                                                            // everyone is done with pass ps
            stage_commit(stg, ps + 1 < NPASS ? 1 : 2);
        }
    }

    // ---- phase 2: softmax over the keys of each query (lane): in-lane over tiles and registers, then across the two lane halves
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
        for (int r = 0; r < 16; ++r) { const float e = expf(acc[kt][r] - m); acc[kt][r] = e; sum += e; }
    sum += __shfl_xor(sum, 32);
    // the probabilities are split as p * 2^10 (values in [0, 1024]: their f16 lo parts stay NORMAL numbers — unscaled, every lo of a
    // p < 1/8 is an f16 subnormal) and the factor is taken out of O again at the end; powers of two, so nothing is rounded.
    const float inv = 1024.f / sum;
    if constexpr (PROBS) {
        // training forward: the normalised probabilities [B * heads][T][T] are kept for the backward (cdae_qkv_attention_bwd).  A lane owns
        // one query row; registers 4 g .. 4 g + 3 of a key tile are four consecutive keys: one 16-byte store each
        float* const prow = probs + ((long)bh * T + q0 + l31) * T + 4 * hh;
        const float pin = inv * (1.f / 1024.f);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(prow + 32 * kt + 8 * g) =
                    make_float4(acc[kt][4 * g] * pin, acc[kt][4 * g + 1] * pin, acc[kt][4 * g + 2] * pin, acc[kt][4 * g + 3] * pin);
    }

    // ---- phase 3: O = P V.  MFMA step sk of key tile kt consumes registers r = 8 sk .. 8 sk + 7 of acc[kt], i.e. the keys
    //      32 kt + 16 sk + 4 hh + {0,1,2,3, 8,9,10,11}: V is read in exactly that order (two transpose reads 8 rows apart).
    f32x16 o[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, chalf = 16 * ((lane >> 4) & 1);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if constexpr (!PIPE) {
            __syncthreads();
            stage_load(stg, ps * PASS, 2);
            stage_commit(stg, 2);
        }
        __syncthreads();                                                                // pass ps of V is in LDS
        if constexpr (PIPE) { if (ps + 1 < NPASS) stage_load(stg, (ps + 1) * PASS, 2); }
#pragma unroll
        for (int t = 0; t < TPP; ++t) {
            const int kt = ps * TPP + t;
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                float pv[8];
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) pv[jx] = acc[kt][8 * sk + jx] * inv;
                u16x8 ph, pl;
                split8(pv, ph, pl);
                const int krow = t * 32 + 16 * sk + 4 * hh + q4;
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const char* src = lds + krow * VP + (j * 32 + chalf + 4 * p4) * 2;
                    u16x8 vh, vl;
                    {
                        const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
                        const fp16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + 8 * VP));
                        const u16x4 a4 = __builtin_bit_cast(u16x4, a), c4 = __builtin_bit_cast(u16x4, c);
                        vh[0] = a4[0]; vh[1] = a4[1]; vh[2] = a4[2]; vh[3] = a4[3]; vh[4] = c4[0]; vh[5] = c4[1]; vh[6] = c4[2]; vh[7] = c4[3];
                    }
                    {
                        const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + PLANE));
                        const fp16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + PLANE + 8 * VP));
                        const u16x4 a4 = __builtin_bit_cast(u16x4, a), c4 = __builtin_bit_cast(u16x4, c);
                        vl[0] = a4[0]; vl[1] = a4[1]; vl[2] = a4[2]; vl[3] = a4[3]; vl[4] = c4[0]; vl[5] = c4[1]; vl[6] = c4[2]; vl[7] = c4[3];
                    }
                    o[j] = mma(pl, vh, o[j]);
                    o[j] = mma(ph, vl, o[j]);
                    o[j] = mma(ph, vh, o[j]);
                }
            }
        }
        if constexpr (PIPE) {
            if (ps + 1 < NPASS) {
                __syncthreads();
                stage_commit(stg, 2);
            }
        }
    }

    // ---- phase 4: O[query row][ch col] -> out[b][q0 + row][h * CH + col]
    float* const obase = out + ((long)b * T + q0) * C + (long)h * CH;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float v = o[j][r] * (1.f / 1024.f);
            obase[(long)row * C + j * 32 + l31] = v;
            if (!__builtin_isfinite(v) && range_flag) *range_flag = 1;      // q / k / v beyond the f16 range of the in-kernel split (cdae_range_status)
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward, query side, in ONE kernel (the structure of attn_fused_kernel with other operands): per (batch, head) and 32 queries per wave
//     dP^T = V dO^T                      (like S^T = K Q^T: lane = query, registers = keys)
//     dS   = P o (dP - rowsum(P o dP))   (softmax backward in registers; P is read from the forward's probabilities, twice, 16 bytes per lane)
//     dQ   = alpha dS K                  (like O = P V)
// dS is also written out ([B * heads][T][T], the layout of the probabilities) for the key-side GEMM dK = alpha dS^T Q; dV = P^T dO needs
// nothing from here.  Replaces GEMM, softmax backward, GEMM.  Gradient operands: every product is bf16 hi/lo x3 (v_mfma_f32_32x32x16_bf16),
// like the other backward contractions.
typedef __bf16 att_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 att_bf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mma_bf(const u16x8& a, const u16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(att_bf8, a), __builtin_bit_cast(att_bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split8_bf(const float* v, u16x8& hi, u16x8& lo) {
    att_bf8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x = v[i];
        asm volatile("" : "+v"(x));
        h[i] = (__bf16)x;
        l[i] = (__bf16)(x - (float)h[i]);
    }
    hi = __builtin_bit_cast(u16x8, h);
    lo = __builtin_bit_cast(u16x8, l);
}

template <int CH, int NKT>
__global__ __launch_bounds__((ATT_WAVES8 && NKT >= 8 && CH <= 96) ? 512 : NKT >= 4 ? 256 : 64 * NKT)
void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ probs, const float* __restrict__ dout, float* __restrict__ dqkv,
                       float* __restrict__ ds, int heads, float alpha, int* __restrict__ range_flag) {
    constexpr int T = 32 * NKT, WAVES = (ATT_WAVES8 && NKT >= 8 && CH <= 96) ? 8 : NKT >= 4 ? 4 : NKT, THREADS = 64 * WAVES;
    constexpr int PASS = T < 128 ? T : 128, NPASS = T / PASS, TPP = PASS / 32;
    constexpr int KSTEPS = CH / 16, CT = CH / 32;
    constexpr int KP = CH * 2 + 16, VP = att_vp(CH);
    constexpr int PLANE = PASS * (KP > VP ? KP : VP);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int q0 = (blockIdx.x * WAVES + wave) * 32;
    const long C3 = (long)heads * 3 * CH, C = (long)heads * CH;
    const float* const base = qkv + (long)b * T * C3 + (long)h * 3 * CH;               // q at +0, k at +CH, v at +2CH

    // ---- phase 0: dO fragments (B operand: lane = query column, 8 consecutive ch per lane and step)
    u16x8 gh[KSTEPS], gl[KSTEPS];
    {
        const float* grow = dout + ((long)b * T + q0 + l31) * C + (long)h * CH;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(grow + s * 16 + 8 * hh);
            const float4 c = *reinterpret_cast<const float4*>(grow + s * 16 + 8 * hh + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            split8_bf(v, gh[s], gl[s]);
        }
    }

    f32x16 acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;

    // rows key0 .. key0 + PASS of the k (which = 1) or v (which = 2) slice as bf16 hi / lo planes, row pitch `pitch` bytes
    auto stage = [&](int key0, int which, int pitch) {
        for (int item = tid; item < PASS * (CH / 4); item += THREADS) {
            const int row = item / (CH / 4), f4 = item - row * (CH / 4);
            const float4 v = *reinterpret_cast<const float4*>(base + (long)(key0 + row) * C3 + which * CH + f4 * 4);
            att_bf4 hi, lo;
            hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
            lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
            lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
            *reinterpret_cast<att_bf4*>(lds + row * pitch + f4 * 8) = hi;
            *reinterpret_cast<att_bf4*>(lds + PLANE + row * pitch + f4 * 8) = lo;
        }
    };

    // ---- phase 1: dP^T = V dO^T.  acc[kt][r] = dP[query = q0 + l31][key = 32 kt + (r & 3) + 8 (r >> 2) + 4 hh]
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (ps) __syncthreads();
        stage(ps * PASS, 2, KP);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPP; ++t) {
            const int kt = ps * TPP + t;
            const char* arow = lds + (t * 32 + l31) * KP + 16 * hh;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const u16x8 vh = *reinterpret_cast<const u16x8*>(arow + s * 32);
                const u16x8 vl = *reinterpret_cast<const u16x8*>(arow + PLANE + s * 32);
                acc[kt] = mma_bf(vl, gh[s], acc[kt]);
                acc[kt] = mma_bf(vh, gl[s], acc[kt]);
                acc[kt] = mma_bf(vh, gh[s], acc[kt]);
            }
        }
    }

    // ---- phase 2: softmax backward in registers.  D = sum_keys P dP (in-lane, then across the two lane halves); dS = P (dP - D)
    const long prow = ((long)bh * T + q0 + l31) * T + 4 * hh;
    float D = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 pv = *reinterpret_cast<const float4*>(probs + prow + 32 * kt + 8 * g);
            D = fmaf(pv.x, acc[kt][4 * g], D); D = fmaf(pv.y, acc[kt][4 * g + 1], D);
            D = fmaf(pv.z, acc[kt][4 * g + 2], D); D = fmaf(pv.w, acc[kt][4 * g + 3], D);
        }
    D += __shfl_xor(D, 32);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 pv = *reinterpret_cast<const float4*>(probs + prow + 32 * kt + 8 * g);
            float4 d;
            d.x = pv.x * (acc[kt][4 * g] - D); d.y = pv.y * (acc[kt][4 * g + 1] - D);
            d.z = pv.z * (acc[kt][4 * g + 2] - D); d.w = pv.w * (acc[kt][4 * g + 3] - D);
            acc[kt][4 * g] = d.x; acc[kt][4 * g + 1] = d.y; acc[kt][4 * g + 2] = d.z; acc[kt][4 * g + 3] = d.w;
            *reinterpret_cast<float4*>(ds + prow + 32 * kt + 8 * g) = d;
        }

    // ---- phase 3: dQ = dS K (alpha at the end).  Same register-to-key permutation as O = P V in the forward
    f32x16 o[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, chalf = 16 * ((lane >> 4) & 1);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        __syncthreads();
        stage(ps * PASS, 1, VP);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPP; ++t) {
            const int kt = ps * TPP + t;
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                float pv[8];
#pragma unroll
                for (int jx = 0; jx < 8; ++jx) pv[jx] = acc[kt][8 * sk + jx];
                u16x8 ph, pl;
                split8_bf(pv, ph, pl);
                const int krow = t * 32 + 16 * sk + 4 * hh + q4;
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const char* src = lds + krow * VP + (j * 32 + chalf + 4 * p4) * 2;
                    u16x8 kh, kl;
                    {
                        const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src));
                        const fp16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + 8 * VP));
                        const u16x4 a4 = __builtin_bit_cast(u16x4, a), c4 = __builtin_bit_cast(u16x4, c);
                        kh[0] = a4[0]; kh[1] = a4[1]; kh[2] = a4[2]; kh[3] = a4[3]; kh[4] = c4[0]; kh[5] = c4[1]; kh[6] = c4[2]; kh[7] = c4[3];
                    }
                    {
                        const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + PLANE));
                        const fp16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + PLANE + 8 * VP));
                        const u16x4 a4 = __builtin_bit_cast(u16x4, a), c4 = __builtin_bit_cast(u16x4, c);
                        kl[0] = a4[0]; kl[1] = a4[1]; kl[2] = a4[2]; kl[3] = a4[3]; kl[4] = c4[0]; kl[5] = c4[1]; kl[6] = c4[2]; kl[7] = c4[3];
                    }
                    o[j] = mma_bf(pl, kh, o[j]);
                    o[j] = mma_bf(ph, kl, o[j]);
                    o[j] = mma_bf(ph, kh, o[j]);
                }
            }
        }
    }

    // ---- phase 4: dQ[query row][ch col] -> dqkv[b][q0 + row][h * 3 CH + col]   (the q slice of the head)
    float* const obase = dqkv + ((long)b * T + q0) * C3 + (long)h * 3 * CH;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float v = o[j][r] * alpha;
            obase[(long)row * C3 + j * 32 + l31] = v;
            if (!__builtin_isfinite(v) && range_flag) *range_flag = 1;
        }
}

template <int CH, int NKT>
int launch_attn_bwd_q(const float* qkv, const float* probs, const float* dout, float* dqkv, float* ds, int B, int heads, hipStream_t st) {
    constexpr int T = 32 * NKT, WAVES = (ATT_WAVES8 && NKT >= 8 && CH <= 96) ? 8 : NKT >= 4 ? 4 : NKT, PASS = T < 128 ? T : 128;
    constexpr int KP = CH * 2 + 16, VP = att_vp(CH);
    constexpr size_t smem = 2 * (size_t)PASS * (KP > VP ? KP : VP);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<CH, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    hipLaunchKernelGGL((attn_bwd_q_kernel<CH, NKT>), dim3(T / (32 * WAVES), B * heads), dim3(64 * WAVES), smem, st, qkv, probs, dout, dqkv, ds, heads,
                       1.f / sqrtf((float)CH), cdae_range_flag_ptr());
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("attn_bwd_q launch failed");
}

template <int CH, int NKT, bool PROBS>
int launch_attn(const float* qkv, float* out, float* probs, int B, int heads, hipStream_t st) {
    constexpr int T = 32 * NKT, WAVES = (ATT_WAVES8 && NKT >= 8 && CH <= 96) ? 8 : NKT >= 4 ? 4 : NKT, PASS = T < 128 ? T : 128;
    constexpr int KP = CH * 2 + 16, VP = att_vp(CH);
    constexpr size_t smem = 2 * (size_t)PASS * (KP > VP ? KP : VP);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fused_kernel<CH, NKT, PROBS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return cdae_fail("hipFuncSetAttribute(max dynamic LDS) failed");
        attr_done = true;
    }
    hipLaunchKernelGGL((attn_fused_kernel<CH, NKT, PROBS>), dim3(T / (32 * WAVES), B * heads), dim3(64 * WAVES), smem, st, qkv, out, probs, heads,
                       1.f / sqrtf((float)CH), cdae_range_flag_ptr());
    return hipGetLastError() == hipSuccess ? 0 : cdae_fail("attn_fused launch failed");
}

}  // namespace

extern "C" int cdae_qkv_attention_fused_supported(int T, int ch) {
    return (T == 64 || T == 256) && (ch == 64 || ch == 96 || ch == 128);
}

// probs != nullptr: also writes the softmax probabilities [B * heads][T][T] (the training forward; one launch instead of GEMM, softmax, GEMM)
extern "C" int cdae_qkv_attention_fwd_fused_p(const float* qkv, float* out, float* probs, int B, int T, int heads, int ch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if ((((size_t)qkv) & 15) || (((size_t)probs) & 15) || !cdae_qkv_attention_fused_supported(T, ch))
        return cdae_fail("attention_fwd_fused: unsupported shape (T in {64, 256}, ch in {64, 96, 128})");
    cdae_prof_begin(PROF_IGEMM, 4.0 * B * heads * (double)T * T * ch, st);
    int rc;
#define ATT(CHV, NK) rc = probs ? launch_attn<CHV, NK, true>(qkv, out, probs, B, heads, st) : launch_attn<CHV, NK, false>(qkv, out, nullptr, B, heads, st)
    if (T == 256) { if (ch == 64) ATT(64, 8); else if (ch == 96) ATT(96, 8); else ATT(128, 8); }
    else { if (ch == 64) ATT(64, 2); else if (ch == 96) ATT(96, 2); else ATT(128, 2); }
#undef ATT
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}

extern "C" int cdae_qkv_attention_fwd_fused(const float* qkv, float* out, int B, int T, int heads, int ch, void* stream) {
    return cdae_qkv_attention_fwd_fused_p(qkv, out, nullptr, B, T, heads, ch, stream);
}

// dS (into `ds`, [B * heads][T][T]) and dQ (into the q slices of dqkv) from the forward's probabilities and dout — the query side of the
// attention backward (reference: autograd through unet.py:239-253) in one launch.  Shapes as cdae_qkv_attention_fused_supported.
extern "C" int cdae_qkv_attention_bwd_q_fused(const float* qkv, const float* probs, const float* dout, float* dqkv, float* ds, int B, int T, int heads,
                                              int ch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (((((size_t)qkv) | ((size_t)probs) | ((size_t)dout) | ((size_t)dqkv) | ((size_t)ds)) & 15) || !cdae_qkv_attention_fused_supported(T, ch))
        return cdae_fail("attention_bwd_q_fused: unsupported shape (T in {64, 256}, ch in {64, 96, 128}) or unaligned operands");
    cdae_prof_begin(PROF_IGEMM, 4.0 * B * heads * (double)T * T * ch, st);
    int rc;
#define ATB(CHV, NK) rc = launch_attn_bwd_q<CHV, NK>(qkv, probs, dout, dqkv, ds, B, heads, st)
    if (T == 256) { if (ch == 64) ATB(64, 8); else if (ch == 96) ATB(96, 8); else ATB(128, 8); }
    else { if (ch == 64) ATB(64, 2); else if (ch == 96) ATB(96, 2); else ATB(128, 2); }
#undef ATB
    cdae_prof_end(PROF_IGEMM, st);
    return rc;
}
