// Shared device helpers of the contraction kernels (igemm.hip: fp32-operand implicit GEMM + the dispatcher; planes.hip: the
// first-generation kernels on pre-split 16-bit planes).  Internal, not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "cdae_internal.h"
#include "../../include/cdae.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// dev ablations of the plane kernels (CDAE_PS_DBG bits) exist only in a -DCW_DEV=1 build (EXTRA_HIPCC_FLAGS=-DCW_DEV=1 build.sh): as
// run-time tests they sat in every K step of the production kernels
#define PDBG(P) (CW_DEV ? (P).dbg : 0)

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;     // row pitch (floats) of a K-contiguous LDS tile

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- conv gather geometry for one GEMM row (an output pixel, or for wgrad a reduction pixel)
struct PixRow {
    long base;     // element offset of image n
    int iy0, ix0;  // top-left input coordinate of the 3x3 window (already * stride - pad)
    int ok;        // row < M
};

// exact floor(n / d) for n < 2^31 with a host-computed (magic, shift): (mulhi(n, magic) + n) >> shift
__device__ __forceinline__ int fdiv(int n, unsigned magic, int shift) {
    return (int)((__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}

// Everything below is straight-line (no branches, no early returns): any control flow in the gather geometry ends
// up between the tile loads and makes hipcc serialise them with vmcnt waits.  The mode switches (stride / fused
// upsample / transposed stride-2 gather) are folded into host-computed constants g_*.
__device__ __forceinline__ PixRow make_pixrow(const GemmParams& p, int m) {
    PixRow r;
    r.ok = m < p.conv_M;
    const int mm = r.ok ? m : 0;
    const int n = fdiv(mm, p.hw_magic, p.hw_shift);
    const int rem = mm - n * p.hw;
    const int oy = fdiv(rem, p.wo_magic, p.wo_shift), ox = rem - oy * p.Wo;
    r.base = (long)n * p.sn;
    r.iy0 = oy * p.g_mul + p.g_add;         // conv: o*stride - 1;  transposed gather: o + 1
    r.ix0 = ox * p.g_mul + p.g_add;
    return r;
}

// offset (elements) of the input pixel under window tap (ky,kx); false when the tap reads padding
__device__ __forceinline__ bool tap_offset(const GemmParams& p, const PixRow& r, int ky, int kx, long& off) {
    const int ty = r.iy0 + p.g_sign * ky, tx = r.ix0 + p.g_sign * kx;      // transposed gather walks the taps backwards
    const bool ok = ((ty | tx) >= 0) & (((ty | tx) & p.g_pm) == 0) & (ty < (p.H << p.g_sh)) & (tx < (p.W << p.g_sh));
    const int iy = ty >> p.g_sh, ix = tx >> p.g_sh;                       // >>1: fused upsample source / stride-2 transpose
    off = ok ? r.base + (long)iy * p.sy + (long)ix * p.sx : 0;
    return ok;
}

// Branch-free guarded loads.  A conditional `ok ? load : 0` makes hipcc branch around every load and wait
// vmcnt(0) before the next one (all tile loads of a K-step serialised), and a value select after the load drags
// the vmcnt wait in front of the MFMAs.  Selecting the ADDRESS instead (a zero-filled device constant when !ok)
// keeps the loads unconditional, back-to-back and un-waited until the LDS store after the MFMAs.
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // non-const: stays in the global address space (a constant-space pointer turns the select into flat loads)

// (zero: the caller's register copy of g_zero16 — see igemm_kernel)
__device__ __forceinline__ float4 ld4_if(const float* zero, const float* p, bool ok) {
    return ld4(ok ? p : zero);
}
__device__ __forceinline__ float ld1_if(const float* zero, const float* p, bool ok) {
    return *(ok ? p : zero);
}

// second output of an epilogue: v as f16 hi/lo planes
__device__ __forceinline__ void store_planes(const GemmParams& p, long addr, float v) {
    asm volatile("" : "+v"(v));        // opaque: no second, differently rounded f16 conversion folded into the producing fma (see attention.hip split8)
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    p.C_hi[addr] = __builtin_bit_cast(unsigned short, h);
    p.C_lo[addr] = __builtin_bit_cast(unsigned short, l);
}

}  // namespace
