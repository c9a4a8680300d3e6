// Dev microbenchmark: the K-step body of convwin_kernel (convwin.hip) in isolation, feature by feature, to find where the loop falls from
// the sustained MFMA rate (tools/hiptests/mfma_peak.hip: 1936 TFLOP/s for v_mfma_f32_16x16x32_f16 at two waves per SIMD) to the 0.57 of the
// nominal roof the in-kernel ablation measured.  Same geometry: 256 threads, 80 KB of LDS (two blocks per CU), 96 MFMAs per wave and step
// (8 row tiles x 4 column tiles x 3 products), 24 ds_read_b128 per wave and step.
//   hipcc --offload-arch=gfx950 -O3 cw_loop.hip -o cw_loop && ./cw_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int A_PLANE = 12288, A_HALF = 2 * A_PLANE, B_BASE = 2 * A_HALF, B_KH = 4096, B_PLANE = 2 * B_KH, B_STAGE = 2 * B_PLANE, LDS = B_BASE + 2 * B_STAGE;

template <int IMM>
__device__ __forceinline__ u32x4 lds_read(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
__device__ __forceinline__ f32x4 mma(const u32x4& x, const u32x4& y, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
}
__device__ __forceinline__ void dma(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane((int)lds_dst)), "v"(voff), "s"(sbase) : "memory");
}

// MODE bits: 1 LDS fragment reads (counted waits) as in the kernel, 2 per-tile address arithmetic (mask select), 4 mid-step barrier,
// 8 weight DMAs (4 per wave and step, 16 KB per block and step from an L2-resident buffer) with the mid-step vmcnt wait,
// 16 setprio alternation, 32 reads one tile EARLIER (two A pairs in flight)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const unsigned* __restrict__ src, float* out, int steps, const char* wbuf, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LDS / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = src[i & 16383];
    __syncthreads();
    const int lr = lane & 15, kg = lane >> 4, pc = kg & 1;
    const bool selb = kg >= 2;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned a_lane = (wm * 128 + lr) * 32 + pc * 16;
    const unsigned b_lane = B_BASE + (selb ? B_KH : 0) + (wn * 64 + lr) * 32 + pc * 16;
    int tapmask[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) tapmask[i] = 0x1ff ^ ((lane + i) & 1 ? 0 : (src[lane + i] & 0x49));      // mostly-set masks, data dependent
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 bh[4], bl[4], ah[2], al[2];
    const unsigned woff = (wave * 32 + (lane >> 1)) * 32u + (lane & 1) * 16u;
    unsigned a_c = a_lane, b_c = b_lane;
    int wtap_c = 0;
#define READ_A(I, BUF, WTAP, ADDR) { unsigned ad_ = ADDR; if (MODE & 2) { const unsigned m_ = (unsigned)__builtin_amdgcn_sbfe(tapmask[I], WTAP, 1); ad_ = (ADDR & m_) | (0x0003C000u & ~m_); } \
        ah[BUF] = lds_read<(I) * 512>(ad_); al[BUF] = lds_read<(I) * 512 + A_PLANE>(ad_); }
#define READ_B(J, ADDR) { bh[J] = lds_read<(J) * 512>(ADDR); bl[J] = lds_read<(J) * 512 + B_PLANE>(ADDR); }
    if (MODE & 1) {
        READ_A(0, 0, wtap_c, a_c);
        READ_B(0, b_c); READ_B(1, b_c); READ_B(2, b_c); READ_B(3, b_c);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { bh[j] = *reinterpret_cast<u32x4*>(lds + b_lane + j * 512); bl[j] = *reinterpret_cast<u32x4*>(lds + b_lane + j * 512 + B_PLANE); }
        ah[0] = ah[1] = *reinterpret_cast<u32x4*>(lds + a_lane); al[0] = al[1] = *reinterpret_cast<u32x4*>(lds + a_lane + A_PLANE);
    }
    if (MODE & 8) { for (int q = 0; q < 4; ++q) dma(wbuf, woff + q * 4096, B_BASE + B_STAGE + wave * 1024 + (q & 1) * B_KH + (q >> 1) * B_PLANE); }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        const int t = s % 9;
        const int wtap_n = (t + 2) % 9;
        const unsigned a_n = a_lane + ((s >> 2) & 1) * A_HALF + ((wtap_n / 3) * 64 + wtap_n % 3) * 32;
        const unsigned b_n = b_lane + ((s + 1) & 1) * B_STAGE;
        if (MODE & 16) { if (s & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cur = i & 1;
            if (i == 4) {
                if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (MODE & 4) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (MODE & 8) { for (int q = 0; q < 4; ++q) dma(wbuf, woff + q * 4096 + ((s * 16384) & 0xFFFFF), B_BASE + (s & 1) * B_STAGE + wave * 1024 + (q & 1) * B_KH + (q >> 1) * B_PLANE); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE & 1) {
                if (MODE & 32) {
                    if (i == 0) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));
                    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[cur]), "+v"(al[cur]));
                } else {
                    if (i == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[cur]), "+v"(al[cur]));
                }
            }
            acc[i][0] = mma(al[cur], bh[0], acc[i][0]);
            acc[i][0] = mma(ah[cur], bl[0], acc[i][0]);
            acc[i][0] = mma(ah[cur], bh[0], acc[i][0]);
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE & 1) && !(MODE & 32)) {
                if (i == 0) { READ_A(1, 1, wtap_c, a_c); }
                else if (i == 1) { READ_A(2, 0, wtap_c, a_c); }
                else if (i == 2) { READ_A(3, 1, wtap_c, a_c); }
                else if (i == 3) { READ_A(4, 0, wtap_c, a_c); }
                else if (i == 4) { READ_A(5, 1, wtap_c, a_c); }
                else if (i == 5) { READ_A(6, 0, wtap_c, a_c); }
                else if (i == 6) { READ_A(7, 1, wtap_c, a_c); }
                else { READ_A(0, 0, wtap_n, a_n); READ_B(0, b_n); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                if ((MODE & 1) && i == 0) {
                    if (MODE & 32) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[j]), "+v"(bl[j]) : "n"(8 - 2 * j));
                    else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[j]), "+v"(bl[j]) : "n"(8 - 2 * j));
                }
                acc[i][j] = mma(al[cur], bh[j], acc[i][j]);
                acc[i][j] = mma(ah[cur], bl[j], acc[i][j]);
                acc[i][j] = mma(ah[cur], bh[j], acc[i][j]);
                if ((MODE & 1) && i == 7) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == 1) { READ_B(1, b_n); } else if (j == 2) { READ_B(2, b_n); } else { READ_B(3, b_n); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wtap_c = wtap_n; a_c = a_n; b_c = b_n;
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && clk) { clk[(blockIdx.x * 4 + wave) * 2] = t1 - t0; clk[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = sum;
}

unsigned long long* g_clk;
template <int MODE>
void run(const unsigned* src, float* out, const char* wbuf, const char* label) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 4000, blocks = 512;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), LDS, 0, src, out, steps, wbuf, g_clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = (double)blocks * 4 * steps * 96 * 2.0 * 16 * 16 * 32;
    static unsigned long long hc[512 * 4 * 2];
    hipMemcpy(hc, g_clk, sizeof(hc), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < 512 * 4; ++i) { cyc += hc[2 * i]; rt += hc[2 * i + 1]; }
    printf("   in-kernel: %.0f shader cycles per step and wave, clock %.3f GHz\n", cyc / (512 * 4) / steps, cyc / rt * 0.1);
    printf("mode %2d  %-78s %8.3f ms  %6.0f TFLOP/s f16 MFMA = %5.0f f16x3-equivalent = %.3f of 833\n", MODE, label, best, fl / best / 1e9, fl / best / 3e9, fl / best / 3e9 / 833.3);
}

int main() {
    unsigned* src; float* out; char* wbuf;
    hipMalloc(&src, 16384 * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&wbuf, 2 << 20);
    unsigned h[16384];
    for (int i = 0; i < 16384; ++i) {
        const _Float16 a = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f), b = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);
        h[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&g_clk, 512 * 4 * 2 * 8);
    for (int off = 0; off < (2 << 20); off += sizeof(h)) hipMemcpy(wbuf + off, h, sizeof(h), hipMemcpyHostToDevice);      // random operands: zeros would flatter the clock
    run<0>(src, out, wbuf, "MFMAs only (operands in registers)");
    run<1>(src, out, wbuf, "+ fragment reads from LDS, counted waits");
    run<3>(src, out, wbuf, "+ per-tile mask select of the read address");
    run<19>(src, out, wbuf, "+ setprio alternation");
    run<7>(src, out, wbuf, "reads + masks + mid-step barrier");
    run<15>(src, out, wbuf, "reads + masks + barrier + weight DMAs (L2-resident source)");
    run<31>(src, out, wbuf, "reads + masks + barrier + DMAs + setprio  (= the kernel's K loop)");
    return 0;
}
