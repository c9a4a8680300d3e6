// Dev microbenchmark (round 6): the K-step body of a ONE-PLANE window conv (the 16-bit torso's form of convwin_kernel: one bf16 MFMA per
// product) for different WAVE TILES — what a redesign of that kernel would gain from a larger register tile, measured instead of reasoned.
// Per K step (K = 32 per v_mfma_f32_16x16x32_bf16) a wave reads TM activation fragments + TN weight fragments (ds_read_b128, 1 KB each per
// wave) and issues TM x TN MFMAs; one block barrier per step; fragments are prefetched one 16-row tile ahead with counted lgkmcnt waits, the
// weight fragments of the next step behind their last use — the structure of convwin.hip's loop without its DMAs and address masks.
//   wave tile 128 x 64  (TM 8,  TN 4): 12 reads / 32 MFMAs, two blocks per CU (today's kernel: 75 % of the LDS pipe at the nominal MFMA rate)
//   wave tile 128 x 128 (TM 8,  TN 8): 16 reads / 64 MFMAs, 256 accumulator registers -> one block (one wave per SIMD) per CU
//   wave tile 256 x 64  (TM 16, TN 4): 20 reads / 64 MFMAs, one block per CU
//   hipcc --offload-arch=gfx950 -O3 cw1_tile.hip -o cw1_tile && ./cw1_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_BYTES = 64 * 1024;

__device__ __forceinline__ u32x4 lds_read(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ f32x4 mma(const u32x4& x, const u32x4& y, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0);
}

// BAR: one block barrier per step (mid-step, as in the kernel); READS: 0 = no LDS traffic in the loop (the MFMA ceiling of the tile shape)
template <int TM, int TN, int MINB, bool BAR, bool READS>
__global__ __launch_bounds__(256, MINB) void k(const unsigned* __restrict__ src, float* out, int steps, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LDS_BYTES / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = src[i & 16383] & 0x3f803f80u;      // small finite bf16 values
    __syncthreads();
    const int lr = lane & 15, kg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // 32-byte rows, two 16-byte pieces per row (the kernel's unswizzled layout); lanes 32-63 read the step's second unit 8 KB further
    const unsigned a_lane = ((wm * TM * 16 + lr) * 32 + (kg & 1) * 16 + (kg >> 1) * 8192) & (LDS_BYTES / 2 - 1);
    const unsigned b_lane = LDS_BYTES / 2 + (((wn * TN * 16 + lr) * 32 + (kg & 1) * 16 + (kg >> 1) * 8192) & (LDS_BYTES / 2 - 1));
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 b[TN], a[2];
    auto addr_a = [&](int i, int s) { return (a_lane + i * 512 + (s & 7) * 64) & (LDS_BYTES / 2 - 1); };            // a tap shift per step
    auto addr_b = [&](int j, int s) { return LDS_BYTES / 2 + ((b_lane + j * 512 + (s & 1) * 16384) & (LDS_BYTES / 2 - 1)); };
    if (READS) {
        a[0] = lds_read(addr_a(0, 0));
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = lds_read(addr_b(j, 0));
    } else {
        a[0] = a[1] = u32x4{0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = u32x4{0x3f803f80u + j, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cur = i & 1;
            if (BAR && i == TM / 2) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            if (READS) {
                if (i == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(TN - 1));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur]));
            }
            acc[i][0] = mma(a[cur], b[0], acc[i][0]);
            __builtin_amdgcn_sched_barrier(0);
            if (READS) {
                if (i + 1 < TM) a[cur ^ 1] = lds_read(addr_a(i + 1, s));
                else { a[0] = lds_read(addr_a(0, s + 1)); b[0] = lds_read(addr_b(0, s + 1)); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < TN; ++j) {
                if (READS && i == 0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[j]) : "n"(TN - j));
                acc[i][j] = mma(a[cur], b[j], acc[i][j]);
                if (READS && i == TM - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    b[j] = lds_read(addr_b(j, s + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int TM, int TN, int MINB, bool BAR, bool READS>
static void run(const char* name, const unsigned* src, float* out, unsigned long long* clk) {
    const int blocks = 256 * MINB, steps = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<TM, TN, MINB, BAR, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TM, TN, MINB, BAR, READS>), dim3(blocks), dim3(256), LDS_BYTES, 0, src, out, steps, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * steps * TM * TN * 16384.0;
    const double tf = flops / (best * 1e-3) / 1e12;
    const double ghz = (double)c / (best * 1e-3) / 1e9;      // s_memtime ticks at 100 MHz on this part would read 0.1; readcyclecounter = shader clock
    printf("%-46s  %7.3f ms  %7.1f TFLOP/s  = %.3f of 2500   reads/MFMA %.3f   LDS KB per wave-step %2d   (cycle counter / time = %.2f GHz)\n", name, best, tf, tf / 2500.0,
           (double)(TM + TN) / (TM * TN), TM + TN, ghz);
}

int main() {
    unsigned* src;
    float* out;
    unsigned long long* clk;
    hipMalloc(&src, 65536);
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&clk, 8);
    unsigned* h = (unsigned*)malloc(65536);
    unsigned seed = 1;
    for (int i = 0; i < 16384; ++i) { seed = seed * 1664525u + 1013904223u; h[i] = seed; }
    hipMemcpy(src, h, 65536, hipMemcpyHostToDevice);
    printf("one-plane window-conv K loop by wave tile (v_mfma_f32_16x16x32_bf16, 4 waves per block)\n");
    run<8, 4, 2, true, false>("128 x 64, 2 blocks/CU, MFMAs only", src, out, clk);
    run<8, 4, 2, true, true>("128 x 64, 2 blocks/CU (today's shape)", src, out, clk);
    run<8, 4, 2, false, true>("128 x 64, 2 blocks/CU, no barrier", src, out, clk);
    run<8, 4, 1, true, true>("128 x 64, 1 block/CU (a lone block)", src, out, clk);
    run<8, 8, 1, true, false>("128 x 128, 1 block/CU, MFMAs only", src, out, clk);
    run<8, 8, 1, true, true>("128 x 128, 1 block/CU", src, out, clk);
    run<8, 8, 1, false, true>("128 x 128, 1 block/CU, no barrier", src, out, clk);
    run<16, 4, 1, true, true>("256 x 64, 1 block/CU", src, out, clk);
    run<8, 6, 1, true, true>("128 x 96, 1 block/CU", src, out, clk);
    run<4, 4, 2, true, true>("64 x 64, 2 blocks/CU", src, out, clk);
    return 0;
}
