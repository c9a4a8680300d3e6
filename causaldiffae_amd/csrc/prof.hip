// prof.hip — error string + opt-in per-kernel-family timing with HIP events on the launch stream.
// Used by bench.py to measure the dominant kernel's average launch duration live (roofline.achieved);
// disabled by default so the timed region and graph capture never see an event record.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <string>
#include <vector>
#include "cdae_internal.h"
#include "../../include/cdae.h"

namespace {
thread_local std::string g_err;
struct Rec { int fam; double work, bytes; hipEvent_t a, b; std::string tag; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local hipEvent_t t_start = nullptr;
thread_local double t_work = 0, t_bytes = 0;
thread_local int t_fam = -1;
thread_local char t_tag[128] = {0};

hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
}  // namespace

int* cdae_range_flag_ptr() {
    static int* host = nullptr;
    static int* dev = nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!host) {
        if (hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocMapped) != hipSuccess) { host = nullptr; return nullptr; }
        host[0] = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) != hipSuccess) dev = host;
    }
    return dev;
}

namespace {
// defaults = the measured optimum on MI355X (dispatch table)
int g_tune[TUNE_N] = {256, 1, 0, 1, 2048, 1, 1, 1, 384, 0, 2, 1, 512, 12, 1};
}
int cdae_tune(int key) { return key >= 0 && key < TUNE_N ? g_tune[key] : 0; }

int cdae_fail(const char* msg) { g_err = msg ? msg : "unknown"; return -1; }

void cdae_prof_begin(int fam, double work, hipStream_t st) {
    if (!g_on) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;      // a launch being captured into a graph has no time of its own
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(g_mu);
    t_start = get_event();
    t_work = work; t_bytes = 0; t_fam = fam; t_tag[0] = 0;
    hipEventRecord(t_start, st);
}

// the launch between begin and end belongs to `fam` instead (a sub-family reported on its own) and moves `bytes` algorithmic bytes
void cdae_prof_note(int fam, double bytes) {
    if (!g_on || !t_start) return;
    t_fam = fam; t_bytes = bytes;
}

// dev aid: a free-form label (shape, tile, split) for the launch between begin and end; dumped per launch by cdae_prof_read when
// CDAE_PROF_DUMP names a file
void cdae_prof_tag(const char* tag) {
    if (!g_on || !t_start || !tag) return;
    snprintf(t_tag, sizeof(t_tag), "%s", tag);
}
bool cdae_prof_on() { return g_on; }

void cdae_prof_end(int fam, hipStream_t st) {
    if (!g_on || !t_start) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t e = get_event();
    hipEventRecord(e, st);
    g_recs.push_back({t_fam >= 0 ? t_fam : fam, t_work, t_bytes, t_start, e, std::string(t_tag)});
    t_start = nullptr;
}

namespace {
typedef float calib_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 calib_half8 __attribute__((ext_vector_type(8)));
// register-resident v_mfma_f32_16x16x32_f16 loop, three dependent MFMAs per accumulator like the f16x3 product (the form of
// tools/hiptests/mfma_peak.hip's best case): what THIS box's matrix cores sustain on random operands at the clock the part holds
// under that load.  clk[0] += shader cycles (s_memtime), clk[1] += 100 MHz ticks (s_memrealtime) of one wave per block.
__global__ void __launch_bounds__(64) calib_mfma_kernel(const calib_half8* __restrict__ in, float* __restrict__ out, int iters,
                                                        unsigned long long* __restrict__ clk) {
    calib_half8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], b0 = in[threadIdx.x + 128], b1 = in[threadIdx.x + 192];
    calib_f32x4 c[8] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c[q], 0, 0, 0);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c[q], 0, 0, 0);
            c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[q], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) s += c[q][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&clk[0], t1 - t0); atomicAdd(&clk[1], r1 - r0); }
}
// flat copy with U 16-byte loads in flight per lane before the first store (a block moves U x 4 KiB per trip; the tail trip is guarded).
// NT: nontemporal loads and stores (no reuse: keeps the stream out of the MALL's way).
template <int U, bool NT>
__global__ void __launch_bounds__(256) calib_copy_kernel(const calib_f32x4* __restrict__ src, calib_f32x4* __restrict__ dst, long n4) {
    const long chunk = 256L * U;
    for (long base = (long)blockIdx.x * chunk; base < n4; base += (long)gridDim.x * chunk) {
        calib_f32x4 v[U];
        const long i0 = base + threadIdx.x;
        if (base + chunk <= n4) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i0 + 256L * u) : src[i0 + 256L * u];
#pragma unroll
            for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i0 + 256L * u); else dst[i0 + 256L * u] = v[u]; }
        } else {
            for (int u = 0; u < U; ++u) if (i0 + 256L * u < n4) dst[i0 + 256L * u] = src[i0 + 256L * u];
        }
    }
}
template <int U, bool NT>
static double calib_copy_run(const void* src, void* dst, long n4, int blocks, int reps, hipEvent_t e0, hipEvent_t e1, hipStream_t st) {
    hipLaunchKernelGGL((calib_copy_kernel<U, NT>), dim3(blocks), dim3(256), 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n4);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((calib_copy_kernel<U, NT>), dim3(blocks), dim3(256), 0, st, (const calib_f32x4*)src, (calib_f32x4*)dst, n4);
    hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * 16.0 * (double)n4 * reps / (ms * 1e-3) / 1e12;
}
}  // namespace

extern "C" {

// Box calibration for bench.py (outside its timed regions): the sustained f16 MFMA rate and the shader clock under that load.
// scratch: >= 4 MiB + 4 KiB of device memory the call may overwrite.
int cdae_calib_mfma(void* scratch, size_t scratch_bytes, int iters, double* tflops, double* sclk_ghz, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int blocks = 256 * 4 * 2;                       // two waves per SIMD, as the window conv kernel runs
    const size_t need = 4096 + 16 + (size_t)blocks * 64 * 4;
    if (!scratch || scratch_bytes < need || iters <= 0 || !tflops || !sclk_ghz) return cdae_fail("calib_mfma: scratch too small or bad arguments");
    _Float16 h[256 * 8];
    unsigned seed = 12345u;
    for (int i = 0; i < 256 * 8; ++i) { seed = seed * 1664525u + 1013904223u; h[i] = (_Float16)((((seed >> 8) & 0xffff) / 65536.f - 0.5f) * 0.05f); }
    char* base = (char*)scratch;
    unsigned long long* clk = (unsigned long long*)(base + 4096);
    float* out = (float*)(base + 4096 + 16);
    if (hipMemcpyAsync(base, h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess) return cdae_fail("calib_mfma: upload failed");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    unsigned long long hclk[2] = {0, 0};
    for (int rep = 0; rep < 2; ++rep) {                   // the first launch warms the clocks up
        hipMemsetAsync(clk, 0, 16, st);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(64), 0, st, (const calib_half8*)base, out, iters, clk);
        hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return cdae_fail("calib_mfma: kernel failed"); }
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
    hipEventDestroy(e0); hipEventDestroy(e1);
    *tflops = (double)blocks * iters * 24 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
    *sclk_ghz = hclk[1] ? (double)hclk[0] / (double)hclk[1] * 0.1 : 0.0;
    return 0;
}

// HBM copy rate (read + write bytes per second) of a flat 16-byte-per-lane copy src -> dst of `bytes` bytes (use >= 1 GiB: beyond the
// 256 MiB MALL): the best of a few forms (1 / 4 / 8 loads in flight per lane, plain or nontemporal, grid = 8 or 16 blocks per CU) — the
// number stands for what the box's memory system delivers, not for one kernel shape (MI355X_MICROARCH.md measures 6.29 TB/s).
int cdae_calib_copy(const void* src, void* dst, size_t bytes, int reps, double* tbps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!src || !dst || bytes < 4096 || bytes % 16 || reps <= 0 || !tbps) return cdae_fail("calib_copy: bad arguments");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long n4 = (long)(bytes / 16);
    double best = 0.0;
    bool bad = false;
    auto take = [&](double v) { if (v < 0) bad = true; else if (v > best) best = v; };
    for (int blocks : {256 * 8, 256 * 16}) {
        take(calib_copy_run<1, false>(src, dst, n4, blocks, reps, e0, e1, st));
        take(calib_copy_run<4, false>(src, dst, n4, blocks, reps, e0, e1, st));
        take(calib_copy_run<8, false>(src, dst, n4, blocks, reps, e0, e1, st));
        take(calib_copy_run<4, true>(src, dst, n4, blocks, reps, e0, e1, st));
        take(calib_copy_run<8, true>(src, dst, n4, blocks, reps, e0, e1, st));
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (bad) return cdae_fail("calib_copy: kernel failed");
    *tbps = best;
    return 0;
}

const char* cdae_last_error(void) { return g_err.c_str(); }

int cdae_version(void) { return CDAE_VERSION; }

// *nonfinite = number-ish (0 / 1) of contraction results that were not finite since the last call, then cleared.  The caller
// synchronises the streams it cares about first (the flag lives in pinned host memory the kernels write to directly).
int cdae_range_status(int* nonfinite) {
    int* f = cdae_range_flag_ptr();
    if (!f || !nonfinite) return cdae_fail("range_status: no flag");
    volatile int* v = f;
    *nonfinite = *v;
    *v = 0;
    return 0;
}

int cdae_tune_set(int key, int value) {
    if (key < 0 || key >= TUNE_N) return cdae_fail("tune_set: unknown key");
    if (value < 0 && key != TUNE_CONVWIN_NJ3 && key != TUNE_CONVWIN_NJ2) return cdae_fail("tune_set: value must be >= 0");
    g_tune[key] = value;
    return 0;
}
int cdae_tune_get(int key) { return key >= 0 && key < TUNE_N ? g_tune[key] : -1; }

int cdae_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}

// Synchronises the device, folds all recorded launches into per-family totals and clears the log.
// ms / work (flops) / bytes (algorithmic, where the dispatcher states them) / launches are arrays of CDAE_PROF_FAMILIES entries.
int cdae_prof_read(double* ms, double* work, double* bytes, long long* launches) {
    if (hipDeviceSynchronize() != hipSuccess) return cdae_fail("hipDeviceSynchronize failed");
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < PROF_NFAM; ++i) { ms[i] = 0; work[i] = 0; bytes[i] = 0; launches[i] = 0; }
    FILE* dump = getenv("CDAE_PROF_DUMP") ? fopen(getenv("CDAE_PROF_DUMP"), "a") : nullptr;
    for (auto& r : g_recs) {
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        if (dump) fprintf(dump, "%d\t%.1f\t%.6g\t%s\n", r.fam, t * 1e3, r.work, r.tag.c_str());
        ms[r.fam] += t; work[r.fam] += r.work; bytes[r.fam] += r.bytes; launches[r.fam] += 1;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    if (dump) fclose(dump);
    g_recs.clear();
    return 0;
}

}  // extern "C"
