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
int g_tune[TUNE_N] = {256, 1, 0, 1};
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

extern "C" {

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
    if (value < 0 && key != TUNE_CONVWIN_NJ3) return cdae_fail("tune_set: value must be >= 0");
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
