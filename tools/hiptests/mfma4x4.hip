// Dev probe: lane mapping of v_mfma_f32_4x4x1_16b_f32 (16 independent 4 x 4 x 1 outer products per instruction; head.hip uses it for the
// UNet output head: 64 pixels x 4 output channels per instruction, one input channel per issue).
// hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4 && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f * (1 + l); }      // a = 1 + lane, b = 100 (1 + lane): d / 100 = (1 + la)(1 + lb)
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            // hypothesis: block = l / 4, column j = l % 4 (from the B lane 4 block + j), row r (from the A lane 4 block + r)
            const float want = (1.f + 4 * (l / 4) + r) * 100.f * (1.f + l);
            if (hd[l * 4 + r] != want) { if (ok) printf("lane %d reg %d: got %g want %g\n", l, r, hd[l * 4 + r], want); ok = 0; }
        }
    printf("mapping D[lane = 4 block + j][reg = i] = A[lane 4 block + i] * B[lane 4 block + j]: %s\n", ok ? "confirmed" : "NOT as assumed");
    return ok ? 0 : 1;
}
