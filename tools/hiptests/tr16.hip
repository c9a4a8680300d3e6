// Dev probe: lane -> element map of ds_read_b64_tr_b16 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define PITCH 160
__global__ void k(float* y) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32 * PITCH];
    for (int i = threadIdx.x; i < 32 * PITCH; i += 64) { int r = i / PITCH, c = i % PITCH; lds[i] = (_Float16)(c < 64 ? r * 64 + c : 0); }
    __syncthreads();
    const int lane = threadIdx.x;
    const int q = (lane & 15) >> 2, pp = lane & 3;
    const int C0 = 16 * ((lane >> 4) & 1), R0 = 8 * (lane >> 5);
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lds + (R0 + q) * PITCH + C0 + 4 * pp));
    half4 h = __builtin_bit_cast(half4, v);
    for (int i = 0; i < 4; ++i) y[lane * 4 + i] = (float)h[i];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        int ch = (lane & 15) + 16 * ((lane >> 4) & 1), k0 = 8 * (lane >> 5);
        for (int e = 0; e < 4; ++e) { int want = (k0 + e) * 64 + ch; if ((int)h[lane * 4 + e] != want) ++bad; }
        if (lane < 4 || lane == 17 || lane == 33 || lane == 50) printf("lane %2d: %g %g %g %g  (want row %d.. col %d)\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3], k0, ch);
    }
    printf("mismatches vs expected map: %d\n", bad);
    return 0;
}
