// What does s_memtime count?  Ratio of s_memtime to s_memrealtime (100 MHz) around a pure-MFMA loop whose achieved TFLOP/s
// pins the real shader clock (154 TF <-> 2.35 GHz).  Build: hipcc --offload-arch=gfx950 -O3 -o clock_ratio clock_ratio.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float a, float b) {
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (s == 12345.f) out[0] = 0;
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 512 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 40000;
    for (int w = 0; w < 3; ++w) k<<<512, 256>>>(d, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<512, 256>>>(d, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double tf = 512.0 * 4 * iters * 6 * 4096.0 / ms / 1e9;
    printf("pure MFMA: %.1f TFLOP/s -> shader clock >= %.0f MHz; s_memtime/s_memrealtime*100MHz = %.0f MHz; MFMA issue cycles by s_memtime: %.1f per MFMA\n",
           tf, tf / 157.3 * 2400.0, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / (iters * 6.0));
    return 0;
}
