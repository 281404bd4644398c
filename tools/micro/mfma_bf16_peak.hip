// Sustained v_mfma_f32_32x32x16_bf16 rate: NACC accumulators per wave, CH consecutive MFMAs per accumulator (a chain on one
// accumulator, as gconv_split.hip issues its six terms), one wave per SIMD, random operands, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_peak mfma_bf16_peak.hip ; run: ./mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int CH>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float scale, unsigned long long* clk) {
    extern __shared__ float lds_pad[];      // only to pin the number of workgroups per CU
    if (iters < 0) lds_pad[threadIdx.x] = a;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 av[6], bv[6];
    for (int q = 0; q < 6; ++q)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; av[q][j] = (__bf16)(((float)(h >> 8) * (1.f / 16777216.f) - 0.5f + a) * scale);
            h = h * 1664525u + 1013904223u; bv[q][j] = (__bf16)(((float)(h >> 8) * (1.f / 16777216.f) - 0.5f + a) * scale);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(i + c) % 6], bv[(i * 3 + c + 1) % 6], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float scale, unsigned long long* clk) {
    extern __shared__ float lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = a;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 av[6], bv[6];
    for (int q = 0; q < 6; ++q)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; av[q][j] = (__bf16)(((float)(h >> 8) * (1.f / 16777216.f) - 0.5f + a) * scale);
            h = h * 1664525u + 1013904223u; bv[q][j] = (__bf16)(((float)(h >> 8) * (1.f / 16777216.f) - 0.5f + a) * scale);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i % 6], bv[(i * 3 + 1) % 6], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}
template <int NACC>
void run16(int wgs, int iters, int reps, const char* tag, float scale = 1.f, int lds = 100 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k16<NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float* out; hipMalloc(&out, 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k16<NACC><<<wgs, 256, lds>>>(out, iters, 0.f, scale, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k16<NACC><<<wgs, 256, lds>>>(out, iters, 0.f, scale, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)reps * wgs * 4 * iters * NACC;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("16x16x32 %-24s NACC=%d wgs=%d scale=%g: %8.2f ms  %7.1f TFLOP/s | wave 0: %.1f shader clk per MFMA, shader clock %.2f GHz\n", tag, NACC, wgs, scale, ms,
           n * 16384.0 / ms / 1e9, (double)hc[0] / ((double)iters * NACC), (double)hc[0] / ((double)hc[1] * 10.0));
}
template <int NACC, int CH>
void run(int wgs, int iters, int reps, const char* tag, float scale = 1.f, int lds = 100 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float* out; hipMalloc(&out, 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, CH><<<wgs, 256, lds>>>(out, iters, 0.f, scale, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<NACC, CH><<<wgs, 256, lds>>>(out, iters, 0.f, scale, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)reps * wgs * 4 * iters * NACC * CH;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-28s NACC=%d CH=%d wgs=%d scale=%g: %8.2f ms  %7.1f TFLOP/s | wave 0: %.1f shader clk per MFMA, shader clock %.2f GHz\n", tag, NACC, CH, wgs, scale, ms,
           n * 32768.0 / ms / 1e9, (double)hc[0] / ((double)iters * NACC * CH), (double)hc[0] / ((double)hc[1] * 10.0) );
    hipFree(out);
}
int main() {
    run<6, 1>(256, 4000, 100, "1 WG/CU, rotating");
    run<6, 6>(256, 700, 100, "1 WG/CU, chains of 6");
    run<6, 1>(512, 4000, 100, "2 WG/CU, rotating", 1.f, 70 * 1024);
    run<6, 6>(512, 700, 100, "2 WG/CU, chains of 6", 1.f, 70 * 1024);
    run<6, 1>(256, 4000, 100, "1 WG/CU, rotating, zeros", 0.f);
    run<6, 1>(512, 4000, 100, "2 WG/CU, rotating, zeros", 0.f, 70 * 1024);
    run<3, 1>(256, 8000, 100, "1 WG/CU, 3 accs rotating");
    run<2, 1>(256, 12000, 100, "1 WG/CU, 2 accs rotating");
    run16<12>(256, 4000, 100, "1 WG/CU");
    run16<12>(512, 4000, 100, "2 WG/CU", 1.f, 70 * 1024);
    run16<12>(256, 4000, 100, "1 WG/CU zeros", 0.f);
    run16<24>(256, 2000, 100, "1 WG/CU 24 accs");
    return 0;
}
