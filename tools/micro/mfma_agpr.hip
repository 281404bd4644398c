// Does the accumulator register class matter?  The same MFMA stream with the accumulators in AGPRs (launch bound 256: the
// compiler allocates a[...]) and in VGPRs (launch bound 512), for the fp32 and bf16 shapes this library uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_agpr mfma_agpr.hip ; run: ./mfma_agpr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE, int LB>
__global__ __launch_bounds__(LB) void k(float* out, int iters, float a, unsigned long long* clk) {
    extern __shared__ float lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = a;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float fv[8];
    bf16x8 av, bv;
    for (int j = 0; j < 8; ++j) { h = h * 1664525u + 1013904223u; fv[j] = (float)(h >> 8) * (1.f / 16777216.f) - 0.5f + a; av[j] = (__bf16)fv[j]; bv[j] = (__bf16)(fv[j] * 0.7f); }
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    float s = 0.f;
    if constexpr (SHAPE == 0) {          // v_mfma_f32_32x32x2_f32
        f32x16 acc[6];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[i & 3], fv[4 + (i & 3)], acc[i], 0, 0, 0);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else if constexpr (SHAPE == 1) {   // v_mfma_f32_16x16x4_f32
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[i & 3], fv[4 + (i & 3)], acc[i], 0, 0, 0);
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    } else {                             // v_mfma_f32_32x32x16_bf16
        f32x16 acc[6];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}
template <int SHAPE, int LB>
void run(const char* tag, int per_iter, double flop) {
    const int iters = 3000, reps = 50;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float* out; (void)hipMalloc(&out, 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SHAPE, LB><<<256, 256, 100 * 1024>>>(out, iters, 0.f, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<SHAPE, LB><<<256, 256, 100 * 1024>>>(out, iters, 0.f, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-44s %7.2f ms %7.1f TFLOP/s | %.1f shader clk per MFMA, %.2f GHz\n", tag, ms, (double)reps * 256 * 4 * iters * per_iter * flop / ms / 1e9,
           (double)hc[0] / ((double)iters * per_iter), (double)hc[0] / ((double)hc[1] * 10.0));
}
int main() {
    run<0, 256>("f32 32x32x2, launch bound 256 (AGPR acc?)", 6, 4096.0);
    run<0, 512>("f32 32x32x2, launch bound 512 (VGPR acc)", 6, 4096.0);
    run<1, 256>("f32 16x16x4, launch bound 256 (AGPR acc?)", 12, 2048.0);
    run<1, 512>("f32 16x16x4, launch bound 512 (VGPR acc)", 12, 2048.0);
    run<2, 256>("bf16 32x32x16, launch bound 256 (AGPR acc?)", 6, 32768.0);
    run<2, 512>("bf16 32x32x16, launch bound 512 (VGPR acc)", 6, 32768.0);
    return 0;
}
