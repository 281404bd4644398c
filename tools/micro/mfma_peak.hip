// Sustained fp32 MFMA rate of the device: NACC independent 32x32x2 accumulators per wave, 2 waves per SIMD, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    // lane-dependent pseudo-random operands that change every iteration (realistic datapath toggling)
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float av[4], bv[4];
    for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; av[j] = (float)(h >> 8) * (1.f / 16777216.f) - 0.5f + a; h = h * 1664525u + 1013904223u; bv[j] = (float)(h >> 8) * (1.f / 16777216.f) - 0.5f + b; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 3], bv[(i * 3 + 1) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.f) out[0] = s;
}
template <int NACC>
void run(int wgs, int iters, int reps, const char* tag) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<wgs, 256>>>(out, iters, 0.f, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<NACC><<<wgs, 256>>>(out, iters, 0.f, 0.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)reps * wgs * 4 * iters * NACC * 4096.0;
    printf("%s NACC=%d wgs=%d iters=%d reps=%d: %.2f ms  %.1f TFLOP/s\n", tag, NACC, wgs, iters, reps, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    run<6>(512, 2000, 1, "short (~1ms)");
    run<6>(512, 2000, 200, "sustained");
    run<6>(512, 2000, 1000, "sustained-long");
    run<9>(512, 2000, 200, "sustained");
    run<6>(256, 4000, 200, "1 wave/SIMD");
    run<2>(512, 6000, 200, "2 acc only");
    return 0;
}
