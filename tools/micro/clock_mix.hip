// Effective shader clock (s_memtime / s_memrealtime) and TFLOP/s of an MFMA stream mixed with LDS reads and fresh operands:
// which activity makes the device leave 2.4 GHz?   hipcc --offload-arch=gfx950 -O3 -o clock_mix clock_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NB, bool FRESH>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float seed) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) { unsigned h = (i + 1) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lds[i] = seed * 0.f + ((float)(h >> 8) * (1.f / 8388608.f) - 1.f); }   /* full-entropy mantissas in [-1, 1) */
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f32x4 a[3], b[2];
    for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<f32x4*>(&lds[(threadIdx.x & 63) * 4 + i * 256]);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<f32x4*>(&lds[(threadIdx.x & 63) * 4 + 1024 + i * 256]);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned off = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        f32x4 na[3], nb[2];
        if (NB > 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < NB) na[i] = *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds) + ((off + i * 1024 + it * 64) & 0xFFF0));
#pragma unroll
            for (int i = 0; i < 2; ++i) if (i + 3 < NB) nb[i] = *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds) + ((off + 8192 + i * 1024 + it * 64) & 0xFFF0));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][kk], b[n][kk], acc[m * 2 + n], 0, 0, 0);
        if (FRESH && NB > 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < NB) a[i] = na[i];
#pragma unroll
            for (int i = 0; i < 2; ++i) if (i + 3 < NB) b[i] = nb[i];
        } else if (NB > 0) {
            float t = 0.f;
            for (int i = 0; i < 3; ++i) if (i < NB) t += na[i][0];
            for (int i = 0; i < 2; ++i) if (i + 3 < NB) t += nb[i][0];
            if (t == 1.2345f) a[0][0] = t;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (s == 12345.f) out[0] = 0;
}
template <int NB, bool FRESH>
void run(const char* tag) {
    unsigned long long* d; (void)hipMalloc(&d, 512 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 8000;
    for (int w = 0; w < 3; ++w) k<NB, FRESH><<<512, 256>>>(d, iters, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NB, FRESH><<<512, 256>>>(d, iters, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double tf = 5.0 * 512.0 * 4 * iters * 24 * 4096.0 / ms / 1e9;
    printf("%-44s %6.1f TFLOP/s   effective clock %4.0f MHz\n", tag, tf, (double)h[0] / (double)h[1] * 100.0);
    (void)hipFree(d);
}
int main() {
    run<0, false>("24 MFMAs per step, operands fixed");
    run<5, false>("+ 5 ds_read_b128 per step, results unused");
    run<5, true>("+ 5 ds_read_b128 per step feeding the MFMAs");
    run<2, true>("+ 2 ds_read_b128 per step feeding the MFMAs");
    return 0;
}
