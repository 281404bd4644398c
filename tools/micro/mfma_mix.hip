// Cost of instructions interleaved with fp32 MFMAs (v_mfma_f32_32x32x2_f32) on gfx950: per step 9 MFMAs on independent
// accumulators + NV VALU ops + ND ds_read_b32 (+ NB ds_read_b128), 2 waves per SIMD.  Prints achieved TFLOP/s and the
// extra clocks per step relative to the bare MFMA stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int ND, int NB, bool BEFORE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float av = threadIdx.x * 0.001f, bv = 1.0f;
    unsigned x = threadIdx.x, y = 3;
    unsigned addr = (threadIdx.x & 63) * 4;
    unsigned addr4 = (threadIdx.x & 63) * 16;
    float r[ND > 0 ? ND : 1];
    f32x4 r4[NB > 0 ? NB : 1];
    for (int it = 0; it < iters; ++it) {
        if (BEFORE) {
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[d]) : "v"(addr), "i"(d * 256));
#pragma unroll
            for (int d = 0; d < NB; ++d) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[d]) : "v"(addr4), "i"(d * 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            if (!BEFORE) {
                if (i < NV) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (i < ND) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i < ND ? i : 0]) : "v"(addr), "i"(i * 256));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ND > 0) av += r[0] * 0.f;
        if (NB > 0) bv += r4[0][0] * 0.f;
    }
    float s = (float)x;
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.f) out[0] = s;
}
template <int NV, int ND, int NB, bool BEFORE>
double run(const char* tag, double base_clk) {
    const int wgs = 512, iters = 3000, reps = 30;
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, ND, NB, BEFORE><<<wgs, 256>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<NV, ND, NB, BEFORE><<<wgs, 256>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)reps * wgs * 4 * iters * 9 * 4096.0;
    double tf = fl / ms / 1e9;
    // clocks per step per SIMD (2 waves share it): peak 154.3 TF <-> 2 * 9 * 64 = 1152 clk per pair of steps
    double clk = 154.3 / tf * 576.0;
    printf("%-34s NV=%2d ND=%2d NB=%d %s: %6.1f TF  %.0f clk/step (+%.0f)\n", tag, NV, ND, NB, BEFORE ? "before" : "interl", tf, clk, clk - base_clk);
    (void)hipFree(out);
    return clk;
}
int main() {
    double b = run<0, 0, 0, true>("bare", 0);
    run<8, 0, 0, true>("8 VALU before", b);
    run<16, 0, 0, true>("16 VALU before", b);
    run<32, 0, 0, true>("32 VALU before", b);
    run<8, 0, 0, false>("8 VALU interleaved", b);
    run<0, 10, 0, true>("10 ds_read_b32 before", b);
    run<0, 20, 0, true>("20 ds_read_b32 before", b);
    run<0, 9, 0, false>("9 ds_read_b32 interleaved", b);
    run<0, 0, 3, true>("3 ds_read_b128 before", b);
    run<0, 0, 6, true>("6 ds_read_b128 before", b);
    run<16, 10, 0, true>("16 VALU + 10 ds_read_b32", b);
    return 0;
}
