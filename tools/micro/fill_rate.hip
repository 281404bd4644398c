// L2 / HBM -> LDS fill rate of one CU with global_load_lds_dwordx4 (1 KiB per wave-instruction), every CU of the chip streaming at once:
// what a convolution's staging can ingest.  W waves per workgroup issue copies back to back, keeping at most D in flight each
// (s_waitcnt vmcnt(D - 1) after every issue once D are out); the source is either ONE region shared by all workgroups (weights: L2 hits
// after the first touch) or a private stream per workgroup (activations: HBM).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/fill_rate tools/micro/fill_rate.hip && tools/micro/fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int D>
__global__ __launch_bounds__(512) void fill(const char* src, size_t region, size_t priv_stride, int iters, unsigned long long* clk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * priv_stride;
    const unsigned lds0 = (unsigned)(size_t)smem + wave * D * 1024;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    size_t off = (size_t)wave * 1024;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + off + lane * 16),
                                             reinterpret_cast<__attribute__((address_space(3))) void*>(lds0 + d * 1024), 16, 0, 0);
            off += (size_t)nw * 1024;
            if (off >= region) off -= region;
            if (i > 0 || d == D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (sink) sink[threadIdx.x] = reinterpret_cast<float*>(smem)[threadIdx.x];
}

template <int D>
static int run(const char* tag, const char* buf, size_t region, size_t stride, int waves, unsigned long long* dclk) {
    const int iters = 400, grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto k = fill<D>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (size_t)waves * D * 1024;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, 0, buf, region, stride, iters, dclk, (float*)nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), dclk, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= grid;
    const double bytes = (double)iters * D * waves * 1024;
    printf("%-34s waves %d depth %2d: %7.1f B/clk/CU  (%.0f clk per workgroup, %.2f TB/s chip, kernel %.3f ms)\n", tag, waves, D, bytes / avg, avg,
           bytes * grid / (ms * 1e-3) / 1e12, ms);
    return 0;
}

int main() {
    const size_t total = (size_t)1 << 30;
    char* buf;
    unsigned long long* dclk;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMemset(buf, 1, total));
    CHECK(hipMalloc(&dclk, 256 * sizeof(unsigned long long)));
    const size_t shared = 288 * 1024, priv = total / 256;
    for (int waves : {1, 2, 4, 8}) {
        if (run<4>("shared 288 KB (L2)", buf, shared, 0, waves, dclk)) return 1;
        if (run<8>("shared 288 KB (L2)", buf, shared, 0, waves, dclk)) return 1;
        if (run<16>("shared 288 KB (L2)", buf, shared, 0, waves, dclk)) return 1;
    }
    for (int waves : {1, 2, 4, 8}) {
        if (run<4>("private 4 MB stream (HBM / MALL)", buf, priv, priv, waves, dclk)) return 1;
        if (run<8>("private 4 MB stream (HBM / MALL)", buf, priv, priv, waves, dclk)) return 1;
        if (run<16>("private 4 MB stream (HBM / MALL)", buf, priv, priv, waves, dclk)) return 1;
    }
    return 0;
}
