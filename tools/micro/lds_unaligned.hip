// Does gfx950 serve ds_read_b128 / ds_read_b64 at addresses that are only 2- or 4-byte aligned, and at what rate?
// (stem weight gradient on the bf16 matrix cores: eight consecutive pixels of a column-parity plane start at any 2-byte offset)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned tools/micro/lds_unaligned.hip && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ u32x2 lds_read64(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// mode 0: b128, mode 1: 2 x b64; misalign = byte offset added to a 16-byte-aligned per-lane address; stride = bytes between lanes
__global__ void probe(unsigned* out, unsigned long long* clk, int mode, int misalign, int stride, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * stride + misalign;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + (it & 15) * 2048;
        if (mode == 0) {
            const u32x4 v = lds_read128(a);
            acc += v;
        } else {
            const u32x2 lo = lds_read64(a), hi = lds_read64(a + 8);
            acc[0] += lo[0]; acc[1] += lo[1]; acc[2] += hi[0]; acc[3] += hi[1];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (iters == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = acc[j];
    } else if (threadIdx.x == 0) {
        clk[0] = t1 - t0;
        out[0] = acc[0] + acc[1] + acc[2] + acc[3];
    }
}

// throughput: 8 independent reads in flight per wave, `nw` waves per CU
__global__ void tput(unsigned* out, unsigned long long* clk, int misalign, int stride, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * stride + misalign + (threadIdx.x >> 6) * 4096;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "n"(j * 160) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { clk[0] = t1 - t0; out[0] = acc[0] + acc[1] + acc[2] + acc[3]; }
    else if (acc[0] == 0x12345678u) out[1] = acc[1];
}

int main() {
    {
        unsigned* o; unsigned long long* c;
        hipMalloc(&o, 64); hipMalloc(&c, 8);
        hipFuncSetAttribute((const void*)tput, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int nw : {1, 4, 8})
            for (int mis : {0, 2, 4, 8})
                for (int stride : {16, 80, 2080}) {
                    tput<<<256, 64 * nw, 65536>>>(o, c, mis, stride, 2048);
                    unsigned long long cc = 0;
                    hipMemcpy(&cc, c, 8, hipMemcpyDeviceToHost);
                    printf("throughput: %d waves/CU, misalign %d B, lane stride %4d B: %.1f clocks per ds_read_b128 per wave -> %.1f clocks of LDS per read\n", nw, mis, stride,
                           (double)cc / 2048 / 8, (double)cc / 2048 / 8 / nw);
                }
    }
    unsigned* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 4 * 4); hipMalloc(&clk, 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int mode = 0; mode < 2; ++mode)
        for (int mis : {0, 2, 4, 6, 8, 14})
            for (int stride : {16, 18, 144}) {
                probe<<<1, 64, 65536>>>(out, clk, mode, mis, stride, 1);
                std::vector<unsigned> h(64 * 4);
                if (hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("mode %d misalign %d: launch failed (%s)\n", mode, mis, hipGetErrorString(hipGetLastError())); return 1; }
                int bad = 0;
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const unsigned e0 = (l * stride + mis) / 2 + 2 * j, want = (e0 & 0xffff) | (((e0 + 1) & 0xffff) << 16);
                        if (h[l * 4 + j] != want) ++bad;
                    }
                probe<<<256, 256, 65536>>>(out, clk, mode, mis, stride, 4096);
                unsigned long long c = 0;
                hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
                printf("%s misalign %2d B, lane stride %3d B: %s, %.1f clocks per wave-read (4 waves per CU in flight, dependent chain)\n", mode ? "2 x ds_read_b64" : "ds_read_b128  ", mis, stride,
                       bad ? "WRONG DATA" : "data ok", (double)c / 4096);
            }
    return 0;
}
