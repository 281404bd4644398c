// Does buffer_load_dwordx4 ... lds write ZEROS into the LDS for lanes whose offset is out of the buffer's range (num_records)?
// (Needed for halo / padding pixels of a patch copied straight into the LDS without a branch or a pre-zeroed buffer.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/buf_lds_oob tools/micro/buf_lds_oob.hip && tools/micro/buf_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const char* p, int n, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    reinterpret_cast<uint4*>(smem)[threadIdx.x] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, n, 0x00020000);
    // even lanes read in range, odd lanes far out of range
    const unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, reinterpret_cast<__attribute__((address_space(3))) void*>((unsigned)(size_t)smem), 16, (int)off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = reinterpret_cast<unsigned*>(smem)[threadIdx.x * 4 + i];
}
int main() {
    char* buf; unsigned* out;
    hipMalloc(&buf, 4096); hipMalloc(&out, 64 * 16);
    unsigned h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    hipMemcpy(buf, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, buf, 4096, out);
    unsigned o[256];
    hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_oob = 0, untouched = 0;
    for (int l = 0; l < 64; ++l) {
        if (l & 1) { zero_oob += o[l * 4] == 0; untouched += o[l * 4] == 0xdeadbeefu; }
        else ok_in += o[l * 4] == 0x1000u + l * 4;
    }
    printf("in-range lanes correct %d/32; out-of-range lanes: zero written %d/32, LDS untouched %d/32 (first oob word 0x%x)\n", ok_in, zero_oob, untouched, o[4]);
    return 0;
}
