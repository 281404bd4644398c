// What ds_read_b64_tr_b16 returns: LDS holds element value = its own element index (u16); every lane passes a byte address and
// prints the four u16 it receives.  Build: hipcc --offload-arch=gfx950 -O3 -o tr_b16 tr_b16.hip ; run: ./tr_b16
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, int pitch_elems) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // 16-lane group g: rows (l & 15) / 4 of a matrix with row pitch `pitch_elems`, 8-byte chunk (l & 3); groups 256 elements apart
    const unsigned addr = (unsigned)(size_t)lds + 2u * (((l & 15) >> 2) * pitch_elems + (l & 3) * 4 + (l >> 4) * 1024);
    unsigned v0, v1;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*reinterpret_cast<unsigned long long*>(&v0)) : "v"(addr) : "memory");
    (void)v1;
    unsigned long long r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[2 * l] = (unsigned)r;
    out[2 * l + 1] = (unsigned)(r >> 32);
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 512);
    for (int pitch : {16, 40}) {
        k<<<1, 64>>>(d, pitch);
        unsigned h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("pitch %d elements\n", pitch);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u%s", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16, (l & 3) == 3 ? "\n" : " | ");
    }
    return 0;
}
