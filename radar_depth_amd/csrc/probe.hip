// Shader-clock probe (round 6): what the matrix kernels' "power-limited" explanation rests on is the clock the chip
// actually runs at INSIDE the training step.  One wave per workgroup reads the shader cycle counter (s_memtime) and the constant
// 100 MHz real-time counter (s_memrealtime) at both ends of a window it spends asleep (s_sleep: no issue slots taken from the step's
// kernels running on the same CU); cycles / real time over the window is the effective clock of that workgroup's XCD while the
// step's kernels execute beside it.  Launched by bench.py on a stream of its own, concurrently with the steps of an extra pass
// OUTSIDE the timed region; n_blocks workgroups land round-robin on the XCDs.
#include "common.h"

namespace rd {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1;
    do {
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < ticks);
    const unsigned long long c1 = __builtin_readcyclecounter();
    unsigned long long* o = out + (size_t)blockIdx.x * 4;
    o[0] = c1 - c0;                                                       // shader clocks
    o[1] = r1 - r0;                                                       // 10 ns ticks
    o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    o[3] = r0;
}
}  // namespace rd

extern "C" int rd_clock_probe(unsigned long long* out, int32_t n_blocks, int32_t duration_us, void* stream) {
    RD_CHECK_ARG(out && n_blocks > 0 && n_blocks <= 64 && duration_us > 0 && duration_us <= 100000, "rd_clock_probe: bad arguments");
    hipLaunchKernelGGL(rd::clock_probe_kernel, dim3(n_blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out,
                       (unsigned long long)duration_us * 100ull);
    RD_CHECK_LAUNCH("clock_probe_kernel");
    return RD_OK;
}
