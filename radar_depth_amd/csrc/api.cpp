// Error reporting, device info, hipGraph helpers of libradardepth_hip.
#include <stdarg.h>

#include "common.h"

namespace rd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int num_cus() {
    static int n = 0;
    if (n == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
}  // namespace rd

extern "C" const char* rd_last_error(void) { return rd::g_err; }
extern "C" int rd_abi_version(void) { return 1; }

extern "C" int rd_device_info(int* n_cu, char* name, int name_len) {
    hipDeviceProp_t prop;
    int dev = 0;
    RD_CHECK_HIP(hipGetDevice(&dev));
    RD_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return RD_OK;
}

extern "C" int rd_graph_begin(void* stream) {
    RD_CHECK_HIP(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return RD_OK;
}
extern "C" int rd_graph_end(void* stream, void** graph_exec) {
    RD_CHECK_ARG(graph_exec != nullptr, "rd_graph_end: null out pointer");
    hipGraph_t g = nullptr;
    RD_CHECK_HIP(hipStreamEndCapture(static_cast<hipStream_t>(stream), &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        rd::set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return RD_ELAUNCH;
    }
    *graph_exec = ge;
    return RD_OK;
}
extern "C" int rd_graph_launch(void* graph_exec, void* stream) {
    RD_CHECK_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
    return RD_OK;
}
extern "C" int rd_graph_destroy(void* graph_exec) {
    if (graph_exec) RD_CHECK_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
    return RD_OK;
}

// Diagnostics: fill the LDS of every CU with NaN bit patterns.  LDS is not cleared between kernels, so a kernel that reads an LDS
// word it never wrote (and, say, multiplies it by a zero operand) then produces NaN instead of silently depending on whatever
// the previous kernel left there (tools/fuzz_conv.py --poison; this is how the 0*NaN hazard of the strip wgrad kernel is kept out).
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned* sink, int words) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = 0xFFFFFFFFu;
    rd::rd_sync();
    if (sink && lds_words[(threadIdx.x * 97) % words] == 0u) sink[0] = 1;      // keep the stores alive
}
extern "C" int rd_debug_poison_lds(void* stream) {
    static std::atomic<unsigned long long> attr_set{0};
    const int bytes = 160 * 1024 - 1024;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(rd::num_cus() * 4), dim3(256), bytes, static_cast<hipStream_t>(stream), (unsigned*)nullptr, bytes / 4);
    RD_CHECK_LAUNCH("poison_lds_kernel");
    return RD_OK;
}

// Events for fork/join between the plan's streams (captured as graph edges under hipStreamBeginCapture)
extern "C" int rd_event_create(void** ev) {
    RD_CHECK_ARG(ev != nullptr, "rd_event_create: null out pointer");
    hipEvent_t e = nullptr;
    RD_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = e;
    return RD_OK;
}
extern "C" int rd_event_destroy(void* ev) {
    if (ev) RD_CHECK_HIP(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return RD_OK;
}
extern "C" int rd_event_record(void* ev, void* stream) {
    RD_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream)));
    return RD_OK;
}
extern "C" int rd_stream_wait_event(void* stream, void* ev) {
    RD_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0));
    return RD_OK;
}
