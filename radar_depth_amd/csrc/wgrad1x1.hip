// Weight gradient of the 1x1 convolutions (conv_fusion 640 -> 512, conv2 512 -> 256, the stride-2 downsample convolutions of
// layer2..4 and layer4_depth; model/models.py:96-112 downsample, :652-657) as a plain GEMM whose reduction index is the pixel:
//   dW[ci][co] = sum over output pixels p of x[pixel(p)][ci] * dout[p][co]          (pixel(p) = p, or (2 oh, 2 ow) at stride 2)
// The generic wgrad_kernel<1,32,...> walks these layers with its tap machinery: one A read, one B read and one pixel-table read
// from LDS per MFMA plus the address arithmetic, 64 x 64 blocks -- 15-30 % of the fp32 peak (conv_fusion 100 us for 3.9 GFLOP).
// Here:
//   * a workgroup owns a (CIB x COB) block of dW, CIB / COB = 64 or 128, and a contiguous range of output pixels (split-K);
//     its 2 x 2 waves hold TI x TO accumulator tiles of 32 x 32 each (up to 64 x 64 per wave): every A fragment feeds TO MFMAs,
//     every B fragment TI -- half the LDS reads and half the staged bytes per MFMA of the 64 x 64 form;
//   * pixels are staged 32 at a time, straight global -> LDS (global_load_lds, no registers), double-buffered: the loads of
//     chunk c+1 are in flight while chunk c is in the matrix cores; one barrier per chunk;
//   * LDS tile layout [32-channel sub-tile][32 pixels][32 channels]: a wave's 64 staging lanes (8 pixels x 8 channel quads)
//     write 1 KB contiguously (what global_load_lds needs), and in the walk every fragment address is a lane constant plus an
//     IMMEDIATE (pixel pair s of the chunk = s * 256 bytes): no address arithmetic, no pixel table;
//   * a thread stages the same pixel lane in every chunk, so its (image, row, column) -- needed for the stride-2 gather -- is
//     carried along instead of divided out;
//   * slabs in wgrad.hip's layout [split][Cin][Cout], reduced by its deterministic two-stage reduction.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "common.h"

namespace rd {

struct Wgrad1x1Args {
    const float* x;
    const float* dout;
    float* slabs;
    int N, Hi, Wi, Cin, ldi, Ho, Wo, Cout, ldo, IS, dh, dw;
    int n_cib, n_cob, n_splits;
    long long P, pix_per_split;      // output pixels in all / per split (a multiple of 32)
};

constexpr int W1_PC = 32;            // pixels per staged chunk

template <int TI, int TO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad1x1_kernel(const Wgrad1x1Args a) {
    constexpr int CIB = 2 * TI * 32, COB = 2 * TO * 32;
    constexpr int XS = CIB * W1_PC, DS = COB * W1_PC;          // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const s_x = smem;                     // [2][CIB/32][32 px][32 ch]
    float* const s_d = smem + 2 * XS;            // [2][COB/32][32 px][32 ch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, hh = lane >> 5;
    const int wci = wave >> 1, wco = wave & 1;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int nblk = a.n_cib * a.n_cob;
    const int blk = vid % nblk, split = vid / nblk;
    const int cib0 = (blk / a.n_cob) * CIB, cob0 = (blk % a.n_cob) * COB;
    const long long p0 = (long long)split * a.pix_per_split;
    const long long p1 = p0 + a.pix_per_split < a.P ? p0 + a.pix_per_split : a.P;
    const int nchunks = p1 > p0 ? (int)((p1 - p0 + W1_PC - 1) / W1_PC) : 0;

    f32x16 acc[TI][TO];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int to = 0; to < TO; ++to)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ti][to][i] = 0.f;

    // staging role of this thread: pixel lane sp of every chunk, channel quad sq of every 32-channel sub-tile
    const int sp = tid >> 3, sq = tid & 7;
    long long pcur = p0 + sp;                    // output pixel this thread stages in the chunk being fetched
    int on, ooh, oow;                            // its (image, row, column)
    {
        const long long hw = (long long)a.Ho * a.Wo;
        on = (int)(pcur / hw);
        const int rem = (int)(pcur - (long long)on * hw);
        ooh = rem / a.Wo;
        oow = rem - ooh * a.Wo;
    }
    // channel quads beyond Cin / Cout are clamped onto the last valid quad: they feed accumulator rows / columns that are never stored
    const int xq = cib0 + sq * 4, dq = cob0 + sq * 4;
    long long pchunk = p0;                       // first pixel of the chunk being fetched
    auto issue = [&](int buf) {
        const float* xs = a.x + (((size_t)on * a.Hi + (size_t)(ooh * a.IS + a.dh)) * a.Wi + (size_t)(oow * a.IS + a.dw)) * a.ldi;
        const float* ds = a.dout + (size_t)pcur * a.ldo;
        float* xb = s_x + buf * XS + wave * 256;     // lane-linear: this wave's 64 lanes fill 1 KB of every sub-tile
        float* db = s_d + buf * DS + wave * 256;
        if (pchunk + W1_PC <= p1) {              // whole chunk inside the split (workgroup-uniform): nothing but copies in flight
#pragma unroll
            for (int j = 0; j < CIB / 32; ++j) {
                const int ch = xq + j * 32;
                glds16(xs + (ch < a.Cin ? ch : a.Cin - 4), xb + j * 1024);
            }
#pragma unroll
            for (int j = 0; j < COB / 32; ++j) {
                const int ch = dq + j * 32;
                glds16(ds + (ch < a.Cout ? ch : a.Cout - 4), db + j * 1024);
            }
        } else {                                 // last chunk of the last split: pixels past the end contribute zeros
            const bool live = pcur < p1;
#pragma unroll
            for (int j = 0; j < CIB / 32; ++j) {
                const int ch = xq + j * 32;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) v = *reinterpret_cast<const float4*>(xs + (ch < a.Cin ? ch : a.Cin - 4));
                *reinterpret_cast<float4*>(xb + j * 1024 + lane * 4) = v;
            }
#pragma unroll
            for (int j = 0; j < COB / 32; ++j) {
                const int ch = dq + j * 32;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) v = *reinterpret_cast<const float4*>(ds + (ch < a.Cout ? ch : a.Cout - 4));
                *reinterpret_cast<float4*>(db + j * 1024 + lane * 4) = v;
            }
        }
        pchunk += W1_PC;
        // advance this thread's pixel by one chunk
        pcur += W1_PC;
        oow += W1_PC;
        while (oow >= a.Wo) {
            oow -= a.Wo;
            if (++ooh == a.Ho) { ooh = 0; ++on; }
        }
    };

    if (nchunks > 0) issue(0);
    for (int c = 0; c < nchunks; ++c) {
        glds_wait();
        rd_sync();                               // chunk c has landed; every wave is done with chunk c-1's buffer
        if (c + 1 < nchunks) issue((c + 1) & 1);
        const float* xa = s_x + (c & 1) * XS + (wci * TI) * 1024 + hh * 32 + lm;
        const float* db = s_d + (c & 1) * DS + (wco * TO) * 1024 + hh * 32 + lm;
        // 16 pixel pairs; fragment (sub-tile t, pair s) = base + t * 1024 + s * 64 floats: immediates (the compiler fetches two
        // consecutive pairs with one ds_read2st64_b32 and issues the next reads behind the 2 * TI * TO MFMAs of the previous two
        // pairs, so all but the last MFMA's 64 cycles cover the LDS latency; an explicit register double buffer compiles to the same)
#pragma unroll
        for (int s = 0; s < W1_PC / 2; ++s) {
            float A[TI], B[TO];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) A[ti] = xa[ti * 1024 + s * 64];
#pragma unroll
            for (int to = 0; to < TO; ++to) B[to] = db[to * 1024 + s * 64];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int to = 0; to < TO; ++to) acc[ti][to] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[ti], B[to], acc[ti][to], 0, 0, 0);
        }
    }

    // slab [Cin][Cout] of this split (every element of the block is written: zeros where the split had no pixels)
    float* slab = a.slabs + (size_t)split * a.Cin * a.Cout;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int to = 0; to < TO; ++to) {
            const int co = cob0 + (wco * TO + to) * 32 + lm;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = cib0 + (wci * TI + ti) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (ci < a.Cin && co < a.Cout) slab[(size_t)ci * a.Cout + co] = acc[ti][to][i];
            }
        }
}

bool wgrad1x1_eligible(const RdConvDesc& d) {
    static const char* off = getenv("RD_WGRAD_NO1X1");      // diagnostics: keep these layers on the generic kernel
    if (off) return false;
    if (d.n_phases != 1 || d.out_stride != 1 || (d.in_stride != 1 && d.in_stride != 2)) return false;
    const RdPhase& p = d.phase[0];
    if (p.n_taps != 1 || p.widx[0] != 0 || p.out_off_h != 0 || p.out_off_w != 0 || p.lh != d.Ho || p.lw != d.Wo) return false;
    if (p.dh[0] < 0 || p.dw[0] < 0) return false;
    if ((d.Ho - 1) * d.in_stride + p.dh[0] >= d.Hi || (d.Wo - 1) * d.in_stride + p.dw[0] >= d.Wi) return false;      // no padding
    if (d.Cin < 64 || d.Cout < 64 || d.Cin % 4 != 0 || d.Cout % 4 != 0 || d.ldi % 4 != 0 || d.ldo % 4 != 0) return false;
    return true;
}

static void w1_blocks(const RdConvDesc& d, int& ti, int& to, int& n_cib, int& n_cob) {
    ti = d.Cin >= 128 ? 2 : 1;
    to = d.Cout >= 128 ? 2 : 1;
    n_cib = cdiv(d.Cin, 2 * ti * 32);
    n_cob = cdiv(d.Cout, 2 * to * 32);
}

// pixel splits: about one workgroup per CU in all (two measured 5-30 % slower: twice the slabs), every split a multiple of the
// 32-pixel chunk
void wgrad1x1_splits(const RdConvDesc& d, int& n_splits, long long& pix_per_split) {
    int ti, to, n_cib, n_cob;
    w1_blocks(d, ti, to, n_cib, n_cob);
    const long long P = (long long)d.N * d.Ho * d.Wo;
    static const char* wpc = getenv("RD_WGRAD1X1_WG_PER_CU");      // diagnostics
    const int want_wgs = (wpc ? atoi(wpc) : 1) * num_cus();
    long long ns = want_wgs / (n_cib * n_cob);
    if (ns < 1) ns = 1;
    const long long max_ns = cdiv64(P, 2 * W1_PC);                  // at least two chunks per split
    if (ns > max_ns) ns = max_ns < 1 ? 1 : max_ns;
    pix_per_split = cdiv64(cdiv64(P, ns), W1_PC) * W1_PC;
    n_splits = (int)cdiv64(P, pix_per_split);
}

template <int TI, int TO>
static int launch_w1(const Wgrad1x1Args& a, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (2 * TI * 32 + 2 * TO * 32) * W1_PC * sizeof(float);
    static std::atomic<unsigned long long> attr{0};
    auto k = wgrad1x1_kernel<TI, TO>;
    RD_SET_ATTR_ONCE(attr, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(a.n_cib * a.n_cob * a.n_splits), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("wgrad1x1_kernel");
    return RD_OK;
}

int launch_wgrad1x1(const RdConvDesc& d, const float* x, const float* dout, float* slabs, hipStream_t s) {
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(dout) % 16 == 0, "wgrad1x1: unaligned tensor");
    Wgrad1x1Args a;
    a.x = x; a.dout = dout; a.slabs = slabs;
    a.N = d.N; a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.ldi = d.ldi; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout; a.ldo = d.ldo;
    a.IS = d.in_stride; a.dh = d.phase[0].dh[0]; a.dw = d.phase[0].dw[0];
    int ti, to;
    w1_blocks(d, ti, to, a.n_cib, a.n_cob);
    a.P = (long long)d.N * d.Ho * d.Wo;
    wgrad1x1_splits(d, a.n_splits, a.pix_per_split);
    if (ti == 2 && to == 2) return launch_w1<2, 2>(a, s);
    if (ti == 1 && to == 2) return launch_w1<1, 2>(a, s);
    if (ti == 2 && to == 1) return launch_w1<2, 1>(a, s);
    return launch_w1<1, 1>(a, s);
}

}  // namespace rd
