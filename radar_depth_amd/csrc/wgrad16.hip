// Weight gradient of the 16 -> 16 channel 3x3 / stride-1 convolutions (the layers conv16.hip serves) on v_mfma_f32_16x16x4_f32:
//   dW[tap][ci][co] = sum over pixels p of x[p + tap][ci] * dout[p][co]
// i.e. nine 16 x 16 GEMMs whose reduction dimension is the pixel index: M = ci, N = co, K = 4 pixels per MFMA.  The generic
// wgrad_kernel<9,16,...> ran these layers at 22-25 % of the fp32 peak (42 us for 1.7 GFLOP on 46 MB of input).
//   * tile = 16 x 16 pixels; x halo patch [18][18][16] and dout tile [16][16][16] in LDS; a wave owns four rows = 16 groups of
//     four consecutive pixels; per group: one dout read, nine x reads (ds_read_b32: lane = channel, lane / 16 = pixel of the
//     group; the 16-float pixel pitch puts pixels k and k + 1 on disjoint halves of the 32 banks), nine MFMAs into nine
//     16 x 16 accumulators (36 registers);
//   * a workgroup walks tiles split, split + n_splits, ... (fixed assignment, fixed order: deterministic), then the four waves'
//     accumulators are summed through LDS and written as one slab [9][16][16] in wgrad.hip's slab layout, reduced by its
//     two-stage slab reduction (rd_wgrad_reduce) like every other weight gradient.
#include <hip/hip_runtime.h>

#include "common.h"

namespace rd {

struct Wgrad16Args {
    const float* x;
    const float* dout;
    float* slabs;
    int ldi, ldo;
    int N, H, W, tiles_h, tiles_w, total_tiles, n_splits;
    int ih_off, iw_off;
    int widx_pos[9];      // weight slab of the tap at patch position (pos / 3, pos % 3)
};

constexpr int W16_T = 16, W16_P = 18;

__global__ __launch_bounds__(256) void wgrad16_kernel(const Wgrad16Args a) {
    // x patch (5184 floats) + dout tile (4096 floats); the whole array is reused for the cross-wave reduction (4 x 2304 floats)
    __shared__ __attribute__((aligned(16))) float s_mem[W16_P * W16_P * 16 + W16_T * W16_T * 16];
    float* const s_x = s_mem;
    float* const s_d = s_mem + W16_P * W16_P * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int per_img = a.tiles_h * a.tiles_w;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += a.n_splits) {
        const int n = tile / per_img, trem = tile - n * per_img;
        const int r0 = (trem / a.tiles_w) * W16_T, c0 = (trem % a.tiles_w) * W16_T;
        const int ih0 = r0 + a.ih_off, iw0 = c0 + a.iw_off;
        const float* x_n = a.x + (size_t)n * a.H * a.W * a.ldi;
        const float* d_n = a.dout + (size_t)n * a.H * a.W * a.ldo;
        rd_sync();          // the previous tile's reads are done
        {
            float4 vx[6], vd[4];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int e = tid + u * 256;
                const int px = e >> 2, q = e & 3;
                const int py = px / W16_P, pxx = px - py * W16_P;
                const int ih = ih0 + py, iw = iw0 + pxx;
                vx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < W16_P * W16_P * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                    vx[u] = *reinterpret_cast<const float4*>(x_n + ((size_t)ih * a.W + iw) * a.ldi + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + u * 256;
                const int px = e >> 2, q = e & 3;
                const int r = r0 + (px >> 4), c = c0 + (px & 15);
                vd[u] = make_float4(0.f, 0.f, 0.f, 0.f);          // pixels outside the image contribute nothing
                if (r < a.H && c < a.W) vd[u] = *reinterpret_cast<const float4*>(d_n + ((size_t)r * a.W + c) * a.ldo + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int e = tid + u * 256;
                if (e < W16_P * W16_P * 4) *reinterpret_cast<float4*>(s_x + e * 4) = vx[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(s_d + (tid + u * 256) * 4) = vd[u];
        }
        rd_sync();
        // accumulators are indexed by PATCH POSITION (the slab index of the tap sitting there is looked up when the slab is
        // written): every LDS address below is one lane-constant base plus a compile-time offset, i.e. an immediate of the
        // ds_read -- with runtime tap offsets the compiler kept 160 address registers and the kernel ran one wave per SIMD
        const float* xl = s_x + ((wave * 4) * W16_P + kq) * 16 + m;
        const float* dl = s_d + ((wave * 4) * W16_T + kq) * 16 + m;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float b = dl[(mb * W16_T + 4 * g) * 16];
#pragma unroll
                for (int pos = 0; pos < 9; ++pos)
                    acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(xl[((mb + pos / 3) * W16_P + 4 * g + pos % 3) * 16], b, acc[pos], 0, 0, 0);
            }
        }
    }

    // sum the four waves' accumulators (C/D layout: lane = co, registers = ci 4 kq .. 4 kq + 3) and write the slab [tap][ci][co]
    rd_sync();
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) s_mem[wave * 9 * 256 + (t * 16 + 4 * kq + v) * 16 + m] = acc[t][v];
    rd_sync();
    float* slab = a.slabs + (size_t)blockIdx.x * 9 * 256;
    for (int e = tid; e < 9 * 256; e += 256) {
        const int t = e >> 8, rest = e & 255;
        const float s = (s_mem[e] + s_mem[9 * 256 + e]) + (s_mem[2 * 9 * 256 + e] + s_mem[3 * 9 * 256 + e]);
        slab[a.widx_pos[t] * 256 + rest] = s;
    }
}

bool wgrad16_eligible(const RdConvDesc& d) {
    static const char* off = getenv("RD_WGRAD_NOW16");   // diagnostics: keep these layers on the generic kernel
    if (off) return false;
    if (d.n_phases != 1 || d.Cin != 16 || d.Cout != 16 || d.in_stride != 1 || d.out_stride != 1) return false;
    const RdPhase& p = d.phase[0];
    if (p.n_taps != 9 || p.dh_max - p.dh_min != 2 || p.dw_max - p.dw_min != 2) return false;
    if (p.out_off_h != 0 || p.out_off_w != 0 || p.lh != d.Ho || p.lw != d.Wo || d.Hi != d.Ho || d.Wi != d.Wo) return false;
    int seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (p.widx[t] < 0 || p.widx[t] >= 9) return false;
        seen |= 1 << ((p.dh[t] - p.dh_min) * 3 + (p.dw[t] - p.dw_min));
    }
    return seen == 0x1ff && d.ldi % 4 == 0 && d.ldo % 4 == 0;      // every position of the 3x3 stencil exactly once
}

// pixel tiles, and how many workgroups (= slabs) share them: about four per CU, every one with the same number of tiles
void wgrad16_splits(const RdConvDesc& d, int& total_tiles, int& n_splits) {
    total_tiles = d.N * cdiv(d.Ho, W16_T) * cdiv(d.Wo, W16_T);
    const int per = cdiv(total_tiles, 4 * num_cus());
    n_splits = cdiv(total_tiles, per);
}

int launch_wgrad16(const RdConvDesc& d, const float* x, const float* dout, float* slabs, hipStream_t s) {
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(dout) % 16 == 0, "wgrad16: unaligned tensor");
    Wgrad16Args a;
    a.x = x; a.dout = dout; a.slabs = slabs; a.ldi = d.ldi; a.ldo = d.ldo;
    a.N = d.N; a.H = d.Ho; a.W = d.Wo; a.tiles_h = cdiv(d.Ho, W16_T); a.tiles_w = cdiv(d.Wo, W16_T);
    wgrad16_splits(d, a.total_tiles, a.n_splits);
    const RdPhase& p = d.phase[0];
    a.ih_off = p.dh_min; a.iw_off = p.dw_min;
    for (int t = 0; t < 9; ++t) a.widx_pos[(p.dh[t] - p.dh_min) * 3 + (p.dw[t] - p.dw_min)] = p.widx[t];
    hipLaunchKernelGGL(wgrad16_kernel, dim3(a.n_splits), dim3(256), 0, s, a);
    RD_CHECK_LAUNCH("wgrad16_kernel");
    return RD_OK;
}

}  // namespace rd
