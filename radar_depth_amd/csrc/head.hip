// Network head: conv3 (3x3, C -> 1, pad 1, no bias; model/models.py:587,661) and the bilinear
// align_corners=True resize to output_size (models.py:588,662), forward and backward.
// All HBM-streaming kernels (one channel out); no matrix cores.
#include <stdlib.h>

#include "common.h"

namespace rd {

int launch_slab_reduce(const float* slabs, int n_splits, int64_t E, float* tmp, float* grad_oihw, int S, int Cin, int Cout,
                       int O, int I, int co_off, int accumulate, hipStream_t s);

// pixel index -> (image, row, column); 32-bit divisions whenever the tensor has fewer than 2^31 pixels (a 64-bit division is
// ~100 instructions, three of them per pixel were a large part of these streaming kernels)
__device__ __forceinline__ void split_pixel(int64_t e, bool small, int H, int W, int& n, int& h, int& wx) {
    if (small) {
        const unsigned u = (unsigned)e, r = u / (unsigned)W;
        wx = (int)(u - r * (unsigned)W);
        n = (int)(r / (unsigned)H);
        h = (int)(r - (unsigned)n * (unsigned)H);
    } else {
        wx = (int)(e % W);
        const int64_t r = e / W;
        h = (int)(r % H);
        n = (int)(r / H);
    }
}

// d[n,h,w] = sum_{kh,kw,c} x[n,h+kh-1,w+kw-1,c] * w[c][kh][kw]      (w: OIHW with O == 1)
// thread = (output pixel, channel quad): the C/4 lanes of a pixel read the 16-byte quads of each neighbour pixel side by side
// (one coalesced 4C-byte access per tap), multiply by their quad of the tap's weights and are summed with two DPP exchanges.
// All nine loads of a thread are issued unconditionally from clamped coordinates (a `continue` per tap made the compiler wait for
// each load before issuing the next; out-of-image taps get a zero weight instead).  Round 2's form -- one thread per pixel walking
// 9 x C/4 sixteen-byte loads behind per-tap branches -- streamed the 98 MB decoder output at 1.5 TB/s (65 us).
template <int C, typename T>
__global__ __launch_bounds__(256) void head_conv_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            int N, int H, int W, float* __restrict__ d) {
    static_assert(C == 16, "four lanes per pixel");
    constexpr int Q = C / 4;
    __shared__ float4 s_w[9 * Q];
    for (int e = threadIdx.x; e < 9 * Q; e += blockDim.x) {
        const int t = e / Q, c = (e - t * Q) * 4;
        s_w[e] = make_float4(w[c * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
    }
    rd_sync();
    const int64_t total = (int64_t)N * H * W;
    const bool small = total < (1ll << 31);
    const int q = threadIdx.x & (Q - 1);
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / Q; p < total; p += (int64_t)gridDim.x * blockDim.x / Q) {
        int wx, h, n;
        split_pixel(p, small, H, W, n, h, wx);
        const T* img = x + (size_t)n * H * W * ldx + q * 4;
        float4 v[9];
        float ok[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = h + kh - 1, ihc = min(max(ih, 0), H - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = wx + kw - 1, iwc = min(max(iw, 0), W - 1);
                ok[kh * 3 + kw] = (ih == ihc && iw == iwc) ? 1.f : 0.f;
                v[kh * 3 + kw] = ld4(img + ((size_t)ihc * W + iwc) * ldx);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wt = s_w[t * Q + q];
            float a = v[t].x * wt.x;
            a = fmaf(v[t].y, wt.y, a); a = fmaf(v[t].z, wt.z, a); a = fmaf(v[t].w, wt.w, a);
            s = fmaf(ok[t], a, s);
        }
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        if (q == 0) d[p] = s;
    }
}

// Tiled form (round 4): a workgroup stages the 10 x 34 pixel halo patch of an 8 x 32 output tile in LDS once (16-byte units, every byte of
// x read from HBM / L2 once instead of nine times through the L1) and takes the nine taps from there; thread = (pixel, channel quad) as above.
// The forward of the head sits alone on the step's dependent chain (49 us for the 98 MB decoder output with the kernel above).
constexpr int HT_H = 8, HT_W = 32, HT_PW = HT_W + 2, HT_PH = HT_H + 2;
template <int C, typename T>
__global__ __launch_bounds__(256) void head_conv_fwd_tiled_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                                  int N, int H, int W, int tiles_h, int tiles_w, float* __restrict__ d) {
    static_assert(C == 16, "four lanes per pixel");
    constexpr int Q = C / 4;
    __shared__ float4 s_w[9 * Q];
    __shared__ float4 s_x[HT_PH * HT_PW * Q];
    const int tid = threadIdx.x;
    for (int e = tid; e < 9 * Q; e += 256) {
        const int t = e / Q, c = (e - t * Q) * 4;
        s_w[e] = make_float4(w[c * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
    }
    const int per_img = tiles_h * tiles_w;
    const int n = blockIdx.x / per_img, trem = blockIdx.x - n * per_img;
    const int r0 = (trem / tiles_w) * HT_H, c0 = (trem % tiles_w) * HT_W;
    const T* img = x + (size_t)n * H * W * ldx;
    constexpr int UNITS = HT_PH * HT_PW * Q, U = (UNITS + 255) / 256;       // 1360 units, 6 per thread
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = tid + u * 256;
        const int px = e / Q, q = e - px * Q;
        const int py = px / HT_PW, pxx = px - py * HT_PW;
        const int ih = r0 - 1 + py, iw = c0 - 1 + pxx;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < UNITS && ih >= 0 && ih < H && iw >= 0 && iw < W) v[u] = ld4(img + ((size_t)ih * W + iw) * ldx + q * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (tid + u * 256 < UNITS) s_x[tid + u * 256] = v[u];
    rd_sync();
#pragma unroll
    for (int k = 0; k < HT_H * HT_W * Q / 256; ++k) {
        const int i = tid + k * 256;
        const int p = i / Q, q = i - p * Q;
        const int row = p / HT_W, col = p - row * HT_W;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 xv = s_x[((row + t / 3) * HT_PW + col + t % 3) * Q + q];
            const float4 wt = s_w[t * Q + q];
            s = fmaf(xv.x, wt.x, s); s = fmaf(xv.y, wt.y, s); s = fmaf(xv.z, wt.z, s); s = fmaf(xv.w, wt.w, s);
        }
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        const int oh = r0 + row, ow = c0 + col;
        if (q == 0 && oh < H && ow < W) d[((size_t)n * H + oh) * W + ow] = s;
    }
}

// dx[n,h,w,c] = sum_{kh,kw} dd[n,h-kh+1,w-kw+1] * w[c][kh][kw]
template <int C, typename T>
__global__ __launch_bounds__(256) void head_conv_dgrad_kernel(const float* __restrict__ dd, const float* __restrict__ w, int N,
                                                              int H, int W, T* __restrict__ dx, int lddx) {
    __shared__ float s_w[9 * C];
    for (int e = threadIdx.x; e < 9 * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C;
        s_w[e] = w[c * 9 + t];
    }
    rd_sync();
    constexpr int Q = C / 4;
    const int64_t total = (int64_t)N * H * W * Q;
    const bool small = total < (1ll << 31);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % Q) * 4;          // (Q is a power of two)
        int wx, h, n;
        split_pixel(e / Q, small, H, W, n, h, wx);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int oh = h - kh + 1;
            if (oh < 0 || oh >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ow = wx - kw + 1;
                if (ow < 0 || ow >= W) continue;
                const float g = dd[((size_t)n * H + oh) * W + ow];
                const float* wt = s_w + (kh * 3 + kw) * C + c;
                s.x = fmaf(g, wt[0], s.x); s.y = fmaf(g, wt[1], s.y); s.z = fmaf(g, wt[2], s.z); s.w = fmaf(g, wt[3], s.w);
            }
        }
        st4(dx + (((size_t)n * H + h) * W + wx) * lddx + c, s);
    }
}

// Tiled form (round 4): the 10 x 34 halo of the one-channel gradient map of an 8 x 32 pixel tile in LDS (zero outside the image), thread =
// (pixel, channel quad), no division or branch per element; the store of a wave is 1 KB contiguous.
template <int C, typename T>
__global__ __launch_bounds__(256) void head_conv_dgrad_tiled_kernel(const float* __restrict__ dd, const float* __restrict__ w, int N, int H, int W,
                                                                    int tiles_h, int tiles_w, T* __restrict__ dx, int lddx) {
    static_assert(C == 16, "four lanes per pixel");
    constexpr int Q = C / 4;
    __shared__ float4 s_w[9 * Q];         // [tap][quad]: w[c..c+3][kh][kw]
    __shared__ float s_d[HT_PH * HT_PW];
    const int tid = threadIdx.x;
    for (int e = tid; e < 9 * Q; e += 256) {
        const int t = e / Q, c = (e - t * Q) * 4;
        s_w[e] = make_float4(w[c * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
    }
    const int per_img = tiles_h * tiles_w;
    const int n = blockIdx.x / per_img, trem = blockIdx.x - n * per_img;
    const int r0 = (trem / tiles_w) * HT_H, c0 = (trem % tiles_w) * HT_W;
    const float* src = dd + (size_t)n * H * W;
    for (int e = tid; e < HT_PH * HT_PW; e += 256) {
        const int py = e / HT_PW, pxx = e - py * HT_PW;
        const int oh = r0 - 1 + py, ow = c0 - 1 + pxx;
        s_d[e] = (oh >= 0 && oh < H && ow >= 0 && ow < W) ? src[(size_t)oh * W + ow] : 0.f;
    }
    rd_sync();
#pragma unroll
    for (int k = 0; k < HT_H * HT_W * Q / 256; ++k) {
        const int i = tid + k * 256;
        const int p = i / Q, q = i - p * Q;
        const int row = p / HT_W, col = p - row * HT_W;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        // dx[h][w] = sum dd[h - kh + 1][w - kw + 1] * w[kh][kw]: patch position (row + 2 - kh, col + 2 - kw)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float g = s_d[(row + 2 - kh) * HT_PW + col + 2 - kw];
                const float4 wt = s_w[(kh * 3 + kw) * Q + q];
                s4.x = fmaf(g, wt.x, s4.x); s4.y = fmaf(g, wt.y, s4.y); s4.z = fmaf(g, wt.z, s4.z); s4.w = fmaf(g, wt.w, s4.w);
            }
        const int h = r0 + row, wx = c0 + col;
        if (h < H && wx < W) st4(dx + (((size_t)n * H + h) * W + wx) * lddx + q * 4, s4);
    }
}

// partial[block][t][c] = sum over the block's INPUT pixels q of x[q][c] * dd[q - tap t]   (= sum over output pixels p of
// x[p + tap t][c] * dd[p]: the same terms, grouped by the pixel of x)
// thread = (pixel lane, channel quad): every x element is read ONCE (16 bytes per thread and pixel) and multiplied by the nine
// neighbouring dd values (a one-channel map: cache hits); the nine partial sums per channel live in registers and the pixel
// lanes are reduced through LDS in a fixed order.  (The first version walked (tap, channel quad, pixel lane) and read x nine
// times with three 64-bit divisions per pixel: 137 us for a 98 MB tensor.)
template <int C, typename T>
__global__ __launch_bounds__(256) void head_conv_wgrad_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ dd,
                                                              int N, int H, int W, int64_t pix_per_block,
                                                              float* __restrict__ partial) {
    constexpr int Q = C / 4, PL = 256 / Q;            // 4 channel quads x 64 pixel lanes
    __shared__ float4 s_red[PL * Q];
    const int q = threadIdx.x % Q, pl = threadIdx.x / Q, c = q * 4;
    const int64_t total = (int64_t)N * H * W;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < total ? p0 + pix_per_block : total;
    float4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        int wx = (int)((p0 + pl) % W), h = (int)(((p0 + pl) / W) % H);      // column / row of the running pixel, carried along
        for (int64_t p = p0 + pl; p < p1; p += PL, wx += PL) {
            while (wx >= W) {
                wx -= W;
                h = h + 1 == H ? 0 : h + 1;
            }
            const float4 v = ld4(x + p * ldx + c);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int oh = h - (kh - 1);                    // output row whose tap (kh, kw) reads this input pixel
                if (oh < 0 || oh >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ow = wx - (kw - 1);
                    if (ow < 0 || ow >= W) continue;
                    const float g = dd[p - (int64_t)(kh - 1) * W - (kw - 1)];
                    float4& a = acc[kh * 3 + kw];
                    a.x = fmaf(g, v.x, a.x); a.y = fmaf(g, v.y, a.y); a.z = fmaf(g, v.z, a.z); a.w = fmaf(g, v.w, a.w);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        rd_sync();
        s_red[pl * Q + q] = acc[t];
        rd_sync();
        if (threadIdx.x < Q) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < PL; ++k) {
                const float4 v = s_red[k * Q + threadIdx.x];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 9 * C + t * C + threadIdx.x * 4) = s;
        }
    }
}

// align_corners=True bilinear; scale and source index computed in fp32 exactly as ATen does for float tensors
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ d, int N, int Hs, int Ws,
                                                           float* __restrict__ out, int Ho, int Wo, float sh, float sw) {
    const int64_t total = (int64_t)N * Ho * Wo;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int ox, oy, n;
        split_pixel(e, total < (1ll << 31), Ho, Wo, n, oy, ox);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oy, sh, Hs, y0, y1, ly);
        src_index(ox, sw, Ws, x0, x1, lx);
        const float* src = d + (size_t)n * Hs * Ws;
        const float hy = 1.f - ly, hx = 1.f - lx;
        out[e] = hy * (hx * src[y0 * Ws + x0] + lx * src[y0 * Ws + x1]) + ly * (hx * src[y1 * Ws + x0] + lx * src[y1 * Ws + x1]);
    }
}

// gather form of the backward: each source pixel sums the output pixels that read it (deterministic, no atomics)
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dout, int N, int Ho, int Wo,
                                                           float* __restrict__ dd, int Hs, int Ws, float sh, float sw) {
    const int64_t total = (int64_t)N * Hs * Ws;
    const float inv_h = sh > 0.f ? 1.f / sh : 0.f, inv_w = sw > 0.f ? 1.f / sw : 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int x, y, n;
        split_pixel(e, total < (1ll << 31), Hs, Ws, n, y, x);
        int oy_lo = (int)floorf((float)(y - 1) * inv_h) - 1, oy_hi = (int)ceilf((float)(y + 1) * inv_h) + 1;
        int ox_lo = (int)floorf((float)(x - 1) * inv_w) - 1, ox_hi = (int)ceilf((float)(x + 1) * inv_w) + 1;
        if (sh == 0.f) { oy_lo = 0; oy_hi = Ho - 1; }
        if (sw == 0.f) { ox_lo = 0; ox_hi = Wo - 1; }
        oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
        ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
        const float* src = dout + (size_t)n * Ho * Wo;
        float s = 0.f;
        if (oy_hi - oy_lo < 8 && ox_hi - ox_lo < 8) {
            // the weights are separable: one row of column weights per source pixel instead of one per candidate output pixel (the
            // 240 x 400 -> 450 x 800 resize has 6 x 6 candidates: 36 index computations became 12; 36 -> ~20 us)
            float wxs[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ox = ox_lo + i;
                int x0, x1;
                float lx;
                src_index(ox < Wo ? ox : Wo - 1, sw, Ws, x0, x1, lx);
                float wx = 0.f;
                if (x0 == x) wx += 1.f - lx;
                if (x1 == x) wx += lx;
                wxs[i] = ox <= ox_hi ? wx : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int oy = oy_lo + j;
                if (oy > oy_hi) break;
                int y0, y1;
                float ly;
                src_index(oy, sh, Hs, y0, y1, ly);
                float wy = 0.f;
                if (y0 == y) wy += 1.f - ly;
                if (y1 == y) wy += ly;
                if (wy == 0.f) continue;
                const float* row = src + (size_t)oy * Wo + ox_lo;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (wxs[i] != 0.f) s = fmaf(wy * wxs[i], row[i], s);
            }
            dd[e] = s;
            continue;
        }
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float ly;
            src_index(oy, sh, Hs, y0, y1, ly);
            float wy = 0.f;
            if (y0 == y) wy += 1.f - ly;
            if (y1 == y) wy += ly;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float lx;
                src_index(ox, sw, Ws, x0, x1, lx);
                float wx = 0.f;
                if (x0 == x) wx += 1.f - lx;
                if (x1 == x) wx += lx;
                if (wx != 0.f) s = fmaf(wy * wx, src[(size_t)oy * Wo + ox], s);
            }
        }
        dd[e] = s;
    }
}

static int ew_grid64(int64_t elems) {
    int64_t g = cdiv64(elems, 256);
    const int64_t cap = (int64_t)num_cus() * 16;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}
static int head_wgrad_blocks(int64_t pixels) {
    int64_t b = cdiv64(pixels, 512);
    // (four blocks per CU: RD_HEAD_WGRAD_BLOCKS_PER_CU = 2 / 4 / 8 / 16 measured 134 / 120 / 133 / 146 us for the whole head backward, round 3)
    static const char* bpc = getenv("RD_HEAD_WGRAD_BLOCKS_PER_CU");      // diagnostics
    const int64_t cap = (int64_t)num_cus() * (bpc ? atoi(bpc) : 4);
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace rd
using namespace rd;

template <typename T>
static int head_conv_fwd_T(const T* x, int32_t ldx, const float* w_oihw, int32_t N, int32_t H, int32_t W, int32_t C, float* d, void* stream) {
    RD_CHECK_ARG(x && w_oihw && d && C == 16 && ldx % 4 == 0, "head_conv_fwd: bad arguments (C must be 16)");
    static const bool tiled = !(getenv("RD_HEAD_FWD_TILED") && atoi(getenv("RD_HEAD_FWD_TILED")) == 0);
    const int tiles_h = cdiv(H, HT_H), tiles_w = cdiv(W, HT_W);
    if (tiled && (int64_t)N * tiles_h * tiles_w < (1ll << 31)) {
        hipLaunchKernelGGL((head_conv_fwd_tiled_kernel<16, T>), dim3(N * tiles_h * tiles_w), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, ldx, w_oihw, N, H, W, tiles_h, tiles_w, d);
        RD_CHECK_LAUNCH("head_conv_fwd_tiled_kernel");
        return RD_OK;
    }
    hipLaunchKernelGGL((head_conv_fwd_kernel<16, T>), dim3(ew_grid64((int64_t)N * H * W * 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, w_oihw, N, H, W, d);
    RD_CHECK_LAUNCH("head_conv_fwd_kernel");
    return RD_OK;
}
extern "C" int rd_head_conv_fwd(const float* x, int32_t ldx, const float* w_oihw, int32_t N, int32_t H, int32_t W, int32_t C,
                                float* d, void* stream) {
    return head_conv_fwd_T<float>(x, ldx, w_oihw, N, H, W, C, d, stream);
}
// storage-typed form: x is an fp32 (RD_DTYPE_F32) or bf16 (RD_DTYPE_BF16) NHWC tensor; the depth map d stays fp32
extern "C" int rd_head_conv_fwd_t(int32_t dtype, const void* x, int32_t ldx, const float* w_oihw, int32_t N, int32_t H, int32_t W,
                                  int32_t C, float* d, void* stream) {
    if (dtype == RD_DTYPE_F32) return head_conv_fwd_T<float>(static_cast<const float*>(x), ldx, w_oihw, N, H, W, C, d, stream);
    if (dtype == RD_DTYPE_BF16) return head_conv_fwd_T<bf16s>(static_cast<const bf16s*>(x), ldx, w_oihw, N, H, W, C, d, stream);
    rd::set_error("head_conv_fwd_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

extern "C" int64_t rd_head_conv_bwd_workspace_floats(int32_t N, int32_t H, int32_t W, int32_t C) {
    const int blocks = head_wgrad_blocks((int64_t)N * H * W);
    return (int64_t)(blocks + 16) * 9 * C;
}

// the two halves of the backward: the input gradient is what the rest of the backward waits for, the weight gradient (+ its slab
// reduction) gates nothing until the gradient bucket closes -- callers may put it on another stream (engine.py does)
template <typename T>
static int head_conv_dgrad_T(const float* w_oihw, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C, T* dx, int32_t lddx, void* stream) {
    RD_CHECK_ARG(w_oihw && dd && dx && C == 16 && lddx % 4 == 0, "head_conv_dgrad: bad arguments (C must be 16)");
    static const bool tiled = !(getenv("RD_HEAD_DGRAD_TILED") && atoi(getenv("RD_HEAD_DGRAD_TILED")) == 0);
    const int tiles_h = cdiv(H, HT_H), tiles_w = cdiv(W, HT_W);
    if (tiled && (int64_t)N * tiles_h * tiles_w < (1ll << 31)) {
        hipLaunchKernelGGL((head_conv_dgrad_tiled_kernel<16, T>), dim3(N * tiles_h * tiles_w), dim3(256), 0, static_cast<hipStream_t>(stream),
                           dd, w_oihw, N, H, W, tiles_h, tiles_w, dx, lddx);
        RD_CHECK_LAUNCH("head_conv_dgrad_tiled_kernel");
        return RD_OK;
    }
    hipLaunchKernelGGL((head_conv_dgrad_kernel<16, T>), dim3(ew_grid64((int64_t)N * H * W * 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dd, w_oihw, N, H, W, dx, lddx);
    RD_CHECK_LAUNCH("head_conv_dgrad_kernel");
    return RD_OK;
}
template <typename T>
static int head_conv_wgrad_T(const T* x, int32_t ldx, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C, float* dw_oihw, float* ws,
                             void* stream) {
    RD_CHECK_ARG(x && dd && dw_oihw && ws && C == 16 && ldx % 4 == 0, "head_conv_wgrad: bad arguments (C must be 16)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t pixels = (int64_t)N * H * W;
    const int blocks = head_wgrad_blocks(pixels);
    hipLaunchKernelGGL((head_conv_wgrad_kernel<16, T>), dim3(blocks), dim3(256), 0, s, x, ldx, dd, N, H, W, cdiv64(pixels, blocks), ws);
    RD_CHECK_LAUNCH("head_conv_wgrad_kernel");
    const int64_t E = 9 * C;
    return launch_slab_reduce(ws, blocks, E, ws + (int64_t)blocks * E, dw_oihw, 9, C, 1, 1, C, 0, 0, s);
}
template <typename T>
static int head_conv_bwd_T(const T* x, int32_t ldx, const float* w_oihw, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C,
                           T* dx, int32_t lddx, float* dw_oihw, float* ws, void* stream) {
    const int rc = head_conv_dgrad_T<T>(w_oihw, dd, N, H, W, C, dx, lddx, stream);
    return rc != RD_OK ? rc : head_conv_wgrad_T<T>(x, ldx, dd, N, H, W, C, dw_oihw, ws, stream);
}
extern "C" int rd_head_conv_dgrad_t(int32_t dtype, const float* w_oihw, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C, void* dx,
                                    int32_t lddx, void* stream) {
    if (dtype == RD_DTYPE_F32) return head_conv_dgrad_T<float>(w_oihw, dd, N, H, W, C, static_cast<float*>(dx), lddx, stream);
    if (dtype == RD_DTYPE_BF16) return head_conv_dgrad_T<bf16s>(w_oihw, dd, N, H, W, C, static_cast<bf16s*>(dx), lddx, stream);
    rd::set_error("head_conv_dgrad_t: bad dtype %d", dtype);
    return RD_EINVAL;
}
extern "C" int rd_head_conv_wgrad_t(int32_t dtype, const void* x, int32_t ldx, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C,
                                    float* dw_oihw, float* ws, void* stream) {
    if (dtype == RD_DTYPE_F32) return head_conv_wgrad_T<float>(static_cast<const float*>(x), ldx, dd, N, H, W, C, dw_oihw, ws, stream);
    if (dtype == RD_DTYPE_BF16) return head_conv_wgrad_T<bf16s>(static_cast<const bf16s*>(x), ldx, dd, N, H, W, C, dw_oihw, ws, stream);
    rd::set_error("head_conv_wgrad_t: bad dtype %d", dtype);
    return RD_EINVAL;
}
extern "C" int rd_head_conv_bwd(const float* x, int32_t ldx, const float* w_oihw, const float* dd, int32_t N, int32_t H, int32_t W,
                                int32_t C, float* dx, int32_t lddx, float* dw_oihw, float* ws, void* stream) {
    return head_conv_bwd_T<float>(x, ldx, w_oihw, dd, N, H, W, C, dx, lddx, dw_oihw, ws, stream);
}
extern "C" int rd_head_conv_bwd_t(int32_t dtype, const void* x, int32_t ldx, const float* w_oihw, const float* dd, int32_t N, int32_t H,
                                  int32_t W, int32_t C, void* dx, int32_t lddx, float* dw_oihw, float* ws, void* stream) {
    if (dtype == RD_DTYPE_F32)
        return head_conv_bwd_T<float>(static_cast<const float*>(x), ldx, w_oihw, dd, N, H, W, C, static_cast<float*>(dx), lddx, dw_oihw, ws, stream);
    if (dtype == RD_DTYPE_BF16)
        return head_conv_bwd_T<bf16s>(static_cast<const bf16s*>(x), ldx, w_oihw, dd, N, H, W, C, static_cast<bf16s*>(dx), lddx, dw_oihw, ws, stream);
    rd::set_error("head_conv_bwd_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

extern "C" int rd_bilinear_fwd(const float* d, int32_t N, int32_t Hs, int32_t Ws, float* out, int32_t Ho, int32_t Wo, void* stream) {
    RD_CHECK_ARG(d && out && N > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "bilinear_fwd: bad arguments");
    const float sh = Ho > 1 ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(ew_grid64((int64_t)N * Ho * Wo)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d, N, Hs, Ws, out, Ho, Wo, sh, sw);
    RD_CHECK_LAUNCH("bilinear_fwd_kernel");
    return RD_OK;
}

extern "C" int rd_bilinear_bwd(const float* dout, int32_t N, int32_t Ho, int32_t Wo, float* dd, int32_t Hs, int32_t Ws, void* stream) {
    RD_CHECK_ARG(dout && dd && N > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "bilinear_bwd: bad arguments");
    const float sh = Ho > 1 ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(ew_grid64((int64_t)N * Hs * Ws)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dout, N, Ho, Wo, dd, Hs, Ws, sh, sw);
    RD_CHECK_LAUNCH("bilinear_bwd_kernel");
    return RD_OK;
}
