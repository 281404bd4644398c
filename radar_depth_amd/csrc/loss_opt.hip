// Losses, the multistage radar filter and the optimizer step -- the tail of the reference's training-step body
// (main.py:416-445): MaskedL1Loss (evaluation/criteria_new.py:44-54), SmoothnessLoss (:8-28), the
// uncertainty-weighted total (main.py:423-429), Filter_layer (model/multistage_model.py:87-119) and
// torch.optim.SGD with momentum + weight decay (main.py:285-290,445).  All scalars stay on the device.
#include "common.h"

namespace rd {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    rd_sync();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    rd_sync();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;
}

// ---------------------------------------------------------------- masked L1
// SQ = false: MaskedL1Loss (sum |t-p|); SQ = true: MaskedMSELoss (criteria_new.py:31-41, sum (t-p)^2)
template <bool SQ>
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                         int64_t n, double* __restrict__ part) {
    __shared__ double sh[4];
    double s = 0.0, c = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float t = target[e];
        if (t > 0.f) {
            const float diff = t - pred[e];
            s += SQ ? (double)(diff * diff) : (double)fabsf(diff);
            c += 1.0;
        }
    }
    s = block_sum_d(s, sh);
    c = block_sum_d(c, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s;
        part[2 * blockIdx.x + 1] = c;
    }
}

__global__ __launch_bounds__(256) void pair_final_kernel(const double* __restrict__ part, int nblocks, int npairs,
                                                         double* __restrict__ out) {
    __shared__ double sh[4];
    for (int k = 0; k < npairs; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += part[(size_t)i * npairs + k];
        s = block_sum_d(s, sh);
        if (threadIdx.x == 0) out[k] = s;
    }
}

template <bool SQ>
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n,
                                                     const double* __restrict__ sums, const float* __restrict__ coef,
                                                     float* __restrict__ dpred, int accumulate) {
    const float k = (float)((double)coef[0] / sums[1]);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float t = target[e];
        float g = 0.f;
        if (t > 0.f) {
            const float diff = t - pred[e];
            g = SQ ? -2.f * diff * k : (diff > 0.f ? -k : (diff < 0.f ? k : 0.f));
        }
        dpred[e] = accumulate ? dpred[e] + g : g;
    }
}

// ---------------------------------------------------------------- evaluation metrics (evaluation/metrics.py:34-58)
// part[block][10]: count, sum d^2, sum |d|, sum |log10 o - log10 t|, sum |d|/t, #(r<1.25), #(r<1.25^2), #(r<1.25^3),
// sum (1/o-1/t)^2, sum |1/o-1/t|   over pixels with t > 0, d = o - t, r = max(o/t, t/o)
__global__ __launch_bounds__(256) void metrics_partial_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                              int64_t n, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float inv_ln10 = 0.4342944819032518f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float t = target[e];
        if (t > 0.f) {
            const float o = out[e];
            const float ad = fabsf(o - t);
            acc[0] += 1.0;
            acc[1] += (double)(ad * ad);
            acc[2] += (double)ad;
            acc[3] += (double)fabsf(logf(o) * inv_ln10 - logf(t) * inv_ln10);
            acc[4] += (double)(ad / t);
            const float r = fmaxf(o / t, t / o);
            if (r < 1.25f) acc[5] += 1.0;
            if (r < 1.25f * 1.25f) acc[6] += 1.0;
            if (r < 1.25f * 1.25f * 1.25f) acc[7] += 1.0;
            const float id = fabsf(1.f / o - 1.f / t);
            acc[8] += (double)(id * id);
            acc[9] += (double)id;
        }
    }
    for (int k = 0; k < 10; ++k) {
        const double s = block_sum_d(acc[k], sh);
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * 10 + k] = s;
    }
}

// ---------------------------------------------------------------- smoothness
// pass 1: per-sample sum of pred -> part[n][block]
__global__ __launch_bounds__(256) void smooth_sum_kernel(const float* __restrict__ pred, int64_t hw, double* __restrict__ part) {
    __shared__ double sh[4];
    const int n = blockIdx.y;
    const float* p = pred + (size_t)n * hw;
    double s = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < hw; e += (int64_t)gridDim.x * blockDim.x) s += (double)p[e];
    s = block_sum_d(s, sh);
    if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = s;
}
// scal[n] = mean + 1e-7 (as the reference computes it in fp32: mean(2).mean(3) then + 1e-7)
__global__ __launch_bounds__(256) void smooth_mean_kernel(const double* __restrict__ part, int nb, int64_t hw, float* __restrict__ sden) {
    __shared__ double sh[4];
    const int n = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[(size_t)n * nb + i];
    s = block_sum_d(s, sh);
    if (threadIdx.x == 0) sden[n] = (float)(s / (double)hw) + 1e-7f;
}

__device__ __forceinline__ float edge_w(const float* __restrict__ img, int C, int64_t hw, int64_t a, int64_t b) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += fabsf(img[c * hw + a] - img[c * hw + b]);
    return expf(-s / (float)C);
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// pass 2: q[n,y,x] = dL/d(dhat) ; partial sums (loss_x, loss_y, sum q*pred) per block
__global__ __launch_bounds__(256) void smooth_main_kernel(const float* __restrict__ pred, const float* __restrict__ image, int N,
                                                          int C, int H, int W, const float* __restrict__ sden,
                                                          float* __restrict__ q, double* __restrict__ part) {
    __shared__ double sh[4];
    const int n = blockIdx.y;
    const int64_t hw = (int64_t)H * W;
    const float* p = pred + (size_t)n * hw;
    const float* img = image + (size_t)n * C * hw;
    const float inv = 1.f / sden[n];
    const float inx = 1.f / ((float)N * H * (W - 1)), iny = 1.f / ((float)N * (H - 1) * W);
    double lx = 0.0, ly = 0.0, tq = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < hw; e += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)((unsigned)e / (unsigned)W), x = (int)((unsigned)e - (unsigned)y * (unsigned)W);   // (one image: e < 2^31, checked by the host)
        const float d = p[e] * inv;
        float qq = 0.f;
        if (x + 1 < W) {
            const float w = edge_w(img, C, hw, e, e + 1);
            const float diff = d - p[e + 1] * inv;
            lx += (double)(fabsf(diff) * w);
            qq += w * sgn(diff) * inx;
        }
        if (x > 0) {
            const float w = edge_w(img, C, hw, e - 1, e);
            qq -= w * sgn(p[e - 1] * inv - d) * inx;
        }
        if (y + 1 < H) {
            const float w = edge_w(img, C, hw, e, e + W);
            const float diff = d - p[e + W] * inv;
            ly += (double)(fabsf(diff) * w);
            qq += w * sgn(diff) * iny;
        }
        if (y > 0) {
            const float w = edge_w(img, C, hw, e - W, e);
            qq -= w * sgn(p[e - W] * inv - d) * iny;
        }
        q[(size_t)n * hw + e] = qq;
        tq += (double)qq * (double)p[e];
    }
    lx = block_sum_d(lx, sh);
    ly = block_sum_d(ly, sh);
    tq = block_sum_d(tq, sh);
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)n * gridDim.x + blockIdx.x) * 3;
        o[0] = lx; o[1] = ly; o[2] = tq;
    }
}
// out[0] = loss ; tsum[n] = sum_k q_k p_k
__global__ __launch_bounds__(256) void smooth_final_kernel(const double* __restrict__ part, int N, int nb, int H, int W,
                                                           double* __restrict__ out, double* __restrict__ tsum) {
    __shared__ double sh[4];
    double LX = 0.0, LY = 0.0;
    for (int n = 0; n < N; ++n) {
        double lx = 0.0, ly = 0.0, t = 0.0;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {
            const double* o = part + ((size_t)n * nb + i) * 3;
            lx += o[0]; ly += o[1]; t += o[2];
        }
        lx = block_sum_d(lx, sh);
        ly = block_sum_d(ly, sh);
        t = block_sum_d(t, sh);
        LX += lx; LY += ly;
        if (threadIdx.x == 0) tsum[n] = t;
    }
    if (threadIdx.x == 0) out[0] = LX / ((double)N * H * (W - 1)) + LY / ((double)N * (H - 1) * W);
}
// dpred = coef * (q / s - T / (HW s^2))
__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float* __restrict__ q, const float* __restrict__ sden,
                                                         const double* __restrict__ tsum, int64_t hw, const float* __restrict__ coef,
                                                         float* __restrict__ dpred, int accumulate) {
    const int n = blockIdx.y;
    const float k = coef[0];
    const float s = sden[n];
    const float a = k / s, b = (float)((double)k * tsum[n] / ((double)hw * (double)s * (double)s));
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < hw; e += (int64_t)gridDim.x * blockDim.x) {
        const size_t i = (size_t)n * hw + e;
        const float g = a * q[i] - b;
        dpred[i] = accumulate ? dpred[i] + g : g;
    }
}

// ---------------------------------------------------------------- filter layer / totals / optimizer
__global__ __launch_bounds__(256) void radar_filter_kernel(const float* __restrict__ x, int Ctot, int c, int64_t hw, int N,
                                                           const float* __restrict__ dense, float* __restrict__ kept,
                                                           float* __restrict__ mask, float log_ratio, float log_alpha) {
    const int64_t total = (int64_t)N * hw;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = e / hw, r = e - n * hw;
        const float sp = x[((size_t)n * Ctot + c) * hw + r];
        const float de = dense[e];
        const float thr = expf(((de * log_ratio) / 100.0f) + log_alpha);
        const float m = fabsf(de - sp) <= thr ? 1.f : 0.f;
        kept[e] = sp * m;
        mask[e] = m;
    }
}

__global__ void uncertainty_total_kernel(const double* s1, const double* s2, const double* sm, const float* w1, const float* w2,
                                         float w_smooth, float* loss4, float* coefs3, float* dw1, float* dw2) {
    const double d1 = s1[0] / s1[1], d2 = s2[0] / s2[1], s = sm[0];
    const double e1 = exp(-(double)w1[0]), e2 = exp(-(double)w2[0]);
    const double st1 = d1 + (double)w_smooth * s;
    loss4[0] = (float)d1; loss4[1] = (float)d2; loss4[2] = (float)s;
    loss4[3] = (float)(e1 * st1 + e2 * d2 + (double)w1[0] + (double)w2[0]);
    coefs3[0] = (float)e1; coefs3[1] = (float)(w_smooth * e1); coefs3[2] = (float)e2;
    dw1[0] = (float)(1.0 - e1 * st1);
    dw2[0] = (float)(1.0 - e2 * d2);
}
__global__ void l1_total_kernel(const double* sums, float* loss, float* coef) {
    loss[0] = (float)(sums[0] / sums[1]);
    coef[0] = 1.f;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n4, int64_t n, float lr, float momentum, float wd, float gscale, int first) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[e];
        const float4 gv = reinterpret_cast<const float4*>(g)[e];
        float4 b;
        float4 d = make_float4(fmaf(wd, pv.x, gscale * gv.x), fmaf(wd, pv.y, gscale * gv.y), fmaf(wd, pv.z, gscale * gv.z),
                               fmaf(wd, pv.w, gscale * gv.w));
        if (first) b = d;
        else {
            b = reinterpret_cast<float4*>(buf)[e];
            b.x = fmaf(momentum, b.x, d.x); b.y = fmaf(momentum, b.y, d.y); b.z = fmaf(momentum, b.z, d.z); b.w = fmaf(momentum, b.w, d.w);
        }
        reinterpret_cast<float4*>(buf)[e] = b;
        pv.x -= lr * b.x; pv.y -= lr * b.y; pv.z -= lr * b.z; pv.w -= lr * b.w;
        reinterpret_cast<float4*>(p)[e] = pv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        const float d = fmaf(wd, p[e], gscale * g[e]);
        const float b = first ? d : fmaf(momentum, buf[e], d);
        buf[e] = b;
        p[e] -= lr * b;
    }
}

static int red_grid(int64_t n) {
    int64_t g = cdiv64(n, 256 * 8);
    if (g > RED_BLOCKS) g = RED_BLOCKS;
    return (int)(g < 1 ? 1 : g);
}
static int ew_grid2(int64_t n) {
    int64_t g = cdiv64(n, 256);
    const int64_t cap = (int64_t)num_cus() * 16;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}
constexpr int SM_BLOCKS = 256;  // blocks per sample in the smoothness passes (64 left the b = 8 passes of config 4 at two blocks per CU: smooth_main 157 us)

}  // namespace rd
using namespace rd;

extern "C" int rd_loss_tiles(int64_t n) { return red_grid(n); }

// ws: 2*rd_loss_tiles(n) doubles (pass as float* with 4*tiles floats, 8-byte aligned)
extern "C" int rd_masked_l1_sums(const float* pred, const float* target, int64_t n, float* ws, double* sums, void* stream) {
    RD_CHECK_ARG(pred && target && ws && sums && n > 0 && ((uintptr_t)ws & 7) == 0, "masked_l1_sums: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int g = red_grid(n);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(l1_partial_kernel<false>, dim3(g), dim3(256), 0, s, pred, target, n, part);
    RD_CHECK_LAUNCH("l1_partial_kernel");
    hipLaunchKernelGGL(pair_final_kernel, dim3(1), dim3(256), 0, s, part, g, 2, sums);
    RD_CHECK_LAUNCH("pair_final_kernel");
    return RD_OK;
}
// MaskedMSELoss (criteria_new.py:31-41): sums[0] = sum (t-p)^2 over t>0, sums[1] = count
extern "C" int rd_masked_l2_sums(const float* pred, const float* target, int64_t n, float* ws, double* sums, void* stream) {
    RD_CHECK_ARG(pred && target && ws && sums && n > 0 && ((uintptr_t)ws & 7) == 0, "masked_l2_sums: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int g = red_grid(n);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(l1_partial_kernel<true>, dim3(g), dim3(256), 0, s, pred, target, n, part);
    RD_CHECK_LAUNCH("l2_partial_kernel");
    hipLaunchKernelGGL(pair_final_kernel, dim3(1), dim3(256), 0, s, part, g, 2, sums);
    RD_CHECK_LAUNCH("pair_final_kernel");
    return RD_OK;
}

// sums[10] as listed above; ws: 10*rd_loss_tiles(n) doubles
extern "C" int rd_depth_metrics(const float* output, const float* target, int64_t n, float* ws, double* sums, void* stream) {
    RD_CHECK_ARG(output && target && ws && sums && n > 0 && ((uintptr_t)ws & 7) == 0, "depth_metrics: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int g = red_grid(n);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(g), dim3(256), 0, s, output, target, n, part);
    RD_CHECK_LAUNCH("metrics_partial_kernel");
    hipLaunchKernelGGL(pair_final_kernel, dim3(1), dim3(256), 0, s, part, g, 10, sums);
    RD_CHECK_LAUNCH("pair_final_kernel");
    return RD_OK;
}

extern "C" int rd_masked_l1_bwd(const float* pred, const float* target, int64_t n, const double* sums, const float* coef,
                                float* dpred, int32_t accumulate, void* stream) {
    RD_CHECK_ARG(pred && target && sums && coef && dpred && n > 0, "masked_l1_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel<false>, dim3(ew_grid2(n)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, target, n, sums, coef,
                       dpred, accumulate);
    RD_CHECK_LAUNCH("l1_bwd_kernel");
    return RD_OK;
}
// dpred = coef * 2 (pred - target) / count on valid pixels, 0 elsewhere
extern "C" int rd_masked_l2_bwd(const float* pred, const float* target, int64_t n, const double* sums, const float* coef,
                                float* dpred, int32_t accumulate, void* stream) {
    RD_CHECK_ARG(pred && target && sums && coef && dpred && n > 0, "masked_l2_bwd: bad arguments");
    hipLaunchKernelGGL(l1_bwd_kernel<true>, dim3(ew_grid2(n)), dim3(256), 0, static_cast<hipStream_t>(stream), pred, target, n, sums, coef,
                       dpred, accumulate);
    RD_CHECK_LAUNCH("l2_bwd_kernel");
    return RD_OK;
}

// workspace layout (floats): q[N*H*W] | sden[N] (padded to even) | doubles: tsum[N], part[N*SM_BLOCKS*3]
extern "C" int64_t rd_smooth_workspace_floats(int32_t N, int32_t H, int32_t W) {
    const int64_t nhw = (int64_t)N * H * W;
    return ((nhw + 1) & ~(int64_t)1) + ((N + 1) & ~1) + 2 * ((int64_t)N + (int64_t)N * SM_BLOCKS * 3);
}

static void smooth_carve(float* ws, int N, int H, int W, float*& q, float*& sden, double*& tsum, double*& part) {
    const int64_t nhw = (int64_t)N * H * W;
    q = ws;
    sden = ws + ((nhw + 1) & ~(int64_t)1);
    tsum = reinterpret_cast<double*>(sden + ((N + 1) & ~1));
    part = tsum + N;
}

extern "C" int rd_smooth_fwd(const float* pred, const float* image, int32_t N, int32_t C, int32_t H, int32_t W, float* ws,
                             double* out, void* stream) {
    RD_CHECK_ARG(pred && image && ws && out && N > 0 && C > 0 && H > 1 && W > 1 && ((uintptr_t)ws & 7) == 0, "smooth_fwd: bad arguments");
    RD_CHECK_ARG((int64_t)H * W < (1ll << 31), "smooth_fwd: one image must have fewer than 2^31 pixels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *q, *sden;
    double *tsum, *part;
    smooth_carve(ws, N, H, W, q, sden, tsum, part);
    const int64_t hw = (int64_t)H * W;
    hipLaunchKernelGGL(smooth_sum_kernel, dim3(SM_BLOCKS, N), dim3(256), 0, s, pred, hw, part);
    RD_CHECK_LAUNCH("smooth_sum_kernel");
    hipLaunchKernelGGL(smooth_mean_kernel, dim3(N), dim3(256), 0, s, part, SM_BLOCKS, hw, sden);
    RD_CHECK_LAUNCH("smooth_mean_kernel");
    hipLaunchKernelGGL(smooth_main_kernel, dim3(SM_BLOCKS, N), dim3(256), 0, s, pred, image, N, C, H, W, sden, q, part);
    RD_CHECK_LAUNCH("smooth_main_kernel");
    hipLaunchKernelGGL(smooth_final_kernel, dim3(1), dim3(256), 0, s, part, N, SM_BLOCKS, H, W, out, tsum);
    RD_CHECK_LAUNCH("smooth_final_kernel");
    return RD_OK;
}

extern "C" int rd_smooth_bwd(int32_t N, int32_t H, int32_t W, const float* ws, const float* coef, float* dpred, int32_t accumulate,
                             void* stream) {
    RD_CHECK_ARG(ws && coef && dpred && N > 0, "smooth_bwd: bad arguments");
    float *q, *sden;
    double *tsum, *part;
    smooth_carve(const_cast<float*>(ws), N, H, W, q, sden, tsum, part);
    const int64_t hw = (int64_t)H * W;
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(SM_BLOCKS, N), dim3(256), 0, static_cast<hipStream_t>(stream), q, sden, tsum, hw,
                       coef, dpred, accumulate);
    RD_CHECK_LAUNCH("smooth_bwd_kernel");
    return RD_OK;
}

extern "C" int rd_radar_filter(const float* x, int32_t N, int32_t Ctot, int32_t c, int64_t hw, const float* dense, float* kept,
                               float* mask, void* stream) {
    RD_CHECK_ARG(x && dense && kept && mask && N > 0 && c >= 0 && c < Ctot && hw > 0, "radar_filter: bad arguments");
    // fp32 constants exactly as the reference's 0-dim tensors evaluate them: log(18/5), log(5)
    const float log_ratio = logf(18.0f / 5.0f), log_alpha = logf(5.0f);
    hipLaunchKernelGGL(radar_filter_kernel, dim3(ew_grid2((int64_t)N * hw)), dim3(256), 0, static_cast<hipStream_t>(stream), x, Ctot,
                       c, hw, N, dense, kept, mask, log_ratio, log_alpha);
    RD_CHECK_LAUNCH("radar_filter_kernel");
    return RD_OK;
}

extern "C" int rd_uncertainty_total(const double* sums1, const double* sums2, const double* smooth, const float* w1, const float* w2,
                                    float w_smooth, float* loss4, float* coefs3, float* dw1, float* dw2, void* stream) {
    RD_CHECK_ARG(sums1 && sums2 && smooth && w1 && w2 && loss4 && coefs3 && dw1 && dw2, "uncertainty_total: null argument");
    hipLaunchKernelGGL(uncertainty_total_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), sums1, sums2, smooth, w1, w2,
                       w_smooth, loss4, coefs3, dw1, dw2);
    RD_CHECK_LAUNCH("uncertainty_total_kernel");
    return RD_OK;
}

extern "C" int rd_l1_total(const double* sums, float* loss, float* coef, void* stream) {
    RD_CHECK_ARG(sums && loss && coef, "l1_total: null argument");
    hipLaunchKernelGGL(l1_total_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), sums, loss, coef);
    RD_CHECK_LAUNCH("l1_total_kernel");
    return RD_OK;
}

extern "C" int rd_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd, float grad_scale,
                           int32_t first_step, void* stream) {
    RD_CHECK_ARG(p && g && buf && n > 0 && ((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)buf & 15) == 0,
                 "sgd_step: bad arguments (arenas must be 16-byte aligned)");
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid2(n / 4 + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, buf, n / 4, n, lr,
                       momentum, wd, grad_scale, first_step);
    RD_CHECK_LAUNCH("sgd_kernel");
    return RD_OK;
}
