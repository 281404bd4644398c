// Training-mode BatchNorm2d + activation + residual joins, and the stem's fused BN/act/max-pool,
// as HBM-streaming NHWC kernels (float4 over channels, per-channel coefficients).
// Replaces nn.BatchNorm2d / ReLU / LeakyReLU / `out += identity` / MaxPool2d(3,2,1) and their backward
// (model/models.py:96-112,203-208,633-650).
#include "common.h"

namespace rd {

// ------------------------------------------------------------------------------------------------
// finalize: reduce [n_tiles][2][C] partial (sum, sumsq) in double -> mean/invstd/scale/shift, running stats
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int n_tiles, int ld, int c0, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* running_mean, float* running_var,
                                                          int64_t* nbt, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    // four tiles per trip with all eight loads issued before the first add: the kernel is a chain of L2 round trips (it sits on
    // the critical path between a convolution and its BatchNorm 54 times per step), so the trips are what there is to save
    int t = threadIdx.x;
    for (; t + 3 * 256 < n_tiles; t += 4 * 256) {
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = part[((size_t)(t + k * 256) * 2 + 0) * ld + c0 + c];
            b[k] = part[((size_t)(t + k * 256) * 2 + 1) * ld + c0 + c];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += (double)a[k]; q += (double)b[k]; }
    }
    for (; t < n_tiles; t += 256) {
        s += (double)part[((size_t)t * 2 + 0) * ld + c0 + c];
        q += (double)part[((size_t)t * 2 + 1) * ld + c0 + c];
    }
    __shared__ double sh[2][4];
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
    rd_sync();
    if (threadIdx.x == 0) {
        s = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double m = s / count;
        double var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        const double r = 1.0 / sqrt(var + (double)eps);
        const double g = gamma[c], b = beta[c];
        mean[c] = (float)m;
        invstd[c] = (float)r;
        scale[c] = (float)(g * r);
        shift[c] = (float)(b - m * g * r);
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
        if (c == 0 && nbt) *nbt += 1;
    }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float r = 1.0f / sqrtf(rv[c] + eps);
        scale[c] = gamma[c] * r;
        shift[c] = beta[c] - rm[c] * gamma[c] * r;
    }
}

// every folded BatchNorm of an inference plan in ONE launch (54 launches of ~4 us each were 18 % of the batch-1 eval forward)
struct EvalCoefJob {
    const float *gamma, *beta, *rm, *rv;
    float *scale, *shift;
    int C, pad_;
};
__global__ __launch_bounds__(256) void bn_eval_coeffs_batched_kernel(const EvalCoefJob* __restrict__ jobs, float eps) {
    const EvalCoefJob j = jobs[blockIdx.x];
    for (int c = threadIdx.x; c < j.C; c += blockDim.x) {
        const float r = 1.0f / sqrtf(j.rv[c] + eps);       // same expression as bn_eval_coeffs_kernel: bit-identical results
        j.scale[c] = j.gamma[c] * r;
        j.shift[c] = j.beta[c] - j.rm[c] * j.gamma[c] * r;
    }
}

// ------------------------------------------------------------------------------------------------
// row-block reductions: thread (q = channel quad, rl = row lane); block covers rows [b*RPB, (b+1)*RPB)
static inline int rows_per_block(int64_t M) {
    int64_t r = cdiv64(M, 2048);
    return (int)(r < 64 ? 64 : r);
}
// Backward reductions: a block reduces RPB rows 256 / (C/4) at a time, so with many channels 64 rows are a long serial walk of
// single 16-byte loads (C = 512: 32 round trips; the layer4 / bn_fusion reductions ran at 1.1 TB/s, 21 us for 25 MB).  Four row
// groups per block keep the loop short and the grid wide: C = 64 -> 64 rows as before, 128 -> 32, 256 -> 16, 512 -> 8.
static inline int bwd_rows_per_block(int64_t M, int C) {
    const int RL = 256 / (C / 4) > 0 ? 256 / (C / 4) : 1;
    int64_t r = cdiv64(M, 2048);
    const int64_t lo = 4 * RL < 64 ? 4 * RL : 64;
    return (int)(r < lo ? lo : r);
}

template <int NS>  // number of per-channel sums
__device__ __forceinline__ void block_reduce_store(float4 (&acc)[NS], int Q, int RL, int q, int rl, bool active, float* out,
                                                   int C, float* sm) {
    // sm: [RL][NS][Q*4]
    if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<float4*>(sm + ((size_t)(rl * NS + s) * Q + q) * 4) = acc[s];
    }
    rd_sync();
    for (int e = threadIdx.x; e < NS * Q * 4; e += blockDim.x) {
        const int s = e / (Q * 4), c = e - s * (Q * 4);
        float v = 0.f;
        for (int r = 0; r < RL; ++r) v += sm[(size_t)(r * NS + s) * Q * 4 + c];
        out[(size_t)s * C + c] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int64_t M, int C, int ldx, int RPB,
                                                       float* __restrict__ part) {
    extern __shared__ float sm[];
    const int Q = C >> 2, RL = 256 / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const bool active = rl < RL;
    const int64_t r0 = (int64_t)blockIdx.x * RPB, r1 = r0 + RPB < M ? r0 + RPB : M;
    float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (active)
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const float4 v = ld4(x + r * ldx + q * 4);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
            acc[1].x += v.x * v.x; acc[1].y += v.y * v.y; acc[1].z += v.z * v.z; acc[1].w += v.w * v.w;
        }
    block_reduce_store<2>(acc, Q, RL, q, rl, active, part + (size_t)blockIdx.x * 2 * C, C, sm);
}

// y = act(s1*x1 + t1 [+ (s2*x2 + t2 | x2)])
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ x1, int ldx1, const float* __restrict__ s1,
                                                     const float* __restrict__ t1, const T* __restrict__ x2, int ldx2,
                                                     const float* __restrict__ s2, const float* __restrict__ t2,
                                                     T* __restrict__ y, int ldy, int64_t M, int C, int act,
                                                     unsigned short* __restrict__ pc, int64_t pplane) {
    const int Q = C >> 2;
    const int lq = quad_log2(Q);
    const int64_t total = M * Q;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int c;
        split_quad(e, Q, lq, r, c);
        const float4 v = ld4(x1 + r * ldx1 + c);
        const float4 a = *reinterpret_cast<const float4*>(s1 + c);
        const float4 b = *reinterpret_cast<const float4*>(t1 + c);
        float4 z = make_float4(fmaf(a.x, v.x, b.x), fmaf(a.y, v.y, b.y), fmaf(a.z, v.z, b.z), fmaf(a.w, v.w, b.w));
        if (x2) {
            float4 u = ld4(x2 + r * ldx2 + c);
            if (s2) {
                const float4 a2 = *reinterpret_cast<const float4*>(s2 + c);
                const float4 b2 = *reinterpret_cast<const float4*>(t2 + c);
                u = make_float4(fmaf(a2.x, u.x, b2.x), fmaf(a2.y, u.y, b2.y), fmaf(a2.z, u.z, b2.z), fmaf(a2.w, u.w, b2.w));
            }
            z.x += u.x; z.y += u.y; z.z += u.z; z.w += u.w;
        }
        z.x = act_fwd(z.x, act); z.y = act_fwd(z.y, act); z.z = act_fwd(z.z, act); z.w = act_fwd(z.w, act);
        st4(y + r * ldy + c, z);
        // the consumers' split arithmetic, done here where the VALU is idle (HBM-bound pass): three bf16 piece planes of y
        if (pc) store_pieces4(pc, pplane, M, r, c, z);
    }
}

// backward pass 1: g = dy * act'(y); sums of g, g*(x1-m1), g*(x2-m2).
// y == nullptr with an activation: the output was act(s1*x1 + t1) alone (no second operand), so its sign is recomputed from
// x1 -- which the kernel reads anyway -- with the forward kernel's own fmaf, instead of reading the activation tensor.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y,
                                                            int ldy, const T* __restrict__ x1, int ldx1,
                                                            const float* __restrict__ m1, const T* __restrict__ x2, int ldx2,
                                                            const float* __restrict__ m2, T* __restrict__ g, int ldg,
                                                            int64_t M, int C, int act, int RPB, float* __restrict__ part,
                                                            const float* __restrict__ s1, const float* __restrict__ t1,
                                                            const float* __restrict__ s2, const float* __restrict__ t2) {
    extern __shared__ float sm[];
    const int Q = C >> 2, RL = 256 / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const bool active = rl < RL;
    const int64_t r0 = (int64_t)blockIdx.x * RPB, r1 = r0 + RPB < M ? r0 + RPB : M;
    float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (active) {
        const int c = q * 4;
        const float4 mu1 = x1 ? *reinterpret_cast<const float4*>(m1 + c) : make_float4(0, 0, 0, 0);
        const float4 mu2 = x2 ? *reinterpret_cast<const float4*>(m2 + c) : make_float4(0, 0, 0, 0);
        const bool from_x = act != RD_ACT_NONE && !y;
        const float4 sa = from_x ? *reinterpret_cast<const float4*>(s1 + c) : make_float4(0, 0, 0, 0);
        const float4 sb = from_x ? *reinterpret_cast<const float4*>(t1 + c) : make_float4(0, 0, 0, 0);
        const bool from_x2 = from_x && x2 && s2;        // act(bn1(x1) + bn2(x2)): both operands are read anyway
        const float4 sa2 = from_x2 ? *reinterpret_cast<const float4*>(s2 + c) : make_float4(0, 0, 0, 0);
        const float4 sb2 = from_x2 ? *reinterpret_cast<const float4*>(t2 + c) : make_float4(0, 0, 0, 0);
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            float4 gv = ld4(dy + r * lddy + c);
            float4 v = make_float4(0, 0, 0, 0), v2 = make_float4(0, 0, 0, 0);
            if (x1) v = ld4(x1 + r * ldx1 + c);
            if (x2) v2 = ld4(x2 + r * ldx2 + c);
            if (act != RD_ACT_NONE) {
                float4 yv;
                if (from_x) {
                    yv = make_float4(fmaf(sa.x, v.x, sb.x), fmaf(sa.y, v.y, sb.y), fmaf(sa.z, v.z, sb.z), fmaf(sa.w, v.w, sb.w));
                    if (from_x2) {     // same order of operations as bn_act_kernel: z1 + fma(s2, x2, t2)
                        yv.x += fmaf(sa2.x, v2.x, sb2.x); yv.y += fmaf(sa2.y, v2.y, sb2.y);
                        yv.z += fmaf(sa2.z, v2.z, sb2.z); yv.w += fmaf(sa2.w, v2.w, sb2.w);
                    }
                } else yv = ld4(y + r * ldy + c);
                gv.x *= act_grad_from_out(yv.x, act); gv.y *= act_grad_from_out(yv.y, act);
                gv.z *= act_grad_from_out(yv.z, act); gv.w *= act_grad_from_out(yv.w, act);
            }
            if (g) st4(g + r * ldg + c, gv);
            acc[0].x += gv.x; acc[0].y += gv.y; acc[0].z += gv.z; acc[0].w += gv.w;
            if (x1) {
                acc[1].x += gv.x * (v.x - mu1.x); acc[1].y += gv.y * (v.y - mu1.y);
                acc[1].z += gv.z * (v.z - mu1.z); acc[1].w += gv.w * (v.w - mu1.w);
            }
            if (x2) {
                acc[2].x += gv.x * (v2.x - mu2.x); acc[2].y += gv.y * (v2.y - mu2.y);
                acc[2].z += gv.z * (v2.z - mu2.z); acc[2].w += gv.w * (v2.w - mu2.w);
            }
        }
    }
    block_reduce_store<3>(acc, Q, RL, q, rl, active, part + (size_t)blockIdx.x * 3 * C, C, sm);
}

// per channel: finish the reduction; dgamma, dbeta; coefficients of dx = A*g + B*(x-mean) + Cc
__global__ __launch_bounds__(256) void bn_bwd_coeffs_kernel(const float* __restrict__ part, int n_tiles, int C, int which,
                                                            double count, const float* __restrict__ gamma,
                                                            const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                                            float* coef) {
    const int c = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    int t = threadIdx.x;
    for (; t + 3 * 256 < n_tiles; t += 4 * 256) {       // (see bn_finalize_kernel: one round trip for four tiles)
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = part[((size_t)(t + k * 256) * 3 + 0) * C + c];
            b[k] = part[((size_t)(t + k * 256) * 3 + which) * C + c];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0 += (double)a[k]; s1 += (double)b[k]; }
    }
    for (; t < n_tiles; t += 256) {
        s0 += (double)part[((size_t)t * 3 + 0) * C + c];
        s1 += (double)part[((size_t)t * 3 + which) * C + c];
    }
    __shared__ double sh[2][4];
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s0; sh[1][w] = s1; }
    rd_sync();
    if (threadIdx.x == 0) {
        s0 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        s1 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double r = invstd[c], gm = gamma[c];
        if (dgamma) dgamma[c] = (float)(r * s1);
        if (dbeta) dbeta[c] = (float)s0;
        const double A = gm * r;
        coef[c] = (float)A;
        coef[C + c] = (float)(-gm * r * r * r * s1 / count);
        coef[2 * C + c] = (float)(-A * s0 / count);
    }
}

// s1 != nullptr: g is the raw output gradient dy of act(s1*x + t1); the activation factor is applied here (the reduce pass then
// does not have to write the masked gradient)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ x, int ldx,
                                                        const float* __restrict__ mean, const float* __restrict__ coef,
                                                        T* __restrict__ dx, int lddx, int64_t M, int C,
                                                        const float* __restrict__ s1, const float* __restrict__ t1, int act,
                                                        unsigned short* __restrict__ pc, int64_t pplane) {
    const int Q = C >> 2;
    const int lq = quad_log2(Q);
    const int64_t total = M * Q;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int c;
        split_quad(e, Q, lq, r, c);
        float4 gv = ld4(g + r * ldg + c);
        const float4 xv = ld4(x + r * ldx + c);
        if (s1) {
            const float4 sa = *reinterpret_cast<const float4*>(s1 + c), sb = *reinterpret_cast<const float4*>(t1 + c);
            gv.x *= act_grad_from_out(fmaf(sa.x, xv.x, sb.x), act); gv.y *= act_grad_from_out(fmaf(sa.y, xv.y, sb.y), act);
            gv.z *= act_grad_from_out(fmaf(sa.z, xv.z, sb.z), act); gv.w *= act_grad_from_out(fmaf(sa.w, xv.w, sb.w), act);
        }
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 A = *reinterpret_cast<const float4*>(coef + c);
        const float4 B = *reinterpret_cast<const float4*>(coef + C + c);
        const float4 K = *reinterpret_cast<const float4*>(coef + 2 * C + c);
        float4 o;
        o.x = fmaf(A.x, gv.x, fmaf(B.x, xv.x - mu.x, K.x));
        o.y = fmaf(A.y, gv.y, fmaf(B.y, xv.y - mu.y, K.y));
        o.z = fmaf(A.z, gv.z, fmaf(B.z, xv.z - mu.z, K.z));
        o.w = fmaf(A.w, gv.w, fmaf(B.w, xv.w - mu.w, K.w));
        st4(dx + r * lddx + c, o);
        if (pc) store_pieces4(pc, pplane, M, r, c, o);
    }
}

// out = act(bn1(x1) + bn2(x2)): both input gradients in one pass from the raw output gradient (the masked gradient is never
// materialised, dy is read once): dx_i = A_i*g + B_i*(x_i - mean_i) + K_i with g = dy * act'(s1*x1+t1 + s2*x2+t2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dx2_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x1, int ldx1,
                                                         const T* __restrict__ x2, int ldx2, const float* __restrict__ m1,
                                                         const float* __restrict__ m2, const float* __restrict__ coef1,
                                                         const float* __restrict__ coef2, const float* __restrict__ s1,
                                                         const float* __restrict__ t1, const float* __restrict__ s2,
                                                         const float* __restrict__ t2, int act, T* __restrict__ dx1, int lddx1,
                                                         T* __restrict__ dx2, int lddx2, int64_t M, int C,
                                                         unsigned short* __restrict__ pc1, int64_t pplane1,
                                                         unsigned short* __restrict__ pc2, int64_t pplane2) {
    const int Q = C >> 2;
    const int lq = quad_log2(Q);
    const int64_t total = M * Q;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int c;
        split_quad(e, Q, lq, r, c);
        float4 gv = ld4(dy + r * lddy + c);
        const float4 v1 = ld4(x1 + r * ldx1 + c), v2 = ld4(x2 + r * ldx2 + c);
        const float4 a1 = *reinterpret_cast<const float4*>(s1 + c), b1 = *reinterpret_cast<const float4*>(t1 + c);
        const float4 a2 = *reinterpret_cast<const float4*>(s2 + c), b2 = *reinterpret_cast<const float4*>(t2 + c);
        gv.x *= act_grad_from_out(fmaf(a1.x, v1.x, b1.x) + fmaf(a2.x, v2.x, b2.x), act);
        gv.y *= act_grad_from_out(fmaf(a1.y, v1.y, b1.y) + fmaf(a2.y, v2.y, b2.y), act);
        gv.z *= act_grad_from_out(fmaf(a1.z, v1.z, b1.z) + fmaf(a2.z, v2.z, b2.z), act);
        gv.w *= act_grad_from_out(fmaf(a1.w, v1.w, b1.w) + fmaf(a2.w, v2.w, b2.w), act);
        const float4 mu1 = *reinterpret_cast<const float4*>(m1 + c), mu2 = *reinterpret_cast<const float4*>(m2 + c);
        const float4 A1 = *reinterpret_cast<const float4*>(coef1 + c), B1 = *reinterpret_cast<const float4*>(coef1 + C + c),
                     K1 = *reinterpret_cast<const float4*>(coef1 + 2 * C + c);
        const float4 A2 = *reinterpret_cast<const float4*>(coef2 + c), B2 = *reinterpret_cast<const float4*>(coef2 + C + c),
                     K2 = *reinterpret_cast<const float4*>(coef2 + 2 * C + c);
        float4 o;
        o.x = fmaf(A1.x, gv.x, fmaf(B1.x, v1.x - mu1.x, K1.x)); o.y = fmaf(A1.y, gv.y, fmaf(B1.y, v1.y - mu1.y, K1.y));
        o.z = fmaf(A1.z, gv.z, fmaf(B1.z, v1.z - mu1.z, K1.z)); o.w = fmaf(A1.w, gv.w, fmaf(B1.w, v1.w - mu1.w, K1.w));
        st4(dx1 + r * lddx1 + c, o);
        if (pc1) store_pieces4(pc1, pplane1, M, r, c, o);
        o.x = fmaf(A2.x, gv.x, fmaf(B2.x, v2.x - mu2.x, K2.x)); o.y = fmaf(A2.y, gv.y, fmaf(B2.y, v2.y - mu2.y, K2.y));
        o.z = fmaf(A2.z, gv.z, fmaf(B2.z, v2.z - mu2.z, K2.z)); o.w = fmaf(A2.w, gv.w, fmaf(B2.w, v2.w - mu2.w, K2.w));
        st4(dx2 + r * lddx2 + c, o);
        if (pc2) store_pieces4(pc2, pplane2, M, r, c, o);
    }
}

// ------------------------------------------------------------------------------------------------
// stem: y = maxpool3x3/2/1(act(scale*x+shift)), argmax position saved (first max wins, like ATen's CPU kernel)
template <typename T>
__global__ __launch_bounds__(256) void bnact_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int act, int N, int H, int W,
                                                                int C, int Ho, int Wo, T* __restrict__ y, int ldy,
                                                                uint8_t* __restrict__ idx, unsigned short* __restrict__ pc, int64_t pplane) {
    const int Q = C >> 2;
    const int lq = quad_log2(Q);
    const int64_t total = (int64_t)N * Ho * Wo * Q;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int c;
        split_quad(e, Q, lq, r, c);
        int ow, oh, n;
        if (r < (1ll << 31)) {          // 32-bit divisions (a 64-bit one is ~100 instructions)
            const unsigned u = (unsigned)r, q1 = u / (unsigned)Wo;
            ow = (int)(u - q1 * (unsigned)Wo);
            n = (int)(q1 / (unsigned)Ho);
            oh = (int)(q1 - (unsigned)n * (unsigned)Ho);
        } else {
            ow = (int)(r % Wo); r /= Wo;
            oh = (int)(r % Ho);
            n = (int)(r / Ho);
        }
        const float4 a = *reinterpret_cast<const float4*>(scale + c);
        const float4 b = *reinterpret_cast<const float4*>(shift + c);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = 2 * oh - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = 2 * ow - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                const float4 v = ld4(x + (((size_t)n * H + ih) * W + iw) * C + c);
                const float z[4] = {act_fwd(fmaf(a.x, v.x, b.x), act), act_fwd(fmaf(a.y, v.y, b.y), act),
                                    act_fwd(fmaf(a.z, v.z, b.z), act), act_fwd(fmaf(a.w, v.w, b.w), act)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (first || z[j] > best[j]) { best[j] = z[j]; bi[j] = kh * 3 + kw; }
                first = false;
            }
        }
        const size_t o = ((size_t)n * Ho + oh) * Wo + ow;
        st4(y + o * ldy + c, make_float4(best[0], best[1], best[2], best[3]));
        *reinterpret_cast<uchar4*>(idx + o * C + c) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
        if (pc) store_pieces4(pc, pplane, (int64_t)N * Ho * Wo, (int64_t)o, c, make_float4(best[0], best[1], best[2], best[3]));
    }
}

// g[n,h,w,c] = act'(scale*x+shift) * sum over windows whose argmax is (h,w) of dy.
template <typename T> __device__ __forceinline__ float round_store(float v);
template <> __device__ __forceinline__ float round_store<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_store<bf16s>(float v) { return __uint_as_float((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v) << 16); }

// part != nullptr: also the BatchNorm-backward sums of the stem (sum g, sum g*(x - mean)) per block, [block][3][C] like
// bn_bwd_reduce_kernel -- g and x are in registers here, so the separate reduce pass over both (the largest tensors of the
// network) disappears.  Needs 256 % (C/4) == 0 so that a thread keeps its channel quad across the grid-stride loop.
template <typename T>
__global__ __launch_bounds__(256) void bnact_maxpool_bwd_kernel(const T* __restrict__ dy, int lddy,
                                                                const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int act, int N, int H, int W, int C, int Ho, int Wo,
                                                                T* __restrict__ g, const float* __restrict__ mean,
                                                                float* __restrict__ part, const float* __restrict__ coef) {
    // g == nullptr (with part): sums only -- the gradient is not materialised.  coef != nullptr: the second pass of that form --
    // the gather is repeated and the BatchNorm input gradient dx = A*g + B*(x - mean) + K (coef = [3][C], bn_bwd_coeffs_kernel)
    // is what gets stored (to g).  Saves a write and a read of the largest tensor of the network against one more read of the
    // quarter-size pooled gradient and its argmax bytes.
    // A thread owns the 2 x 2 block of input positions (2i..2i+1, 2j..2j+1) of one channel quad: the four pooling windows that reach it
    // -- (i..i+1) x (j..j+1) -- are loaded once (gradient + argmax bytes) and serve all four positions: 4 memory instructions per
    // position instead of 10 (every position used to look its up-to-four windows up by itself, each pooled value was fetched nine
    // times and the kernel ran at the load units' request rate, 3.3 TB/s).  Contributions are added in the old order (oh, then ow).
    extern __shared__ float sm[];
    const int Q = C >> 2;
    const int lq = quad_log2(Q);
    const int64_t total = (int64_t)N * Ho * Wo * Q;          // Ho == ceil(H/2), Wo == ceil(W/2): the launchers pass (H + 2 - 3) / 2 + 1, which is that
    float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    const float4 mu = (part || coef) ? *reinterpret_cast<const float4*>(mean + (threadIdx.x % Q) * 4) : make_float4(0, 0, 0, 0);
    float4 cA = make_float4(0, 0, 0, 0), cB = cA, cK = cA;
    if (coef) {
        const int c_ = (threadIdx.x % Q) * 4;
        cA = *reinterpret_cast<const float4*>(coef + c_);
        cB = *reinterpret_cast<const float4*>(coef + C + c_);
        cK = *reinterpret_cast<const float4*>(coef + 2 * C + c_);
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int c;
        split_quad(e, Q, lq, r, c);
        const float4 sa = *reinterpret_cast<const float4*>(scale + c);
        const float4 sb = *reinterpret_cast<const float4*>(shift + c);
        int j, i, n;
        if (r < (1ll << 31)) {
            const unsigned u = (unsigned)r, q1 = u / (unsigned)Wo;
            j = (int)(u - q1 * (unsigned)Wo);
            n = (int)(q1 / (unsigned)Ho);
            i = (int)(q1 - (unsigned)n * (unsigned)Ho);
        } else {
            j = (int)(r % Wo); r /= Wo;
            i = (int)(r % Ho);
            n = (int)(r / Ho);
        }
        // the four windows (i + a, j + b)
        float4 d[2][2];
        uchar4 id[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bool ok = i + a < Ho && j + b < Wo;
                const size_t o = ((size_t)n * Ho + (ok ? i + a : i)) * Wo + (ok ? j + b : j);
                id[a][b] = *reinterpret_cast<const uchar4*>(idx + o * C + c);
                d[a][b] = ld4(dy + o * lddy + c);
                if (!ok) { id[a][b] = make_uchar4(255, 255, 255, 255); }
            }
        float4 v[2][2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int h = 2 * i + dh, w = 2 * j + dw;
                const bool ok = h < H && w < W;
                v[dh][dw] = ld4(x + (((size_t)n * H + (ok ? h : 2 * i)) * W + (ok ? w : 2 * j)) * C + c);
            }
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int h = 2 * i + dh, w = 2 * j + dw;
                if (h >= H || w >= W) continue;
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                // windows that hold (h, w): oh = i (kh = 1 + dh) and, for the odd row, oh = i + 1 (kh = 0); same for the columns
#pragma unroll
                for (int a = 0; a <= dh; ++a)
#pragma unroll
                    for (int b = 0; b <= dw; ++b) {
                        const int pos = (a ? 0 : 1 + dh) * 3 + (b ? 0 : 1 + dw);
                        if (id[a][b].x == pos) s[0] += d[a][b].x;
                        if (id[a][b].y == pos) s[1] += d[a][b].y;
                        if (id[a][b].z == pos) s[2] += d[a][b].z;
                        if (id[a][b].w == pos) s[3] += d[a][b].w;
                    }
                const size_t ix = (((size_t)n * H + h) * W + w) * C + c;
                const float4 vv = v[dh][dw];
                float4 o4;
                o4.x = s[0] * act_grad_from_out(fmaf(sa.x, vv.x, sb.x), act);
                o4.y = s[1] * act_grad_from_out(fmaf(sa.y, vv.y, sb.y), act);
                o4.z = s[2] * act_grad_from_out(fmaf(sa.z, vv.z, sb.z), act);
                o4.w = s[3] * act_grad_from_out(fmaf(sa.w, vv.w, sb.w), act);
                if (coef) {
                    // g as the one-pass form would have stored it (bf16 storage: rounded), then bn_bwd_dx_kernel's expression
                    const float4 gr = make_float4(round_store<T>(o4.x), round_store<T>(o4.y), round_store<T>(o4.z), round_store<T>(o4.w));
                    float4 d4;
                    d4.x = fmaf(cA.x, gr.x, fmaf(cB.x, vv.x - mu.x, cK.x));
                    d4.y = fmaf(cA.y, gr.y, fmaf(cB.y, vv.y - mu.y, cK.y));
                    d4.z = fmaf(cA.z, gr.z, fmaf(cB.z, vv.z - mu.z, cK.z));
                    d4.w = fmaf(cA.w, gr.w, fmaf(cB.w, vv.w - mu.w, cK.w));
                    st4(g + ix, d4);
                } else if (g) {
                    st4(g + ix, o4);
                }
                if (part) {
                    acc[0].x += o4.x; acc[0].y += o4.y; acc[0].z += o4.z; acc[0].w += o4.w;
                    acc[1].x += o4.x * (vv.x - mu.x); acc[1].y += o4.y * (vv.y - mu.y);
                    acc[1].z += o4.z * (vv.z - mu.z); acc[1].w += o4.w * (vv.w - mu.w);
                }
            }
    }
    if (part) block_reduce_store<3>(acc, Q, 256 / Q, threadIdx.x % Q, threadIdx.x / Q, true, part + (size_t)blockIdx.x * 3 * C, C, sm);
}

static int ew_grid(int64_t elems) {
    int64_t g = cdiv64(elems, 256);
    const int64_t cap = (int64_t)num_cus() * 16;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace rd
using namespace rd;

extern "C" int rd_bn_finalize(const float* stat_partial, int32_t n_tiles, int32_t ld, int32_t c0, int32_t C, int64_t count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              int64_t* nbt, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    RD_CHECK_ARG(stat_partial && gamma && beta && mean && invstd && scale && shift && n_tiles > 0 && C > 0 && count > 0 &&
                     c0 >= 0 && c0 + C <= ld, "bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), stat_partial, n_tiles, ld, c0,
                       (double)count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean, invstd, scale, shift);
    RD_CHECK_LAUNCH("bn_finalize_kernel");
    return RD_OK;
}

extern "C" int rd_bn_eval_coeffs(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* scale, float* shift, void* stream) {
    RD_CHECK_ARG(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "bn_eval_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    RD_CHECK_LAUNCH("bn_eval_coeffs_kernel");
    return RD_OK;
}

extern "C" int rd_bn_eval_coeffs_batched(const void* jobs_dev, int32_t n_jobs, float eps, void* stream) {
    RD_CHECK_ARG(jobs_dev && n_jobs > 0, "bn_eval_coeffs_batched: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_batched_kernel, dim3(n_jobs), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const EvalCoefJob*>(jobs_dev), eps);
    RD_CHECK_LAUNCH("bn_eval_coeffs_batched_kernel");
    return RD_OK;
}

extern "C" int rd_bn_stats_tiles(int64_t M) { return (int)cdiv64(M, rows_per_block(M)); }
extern "C" int rd_bn_bwd_tiles(int64_t M, int32_t C) { return (C >= 4 && C % 4 == 0 && M > 0) ? (int)cdiv64(M, bwd_rows_per_block(M, C)) : RD_EINVAL; }

template <typename T>
static int rd_bn_stats_T(const T* x, int64_t M, int32_t C, int32_t ldx, float* stat_partial, int32_t* n_tiles, void* stream) {
    RD_CHECK_ARG(x && stat_partial && M > 0 && C >= 4 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0, "bn_stats: bad arguments");
    const int RPB = rows_per_block(M), grid = (int)cdiv64(M, RPB);
    const int Q = C / 4, RL = 256 / Q;
    hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(grid), dim3(256), (size_t)RL * 2 * C * sizeof(float),
                       static_cast<hipStream_t>(stream), x, M, C, ldx, RPB, stat_partial);
    RD_CHECK_LAUNCH("bn_stats_kernel");
    if (n_tiles) *n_tiles = grid;
    return RD_OK;
}
extern "C" int rd_bn_stats(const float* x, int64_t M, int32_t C, int32_t ldx, float* stat_partial, int32_t* n_tiles, void* stream) {
    return rd_bn_stats_T<float>(x, M, C, ldx, stat_partial, n_tiles, stream);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_stats_t(int32_t dtype, const void* x, int64_t M, int32_t C, int32_t ldx, float* stat_partial, int32_t* n_tiles, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_stats_T<float>(static_cast<const float*>(x), M, C, ldx, stat_partial, n_tiles, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_stats_T<bf16s>(static_cast<const bf16s*>(x), M, C, ldx, stat_partial, n_tiles, stream);
    rd::set_error("rd_bn_stats_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

template <typename T>
static int rd_bn_act_T(const T* x1, int32_t ldx1, const float* scale1, const float* shift1, const T* x2, int32_t ldx2, const float* scale2, const float* shift2, T* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* stream,
                       void* pieces = nullptr, int64_t piece_elems = 0) {
    RD_CHECK_ARG(x1 && scale1 && shift1 && y && M > 0 && C % 4 == 0 && ldx1 % 4 == 0 && ldy % 4 == 0 && (!x2 || ldx2 % 4 == 0),
                 "bn_act: bad arguments");
    RD_CHECK_ARG(!pieces || (C % 16 == 0 && piece_elems >= (int64_t)C * M && reinterpret_cast<uintptr_t>(pieces) % 16 == 0), "bn_act: bad piece planes");
    hipLaunchKernelGGL((bn_act_kernel<T>), dim3(ew_grid(M * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream), x1, ldx1,
                       scale1, shift1, x2, ldx2, scale2, shift2, y, ldy, M, C, act, static_cast<unsigned short*>(pieces), piece_elems);
    RD_CHECK_LAUNCH("bn_act_kernel");
    return RD_OK;
}
extern "C" int rd_bn_act(const float* x1, int32_t ldx1, const float* scale1, const float* shift1, const float* x2, int32_t ldx2, const float* scale2, const float* shift2, float* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* stream) {
    return rd_bn_act_T<float>(x1, ldx1, scale1, shift1, x2, ldx2, scale2, shift2, y, ldy, M, C, act, stream);
}
// + piece planes of y for the pre-split convolutions (include/radar_depth_hip.h, rd_split_pieces): pieces may be NULL
extern "C" int rd_bn_act_p(const float* x1, int32_t ldx1, const float* scale1, const float* shift1, const float* x2, int32_t ldx2, const float* scale2, const float* shift2, float* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* pieces, int64_t piece_elems, void* stream) {
    return rd_bn_act_T<float>(x1, ldx1, scale1, shift1, x2, ldx2, scale2, shift2, y, ldy, M, C, act, stream, pieces, piece_elems);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_act_t(int32_t dtype, const void* x1, int32_t ldx1, const float* scale1, const float* shift1, const void* x2, int32_t ldx2, const float* scale2, const float* shift2, void* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_act_T<float>(static_cast<const float*>(x1), ldx1, scale1, shift1, static_cast<const float*>(x2), ldx2, scale2, shift2, static_cast<float*>(y), ldy, M, C, act, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_act_T<bf16s>(static_cast<const bf16s*>(x1), ldx1, scale1, shift1, static_cast<const bf16s*>(x2), ldx2, scale2, shift2, static_cast<bf16s*>(y), ldy, M, C, act, stream);
    rd::set_error("rd_bn_act_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

template <typename T>
static int bn_bwd_reduce_impl(const T* dy, int32_t lddy, const T* y, int32_t ldy, const T* x1, int32_t ldx1,
                              const float* mean1, const T* x2, int32_t ldx2, const float* mean2, T* g, int32_t ldg,
                              int64_t M, int32_t C, int32_t act, float* red_partial, const float* scale1, const float* shift1,
                              void* stream, const float* scale2 = nullptr, const float* shift2 = nullptr) {
    RD_CHECK_ARG(dy && red_partial && M > 0 && C >= 4 && C % 4 == 0 && C <= 1024, "bn_bwd_reduce: bad arguments");
    RD_CHECK_ARG(act == RD_ACT_NONE || y || (scale1 && shift1 && x1 && (!x2 || (scale2 && shift2))),
                 "bn_bwd_reduce: activation needs y (or the scale/shift of every operand)");
    RD_CHECK_ARG((!x1 || mean1) && (!x2 || mean2), "bn_bwd_reduce: x without mean");
    const int RPB = bwd_rows_per_block(M, C), grid = (int)cdiv64(M, RPB);
    const int Q = C / 4, RL = 256 / Q;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(256), (size_t)RL * 3 * C * sizeof(float),
                       static_cast<hipStream_t>(stream), dy, lddy, y, ldy, x1, ldx1, mean1, x2, ldx2, mean2, g, ldg, M, C, act, RPB,
                       red_partial, scale1, shift1, scale2, shift2);
    RD_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    return RD_OK;
}
template <typename T>
static int rd_bn_bwd_reduce_T(const T* dy, int32_t lddy, const T* y, int32_t ldy, const T* x1, int32_t ldx1, const float* mean1, const T* x2, int32_t ldx2, const float* mean2, T* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    return bn_bwd_reduce_impl<T>(dy, lddy, y, ldy, x1, ldx1, mean1, x2, ldx2, mean2, g, ldg, M, C, act, red_partial, nullptr, nullptr, stream);
}
extern "C" int rd_bn_bwd_reduce(const float* dy, int32_t lddy, const float* y, int32_t ldy, const float* x1, int32_t ldx1, const float* mean1, const float* x2, int32_t ldx2, const float* mean2, float* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    return rd_bn_bwd_reduce_T<float>(dy, lddy, y, ldy, x1, ldx1, mean1, x2, ldx2, mean2, g, ldg, M, C, act, red_partial, stream);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_bwd_reduce_t(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x1, int32_t ldx1, const float* mean1, const void* x2, int32_t ldx2, const float* mean2, void* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_bwd_reduce_T<float>(static_cast<const float*>(dy), lddy, static_cast<const float*>(y), ldy, static_cast<const float*>(x1), ldx1, mean1, static_cast<const float*>(x2), ldx2, mean2, static_cast<float*>(g), ldg, M, C, act, red_partial, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_bwd_reduce_T<bf16s>(static_cast<const bf16s*>(dy), lddy, static_cast<const bf16s*>(y), ldy, static_cast<const bf16s*>(x1), ldx1, mean1, static_cast<const bf16s*>(x2), ldx2, mean2, static_cast<bf16s*>(g), ldg, M, C, act, red_partial, stream);
    rd::set_error("rd_bn_bwd_reduce_t: bad dtype %d", dtype);
    return RD_EINVAL;
}
// out = act(scale1 * x1 + shift1): the activation's sign is recomputed from x1 (one tensor read less than rd_bn_bwd_reduce)
template <typename T>
static int rd_bn_bwd_reduce_x_T(const T* dy, int32_t lddy, const T* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, T* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    return bn_bwd_reduce_impl<T>(dy, lddy, nullptr, 0, x1, ldx1, mean1, nullptr, 0, nullptr, g, ldg, M, C, act, red_partial, scale1, shift1, stream);
}
extern "C" int rd_bn_bwd_reduce_x(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, float* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    return rd_bn_bwd_reduce_x_T<float>(dy, lddy, x1, ldx1, mean1, scale1, shift1, g, ldg, M, C, act, red_partial, stream);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_bwd_reduce_x_t(int32_t dtype, const void* dy, int32_t lddy, const void* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, void* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_bwd_reduce_x_T<float>(static_cast<const float*>(dy), lddy, static_cast<const float*>(x1), ldx1, mean1, scale1, shift1, static_cast<float*>(g), ldg, M, C, act, red_partial, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_bwd_reduce_x_T<bf16s>(static_cast<const bf16s*>(dy), lddy, static_cast<const bf16s*>(x1), ldx1, mean1, scale1, shift1, static_cast<bf16s*>(g), ldg, M, C, act, red_partial, stream);
    rd::set_error("rd_bn_bwd_reduce_x_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

namespace rd {
// (for rd_stem_wgrad_split_bn_t, csrc/stem_wgrad_split.hip: the coefficient kernel without the apply pass)
int launch_bn_bwd_coeffs(const float* red_partial, int n_tiles, int C, int which, double count, const float* gamma, const float* invstd,
                         float* dgamma, float* dbeta, float* coef_ws, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, s, red_partial, n_tiles, C, which, count, gamma, invstd, dgamma, dbeta,
                       coef_ws);
    RD_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
    return RD_OK;
}
}  // namespace rd

template <typename T>
static int rd_bn_bwd_apply_T(const T* g, int32_t ldg, const T* x, int32_t ldx, const float* red_partial, int32_t n_tiles, int32_t which, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, T* dx, int32_t lddx, int64_t M, int32_t C, void* stream,
                             void* pieces = nullptr, int64_t piece_elems = 0) {
    RD_CHECK_ARG(g && x && red_partial && gamma && mean && invstd && coef_ws && dx && (which == 1 || which == 2) && M > 0 &&
                     C % 4 == 0, "bn_bwd_apply: bad arguments");
    RD_CHECK_ARG(!pieces || (C % 16 == 0 && piece_elems >= (int64_t)C * M && reinterpret_cast<uintptr_t>(pieces) % 16 == 0), "bn_bwd_apply: bad piece planes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, s, red_partial, n_tiles, C, which, (double)M, gamma, invstd,
                       dgamma, dbeta, coef_ws);
    RD_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
    hipLaunchKernelGGL((bn_bwd_dx_kernel<T>), dim3(ew_grid(M * (C / 4))), dim3(256), 0, s, g, ldg, x, ldx, mean, coef_ws, dx, lddx, M, C,
                       (const float*)nullptr, (const float*)nullptr, RD_ACT_NONE, static_cast<unsigned short*>(pieces), piece_elems);
    RD_CHECK_LAUNCH("bn_bwd_dx_kernel");
    return RD_OK;
}
extern "C" int rd_bn_bwd_apply(const float* g, int32_t ldg, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles, int32_t which, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C, void* stream) {
    return rd_bn_bwd_apply_T<float>(g, ldg, x, ldx, red_partial, n_tiles, which, gamma, mean, invstd, dgamma, dbeta, coef_ws, dx, lddx, M, C, stream);
}
extern "C" int rd_bn_bwd_apply_p(const float* g, int32_t ldg, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles, int32_t which, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C, void* pieces, int64_t piece_elems, void* stream) {
    return rd_bn_bwd_apply_T<float>(g, ldg, x, ldx, red_partial, n_tiles, which, gamma, mean, invstd, dgamma, dbeta, coef_ws, dx, lddx, M, C, stream, pieces, piece_elems);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_bwd_apply_t(int32_t dtype, const void* g, int32_t ldg, const void* x, int32_t ldx, const float* red_partial, int32_t n_tiles, int32_t which, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, void* dx, int32_t lddx, int64_t M, int32_t C, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_bwd_apply_T<float>(static_cast<const float*>(g), ldg, static_cast<const float*>(x), ldx, red_partial, n_tiles, which, gamma, mean, invstd, dgamma, dbeta, coef_ws, static_cast<float*>(dx), lddx, M, C, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_bwd_apply_T<bf16s>(static_cast<const bf16s*>(g), ldg, static_cast<const bf16s*>(x), ldx, red_partial, n_tiles, which, gamma, mean, invstd, dgamma, dbeta, coef_ws, static_cast<bf16s*>(dx), lddx, M, C, stream);
    rd::set_error("rd_bn_bwd_apply_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

// out = act(bn1(x1) + bn2(x2)) (downsample blocks, UpProj joins): sums and both input gradients straight from the raw output
// gradient -- the activation output is not read, the masked gradient is not written, dy is read once per pass.
template <typename T>
static int rd_bn_bwd_reduce_x2_T(const T* dy, int32_t lddy, const T* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, const T* x2, int32_t ldx2, const float* mean2, const float* scale2, const float* shift2, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    RD_CHECK_ARG(x1 && x2 && scale1 && shift1 && scale2 && shift2 && act != RD_ACT_NONE, "bn_bwd_reduce_x2: bad arguments");
    return bn_bwd_reduce_impl<T>(dy, lddy, nullptr, 0, x1, ldx1, mean1, x2, ldx2, mean2, nullptr, 0, M, C, act, red_partial, scale1, shift1, stream,
                              scale2, shift2);
}
extern "C" int rd_bn_bwd_reduce_x2(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, const float* x2, int32_t ldx2, const float* mean2, const float* scale2, const float* shift2, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    return rd_bn_bwd_reduce_x2_T<float>(dy, lddy, x1, ldx1, mean1, scale1, shift1, x2, ldx2, mean2, scale2, shift2, M, C, act, red_partial, stream);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_bwd_reduce_x2_t(int32_t dtype, const void* dy, int32_t lddy, const void* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, const void* x2, int32_t ldx2, const float* mean2, const float* scale2, const float* shift2, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_bwd_reduce_x2_T<float>(static_cast<const float*>(dy), lddy, static_cast<const float*>(x1), ldx1, mean1, scale1, shift1, static_cast<const float*>(x2), ldx2, mean2, scale2, shift2, M, C, act, red_partial, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_bwd_reduce_x2_T<bf16s>(static_cast<const bf16s*>(dy), lddy, static_cast<const bf16s*>(x1), ldx1, mean1, scale1, shift1, static_cast<const bf16s*>(x2), ldx2, mean2, scale2, shift2, M, C, act, red_partial, stream);
    rd::set_error("rd_bn_bwd_reduce_x2_t: bad dtype %d", dtype);
    return RD_EINVAL;
}
template <typename T>
static int rd_bn_bwd_apply_x2_T(const T* dy, int32_t lddy, const T* x1, int32_t ldx1, const T* x2, int32_t ldx2, const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1, const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2, const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* coef_ws6, T* dx1, int32_t lddx1, T* dx2, int32_t lddx2, int64_t M, int32_t C, void* stream,
                                void* pieces1 = nullptr, int64_t piece_elems1 = 0, void* pieces2 = nullptr, int64_t piece_elems2 = 0) {
    RD_CHECK_ARG(dy && x1 && x2 && red_partial && gamma1 && gamma2 && mean1 && mean2 && invstd1 && invstd2 && scale1 && shift1 && scale2 &&
                     shift2 && coef_ws6 && dx1 && dx2 && M > 0 && C % 4 == 0, "bn_bwd_apply_x2: bad arguments");
    RD_CHECK_ARG((!pieces1 || (C % 16 == 0 && piece_elems1 >= (int64_t)C * M)) && (!pieces2 || (C % 16 == 0 && piece_elems2 >= (int64_t)C * M)), "bn_bwd_apply_x2: bad piece planes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, s, red_partial, n_tiles, C, 1, (double)M, gamma1, invstd1, dgamma1, dbeta1,
                       coef_ws6);
    RD_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, s, red_partial, n_tiles, C, 2, (double)M, gamma2, invstd2, dgamma2, dbeta2,
                       coef_ws6 + 3 * C);
    RD_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
    hipLaunchKernelGGL((bn_bwd_dx2_kernel<T>), dim3(ew_grid(M * (C / 4))), dim3(256), 0, s, dy, lddy, x1, ldx1, x2, ldx2, mean1, mean2, coef_ws6,
                       coef_ws6 + 3 * C, scale1, shift1, scale2, shift2, act, dx1, lddx1, dx2, lddx2, M, C,
                       static_cast<unsigned short*>(pieces1), piece_elems1, static_cast<unsigned short*>(pieces2), piece_elems2);
    RD_CHECK_LAUNCH("bn_bwd_dx2_kernel");
    return RD_OK;
}
extern "C" int rd_bn_bwd_apply_x2(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* x2, int32_t ldx2, const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1, const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2, const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* coef_ws6, float* dx1, int32_t lddx1, float* dx2, int32_t lddx2, int64_t M, int32_t C, void* stream) {
    return rd_bn_bwd_apply_x2_T<float>(dy, lddy, x1, ldx1, x2, ldx2, red_partial, n_tiles, gamma1, mean1, invstd1, scale1, shift1, gamma2, mean2, invstd2, scale2, shift2, act, dgamma1, dbeta1, dgamma2, dbeta2, coef_ws6, dx1, lddx1, dx2, lddx2, M, C, stream);
}
extern "C" int rd_bn_bwd_apply_x2_p(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* x2, int32_t ldx2, const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1, const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2, const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* coef_ws6, float* dx1, int32_t lddx1, float* dx2, int32_t lddx2, int64_t M, int32_t C, void* pieces1, int64_t piece_elems1, void* pieces2, int64_t piece_elems2, void* stream) {
    return rd_bn_bwd_apply_x2_T<float>(dy, lddy, x1, ldx1, x2, ldx2, red_partial, n_tiles, gamma1, mean1, invstd1, scale1, shift1, gamma2, mean2, invstd2, scale2, shift2, act, dgamma1, dbeta1, dgamma2, dbeta2, coef_ws6, dx1, lddx1, dx2, lddx2, M, C, stream, pieces1, piece_elems1, pieces2, piece_elems2);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_bwd_apply_x2_t(int32_t dtype, const void* dy, int32_t lddy, const void* x1, int32_t ldx1, const void* x2, int32_t ldx2, const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1, const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2, const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* coef_ws6, void* dx1, int32_t lddx1, void* dx2, int32_t lddx2, int64_t M, int32_t C, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_bwd_apply_x2_T<float>(static_cast<const float*>(dy), lddy, static_cast<const float*>(x1), ldx1, static_cast<const float*>(x2), ldx2, red_partial, n_tiles, gamma1, mean1, invstd1, scale1, shift1, gamma2, mean2, invstd2, scale2, shift2, act, dgamma1, dbeta1, dgamma2, dbeta2, coef_ws6, static_cast<float*>(dx1), lddx1, static_cast<float*>(dx2), lddx2, M, C, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_bwd_apply_x2_T<bf16s>(static_cast<const bf16s*>(dy), lddy, static_cast<const bf16s*>(x1), ldx1, static_cast<const bf16s*>(x2), ldx2, red_partial, n_tiles, gamma1, mean1, invstd1, scale1, shift1, gamma2, mean2, invstd2, scale2, shift2, act, dgamma1, dbeta1, dgamma2, dbeta2, coef_ws6, static_cast<bf16s*>(dx1), lddx1, static_cast<bf16s*>(dx2), lddx2, M, C, stream);
    rd::set_error("rd_bn_bwd_apply_x2_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

// Apply pass paired with rd_bn_bwd_reduce_x(g = NULL): dy is the raw output gradient, the activation factor is recomputed from x.
template <typename T>
static int rd_bn_bwd_apply_x_T(const T* dy, int32_t lddy, const T* x, int32_t ldx, const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift, int32_t act, float* dgamma, float* dbeta, float* coef_ws, T* dx, int32_t lddx, int64_t M, int32_t C, void* stream,
                               void* pieces = nullptr, int64_t piece_elems = 0) {
    RD_CHECK_ARG(dy && x && red_partial && gamma && mean && invstd && scale && shift && coef_ws && dx && M > 0 && C % 4 == 0,
                 "bn_bwd_apply_x: bad arguments");
    RD_CHECK_ARG(!pieces || (C % 16 == 0 && piece_elems >= (int64_t)C * M && reinterpret_cast<uintptr_t>(pieces) % 16 == 0), "bn_bwd_apply_x: bad piece planes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, s, red_partial, n_tiles, C, 1, (double)M, gamma, invstd, dgamma,
                       dbeta, coef_ws);
    RD_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
    hipLaunchKernelGGL((bn_bwd_dx_kernel<T>), dim3(ew_grid(M * (C / 4))), dim3(256), 0, s, dy, lddy, x, ldx, mean, coef_ws, dx, lddx, M, C,
                       scale, shift, act, static_cast<unsigned short*>(pieces), piece_elems);
    RD_CHECK_LAUNCH("bn_bwd_dx_kernel");
    return RD_OK;
}
extern "C" int rd_bn_bwd_apply_x(const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift, int32_t act, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C, void* stream) {
    return rd_bn_bwd_apply_x_T<float>(dy, lddy, x, ldx, red_partial, n_tiles, gamma, mean, invstd, scale, shift, act, dgamma, dbeta, coef_ws, dx, lddx, M, C, stream);
}
extern "C" int rd_bn_bwd_apply_x_p(const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift, int32_t act, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C, void* pieces, int64_t piece_elems, void* stream) {
    return rd_bn_bwd_apply_x_T<float>(dy, lddy, x, ldx, red_partial, n_tiles, gamma, mean, invstd, scale, shift, act, dgamma, dbeta, coef_ws, dx, lddx, M, C, stream, pieces, piece_elems);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bn_bwd_apply_x_t(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift, int32_t act, float* dgamma, float* dbeta, float* coef_ws, void* dx, int32_t lddx, int64_t M, int32_t C, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bn_bwd_apply_x_T<float>(static_cast<const float*>(dy), lddy, static_cast<const float*>(x), ldx, red_partial, n_tiles, gamma, mean, invstd, scale, shift, act, dgamma, dbeta, coef_ws, static_cast<float*>(dx), lddx, M, C, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bn_bwd_apply_x_T<bf16s>(static_cast<const bf16s*>(dy), lddy, static_cast<const bf16s*>(x), ldx, red_partial, n_tiles, gamma, mean, invstd, scale, shift, act, dgamma, dbeta, coef_ws, static_cast<bf16s*>(dx), lddx, M, C, stream);
    rd::set_error("rd_bn_bwd_apply_x_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

template <typename T>
static int rd_bnact_maxpool_fwd_T(const T* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, T* y, int32_t ldy, uint8_t* idx, void* stream,
                                  void* pieces = nullptr, int64_t piece_elems = 0) {
    RD_CHECK_ARG(x && scale && shift && y && idx && C % 4 == 0 && ldy % 4 == 0, "bnact_maxpool_fwd: bad arguments");
    RD_CHECK_ARG(!pieces || (C % 16 == 0 && reinterpret_cast<uintptr_t>(pieces) % 16 == 0), "bnact_maxpool_fwd: bad piece planes");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL((bnact_maxpool_fwd_kernel<T>), dim3(ew_grid((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, scale, shift, act, N, H, W, C, Ho, Wo, y, ldy, idx, static_cast<unsigned short*>(pieces), piece_elems);
    RD_CHECK_LAUNCH("bnact_maxpool_fwd_kernel");
    return RD_OK;
}
extern "C" int rd_bnact_maxpool_fwd(const float* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, float* y, int32_t ldy, uint8_t* idx, void* stream) {
    return rd_bnact_maxpool_fwd_T<float>(x, scale, shift, act, N, H, W, C, y, ldy, idx, stream);
}
extern "C" int rd_bnact_maxpool_fwd_p(const float* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, float* y, int32_t ldy, uint8_t* idx, void* pieces, int64_t piece_elems, void* stream) {
    return rd_bnact_maxpool_fwd_T<float>(x, scale, shift, act, N, H, W, C, y, ldy, idx, stream, pieces, piece_elems);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bnact_maxpool_fwd_t(int32_t dtype, const void* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, void* y, int32_t ldy, uint8_t* idx, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bnact_maxpool_fwd_T<float>(static_cast<const float*>(x), scale, shift, act, N, H, W, C, static_cast<float*>(y), ldy, idx, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bnact_maxpool_fwd_T<bf16s>(static_cast<const bf16s*>(x), scale, shift, act, N, H, W, C, static_cast<bf16s*>(y), ldy, idx, stream);
    rd::set_error("rd_bnact_maxpool_fwd_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

template <typename T>
static int rd_bnact_maxpool_bwd_T(const T* dy, int32_t lddy, const uint8_t* idx, const T* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, T* g, void* stream) {
    RD_CHECK_ARG(dy && idx && x && scale && shift && g && C % 4 == 0 && lddy % 4 == 0, "bnact_maxpool_bwd: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL((bnact_maxpool_bwd_kernel<T>), dim3(ew_grid((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dy, lddy, idx, x, scale, shift, act, N, H, W, C, Ho, Wo, g,
                       (const float*)nullptr, (float*)nullptr, (const float*)nullptr);
    RD_CHECK_LAUNCH("bnact_maxpool_bwd_kernel");
    return RD_OK;
}
extern "C" int rd_bnact_maxpool_bwd(const float* dy, int32_t lddy, const uint8_t* idx, const float* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, float* g, void* stream) {
    return rd_bnact_maxpool_bwd_T<float>(dy, lddy, idx, x, scale, shift, act, N, H, W, C, g, stream);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bnact_maxpool_bwd_t(int32_t dtype, const void* dy, int32_t lddy, const uint8_t* idx, const void* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, void* g, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bnact_maxpool_bwd_T<float>(static_cast<const float*>(dy), lddy, idx, static_cast<const float*>(x), scale, shift, act, N, H, W, C, static_cast<float*>(g), stream);
    if (dtype == RD_DTYPE_BF16) return rd_bnact_maxpool_bwd_T<bf16s>(static_cast<const bf16s*>(dy), lddy, idx, static_cast<const bf16s*>(x), scale, shift, act, N, H, W, C, static_cast<bf16s*>(g), stream);
    rd::set_error("rd_bnact_maxpool_bwd_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

// Same, and the stem BatchNorm's backward sums in the same pass: red_partial [rd_bnact_maxpool_bwd_tiles(...)][3][C]
// (slot 0 = sum g, slot 1 = sum g*(x - mean)), consumed by rd_bn_bwd_apply(which = 1).
// (the kernel's work item is one 2 x 2 block of positions x one channel quad: N * ceil(H/2) * ceil(W/2) * C/4 of them -- grid and tile
//  count are sized from THAT everywhere, ADVICE r5: sized from N*H*W*C/4, small shapes launched up to 4x the blocks, all writing zero rows)
extern "C" int rd_bnact_maxpool_bwd_tiles(int32_t N, int32_t H, int32_t W, int32_t C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    return ew_grid((int64_t)N * Ho * Wo * (C / 4));
}
template <typename T>
static int rd_bnact_maxpool_bwd_stats_T(const T* dy, int32_t lddy, const uint8_t* idx, const T* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, T* g, const float* mean, float* red_partial, void* stream) {
    RD_CHECK_ARG(dy && idx && x && scale && shift && mean && red_partial && C % 4 == 0 && lddy % 4 == 0,
                 "bnact_maxpool_bwd_stats: bad arguments");      // (g may be NULL: sums only, see rd_bnact_maxpool_bwd_apply_t)
    RD_CHECK_ARG(C >= 4 && 256 % (C / 4) == 0, "bnact_maxpool_bwd_stats: C/4 = %d must divide 256", C / 4);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int Q = C / 4, RL = 256 / Q;
    hipLaunchKernelGGL((bnact_maxpool_bwd_kernel<T>), dim3(ew_grid((int64_t)N * Ho * Wo * (C / 4))), dim3(256),
                       (size_t)RL * 3 * C * sizeof(float), static_cast<hipStream_t>(stream), dy, lddy, idx, x, scale, shift, act, N, H,
                       W, C, Ho, Wo, g, mean, red_partial, (const float*)nullptr);
    RD_CHECK_LAUNCH("bnact_maxpool_bwd_kernel");
    return RD_OK;
}
extern "C" int rd_bnact_maxpool_bwd_stats(const float* dy, int32_t lddy, const uint8_t* idx, const float* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, float* g, const float* mean, float* red_partial, void* stream) {
    return rd_bnact_maxpool_bwd_stats_T<float>(dy, lddy, idx, x, scale, shift, act, N, H, W, C, g, mean, red_partial, stream);
}
// storage-typed form: dtype = RD_DTYPE_F32 / RD_DTYPE_BF16 selects the element type of the NHWC tensors (strides in elements)
extern "C" int rd_bnact_maxpool_bwd_stats_t(int32_t dtype, const void* dy, int32_t lddy, const uint8_t* idx, const void* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, void* g, const float* mean, float* red_partial, void* stream) {
    if (dtype == RD_DTYPE_F32) return rd_bnact_maxpool_bwd_stats_T<float>(static_cast<const float*>(dy), lddy, idx, static_cast<const float*>(x), scale, shift, act, N, H, W, C, static_cast<float*>(g), mean, red_partial, stream);
    if (dtype == RD_DTYPE_BF16) return rd_bnact_maxpool_bwd_stats_T<bf16s>(static_cast<const bf16s*>(dy), lddy, idx, static_cast<const bf16s*>(x), scale, shift, act, N, H, W, C, static_cast<bf16s*>(g), mean, red_partial, stream);
    rd::set_error("rd_bnact_maxpool_bwd_stats_t: bad dtype %d", dtype);
    return RD_EINVAL;
}

// Second pass of the stem's BatchNorm + pool backward when the first pass (rd_bnact_maxpool_bwd_stats_t with g == NULL) only took
// the sums: finishes them into dgamma / dbeta / the dx coefficients (bn_bwd_coeffs_kernel) and repeats the pool gather, storing
// dx = A*g + B*(x - mean) + K directly.  x / dx are the full-resolution [N,H,W,C] stem tensors, dy / idx the pooled ones.
template <typename T>
static int rd_bnact_maxpool_bwd_apply_T(const T* dy, int32_t lddy, const uint8_t* idx, const T* x, const float* scale, const float* shift,
                                        int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, const float* red_partial, int32_t n_tiles,
                                        const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                        float* coef_ws, T* dx, void* stream) {
    RD_CHECK_ARG(dy && idx && x && scale && shift && red_partial && gamma && mean && invstd && coef_ws && dx && C % 4 == 0 && lddy % 4 == 0 &&
                     n_tiles > 0, "bnact_maxpool_bwd_apply: bad arguments");
    RD_CHECK_ARG(C >= 4 && 256 % (C / 4) == 0, "bnact_maxpool_bwd_apply: C/4 = %d must divide 256", C / 4);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, s, red_partial, n_tiles, C, 1, (double)((int64_t)N * H * W), gamma, invstd,
                       dgamma, dbeta, coef_ws);
    RD_CHECK_LAUNCH("bn_bwd_coeffs_kernel");
    hipLaunchKernelGGL((bnact_maxpool_bwd_kernel<T>), dim3(ew_grid((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, s, dy, lddy, idx, x, scale,
                       shift, act, N, H, W, C, Ho, Wo, dx, mean, (float*)nullptr, (const float*)coef_ws);
    RD_CHECK_LAUNCH("bnact_maxpool_bwd_kernel");
    return RD_OK;
}
extern "C" int rd_bnact_maxpool_bwd_apply_t(int32_t dtype, const void* dy, int32_t lddy, const uint8_t* idx, const void* x, const float* scale,
                                            const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C,
                                            const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean,
                                            const float* invstd, float* dgamma, float* dbeta, float* coef_ws, void* dx, void* stream) {
    if (dtype == RD_DTYPE_F32)
        return rd_bnact_maxpool_bwd_apply_T<float>(static_cast<const float*>(dy), lddy, idx, static_cast<const float*>(x), scale, shift, act, N, H, W, C,
                                                   red_partial, n_tiles, gamma, mean, invstd, dgamma, dbeta, coef_ws, static_cast<float*>(dx), stream);
    if (dtype == RD_DTYPE_BF16)
        return rd_bnact_maxpool_bwd_apply_T<bf16s>(static_cast<const bf16s*>(dy), lddy, idx, static_cast<const bf16s*>(x), scale, shift, act, N, H, W, C,
                                                   red_partial, n_tiles, gamma, mean, invstd, dgamma, dbeta, coef_ws, static_cast<bf16s*>(dx), stream);
    rd::set_error("rd_bnact_maxpool_bwd_apply_t: bad dtype %d", dtype);
    return RD_EINVAL;
}
