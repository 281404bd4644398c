// 3x3 / stride-1 convolutions with 16 input and 16 output channels (the depth encoder's layer1, the last decoder stage's
// conv2: model/models.py:30-72 BasicBlock with 16 planes, model/models.py:96-112 UpProjModule conv2 at out_channels 16) on
// v_mfma_f32_16x16x4_f32.
//
// gconv.hip's kernels are built on the 32x32x2 fp32 MFMA: with 16 output channels half of every B operand is padding, and these
// layers ran at 22-28 % of the fp32 peak (dec4 conv2: 164 us for 7.1 GFLOP).  The 16x16x4 instruction has the same rate per MAC
// and exactly this layer's N.  What makes the kernel small:
//   * K = 9 taps x 16 channels = 36 MFMA steps, and the WHOLE weight operand is 36 registers per lane: lane (n = lane % 16,
//     kq = lane / 16) holds w[tap][4 kq .. 4 kq + 3][n] for the nine taps -- exactly the 16-byte quads of the packed layout
//     ([tap][Cin/4][Cout] x 4 input channels, rd_pack_weights), loaded once per workgroup; no weight traffic through LDS;
//   * the k index is ordered (tap, j, kq) with input channel 4 kq + j, so one ds_read_b128 of a patch pixel feeds the four MFMAs
//     of a tap; the patch pitch of 24 floats makes those reads conflict-free for the b128 lane groups (MI355X_MICROARCH LDS table);
//   * tile = 16 x 16 output pixels, a wave owns four rows (four 16-pixel M blocks, 16 accumulator registers).
// Same contract as rd_gconv: single-phase descriptor (forward or the stride-1 dgrad: taps and weight indices come from the
// descriptor), optional residual addend, optional BatchNorm partial sums [tile][2][16] (tile index = blockIdx, n-major).
#include <hip/hip_runtime.h>

#include "common.h"

namespace rd {

struct Conv16Args {
    const float* in;
    const float* w;       // packed quads [9][4][16][4]
    float* out;
    const float* addend;
    float* stat;
    int ldi, ldo, ld_add;
    int N, H, W;          // output grid == input grid (unit strides)
    int tiles_h, tiles_w;
    int ih_off, iw_off;   // dh_min, dw_min
    int widx_pos[9];      // weight slab of the tap at patch position (pos / 3, pos % 3)
};

constexpr int C16_T = 16, C16_P = 18, C16_PS = 24;

template <bool STAT, bool ADD>
__global__ __launch_bounds__(256) void conv16_kernel(const Conv16Args a) {
    __shared__ __attribute__((aligned(16))) float s_patch[C16_P * C16_P * C16_PS];
    __shared__ float s_red[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int bid = blockIdx.x;
    const int per_img = a.tiles_h * a.tiles_w;
    const int n = bid / per_img, trem = bid - n * per_img;
    const int r0 = (trem / a.tiles_w) * C16_T, c0 = (trem % a.tiles_w) * C16_T;
    const int ih0 = r0 + a.ih_off, iw0 = c0 + a.iw_off;

    // the weight operand: nine quads per lane, straight from the packed layout, ordered by PATCH POSITION (pos / 3, pos % 3) so
    // that every LDS address of the walk is one lane-constant base plus a compile-time offset (an immediate of the ds_read)
    float4 wq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[t] = *reinterpret_cast<const float4*>(a.w + ((size_t)(a.widx_pos[t] * 4 + kq) * 16 + m) * 4);

    // halo patch [18][18][16] -> LDS (zero outside the image), 16-byte units, all loads of a thread in flight before its first write
    const float* in_n = a.in + (size_t)n * a.H * a.W * a.ldi;
    {
        constexpr int U = 6;       // 18*18*4 = 1296 units <= 6 * 256
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + u * 256;
            const int px = e >> 2, q = e & 3;
            const int py = px / C16_P, pxx = px - py * C16_P;
            const int ih = ih0 + py, iw = iw0 + pxx;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < C16_P * C16_P * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                v[u] = *reinterpret_cast<const float4*>(in_n + ((size_t)ih * a.W + iw) * a.ldi + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + u * 256;
            if (e < C16_P * C16_P * 4) *reinterpret_cast<float4*>(s_patch + (e >> 2) * C16_PS + 4 * (e & 3)) = v[u];
        }
    }
    rd_sync();

    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* al = s_patch + ((wave * 4) * C16_P + m) * C16_PS + 4 * kq;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 av = *reinterpret_cast<const float4*>(al + ((mb + t / 3) * C16_P + t % 3) * C16_PS);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wq[t].x, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wq[t].y, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wq[t].z, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wq[t].w, acc[mb], 0, 0, 0);
        }
    }

    // epilogue: in the C/D layout of the 16x16 MFMA a lane holds channel m of the pixels 4 kq .. 4 kq + 3 of the block's row; the 4x4
    // block (4 registers x the 4 lanes of a quad = 4 pixels x 4 channels) is transposed in registers (common.h), after which lane q
    // of a quad holds the channels 4 (m / 4) .. + 3 of pixel 4 kq + q: one 16-byte store (and addend load) per M block and lane
    const int q = m & 3, cq = m & ~3;
    const bool odd1 = q & 1, odd2 = q & 2;
    float4 ssum4 = make_float4(0.f, 0.f, 0.f, 0.f), ssq4 = ssum4;
    const int c = c0 + 4 * kq + q;
    float4 addv[4];
    if (ADD) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int r = r0 + wave * 4 + mb;
            addv[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.H && c < a.W) addv[mb] = *reinterpret_cast<const float4*>(a.addend + (((size_t)n * a.H + r) * a.W + c) * a.ld_add + cq);
        }
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int r = r0 + wave * 4 + mb;
        float e0 = acc[mb][0], e1 = acc[mb][1], e2 = acc[mb][2], e3 = acc[mb][3];
        quad_transpose(e0, e1, e2, e3, odd1, odd2);
        float4 val = make_float4(e0, e1, e2, e3);
        if (ADD) { val.x += addv[mb].x; val.y += addv[mb].y; val.z += addv[mb].z; val.w += addv[mb].w; }
        if (r < a.H && c < a.W) {
            *reinterpret_cast<float4*>(a.out + (((size_t)n * a.H + r) * a.W + c) * a.ldo + cq) = val;
            ssum4.x += val.x; ssum4.y += val.y; ssum4.z += val.z; ssum4.w += val.w;
            ssq4.x += val.x * val.x; ssq4.y += val.y * val.y; ssq4.z += val.z * val.z; ssq4.w += val.w * val.w;
        }
    }
    if (STAT) {
        // per channel: sum the quad's four pixels (DPP), then the four pixel groups kq (lanes 16 apart)
        float4 s4 = ssum4, q4 = ssq4;
        s4.x += dpp_xor1(s4.x); s4.y += dpp_xor1(s4.y); s4.z += dpp_xor1(s4.z); s4.w += dpp_xor1(s4.w);
        q4.x += dpp_xor1(q4.x); q4.y += dpp_xor1(q4.y); q4.z += dpp_xor1(q4.z); q4.w += dpp_xor1(q4.w);
        s4.x += dpp_xor2(s4.x); s4.y += dpp_xor2(s4.y); s4.z += dpp_xor2(s4.z); s4.w += dpp_xor2(s4.w);
        q4.x += dpp_xor2(q4.x); q4.y += dpp_xor2(q4.y); q4.z += dpp_xor2(q4.z); q4.w += dpp_xor2(q4.w);
        float ssum = odd2 ? (odd1 ? s4.w : s4.z) : (odd1 ? s4.y : s4.x);      // lane m: channel m
        float ssq = odd2 ? (odd1 ? q4.w : q4.z) : (odd1 ? q4.y : q4.x);
        ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
        ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16) {
            s_red[(wave * 2 + 0) * 16 + m] = ssum;
            s_red[(wave * 2 + 1) * 16 + m] = ssq;
        }
        rd_sync();
        if (tid < 32) {
            const int which = tid >> 4, j = tid & 15;
            a.stat[((size_t)bid * 2 + which) * 16 + j] =
                s_red[(0 * 2 + which) * 16 + j] + s_red[(1 * 2 + which) * 16 + j] + s_red[(2 * 2 + which) * 16 + j] + s_red[(3 * 2 + which) * 16 + j];
        }
    }
}

// single phase, nine taps on a 3x3 stencil, unit strides, 16 -> 16 channels, output grid == input grid
bool conv16_eligible(const RdConvDesc& d) {
    static const char* off = getenv("RD_GCONV_NOC16");   // diagnostics: keep these layers on the 32x32 kernels
    if (off) return false;
    if (d.n_phases != 1 || d.Cin != 16 || d.Cout != 16 || d.in_stride != 1 || d.out_stride != 1) return false;
    const RdPhase& p = d.phase[0];
    if (p.n_taps != 9 || p.dh_max - p.dh_min != 2 || p.dw_max - p.dw_min != 2) return false;
    if (p.out_off_h != 0 || p.out_off_w != 0 || p.lh != d.Ho || p.lw != d.Wo || d.Hi != d.Ho || d.Wi != d.Wo) return false;
    if (d.ldi % 4 != 0 || d.ldo % 4 != 0) return false;
    int seen = 0;
    for (int t = 0; t < 9; ++t) seen |= 1 << ((p.dh[t] - p.dh_min) * 3 + (p.dw[t] - p.dw_min));
    return seen == 0x1ff;      // every position of the 3x3 stencil exactly once
}

int conv16_tiles_per_image(const RdConvDesc& d) { return cdiv(d.Ho, C16_T) * cdiv(d.Wo, C16_T); }

int launch_conv16(const RdConvDesc& d, const float* in, const float* w_packed, float* out, const float* addend, int ld_add,
                  float* stat, hipStream_t s) {
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(w_packed) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0 && (!addend || (reinterpret_cast<uintptr_t>(addend) % 16 == 0 && ld_add % 4 == 0)),
                 "conv16: tensors must be 16-byte aligned with channel strides that are multiples of 4");
    Conv16Args a;
    a.in = in; a.w = w_packed; a.out = out; a.addend = addend; a.stat = stat;
    a.ldi = d.ldi; a.ldo = d.ldo; a.ld_add = ld_add;
    a.N = d.N; a.H = d.Ho; a.W = d.Wo;
    a.tiles_h = cdiv(d.Ho, C16_T); a.tiles_w = cdiv(d.Wo, C16_T);
    const RdPhase& p = d.phase[0];
    a.ih_off = p.dh_min; a.iw_off = p.dw_min;
    for (int t = 0; t < 9; ++t) a.widx_pos[(p.dh[t] - p.dh_min) * 3 + (p.dw[t] - p.dw_min)] = p.widx[t];
    const int grid = d.N * a.tiles_h * a.tiles_w;
    if (stat) {
        if (addend) hipLaunchKernelGGL((conv16_kernel<true, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv16_kernel<true, false>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (addend) hipLaunchKernelGGL((conv16_kernel<false, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv16_kernel<false, false>), dim3(grid), dim3(256), 0, s, a);
    }
    RD_CHECK_LAUNCH("conv16_kernel");
    return RD_OK;
}

}  // namespace rd
