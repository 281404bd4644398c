// 3x3 / stride-1 convolutions with 16 input and 16 output channels (the depth encoder's layer1, the last decoder stage's conv2:
// model/models.py:30-72, 96-112) with fp32 arithmetic on the bf16 matrix cores: the split plans' form of conv16.hip.
//
// Operands are split into three bf16 pieces while they are staged (x = x0 + x1 + x2 exactly, gconv_split.hip), every product is rebuilt
// from six v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  conv16.hip's 16x16x4 fp32 MFMA needs 36 steps of 32 cycles per 16-pixel
// block (the layers ran at ~47 % of that rate: 25.8 us for the depth encoder's layer1, 73 us for dec4 conv2); here a block is
// 5 steps x 6 terms x 16 cycles.  Same contract, tile (16 x 16 output pixels, a wave owns four rows), packed fp32 weight operand
// ([tap][Cin/4][Cout] x 4, rd_pack_weights quad layout) and BatchNorm partial-sum layout as conv16.hip:
//   * K = 9 taps x 16 channels, two taps per MFMA step (tap 9 of the fifth step meets zero weights): lane (m = lane % 16, g = lane / 16)
//     feeds the 8 consecutive k = (tap 2 s + g / 2, channels 8 (g % 2) .. + 7) -- for the A operand the 16 bytes of one piece of patch
//     pixel (row + dy, m + dx), for the B operand eight weights of output channel m;
//   * the whole weight operand lives in registers: 5 steps x 3 pieces x 4 registers per lane, split once per workgroup;
//   * patch [piece][18 x 18 pixels][16 channels bf16] with the two 16-byte halves of a pixel swapped on every second group of eight pixels
//     (conflict-free b128 reads at a 32-byte pitch: 31 KB of LDS, four workgroups per CU).
#include <hip/hip_runtime.h>

#include "common.h"

namespace rd {

typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));

struct Conv16SpArgs {
    const float* in;
    const float* w;       // packed quads [9][4][16][4] (fp32)
    float* out;
    const float* addend;
    float* stat;
    int ldi, ldo, ld_add;
    int N, H, W;          // output grid == input grid (unit strides)
    int tiles_h, tiles_w;
    int ih_off, iw_off;   // dh_min, dw_min
    int widx_pos[9];      // weight slab of the tap at patch position (pos / 3, pos % 3)
};

constexpr int CS_T = 16, CS_P = 18;
constexpr int CS_PIX = 32;                          // bytes per patch pixel of one piece plane: two 16-byte halves (8 channels each),
                                                    // stored at half ^ ((pixel >> 3) & 1): the 16 lanes of a b128 read pass (16 consecutive pixels,
                                                    // one half) then fall into 16 bank groups (pixels p and p + 8 take opposite halves)
constexpr int CS_PL = CS_P * CS_P * CS_PIX;         // bytes per piece plane (10368): 31 KB for the three, four workgroups per CU
__device__ __forceinline__ int cs_unit(int px, int half) { return px * CS_PIX + ((half ^ ((px >> 3) & 1)) << 4); }

// two fp32 values -> their three bf16 pieces, packed (low half = first value); round to nearest even at every level
__device__ __forceinline__ void cs_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(a, b);
    a -= __uint_as_float(p0 << 16); b -= __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(a, b);
    a -= __uint_as_float(p1 << 16); b -= __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(a, b);
}

template <bool STAT, bool ADD>
__global__ __launch_bounds__(256) void conv16_split_kernel(const Conv16SpArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_patch[3 * CS_PL];
    __shared__ float s_red[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int bid = blockIdx.x;
    const int per_img = a.tiles_h * a.tiles_w;
    const int n = bid / per_img, trem = bid - n * per_img;
    const int r0 = (trem / a.tiles_w) * CS_T, c0 = (trem % a.tiles_w) * CS_T;
    const int ih0 = r0 + a.ih_off, iw0 = c0 + a.iw_off;

    // ---- the weight operand: step s, lane (m, g): tap position 2 s + g / 2, input channels 8 (g % 2) .. + 7, output channel m
    cbf16x8 Bw[5][3];
    int apx[5];           // patch pixel index of the lane's A unit in step s, relative to the wave's first row
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int pos = 2 * s + (g >> 1);
        const bool real = pos < 9;
        const int posc = real ? pos : 8;
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
        if (real) {
            const int slab = a.widx_pos[posc];
            const float* wp = a.w + ((size_t)(slab * 4 + (g & 1) * 2) * 16 + m) * 4;
            w0 = *reinterpret_cast<const float4*>(wp);
            w1 = *reinterpret_cast<const float4*>(wp + 16 * 4);
        }
        unsigned p[3][4];
        cs_split2(w0.x, w0.y, p[0][0], p[1][0], p[2][0]);
        cs_split2(w0.z, w0.w, p[0][1], p[1][1], p[2][1]);
        cs_split2(w1.x, w1.y, p[0][2], p[1][2], p[2][2]);
        cs_split2(w1.z, w1.w, p[0][3], p[1][3], p[2][3]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Bw[s][pc] = __builtin_bit_cast(cbf16x8, cu32x4{p[pc][0], p[pc][1], p[pc][2], p[pc][3]});
        apx[s] = (posc / 3) * CS_P + (posc % 3) + m;
    }

    // ---- halo patch [18][18][16] -> three piece planes in LDS (zero outside the image); all loads of a thread in flight before its first write
    const float* in_n = a.in + (size_t)n * a.H * a.W * a.ldi;
    {
        constexpr int U = 6;       // 18*18*4 = 1296 four-channel units <= 6 * 256
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + u * 256;
            const int px = e >> 2, q = e & 3;
            const int py = px / CS_P, pxx = px - py * CS_P;
            const int ih = ih0 + py, iw = iw0 + pxx;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < CS_P * CS_P * 4 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W)
                v[u] = *reinterpret_cast<const float4*>(in_n + ((size_t)ih * a.W + iw) * a.ldi + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + u * 256;
            if (e < CS_P * CS_P * 4) {
                unsigned p[3][2];
                cs_split2(v[u].x, v[u].y, p[0][0], p[1][0], p[2][0]);
                cs_split2(v[u].z, v[u].w, p[0][1], p[1][1], p[2][1]);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    *reinterpret_cast<cu32x2*>(s_patch + pc * CS_PL + cs_unit(e >> 2, (e >> 1) & 1) + 8 * (e & 1)) = cu32x2{p[pc][0], p[pc][1]};
            }
        }
    }
    rd_sync();

    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int px_w = (wave * 4) * CS_P;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        cbf16x8 A[4][3];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int off = cs_unit(px_w + mb * CS_P + apx[s], g & 1);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) A[mb][pc] = *reinterpret_cast<const cbf16x8*>(s_patch + pc * CS_PL + off);
        }
        // the six kept terms, smallest first; term-major: consecutive MFMAs go to different accumulators
#define RD_CS_TERM(pa, pb)                                                                                            \
    _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                                                   \
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mb][pa], Bw[s][pb], acc[mb], 0, 0, 0);
        RD_CS_TERM(2, 0) RD_CS_TERM(1, 1) RD_CS_TERM(0, 2) RD_CS_TERM(1, 0) RD_CS_TERM(0, 1) RD_CS_TERM(0, 0)
#undef RD_CS_TERM
    }

    // ---- epilogue (conv16.hip's): in the C/D layout of the 16x16 MFMA a lane holds channel m of the pixels 4 g .. 4 g + 3 of the block's
    // row; the 4x4 block is transposed in registers, after which lane q of a quad holds the channels 4 (m / 4) .. + 3 of pixel 4 g + q
    const int q = m & 3, cq = m & ~3;
    const bool odd1 = q & 1, odd2 = q & 2;
    float4 ssum4 = make_float4(0.f, 0.f, 0.f, 0.f), ssq4 = ssum4;
    const int c = c0 + 4 * g + q;
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

}  // namespace rd

using namespace rd;

// 1: rd_conv16_split serves this descriptor (exactly the descriptors conv16.hip serves inside rd_gconv)
extern "C" int rd_conv16_split_supported(const RdConvDesc* d) { return d && conv16_eligible(*d) ? 1 : 0; }

// rd_gconv's contract for those descriptors (fp32 tensors, the fp32 quad-packed weight operand of rd_pack_weights, optional residual
// addend, optional BatchNorm partial sums [N * tiles][2][16] with conv16.hip's 16 x 16 tiling: rd_gconv_stat_tiles_ws), fp32 arithmetic
// rebuilt from six bf16 MFMAs per product
extern "C" int rd_conv16_split(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* addend, int32_t ld_add,
                               float* stat_partial, void* stream) {
    RD_CHECK_ARG(d && in && w_packed && out, "conv16_split: null argument");
    RD_CHECK_ARG(conv16_eligible(*d), "conv16_split: not a 16 -> 16 channel 3x3 unit-stride descriptor (rd_conv16_split_supported)");
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(w_packed) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(out) % 16 == 0 && (!addend || (reinterpret_cast<uintptr_t>(addend) % 16 == 0 && ld_add % 4 == 0)),
                 "conv16_split: tensors must be 16-byte aligned with channel strides that are multiples of 4");
    Conv16SpArgs a;
    a.in = in; a.w = w_packed; a.out = out; a.addend = addend; a.stat = stat_partial;
    a.ldi = d->ldi; a.ldo = d->ldo; a.ld_add = ld_add;
    a.N = d->N; a.H = d->Ho; a.W = d->Wo;
    a.tiles_h = cdiv(d->Ho, CS_T); a.tiles_w = cdiv(d->Wo, CS_T);
    const RdPhase& p = d->phase[0];
    a.ih_off = p.dh_min; a.iw_off = p.dw_min;
    for (int t = 0; t < 9; ++t) a.widx_pos[(p.dh[t] - p.dh_min) * 3 + (p.dw[t] - p.dw_min)] = p.widx[t];
    const int grid = d->N * a.tiles_h * a.tiles_w;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (stat_partial) {
        if (addend) hipLaunchKernelGGL((conv16_split_kernel<true, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv16_split_kernel<true, false>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (addend) hipLaunchKernelGGL((conv16_split_kernel<false, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv16_split_kernel<false, false>), dim3(grid), dim3(256), 0, s, a);
    }
    RD_CHECK_LAUNCH("conv16_split_kernel");
    return RD_OK;
}
