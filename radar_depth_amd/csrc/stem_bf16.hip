// bf16-operand form of the stem convolution (7x7, stride 2, pad 3, Cin = 1..3 planes of the NCHW network input -> NHWC
// [N,Ho,Wo,Cout]; models.py:539,559,633,643) on v_mfma_f32_32x32x16_bf16.  Opt-in with the rest of the bf16 path; rd_stem_fwd
// (fp32) stays the default.  Same arguments, same tile geometry (8 x 32 output pixels, so the BN partial sums have the layout
// rd_stem_stat_tiles promises) and the same packed fp32 weights as rd_stem_fwd.
//
// The reduction is lowered WITHOUT an im2col buffer: the K index is ordered (plane c, kernel row kh, kernel column kw) with kw
// padded from 7 to 8, so that the eight consecutive k a lane feeds to one MFMA are exactly eight consecutive columns of one
// patch row -- the A fragment of output pixel (r, col) for group (c, kh) is the 16 bytes at patch[c][2r + kh][2 col .. 2 col + 7]
// of the bf16 halo patch in LDS.  The eighth column meets a zero weight.  K = 8 * 7 * Cin (168 for RGB, 10.5 MFMA steps).
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace rd {

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));

struct StemBfArgs {
    const float* plane[3];
    long long stride[3];  // elements between consecutive images of each plane
    const float* w;       // packed fp32 [49][Cin][Cout]
    void* out;            // NHWC [N,Ho,Wo,Cout]: fp32, or bf16 when io16
    float* stat;
    int io16;
    int Cin, N, H, W, Ho, Wo, Cout, tiles_h, tiles_w;
};

constexpr int SB_TH = 8, SB_TW = 32;
constexpr int SB_PH = 2 * SB_TH + 5;         // 21 patch rows
constexpr int SB_PW = 72;                    // staged patch columns (2*32 + 5 = 69 needed, 2*31 + 8 = 70 read)
constexpr int SB_KS = 11;                    // 16-wide MFMA steps for Cin = 3 (21 groups of 8 -> 22)

template <int NT>
__global__ __launch_bounds__(256) void stem_fwd_bf16_kernel(const StemBfArgs a) {
    constexpr int BN = NT * 32, MT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned short ssm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int G = a.Cin * 7;                 // (plane, kernel row) groups of eight k
    const int ksteps = (G + 1) >> 1;
    unsigned short* s_patch = ssm;                                   // [Cin][21][72] bf16
    unsigned short* s_w = ssm + 3 * SB_PH * SB_PW;                   // [2*ksteps groups][BN][8] bf16 (16-byte aligned: 3*21*72*2 B)

    const int tiles_img = a.tiles_h * a.tiles_w;
    const int total_tiles = a.N * tiles_img;

    // ---- weights (once per workgroup: the workgroups are persistent and walk the tiles with a grid stride): group g = (c, kh), element j = kw (j = 7 and the padding group: zero)
    for (int u = tid; u < 2 * ksteps * BN; u += 256) {
        const int g = u / BN, co = u - g * BN;
        const int c = g / 7, kh = g - c * 7;
        sbf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = 0.f;
            if (g < G && j < 7 && co < a.Cout) f = a.w[((size_t)(kh * 7 + j) * a.Cin + c) * a.Cout + co];
            v[j] = (__bf16)f;
        }
        *reinterpret_cast<sbf16x8*>(s_w + (size_t)u * 8) = v;
    }
    int goff[SB_KS];                         // ushort offset of the lane's group in step s (clamped: the padding group reads
#pragma unroll                               //   real data against zero weights)
    for (int s = 0; s < SB_KS; ++s) {
        const int g = min(2 * s + hh, G - 1);
        const int c = g / 7, kh = g - c * 7;
        goff[s] = (c * SB_PH + kh) * SB_PW + 2 * l31;
    }

    // ---- halo patch (zero outside the image).  The NEXT tile's patch is fetched into registers before the MFMA walk of the current
    // one (stem.hip's scheme: slot q of a thread is patch element tid + 256 q of every plane, (row, column) computed once per
    // workgroup, one buffer descriptor per plane with out-of-range -> 0) and rounded to bf16 when it is written to LDS.
    constexpr int UPB = (SB_PH * SB_PW + 255) / 256;      // 6 patch elements per thread and plane
    int pyx[UPB];                                         // row << 16 | column, row = 30000 for slots past the plane
#pragma unroll
    for (int q = 0; q < UPB; ++q) {
        const int u = tid + q * 256;
        const int row = u / SB_PW, col = u - row * SB_PW;
        pyx[q] = ((u < SB_PH * SB_PW ? row : 30000) << 16) | col;
    }
    auto fetch_patch = [&](int tile_, float (&f)[3][UPB]) {
        const int n_ = tile_ / tiles_img, trem_ = tile_ - n_ * tiles_img;
        const int ih0_ = 2 * (trem_ / a.tiles_w) * SB_TH - 3, iw0_ = 2 * (trem_ % a.tiles_w) * SB_TW - 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c < a.Cin) {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(a.plane[c] + (size_t)n_ * a.stride[c]), 0, (unsigned)(a.H * a.W) * 4u, 0x00020000);
#pragma unroll
                for (int q = 0; q < UPB; ++q) {
                    const int ih = ih0_ + (pyx[q] >> 16), iw = iw0_ + (pyx[q] & 0xffff);
                    const unsigned off = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? (unsigned)(ih * a.W + iw) * 4u : 0x80000000u;
                    f[c][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
                }
            }
        }
    };
    float fnext[3][UPB];
    if ((int)blockIdx.x < total_tiles) fetch_patch(blockIdx.x, fnext);
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int n = tile / tiles_img, trem = tile - n * tiles_img;
    const int r0 = (trem / a.tiles_w) * SB_TH, c0 = (trem % a.tiles_w) * SB_TW;
    rd_sync();                         // the previous tile's MFMAs / partial sums are done with the patch
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < a.Cin) {
#pragma unroll
            for (int q = 0; q < UPB; ++q) {
                const __bf16 b = (__bf16)fnext[c][q];
                if (tid + q * 256 < SB_PH * SB_PW) s_patch[c * SB_PH * SB_PW + tid + q * 256] = __builtin_bit_cast(unsigned short, b);
            }
        }
    rd_sync();
    if (tile + (int)gridDim.x < total_tiles) fetch_patch(tile + gridDim.x, fnext);      // in flight during the walk and the epilogue

    // ---- MFMA walk.  Wave w owns tile rows 2w, 2w+1 (M-tile = one tile row, lane l31 = column); lane half hh takes group 2s+hh.
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
#pragma unroll
    for (int s = 0; s < SB_KS; ++s) {
        if (s < ksteps) {
            sbf16x8 B[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                B[nt] = *reinterpret_cast<const sbf16x8*>(s_w + ((size_t)(2 * s + hh) * BN + nt * 32 + l31) * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned* p = reinterpret_cast<const unsigned*>(s_patch + goff[s] + 2 * (wave * MT + mt) * SB_PW);
                su32x4 av;
                av[0] = p[0]; av[1] = p[1]; av[2] = p[2]; av[3] = p[3];
                const sbf16x8 A = __builtin_bit_cast(sbf16x8, av);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: NHWC store + BatchNorm partial sums (accumulator row = pixel column of the tile row, lane = channel)
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    const bool full = r0 + SB_TH <= a.Ho && c0 + SB_TW <= a.Wo && BN <= a.Cout;      // workgroup-uniform: no masking at all
    const int q4l = l31 & 3, k4l = l31 >> 2;
    const bool odd1 = q4l & 1, odd2 = q4l & 2;
    float4 ssum4[NT], ssq4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum4[nt] = ssq4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = r0 + wave * MT + mt;
        if (full) {
            // row-major: each 4x4 block (4 registers x the 4 lanes of a quad) transposed with two DPP exchanges, then a lane
            // stores four consecutive channels of one pixel -- a quarter of the (issue-bound) store instructions
            const size_t rowo = (((size_t)n * a.Ho + r) * a.Wo + c0 + q4l + 4 * hh) * a.Cout + 4 * k4l;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float e0 = acc[mt][nt][4 * g], e1 = acc[mt][nt][4 * g + 1], e2 = acc[mt][nt][4 * g + 2], e3 = acc[mt][nt][4 * g + 3];
                    quad_transpose(e0, e1, e2, e3, odd1, odd2);
                    const float4 v = make_float4(e0, e1, e2, e3);
                    const size_t o = rowo + (size_t)(8 * g) * a.Cout + nt * 32;
                    if (a.io16) st4(static_cast<bf16s*>(a.out) + o, v);
                    else st4(static_cast<float*>(a.out) + o, v);
                    ssum4[nt].x += v.x; ssum4[nt].y += v.y; ssum4[nt].z += v.z; ssum4[nt].w += v.w;
                    ssq4[nt].x += v.x * v.x; ssq4[nt].y += v.y * v.y; ssq4[nt].z += v.z * v.z; ssq4[nt].w += v.w * v.w;
                }
            }
            continue;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 32 + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = c0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (r < a.Ho && c < a.Wo && co < a.Cout) {
                    const float v = acc[mt][nt][i];
                    const size_t o = (((size_t)n * a.Ho + r) * a.Wo + c) * a.Cout + co;
                    if (a.io16) st1(static_cast<bf16s*>(a.out) + o, v);
                    else static_cast<float*>(a.out)[o] = v;
                    ssum[nt] += v;
                    ssq[nt] += v * v;
                }
            }
        }
    }
    if (full) {      // back to one channel per lane: sum the quad's four pixels, lane q keeps channel q
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float4 s4 = ssum4[nt], q4 = ssq4[nt];
            s4.x += dpp_xor1(s4.x); s4.y += dpp_xor1(s4.y); s4.z += dpp_xor1(s4.z); s4.w += dpp_xor1(s4.w);
            q4.x += dpp_xor1(q4.x); q4.y += dpp_xor1(q4.y); q4.z += dpp_xor1(q4.z); q4.w += dpp_xor1(q4.w);
            s4.x += dpp_xor2(s4.x); s4.y += dpp_xor2(s4.y); s4.z += dpp_xor2(s4.z); s4.w += dpp_xor2(s4.w);
            q4.x += dpp_xor2(q4.x); q4.y += dpp_xor2(q4.y); q4.z += dpp_xor2(q4.z); q4.w += dpp_xor2(q4.w);
            ssum[nt] = odd2 ? (odd1 ? s4.w : s4.z) : (odd1 ? s4.y : s4.x);
            ssq[nt] = odd2 ? (odd1 ? q4.w : q4.z) : (odd1 ? q4.y : q4.x);
        }
    }
    if (a.stat) {
        rd_sync();
        float* red = reinterpret_cast<float*>(ssm);       // [4 waves][2][BN] (over the patch: every wave is past its MFMAs)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
            const float q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
            if (hh == 0) {
                red[(wave * 2 + 0) * BN + nt * 32 + l31] = s;
                red[(wave * 2 + 1) * BN + nt * 32 + l31] = q;
            }
        }
        rd_sync();
        if (tid < 2 * BN) {
            const int which = tid / BN, j = tid - which * BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * BN + j];
            if (j < a.Cout) a.stat[((size_t)tile * 2 + which) * a.Cout + j] = s;
        }
    }
    }   // tile loop
}

}  // namespace rd

using namespace rd;

static int stem_fwd_bf16_impl(int io16, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                              const float* w_packed, int32_t Cout, void* out, float* stat_partial, void* stream) {
    RD_CHECK_ARG(planes && strides && Cin >= 1 && Cin <= 3 && N > 0 && H > 6 && W > 6, "stem_bf16: bad arguments");
    RD_CHECK_ARG(Cout == 64 || Cout == 16 || Cout == 32, "stem_bf16: Cout=%d unsupported", Cout);
    RD_CHECK_ARG(w_packed && out, "stem_bf16: null tensor");
    StemBfArgs a;
    for (int i = 0; i < 3; ++i) {
        a.plane[i] = i < Cin ? planes[i] : nullptr;
        a.stride[i] = i < Cin ? strides[i] : 0;
        RD_CHECK_ARG(i >= Cin || planes[i], "stem_bf16: null plane %d", i);
    }
    a.w = w_packed; a.out = out; a.stat = stat_partial; a.io16 = io16;
    a.Cin = Cin; a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_h = cdiv(a.Ho, SB_TH); a.tiles_w = cdiv(a.Wo, SB_TW);
    const int total = N * a.tiles_h * a.tiles_w;
    const int per_cu = Cout > 32 ? 2 : 3;                    // resident workgroups per CU (register-limited)
    const int grid = total < per_cu * num_cus() ? total : per_cu * num_cus();
    const int NT = Cout > 32 ? 2 : 1;
    const int ksteps = (Cin * 7 + 1) / 2;
    const size_t lds = ((size_t)3 * SB_PH * SB_PW + (size_t)2 * ksteps * NT * 32 * 8) * 2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (NT == 2) hipLaunchKernelGGL(stem_fwd_bf16_kernel<2>, dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(stem_fwd_bf16_kernel<1>, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("stem_fwd_bf16_kernel");
    return RD_OK;
}

extern "C" int rd_stem_fwd_bf16(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                                const float* w_packed, int32_t Cout, float* out, float* stat_partial, void* stream) {
    return stem_fwd_bf16_impl(0, planes, strides, Cin, N, H, W, w_packed, Cout, out, stat_partial, stream);
}
// storage-typed form: dtype selects the element type of the NHWC output tensor
extern "C" int rd_stem_fwd_bf16_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                                  int32_t W, const float* w_packed, int32_t Cout, void* out, float* stat_partial, void* stream) {
    RD_CHECK_ARG(dtype == RD_DTYPE_F32 || dtype == RD_DTYPE_BF16, "stem_fwd_bf16_t: bad dtype %d", dtype);
    return stem_fwd_bf16_impl(dtype == RD_DTYPE_BF16, planes, strides, Cin, N, H, W, w_packed, Cout, out, stat_partial, stream);
}
