// fp32 stem convolution on the bf16 matrix cores (7x7, stride 2, pad 3, Cin = 1..3 planes of the NCHW network input -> NHWC
// [N,Ho,Wo,Cout]; models.py:539,559,633,643): the split plans' form of rd_stem_fwd.  Input and weights are split into three bf16
// pieces while they are staged (x = x0 + x1 + x2 exactly, gconv_split.hip) and every product is rebuilt from six
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32 arithmetic at 2.67x the fp32 MFMA rate.
//
// Lowering = stem_bf16.hip's: no im2col buffer, K ordered (plane c, kernel row kh, kernel column kw padded 7 -> 8), so the A fragment of
// output pixel (r, col) for group (c, kh) is the 16 bytes at patch[c][2r + kh][2 col .. 2 col + 7] of the halo patch in LDS; same 8 x 32
// output tile, same partial-sum layout (rd_stem_stat_tiles).
//
// Schedule: measured on stem_bf16.hip's own schedule with three-piece operands (one role, two workgroups per CU) the MFMA walk (126 us of
// the launch at b = 16) simply adds to the staging / store / latency skeleton (120-150 us): nothing overlaps it.  Here a workgroup has
// EIGHT waves, one per CU: waves 0-3 own the accumulators and do fragment reads, MFMAs and the epilogue of tile i; waves 4-7 fetch the
// halo patch of tile i+1, split it and write its three piece planes into the other patch buffer meanwhile.  One barrier per tile hands the
// buffers over (a second one inside the partial-sum reduction).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace rd {

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));

struct StemSpArgs {
    const float* plane[3];
    long long stride[3];  // elements between consecutive images of each plane
    const float* w;       // packed fp32 [49][Cin][Cout]
    float* out;           // NHWC [N,Ho,Wo,Cout]
    float* stat;
    int Cin, N, H, W, Ho, Wo, Cout, tiles_h, tiles_w;
    int dbg;              // diagnostics (RD_STEM_SPLIT_DEBUG; results garbage): 2 no patch staging, 4 no output stores (run time); 1 no MFMA walk
                          // (an instantiation of the 64-channel RGB kernel)
};

constexpr int SS_TH = 8, SS_TW = 32;
constexpr int SS_PH = 2 * SS_TH + 5;         // 21 patch rows
constexpr int SS_PW = 72;                    // staged patch columns (2*32 + 5 = 69 needed, 2*31 + 8 = 70 read)
constexpr int SS_PPL = 3 * SS_PH * SS_PW;    // ushorts per piece plane of a patch buffer (9072 B: a multiple of 16)

// f(integral_constant<int, S>) for S = S0 .. ksteps-1: an unrolled loop whose index is a constant expression inside the body
template <int S, int N, typename F>
__device__ __forceinline__ void static_steps(F&& f) {
    if constexpr (S < N) {
        f(std::integral_constant<int, S>{});
        static_steps<S + 1, N>(f);
    }
}

// CIN and DBG are compile-time: a run-time test inside the unrolled walk (`if (s < ksteps)`, a diagnostic flag) cuts it into basic blocks
// and stops the scheduler from moving the next step's fragment reads over this step's MFMAs (213.7 -> 201.8 us).  DBG 1: no MFMA walk.
template <int NT, int CIN, int DBG>
__global__ __launch_bounds__(512) void stem_fwd_split_kernel(const StemSpArgs a) {
    constexpr int BN = NT * 32, MT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned short ssm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const bool loader = wave >= 4;
    constexpr int G = CIN * 7;               // (plane, kernel row) groups of eight k
    constexpr int ksteps = (G + 1) >> 1;
    constexpr int WPL = 2 * ksteps * BN * 8; // ushorts per piece plane of the weights
    unsigned short* s_patch = ssm;                                   // [2 buffers][3 pieces][Cin][21][72] bf16
    unsigned short* s_w = ssm + 2 * 3 * SS_PPL;                      // [3 pieces][2*ksteps groups][BN][8] bf16
    float* s_red = reinterpret_cast<float*>(s_w + 3 * WPL);          // [4 waves][2][BN]

    const int tiles_img = a.tiles_h * a.tiles_w;
    const int total_tiles = a.N * tiles_img;

    // ---- weights, once per workgroup (persistent: the tiles are walked with a grid stride): group g = (c, kh), element j = kw (j = 7 and the
    // padding group: zero); three pieces of every value
    for (int u = tid; u < 2 * ksteps * BN; u += 512) {
        const int g = u / BN, co = u - g * BN;
        const int c = g / 7, kh = g - c * 7;
        sbf16x8 v[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = 0.f;
            if (g < G && j < 7 && co < a.Cout) f = a.w[((size_t)(kh * 7 + j) * CIN + c) * a.Cout + co];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                v[pc][j] = (__bf16)f;
                f -= (float)v[pc][j];        // (exact: the remainder of a round-to-nearest bf16 fits fp32)
            }
        }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<sbf16x8*>(s_w + (size_t)pc * WPL + (size_t)u * 8) = v[pc];
    }

    if (loader) {
        // ================= staging waves: patch of the NEXT tile -> the other buffer (zero outside the image)
        const int lt = tid - 256;
        constexpr int UPB = (SS_PH * SS_PW + 255) / 256;      // 6 patch elements per thread and plane
        int pyx[UPB];                                         // row << 16 | column, row = 30000 for slots past the plane
#pragma unroll
        for (int q = 0; q < UPB; ++q) {
            const int u = lt + q * 256;
            const int row = u / SS_PW, col = u - row * SS_PW;
            pyx[q] = ((u < SS_PH * SS_PW ? row : 30000) << 16) | col;
        }
        auto stage = [&](int tile_, int buf) {
            const int n_ = tile_ / tiles_img, trem_ = tile_ - n_ * tiles_img;
            const int ih0_ = 2 * (trem_ / a.tiles_w) * SS_TH - 3, iw0_ = 2 * (trem_ % a.tiles_w) * SS_TW - 3;
            float f[3][UPB];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c < CIN) {
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
            if (a.dbg & 2) return;
            unsigned short* pb = s_patch + (size_t)buf * 3 * SS_PPL;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < CIN) {
#pragma unroll
                    for (int q = 0; q < UPB; ++q) {
                        float x = f[c][q];
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            const __bf16 b = (__bf16)x;
                            x -= (float)b;
                            if (lt + q * 256 < SS_PH * SS_PW) pb[pc * SS_PPL + c * SS_PH * SS_PW + lt + q * 256] = __builtin_bit_cast(unsigned short, b);
                        }
                    }
                }
        };
        if ((int)blockIdx.x < total_tiles) stage(blockIdx.x, 0);
        int it = 0;
        for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x, ++it) {
            rd_sync();                                        // (1) buffer it & 1 is complete; the compute waves are done with the other one
            if (tile + (int)gridDim.x < total_tiles) stage(tile + gridDim.x, (it + 1) & 1);
            if (a.stat) { rd_sync(); }                        // (2) the compute waves' partial-sum hand-over
        }
        return;
    }

    // ================= compute waves
    int goff[ksteps];                        // ushort offset of the lane's group in step s (clamped: the padding group reads real
#pragma unroll                               //   data against zero weights)
    for (int s = 0; s < ksteps; ++s) {
        const int g = min(2 * s + hh, G - 1);
        const int c = g / 7, kh = g - c * 7;
        goff[s] = (c * SS_PH + kh) * SS_PW + 2 * l31;
    }
    int it = 0;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x, ++it) {
        const int n = tile / tiles_img, trem = tile - n * tiles_img;
        const int r0 = (trem / a.tiles_w) * SS_TH, c0 = (trem % a.tiles_w) * SS_TW;
        rd_sync();                                            // (1)
        const unsigned short* pb = s_patch + (size_t)(it & 1) * 3 * SS_PPL;

        // ---- MFMA walk.  Wave w owns tile rows 2w, 2w+1 (M-tile = one tile row, lane l31 = column); lane half hh takes group 2s+hh.
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        // Tile row 2w+1's fragment of group (c, kh) is tile row 2w's fragment of group (c, kh + 2) -- the same patch row -- which is the same
        // lane half's group of the NEXT step: where that holds for both lane halves (kh <= 4 and a real group: 6 of the 11 steps of the RGB
        // stem) row 2w's next fragment is read one step early and serves both (the b32 fragment reads are the walk's second bottleneck)
        auto load_a = [&](int s, int mt, sbf16x8 (&A)[3]) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const unsigned* p = reinterpret_cast<const unsigned*>(pb + pc * SS_PPL + goff[s] + 2 * (wave * MT + mt) * SS_PW);
                su32x4 av;
                av[0] = p[0]; av[1] = p[1]; av[2] = p[2]; av[3] = p[3];
                A[pc] = __builtin_bit_cast(sbf16x8, av);
            }
        };
        sbf16x8 Acarry[3];
        static_steps<0, ksteps>([&](auto SC) {
            constexpr int s = decltype(SC)::value;
            constexpr bool reuse = s + 1 < ksteps && 2 * s + 1 < G && (2 * s) % 7 <= 4 && (2 * s + 1) % 7 <= 4;
            constexpr bool carried = s > 0 && (s < ksteps && 2 * (s - 1) + 1 < G && (2 * (s - 1)) % 7 <= 4 && (2 * (s - 1) + 1) % 7 <= 4);
            if constexpr (!(DBG & 1)) {
                sbf16x8 B[NT][3];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        B[nt][pc] = *reinterpret_cast<const sbf16x8*>(s_w + (size_t)pc * WPL + ((size_t)(2 * s + hh) * BN + nt * 32 + l31) * 8);
                sbf16x8 A[MT][3];
                if constexpr (carried) { A[0][0] = Acarry[0]; A[0][1] = Acarry[1]; A[0][2] = Acarry[2]; }
                else load_a(s, 0, A[0]);
                if constexpr (reuse) { load_a(s + 1, 0, A[1]); Acarry[0] = A[1][0]; Acarry[1] = A[1][1]; Acarry[2] = A[1][2]; }
                else load_a(s, 1, A[1]);
                // the six kept terms, smallest first (gconv_split.hip); term-major so that consecutive MFMAs go to different accumulators
#define RD_SS_TERM(pa, pbb)                                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                 \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][pa], B[nt][pbb], acc[mt][nt], 0, 0, 0);
                RD_SS_TERM(2, 0) RD_SS_TERM(1, 1) RD_SS_TERM(0, 2) RD_SS_TERM(1, 0) RD_SS_TERM(0, 1) RD_SS_TERM(0, 0)
#undef RD_SS_TERM
            }
        });

        // ---- epilogue: NHWC store + BatchNorm partial sums (accumulator row = pixel column of the tile row, lane = channel)
        float ssum[NT], ssq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
        const bool full = r0 + SS_TH <= a.Ho && c0 + SS_TW <= a.Wo && BN <= a.Cout;      // workgroup-uniform: no masking at all
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
                // stores four consecutive channels of one pixel
                const size_t rowo = (((size_t)n * a.Ho + r) * a.Wo + c0 + q4l + 4 * hh) * a.Cout + 4 * k4l;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float e0 = acc[mt][nt][4 * g], e1 = acc[mt][nt][4 * g + 1], e2 = acc[mt][nt][4 * g + 2], e3 = acc[mt][nt][4 * g + 3];
                        quad_transpose(e0, e1, e2, e3, odd1, odd2);
                        const float4 v = make_float4(e0, e1, e2, e3);
                        const size_t o = rowo + (size_t)(8 * g) * a.Cout + nt * 32;
                        if (!(a.dbg & 4)) st4(a.out + o, v);
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
                        a.out[(((size_t)n * a.Ho + r) * a.Wo + c) * a.Cout + co] = v;
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
            // (s_red is read by the first 2 BN threads after barrier (2) and rewritten only after the next tile's barrier (1))
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
                const float q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
                if (hh == 0) {
                    s_red[(wave * 2 + 0) * BN + nt * 32 + l31] = s;
                    s_red[(wave * 2 + 1) * BN + nt * 32 + l31] = q;
                }
            }
            rd_sync();                                    // (2)
            if (tid < 2 * BN) {
                const int which = tid / BN, j = tid - which * BN;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) s += s_red[(w * 2 + which) * BN + j];
                if (j < a.Cout) a.stat[((size_t)tile * 2 + which) * a.Cout + j] = s;
            }
        }
    }
}

}  // namespace rd

using namespace rd;

extern "C" int rd_stem_fwd_split(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                                 const float* w_packed, int32_t Cout, float* out, float* stat_partial, void* stream) {
    RD_CHECK_ARG(planes && strides && Cin >= 1 && Cin <= 3 && N > 0 && H > 6 && W > 6, "stem_split: bad arguments");
    RD_CHECK_ARG(Cout == 64 || Cout == 16 || Cout == 32, "stem_split: Cout=%d unsupported", Cout);
    RD_CHECK_ARG(w_packed && out, "stem_split: null tensor");
    StemSpArgs a;
    for (int i = 0; i < 3; ++i) {
        a.plane[i] = i < Cin ? planes[i] : nullptr;
        a.stride[i] = i < Cin ? strides[i] : 0;
        RD_CHECK_ARG(i >= Cin || planes[i], "stem_split: null plane %d", i);
    }
    a.w = w_packed; a.out = out; a.stat = stat_partial;
    a.Cin = Cin; a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_h = cdiv(a.Ho, SS_TH); a.tiles_w = cdiv(a.Wo, SS_TW);
    static const int dbg = getenv("RD_STEM_SPLIT_DEBUG") ? atoi(getenv("RD_STEM_SPLIT_DEBUG")) : 0;
    a.dbg = dbg;
    const int total = N * a.tiles_h * a.tiles_w;
    const int grid = total < num_cus() ? total : num_cus();      // one 8-wave workgroup per CU
    const int NT = Cout > 32 ? 2 : 1;
    const int ksteps = (Cin * 7 + 1) / 2;
    const size_t lds = ((size_t)2 * 3 * SS_PPL + (size_t)3 * 2 * ksteps * NT * 32 * 8) * 2 + (size_t)4 * 2 * NT * 32 * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto launch = [&](auto k) -> int {
        static std::atomic<unsigned long long> attr_done{0};      // (one flag per instantiation of this generic lambda)
        RD_SET_ATTR_ONCE(attr_done, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
        return RD_OK;
    };
    int rc;
    if (NT == 2 && Cin == 3) rc = (dbg & 1) ? launch(stem_fwd_split_kernel<2, 3, 1>) : launch(stem_fwd_split_kernel<2, 3, 0>);
    else if (NT == 2) rc = Cin == 2 ? launch(stem_fwd_split_kernel<2, 2, 0>) : launch(stem_fwd_split_kernel<2, 1, 0>);
    else rc = Cin == 3 ? launch(stem_fwd_split_kernel<1, 3, 0>) : Cin == 2 ? launch(stem_fwd_split_kernel<1, 2, 0>) : launch(stem_fwd_split_kernel<1, 1, 0>);
    if (rc != RD_OK) return rc;
    RD_CHECK_LAUNCH("stem_fwd_split_kernel");
    return RD_OK;
}
