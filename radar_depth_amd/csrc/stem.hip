// Stem convolutions: 7x7, stride 2, pad 3, read straight from the network's NCHW input planes
// (model/models.py:539,559,633,643; multistage_model.py:163-164,236-241), output NHWC.
//   forward : implicit GEMM  [pixels] x [K = 49*Cin] x [Cout]  on v_mfma_f32_32x32x2_f32; the input patch of
//             an 8x32 output tile lives in LDS as planes, an LDS table maps k -> (ci,kh,kw) patch offset.
//   wgrad   : [K] x [pixels] x [Cout] with the pixel walk split over the four waves, slabs reduced like wgrad.hip.
//   dgrad   : only the stage-2 dense-depth input channel needs one (stage-1 output is not detached).
// Weights use the packed layout [49][Cin][Cout] (rd_pack_weights), i.e. k = (kh*7+kw)*Cin + ci.
#include "common.h"

namespace rd {

int launch_slab_reduce(const float* slabs, int n_splits, int64_t E, float* tmp, float* grad_oihw, int S, int Cin, int Cout,
                       int O, int I, int co_off, int accumulate, hipStream_t s);

struct StemArgs {
    const float* plane[3];
    long long stride[3];  // elements between consecutive images of each plane
    const float* w;       // packed [49][Cin][Cout]
    const void* dout;     // wgrad: NHWC [N,Ho,Wo,Cout], fp32 or (io16) bf16
    void* out;            // forward: NHWC [N,Ho,Wo,Cout], fp32 or (io16) bf16
    float* stat;
    int io16;             // bf16-storage plans: the NHWC tensors are bf16 (the network input planes stay fp32)
    int Cin, N, H, W, Ho, Wo, Cout, tiles_h, tiles_w;
    int total_tiles, tiles_per_split;  // wgrad
    int debug;            // ablation bits (RD_STEM_DEBUG): 1 stage only the first tile's patch, 2 skip the MFMA walk, 4 skip the stores
                          // (weight gradient, RD_STEM_WGRAD_DEBUG: 1 stage dout once per workgroup, 2 skip the MFMA walk)
};

constexpr int ST_TW = 32;
constexpr int ST_PW = 2 * ST_TW + 5;  // 69

template <int NT>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemArgs a) {
    constexpr int TH = 8, PH = 2 * TH + 5, BN = NT * 32, MT = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int K = 49 * a.Cin, Kp = (K + 1) & ~1;
    const int PLANE = PH * ST_PW;
    const int Kq = Kp + 4;                              // slack rows: the pipelined loop prefetches up to two steps past the end
    int* s_koff = reinterpret_cast<int*>(smem);        // [Kq]
    float* s_w = smem + ((Kq + 3) & ~3);               // [Kq][BN]
    float* s_patch = s_w + (size_t)Kq * BN;            // [Cin][PH][PW]
    float* s_red = s_patch + (size_t)a.Cin * PLANE;    // [4][2][BN] statistics scratch (the weights stay resident)

    for (int k = tid; k < Kq; k += 256) {
        int off = 0;
        if (k < K) {
            const int t = k / a.Cin, ci = k - t * a.Cin;
            off = ci * PLANE + (t / 7) * ST_PW + (t % 7);
        }
        s_koff[k] = off;
    }
    // staging is batched (U independent loads per thread in flight before the first LDS write)
    constexpr int U = 8;
    for (int base = tid; base < Kq * (BN / 4); base += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 256;
            const int k = e / (BN / 4), j = (e - k * (BN / 4)) * 4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Kq * (BN / 4) && k < K && j < a.Cout) v[u] = *reinterpret_cast<const float4*>(a.w + (size_t)k * a.Cout + j);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * 256;
            if (e < Kq * (BN / 4)) *reinterpret_cast<float4*>(s_w + (size_t)e * 4) = v[u];
        }
    }
    // PERSISTENT workgroups: the k->offset table and the weight operand (38 KB for the RGB stem -- more than twice the patch of
    // a tile, for 4.8 MFLOP of work) are staged once per workgroup, not once per tile; the grid is two workgroups per CU and each
    // walks the tiles bid, bid + grid, ...
    // The patch of the NEXT tile is fetched into registers before the MFMA walk of the current tile and written to LDS after it
    // (the RD_STEM_DEBUG ablation charged 140 of the kernel's 411 us to staging: three exposed memory round trips per tile plus
    // ~40 VALU instructions of index arithmetic per element on the lanes the co-resident workgroup's fp32 MFMAs need).  A thread's
    // elements are (plane ci, slot u): patch pixel e = tid + 256 u of EVERY plane, so (row, column) are computed once per
    // workgroup, the plane is compile-time, and each plane is read through its own buffer descriptor (out-of-range -> 0).
    const int total_tiles = a.N * a.tiles_h * a.tiles_w;
    constexpr int UP = (PH * ST_PW + 255) / 256;          // 6 patch pixels per thread and plane
    int prel[UP];                                         // py * W + px, or a huge value for slots past the plane
    short ppy[UP], ppx[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int e = tid + u * 256;
        const int py = e / ST_PW, px = e - py * ST_PW;
        ppy[u] = (short)(e < PLANE ? py : 30000);
        ppx[u] = (short)px;
        prel[u] = py * a.W + px;
    }
    auto fetch_patch = [&](int bid_, float (&v)[3][UP]) {
        const int n_ = bid_ / (a.tiles_h * a.tiles_w);
        const int trem_ = bid_ - n_ * (a.tiles_h * a.tiles_w);
        const int ih0_ = 2 * (trem_ / a.tiles_w) * TH - 3, iw0_ = 2 * (trem_ % a.tiles_w) * ST_TW - 3;
        const int org = ih0_ * a.W + iw0_;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            if (ci < a.Cin) {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(a.plane[ci] + (size_t)n_ * a.stride[ci]), 0, (unsigned)(a.H * a.W) * 4u, 0x00020000);
#pragma unroll
                for (int u = 0; u < UP; ++u) {
                    const int ih = ih0_ + ppy[u], iw = iw0_ + ppx[u];
                    const unsigned off = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? (unsigned)(org + prel[u]) * 4u : 0x80000000u;
                    v[ci][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
                }
            }
        }
    };
    float vnext[3][UP];
    if ((int)blockIdx.x < total_tiles) fetch_patch(blockIdx.x, vnext);
    for (int bid = blockIdx.x; bid < total_tiles; bid += gridDim.x) {
    const int n = bid / (a.tiles_h * a.tiles_w);
    const int trem = bid - n * (a.tiles_h * a.tiles_w);
    const int r0 = (trem / a.tiles_w) * TH, c0 = (trem % a.tiles_w) * ST_TW;
    rd_sync();          // the previous tile's MFMAs are done with the patch (and its statistics with s_red)
    if (!((a.debug & 1) && bid != (int)blockIdx.x)) {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
            if (ci < a.Cin) {
#pragma unroll
                for (int u = 0; u < UP; ++u)
                    if (tid + u * 256 < PLANE) s_patch[ci * PLANE + tid + u * 256] = vnext[ci][u];
            }
    }
    rd_sync();
    if (bid + (int)gridDim.x < total_tiles) fetch_patch(bid + gridDim.x, vnext);      // in flight during the walk and the epilogue

    // wave w owns output rows 2w, 2w+1 of the tile (two 32-pixel M tiles)
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) abase[mt] = (2 * (wave * MT + mt)) * ST_PW + 2 * l31;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    // K walk, two k per MFMA.  Software pipeline pinned with sched_barrier (see gconv.hip): two register sets ping-pong;
    // while one step's MT*NT MFMAs issue, the next step's fragments and the k->offset entry after that are in flight.
    const int nst = (a.debug & 2) ? 2 : Kp / 2;              // Kp is even; s_koff / s_w have >= 4 slack entries past Kp (host LDS sizing)
    float a0[MT], b0[NT], a1[MT], b1[NT];
    int ko_n = s_koff[hh];
#define RD_ST_LOAD(AV, BV, STEP)                                                             \
    {                                                                                        \
        const int k_ = 2 * (STEP) + hh;                                                      \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) AV[mt] = s_patch[abase[mt] + ko_n]; \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) BV[nt] = s_w[(size_t)k_ * BN + nt * 32 + l31]; \
        ko_n = s_koff[k_ + 2];                                                               \
    }
#define RD_ST_MFMA(AV, BV)                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                        \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[mt], BV[nt], acc[mt][nt], 0, 0, 0);
    RD_ST_LOAD(a0, b0, 0)
    for (int st = 0; st < nst; st += 2) {
        RD_ST_LOAD(a1, b1, st + 1)
        __builtin_amdgcn_sched_barrier(0);
        RD_ST_MFMA(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < nst) {
            RD_ST_LOAD(a0, b0, st + 2)
            __builtin_amdgcn_sched_barrier(0);
            RD_ST_MFMA(a1, b1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef RD_ST_LOAD
#undef RD_ST_MFMA

    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    // Full tiles store ROW-MAJOR: in the C/D layout a lane holds one channel of 16 pixels -- 64 four-byte (two-byte) stores per
    // lane and tile, and the store pipe is issue-bound (this epilogue, not the MFMAs, set the kernel's time: 43 % of the fp32
    // peak).  Each 4x4 block (4 registers x the 4 lanes of a quad) is transposed with two DPP exchanges (common.h), after which
    // a lane holds four consecutive channels of one pixel: a quarter of the store instructions.
    const bool full = r0 + TH <= a.Ho && c0 + ST_TW <= a.Wo && (a.Cout & 3) == 0;      // workgroup-uniform
    if (full) {
        const int q4l = l31 & 3, k4l = l31 >> 2;
        const bool odd1 = q4l & 1, odd2 = q4l & 2;
        float4 ssum4[NT], ssq4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ssum4[nt] = ssq4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oh = r0 + wave * MT + mt;
            const size_t rowo = (((size_t)n * a.Ho + oh) * a.Wo + c0 + q4l + 4 * hh) * a.Cout + 4 * k4l;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool cok4 = 4 * k4l + nt * 32 < a.Cout;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float e0 = acc[mt][nt][4 * g], e1 = acc[mt][nt][4 * g + 1], e2 = acc[mt][nt][4 * g + 2], e3 = acc[mt][nt][4 * g + 3];
                    quad_transpose(e0, e1, e2, e3, odd1, odd2);
                    const float4 v = make_float4(e0, e1, e2, e3);
                    if (cok4) {
                        const size_t o = rowo + (size_t)(8 * g) * a.Cout + nt * 32;
                        if (a.debug & 4) {} else if (a.io16) st4(static_cast<bf16s*>(a.out) + o, v);
                        else st4(static_cast<float*>(a.out) + o, v);
                        ssum4[nt].x += v.x; ssum4[nt].y += v.y; ssum4[nt].z += v.z; ssum4[nt].w += v.w;
                        ssq4[nt].x += v.x * v.x; ssq4[nt].y += v.y * v.y; ssq4[nt].z += v.z * v.z; ssq4[nt].w += v.w * v.w;
                    }
                }
            }
        }
        // back to one channel per lane: sum the quad's four pixels, lane q keeps channel q
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
    } else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int oh = r0 + wave * MT + mt;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ow = c0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (oh < a.Ho && ow < a.Wo) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = nt * 32 + l31;
                    if (co < a.Cout) {
                        const float v = acc[mt][nt][i];
                        const size_t o = (((size_t)n * a.Ho + oh) * a.Wo + ow) * a.Cout + co;
                        if (a.io16) st1(static_cast<bf16s*>(a.out) + o, v);
                        else static_cast<float*>(a.out)[o] = v;
                        ssum[nt] += v;
                        ssq[nt] += v * v;
                    }
                }
            }
        }
    }
    if (a.stat) {
        rd_sync();
        float* red = s_red;  // [4][2][BN]
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
            const float s = red[(0 * 2 + which) * BN + j] + red[(1 * 2 + which) * BN + j] + red[(2 * 2 + which) * BN + j] +
                            red[(3 * 2 + which) * BN + j];
            if (j < a.Cout) a.stat[((size_t)bid * 2 + which) * a.Cout + j] = s;
        }
    }
    }   // tiles of this workgroup
}

// wgrad: D[k][co] += sum_pixels patch(k, pixel) * dout[pixel][co];  MTK = ceil(K/32) row tiles
template <int MTK, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void stem_wgrad_kernel(const StemArgs a, float* __restrict__ slabs) {
    constexpr int TH = 4, PH = 2 * TH + 5, BN = NT * 32, NPIX = TH * ST_TW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int K = 49 * a.Cin;
    const int PLANE = PH * ST_PW;
    float* s_do = smem;                                 // [NPIX][BN]
    float* s_patch = s_do + (size_t)NPIX * BN;          // [Cin][PH][PW]

    int koff[MTK];
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt) {
        const int k = mt * 32 + l31;
        int off = 0;
        if (k < K) {
            const int t = k / a.Cin, ci = k - t * a.Cin;
            off = ci * PLANE + (t / 7) * ST_PW + (t % 7);
        }
        koff[mt] = off;
    }
    f32x16 acc[MTK][NT];
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int split = blockIdx.x;
    const int tile_begin = split * a.tiles_per_split;
    const int tile_end = min(tile_begin + a.tiles_per_split, a.total_tiles);
    // The input patch of the NEXT tile is fetched into registers before the pixel walk of the current one (same scheme as the
    // forward kernel: slot u of a thread is patch pixel tid + 256 u of every plane, (row, column) computed once, one buffer
    // descriptor per plane with out-of-range -> 0); the dout tile is staged per tile as before (16-byte loads, shifts only).
    constexpr int UPW = (PH * ST_PW + 255) / 256;          // 4 patch pixels per thread and plane
    int pyx[UPW];                                          // py << 16 | px, py = 30000 for slots past the plane
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int e = tid + u * 256;
        const int py = e / ST_PW, px = e - py * ST_PW;
        pyx[u] = ((e < PLANE ? py : 30000) << 16) | px;
    }
    auto fetch_patch = [&](int tile_, float (&v)[3][UPW]) {
        const int n_ = tile_ / (a.tiles_h * a.tiles_w);
        const int trem_ = tile_ - n_ * (a.tiles_h * a.tiles_w);
        const int ih0_ = 2 * (trem_ / a.tiles_w) * TH - 3, iw0_ = 2 * (trem_ % a.tiles_w) * ST_TW - 3;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            if (ci < a.Cin) {
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(a.plane[ci] + (size_t)n_ * a.stride[ci]), 0, (unsigned)(a.H * a.W) * 4u, 0x00020000);
#pragma unroll
                for (int u = 0; u < UPW; ++u) {
                    const int ih = ih0_ + (pyx[u] >> 16), iw = iw0_ + (pyx[u] & 0xffff);
                    const unsigned off = (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? (unsigned)(ih * a.W + iw) * 4u : 0x80000000u;
                    v[ci][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
                }
            }
        }
    };
    float vnext[3][UPW];
    if (tile_begin < tile_end) fetch_patch(tile_begin, vnext);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int n = tile / (a.tiles_h * a.tiles_w);
        const int trem = tile - n * (a.tiles_h * a.tiles_w);
        const int r0 = (trem / a.tiles_w) * TH, c0 = (trem % a.tiles_w) * ST_TW;
        rd_sync();
        constexpr int U = 8;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
            if (ci < a.Cin) {
#pragma unroll
                for (int u = 0; u < UPW; ++u)
                    if (tid + u * 256 < PLANE) s_patch[ci * PLANE + tid + u * 256] = vnext[ci][u];
            }
        for (int base = tid; base < NPIX * (BN / 4) && !((a.debug & 1) && tile > tile_begin); base += 256 * U) {   // (ablation: 1 = stage dout once)
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * 256;
                const int p = e / (BN / 4), j = (e - p * (BN / 4)) * 4;
                const int oh = r0 + (p >> 5), ow = c0 + (p & 31);
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < NPIX * (BN / 4) && oh < a.Ho && ow < a.Wo && j < a.Cout)
                    v[u] = a.io16 ? ld4(static_cast<const bf16s*>(a.dout) + (((size_t)n * a.Ho + oh) * a.Wo + ow) * a.Cout + j)
                                  : ld4(static_cast<const float*>(a.dout) + (((size_t)n * a.Ho + oh) * a.Wo + ow) * a.Cout + j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * 256;
                if (e < NPIX * (BN / 4)) *reinterpret_cast<float4*>(s_do + (size_t)e * 4) = v[u];
            }
        }
        rd_sync();
        if (tile + 1 < tile_end) fetch_patch(tile + 1, vnext);          // in flight during the pixel walk
        // pixel walk (two pixels per MFMA), pipelined like the forward K walk; NPIX/2/4 = 16 steps per wave (even)
        if (!(a.debug & 2)) {
            float a0[MTK], b0[NT], a1[MTK], b1[NT];
#define RD_SW_LOAD(AV, BV, Q)                                                                    \
            {                                                                                        \
                const int p_ = 2 * (Q) + hh;                                                         \
                const int aoff_ = (2 * (p_ >> 5)) * ST_PW + 2 * (p_ & 31);                           \
                _Pragma("unroll") for (int mt = 0; mt < MTK; ++mt) AV[mt] = s_patch[aoff_ + koff[mt]]; \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) BV[nt] = s_do[(size_t)p_ * BN + nt * 32 + l31]; \
            }
#define RD_SW_MFMA(AV, BV)                                                                       \
            _Pragma("unroll") for (int mt = 0; mt < MTK; ++mt)                                       \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[mt], BV[nt], acc[mt][nt], 0, 0, 0);
            RD_SW_LOAD(a0, b0, wave)
            for (int q = wave; q < NPIX / 2; q += 8) {
                RD_SW_LOAD(a1, b1, q + 4)
                __builtin_amdgcn_sched_barrier(0);
                RD_SW_MFMA(a0, b0)
                __builtin_amdgcn_sched_barrier(0);
                RD_SW_LOAD(a0, b0, (q + 8 < NPIX / 2 ? q + 8 : wave))
                __builtin_amdgcn_sched_barrier(0);
                RD_SW_MFMA(a1, b1)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef RD_SW_LOAD
#undef RD_SW_MFMA
        }
    }
    // combine the four waves through LDS, one accumulator tile at a time; slab layout [K][Cout]
    float* slab = slabs + (size_t)split * K * a.Cout;
    float* red = smem;  // [4][16][64]
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            rd_sync();
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[mt][nt][i];
            rd_sync();
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = ii * 4 + wave;
                const float v = red[(0 * 16 + i) * 64 + lane] + red[(1 * 16 + i) * 64 + lane] + red[(2 * 16 + i) * 64 + lane] +
                                red[(3 * 16 + i) * 64 + lane];
                const int k = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const int co = nt * 32 + l31;
                if (k < K && co < a.Cout) slab[(size_t)k * a.Cout + co] = v;
            }
        }
}

// dx[n,h,w] = sum_{co,kh,kw} dout[n,(h+3-kh)/2,(w+3-kw)/2,co] * w[(kh*7+kw)][ci][co]  over taps of matching parity
template <typename T>
__global__ __launch_bounds__(256) void stem_dgrad_channel_kernel(const T* __restrict__ dout, const float* __restrict__ wp,
                                                                 int N, int H, int W, int Ho, int Wo, int Cin, int ci, int Cout,
                                                                 float* __restrict__ dx) {
    __shared__ float s_w[49 * 64];
    for (int e = threadIdx.x; e < 49 * Cout; e += blockDim.x) {
        const int t = e / Cout, co = e - t * Cout;
        s_w[e] = wp[((size_t)t * Cin + ci) * Cout + co];
    }
    rd_sync();
    const int64_t total = (int64_t)N * H * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int w, h, n;
        if (total < (1ll << 31)) {          // 32-bit divisions (a 64-bit one is ~100 instructions)
            const unsigned u = (unsigned)e, r = u / (unsigned)W;
            w = (int)(u - r * (unsigned)W);
            n = (int)(r / (unsigned)H);
            h = (int)(r - (unsigned)n * (unsigned)H);
        } else {
            w = (int)(e % W);
            const int64_t r = e / W;
            h = (int)(r % H);
            n = (int)(r / H);
        }
        float s = 0.f;
        for (int kh = (h + 3) & 1; kh < 7; kh += 2) {
            const int oh = (h + 3 - kh) >> 1;
            if (oh < 0 || oh >= Ho) continue;
            for (int kw = (w + 3) & 1; kw < 7; kw += 2) {
                const int ow = (w + 3 - kw) >> 1;
                if (ow < 0 || ow >= Wo) continue;
                const T* d = dout + (((size_t)n * Ho + oh) * Wo + ow) * Cout;
                const float* wv = s_w + (kh * 7 + kw) * Cout;
                for (int co = 0; co < Cout; co += 4) {
                    const float4 dv = ld4(d + co);
                    s = fmaf(dv.x, wv[co], s); s = fmaf(dv.y, wv[co + 1], s);
                    s = fmaf(dv.z, wv[co + 2], s); s = fmaf(dv.w, wv[co + 3], s);
                }
            }
        }
        dx[e] = s;
    }
}

// Tiled form for the 16-channel depth stem (round 4; the only caller: stage 2's dense-depth channel, once per multistage step, alone on the
// chain between the two stages' backward passes -- 175 us for 46 MB with the kernel above, whose every pixel walks up to 16 x 16 global
// loads).  A workgroup stages the 12 x 20 pixel patch of dout behind a 16 x 32 pixel tile of dx in LDS; wave = parity class (h & 1, w & 1) of
// the pixels it computes, so its taps (3 or 4 kernel rows x 3 or 4 columns) are the same for all of them and its weights live in registers;
// lane = (pixel column, channel quad), summed over the quads with two DPP exchanges.
template <typename T>
__global__ __launch_bounds__(256) void stem_dgrad_channel16_kernel(const T* __restrict__ dout, const float* __restrict__ wp, int N, int H, int W,
                                                                   int Ho, int Wo, int Cin, int ci, int tiles_h, int tiles_w,
                                                                   float* __restrict__ dx) {
    constexpr int Cout = 16, TH = 16, TW = 32, PR = 12, PC = 20;
    __shared__ float4 s_d[PR * PC * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane & 3, pl = lane >> 2;
    const int per_img = tiles_h * tiles_w;
    const int n = blockIdx.x / per_img, trem = blockIdx.x - n * per_img;
    const int h0 = (trem / tiles_w) * TH, w0 = (trem % tiles_w) * TW;
    const int oh0 = (h0 - 3) >> 1, ow0 = (w0 - 3) >> 1;          // first dout row / column any pixel of the tile reads (may be negative)
    const T* src = dout + (size_t)n * Ho * Wo * Cout;
    for (int e = tid; e < PR * PC * 4; e += 256) {
        const int px = e >> 2, qq = e & 3;
        const int r = px / PC, c = px - r * PC;
        const int oh = oh0 + r, ow = ow0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oh >= 0 && oh < Ho && ow >= 0 && ow < Wo) v = ld4(src + ((size_t)oh * Wo + ow) * Cout + 4 * qq);
        s_d[e] = v;
    }
    const int ph = wave >> 1, pw = wave & 1;
    const int kh0 = (ph + 3) & 1, kw0 = (pw + 3) & 1;             // taps of matching parity: kh0, kh0 + 2, ... < 7
    float4 wr[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kh = kh0 + 2 * i, kw = kw0 + 2 * j;
            wr[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kh < 7 && kw < 7) wr[i][j] = *reinterpret_cast<const float4*>(wp + ((size_t)(kh * 7 + kw) * Cin + ci) * Cout + 4 * q);
        }
    rd_sync();
    const int w = w0 + 2 * pl + pw;
#pragma unroll
    for (int a = 0; a < TH / 2; ++a) {
        const int h = h0 + 2 * a + ph;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ((h + 3 - (kh0 + 2 * i)) >> 1) - oh0;      // (taps with kh >= 7 have zero weights; their row index is clamped)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = ((w + 3 - (kw0 + 2 * j)) >> 1) - ow0;
                const float4 dv = s_d[((r < 0 ? 0 : r) * PC + (c < 0 ? 0 : c)) * 4 + q];
                const float4 wt = wr[i][j];
                s = fmaf(dv.x, wt.x, s); s = fmaf(dv.y, wt.y, s); s = fmaf(dv.z, wt.z, s); s = fmaf(dv.w, wt.w, s);
            }
        }
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        if (q == 0 && h < H && w < W) dx[((size_t)n * H + h) * W + w] = s;
    }
}

static int stem_fill(StemArgs& a, const float* const* planes, const int64_t* strides, int Cin, int N, int H, int W, int Cout) {
    RD_CHECK_ARG(planes && strides && Cin >= 1 && Cin <= 3 && N > 0 && H > 6 && W > 6, "stem: bad arguments");
    RD_CHECK_ARG(Cout == 64 || Cout == 16 || Cout == 32, "stem: Cout=%d unsupported", Cout);
    for (int i = 0; i < 3; ++i) {
        a.plane[i] = i < Cin ? planes[i] : nullptr;
        a.stride[i] = i < Cin ? strides[i] : 0;
        RD_CHECK_ARG(i >= Cin || planes[i], "stem: null plane %d", i);
    }
    a.Cin = Cin; a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    a.Ho = (H + 6 - 7) / 2 + 1;
    a.Wo = (W + 6 - 7) / 2 + 1;
    return RD_OK;
}

static int stem_wgrad_splits(int total_tiles) {
    static const char* spc = getenv("RD_STEM_WGRAD_SPLITS_PER_CU");      // diagnostics
    int want = (spc ? atoi(spc) : 2) * num_cus();
    return want < total_tiles ? want : total_tiles;
}

}  // namespace rd
using namespace rd;

extern "C" int rd_stem_stat_tiles(int32_t N, int32_t H, int32_t W) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    return N * cdiv(Ho, 8) * cdiv(Wo, ST_TW);
}

static int stem_fwd_impl(int io16, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                         const float* w_packed, int32_t Cout, void* out, float* stat_partial, void* stream) {
    StemArgs a;
    a.io16 = io16;
    int rc = stem_fill(a, planes, strides, Cin, N, H, W, Cout);
    if (rc != RD_OK) return rc;
    RD_CHECK_ARG(w_packed && out, "stem_fwd: null tensor");
    if (stem16_eligible(Cin, Cout))       // the 16-channel depth stem: its own 16-wide kernel (same tiles, same statistics layout)
        return launch_stem16_fwd(io16, planes, strides, Cin, N, H, W, w_packed, Cout, out, stat_partial, static_cast<hipStream_t>(stream));
    a.w = w_packed; a.out = out; a.stat = stat_partial; a.dout = nullptr;
    { static const char* dbg = getenv("RD_STEM_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
    a.tiles_h = cdiv(a.Ho, 8); a.tiles_w = cdiv(a.Wo, ST_TW);
    const int total = N * a.tiles_h * a.tiles_w;
    // persistent: two workgroups per CU walk the tiles.  (More resident workgroups do not help the latency-bound one-plane depth stem:
    // at 164 VGPRs only two fit a SIMD -- RD_STEM_WG_PER_CU = 3 / 4 / 6 / 8 measured 85 / 79 / 80 / 81 us against 77 us, round 3.)
    static const char* wpc_env = getenv("RD_STEM_WG_PER_CU");
    const int wpc = wpc_env ? atoi(wpc_env) : 2;
    const int grid = total < wpc * num_cus() ? total : wpc * num_cus();
    const int NT = Cout > 32 ? 2 : 1, BN = NT * 32;
    const int Kp = (49 * Cin + 1) & ~1;
    const int Kq = Kp + 4;
    const size_t lds = ((size_t)((Kq + 3) & ~3) + (size_t)Kq * BN + (size_t)Cin * 21 * ST_PW + (size_t)8 * BN) * 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static std::atomic<unsigned long long> attr{0}, attr1{0};
    RD_SET_ATTR_ONCE(attr, hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    RD_SET_ATTR_ONCE(attr1, hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    if (NT == 2) hipLaunchKernelGGL(stem_fwd_kernel<2>, dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(stem_fwd_kernel<1>, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("stem_fwd_kernel");
    return RD_OK;
}

extern "C" int rd_stem_fwd(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                           const float* w_packed, int32_t Cout, float* out, float* stat_partial, void* stream) {
    return stem_fwd_impl(0, planes, strides, Cin, N, H, W, w_packed, Cout, out, stat_partial, stream);
}
// storage-typed forms: dtype selects the element type of the NHWC output / output-gradient tensor (the input planes stay fp32)
extern "C" int rd_stem_fwd_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                             int32_t W, const float* w_packed, int32_t Cout, void* out, float* stat_partial, void* stream) {
    RD_CHECK_ARG(dtype == RD_DTYPE_F32 || dtype == RD_DTYPE_BF16, "stem_fwd_t: bad dtype %d", dtype);
    return stem_fwd_impl(dtype == RD_DTYPE_BF16, planes, strides, Cin, N, H, W, w_packed, Cout, out, stat_partial, stream);
}

extern "C" int64_t rd_stem_wgrad_workspace_floats(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int total = N * cdiv(Ho, 4) * cdiv(Wo, ST_TW);
    const int splits = stem_wgrad_splits(total);
    const int tps = cdiv(total, splits);
    const int n_splits = cdiv(total, tps);
    const int J = n_splits < 16 ? n_splits : 16;
    return (int64_t)(n_splits + J) * 49 * Cin * Cout;
}

static int stem_wgrad_impl(int io16, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                           const void* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream) {
    StemArgs a;
    a.io16 = io16;
    int rc = stem_fill(a, planes, strides, Cin, N, H, W, Cout);
    if (rc != RD_OK) return rc;
    RD_CHECK_ARG(dout && grad_oihw && ws, "stem_wgrad: null tensor");
    a.w = nullptr; a.out = nullptr; a.stat = nullptr; a.dout = dout;
    { static const char* dbg = getenv("RD_STEM_WGRAD_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }     // ablation bits (tools/bench_stem.py)
    a.tiles_h = cdiv(a.Ho, 4); a.tiles_w = cdiv(a.Wo, ST_TW);
    a.total_tiles = N * a.tiles_h * a.tiles_w;
    const int splits = stem_wgrad_splits(a.total_tiles);
    a.tiles_per_split = cdiv(a.total_tiles, splits);
    const int n_splits = cdiv(a.total_tiles, a.tiles_per_split);
    const int K = 49 * Cin, MTK = cdiv(K, 32), NT = Cout > 32 ? 2 : 1, BN = NT * 32;
    size_t lds = ((size_t)4 * 32 * BN + (size_t)Cin * 13 * ST_PW) * 4;
    if (lds < 4 * 16 * 64 * 4) lds = 4 * 16 * 64 * 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RD_SW(M_, N_)                                                                             \
    if (MTK == M_ && NT == N_) {                                                                  \
        hipLaunchKernelGGL((stem_wgrad_kernel<M_, N_>), dim3(n_splits), dim3(256), lds, s, a, ws); \
        RD_CHECK_LAUNCH("stem_wgrad_kernel");                                                     \
    } else
    RD_SW(5, 2) RD_SW(2, 1) RD_SW(4, 1) {
        set_error("stem_wgrad: unsupported shape Cin=%d Cout=%d", Cin, Cout);
        return RD_EINVAL;
    }
#undef RD_SW
    const int64_t E = (int64_t)K * Cout;
    return launch_slab_reduce(ws, n_splits, E, ws + (int64_t)n_splits * E, grad_oihw, 49, Cin, Cout, Cout, Cin, 0, 0, s);
}

extern "C" int rd_stem_wgrad(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                             const float* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream) {
    return stem_wgrad_impl(0, planes, strides, Cin, N, H, W, dout, Cout, grad_oihw, ws, stream);
}
extern "C" int rd_stem_wgrad_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                               int32_t W, const void* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream) {
    RD_CHECK_ARG(dtype == RD_DTYPE_F32 || dtype == RD_DTYPE_BF16, "stem_wgrad_t: bad dtype %d", dtype);
    return stem_wgrad_impl(dtype == RD_DTYPE_BF16, planes, strides, Cin, N, H, W, dout, Cout, grad_oihw, ws, stream);
}

template <typename T>
static int stem_dgrad_channel_T(const T* dout, const float* w_packed, int32_t N, int32_t H, int32_t W, int32_t Cin,
                                int32_t ci, int32_t Cout, float* dx, void* stream) {
    RD_CHECK_ARG(dout && w_packed && dx && ci >= 0 && ci < Cin && Cout % 4 == 0 && Cout <= 64, "stem_dgrad_channel: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int64_t g = cdiv64((int64_t)N * H * W, 256);
    if (g > (int64_t)num_cus() * 16) g = (int64_t)num_cus() * 16;
    static const bool tiled = !(getenv("RD_STEM_DGRAD_TILED") && atoi(getenv("RD_STEM_DGRAD_TILED")) == 0);
    const int tiles_h = cdiv(H, 16), tiles_w = cdiv(W, 32);
    if (tiled && Cout == 16 && (int64_t)N * tiles_h * tiles_w < (1ll << 31)) {
        hipLaunchKernelGGL(stem_dgrad_channel16_kernel<T>, dim3(N * tiles_h * tiles_w), dim3(256), 0, static_cast<hipStream_t>(stream), dout,
                           w_packed, N, H, W, Ho, Wo, Cin, ci, tiles_h, tiles_w, dx);
        RD_CHECK_LAUNCH("stem_dgrad_channel16_kernel");
        return RD_OK;
    }
    hipLaunchKernelGGL(stem_dgrad_channel_kernel<T>, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), dout, w_packed, N,
                       H, W, Ho, Wo, Cin, ci, Cout, dx);
    RD_CHECK_LAUNCH("stem_dgrad_channel_kernel");
    return RD_OK;
}
extern "C" int rd_stem_dgrad_channel(const float* dout, const float* w_packed, int32_t N, int32_t H, int32_t W, int32_t Cin,
                                     int32_t ci, int32_t Cout, float* dx, void* stream) {
    return stem_dgrad_channel_T<float>(dout, w_packed, N, H, W, Cin, ci, Cout, dx, stream);
}
extern "C" int rd_stem_dgrad_channel_t(int32_t dtype, const void* dout, const float* w_packed, int32_t N, int32_t H, int32_t W,
                                       int32_t Cin, int32_t ci, int32_t Cout, float* dx, void* stream) {
    if (dtype == RD_DTYPE_F32) return stem_dgrad_channel_T<float>(static_cast<const float*>(dout), w_packed, N, H, W, Cin, ci, Cout, dx, stream);
    if (dtype == RD_DTYPE_BF16) return stem_dgrad_channel_T<bf16s>(static_cast<const bf16s*>(dout), w_packed, N, H, W, Cin, ci, Cout, dx, stream);
    rd::set_error("stem_dgrad_channel_t: bad dtype %d", dtype);
    return RD_EINVAL;
}
