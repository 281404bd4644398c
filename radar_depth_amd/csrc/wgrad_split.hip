// fp32 weight gradient of the 3x3 / stride-1 convolutions and of the UpProj 5x5 (its four parity phases: 3x3 / 2x3 / 3x2 / 2x2 sub-stencils
// against the output gradient sampled at stride 2, one launch each) on the bf16 matrix cores: both operands of
//     dW[tap][ci][co] = sum over output pixels p of  x[p + tap][ci] * dy[p][co]
// are activations, so both are split into three bf16 pieces while they are staged (x = x0 + x1 + x2 exactly) and every product is
// rebuilt from the six bf16 MFMAs whose terms are not below 2^-24 of it, with fp32 accumulation -- gconv_split.hip's arithmetic
// (error analysis there), applied to the GEMM whose reduction index is the PIXEL.
//
// The bf16 MFMA wants eight consecutive reduction elements (pixels) per lane for a fixed row (channel), i.e. channel-major
// operands, while the tensors are NHWC.  wgrad_bf16.hip transposes in its staging waves (pixel pairs packed per dword, v_alignbyte
// for the horizontal taps).  Here the LDS images stay PIXEL-major -- [piece][32-channel tile][pixel][32 channels], 64 bytes per
// pixel -- and the fragments are read with ds_read_b64_tr_b16: the 16 lanes of a group pass the addresses of four pixel rows x
// four 8-byte chunks and each lane receives ONE channel of those four pixels (tools/micro/tr_b16.hip prints the mapping).  A
// tap is then nothing but an immediate offset of (dh * 34 + dw) pixels: no alignbyte, no address arithmetic in the walk.  A
// 32-lane read pass covers 4 consecutive pixels x 64 bytes = 256 contiguous bytes: all 64 banks once.
//
//   workgroup : 8 waves, one per CU.  Waves 0-3: one 32 x 32 (ci, co) pair each of a 64 x 64 block of dW, all nine taps: nine
//               accumulators = 144 registers, and nothing but fragment reads and MFMAs.  Waves 4-7 stage: 8-channel units of the
//               next pixel tile through registers (split into pieces there), two LDS buffers, ONE barrier per tile.
//   pixel tile: 64 pixels of one image, 2 rows x 32 columns (x patch 4 x 34 with the halo) or 4 x 16 (6 x 18), whichever pads the
//               grid less; four 16-pixel reduction steps.
//   split-K   : contiguous ranges of pixel tiles per workgroup, per-split slabs [split][tap][Cin][Cout] like rd_wgrad, reduced in
//               a fixed order by the same slab reduction.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <unordered_map>

#include "common.h"

namespace rd {

int launch_slab_reduce(const float* slabs, int n_splits, int64_t E, float* tmp, float* grad_oihw, int S, int Cin, int Cout,
                       int O, int I, int co_off, int accumulate, hipStream_t s);   // wgrad.hip

typedef __bf16 wsbf16x8 __attribute__((ext_vector_type(8)));
typedef short wss16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wsu32x4 __attribute__((ext_vector_type(4)));
typedef float wsf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wsbf16x2 __attribute__((ext_vector_type(2)));

constexpr unsigned WS_OOB = 0x80000000u;
// pixel-tile geometries: 64 output pixels as R rows x TW columns (TW a multiple of the 16-pixel reduction step)
template <int R_, int TW_>
struct WsGeo {
    static constexpr int R = R_, TW = TW_;
    static constexpr int XW = TW + 2, XH = R + 2;                // x patch with the 3x3 halo
    static constexpr int XPIX = XW * XH, YPIX = R * TW;
    static constexpr int XPLANE = ((XPIX * 64 + 255) / 256) * 256;      // bytes of one [pixel][32 channels] bf16 plane of the patch (multiple of 256)
    static constexpr int YPLANE = YPIX * 64;
    static constexpr int XBYTES = 3 * 2 * XPLANE;                // [piece][channel tile][plane]
    static constexpr int YBYTES = 3 * 2 * YPLANE;
    static constexpr int BUF = XBYTES + YBYTES;                  // bytes per buffer, two of them
    static constexpr int XUNITS = XPIX * 8;                      // 8-channel units of a 64-channel patch ...
    static constexpr int XUNITS_PAD = ((XUNITS + 63) / 64) * 64; // ... rounded up to whole waves: a wave's unit is entirely x or entirely dy
    static constexpr int UNITS = XUNITS_PAD + YPIX * 8;
    static constexpr int UPT = (UNITS + 255) / 256;              // units per staging thread
    static_assert(YPIX == 64 && TW % 16 == 0, "a tile is 64 pixels in 16-pixel reduction steps");
    // PRE (operands split by their producers, staged by global_load_lds): [piece][16-channel block (4)][pixel][16 channels], 32 bytes
    // per pixel; a block plane is = 128 mod 256 bytes so that the two halves of a transposing read pass (channels 0-15 / 16-31 of a
    // 32-channel tile = two neighbouring block planes) fall into different halves of the 64 banks
    static constexpr int XP16 = ((XPIX * 32 + 127) / 256) * 256 + 128;
    static constexpr int YP16 = ((YPIX * 32 + 127) / 256) * 256 + 128;
    static constexpr int XBYTES_P = 3 * 4 * XP16;
    static constexpr int YBYTES_P = 3 * 4 * YP16;
    static constexpr int BUF_P = XBYTES_P + YBYTES_P;
    static constexpr int XSEG = (XW + 31) / 32, YSEG = (TW + 31) / 32;      // 64-lane copies (32 pixels x two 16-byte halves) per row
    static constexpr int XU_P = 3 * 4 * XH * XSEG;               // copies per tile: (piece, block, patch row, segment)
    static constexpr int YU_P = 3 * 4 * R * YSEG;
};
typedef WsGeo<2, 32> WsWide;      // 2 x 32: 76800 bytes per buffer, 7 units per thread
typedef WsGeo<4, 16> WsTall;      // 4 x 16: less halo (1.69 vs 2.1 patch pixels per output pixel), fits 100- and 200-column grids better

struct WsArgs {
    const float* x;
    const float* dy;
    float* slabs;
    int N, Hi, Wi, Cin, ldi, Ho, Wo, Cout, ldo;
    int tiles_h, tiles_w, total_tiles, tiles_per_split, n_splits, n_cib, n_cob;
    int dh0, dw0;               // offset of tap (0, 0): x pixel = logical pixel + (dh0 + i, dw0 + j)
    int OS, off_h, off_w;       // dy pixel = OS * logical pixel + (off_h, off_w)
    int lh, lw;                 // logical grid (= the x grid for these descriptors)
    int S;                      // weight slabs per split
    int widx[9];                // slab of tap (i, j) = widx[i * TC + j]
    // PRE: piece planes [piece][C/16][pixels][16] bf16 of x and dy (rd_split_pieces / the producers' epilogues)
    const unsigned short* xp;
    const unsigned short* yp;
    long long xplane, yplane;   // bytes per piece plane
    long long xm, ym;           // pixels per 16-channel block of a plane (N*Hi*Wi, N*Ho*Wo)
    int dbg;                    // diagnostics (RD_WGRAD_SPLIT_DEBUG; results are garbage): 1 staging without the split arithmetic, 2 no LDS stores
                                // in the staging waves, 4 no global loads in the staging waves
};

__device__ __forceinline__ unsigned ws_cvt_pk(float a, float b) {
    wsf32x2 v;
    v[0] = a; v[1] = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wsbf16x2));
}
// three bf16 pieces of eight fp32 values, a pair at a time (see gconv_split.hip)
__device__ __forceinline__ void ws_split8(const float4 v0, const float4 v1, wsu32x4& w0, wsu32x4& w1, wsu32x4& w2) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = x[2 * i], b = x[2 * i + 1];
        const unsigned u0 = ws_cvt_pk(a, b);
        a -= __uint_as_float(u0 << 16);
        b -= __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = ws_cvt_pk(a, b);
        a -= __uint_as_float(u1 << 16);
        b -= __uint_as_float(u1 & 0xffff0000u);
        w0[i] = u0;
        w1[i] = u1;
        w2[i] = ws_cvt_pk(a, b);
    }
}

// eight consecutive pixels of one channel: two transposing reads of four pixels each
template <int PITCH = 64>
__device__ __forceinline__ wsbf16x8 ws_frag(unsigned base, int off) {
    typedef __attribute__((address_space(3))) wss16x4* lp;
    const wss16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lp>(base + off));
    const wss16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lp>(base + off + 4 * PITCH));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(wsbf16x8, r);
}

template <int TR, int TC, typename G, bool PRE>
__global__ __launch_bounds__(512) void wgrad_split_kernel(const WsArgs a) {
    constexpr int NT = TR * TC;
    constexpr int BUFB = PRE ? G::BUF_P : G::BUF;          // bytes per LDS buffer
    constexpr int PIT = PRE ? 32 : 64;                     // bytes per pixel of an LDS plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const unsigned lds0 = (unsigned)(size_t)smem;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int nblk = a.n_cib * a.n_cob;
    const int blk = vid % nblk, split = vid / nblk;
    const int cib0 = (blk / a.n_cob) * 64, cob0 = (blk % a.n_cob) * 64;
    const int tile_begin = split * a.tiles_per_split;
    const int tile_end = min(tile_begin + a.tiles_per_split, a.total_tiles);
    const int ntiles = tile_end - tile_begin;

    f32x16 acc[NT];      // (zeroed in the compute branch only: live registers of the staging waves otherwise)

    if (loader && PRE) {
        // ------------------------------------------------------------------------------------------ staging waves, pre-split operands:
        // one global_load_lds per (piece, 16-channel block, patch row): lane l = (pixel l >> 1, 16-byte half l & 1) of the row, which is
        // both contiguous in the plane and lane-linear in the LDS image; lanes outside the image / the logical grid / the channel
        // range store zeros instead (the buffers are reused from tile to tile, every slot is written for every tile)
        const int lw = wave - 4;
        const char* xsrc = reinterpret_cast<const char*>(a.xp);
        const char* ysrc = reinterpret_cast<const char*>(a.yp);
        const int pxl = lane >> 1, hf = lane & 1;
        auto stage = [&](int tile, int buf) {
            const int n = tile / (a.tiles_h * a.tiles_w), tr = tile - n * (a.tiles_h * a.tiles_w);
            const int r0 = (tr / a.tiles_w) * G::R, c0 = (tr % a.tiles_w) * G::TW;
            const unsigned base = lds0 + buf * BUFB;
            for (int u = lw; u < G::XU_P + G::YU_P; u += 4) {
                if (u < G::XU_P) {
                    const int sg = u % G::XSEG, u1 = u / G::XSEG;
                    const int row = u1 % G::XH, t = u1 / G::XH, blk = t & 3, p = t >> 2;
                    const int px = sg * 32 + pxl;
                    const int ih = r0 + a.dh0 + row, iw = c0 + a.dw0 + px, ch = cib0 + blk * 16;
                    const bool rok = ih >= 0 && ih < a.Hi && ch < a.Cin;                    // wave-uniform
                    const bool ok = rok && px < G::XW && iw >= 0 && iw < a.Wi;
                    const unsigned dst = base + (p * 4 + blk) * G::XP16 + (row * G::XW + sg * 32) * 32;   // wave base: lane l lands at + 16 l
                    if (ok) glds16(reinterpret_cast<const float*>(xsrc + p * a.xplane + (((size_t)(ch >> 4) * a.xm + ((size_t)n * a.Hi + ih) * a.Wi + iw) * 32 + hf * 16)),
                                   reinterpret_cast<float*>((size_t)dst));
                    else if (px < G::XW) asm volatile("ds_write_b128 %0, %1" ::"v"(dst + lane * 16), "v"(wsu32x4{0u, 0u, 0u, 0u}) : "memory");
                } else {
                    const int v = u - G::XU_P;
                    const int sg = v % G::YSEG, v1 = v / G::YSEG;
                    const int row = v1 % G::R, t = v1 / G::R, blk = t & 3, p = t >> 2;
                    const int px = sg * 32 + pxl;
                    const int lr = r0 + row, lc = c0 + px, ch = cob0 + blk * 16;
                    const int oh = a.OS * lr + a.off_h, ow = a.OS * lc + a.off_w;
                    const bool ok = lr < a.lh && ch < a.Cout && px < G::TW && lc < a.lw;
                    const unsigned dst = base + G::XBYTES_P + (p * 4 + blk) * G::YP16 + (row * G::TW + sg * 32) * 32;
                    if (ok) glds16(reinterpret_cast<const float*>(ysrc + p * a.yplane + (((size_t)(ch >> 4) * a.ym + ((size_t)n * a.Ho + oh) * a.Wo + ow) * 32 + hf * 16)),
                                   reinterpret_cast<float*>((size_t)dst));
                    else if (px < G::TW) asm volatile("ds_write_b128 %0, %1" ::"v"(dst + lane * 16), "v"(wsu32x4{0u, 0u, 0u, 0u}) : "memory");
                }
            }
        };
        // tile i lives in buffer i & 1; iteration i: barrier B(i) (tile i published, buffer (i + 1) & 1 free), copy tile i + 1, wait for it
        if (ntiles > 0) stage(tile_begin, 0);
        glds_wait();
        for (int i = 0; i < ntiles; ++i) {
            rd_sync();                            // B(i)
            if (i + 1 < ntiles) stage(tile_begin + i + 1, (i + 1) & 1);
            glds_wait();
        }
        rd_sync();                                // matches the compute waves' final barrier
    } else if (loader) {
        // ------------------------------------------------------------------------------------------ staging waves
        const int ltid = tid - 256;
        // this thread's units: e = ltid + 256 u; e < 1088: x patch unit (tile t, pixel, quad q), else dy unit.  1088 = 17 * 64, so a
        // wave's unit u is entirely x or entirely dy (wave-uniform buffer resource)
        int upr[G::UPT], upc[G::UPT], uch[G::UPT], udst[G::UPT];
#pragma unroll
        for (int u = 0; u < G::UPT; ++u) {
            const int e = ltid + 256 * u;
            if (e < G::XUNITS) {
                const int t = e / (G::XPIX * 4), rem = e - t * (G::XPIX * 4);
                const int px = rem >> 2, q = rem & 3;
                upr[u] = px / G::XW; upc[u] = px - upr[u] * G::XW;
                uch[u] = t * 32 + q * 8;
                udst[u] = t * G::XPLANE + rem * 16;
            } else if (e >= G::XUNITS_PAD && e < G::UNITS) {
                const int e2 = e - G::XUNITS_PAD;
                const int t = e2 >> 8, rem = e2 & 255;
                const int px = rem >> 2, q = rem & 3;
                upr[u] = px / G::TW; upc[u] = px - upr[u] * G::TW;
                uch[u] = t * 32 + q * 8;
                udst[u] = G::XBYTES + t * G::YPLANE + rem * 16;
            } else {
                upr[u] = upc[u] = uch[u] = 0;
                udst[u] = -1;
            }
        }
        const unsigned ximg = (unsigned)(a.Hi * a.Wi * a.ldi) * 4u, yimg = (unsigned)(a.Ho * a.Wo * a.ldo) * 4u;
        // two register sets: the loads of tile i + 2 are issued BEFORE tile i + 1 is split and stored, so they have a whole iteration in
        // flight (with one set they were issued behind the split and the next iteration's split waited for them: the ablation without global
        // loads ran 22 % faster)
        float4 va0[G::UPT], va1[G::UPT], vb0[G::UPT], vb1[G::UPT];
        auto fetch = [&](int tile, float4 (&v0)[G::UPT], float4 (&v1)[G::UPT]) {
            if (a.dbg & 4) return;
            const int n = tile / (a.tiles_h * a.tiles_w), tr = tile - n * (a.tiles_h * a.tiles_w);
            const int r0 = (tr / a.tiles_w) * G::R, c0 = (tr % a.tiles_w) * G::TW;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * a.Hi * a.Wi * a.ldi, 0, ximg, 0x00020000);
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + (size_t)n * a.Ho * a.Wo * a.ldo, 0, yimg, 0x00020000);
#pragma unroll
            for (int u = 0; u < G::UPT; ++u) {
                const bool is_x = __builtin_amdgcn_readfirstlane(ltid + 256 * u) < G::XUNITS_PAD;      // wave-uniform
                unsigned off;
                if (is_x) {
                    const int ih = r0 + a.dh0 + upr[u], iw = c0 + a.dw0 + upc[u], ch = cib0 + uch[u];
                    off = (ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi && ch < a.Cin && udst[u] >= 0) ? (unsigned)(((ih * a.Wi + iw) * a.ldi + ch) * 4) : WS_OOB;
                    v0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
                    v1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off + 16, 0, 0));
                } else {
                    const int lr = r0 + upr[u], lc = c0 + upc[u], ch = cob0 + uch[u];
                    const int oh = a.OS * lr + a.off_h, ow = a.OS * lc + a.off_w;
                    off = (lr < a.lh && lc < a.lw && ch < a.Cout && udst[u] >= 0) ? (unsigned)(((oh * a.Wo + ow) * a.ldo + ch) * 4) : WS_OOB;
                    v0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off, 0, 0));
                    v1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off + 16, 0, 0));
                }
            }
        };
        auto split_put = [&](int buf, float4 (&v0)[G::UPT], float4 (&v1)[G::UPT]) {
            const unsigned base = lds0 + buf * BUFB;
#pragma unroll
            for (int u = 0; u < G::UPT; ++u) {
                wsu32x4 w0, w1, w2;
                if (a.dbg & 1) { w0 = __builtin_bit_cast(wsu32x4, v0[u]); w1 = __builtin_bit_cast(wsu32x4, v1[u]); w2 = w0; }
                else ws_split8(v0[u], v1[u], w0, w1, w2);
                if (udst[u] >= 0 && !(a.dbg & 2)) {
                    const bool is_x = udst[u] < G::XBYTES;
                    const unsigned pstride = is_x ? 2 * G::XPLANE : 2 * G::YPLANE;
                    const unsigned ad = base + udst[u];
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(w0) : "memory");
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ad + pstride), "v"(w1) : "memory");
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ad + 2 * pstride), "v"(w2) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // tile i of this workgroup lives in buffer i & 1.  Iteration i: barrier B(i) (tile i published, buffer (i + 1) & 1 free);
        // split and store tile i + 1 (fetched during iteration i - 1); fetch tile i + 2.
        // (tile j's values live in set A for even j, in set B for odd j)
        if (ntiles > 0) {
            fetch(tile_begin, va0, va1);
            if (ntiles > 1) fetch(tile_begin + 1, vb0, vb1);
            split_put(0, va0, va1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < ntiles; i += 2) {
            rd_sync();                            // B(i), i even: tile i + 1 is in set B, set A is free
            if (i + 2 < ntiles) fetch(tile_begin + i + 2, va0, va1);
            if (i + 1 < ntiles) split_put(1, vb0, vb1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i + 1 >= ntiles) break;
            rd_sync();                            // B(i + 1): tile i + 2 is in set A, set B is free
            if (i + 3 < ntiles) fetch(tile_begin + i + 3, vb0, vb1);
            if (i + 2 < ntiles) split_put(0, va0, va1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        rd_sync();                                // matches the compute waves' final barrier
    } else {
        // ------------------------------------------------------------------------------------------ compute waves
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const int ti = wave >> 1, to = wave & 1;          // 32-channel tile of the block on the ci / co side
        // lane part of every fragment address: pixel row (lane & 15) / 4 of the group's four, 8-byte chunk lane & 3, second 16
        // channels for lanes 16..31 of each half, pixels 8.. for the upper half wave
        // (PRE: 32 bytes per pixel, the second 16 channels live in the next 16-channel block plane)
        const unsigned lrow = (unsigned)(((lane & 15) >> 2) + (lane >> 5) * 8) * PIT + (lane & 3) * 8;
        const unsigned xb = PRE ? lds0 + (2 * ti + ((lane >> 4) & 1)) * G::XP16 + lrow : lds0 + ti * G::XPLANE + lrow + ((lane >> 4) & 1) * 32;
        const unsigned yb = PRE ? lds0 + G::XBYTES_P + (2 * to + ((lane >> 4) & 1)) * G::YP16 + lrow : lds0 + G::XBYTES + to * G::YPLANE + lrow + ((lane >> 4) & 1) * 32;
        constexpr int XPS = PRE ? 4 * G::XP16 : 2 * G::XPLANE;      // bytes between pieces
        constexpr int YPS = PRE ? 4 * G::YP16 : 2 * G::YPLANE;
        for (int i = 0; i < ntiles; ++i) {
            rd_sync();                            // B(i)
            const unsigned xa = xb + (i & 1) * BUFB, ya = yb + (i & 1) * BUFB;
            // 4 NT steps per tile = (row, 16-pixel reduction step, tap); the fragments of step s + 1 are read in front of the MFMAs of
            // step s and the order is pinned (left alone the compiler hoists a whole reduction step's reads and spills them)
            constexpr int NSTEP = G::R * (G::TW / 16) * NT;
            wsbf16x8 A[2][3], B[2][3];
            auto loadA = [&](int s_, wsbf16x8 (&F)[3]) {
                const int rk = s_ / NT, t = s_ % NT, r = rk / (G::TW / 16), ks = rk % (G::TW / 16);
#pragma unroll
                for (int p = 0; p < 3; ++p) F[p] = ws_frag<PIT>(xa, p * XPS + ((r + t / TC) * G::XW + ks * 16 + t % TC) * PIT);
            };
            auto loadB = [&](int rk, wsbf16x8 (&F)[3]) {
                const int r = rk / (G::TW / 16), ks = rk % (G::TW / 16);
#pragma unroll
                for (int p = 0; p < 3; ++p) F[p] = ws_frag<PIT>(ya, p * YPS + (r * G::TW + ks * 16) * PIT);
            };
            loadB(0, B[0]);
            loadA(0, A[0]);
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                const int rk = s_ / NT, t = s_ % NT;
                if (s_ + 1 < NSTEP) {
                    loadA(s_ + 1, A[(s_ + 1) & 1]);
                    if (t == NT - 1) loadB(rk + 1, B[(rk + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 c = acc[t];
                RD_SPLIT_TERMS(c, A[s_ & 1][0], A[s_ & 1][1], A[s_ & 1][2], B[rk & 1][0], B[rk & 1][1], B[rk & 1][2])
                acc[t] = c;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rd_sync();
        // slab [tap][Cin][Cout] of this split (zeros when the split has no tiles: every element of the block is written)
        const int l31 = lane & 31, hh = lane >> 5;
        float* slab = a.slabs + (size_t)split * a.S * a.Cin * a.Cout;
        const int co = cob0 + to * 32 + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float* dst = slab + (size_t)a.widx[t] * a.Cin * a.Cout;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = cib0 + ti * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (ci < a.Cin && co < a.Cout) dst[(size_t)ci * a.Cout + co] = acc[t][i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host
struct WsPlan {
    int ok, tiles_h, tiles_w, total_tiles, tiles_per_split, n_splits, n_cib, n_cob, J, S;
    int tall;       // pixel tile 4 x 16 (WsTall) instead of 2 x 32 (WsWide)
};

// taps of a phase = a full TR x TC rectangle in row-major order, TR, TC in {2, 3}
static bool ws_phase_ok(const RdPhase& p, int& tr, int& tc) {
    tr = p.dh_max - p.dh_min + 1;
    tc = p.dw_max - p.dw_min + 1;
    if (tr < 2 || tr > 3 || tc < 2 || tc > 3 || p.n_taps != tr * tc) return false;
    for (int t = 0; t < p.n_taps; ++t)
        if (p.dh[t] != p.dh_min + t / tc || p.dw[t] != p.dw_min + t % tc || p.widx[t] < 0) return false;
    return true;
}

static bool ws_shape_ok(const RdConvDesc& d) {
    static const char* off = getenv("RD_WGRAD_NO_SPLIT");       // diagnostics: keep every weight gradient on the fp32 kernels
    if (off) return false;
    if (d.n_phases < 1 || d.n_phases > RD_MAX_PHASES || d.in_stride != 1 || d.out_stride < 1 || d.out_stride > 2) return false;
    if (d.n_phases == 1 && d.out_stride != 1) return false;      // the 3x3 convolution ...
    if (d.n_phases > 1 && d.out_stride != 2) return false;       // ... or the UpProj parity phases
    for (int i = 0; i < d.n_phases; ++i) {
        const RdPhase& p = d.phase[i];
        int tr, tc;
        if (!ws_phase_ok(p, tr, tc)) return false;
        if (p.lh != d.phase[0].lh || p.lw != d.phase[0].lw || p.lh > d.Hi || p.lw > d.Wi) return false;
        if (d.out_stride * (p.lh - 1) + p.out_off_h >= d.Ho || d.out_stride * (p.lw - 1) + p.out_off_w >= d.Wo) return false;
    }
    if (d.n_phases == 1 && d.phase[0].n_taps != 9) return false;
    if (d.Cin < 64 || d.Cout < 64 || d.Cin % 8 != 0 || d.Cout % 8 != 0 || d.ldi % 4 != 0 || d.ldo % 4 != 0) return false;
    if ((int64_t)d.Hi * d.Wi * d.ldi * 4 >= (int64_t)WS_OOB || (int64_t)d.Ho * d.Wo * d.ldo * 4 >= (int64_t)WS_OOB) return false;
    return true;
}

static WsPlan ws_plan(const RdConvDesc& d) {
    WsPlan pl{};
    pl.ok = ws_shape_ok(d) ? 1 : 0;
    if (!pl.ok) return pl;
    // tile geometry: the tall one (fewer staged halo pixels: layer1 158 -> 143 us, layer2 171 -> 153, UpProj-64 192 -> 163) unless it
    // pads the logical grid more than 10 % beyond the wide one (measured level within 7 %: 29 / 30 x 50 grids)
    {
        const int lh = d.phase[0].lh, lw = d.phase[0].lw;
        const long long wide_px = (long long)cdiv(lh, WsWide::R) * WsWide::R * cdiv(lw, WsWide::TW) * WsWide::TW;
        const long long tall_px = (long long)cdiv(lh, WsTall::R) * WsTall::R * cdiv(lw, WsTall::TW) * WsTall::TW;
        static const char* force = getenv("RD_WGRAD_SPLIT_TILE");       // diagnostics: "wide" / "tall"
        pl.tall = force ? (force[0] == 't') : (tall_px * 10 <= wide_px * 11);
        pl.tiles_h = cdiv(lh, pl.tall ? WsTall::R : WsWide::R);
        pl.tiles_w = cdiv(lw, pl.tall ? WsTall::TW : WsWide::TW);
    }
    pl.total_tiles = d.N * pl.tiles_h * pl.tiles_w;
    pl.n_cib = cdiv(d.Cin, 64);
    pl.n_cob = cdiv(d.Cout, 64);
    for (int i = 0; i < d.n_phases; ++i)
        for (int t = 0; t < d.phase[i].n_taps; ++t) pl.S = pl.S > d.phase[i].widx[t] + 1 ? pl.S : d.phase[i].widx[t] + 1;
    // one workgroup per CU: about num_cus workgroups in all, at least four tiles per split (the first tile's staging is exposed)
    static const char* wpc = getenv("RD_WGRAD_SPLIT_WG_PER_CU");      // diagnostics
    int ns = (wpc ? atoi(wpc) : 1) * num_cus() / (pl.n_cib * pl.n_cob);
    if (ns < 1) ns = 1;
    const int max_ns = cdiv(pl.total_tiles, 4);
    if (ns > max_ns) ns = max_ns < 1 ? 1 : max_ns;
    pl.tiles_per_split = cdiv(pl.total_tiles, ns);
    pl.n_splits = cdiv(pl.total_tiles, pl.tiles_per_split);
    pl.J = pl.n_splits < 16 ? pl.n_splits : 16;
    return pl;
}

template <int TR, int TC, typename G, bool PRE>
static int launch_ws(const WsArgs& a, int grid, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = wgrad_split_kernel<TR, TC, G, PRE>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 2 * (PRE ? G::BUF_P : G::BUF), s, a);
    RD_CHECK_LAUNCH("wgrad_split_kernel");
    return RD_OK;
}

}  // namespace rd

using namespace rd;

extern "C" int rd_wgrad_split_supported(const RdConvDesc* d) { return d && ws_shape_ok(*d) ? 1 : 0; }

extern "C" int64_t rd_wgrad_split_workspace_floats(const RdConvDesc* d) {
    if (!d) return RD_EINVAL;
    const WsPlan pl = ws_plan(*d);
    if (!pl.ok) return RD_EINVAL;
    return (int64_t)(pl.n_splits + pl.J) * pl.S * d->Cin * d->Cout;
}

// diagnostics: out[0..3] = splits, tiles per split, workgroups per launch, tiles (+ 2^30 when the pixel tile is 4 x 16)
extern "C" int rd_wgrad_split_plan_info(const RdConvDesc* d, int32_t* out) {
    if (!d || !out) return RD_EINVAL;
    const WsPlan pl = ws_plan(*d);
    if (!pl.ok) return RD_EINVAL;
    out[0] = pl.n_splits; out[1] = pl.tiles_per_split; out[2] = pl.n_splits * pl.n_cib * pl.n_cob; out[3] = pl.total_tiles + (pl.tall ? 1 << 30 : 0);
    return RD_OK;
}

template <bool PRE>
static int ws_launch_all(const RdConvDesc* d, WsArgs& a, const WsPlan& pl, hipStream_t s) {
    const int grid = pl.n_splits * pl.n_cib * pl.n_cob;
    // one launch per phase (the 3x3 convolution has one); the phases write disjoint slabs of the same splits
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        int tr, tc;
        ws_phase_ok(p, tr, tc);
        a.dh0 = p.dh_min; a.dw0 = p.dw_min; a.off_h = p.out_off_h; a.off_w = p.out_off_w; a.lh = p.lh; a.lw = p.lw;
        for (int t = 0; t < 9; ++t) a.widx[t] = t < p.n_taps ? p.widx[t] : 0;
        int rc;
        if (pl.tall) {
            if (tr == 3 && tc == 3) rc = launch_ws<3, 3, WsTall, PRE>(a, grid, s);
            else if (tr == 2 && tc == 3) rc = launch_ws<2, 3, WsTall, PRE>(a, grid, s);
            else if (tr == 3 && tc == 2) rc = launch_ws<3, 2, WsTall, PRE>(a, grid, s);
            else rc = launch_ws<2, 2, WsTall, PRE>(a, grid, s);
        } else {
            if (tr == 3 && tc == 3) rc = launch_ws<3, 3, WsWide, PRE>(a, grid, s);
            else if (tr == 2 && tc == 3) rc = launch_ws<2, 3, WsWide, PRE>(a, grid, s);
            else if (tr == 3 && tc == 2) rc = launch_ws<3, 2, WsWide, PRE>(a, grid, s);
            else rc = launch_ws<2, 2, WsWide, PRE>(a, grid, s);
        }
        if (rc != RD_OK) return rc;
    }
    return RD_OK;
}

static void ws_fill_args(const RdConvDesc* d, const WsPlan& pl, WsArgs& a) {
    a.N = d->N; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin; a.ldi = d->ldi; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.ldo = d->ldo;
    a.tiles_h = pl.tiles_h; a.tiles_w = pl.tiles_w; a.total_tiles = pl.total_tiles; a.tiles_per_split = pl.tiles_per_split;
    a.n_splits = pl.n_splits; a.n_cib = pl.n_cib; a.n_cob = pl.n_cob;
    a.OS = d->out_stride; a.S = pl.S;
    a.x = nullptr; a.dy = nullptr; a.xp = nullptr; a.yp = nullptr; a.xplane = a.yplane = 0;
    { const char* dbg = getenv("RD_WGRAD_SPLIT_DEBUG"); a.dbg = dbg ? atoi(dbg) : 0; }
    a.xm = (long long)d->N * d->Hi * d->Wi; a.ym = (long long)d->N * d->Ho * d->Wo;
}

extern "C" int rd_wgrad_split(const RdConvDesc* d, const float* in, const float* dout, float* slabs, void* stream) {
    RD_CHECK_ARG(d && in && dout && slabs, "wgrad_split: null argument");
    const WsPlan pl = ws_plan(*d);
    if (!pl.ok) { set_error("wgrad_split: descriptor not supported (rd_wgrad_split_supported)"); return RD_EINVAL; }
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(dout) % 16 == 0, "wgrad_split: unaligned tensor");
    WsArgs a;
    ws_fill_args(d, pl, a);
    a.x = in; a.dy = dout; a.slabs = slabs;
    return ws_launch_all<false>(d, a, pl, static_cast<hipStream_t>(stream));
}

// Both operands already split by their producers (piece planes [piece][C/16][pixels][16] bf16, rd_split_pieces): the staging waves
// issue nothing but global_load_lds copies.  Same slabs, same plan, same reduction as rd_wgrad_split; the channel counts must be
// multiples of 16 (rd_wgrad_split_pre_supported).
extern "C" int rd_wgrad_split_pre_supported(const RdConvDesc* d) { return d && ws_shape_ok(*d) && d->Cin % 16 == 0 && d->Cout % 16 == 0 ? 1 : 0; }

extern "C" int rd_wgrad_split_pre(const RdConvDesc* d, const void* x_pieces, int64_t x_piece_elems, const void* dy_pieces, int64_t dy_piece_elems,
                                  float* slabs, void* stream) {
    RD_CHECK_ARG(d && x_pieces && dy_pieces && slabs, "wgrad_split_pre: null argument");
    const WsPlan pl = ws_plan(*d);
    if (!pl.ok || d->Cin % 16 || d->Cout % 16) { set_error("wgrad_split_pre: descriptor not supported (rd_wgrad_split_pre_supported)"); return RD_EINVAL; }
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(x_pieces) % 16 == 0 && reinterpret_cast<uintptr_t>(dy_pieces) % 16 == 0, "wgrad_split_pre: unaligned tensor");
    WsArgs a;
    ws_fill_args(d, pl, a);
    RD_CHECK_ARG(x_piece_elems >= (int64_t)d->Cin * a.xm && dy_piece_elems >= (int64_t)d->Cout * a.ym && x_piece_elems % 8 == 0 && dy_piece_elems % 8 == 0,
                 "wgrad_split_pre: piece stride too small");
    a.xp = static_cast<const unsigned short*>(x_pieces); a.yp = static_cast<const unsigned short*>(dy_pieces);
    a.xplane = (long long)x_piece_elems * 2; a.yplane = (long long)dy_piece_elems * 2;
    a.slabs = slabs;
    return ws_launch_all<true>(d, a, pl, static_cast<hipStream_t>(stream));
}

extern "C" int rd_wgrad_split_reduce(const RdConvDesc* d, const float* slabs, float* grad_oihw, int32_t O, int32_t I, int32_t KH, int32_t KW,
                                     int32_t co_off, int32_t accumulate, void* stream) {
    RD_CHECK_ARG(d && slabs && grad_oihw, "wgrad_split_reduce: null argument");
    const WsPlan pl = ws_plan(*d);
    if (!pl.ok) { set_error("wgrad_split_reduce: descriptor not supported"); return RD_EINVAL; }
    RD_CHECK_ARG(KH * KW == pl.S && I == d->Cin && co_off >= 0 && co_off + O <= d->Cout, "wgrad_split_reduce: gradient shape does not match the descriptor");
    const int64_t E = (int64_t)pl.S * d->Cin * d->Cout;
    float* tmp = const_cast<float*>(slabs) + (int64_t)pl.n_splits * E;
    return launch_slab_reduce(slabs, pl.n_splits, E, tmp, grad_oihw, pl.S, d->Cin, d->Cout, O, I, co_off, accumulate, static_cast<hipStream_t>(stream));
}
