// fp32 weight gradient of the 7x7 / stride-2 stems on the bf16 matrix cores (models.py:539,559,633,643; the step's last kernel):
//     dW[k = (kh, kw, c)][co] = sum over output pixels (n, oh, ow) of  in[n, c, 2 oh + kh - 3, 2 ow + kw - 3] * dout[n, oh, ow, co]
// with both operands split into three bf16 pieces while they are staged and every product rebuilt from six v_mfma_f32_32x32x16_bf16
// (gconv_split.hip's arithmetic and error analysis; a bf16-storage dout is its own single piece: three MFMAs).  The fp32 kernel
// (stem.hip: v_mfma_f32_32x32x2_f32, 292 of its 328 us at b = 16 are the MFMA walk) sits alone at the end of the step.
//
// The reduction index is the PIXEL, and the bf16 MFMA wants eight consecutive pixels per lane for a fixed row:
//   * dout is pixel-major in HBM and in LDS ([piece][32-channel tile][pixel][32 channels], 64 bytes per pixel) and its fragments are
//     transposing reads (ds_read_b64_tr_b16), exactly wgrad_split.hip's;
//   * the input row of a lane is a stride-2 walk over an image row: column 2 ow + kw - 3.  The halo patch is therefore staged
//     de-interleaved by column parity -- [piece][plane c][patch row][parity][40] bf16, patch column x = 2 (ow - c0) + kw + 1 lives in
//     parity plane x & 1 at index x >> 1 -- so that eight consecutive output pixels are eight consecutive elements, starting at
//     element 8 q + sh with sh = (kw + 1) >> 1 in 0..3.  gfx950 serves a ds_read_b128 at such a 2-byte-aligned address correctly but at
//     1/8 to 1/12 of the aligned rate (tools/micro/lds_unaligned.hip: 48-64 LDS clocks per read against 4-8; the first version of this
//     kernel was LDS-bound on them, 285 us).  A fragment is therefore five 4-byte-aligned dwords from element 8 q + 2 (sh >> 1) on and
//     four v_alignbit_b32 by 16 (sh & 1) bits: per-lane constants.  No im2col image, no gather in the walk.
//
//   workgroup : 8 waves, one per CU.  Waves 4-7 stage tile i + 1 (patch: 4-byte loads, split, 2-byte stores; dout: 8-channel units,
//               split, 16-byte stores) into the other LDS buffer; one barrier per tile.  Waves 0-3 walk tile i: the tile's eight
//               16-pixel reduction steps are dealt to them two each (split-K inside the workgroup), every wave holds the whole
//               160 x 64 accumulator block (MTK x NT tiles of 32 x 32) and the four are summed through LDS at the end.
//   pixel tile: 4 rows x 32 columns of one image.
//   split-K   : contiguous tile ranges per workgroup, slabs [split][k][Cout] with k = (kh * 7 + kw) * Cin + c -- stem.hip's layout,
//               reduced by the same deterministic slab reduction.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace rd {

int launch_slab_reduce(const float* slabs, int n_splits, int64_t E, float* tmp, float* grad_oihw, int S, int Cin, int Cout,
                       int O, int I, int co_off, int accumulate, hipStream_t s);   // wgrad.hip

typedef __bf16 swbf16x8 __attribute__((ext_vector_type(8)));
typedef short sws16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int swu32x4 __attribute__((ext_vector_type(4)));
typedef float swf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 swbf16x2 __attribute__((ext_vector_type(2)));

constexpr int SW_R = 4, SW_TW = 32, SW_PIX = SW_R * SW_TW;       // 128 output pixels per tile, eight reduction steps
constexpr int SW_PR = 2 * SW_R + 5;                              // 13 patch rows
constexpr int SW_PC = 72;                                        // staged patch columns x = 0 .. 71 (read: 1 .. 70)
constexpr int SW_PITCH = 40;                                     // elements per (row, parity) line (36 used)
constexpr int SW_PPLANE = ((3 * SW_PR * 2 * SW_PITCH * 2 + 255) / 256) * 256;   // bytes of one piece of the patch (three input planes)
constexpr int SW_XBYTES = 3 * SW_PPLANE;
constexpr int SW_YPLANE = SW_PIX * 64;                           // bytes of one [pixel][32 channels] bf16 plane of dout
constexpr unsigned SW_OOB = 0x80000000u;

struct StemWsArgs {
    const float* plane[3];
    long long stride[3];        // elements between consecutive images of each input plane
    const void* dout;           // NHWC [N,Ho,Wo,Cout], fp32 or (io16) bf16
    float* slabs;
    int Cin, N, H, W, Ho, Wo, Cout, tiles_h, tiles_w, total_tiles, tiles_per_split;
    // BNF (rd_stem_wgrad_split_bn_t): dout is the gradient g at the stem BatchNorm's OUTPUT, bn_x the BatchNorm input (the stem's raw
    // output), and the staged operand is that BatchNorm's input gradient A*g + B*(x - mean) + K per channel (bn_coef = [3][Cout] as
    // bn_bwd_coeffs_kernel writes it) -- bn_bwd_dx_kernel's expression, term for term; a bf16-storage plan rounds it to bf16 as that
    // kernel's store would
    const void* bn_x;
    const float* bn_coef;
    const float* bn_mean;
    int dbg;                    // ablation bits (RD_STEM_WGRAD_SPLIT_DEBUG; results garbage): 1 no MFMA walk, 2 no staging after the first tile,
                                // 4 staging without the split arithmetic, 8 without its LDS stores, 16 without its global loads
};

__device__ __forceinline__ unsigned sw_cvt_pk(float a, float b) {
    swf32x2 v;
    v[0] = a; v[1] = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, swbf16x2));
}
// three bf16 pieces of eight fp32 values, a pair at a time (gconv_split.hip)
__device__ __forceinline__ void sw_split8(const float4 v0, const float4 v1, swu32x4& w0, swu32x4& w1, swu32x4& w2) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = x[2 * i], b = x[2 * i + 1];
        const unsigned u0 = sw_cvt_pk(a, b);
        a -= __uint_as_float(u0 << 16);
        b -= __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = sw_cvt_pk(a, b);
        a -= __uint_as_float(u1 << 16);
        b -= __uint_as_float(u1 & 0xffff0000u);
        w0[i] = u0;
        w1[i] = u1;
        w2[i] = sw_cvt_pk(a, b);
    }
}
// eight consecutive pixels of one dout channel: two transposing reads of four pixels each (64 bytes per pixel)
__device__ __forceinline__ swbf16x8 sw_frag_tr(unsigned addr) {
    typedef __attribute__((address_space(3))) sws16x4* lp;
    const sws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lp>(addr));
    const sws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lp>(addr + 4 * 64));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(swbf16x8, r);
}
// eight consecutive elements of a parity line from a 2-byte-aligned position, in two halves so that the MFMAs of the previous row tile
// sit between the reads and the first use of their data: sw_row_load fetches the five dwords from the 4-byte-aligned address of the
// dword that holds the first element, sw_row_finish shifts them by sh16 = 16 bits if that element is the dword's upper half
__device__ __forceinline__ void sw_row_load(unsigned addr, unsigned (&e)[5]) {
    typedef __attribute__((address_space(3))) const unsigned* lp;
    const lp q = reinterpret_cast<lp>(addr);
#pragma unroll
    for (int i = 0; i < 5; ++i) e[i] = q[i];
}
__device__ __forceinline__ swbf16x8 sw_row_finish(const unsigned (&e)[5], unsigned sh16) {
    swu32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_alignbit(e[i + 1], e[i], sh16);
    return __builtin_bit_cast(swbf16x8, v);
}

// MTK = ceil(49 Cin / 32) row tiles, NT = 32-channel tiles of dout, B16: dout is bf16 (one piece), BNF: dout is formed from (g, x,
// coefficients) by the staging waves
template <int MTK, int NT, bool B16, bool BNF = false>
__global__ __launch_bounds__(512) void stem_wgrad_split_kernel(const StemWsArgs a) {
    constexpr int YBYTES = 3 * NT * SW_YPLANE;          // (B16 plans use the first piece only)
    constexpr int BUF = SW_XBYTES + YBYTES;
    constexpr int NPB = B16 ? 1 : 3;
    constexpr int CIN = MTK == 5 ? 3 : MTK == 4 ? 2 : 1;      // (49 Cin rows in MTK tiles of 32)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const unsigned lds0 = (unsigned)(size_t)smem;
    constexpr int K = 49 * CIN;

    const int split = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_begin = split * a.tiles_per_split;
    const int tile_end = min(tile_begin + a.tiles_per_split, a.total_tiles);
    const int ntiles = tile_end - tile_begin;
    const int tiles_img = a.tiles_h * a.tiles_w;

    f32x16 acc[MTK][NT];      // (zeroed in the compute branch only)

    if (loader) {
        // ------------------------------------------------------------------------------------------ staging waves
        const int lt = tid - 256;
        // patch slot of this thread: four consecutive columns x0 .. x0 + 3 (x0 = 4 (lt % 18)) of patch row lt / 18 -- ONE 16-byte load per
        // plane; (x0, x0 + 2) are neighbours of the even-column line, (x0 + 1, x0 + 3) of the odd one: a dword store each per piece
        // (2-byte stores of single elements: twice the LDS instructions, and two lanes per dword)
        constexpr int XG = SW_PC / 4;                          // 18 groups per row
        const bool pin = lt < SW_PR * XG;
        const int ppr = pin ? lt / XG : 0, px0 = pin ? 4 * (lt - (lt / XG) * XG) : 0;
        const unsigned pdst = pin ? (unsigned)((ppr * 2) * SW_PITCH + (px0 >> 1)) * 2u : 36u * 2u;      // (idle threads: padding elements of line 0)
        // dout units of this thread: 8-channel group q8 = lt % (4 NT) of pixels lt / (4 NT) + (256 / (4 NT)) j -- one channel group per thread
        // (a wave's load covers whole 128- / 256-byte pixel rows; with BNF one set of coefficients per thread)
        constexpr int UQ = 4 * NT;                            // 8-channel units per pixel
        constexpr int UPY = SW_PIX * UQ / 256;                // 2 NT
        const int yq8 = lt % UQ;
        const int ych0 = yq8 * 8;
        int ypx[UPY], ydst[UPY];
#pragma unroll
        for (int j = 0; j < UPY; ++j) {
            ypx[j] = lt / UQ + (256 / UQ) * j;
            ydst[j] = SW_XBYTES + (yq8 >> 2) * SW_YPLANE + (ypx[j] * 4 + (yq8 & 3)) * 16;
        }
        const unsigned yimg = (unsigned)(a.Ho * a.Wo * a.Cout) * (B16 ? 2u : 4u);
        // two register sets: the loads of tile i + 2 are issued BEFORE tile i + 1 is split and stored, so that they have a whole iteration in
        // flight (with one set, issued behind the split, the next iteration's split waited for them: 195 us, 142 us without the loads)
        struct Regs { float4 pf[3]; int pnval; float4 y0[UPY], y1[UPY]; float4 x0[BNF ? UPY : 1], x1[BNF ? UPY : 1]; unsigned ok; };
        Regs ra, rb;
        ra.pnval = rb.pnval = 4;
        ra.ok = rb.ok = 0;
        // (BNF) a thread's units all cover the same eight channels: their coefficients are kept in registers (the compute waves set the
        // kernel's register budget; these waves have room)
        float cfA[8], cfB[8], cfK[8], cfM[8];
        if constexpr (BNF) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ch = ych0 + k;
                const bool in = ch < a.Cout;
                cfA[k] = in ? a.bn_coef[ch] : 0.f;
                cfB[k] = in ? a.bn_coef[a.Cout + ch] : 0.f;
                cfK[k] = in ? a.bn_coef[2 * a.Cout + ch] : 0.f;
                cfM[k] = in ? a.bn_mean[ch] : 0.f;
            }
        }
        auto fetch = [&](int tile, Regs& rg) {
            float4 (&pf)[3] = rg.pf; int& pnval = rg.pnval; float4 (&y0)[UPY] = rg.y0; float4 (&y1)[UPY] = rg.y1;
            if (a.dbg & 16) return;
            const int n = tile / tiles_img, tr = tile - n * tiles_img;
            const int r0 = (tr / a.tiles_w) * SW_R, c0 = (tr % a.tiles_w) * SW_TW;
            const int ih0 = 2 * r0 - 3, iw0 = 2 * c0 - 4;          // patch row 0 / patch column x = 0
            const int ih = ih0 + ppr, iw = iw0 + px0;          // (iw0 and x0 are multiples of 4 columns apart from -4: a group is never split by the left edge)
            const int nval = a.W - iw;                         // columns of the group inside the image
            pnval = nval;                                      // (the mask is applied when the values are split: nothing here may wait for a load)
            const unsigned poff = (pin && ih >= 0 && ih < a.H && iw >= 0 && nval > 0) ? (unsigned)(ih * a.W + iw) * 4u : SW_OOB;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c < CIN) {
                    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(a.plane[c] + (size_t)n * a.stride[c]), 0, (unsigned)(a.H * a.W) * 4u, 0x00020000);
                    pf[c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)poff, 0, 0));
                }
            }
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(static_cast<const char*>(a.dout)) + (size_t)n * a.Ho * a.Wo * a.Cout * (B16 ? 2 : 4), 0, yimg, 0x00020000);
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(static_cast<const char*>(BNF ? a.bn_x : a.dout)) + (size_t)n * a.Ho * a.Wo * a.Cout * (B16 ? 2 : 4), 0, yimg, 0x00020000);
            unsigned okm = 0;
#pragma unroll
            for (int j = 0; j < UPY; ++j) {
                const int oh = r0 + (ypx[j] >> 5), ow = c0 + (ypx[j] & 31);
                const bool ok = oh < a.Ho && ow < a.Wo && ych0 < a.Cout;
                okm |= ok ? 1u << j : 0u;
                if (B16) {
                    const unsigned off = ok ? (unsigned)(((oh * a.Wo + ow) * a.Cout + ych0) * 2) : SW_OOB;
                    y0[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off, 0, 0));
                    if constexpr (BNF) rg.x0[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
                } else {
                    const unsigned off = ok ? (unsigned)(((oh * a.Wo + ow) * a.Cout + ych0) * 4) : SW_OOB;
                    y0[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off, 0, 0));
                    y1[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off + 16, 0, 0));
                    if constexpr (BNF) {
                        rg.x0[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
                        rg.x1[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off + 16, 0, 0));
                    }
                }
            }
            rg.ok = okm;
        };
        auto split_put = [&](int buf, Regs& rg) {
            float4 (&pf)[3] = rg.pf; const int pnval = rg.pnval; float4 (&y0)[UPY] = rg.y0; float4 (&y1)[UPY] = rg.y1;
            const unsigned base = lds0 + buf * BUF;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c < CIN) {
                    // (columns past the right edge are the next row's first ones, or past the image: zero them)
                    float e0 = pf[c].x, e2 = pnval > 2 ? pf[c].z : 0.f, o0 = pnval > 1 ? pf[c].y : 0.f, o2 = pnval > 3 ? pf[c].w : 0.f;
                    const unsigned ad = base + c * (SW_PR * 2 * SW_PITCH * 2) + pdst;
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        unsigned ue, uo;
                        if (a.dbg & 4) { ue = __float_as_uint(e0); uo = __float_as_uint(o0); }
                        else {
                            ue = sw_cvt_pk(e0, e2); uo = sw_cvt_pk(o0, o2);
                            e0 -= __uint_as_float(ue << 16); e2 -= __uint_as_float(ue & 0xffff0000u);
                            o0 -= __uint_as_float(uo << 16); o2 -= __uint_as_float(uo & 0xffff0000u);
                        }
                        typedef __attribute__((address_space(3))) unsigned* lp;
                        if (!(a.dbg & 8)) {
                            *reinterpret_cast<lp>(ad + pc * SW_PPLANE) = ue;
                            *reinterpret_cast<lp>(ad + pc * SW_PPLANE + SW_PITCH * 2) = uo;
                        }
                    }
                }
            }
            if constexpr (BNF) {
                // the BatchNorm input gradient of the unit's eight channels, in place (units outside the tile / the channel range stay zero)
#pragma unroll
                for (int j = 0; j < UPY; ++j) {
                    float g8[8], x8[8];
                    if (B16) {
                        const swu32x4 gu = __builtin_bit_cast(swu32x4, y0[j]), xu = __builtin_bit_cast(swu32x4, rg.x0[j]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            g8[2 * k] = __uint_as_float(gu[k] << 16); g8[2 * k + 1] = __uint_as_float(gu[k] & 0xffff0000u);
                            x8[2 * k] = __uint_as_float(xu[k] << 16); x8[2 * k + 1] = __uint_as_float(xu[k] & 0xffff0000u);
                        }
                    } else {
                        g8[0] = y0[j].x; g8[1] = y0[j].y; g8[2] = y0[j].z; g8[3] = y0[j].w; g8[4] = y1[j].x; g8[5] = y1[j].y; g8[6] = y1[j].z; g8[7] = y1[j].w;
                        x8[0] = rg.x0[j].x; x8[1] = rg.x0[j].y; x8[2] = rg.x0[j].z; x8[3] = rg.x0[j].w;
                        x8[4] = rg.x1[j].x; x8[5] = rg.x1[j].y; x8[6] = rg.x1[j].z; x8[7] = rg.x1[j].w;
                    }
                    const bool ok = (rg.ok >> j) & 1u;
                    float d8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d = fmaf(cfA[k], g8[k], fmaf(cfB[k], x8[k] - cfM[k], cfK[k]));
                        d8[k] = ok ? d : 0.f;
                    }
                    if (B16) {
                        swu32x4 w;
#pragma unroll
                        for (int k = 0; k < 4; ++k) w[k] = sw_cvt_pk(d8[2 * k], d8[2 * k + 1]);      // (round to nearest even: the separate pass's store)
                        y0[j] = __builtin_bit_cast(float4, w);
                    } else {
                        y0[j] = make_float4(d8[0], d8[1], d8[2], d8[3]);
                        y1[j] = make_float4(d8[4], d8[5], d8[6], d8[7]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UPY; ++j) {
                const unsigned ad = base + ydst[j];
                if (B16) {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(__builtin_bit_cast(swu32x4, y0[j])) : "memory");
                } else {
                    swu32x4 w0, w1, w2;
                    if (a.dbg & 4) { w0 = __builtin_bit_cast(swu32x4, y0[j]); w1 = __builtin_bit_cast(swu32x4, y1[j]); w2 = w0; }
                    else sw_split8(y0[j], y1[j], w0, w1, w2);
                    if (!(a.dbg & 8)) {
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(w0) : "memory");
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ad + NT * SW_YPLANE), "v"(w1) : "memory");
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ad + 2 * NT * SW_YPLANE), "v"(w2) : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // tile i of this workgroup lives in buffer i & 1 and, while in registers, in set A for even i, set B for odd i.  Iteration i: barrier
        // B(i) (tile i published, buffer (i + 1) & 1 free); fetch tile i + 2; split and store tile i + 1 (fetched during iteration i - 1)
        const bool nostage = a.dbg & 2;
        if (ntiles > 0) {
            fetch(tile_begin, ra);
            if (ntiles > 1) fetch(tile_begin + 1, rb);
            split_put(0, ra);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < ntiles; i += 2) {
            rd_sync();                            // B(i), i even: tile i + 1 is in set B, set A is free
            if (i + 2 < ntiles && !nostage) fetch(tile_begin + i + 2, ra);
            if (i + 1 < ntiles && !nostage) split_put(1, rb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i + 1 >= ntiles) break;
            rd_sync();                            // B(i + 1): tile i + 2 is in set A, set B is free
            if (i + 3 < ntiles && !nostage) fetch(tile_begin + i + 3, rb);
            if (i + 2 < ntiles && !nostage) split_put(0, ra);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        rd_sync();                                // matches the compute waves' final barrier
    } else {
        // ------------------------------------------------------------------------------------------ compute waves
#pragma unroll
        for (int mt = 0; mt < MTK; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        const int l31 = lane & 31, g = lane >> 5;
        // lane part of a patch fragment address: row m = mt * 32 + l31 = (kh * 7 + kw) * Cin + c; rows past K read row 0 (never stored)
        unsigned arow[MTK], ash[MTK];
#pragma unroll
        for (int mt = 0; mt < MTK; ++mt) {
            int m = mt * 32 + l31;
            if (m >= K) m = 0;
            const int t = m / CIN, c = m - t * CIN, kh = t / 7, kw = t - kh * 7;
            const int sh = (kw + 1) >> 1;
            arow[mt] = lds0 + (unsigned)(((c * SW_PR + kh) * 2 + ((kw + 1) & 1)) * SW_PITCH + 2 * (sh >> 1) + 8 * g) * 2u;
            ash[mt] = 16u * (sh & 1);
        }
        // lane part of a dout fragment address (wgrad_split.hip): pixel row (lane & 15) / 4 of the group's four, 8-byte chunk lane & 3,
        // second 16 channels for lanes 16..31 of each half, pixels 8.. for the upper half wave
        const unsigned yb = lds0 + SW_XBYTES + (unsigned)(((lane & 15) >> 2) + g * 8) * 64u + (lane & 3) * 8u + ((lane >> 4) & 1) * 32u;
        for (int i = 0; i < ntiles; ++i) {
            rd_sync();                            // B(i)
            if (a.dbg & 1) continue;
            const unsigned bo = (i & 1) * BUF;
            // this wave's two reduction steps of the tile: ks = wave and wave + 4, step ks = (row ks >> 1, column half ks & 1)
            unsigned xo[2], yo[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ks = wave + 4 * h;
                const int rl = ks >> 1, s = ks & 1;
                xo[h] = bo + (unsigned)(4 * rl * SW_PITCH + 16 * s) * 2u;
                yo[h] = bo + (unsigned)(rl * 32 + 16 * s) * 64u;
            }
            unsigned R[3][5];
            swbf16x8 A[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) sw_row_load(arow[0] + xo[0] + p * SW_PPLANE, R[p]);
#pragma unroll
            for (int p = 0; p < 3; ++p) A[p] = sw_row_finish(R[p], ash[0]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                swbf16x8 B[NT][NPB];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int p = 0; p < NPB; ++p) B[nt][p] = sw_frag_tr(yb + yo[h] + nt * SW_YPLANE + p * NT * SW_YPLANE);
#pragma unroll
                for (int mt = 0; mt < MTK; ++mt) {
                    // the next row tile's dwords (the first one of the second step included) are requested in front of this one's MFMAs and
                    // shifted into place behind them; the order is pinned (left alone the compiler hoists the reads of a whole step)
                    const bool more = mt + 1 < MTK || h == 0;
                    const int nmt = mt + 1 < MTK ? mt + 1 : 0, nh = mt + 1 < MTK ? h : 1;
                    if (more) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) sw_row_load(arow[nmt] + xo[nh] + p * SW_PPLANE, R[p]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x16 c = acc[mt][nt];
                        if (B16) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[nt][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[nt][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[nt][0], c, 0, 0, 0);
                        } else {
                            RD_SPLIT_TERMS(c, A[0], A[1], A[2], B[nt][0], B[nt][NPB > 1 ? 1 : 0], B[nt][NPB > 2 ? 2 : 0])
                        }
                        acc[mt][nt] = c;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) A[p] = sw_row_finish(R[p], ash[nmt]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        rd_sync();
    }
    // ---- the four compute waves' partial blocks summed through LDS, one accumulator tile at a time (every wave takes part in the
    // barriers); slab [k][Cout] of this split (zeros when the split has no tiles)
    float* red = smem;      // [4][16][64]
    float* slab = a.slabs + (size_t)split * K * a.Cout;
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            rd_sync();
            if (!loader) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[mt][nt][i];
            }
            rd_sync();
            if (!loader) {
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
}

}  // namespace rd
using namespace rd;

extern "C" int64_t rd_stem_wgrad_workspace_floats(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout);     // stem.hip: an upper bound for this kernel's splits too

extern "C" int rd_stem_wgrad_split_supported(int32_t Cin, int32_t Cout) {
    static const char* off = getenv("RD_STEM_WGRAD_SPLIT");      // RD_STEM_WGRAD_SPLIT=0: diagnostics
    if (off && atoi(off) == 0) return 0;
    return ((Cin == 3 && Cout == 64) || (Cin >= 1 && Cin <= 2 && Cout == 16)) ? 1 : 0;
}

namespace rd {
int launch_bn_bwd_coeffs(const float* red_partial, int n_tiles, int C, int which, double count, const float* gamma, const float* invstd,
                         float* dgamma, float* dbeta, float* coef_ws, hipStream_t s);     // norm_act.hip
}

static int stem_wgrad_split_impl(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                                 int32_t W, const void* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream, const void* bn_x,
                                 const float* bn_coef, const float* bn_mean) {
    RD_CHECK_ARG(dtype == RD_DTYPE_F32 || dtype == RD_DTYPE_BF16, "stem_wgrad_split_t: bad dtype %d", dtype);
    RD_CHECK_ARG(planes && strides && dout && grad_oihw && ws && N > 0 && H > 0 && W > 0, "stem_wgrad_split: null tensor / empty shape");
    RD_CHECK_ARG(rd_stem_wgrad_split_supported(Cin, Cout) == 1 && cdiv(49 * Cin, 32) == (Cin == 3 ? 5 : Cin == 2 ? 4 : 2), "stem_wgrad_split: unsupported shape Cin=%d Cout=%d", Cin, Cout);
    RD_CHECK_ARG((int64_t)H * W * 4 < (int64_t)SW_OOB && (int64_t)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * Cout * 4 < (int64_t)SW_OOB,
                 "stem_wgrad_split: an image exceeds the 2 GB buffer-addressing range");
    StemWsArgs a;
    for (int c = 0; c < 3; ++c) {
        a.plane[c] = c < Cin ? planes[c] : nullptr;
        a.stride[c] = c < Cin ? strides[c] : 0;
        RD_CHECK_ARG(c >= Cin || planes[c] != nullptr, "stem_wgrad_split: null input plane %d", c);
    }
    a.dout = dout; a.slabs = ws;
    a.bn_x = bn_x; a.bn_coef = bn_coef; a.bn_mean = bn_mean;
    a.Cin = Cin; a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    a.tiles_h = cdiv(a.Ho, SW_R); a.tiles_w = cdiv(a.Wo, SW_TW);
    a.total_tiles = N * a.tiles_h * a.tiles_w;
    { static const char* dbg = getenv("RD_STEM_WGRAD_SPLIT_DEBUG"); a.dbg = dbg ? atoi(dbg) : 0; }
    // one workgroup per CU, at least two tiles per split (the first tile's staging is exposed); never more splits than stem.hip's
    // kernel would use (the workspace is sized for those)
    int ns = num_cus();
    const int max_ns = cdiv(a.total_tiles, 2);
    if (ns > max_ns) ns = max_ns < 1 ? 1 : max_ns;
    a.tiles_per_split = cdiv(a.total_tiles, ns);
    const int n_splits = cdiv(a.total_tiles, a.tiles_per_split);
    const int K = 49 * Cin;
    const int64_t E = (int64_t)K * Cout;
    RD_CHECK_ARG((int64_t)(n_splits + (n_splits < 16 ? n_splits : 16)) * E <= rd_stem_wgrad_workspace_floats(N, H, W, Cin, Cout),
                 "stem_wgrad_split: workspace smaller than this kernel's %d splits need", n_splits);
    const int MTK = cdiv(K, 32), NT = Cout > 32 ? 2 : 1;
    const bool b16 = dtype == RD_DTYPE_BF16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = 2 * ((size_t)SW_XBYTES + 3 * (size_t)NT * SW_YPLANE);
    const bool bnf = bn_x != nullptr;
#define RD_SWS(M_, N_, B_, F_)                                                                                       \
    if (MTK == M_ && NT == N_ && b16 == B_ && bnf == F_) {                                                           \
        static std::atomic<unsigned long long> attr_set{0};       /* (one flag set per instantiation) */             \
        RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(stem_wgrad_split_kernel<M_, N_, B_, F_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL((stem_wgrad_split_kernel<M_, N_, B_, F_>), dim3(n_splits), dim3(512), lds, s, a);         \
        RD_CHECK_LAUNCH("stem_wgrad_split_kernel");                                                                  \
    } else
    RD_SWS(5, 2, false, false) RD_SWS(5, 2, true, false) RD_SWS(2, 1, false, false) RD_SWS(2, 1, true, false) RD_SWS(4, 1, false, false)
    RD_SWS(4, 1, true, false) RD_SWS(5, 2, false, true) RD_SWS(5, 2, true, true) RD_SWS(2, 1, false, true) RD_SWS(2, 1, true, true) {
        set_error("stem_wgrad_split: no instantiation for Cin=%d Cout=%d%s", Cin, Cout, bnf ? " with the BatchNorm apply pass folded in" : "");
        return RD_EINVAL;
    }
#undef RD_SWS
    return launch_slab_reduce(ws, n_splits, E, ws + (int64_t)n_splits * E, grad_oihw, 49, Cin, Cout, Cout, Cin, 0, 0, s);
}

// Same contract as rd_stem_wgrad_t (weight gradient OIHW, overwritten; ws of rd_stem_wgrad_workspace_floats floats).
extern "C" int rd_stem_wgrad_split_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                                     int32_t W, const void* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream) {
    return stem_wgrad_split_impl(dtype, planes, strides, Cin, N, H, W, dout, Cout, grad_oihw, ws, stream, nullptr, nullptr, nullptr);
}

// The same with the stem BatchNorm's backward apply pass folded into the staging waves -- for a stem nobody asks an input gradient of
// (the RGB stem; the depth stem outside the multistage network's second stage), whose BatchNorm input gradient only this kernel would
// read: a read of g and x and a write of dx (1.1 GB at b = 16, the last kernel of the main chain) disappear.  g: gradient at the
// BatchNorm OUTPUT (what rd_bnact_maxpool_bwd_stats_t stores); x: the stem's raw output; red_partial / n_tiles: that call's sums;
// coef_ws: 3 * Cout floats.  Equivalent to rd_bn_bwd_apply_t(g, x, ..., which = 1, dx) + rd_stem_wgrad_split_t(dx): same dgamma / dbeta,
// same weight-gradient bits (tests/test_gpu_stem.py).  Shapes: 3 -> 64 and 1 -> 16.
extern "C" int rd_stem_wgrad_split_bn_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                                        int32_t W, const void* g, const void* x, const float* red_partial, int32_t n_tiles,
                                        const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                        float* coef_ws, int32_t Cout, float* grad_oihw, float* ws, void* stream) {
    RD_CHECK_ARG(g && x && red_partial && gamma && mean && invstd && coef_ws && n_tiles > 0, "stem_wgrad_split_bn_t: null argument");
    RD_CHECK_ARG(Cin != 2, "stem_wgrad_split_bn_t: the two-plane stem needs its BatchNorm input gradient (rd_stem_dgrad_channel)");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int rc = launch_bn_bwd_coeffs(red_partial, n_tiles, Cout, 1, (double)N * Ho * Wo, gamma, invstd, dgamma, dbeta, coef_ws,
                                  static_cast<hipStream_t>(stream));
    if (rc != RD_OK) return rc;
    return stem_wgrad_split_impl(dtype, planes, strides, Cin, N, H, W, g, Cout, grad_oihw, ws, stream, x, coef_ws, mean);
}
