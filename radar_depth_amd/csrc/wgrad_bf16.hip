// bf16-operand weight gradient of the gconv-lowered convolutions:
//     dW[slab][ci][co] = sum over the phase's logical grid (r,c) of  x[IS*(r,c) + (dh,dw)][ci] * dy[OS*(r,c) + off][co]
// (3x3 / 1x1 at stride 1 and 2, the four UpProj parity phases of the zero-skipped 5x5) on v_mfma_f32_32x32x16_bf16 (fp32 tensors in HBM, operands rounded to bf16 while they are staged, fp32 accumulation).
// BASELINE.json configs 2/4 (bf16) -- opt-in like gconv_bf16.hip; rd_wgrad (fp32) stays the default and the parity reference.
//
// The reduction dimension of this GEMM is the PIXEL index, and the bf16 MFMA wants eight consecutive k per lane, so both
// operands are transposed on their way into LDS: XT[ci][row][col] and YT[co][row][col] (bf16, columns contiguous).  A lane then
// reads eight consecutive pixels of one channel with a single 16-byte LDS read.  The three horizontal taps of one patch row are
// the same 16 bytes shifted by one pixel: they are built from the aligned read plus the neighbouring dwords with
// v_alignbyte, the vertical taps are row offsets.
//
// A descriptor is decomposed into at most four PASSES, each a stride-1 3x3-shaped problem over decimated tensors: a pass owns
// the taps of one descriptor phase whose offsets share the same residue modulo the input stride -- x sampled at IS*q + (xa,xb),
// dy at OS*q + (ya,yb), tap shifts in [-1,1]^2 (stride-2 conv: 4 input-parity passes of 4/2/2/1 taps; UpProj: its 4 phases of
// 9/6/6/4 taps; 1x1: one tap).  Passes write disjoint weight slabs.
//
//   workgroup : one (<=64 input channels) x (<=64 output channels) block of dW, the pass's taps, a range of pixel tiles (split)
//   pixel tile: R rows x 32 columns of one image; X patch (R+2) x 36 columns (halo), staged with pixel PAIRS packed per dword
//   wave      : one 32x32 (ci, co) tile pair of the block and all nine taps (nine 32x32 accumulators); when the block has fewer
//               than four tile pairs the spare waves take alternate k-steps and write their own partial slab
//   pipeline  : one 8-wave workgroup per CU, two LDS tile buffers: waves 0-3 run the MFMAs of tile i while waves 4-7 load,
//               convert and write tile i+1 (role split instead of software pipelining: loads, conversions and matrix
//               instructions of different tiles are then issued by different waves of the same SIMD)
//   output    : per-split slabs [split][tap][Cin][Cout] like rd_wgrad, reduced in a fixed order by the same slab reduction
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <unordered_map>

#include "common.h"

namespace rd {

int launch_slab_reduce(const float* slabs, int n_splits, int64_t E, float* tmp, float* grad_oihw, int S, int Cin, int Cout,
                       int O, int I, int co_off, int accumulate, hipStream_t s);   // wgrad.hip

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

struct WgradBfArgs {
    const void* x;        // NHWC activations / output gradients: fp32, or bf16 when the kernel is instantiated with IO16
    const void* dy;
    float* slabs;
    int N, Hx, Wx, Hy, Wy, lh, lw, Cin, Cout, ldi, ldo, IS, OS, S;
    int tiles_h, tiles_w, total_tiles, tiles_per_split, n_splits;
    int n_cib, n_cob, cpi, cpo;      // channel blocks; 32-channel tiles per block (1 or 2) on the input / output side
    int n_pass;
    int xa[4], xb[4], ya[4], yb[4];  // per pass: sampling offsets of x (input stride IS) and dy (output stride OS)
    int slab_of_tap[4][9];           // per pass: weight slab index of tap shift (sh+1)*3 + (sw+1), -1: tap absent
    unsigned long long* trace;       // diagnostics (RD_WGRAD_BF16_TRACE=1 compute wave, =2 loader wave): 32 stamps per workgroup
    int trace_loader;
    int xcd;                         // XCD-aware workgroup order (diagnostics: RD_WGRAD_BF16_NOXCD=1 disables)
};

constexpr int WB_R = 4;                      // rows per pixel tile
constexpr int WB_TW = 32;                    // columns per pixel tile
constexpr int WB_XROW = 24;                  // dwords per staged X row: 48 bf16 columns, column c0 of the tile at index 8
// dwords per channel plane = 4 * odd: planes stay 16-byte aligned AND the 16 lanes of a b128 read pass (consecutive channels,
// same pixel) fall into 16 different bank groups (a pitch of 144 / 64 dwords made these reads 8- / 32-way conflicted)
constexpr int WB_XPLANE = (WB_R + 2) * WB_XROW + 4;
constexpr int WB_YROW = 16;                  // dwords per staged dY row (32 bf16 columns)
constexpr int WB_YPLANE = WB_R * WB_YROW + 4;
constexpr int WB_XP0 = 3, WB_XPN = 18;       // staged column pairs of X: 3..20 = plane columns 6..41 (image columns c0-2 .. c0+33)

// 32 bytes through a raw buffer descriptor: offsets at or beyond num_records return zeros (halo / padding without branches)
constexpr unsigned WB_OOB = 0x80000000u;
__device__ __forceinline__ void wb_load8(__amdgpu_buffer_rsrc_t r, unsigned off, float4& v0, float4& v1) {
    v0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    v1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off + 16, 0, 0));
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// FULL: all nine taps present (stride-1 3x3, UpProj phase 0) -> no tap tests in the MFMA phase; NK: k-parts = 4 / tile pairs
// IO16: x and dy are bf16 tensors (bf16-storage plans): a unit is one 16-byte load per pixel and the staging waves only interleave
// the pixel pair (no conversion) -- the staging waves are this kernel's critical path, so this is where bf16 storage pays twice.
template <bool FULL, int NK, bool IO16>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_bf16_kernel(const WgradBfArgs a) {
    constexpr int ESZ = IO16 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned wsm[];
    // eight waves, two roles: waves 0-3 own the accumulators and issue the MFMAs, waves 4-7 stage the NEXT tile meanwhile
    // (global loads -> registers -> bf16 -> LDS); one barrier per tile hands the buffers over
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int tid = threadIdx.x & 255;        // index inside the role
    const int l31 = lane & 31, hh = lane >> 5;
    int n_stamp = 0;
#define WB_STAMP() \
    if (a.trace && threadIdx.x == (unsigned)(a.trace_loader ? 256 : 0) && n_stamp < 31) a.trace[(size_t)blockIdx.x * 32 + 1 + n_stamp++] = __builtin_readcyclecounter();
    WB_STAMP()
    const int nblk = a.n_cib * a.n_cob;
    // consecutive logical ids = the channel blocks of ONE pixel split: they read the same x / dy tiles, so they are mapped to the
    // same XCD (hardware deals workgroup ids round-robin over the eight XCDs, each with its own L2)
    const int vid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int blk = vid % nblk;
    const int sp_ = vid / nblk;
    const int pass = __builtin_amdgcn_readfirstlane(sp_ / a.n_splits), split = sp_ - pass * a.n_splits;
    const int xa = a.xa[pass], xb = a.xb[pass], ya = a.ya[pass], yb = a.yb[pass];
    unsigned tapmask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
        if (a.slab_of_tap[pass][t] >= 0) tapmask |= 1u << t;
    const int cib = blk / a.n_cob, cob = blk - cib * a.n_cob;
    const int ci0 = cib * 32 * a.cpi, co0 = cob * 32 * a.cpo;
    const int n_pairs = a.cpi * a.cpo;
    constexpr int n_kparts = NK;
    const int pair = wave % n_pairs, kpart = wave / n_pairs;
    const int wci = pair / a.cpo, wco = pair - wci * a.cpo;
    const int ncgi = a.cpi * 4, ncgo = a.cpo * 4;      // 8-channel groups of the block

    unsigned* XT = wsm;                                // [32*cpi][WB_XPLANE]
    unsigned* YT = wsm + 32 * a.cpi * WB_XPLANE;       // [32*cpo][WB_YPLANE]

    const int t_lo = split * a.tiles_per_split, t_hi = min(t_lo + a.tiles_per_split, a.total_tiles);
    const int tiles_img = a.tiles_h * a.tiles_w;
    const int buf_dwords = 32 * a.cpi * WB_XPLANE + 32 * a.cpo * WB_YPLANE;       // one (XT, YT) buffer pair

    // ---- staging, split in two so that a tile's global loads fly under the previous tile's MFMAs (one workgroup per CU, 512
    // registers per lane: the accumulators and a whole tile of staged activations are live together):
    //   fetch: every thread's share of the X patch (UX units) and of the dY tile (UY units) into registers; unit = (8-channel
    //          group, row, column PAIR) = 2 pixels x 8 channels = 64 bytes; halo / padding through out-of-range buffer offsets;
    //   put  : convert to bf16, pack the pixel pair per dword and write one dword per channel plane (consecutive lanes =
    //          consecutive column pairs: conflict-free).
    constexpr int UX = 4, UY = 2;             // 4*256 >= 8 groups * 6 rows * 18 pairs, 2*256 >= 8 groups * 4 rows * 16 pairs
    constexpr int XPER = (WB_R + 2) * WB_XPN, YPER = WB_R * WB_YROW;
    const int nux = ncgi * XPER, nuy = ncgo * YPER;
    // unit u -> (8-channel group, row, column pair): four channel groups vary fastest, so that four neighbouring lanes read the
    // four 32-byte pieces of one pixel's 128-byte line (16 lines per load instruction instead of 64), then the pairs of a row
    auto xunit = [&](int u, int& cg, int& rr, int& pp) {
        const int pos = u >> 2, cgh = pos / XPER, rem = pos - cgh * XPER;
        cg = cgh * 4 + (u & 3);
        rr = rem / WB_XPN;
        pp = rem - rr * WB_XPN + WB_XP0;
        return u < nux;
    };
    auto yunit = [&](int u, int& cg, int& rr, int& pp) {
        const int pos = u >> 2, cgh = pos / YPER, rem = pos - cgh * YPER;
        cg = cgh * 4 + (u & 3);
        rr = rem / WB_YROW;
        pp = rem - rr * WB_YROW;
        return u < nuy;
    };
    auto fetch = [&](int tile, float4 (&vx)[UX][4], float4 (&vy)[UY][4]) {
        const int n = tile / tiles_img, tr = tile - n * tiles_img;
        const int th = tr / a.tiles_w, tw = tr - th * a.tiles_w;
        const int r0 = th * WB_R, c0 = tw * WB_TW;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(static_cast<const char*>(a.x) + (size_t)n * a.Hx * a.Wx * a.ldi * ESZ), 0, a.Hx * a.Wx * a.ldi * ESZ, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(static_cast<const char*>(a.dy) + (size_t)n * a.Hy * a.Wy * a.ldo * ESZ), 0, a.Hy * a.Wy * a.ldo * ESZ, 0x00020000);
        // one pixel's 8 channels: two float4 (fp32 storage) or 16 raw bytes in the first slot (bf16 storage)
        auto load_px = [&](__amdgpu_buffer_rsrc_t r, unsigned off, float4& v0, float4& v1) {
            if constexpr (IO16) v0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
            else wb_load8(r, off, v0, v1);
        };
#pragma unroll
        for (int q = 0; q < UX; ++q) {
            int cg, rr, pp;
            const bool live = xunit(tid + q * 256, cg, rr, pp);
            const int ih = a.IS * (r0 - 1 + rr) + xa, c = ci0 + cg * 8;
            const bool rowok = live && ih >= 0 && ih < a.Hx && c < a.Cin;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int iw = a.IS * (c0 + 2 * pp - 8 + px) + xb;
                const bool ok = rowok && iw >= 0 && iw < a.Wx;
                load_px(xr, ok ? (unsigned)(((ih * a.Wx + iw) * a.ldi + c) * ESZ) : WB_OOB, vx[q][2 * px], vx[q][2 * px + 1]);
            }
        }
#pragma unroll
        for (int q = 0; q < UY; ++q) {
            int cg, rr, pp;
            const bool live = yunit(tid + q * 256, cg, rr, pp);
            const int qr = r0 + rr, ih = a.OS * qr + ya, c = co0 + cg * 8;
            const bool rowok = live && qr < a.lh && ih < a.Hy && c < a.Cout;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int qc = c0 + 2 * pp + px, iw = a.OS * qc + yb;
                const bool ok = rowok && qc < a.lw && iw < a.Wy;
                load_px(yr, ok ? (unsigned)(((ih * a.Wy + iw) * a.ldo + c) * ESZ) : WB_OOB, vy[q][2 * px], vy[q][2 * px + 1]);
            }
        }
    };
    auto put8 = [&](unsigned* d, int plane, const float4 (&v)[4]) {
        if constexpr (IO16) {
            // v[0] / v[2]: the two pixels' eight bf16 channels as stored; dword k of the plane = (pixel 0 ch k, pixel 1 ch k)
            const wu32x4 p0 = __builtin_bit_cast(wu32x4, v[0]), p1 = __builtin_bit_cast(wu32x4, v[2]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[(2 * j) * plane] = (p0[j] & 0xffffu) | (p1[j] << 16);
                d[(2 * j + 1) * plane] = (p0[j] >> 16) | (p1[j] & 0xffff0000u);
            }
            return;
        }
        d[0 * plane] = pack2(v[0].x, v[2].x);
        d[1 * plane] = pack2(v[0].y, v[2].y);
        d[2 * plane] = pack2(v[0].z, v[2].z);
        d[3 * plane] = pack2(v[0].w, v[2].w);
        d[4 * plane] = pack2(v[1].x, v[3].x);
        d[5 * plane] = pack2(v[1].y, v[3].y);
        d[6 * plane] = pack2(v[1].z, v[3].z);
        d[7 * plane] = pack2(v[1].w, v[3].w);
    };
    auto put = [&](int buf, const float4 (&vx)[UX][4], const float4 (&vy)[UY][4]) {
        unsigned* xt = XT + buf * buf_dwords;
        unsigned* yt = YT + buf * buf_dwords;
#pragma unroll
        for (int q = 0; q < UX; ++q) {
            int cg, rr, pp;
            if (xunit(tid + q * 256, cg, rr, pp)) put8(xt + (cg * 8) * WB_XPLANE + rr * WB_XROW + pp, WB_XPLANE, vx[q]);
        }
#pragma unroll
        for (int q = 0; q < UY; ++q) {
            int cg, rr, pp;
            if (yunit(tid + q * 256, cg, rr, pp)) put8(yt + (cg * 8) * WB_YPLANE + rr * WB_YROW + pp, WB_YPLANE, vy[q]);
        }
    };

    if (loader) {
        // ---- staging role: tile i+1 while the other four waves run the MFMAs of tile i (same barrier sequence as below)
        if (t_lo < t_hi) {
            float4 vx[UX][4], vy[UY][4];
            fetch(t_lo, vx, vy);
            put(0, vx, vy);
        }
        int bs = 0;
        for (int tile = t_lo; tile < t_hi; ++tile, bs ^= 1) {
            rd_sync();
            WB_STAMP()
            if (tile + 1 < t_hi) {
                float4 vx[UX][4], vy[UY][4];
                fetch(tile + 1, vx, vy);
                WB_STAMP()
                put(bs ^ 1, vx, vy);
                WB_STAMP()
            }
        }
        if (a.trace && threadIdx.x == 256 && a.trace_loader) a.trace[(size_t)blockIdx.x * 32] = n_stamp;
        return;
    }

    // ---- MFMA role
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    int bsel = 0;
    for (int tile = t_lo; tile < t_hi; ++tile, bsel ^= 1) {
        rd_sync();      // buffer `bsel` is complete; the MFMAs of the previous tile are done with the other one
        WB_STAMP()
        const unsigned* XTc = XT + bsel * buf_dwords;
        const unsigned* YTc = YT + bsel * buf_dwords;
        // ---- k-steps of this wave: 16 consecutive pixels of one tile row; lane (l31, hh) holds channel l31 of its 32-channel
        // tile and pixels hh*8 .. hh*8+7 of the step
        const unsigned* xq = XTc + (wci * 32 + l31) * WB_XPLANE + 4 + hh * 4;     // dword of plane column 8 + hh*8 in row 0
        const unsigned* yq = YTc + (wco * 32 + l31) * WB_YPLANE + hh * 4;
        // Every patch row fragment is read (and its two shifted windows built) ONCE and feeds up to three tile rows: patch row
        // rr is tap row dh of tile row r = rr - dh.  The dY fragments of the wave's k-steps are read up front.
        wu32x4 Bf[2 * WB_R];
#pragma unroll
        for (int ks = 0; ks < 2 * WB_R; ++ks)
            if (NK == 1 || (ks % NK) == kpart) Bf[ks] = *reinterpret_cast<const wu32x4*>(yq + (ks >> 1) * WB_YROW + (ks & 1) * 8);
#pragma unroll
        for (int rr = 0; rr < WB_R + 2; ++rr) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned* p = xq + rr * WB_XROW + h * 8;
                const wu32x4 cur = *reinterpret_cast<const wu32x4*>(p);
                const unsigned prev = p[-1], next = p[4];
                wu32x4 lft, rgt;       // windows starting one pixel earlier / later
                lft[0] = __builtin_amdgcn_alignbyte(cur[0], prev, 2);
                lft[1] = __builtin_amdgcn_alignbyte(cur[1], cur[0], 2);
                lft[2] = __builtin_amdgcn_alignbyte(cur[2], cur[1], 2);
                lft[3] = __builtin_amdgcn_alignbyte(cur[3], cur[2], 2);
                rgt[0] = __builtin_amdgcn_alignbyte(cur[1], cur[0], 2);
                rgt[1] = __builtin_amdgcn_alignbyte(cur[2], cur[1], 2);
                rgt[2] = __builtin_amdgcn_alignbyte(cur[3], cur[2], 2);
                rgt[3] = __builtin_amdgcn_alignbyte(next, cur[3], 2);
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    const int r = rr - dh;
                    if (r < 0 || r >= WB_R) continue;                                   // compile time
                    const int ks = 2 * r + h;
                    if (NK != 1 && (ks % NK) != kpart) continue;                        // another wave's k-step (wave-uniform)
                    const wbf16x8 B = __builtin_bit_cast(wbf16x8, Bf[ks]);
                    if (FULL || (tapmask & (1u << (dh * 3 + 0))))
                        acc[dh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, lft), B, acc[dh * 3 + 0], 0, 0, 0);
                    if (FULL || (tapmask & (1u << (dh * 3 + 1))))
                        acc[dh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, cur), B, acc[dh * 3 + 1], 0, 0, 0);
                    if (FULL || (tapmask & (1u << (dh * 3 + 2))))
                        acc[dh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, rgt), B, acc[dh * 3 + 2], 0, 0, 0);
                }
            }
        }
        WB_STAMP()
    }

    // ---- this wave's partial slab: [tap][Cin][Cout], accumulator row = input channel, lane = output channel
    float* slab = a.slabs + (size_t)(split * n_kparts + kpart) * a.S * a.Cin * a.Cout;
    const int co = co0 + wco * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (!(tapmask & (1u << t))) continue;
        float* dst = slab + (size_t)a.slab_of_tap[pass][t] * a.Cin * a.Cout;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (ci < a.Cin && co < a.Cout) dst[(size_t)ci * a.Cout + co] = acc[t][i];
        }
    }
    WB_STAMP()
    if (a.trace && threadIdx.x == 0 && !a.trace_loader) a.trace[(size_t)blockIdx.x * 32] = n_stamp;
#undef WB_STAMP
}

// ------------------------------------------------------------------------------------------ host
struct WgradBfPlan {
    int cpi, cpo, n_cib, n_cob, tiles_h, tiles_w, total_tiles, tiles_per_split, n_splits, slab_splits;
    size_t lds_bytes;
    int S, lh, lw, n_pass;
    int xa[4], xb[4], ya[4], yb[4];
    int slab_of_tap[4][9];
};

static inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Decompose the (forward) descriptor into passes; false when it does not fit (then the caller keeps rd_wgrad).
static bool wgrad_bf16_plan(const RdConvDesc* d, WgradBfPlan& pl) {
    if (!d || d->n_phases < 1 || d->n_phases > RD_MAX_PHASES) return false;
    if (d->in_stride < 1 || d->in_stride > 2 || d->out_stride < 1 || d->out_stride > 2) return false;
    if (d->Cin % 16 != 0 || d->Cout % 16 != 0 || d->ldi % 4 != 0 || d->ldo % 4 != 0) return false;
    if ((int64_t)d->Hi * d->Wi * d->ldi * 4 >= (int64_t)WB_OOB || (int64_t)d->Ho * d->Wo * d->ldo * 4 >= (int64_t)WB_OOB) return false;
    const int IS = d->in_stride;
    pl.lh = d->phase[0].lh; pl.lw = d->phase[0].lw;
    pl.n_pass = 0;
    int S = 0;
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        if (p.lh != pl.lh || p.lw != pl.lw || p.n_taps < 1 || p.n_taps > RD_MAX_TAPS) return false;
        int pass_of[2][2] = {{-1, -1}, {-1, -1}};
        for (int t = 0; t < p.n_taps; ++t) {
            const int sh = floor_div(p.dh[t], IS), sw = floor_div(p.dw[t], IS);
            const int ra = p.dh[t] - sh * IS, rb = p.dw[t] - sw * IS;          // residues 0..IS-1
            if (sh < -1 || sh > 1 || sw < -1 || sw > 1 || p.widx[t] < 0) return false;
            int& ps = pass_of[ra][rb];
            if (ps < 0) {
                if (pl.n_pass == 4) return false;
                ps = pl.n_pass++;
                pl.xa[ps] = ra; pl.xb[ps] = rb; pl.ya[ps] = p.out_off_h; pl.yb[ps] = p.out_off_w;
                for (int k = 0; k < 9; ++k) pl.slab_of_tap[ps][k] = -1;
            }
            int& slot = pl.slab_of_tap[ps][(sh + 1) * 3 + (sw + 1)];
            if (slot >= 0) return false;                                         // two taps on the same shift
            slot = p.widx[t];
            S = S > p.widx[t] + 1 ? S : p.widx[t] + 1;
        }
    }
    // every slab must be produced exactly once (the reduction reads all S of every split)
    if (S < 1 || S > 25) return false;
    int seen[25] = {0};
    for (int ps = 0; ps < pl.n_pass; ++ps)
        for (int k = 0; k < 9; ++k)
            if (pl.slab_of_tap[ps][k] >= 0 && seen[pl.slab_of_tap[ps][k]]++) return false;
    for (int k = 0; k < S; ++k)
        if (!seen[k]) return false;
    pl.S = S;
    pl.cpi = d->Cin >= 64 ? 2 : 1;
    pl.cpo = d->Cout >= 64 ? 2 : 1;
    pl.n_cib = cdiv(d->Cin, 32 * pl.cpi);
    pl.n_cob = cdiv(d->Cout, 32 * pl.cpo);
    pl.tiles_h = cdiv(pl.lh, WB_R);
    pl.tiles_w = cdiv(pl.lw, WB_TW);
    pl.total_tiles = d->N * pl.tiles_h * pl.tiles_w;
    const int nblk = pl.n_cib * pl.n_cob * pl.n_pass;
    static const char* wgs_env = getenv("RD_WGRAD_BF16_WGS");      // diagnostics: target workgroup count
    const int target = wgs_env ? atoi(wgs_env) : 256;   // one workgroup per CU: fewer slabs to reduce (13.01 vs 13.18 ms/step at 512)
    int splits = cdiv(target, nblk);
    if (splits > pl.total_tiles) splits = pl.total_tiles;
    if (splits < 1) splits = 1;
    pl.tiles_per_split = cdiv(pl.total_tiles, splits);
    pl.n_splits = cdiv(pl.total_tiles, pl.tiles_per_split);
    pl.slab_splits = pl.n_splits * (4 / (pl.cpi * pl.cpo));
    pl.lds_bytes = (size_t)(32 * pl.cpi * WB_XPLANE + 32 * pl.cpo * WB_YPLANE) * 4 * 2;      // two tile buffers
    return true;
}

}  // namespace rd

using namespace rd;

static unsigned long long* g_wb_trace = nullptr;
// diagnostics: stamps of the last traced launch (32 slots per workgroup: count, cycle-counter values)
extern "C" int rd_wgrad_bf16_trace_read(unsigned long long* host, int n_wg) {
    if (!g_wb_trace) return RD_EINVAL;
    RD_CHECK_HIP(hipMemcpy(host, g_wb_trace, (size_t)n_wg * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_wgrad_bf16_supported(const RdConvDesc* d) {
    WgradBfPlan pl;
    return wgrad_bf16_plan(d, pl) ? 1 : 0;
}

extern "C" int64_t rd_wgrad_bf16_workspace_floats(const RdConvDesc* d) {
    WgradBfPlan pl;
    if (!wgrad_bf16_plan(d, pl)) { set_error("wgrad_bf16: unsupported descriptor"); return RD_EINVAL; }
    return (int64_t)(pl.slab_splits + 16) * pl.S * d->Cin * d->Cout;
}

// diagnostics: out[0..6] = ci tiles per block, co tiles per block, channel blocks, pixel splits, slabs, lds bytes, passes
extern "C" int rd_wgrad_bf16_plan_info(const RdConvDesc* d, int32_t* out) {
    WgradBfPlan pl;
    if (!wgrad_bf16_plan(d, pl)) return RD_EINVAL;
    out[0] = pl.cpi; out[1] = pl.cpo; out[2] = pl.n_cib * pl.n_cob; out[3] = pl.n_splits; out[4] = pl.slab_splits; out[5] = (int)pl.lds_bytes;
    out[6] = pl.n_pass;
    return RD_OK;
}

static int wgrad_bf16_impl(bool io16, const RdConvDesc* d, const void* in, const void* dout, float* slabs, void* stream) {
    RD_CHECK_ARG(d && in && dout && slabs, "wgrad_bf16: null argument");
    RD_CHECK_ARG(!io16 || (d->ldi % 8 == 0 && d->ldo % 8 == 0), "wgrad_bf16: bf16 storage needs channel strides that are multiples of 8");
    WgradBfPlan pl;
    if (!wgrad_bf16_plan(d, pl)) { set_error("wgrad_bf16: unsupported descriptor"); return RD_EINVAL; }
    WgradBfArgs a;
    a.x = in; a.dy = dout; a.slabs = slabs;
    a.N = d->N; a.Hx = d->Hi; a.Wx = d->Wi; a.Hy = d->Ho; a.Wy = d->Wo; a.lh = pl.lh; a.lw = pl.lw;
    a.Cin = d->Cin; a.Cout = d->Cout; a.ldi = d->ldi; a.ldo = d->ldo; a.IS = d->in_stride; a.OS = d->out_stride; a.S = pl.S;
    a.tiles_h = pl.tiles_h; a.tiles_w = pl.tiles_w; a.total_tiles = pl.total_tiles; a.tiles_per_split = pl.tiles_per_split;
    a.n_splits = pl.n_splits; a.n_cib = pl.n_cib; a.n_cob = pl.n_cob; a.cpi = pl.cpi; a.cpo = pl.cpo; a.n_pass = pl.n_pass;
    for (int ps = 0; ps < 4; ++ps) {
        a.xa[ps] = pl.xa[ps]; a.xb[ps] = pl.xb[ps]; a.ya[ps] = pl.ya[ps]; a.yb[ps] = pl.yb[ps];
        for (int t = 0; t < 9; ++t) a.slab_of_tap[ps][t] = ps < pl.n_pass ? pl.slab_of_tap[ps][t] : -1;
    }
    a.trace = nullptr;
    a.trace_loader = 0;
    { static const char* nox = getenv("RD_WGRAD_BF16_NOXCD"); a.xcd = nox ? 0 : 1; }
    {
        static const char* tr = getenv("RD_WGRAD_BF16_TRACE");
        if (tr && atoi(tr)) {
            a.trace_loader = atoi(tr) == 2;
            if (!g_wb_trace) RD_CHECK_HIP(hipMalloc(&g_wb_trace, (size_t)65536 * 32 * sizeof(unsigned long long)));
            a.trace = g_wb_trace;
        }
    }
    bool full = true;
    for (int ps = 0; ps < pl.n_pass; ++ps)
        for (int t = 0; t < 9; ++t) full = full && pl.slab_of_tap[ps][t] >= 0;
    const int nk = 4 / (pl.cpi * pl.cpo);
    const dim3 grid(pl.n_cib * pl.n_cob * pl.n_splits * pl.n_pass);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define RD_WB(FULL_, NK_) RD_WB2(FULL_, NK_, false) RD_WB2(FULL_, NK_, true)
#define RD_WB2(FULL_, NK_, IO_)                                                                                                 \
    if (full == FULL_ && nk == NK_ && io16 == IO_) {                                                                            \
        static std::atomic<unsigned long long> attr_set{0};                                                                                           \
        auto k = wgrad_bf16_kernel<FULL_, NK_, IO_>;                                                                            \
        RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(512), pl.lds_bytes, st, a);                                                            \
        RD_CHECK_LAUNCH("wgrad_bf16_kernel");                                                                                   \
        return RD_OK;                                                                                                           \
    }
    RD_WB(true, 1) RD_WB(true, 2) RD_WB(true, 4) RD_WB(false, 1) RD_WB(false, 2) RD_WB(false, 4)
#undef RD_WB
#undef RD_WB2
    set_error("wgrad_bf16: no kernel for %d k-parts", nk);
    return RD_EINVAL;
}
extern "C" int rd_wgrad_bf16(const RdConvDesc* d, const float* in, const float* dout, float* slabs, void* stream) {
    return wgrad_bf16_impl(false, d, in, dout, slabs, stream);
}
// storage-typed form: dtype = RD_DTYPE_BF16 -> in / dout are bf16 NHWC tensors (strides in elements); slabs stay fp32
extern "C" int rd_wgrad_bf16_t(int32_t dtype, const RdConvDesc* d, const void* in, const void* dout, float* slabs, void* stream) {
    RD_CHECK_ARG(dtype == RD_DTYPE_F32 || dtype == RD_DTYPE_BF16, "wgrad_bf16_t: bad dtype %d", dtype);
    return wgrad_bf16_impl(dtype == RD_DTYPE_BF16, d, in, dout, slabs, stream);
}

extern "C" int rd_wgrad_bf16_reduce(const RdConvDesc* d, const float* slabs, float* grad_oihw, int32_t O, int32_t I, int32_t KH,
                                    int32_t KW, int32_t co_off, int32_t accumulate, void* stream) {
    RD_CHECK_ARG(d && slabs && grad_oihw, "wgrad_bf16_reduce: null argument");
    WgradBfPlan pl;
    if (!wgrad_bf16_plan(d, pl)) { set_error("wgrad_bf16_reduce: unsupported descriptor"); return RD_EINVAL; }
    RD_CHECK_ARG(KH * KW == pl.S && I == d->Cin && co_off + O <= d->Cout, "wgrad_bf16_reduce: shape mismatch");
    const int64_t E = (int64_t)pl.S * d->Cin * d->Cout;
    float* tmp = const_cast<float*>(slabs) + (int64_t)pl.slab_splits * E;
    return launch_slab_reduce(slabs, pl.slab_splits, E, tmp, grad_oihw, pl.S, d->Cin, d->Cout, O, I, co_off, accumulate,
                              static_cast<hipStream_t>(stream));
}

// shape of this descriptor's slab reduction (rd_wgrad_reduce_job, wgrad.hip)
namespace rd {
bool wgrad_bf16_reduce_shape(const RdConvDesc* d, int* slab_splits, int* S) {
    WgradBfPlan pl;
    if (!wgrad_bf16_plan(d, pl)) return false;
    *slab_splits = pl.slab_splits;
    *S = pl.S;
    return true;
}
}  // namespace rd
