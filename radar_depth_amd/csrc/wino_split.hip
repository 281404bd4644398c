// Winograd F(2x2, 3x3) on the split-bf16 matrix pipeline (round 6): the 3x3 / stride-1 / pad-1 convolutions with
// >= 64 channels a side (forward, and the input gradient = the same kernel on 180-degree-rotated, transposed weights) with 16 instead of
// 36 multiplications per 2x2 output tile and (ci, co) pair -- 2.25x fewer of the v_mfma_f32_32x32x16_bf16 instructions the device is
// power-limited on.  It is what cuDNN runs for the reference's fp32 3x3 layers under cudnn.benchmark = True
// (/root/reference/main.py:11,47; layers: /root/reference/model/models.py:96-112).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
//   * B^T d B in fp32 VALU (adds only) by the staging waves, straight from global memory (NHWC: a lane owns one tile and two input
//     channels -- sixteen 8-byte loads per 16-channel chunk), then split into three bf16 pieces exactly as gconv_split.hip splits its
//     patch (x = x0 + x1 + x2 bitwise) and stored as the A operand of sixteen independent GEMMs [32 tiles x 16 ci] x [16 ci x 64 co];
//   * G g G^T once per step by rd_wino_pack (fp64, rounded once to fp32, three pieces), laid out so that the B image of a phase is
//     one linear 48 KB global_load_lds copy;
//   * six MFMA terms per product, fp32 accumulation: gconv_split.hip's arithmetic (RD_SPLIT_TERMS);
//   * A^T m A in the epilogue: rows inside the wave that owns a column of the 4x4 position grid, columns across the four MFMA waves
//     through LDS; BatchNorm partial sums, residual addend and the NHWC store from there.
//
//   workgroup : 8 waves, ONE per CU.  Waves 0-3 own the accumulators -- wave w holds column j = w of the 4x4 position grid, all four
//               rows i, for 32 tiles x 64 output channels: 4 x 2 tiles of 32 x 32 = 128 accumulator registers -- and do nothing but
//               fragment reads and MFMAs; waves 4-7 stage the A operand.
//   operands  : A (transformed patch) through LDS, one 16-channel chunk = all sixteen positions per phase (48 KB, two buffers, one
//               barrier per chunk); B (transformed weights) straight from global memory / L2 into the MFMA waves' registers, two steps
//               (positions) ahead -- the first version copied it through LDS (48 KB per half-chunk phase, issued and awaited by the
//               staging waves inside the phase that preceded its use): the copy's latency was exposed in every phase and the kernel ran
//               at 0.85x of the direct split kernel (profiles/r06_wino_gate.txt).
//   staging   : a lane owns (tile, channel pair): sixteen 8-byte loads per chunk, fetched TWO chunks ahead (HBM latency under load is
//               ~3 k clocks, a chunk ~1.6 k), 64 adds, sixteen three-way splits, 48 ds_write_b32.
//   tile block: 4 x 8 Winograd tiles = 8 x 16 output pixels; odd sizes (113, 57, 29, 15) by masking the last tile row / column.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace rd {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));

constexpr int WN_TBH = 4, WN_TBW = 8;              // tiles per block
constexpr int WN_T = WN_TBH * WN_TBW;              // 32 tiles = one MFMA M tile
constexpr int WN_CB = 64;                          // output channels per workgroup
constexpr int WN_AH = 16 * 3 * 2 * 512;            // A image of a chunk: [pos 16][piece 3][k half 2][tile 32] x 16 B = 48 KB
constexpr int WN_BH = 16 * 3 * 2 * 1024;           // B operand of a chunk and a 64-channel block: [pos 16][piece 3][k half 2][co 64] x 16 B = 96 KB
constexpr int WN_BUF = WN_AH;
constexpr int WN_ZP = 68;                          // floats per (tile) row of the epilogue's exchange buffer (64 + 4: 16-byte aligned rows)
constexpr unsigned WN_OOB = 0x80000000u;

struct WinoArgs {
    const float* in;
    const unsigned short* u;      // rd_wino_pack's operand: [cot][chunk][WN_BH bytes]
    float* out;
    const float* addend;
    float* stat;
    int N, H, W, Cin, Cout, ldi, ldo, ld_add;
    int tiles_h, tiles_w, bh, bw, n_cot;
    // rd_wino_conv3x3_bnbwd: the launch is an input gradient whose epilogue also emits the BatchNorm-backward sums of the BatchNorm in front of
    // the convolution (bnb_x = that BatchNorm's input at the output pixels; stat then holds [tile][3][Cout]: sum g, sum g (x - mean))
    const float* bnb_x;
    const float* bnb_scale;
    const float* bnb_shift;
    const float* bnb_mean;
    int bnb_ld, bnb_act;
    int dbg;                      // diagnostics (RD_WINO_DEBUG; results are then garbage): 1 no MFMAs, 2 no split / A stores, 4 no weight loads, 8 no epilogue, 32 / 64 staging / MFMA waves idle, 128 pixel-block-major order, 256 no patch loads
};

__device__ __forceinline__ unsigned wn_cvt_pk(float a, float b) {
    wf32x2 v;
    v[0] = a;
    v[1] = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wbf16x2));
}

template <bool NOMMA>      // (diagnostics instantiation: no MFMAs -- a run-time test would split the basic blocks the load / MFMA schedule lives in)
__global__ __launch_bounds__(512) void wino_split_kernel(const WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int l31 = lane & 31, hh = lane >> 5;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int nb_img = a.bh * a.bw;
    // Many output-channel blocks (>= 256 channels): channel-block-major order -- xcd_remap hands every XCD a contiguous range of ids, so an XCD
    // then works on ONE or two 64-channel blocks' weights (3.1 MB at 512 channels: inside its 4 MB L2) over all pixel blocks, instead of every
    // XCD streaming all 25 MB of transformed weights out of the memory-side cache.  Few blocks: pixel-block-major (neighbours share halos).
    const int n_pt = a.N * nb_img;
    const bool cot_major = a.n_cot >= 4 && !(a.dbg & 128);
    const int cot = cot_major ? vid / n_pt : vid % a.n_cot;
    const int pt = cot_major ? vid - cot * n_pt : vid / a.n_cot;
    const int n = pt / nb_img;
    const int rem = pt - n * nb_img;
    const int by = rem / a.bw, bx = rem - by * a.bw;
    const int nchunks = a.Cin >> 4;
    // Every workgroup walks the 16-channel chunks in its own rotation (first chunk = workgroup index mod the chunk count): workgroups that
    // run side by side would otherwise ask the L2 for the SAME 96 KB weight block at the same moment -- one channel serving 32 CUs
    // (the "2-4 k clocks for a copy to land" of gconv_split.hip, with 5x its weight traffic per output pixel).  The summation order over the
    // chunks therefore depends on the tile block; it is fixed for a given geometry (run-to-run bit-identical).
    // MEASURED (profiles/r06_wino_gate.txt): 512-channel layer 134 -> 223 us, 256-channel 137 -> 147, 64-channel unchanged -- rotated, the chip
    // touches every chunk's weights at once (25 MB at 512 channels: out of the 4 MB L2s), in lockstep it streams them.  Off; dbg 16 turns it on.
    const int c_rot = (a.dbg & 16) ? vid % nchunks : 0;
    auto rot = [&](int c) { const int cc = c + c_rot; return cc >= nchunks ? cc - nchunks : cc; };

    f32x16 acc[4][2];

    if (loader) {
        // ---------------------------------------------------------------------------------------------- staging waves
        const int sw = wave - 4;
        const int tl = (sw & 1) * 16 + (lane >> 2);      // tile of this lane
        const int q4 = lane & 3, kh = sw >> 1;           // channel pair q4 of k half kh: channels kh * 8 + q4 * 2, + 1 of the chunk
        const int ty = by * WN_TBH + (tl >> 3), tx = bx * WN_TBW + (tl & 7);
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
        unsigned off[16];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int iy = iy0 + r, ix = ix0 + c;
                const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                off[r * 4 + c] = ok ? (unsigned)(((iy * a.W + ix) * a.ldi + kh * 8 + q4 * 2) * 4) : WN_OOB;
            }
        const char* in_n = reinterpret_cast<const char*>(a.in) + (size_t)n * a.H * a.W * a.ldi * 4;
        const unsigned img_bytes = (unsigned)(a.H * a.W * a.ldi) * 4u;
        const unsigned a_dst = (unsigned)(size_t)wsm + (unsigned)(kh * 512 + tl * 16 + q4 * 4);       // inside one (position, piece) plane pair

        wf32x2 raw[2][16];
        auto fetch = [&](int c, wf32x2 (&rw)[16]) {
            if (a.dbg & 256) {      // diagnostics: no patch loads (every value = the lane's tile index)
#pragma unroll
                for (int p = 0; p < 16; ++p) { rw[p][0] = (float)tl; rw[p][1] = (float)(p + c); }
                return;
            }
            const int cc = rot(c);
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_n + cc * 64), 0, img_bytes - cc * 64, 0x00020000);
#pragma unroll
            for (int p = 0; p < 16; ++p) rw[p] = __builtin_bit_cast(wf32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off[p], 0, 0));
        };
        // split the two channels' values of position p into three packed pieces and store them
        auto put = [&](unsigned abuf, int p, wf32x2 v) {
            float x = v[0], y = v[1];
            const unsigned u0 = wn_cvt_pk(x, y);
            x -= __uint_as_float(u0 << 16);
            y -= __uint_as_float(u0 & 0xffff0000u);
            const unsigned u1 = wn_cvt_pk(x, y);
            x -= __uint_as_float(u1 << 16);
            y -= __uint_as_float(u1 & 0xffff0000u);
            const unsigned u2 = wn_cvt_pk(x, y);
            const unsigned ad = abuf + p * (3 * 1024);
            asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(u0) : "memory");
            asm volatile("ds_write_b32 %0, %1 offset:1024" ::"v"(ad), "v"(u1) : "memory");
            asm volatile("ds_write_b32 %0, %1 offset:2048" ::"v"(ad), "v"(u2) : "memory");
        };
        // B^T d B of a fetched chunk, split and stored into A buffer `slot`
        auto transform = [&](const wf32x2 (&rw)[16], int slot) {
            const unsigned abuf = a_dst + slot * WN_BUF;
            wf32x2 t[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = rw[0 * 4 + c] - rw[2 * 4 + c];
                t[1][c] = rw[1 * 4 + c] + rw[2 * 4 + c];
                t[2][c] = rw[2 * 4 + c] - rw[1 * 4 + c];
                t[3][c] = rw[1 * 4 + c] - rw[3 * 4 + c];
            }
            if (a.dbg & 2) {
                // (diagnostics: keep the adds alive without the split arithmetic and the stores)
                wf32x2 acc_ = t[0][0];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc_ += t[i][c];
                if (acc_[0] == 12345.678f) put(abuf, 0, acc_);
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                put(abuf, i * 4 + 0, t[i][0] - t[i][2]);
                put(abuf, i * 4 + 1, t[i][1] + t[i][2]);
                put(abuf, i * 4 + 2, t[i][2] - t[i][1]);
                put(abuf, i * 4 + 3, t[i][1] - t[i][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (a.dbg & 32) {      // diagnostics: the staging waves only keep the barriers
            for (int c = 0; c <= nchunks; ++c) __builtin_amdgcn_s_barrier();
            goto epilogue;
        }
        // prologue: chunk 0 into buffer 0; chunks 1 and 2 in flight
        fetch(0, raw[0]);
        if (nchunks > 1) fetch(1, raw[1]);
        if (nchunks > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        transform(raw[0], 0);
        if (nchunks > 2) fetch(2, raw[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // interval c: the compute waves read buffer c & 1; buffer (c + 1) & 1 <- chunk c + 1 (fetched two intervals ago); chunk c + 3 goes out
        auto interval = [&](int c, auto par_) {
            constexpr int par = decltype(par_)::value;      // == (c + 1) & 1: the register set AND the buffer of chunk c + 1
            if (c + 1 < nchunks) {
                if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // chunk c + 1 is in, chunk c + 2 may still be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                transform(raw[par], par);
                if (c + 3 < nchunks) fetch(c + 3, raw[par]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        int c = 0;
        for (; c + 2 <= nchunks; c += 2) {
            interval(c, std::integral_constant<int, 1>{});
            interval(c + 1, std::integral_constant<int, 0>{});
        }
        if (c < nchunks) interval(c, std::integral_constant<int, 1>{});
    } else {
        // ---------------------------------------------------------------------------------------------- compute waves
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][nb][k] = 0.f;
        if (a.dbg & 64) {      // diagnostics: the compute waves only keep the barriers
            for (int c = 0; c <= nchunks; ++c) __builtin_amdgcn_s_barrier();
            goto epilogue;
        }
        const int a_off = hh * 512 + l31 * 16 + wave * (3 * 1024);
        // B fragments of step (chunk c, row i): position i * 4 + wave of the chunk's block, 3 pieces x 2 N tiles, 16 bytes per lane each
        const char* u_lane = reinterpret_cast<const char*>(a.u) + (size_t)cot * nchunks * WN_BH + wave * (3 * 2048) + hh * 1024 + l31 * 16;
        const int nsteps = 4 * nchunks;
        // (branch-free: steps beyond the last re-read the last step's fragments, never used; dbg 4: every chunk reads chunk 0's block --
        //  L2 / L1 hits instead of fresh lines.  A conditional here makes the compiler wait for ALL outstanding loads at every step.)
        const size_t ustride = (a.dbg & 4) ? 0 : (size_t)WN_BH;
        auto loadB = [&](int step, wbf16x8 (&B)[3][2]) {
            step = min(step, nsteps - 1);
            const char* src = u_lane + (size_t)rot(step >> 2) * ustride + (step & 3) * (4 * 3 * 2048);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) B[p][nb] = *reinterpret_cast<const wbf16x8*>(src + p * 2048 + nb * 512);
        };
        wbf16x8 Bq[3][3][2];      // ring of three steps: B of step s lives in Bq[s % 3]
        wbf16x8 A[2][3];
        loadB(0, Bq[0]);
        loadB(1, Bq[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // twelve steps (three chunks) per trip: ring index and accumulator row are compile-time
        auto step = [&](const char* abuf, int st, auto i_, auto ring_) {
            constexpr int i = decltype(i_)::value, ring = decltype(ring_)::value;
            // this step's A fragments (the phase's first are read behind its barrier, the others one step ahead -- see below)
            loadB(st + 2, Bq[(ring + 2) % 3]);
            if (i < 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p) A[(i + 1) & 1][p] = *reinterpret_cast<const wbf16x8*>(abuf + a_off + ((i + 1) * 4 * 3 + p) * 1024);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x16 cc = acc[i][nb];
                if constexpr (NOMMA) {
                    cc[0] += (float)A[i & 1][0][0] + (float)Bq[ring][0][nb][0] + (float)A[i & 1][1][0] + (float)Bq[ring][1][nb][0] + (float)A[i & 1][2][0] + (float)Bq[ring][2][nb][0];
                } else {
                    RD_SPLIT_TERMS(cc, A[i & 1][0], A[i & 1][1], A[i & 1][2], Bq[ring][0][nb], Bq[ring][1][nb], Bq[ring][2][nb])
                }
                acc[i][nb] = cc;
            }
        };
        auto chunk = [&](int c, auto r0_) {
            constexpr int r0 = decltype(r0_)::value;       // ring index of the chunk's first step = (4 c) % 3
            const char* abuf = wsm + (c & 1) * WN_BUF;
#pragma unroll
            for (int p = 0; p < 3; ++p) A[0][p] = *reinterpret_cast<const wbf16x8*>(abuf + a_off + p * 1024);
            step(abuf, 4 * c + 0, std::integral_constant<int, 0>{}, std::integral_constant<int, (r0 + 0) % 3>{});
            step(abuf, 4 * c + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, (r0 + 1) % 3>{});
            step(abuf, 4 * c + 2, std::integral_constant<int, 2>{}, std::integral_constant<int, (r0 + 2) % 3>{});
            step(abuf, 4 * c + 3, std::integral_constant<int, 3>{}, std::integral_constant<int, (r0 + 3) % 3>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        int c = 0;
        for (; c + 3 <= nchunks; c += 3) {
            chunk(c, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{});
            chunk(c + 2, std::integral_constant<int, 2>{});
        }
        if (c < nchunks) { chunk(c, std::integral_constant<int, 0>{}); ++c; }
        if (c < nchunks) { chunk(c, std::integral_constant<int, 1>{}); ++c; }
    }
epilogue:
    if (a.dbg & 8) return;

    // ---- epilogue.  Rows of A^T m A inside the wave (Z_r = sum_i A^T[r][i] m_i, the wave's column j), columns across the waves:
    // Z goes to LDS as [j][r][tile][co], every thread then combines the four j of (tile, r, four channels): Y_r0 = Z0 + Z1 + Z2,
    // Y_r1 = Z1 - Z2 - Z3 -- two horizontally adjacent output pixels, 16-byte stores.
    float* zb = reinterpret_cast<float*>(wsm);
    if (!loader) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x16 z0 = acc[0][nb] + acc[1][nb] + acc[2][nb];
            const f32x16 z1 = acc[1][nb] - acc[2][nb] - acc[3][nb];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = 8 * (k >> 2) + 4 * hh + (k & 3);
                zb[((wave * 2 + 0) * WN_T + m) * WN_ZP + nb * 32 + l31] = z0[k];
                zb[((wave * 2 + 1) * WN_T + m) * WN_ZP + nb * 32 + l31] = z1[k];
            }
        }
    }
    rd_sync();
    const int c4 = tid & 15;                 // channel quad of this thread (both of its items)
    const int co = cot * WN_CB + c4 * 4;
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;
    const bool bnb = a.bnb_x != nullptr;
    float4 bS = ssum, bT = ssum, bM = ssum;
    if (bnb && co < a.Cout) { bS = ld4(a.bnb_scale + co); bT = ld4(a.bnb_shift + co); bM = ld4(a.bnb_mean + co); }
    // statistics of one stored pixel: its values and squares, or (bnb) g = y * act'(scale x + shift) and g (x - mean)
    auto account = [&](const float4 y, size_t px) {
        if (bnb) {
            const float4 xv = ld4(a.bnb_x + px * a.bnb_ld + co);
            const float gx = y.x * act_grad_from_out(fmaf(bS.x, xv.x, bT.x), a.bnb_act), gy = y.y * act_grad_from_out(fmaf(bS.y, xv.y, bT.y), a.bnb_act);
            const float gz = y.z * act_grad_from_out(fmaf(bS.z, xv.z, bT.z), a.bnb_act), gw = y.w * act_grad_from_out(fmaf(bS.w, xv.w, bT.w), a.bnb_act);
            ssum.x += gx; ssum.y += gy; ssum.z += gz; ssum.w += gw;
            ssq.x += gx * (xv.x - bM.x); ssq.y += gy * (xv.y - bM.y); ssq.z += gz * (xv.z - bM.z); ssq.w += gw * (xv.w - bM.w);
        } else {
            ssum.x += y.x; ssum.y += y.y; ssum.z += y.z; ssum.w += y.w;
            ssq.x += y.x * y.x; ssq.y += y.y * y.y; ssq.z += y.z * y.z; ssq.w += y.w * y.w;
        }
    };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int rt = (tid >> 4) + 32 * it;   // 0..63 = tile * 2 + r
        const int m = rt >> 1, r = rt & 1;
        float4 z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = *reinterpret_cast<const float4*>(zb + ((j * 2 + r) * WN_T + m) * WN_ZP + c4 * 4);
        float4 y0 = make_float4(z[0].x + z[1].x + z[2].x, z[0].y + z[1].y + z[2].y, z[0].z + z[1].z + z[2].z, z[0].w + z[1].w + z[2].w);
        float4 y1 = make_float4(z[1].x - z[2].x - z[3].x, z[1].y - z[2].y - z[3].y, z[1].z - z[2].z - z[3].z, z[1].w - z[2].w - z[3].w);
        const int oy = 2 * (by * WN_TBH + (m >> 3)) + r, ox = 2 * (bx * WN_TBW + (m & 7));
        const bool okr = oy < a.H && co < a.Cout;
        const bool ok0 = okr && ox < a.W, ok1 = okr && ox + 1 < a.W;
        const size_t pix = ((size_t)n * a.H + oy) * a.W + ox;
        if (a.addend) {
            if (ok0) { const float4 v = ld4(a.addend + pix * a.ld_add + co); y0.x += v.x; y0.y += v.y; y0.z += v.z; y0.w += v.w; }
            if (ok1) { const float4 v = ld4(a.addend + (pix + 1) * a.ld_add + co); y1.x += v.x; y1.y += v.y; y1.z += v.z; y1.w += v.w; }
        }
        if (ok0) {
            st4(a.out + pix * a.ldo + co, y0);
            if (a.stat) account(y0, pix);
        }
        if (ok1) {
            st4(a.out + (pix + 1) * a.ldo + co, y1);
            if (a.stat) account(y1, pix + 1);
        }
    }
    if (a.stat) {
        // BatchNorm partial sums of this workgroup's 128 pixels x 64 channels: [pt][2][Cout] like the other convolution kernels
        rd_sync();                                     // (every thread is done reading the exchange buffer)
        float* red = reinterpret_cast<float*>(wsm);    // [32 thread rows][2][64]
        const int row = tid >> 4;
        *reinterpret_cast<float4*>(red + (row * 2 + 0) * 64 + c4 * 4) = ssum;
        *reinterpret_cast<float4*>(red + (row * 2 + 1) * 64 + c4 * 4) = ssq;
        rd_sync();
        if (tid < 128) {
            const int which = tid >> 6, j = tid & 63;
            float s = 0.f;
#pragma unroll 8
            for (int rw = 0; rw < 32; ++rw) s += red[(rw * 2 + which) * 64 + j];
            if (cot * WN_CB + j < a.Cout) a.stat[((size_t)pt * (bnb ? 3 : 2) + which) * a.Cout + cot * WN_CB + j] = s;
        }
    }
}

// U = G g G^T of every (reduction channel, output channel) pair in fp64, rounded once to fp32, split into three bf16 pieces, in the
// layout the MFMA waves read as fragments: [cot][chunk][pos 16][piece 3][k half 2][co 64][8 reduction channels].
// flip = 0: forward (reduction = the I of OIHW, output = O); flip = 1: input gradient (reduction = O, output = I, taps rotated by 180 degrees).
__device__ __forceinline__ void wino_pack_body(const float* __restrict__ w, int O, int I, int flip, unsigned short* __restrict__ u, int e) {
    const int R = flip ? O : I, Q = flip ? I : O;
    const int r8n = R >> 3;
    if (e >= Q * r8n) return;
    const int q = e / r8n, r8 = e - q * r8n;
    const int nck = R >> 4;
    const int cot = q >> 6, ql = q & 63;
    const int chunk = r8 >> 1, kh8 = r8 & 1;
    unsigned short pc[16][3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = r8 * 8 + k;
        const float* g = flip ? w + ((size_t)r * I + q) * 9 : w + ((size_t)q * I + r) * 9;
        double gg[3][3];
#pragma unroll
        for (int y = 0; y < 3; ++y)
#pragma unroll
            for (int x = 0; x < 3; ++x) gg[y][x] = flip ? (double)g[(2 - y) * 3 + (2 - x)] : (double)g[y * 3 + x];
        double t[4][3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            t[0][x] = gg[0][x];
            t[1][x] = 0.5 * (gg[0][x] + gg[1][x] + gg[2][x]);
            t[2][x] = 0.5 * (gg[0][x] - gg[1][x] + gg[2][x]);
            t[3][x] = gg[2][x];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double uu[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = (float)uu[j];
                const __bf16 p0 = (__bf16)f;
                f -= (float)p0;
                const __bf16 p1 = (__bf16)f;
                f -= (float)p1;
                const __bf16 p2 = (__bf16)f;
                pc[i * 4 + j][0][k] = __builtin_bit_cast(unsigned short, p0);
                pc[i * 4 + j][1][k] = __builtin_bit_cast(unsigned short, p1);
                pc[i * 4 + j][2][k] = __builtin_bit_cast(unsigned short, p2);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const size_t o = ((size_t)cot * nck + chunk) * WN_BH + (((i * 4 + j) * 3 + p) * 2 + kh8) * 1024 + ql * 16;
                uint4 v;
                v.x = pc[i * 4 + j][p][0] | ((unsigned)pc[i * 4 + j][p][1] << 16);
                v.y = pc[i * 4 + j][p][2] | ((unsigned)pc[i * 4 + j][p][3] << 16);
                v.z = pc[i * 4 + j][p][4] | ((unsigned)pc[i * 4 + j][p][5] << 16);
                v.w = pc[i * 4 + j][p][6] | ((unsigned)pc[i * 4 + j][p][7] << 16);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(u) + o) = v;
            }
}

__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, int O, int I, int flip, unsigned short* __restrict__ u) {
    wino_pack_body(w, O, I, flip, u, blockIdx.x * blockDim.x + threadIdx.x);
}

// one launch for every Winograd operand of a plan (rd_wino_pack_batched): block -> job through a block table, like rd_pack_weights_batched
struct WinoPackJob {
    const float* w;
    unsigned short* u;
    int32_t O, I, flip, first_block;
};
__global__ __launch_bounds__(256) void wino_pack_batched_kernel(const WinoPackJob* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    const WinoPackJob j = jobs[block_job[blockIdx.x]];
    wino_pack_body(j.w, j.O, j.I, j.flip, j.u, (blockIdx.x - j.first_block) * blockDim.x + threadIdx.x);
}

static bool wino_shape_ok(int H, int W, int Cin, int Cout, int ldi, int ldo) {
    return H >= 2 && W >= 2 && Cin >= 64 && Cin % 16 == 0 && Cout >= 64 && Cout % 64 == 0 && ldi % 4 == 0 && ldo % 4 == 0 &&
           (long long)H * W * ldi * 4 < 0x7fffffffll;
}
}  // namespace rd

// bytes of the packed operand for an O x I x 3 x 3 weight tensor (flip as in rd_wino_pack)
extern "C" int64_t rd_wino_packed_bytes(int32_t O, int32_t I, int32_t flip) {
    const int R = flip ? O : I, Q = flip ? I : O;
    return (int64_t)((Q + 63) / 64) * (R / 16) * rd::WN_BH;
}

extern "C" int rd_wino_pack(const float* w_oihw, int32_t O, int32_t I, int32_t flip, void* u_packed, void* stream) {
    RD_CHECK_ARG(w_oihw && u_packed && O > 0 && I > 0, "rd_wino_pack: bad arguments");
    const int R = flip ? O : I, Q = flip ? I : O;
    RD_CHECK_ARG(R % 16 == 0 && Q % 64 == 0, "rd_wino_pack: reduction channels %d %% 16, output channels %d %% 64", R, Q);
    const int n = Q * (R / 8);
    hipLaunchKernelGGL(rd::wino_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw, O, I, flip,
                       static_cast<unsigned short*>(u_packed));
    RD_CHECK_LAUNCH("wino_pack_kernel");
    return RD_OK;
}

// jobs: device array of { const float* w; void* u; int32 O, I, flip, first_block } (24 bytes + 8 of pointers = 32-byte records);
// block_job[b] = index of the job block b works on; a job of an O x I tensor takes ceil(Q * (R / 8) / 256) blocks (rd_wino_pack_blocks)
extern "C" int rd_wino_pack_blocks(int32_t O, int32_t I, int32_t flip) {
    const int R = flip ? O : I, Q = flip ? I : O;
    return (Q * (R / 8) + 255) / 256;
}
extern "C" int rd_wino_pack_batched(const void* jobs, const int32_t* block_job, int32_t n_blocks, void* stream) {
    RD_CHECK_ARG(jobs && block_job && n_blocks > 0, "rd_wino_pack_batched: bad arguments");
    static_assert(sizeof(rd::WinoPackJob) == 32, "job record layout");
    hipLaunchKernelGGL(rd::wino_pack_batched_kernel, dim3(n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const rd::WinoPackJob*>(jobs), block_job);
    RD_CHECK_LAUNCH("wino_pack_batched_kernel");
    return RD_OK;
}

// Planner rule, from the kernel-level gate at b = 16 (profiles/r06_wino_gate.txt): the Winograd form beats the direct split kernels
// where the direct kernels are weakest -- 512-channel layers (1.30x) and the small maps with <= 128 channels (29 x 50 / 30 x 50 / 15 x 25:
// 1.32-1.41x) -- and loses or ties on the large maps (64 channels at 113 x 200: 0.94x; 128 at 57 x 100: 1.10x; 256 at 29 x 50: 1.00x),
// where its 16 / 9 larger weight operand (96 KB per 16-channel chunk and 32-tile block, all of it through the CU's 64 B/clk vector L1)
// and the un-coalesced 4 x 4 patch gathers cost more than the 2.25x fewer MFMAs save.  RD_WINO=0 never, RD_WINO=all wherever supported.
extern "C" int rd_wino_preferred(int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ldi, int32_t ldo) {
    if (!rd::wino_shape_ok(H, W, Cin, Cout, ldi, ldo)) return 0;
    static const char* env = getenv("RD_WINO");
    if (env && !strcmp(env, "0")) return 0;
    if (env && !strcmp(env, "all")) return 1;
    return (Cin >= 512 || ((long long)H * W <= 1536 && Cin <= 128)) ? 1 : 0;
}

extern "C" int rd_wino_supported(int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ldi, int32_t ldo) {
    return rd::wino_shape_ok(H, W, Cin, Cout, ldi, ldo) ? 1 : 0;
}

// rows of the [tiles][2][Cout] BatchNorm partial-sum buffer rd_wino_conv3x3 writes
extern "C" int rd_wino_stat_tiles(int32_t N, int32_t H, int32_t W) {
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    return N * ((th + rd::WN_TBH - 1) / rd::WN_TBH) * ((tw + rd::WN_TBW - 1) / rd::WN_TBW);
}

static int wino_launch(const float* in, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldi, const void* u_packed, float* out,
                       int32_t Cout, int32_t ldo, const float* addend, int32_t ld_add, float* stat_partial, void* stream, const float* bn_x,
                       int32_t bn_ld, const float* bn_mean, const float* bn_scale, const float* bn_shift, int32_t bn_act) {
    using namespace rd;
    RD_CHECK_ARG(in && u_packed && out && N > 0, "rd_wino_conv3x3: bad arguments");
    RD_CHECK_ARG(wino_shape_ok(H, W, Cin, Cout, ldi, ldo), "rd_wino_conv3x3: unsupported shape %dx%d %d->%d (ld %d / %d)", H, W, Cin, Cout, ldi, ldo);
    RD_CHECK_ARG(!addend || ld_add % 4 == 0, "rd_wino_conv3x3: addend stride %d", ld_add);
    WinoArgs a;
    a.in = in;
    a.u = static_cast<const unsigned short*>(u_packed);
    a.out = out;
    a.addend = addend;
    a.stat = stat_partial;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldi = ldi; a.ldo = ldo; a.ld_add = ld_add;
    a.bnb_x = bn_x; a.bnb_ld = bn_ld; a.bnb_mean = bn_mean; a.bnb_scale = bn_scale; a.bnb_shift = bn_shift; a.bnb_act = bn_act;
    RD_CHECK_ARG(!bn_x || (bn_mean && bn_scale && bn_shift && stat_partial && !addend && bn_ld % 4 == 0), "rd_wino_conv3x3_bnbwd: bad arguments");
    a.tiles_h = (H + 1) / 2;
    a.tiles_w = (W + 1) / 2;
    a.bh = (a.tiles_h + WN_TBH - 1) / WN_TBH;
    a.bw = (a.tiles_w + WN_TBW - 1) / WN_TBW;
    a.n_cot = Cout / WN_CB;
    static const int dbg = getenv("RD_WINO_DEBUG") ? atoi(getenv("RD_WINO_DEBUG")) : 0;
    a.dbg = dbg;
    const int grid = N * a.bh * a.bw * a.n_cot;
    const size_t lds = 2 * (size_t)WN_BUF;
    static std::atomic<unsigned long long> attr_set{0};
    if (dbg & 1) {
        static std::atomic<unsigned long long> attr_dbg{0};
        RD_SET_ATTR_ONCE(attr_dbg, hipFuncSetAttribute(reinterpret_cast<const void*>(wino_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(wino_split_kernel<true>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
        RD_CHECK_LAUNCH("wino_split_kernel");
        return RD_OK;
    }
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(wino_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(wino_split_kernel<false>, dim3(grid), dim3(512), lds, static_cast<hipStream_t>(stream), a);
    RD_CHECK_LAUNCH("wino_split_kernel");
    return RD_OK;
}

extern "C" int rd_wino_conv3x3(const float* in, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldi, const void* u_packed, float* out,
                               int32_t Cout, int32_t ldo, const float* addend, int32_t ld_add, float* stat_partial, void* stream) {
    return wino_launch(in, N, H, W, Cin, ldi, u_packed, out, Cout, ldo, addend, ld_add, stat_partial, stream, nullptr, 0, nullptr, nullptr, nullptr, 0);
}

// The input gradient of a Winograd layer (u_packed = the flipped operand) that also emits the backward sums of the BatchNorm in front of the
// convolution -- rd_gconv_bnbwd's contract: bn_x [N,H,W,Cout] (channel stride bn_ld) = that BatchNorm's input, red_partial
// [rd_wino_stat_tiles(N,H,W)][3][Cout]: slot 0 = sum g, slot 1 = sum g (x - mean), g = dx * act'(scale x + shift).
extern "C" int rd_wino_conv3x3_bnbwd(const float* in, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldi, const void* u_packed, float* out,
                                     int32_t Cout, int32_t ldo, const float* bn_x, int32_t bn_ld, const float* mean, const float* scale,
                                     const float* shift, int32_t bn_act, float* red_partial, void* stream) {
    RD_CHECK_ARG(bn_x != nullptr, "rd_wino_conv3x3_bnbwd: null argument");
    return wino_launch(in, N, H, W, Cin, ldi, u_packed, out, Cout, ldo, nullptr, 0, red_partial, stream, bn_x, bn_ld, mean, scale, shift, bn_act);
}
