// Weight gradient of the generalised convolution on the matrix cores:
//   dW[slab][ci][co] = sum over (n, oh, ow) of in[n, oh*IS+dh, ow*IS+dw, ci] * dout[n, oh*OS+off_h, ow*OS+off_w, co]
// i.e. a GEMM whose K dimension is the pixel index.  Replaces the weight half of ATen's
// convolution_backward for every conv of the reference's step (autograd of model/models.py:96-112,203-206,
// 652-657; ~50 % of the reference's CPU step time).
//
// A workgroup owns a (ci-block, co-block, tap-group) of dW and a contiguous range of pixel tiles
// (split-K over pixels).  Per tile it stages the input halo patch [pixels][CIB] and the dout tile
// [pixels][COB] in LDS; each wave keeps TG accumulator tiles (one per tap of its group) and walks the
// tile's pixels 2 (32x32x2) or 4 (16x16x4) at a time: A[ci][pixel] and B[pixel][co] fragments are both
// contiguous-in-channel ds_read_b32 (conflict-free).  Partial results go to per-split slabs that
// rd_wgrad_reduce sums in a fixed order (deterministic, no atomics) straight into OIHW gradients.
#include <stdlib.h>

#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace rd {

// compile-time loop: f(std::integral_constant<int, I>) for I = B, B+ST, ... < E
template <int B, int E, int ST, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + ST, E, ST>(f);
    }
}

constexpr int WG_MAX_TG = 9;     // taps per group
constexpr int WG_MAX_GROUPS = 5;
constexpr int WG_MAX_PIX = 512;  // logical pixels per tile
constexpr int WG_PITCH = 27;     // patch pitch of the immediate-offset 3x3 kernel: 25-wide tiles (stride 1), 13-wide (stride 2)

struct WgTapGroup {
    int n;
    int8_t dh[WG_MAX_TG], dw[WG_MAX_TG], oh[WG_MAX_TG], ow[WG_MAX_TG];
    int16_t widx[WG_MAX_TG];
};

struct WgradArgs {
    const float* in;
    const float* dout;
    float* slabs;
    int N, Hi, Wi, Cin, ldi, Ho, Wo, Cout, ldo, IS, OS;
    int lh, lw, TH, TW, tiles_h, tiles_w, total_tiles, tiles_per_split, n_splits;
    int dh_min, dh_max, dw_min, dw_max;
    int n_cib, n_cob, n_tg;
    int S;  // slabs (= kernel taps) per split
    // compact != 0: one phase of a strided-output stencil (UpProj): only this phase's dout pixels (oh*OS+off_h, ow*OS+off_w)
    // are staged, densely, in LDS
    int compact, off_h, off_w;
    int debug;  // ablation bits (RD_WGRAD_DEBUG): 1 skip staging after the first tile, 2 skip the MFMA walk
    WgTapGroup tg[WG_MAX_GROUPS];
};

// LAYOUT_A: waves arranged 2 (ci) x 2 (co), all see every pixel.  Otherwise: one (ci,co) block, the four
// waves take interleaved pixel groups and each writes its own slab.
// SHB: every tap of the group reads the same dout pixel (stride-1/2 k x k convs) -> one shared B fragment per step.
// PITCH > 0: the group is a full rectangular stencil, RW taps per row (taps in row-major order: 3x3, or the 3x2 / 2x3 / 2x2
// sub-stencils of the UpProj phases) and the LDS patch row pitch is the compile-time
// PITCH, so the nine tap offsets are instruction immediates: the walk then costs 2 VALU instructions per 9 MFMAs instead of
// 16 (fp32 MFMAs share the SIMD's fp32 lanes with the VALU: every VALU instruction in the walk is lost MFMA time, see
// tools/micro/mfma_mix.hip).
template <int TG, int MF, bool LAYOUT_A, bool SHB, int PITCH, int RW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_kernel(const WgradArgs a) {
    constexpr int CIB = LAYOUT_A ? 64 : MF;
    constexpr int COB = LAYOUT_A ? 64 : MF;
    constexpr int KP = MF == 32 ? 2 : 4;   // pixels per MFMA
    constexpr int WLP = LAYOUT_A ? 1 : 4;  // waves splitting the pixel walk
    constexpr int NACC = MF == 32 ? 16 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & (MF - 1), lk = lane / MF;
    const int wci = LAYOUT_A ? (wave >> 1) : 0, wco = LAYOUT_A ? (wave & 1) : 0;
    const int wpix = LAYOUT_A ? 0 : wave;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_groups = a.n_cib * a.n_cob * a.n_tg;
    const int g = vid % n_groups, split = vid / n_groups;
    const int tgi = g % a.n_tg;
    const int cob0 = ((g / a.n_tg) % a.n_cob) * COB;
    const int cib0 = (g / (a.n_tg * a.n_cob)) * CIB;
    const WgTapGroup& G = a.tg[tgi];

    const int PWmax = PITCH > 0 ? PITCH : (a.TW - 1) * a.IS + (a.dw_max - a.dw_min) + 1;
    const int PHmax = (a.TH - 1) * a.IS + (a.dh_max - a.dh_min) + 1;
    const int LOS = a.compact ? 1 : a.OS;          // dout pixel stride inside LDS
    const int DW = a.TW * LOS, DHmax = a.TH * LOS;
    int2* s_tab = reinterpret_cast<int2*>(smem);                        // [WG_MAX_PIX + 64]
    float* s_in = smem + 2 * (WG_MAX_PIX + 64);                         // [PHmax*PWmax][CIB]
    float* s_do = s_in + (size_t)PHmax * PWmax * CIB;                    // [DHmax*DW + 1][COB], last row zero (SHB padding)
    const int zero_row = DHmax * DW;
    if (tid < COB) s_do[(size_t)zero_row * COB + tid] = 0.f;

    // per-tap LDS offsets (loop invariant)
    int tin[TG], tout[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        tin[t] = (PITCH > 0 ? ((t / RW) * PITCH + (t % RW)) * CIB : ((G.dh[t] - a.dh_min) * PWmax + (G.dw[t] - a.dw_min)) * CIB) + wci * 32 + lm;
        tout[t] = (G.oh[t] * DW + G.ow[t]) * COB + wco * 32 + lm;
    }

    typedef float accv __attribute__((ext_vector_type(NACC)));
    accv acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[t][i] = 0.f;

    const int tile_begin = split * a.tiles_per_split;
    const int tile_end = min(tile_begin + a.tiles_per_split, a.total_tiles);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int n = tile / (a.tiles_h * a.tiles_w);
        const int trem = tile - n * (a.tiles_h * a.tiles_w);
        const int r0 = (trem / a.tiles_w) * a.TH, c0 = (trem % a.tiles_w) * a.TW;
        const int th_n = min(a.TH, a.lh - r0), tw_n = min(a.TW, a.lw - c0);
        const int npix = th_n * tw_n;
        const int npad = ((npix + 2 * KP * WLP - 1) / (2 * KP * WLP)) * (2 * KP * WLP);   // even number of steps
        rd_sync();  // previous tile fully consumed
        if (!((a.debug & 1) && tile > tile_begin)) {
        for (int p = tid; p < npad + 3 * KP * WLP; p += 256) {
            int2 e = make_int2(0, SHB ? zero_row * COB : 0);  // padding: A reads a valid location, B reads zeros (SHB) or is masked
            if (p < npix) {
                const int r = p / tw_n, c = p - r * tw_n;
                e.x = ((r * a.IS) * PWmax + c * a.IS) * CIB;
                e.y = ((r * LOS) * DW + c * LOS) * COB;
            }
            s_tab[p] = e;
        }
        // Staging goes global -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16 B, which is
        // exactly the [pixel][channel] order both tiles have), so no registers are held and every load of the tile is in
        // flight at once: one memory round trip per tile.  Elements outside the image / tile / channel range are written
        // as zeros by their (exec-masked-out) lanes with an ordinary ds_write.
        {   // input halo patch
            const int PH = (th_n - 1) * a.IS + (a.dh_max - a.dh_min) + 1;
            const int ih0 = r0 * a.IS + a.dh_min, iw0 = c0 * a.IS + a.dw_min;
            const float* in_n = a.in + (size_t)n * a.Hi * a.Wi * a.ldi;
            constexpr int q4 = CIB / 4;
            const int elems = PH * PWmax * q4;
            for (int e = tid; e < elems; e += 256) {
                const int pix = e / q4, qq = e - pix * q4;
                const int py = pix / PWmax, px = pix - py * PWmax;
                const int ih = ih0 + py, iw = iw0 + px, c = cib0 + qq * 4;
                if (ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi && c < a.Cin)
                    glds16(in_n + ((size_t)ih * a.Wi + iw) * a.ldi + c, s_in + (size_t)(e - lane) * 4);
                else
                    *reinterpret_cast<float4*>(s_in + (size_t)e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);   // [pix][CIB] is linear in e
            }
        }
        {   // dout tile (rows/cols beyond the image or beyond the tile's valid extent are zero)
            const int DH = th_n * LOS;
            const int gst = a.compact ? a.OS : 1;          // global pixel step per LDS pixel
            const int oh0 = r0 * a.OS + (a.compact ? a.off_h : 0), ow0 = c0 * a.OS + (a.compact ? a.off_w : 0);
            const float* do_n = a.dout + (size_t)n * a.Ho * a.Wo * a.ldo;
            constexpr int q4 = COB / 4;
            const int elems = DH * DW * q4;
            const int px_lim = tw_n * LOS;
            for (int e = tid; e < elems; e += 256) {
                const int pix = e / q4, qq = e - pix * q4;
                const int py = pix / DW, px = pix - py * DW;
                const int oh = oh0 + py * gst, ow = ow0 + px * gst, c = cob0 + qq * 4;
                if (oh < a.Ho && ow < a.Wo && px < px_lim && c < a.Cout)
                    glds16(do_n + ((size_t)oh * a.Wo + ow) * a.ldo + c, s_do + (size_t)(e - lane) * 4);
                else
                    *reinterpret_cast<float4*>(s_do + (size_t)e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        }
        glds_wait();
        rd_sync();
        // Software pipeline pinned with sched_barrier (hipcc otherwise sinks every ds_read next to its MFMA and waits
        // lgkmcnt(0) per instruction).  Two register sets ping-pong (no copy moves): while one set's TG MFMAs issue, the
        // other set's fragments are in flight from LDS and the pixel-table entry of the step after is being fetched.
        constexpr int NB = SHB ? 1 : TG;
        const int nsteps = npad / (KP * WLP);          // even
        float a0[TG], b0[NB], a1[TG], b1[NB];
        bool pv0 = true, pv1 = true;
        (void)pv0; (void)pv1;
        int2 e_nxt;
#define RD_WG_LOAD(AV, BV, PV, STEP)                                          \
        {                                                                         \
            const int2 e_ = e_nxt;                                                \
            if constexpr (PITCH > 0) {                                            \
                const float* pa_ = s_in + (e_.x + wci * 32 + lm);                 \
                _Pragma("unroll") for (int t = 0; t < TG; ++t) AV[t] = pa_[((t / RW) * PITCH + (t % RW)) * CIB]; \
            } else {                                                              \
                _Pragma("unroll") for (int t = 0; t < TG; ++t) AV[t] = s_in[e_.x + tin[t]]; \
            }                                                                     \
            _Pragma("unroll") for (int t = 0; t < NB; ++t) BV[t] = s_do[e_.y + tout[t]]; \
            const int pn_ = (((STEP) + 1) * WLP + wpix) * KP + lk;                \
            if (!SHB) PV = (((STEP)) * WLP + wpix) * KP + lk < npix;              \
            e_nxt = s_tab[pn_];                                                   \
        }
#define RD_WG_MFMA(AV, BV, PV)                                                \
        _Pragma("unroll") for (int t = 0; t < TG; ++t) {                          \
            const float bsel_ = SHB ? BV[0] : (PV ? BV[t < NB ? t : 0] : 0.f);    \
            if constexpr (MF == 32)                                               \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[t], bsel_, acc[t], 0, 0, 0); \
            else                                                                  \
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[t], bsel_, acc[t], 0, 0, 0); \
        }
        // one half-iteration: address VALU first, then MFMAs with the LDS reads of the next step slotted between them
        // (an LDS read issued between two MFMAs costs ~3.5 clk of MFMA time, ~10 clk when issued in a block)
#define RD_WG_SCHED()                                                         \
        if constexpr (PITCH > 0) {                                                \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                    \
            _Pragma("unroll") for (int i = 2; i < TG; ++i) {                      \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                \
            }                                                                     \
        }                                                                         \
        __builtin_amdgcn_sched_barrier(0);
        e_nxt = s_tab[wpix * KP + lk];
        RD_WG_LOAD(a0, b0, pv0, 0)
        __builtin_amdgcn_sched_barrier(0);
        for (int st = 0; st < ((a.debug & 2) ? 0 : nsteps); st += 2) {
            if constexpr (PITCH > 0) {
                RD_WG_LOAD(a1, b1, pv1, st + 1)
                RD_WG_MFMA(a0, b0, pv0)
                RD_WG_SCHED()
                RD_WG_LOAD(a0, b0, pv0, st + 2)
                RD_WG_MFMA(a1, b1, pv1)
                RD_WG_SCHED()
            } else {
                RD_WG_LOAD(a1, b1, pv1, st + 1)
                __builtin_amdgcn_sched_barrier(0);
                RD_WG_MFMA(a0, b0, pv0)
                __builtin_amdgcn_sched_barrier(0);
                RD_WG_LOAD(a0, b0, pv0, st + 2)
                __builtin_amdgcn_sched_barrier(0);
                RD_WG_MFMA(a1, b1, pv1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef RD_WG_LOAD
#undef RD_WG_MFMA
#undef RD_WG_SCHED
    }

    // ---- write this workgroup's partial slab
    float* slab = a.slabs + (size_t)split * a.S * a.Cin * a.Cout;
    const int co = cob0 + wco * 32 + lm;
    if constexpr (LAYOUT_A) {
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            float* dst = slab + (size_t)G.widx[t] * a.Cin * a.Cout;
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const int ci = cib0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
                if (ci < a.Cin && co < a.Cout) dst[(size_t)ci * a.Cout + co] = acc[t][i];
            }
        }
    } else {
        // the four waves walked disjoint pixels of the same (ci,co) block: combine them through LDS, one tap
        // at a time; wave w then owns accumulator rows i == w (mod 4)
        float* red = smem;  // [4][NACC][64]
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            rd_sync();
#pragma unroll
            for (int i = 0; i < NACC; ++i) red[(wave * NACC + i) * 64 + lane] = acc[t][i];
            rd_sync();
            float* dst = slab + (size_t)G.widx[t] * a.Cin * a.Cout;
#pragma unroll
            for (int ii = 0; ii < NACC / 4; ++ii) {
                const int i = ii * 4 + wave;
                const float v = red[(0 * NACC + i) * 64 + lane] + red[(1 * NACC + i) * 64 + lane] +
                                red[(2 * NACC + i) * 64 + lane] + red[(3 * NACC + i) * 64 + lane];
                const int row = MF == 32 ? ((i & 3) + 8 * (i >> 2) + 4 * lk) : (lk * 4 + i);
                const int ci = cib0 + row;
                if (ci < a.Cin && co < a.Cout) dst[(size_t)ci * a.Cout + co] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Strip kernel for rectangular RH x RW stencils at unit input stride on 64-channel blocks (every 3x3 conv with C >= 64 and
// the UpProj phase sub-stencils): the workgroup walks DOWN a 25-pixel-wide column strip.  Input rows and dout rows live in
// two LDS rings; while the MFMAs of output rows (2i, 2i+1) issue, the two input rows and two dout rows of iteration i+1 are
// already in flight into their ring slots (global_load_lds: no staging registers), so the memory latency that cost the tiled
// kernel 12 % of its time (RD_WGRAD_DEBUG=1 ablation) is covered, vertical halo rows are never re-staged, and there is one
// barrier per 234 MFMAs.  All LDS offsets of the 26 steps of an iteration are instruction immediates relative to RH+1 row
// base registers: the walk issues no VALU instruction at all besides those bases.
struct WgradStripArgs {
    int seg_rows, n_segs, n_strips, n_units, units_per_split;
};
template <int RH, int RW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_strip_kernel(const WgradArgs a, const WgradStripArgs sa) {
    constexpr int TG = RH * RW, CIB = 64, COB = 64;
    constexpr int TWS = 25, PW = WG_PITCH, DW = 26;        // strip width, patch row pixels, dout row slots (slot 25 stays zero)
    constexpr int RP = 6, RD = 4;                          // ring depths in rows
    // The 13th step of a row multiplies the zero pad pixel of the dout row (slot 25) by patch pixel 25 + dw, i.e. pixel 27 for
    // dw = 2: one pixel past the 27 staged ones.  It must be FINITE (0 * NaN = NaN), so every ring row carries a 28th pixel that
    // is zeroed once per workgroup and never written again (uninitialised LDS can hold NaN bit patterns from other kernels:
    // under concurrency this showed up as whole taps of the gradient turning NaN).
    constexpr int PWR = PW + 1;
    constexpr int PROW = PWR * CIB, DROW = DW * COB;       // floats per ring row
    constexpr int STEPS = (TWS + 1) / 2;                   // 13 two-pixel steps per output row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                                    // [RP][PW][CIB]
    float* s_do = smem + RP * PROW;                        // [RD][DW][COB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 31, lk = lane >> 5;
    const int wci = wave >> 1, wco = wave & 1;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_groups = a.n_cib * a.n_cob;
    const int g = vid % n_groups, split = vid / n_groups;
    const int cob0 = (g % a.n_cob) * COB, cib0 = (g / a.n_cob) * CIB;
    const WgTapGroup& G = a.tg[0];

    typedef float accv __attribute__((ext_vector_type(16)));
    accv acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int i = tid; i < RP * CIB; i += 256) s_in[(i / CIB) * PROW + PW * CIB + (i % CIB)] = 0.f;     // the pad pixel of every ring row
    const int laneA = lk * CIB + wci * 32 + lm, laneB = lk * COB + wco * 32 + lm;
    const int gst = a.compact ? a.OS : 1;                  // global dout pixel step per logical pixel
    const int u_begin = split * sa.units_per_split, u_end = min(u_begin + sa.units_per_split, sa.n_units);
    for (int unit = u_begin; unit < u_end; ++unit) {
        const int seg = unit % sa.n_segs, uj = unit / sa.n_segs;
        const int strip = uj % sa.n_strips, n = uj / sa.n_strips;
        const int r_b = seg * sa.seg_rows, r_e = min(a.lh, r_b + sa.seg_rows);
        const int c0 = strip * TWS, tw_n = min(TWS, a.lw - c0);
        const int niter = (r_e - r_b + 1) >> 1;
        const int ih_base = r_b + a.dh_min, iw_base = c0 + a.dw_min;
        const float* in_n = a.in + (size_t)n * a.Hi * a.Wi * a.ldi;
        const float* do_n = a.dout + (size_t)n * a.Ho * a.Wo * a.ldo;
        // this thread's share of one staged row (fixed for the unit): element e = tid + k*256 -> (pixel, channel quad)
        int p_off[2], d_off[2];        // float offset inside the global row, -1: write zeros
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * 256;
            const int px = e >> 4, q = e & 15;
            const int iw = iw_base + px, ci = cib0 + q * 4;
            p_off[k] = (e < PW * 16 && iw >= 0 && iw < a.Wi && ci < a.Cin) ? iw * a.ldi + ci : -1;
            const int ow = (c0 + px) * gst + (a.compact ? a.off_w : 0), co = cob0 + q * 4;
            d_off[k] = (e < DW * 16 && px < tw_n && ow < a.Wo && co < a.Cout) ? ow * a.ldo + co : -1;
        }
        auto stage_patch_row = [&](int p) {                // p: patch row of the unit (0 = input row ih_base)
            const int ih = ih_base + p;
            const bool rowok = ih >= 0 && ih < a.Hi;
            float* dst = s_in + (p % RP) * PROW;
            const float* src = in_n + (size_t)ih * a.Wi * a.ldi;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = tid + k * 256;
                if (e < PW * 16) {
                    if (rowok && p_off[k] >= 0) glds16(src + p_off[k], dst + (e - lane) * 4);
                    else *reinterpret_cast<float4*>(dst + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto stage_dout_row = [&](int q) {                 // q: output row of the unit (0 = logical row r_b)
            const int r = r_b + q;
            const int oh = r * gst + (a.compact ? a.off_h : 0);
            const bool rowok = r < r_e && oh < a.Ho;
            float* dst = s_do + (q % RD) * DROW;
            const float* src = do_n + (size_t)oh * a.Wo * a.ldo;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = tid + k * 256;
                if (e < DW * 16) {
                    if (rowok && d_off[k] >= 0) glds16(src + d_off[k], dst + (e - lane) * 4);
                    else *reinterpret_cast<float4*>(dst + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        rd_sync();                                   // the previous unit's walk is over: the rings are free
        for (int p = 0; p <= RH; ++p) stage_patch_row(p);
        stage_dout_row(0);
        stage_dout_row(1);
        for (int it = 0; it < niter; ++it) {
            glds_wait();
            rd_sync();                               // rows of iteration `it` have landed; iteration it-1 is fully consumed
            if (it + 1 < niter && !(a.debug & 1)) {
                stage_patch_row(2 * it + RH + 1);
                stage_patch_row(2 * it + RH + 2);
                stage_dout_row(2 * it + 2);
                stage_dout_row(2 * it + 3);
            }
            if (a.debug & 2) continue;
            // row bases of the two output rows of this iteration
            const float* ab[2][RH];
            const float* bb[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
                for (int dh = 0; dh < RH; ++dh) ab[rr][dh] = s_in + ((2 * it + rr + dh) % RP) * PROW + laneA;
                bb[rr] = s_do + ((2 * it + rr) % RD) * DROW + laneB;
            }
            float a0[TG], a1[TG], b0, b1;
#define RD_WS_LOAD(AV, BV, S_)                                                                     \
            {                                                                                          \
                constexpr int rr_ = (S_) / STEPS, sc_ = (S_) % STEPS;                                  \
                _Pragma("unroll") for (int t = 0; t < TG; ++t) AV[t] = ab[rr_][t / RW][((t % RW) + 2 * sc_) * CIB]; \
                BV = bb[rr_][2 * sc_ * COB];                                                           \
            }
#define RD_WS_MFMA(AV, BV)                                                                         \
            _Pragma("unroll") for (int t = 0; t < TG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[t], BV, acc[t], 0, 0, 0);
#define RD_WS_SCHED()                                                                              \
            _Pragma("unroll") for (int i = 0; i < TG; ++i) {                                           \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                     \
            }                                                                                          \
            __builtin_amdgcn_sched_barrier(0);
            RD_WS_LOAD(a0, b0, 0)
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 2 * STEPS, 2>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                RD_WS_LOAD(a1, b1, S + 1)
                RD_WS_MFMA(a0, b0)
                RD_WS_SCHED()
                if constexpr (S + 2 < 2 * STEPS) { RD_WS_LOAD(a0, b0, S + 2) }
                RD_WS_MFMA(a1, b1)
                RD_WS_SCHED()
            });
#undef RD_WS_LOAD
#undef RD_WS_MFMA
#undef RD_WS_SCHED
        }
    }

    // ---- write this workgroup's partial slab (same layout as the tiled kernel)
    float* slab = a.slabs + (size_t)split * a.S * a.Cin * a.Cout;
    const int co = cob0 + wco * 32 + lm;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        float* dst = slab + (size_t)G.widx[t] * a.Cin * a.Cout;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = cib0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;
            if (ci < a.Cin && co < a.Cout) dst[(size_t)ci * a.Cout + co] = acc[t][i];
        }
    }
}

// Deterministic two-stage reduction of the per-split slabs.
// stage 1: tmp[j][e] = sum over splits k == j (mod J) of slabs[k][e]        (J partial sums, wide parallelism)
// stage 2: grad[o][i][t] (+)= sum_j tmp[j][(t*Cin+i)*Cout + co_off + o]
__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float* __restrict__ slabs, float* __restrict__ tmp,
                                                            int n_splits, int J, int64_t E4) {
    const int j = blockIdx.y;
    const float4* src = reinterpret_cast<const float4*>(slabs);
    float4* dst = reinterpret_cast<float4*>(tmp);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int k = j; k < n_splits; k += J) {
            const float4 v = src[(int64_t)k * E4 + e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        dst[(int64_t)j * E4 + e] = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce2_kernel(const float* __restrict__ tmp, float* __restrict__ grad, int J,
                                                            int64_t E, int S, int Cin, int Cout, int O, int I, int co_off,
                                                            int accumulate) {
    const int64_t total = (int64_t)S * I * O;
    const bool small = total < (1ll << 31);          // 32-bit divisions (a 64-bit one is ~100 instructions, three per element)
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int o, i, t;
        if (small) {
            const unsigned u = (unsigned)e, r = u / (unsigned)O;
            o = (int)(u - r * (unsigned)O);
            t = (int)(r / (unsigned)I);
            i = (int)(r - (unsigned)t * (unsigned)I);
        } else {
            o = (int)(e % O);
            const int64_t r = e / O;
            i = (int)(r % I);
            t = (int)(r / I);
        }
        const float* src = tmp + ((int64_t)t * Cin + i) * Cout + co_off + o;
        float s = 0.f;
        for (int k = 0; k < J; ++k) s += src[k * E];
        float* dst = grad + ((int64_t)o * I + i) * S + t;
        *dst = accumulate ? *dst + s : s;
    }
}

// stage 1 (once per slab set, when co_off == 0: callers reduce column ranges in increasing co_off order) + stage 2
int launch_slab_reduce(const float* slabs, int n_splits, int64_t E, float* tmp, float* grad_oihw, int S, int Cin, int Cout,
                       int O, int I, int co_off, int accumulate, hipStream_t s) {
    RD_CHECK_ARG(E % 4 == 0, "slab_reduce: slab size must be a multiple of 4");
    const int J = n_splits < 16 ? n_splits : 16;
    const int64_t E4 = E / 4;
    // up to 16 splits stage 1 would be a plain copy (tmp[j] = slabs[j]): stage 2 reads the slabs themselves, same order, same bits
    if (n_splits <= 16) tmp = const_cast<float*>(slabs);
    else if (co_off == 0) {
        int64_t g1 = cdiv64(E4, 256);
        if (g1 > 2048) g1 = 2048;
        hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((int)g1, J), dim3(256), 0, s, slabs, tmp, n_splits, J, E4);
        RD_CHECK_LAUNCH("wgrad_reduce1_kernel");
    }
    const int64_t total = (int64_t)S * I * O;
    int64_t g = cdiv64(total, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((int)g), dim3(256), 0, s, tmp, grad_oihw, J, E, S, Cin, Cout, O, I, co_off,
                       accumulate);
    RD_CHECK_LAUNCH("wgrad_reduce2_kernel");
    return RD_OK;
}

// ---- batched form: the reductions of MANY weight-gradient launches (one backward segment) in two launches.  A job is one
// rd_wgrad_reduce call; blocks are dealt to jobs through a block -> job table.  Same summation order as the single-job kernels
// above, hence the same bits (tests/test_gpu_wgrad.py).
__global__ __launch_bounds__(256) void wgrad_reduce1_batched_kernel(const RdReduceJob* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    const RdReduceJob jb = jobs[block_job[blockIdx.x]];
    const int lb = (int)blockIdx.x - jb.first_block1;
    const int J = jb.J, j = lb % J, bx = lb / J, nbx = jb.n_blocks1 / J;
    const int64_t E4 = jb.E / 4;
    const float4* src = reinterpret_cast<const float4*>(jb.slabs);
    float4* dst = reinterpret_cast<float4*>(jb.tmp);
    for (int64_t e = (int64_t)bx * 256 + threadIdx.x; e < E4; e += (int64_t)nbx * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int k = j; k < jb.n_splits; k += J) {
            const float4 v = src[(int64_t)k * E4 + e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        dst[(int64_t)j * E4 + e] = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce2_batched_kernel(const RdReduceJob* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    const RdReduceJob jb = jobs[block_job[blockIdx.x]];
    const int lb = (int)blockIdx.x - jb.first_block2;
    const unsigned total = (unsigned)jb.S * (unsigned)jb.I * (unsigned)jb.O;       // < 2^31 (checked when the job is made)
    const float* tmp = jb.n_splits <= 16 ? jb.slabs : jb.tmp;
    for (unsigned u = (unsigned)lb * 256u + threadIdx.x; u < total; u += (unsigned)jb.n_blocks2 * 256u) {
        const unsigned r = u / (unsigned)jb.O;
        const int o = (int)(u - r * (unsigned)jb.O);
        const int t = (int)(r / (unsigned)jb.I);
        const int i = (int)(r - (unsigned)t * (unsigned)jb.I);
        const float* src = tmp + ((int64_t)t * jb.Cin + i) * jb.Cout + jb.co_off + o;
        float s = 0.f;
        for (int k = 0; k < jb.J; ++k) s += src[k * jb.E];
        float* dst = jb.grad + ((int64_t)o * jb.I + i) * jb.S + t;
        *dst = jb.accumulate ? *dst + s : s;
    }
}

// Fills a job from the reduction's shape (block ranges are the caller's: rd_wgrad_reduce_job)
int make_reduce_job(const float* slabs, int n_splits, int64_t E, float* grad_oihw, int S, int Cin, int Cout, int O, int I, int co_off,
                    int accumulate, RdReduceJob* jb) {
    RD_CHECK_ARG(E % 4 == 0, "slab_reduce: slab size must be a multiple of 4");
    RD_CHECK_ARG((int64_t)S * I * O < (1ll << 31), "wgrad_reduce_job: one weight tensor must stay below 2^31 elements");
    const int J = n_splits < 16 ? n_splits : 16;
    jb->slabs = slabs; jb->tmp = const_cast<float*>(slabs) + (int64_t)n_splits * E; jb->grad = grad_oihw;
    jb->E = E; jb->n_splits = n_splits; jb->J = J; jb->S = S; jb->Cin = Cin; jb->Cout = Cout; jb->O = O; jb->I = I;
    jb->co_off = co_off; jb->accumulate = accumulate;
    int64_t g1 = cdiv64(E / 4, 256);
    if (g1 > 2048) g1 = 2048;
    jb->n_blocks1 = (n_splits > 16 && co_off == 0) ? (int)g1 * J : 0;      // stage 1 once per slab set (column ranges in increasing co_off order)
    int64_t g2 = cdiv64((int64_t)S * I * O, 256);
    if (g2 > 4096) g2 = 4096;
    jb->n_blocks2 = (int)g2;
    jb->first_block1 = jb->first_block2 = 0;
    jb->pad_ = 0;
    return RD_OK;
}

struct WgradPlan {
    int TG, MF, layoutA;
    int TH, TW, tiles_h, tiles_w, total_tiles, tiles_per_split, n_splits, slab_splits;
    int n_cib, n_cob, n_tg, S, J, shb;
    int pitch;      // > 0: fixed LDS patch pitch with immediate tap offsets (full rectangular stencil)
    int rw;         // taps per stencil row (pitch > 0)
    int per_phase;  // UpProj: one launch per output phase (each a rectangular sub-stencil with a shared dout pixel)
    int strip;      // column-strip kernel with LDS row rings (rectangular stencil, unit input stride, 64-channel blocks)
    int w16;        // served by wgrad16.hip (16 -> 16 channels, 3x3, unit strides, 16x16x4 MFMA): only S, n_splits, slab_splits, J are used
    int w1x1;       // served by wgrad1x1.hip (one tap, >= 64 channels each side): only S, n_splits, slab_splits, J are used
    WgradStripArgs sa;
    size_t lds;
};

static inline bool tile_is_tiled(const struct WgradPlan* tile);
// ph >= 0: plan one phase of a multi-phase descriptor on its own (compact dout staging); tile: reuse this plan's tile.
static int plan_wgrad(const RdConvDesc& d_in, WgradPlan& pl, WgradArgs* out, int ph = -1, const WgradPlan* tile = nullptr) {
    RdConvDesc d = d_in;
    int S_all = 0;
    for (int i = 0; i < d_in.n_phases; ++i)
        for (int t = 0; t < d_in.phase[i].n_taps; ++t) S_all = S_all > d_in.phase[i].widx[t] + 1 ? S_all : d_in.phase[i].widx[t] + 1;
    if (ph >= 0) {
        d.n_phases = 1;
        d.phase[0] = d_in.phase[ph];
    }
    const RdPhase& P0 = d.phase[0];
    int ntaps = 0, S = 0;
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int i = 0; i < d.n_phases; ++i) {
        const RdPhase& p = d.phase[i];
        RD_CHECK_ARG(p.lh == P0.lh && p.lw == P0.lw, "wgrad: phases must share one logical grid");
        ntaps += p.n_taps;
        for (int t = 0; t < p.n_taps; ++t) S = S > p.widx[t] + 1 ? S : p.widx[t] + 1;
        dh_min = dh_min < p.dh_min ? dh_min : p.dh_min;
        dh_max = dh_max > p.dh_max ? dh_max : p.dh_max;
        dw_min = dw_min < p.dw_min ? dw_min : p.dw_min;
        dw_max = dw_max > p.dw_max ? dw_max : p.dw_max;
    }
    S = S_all;
    // rectangular stencil in row-major tap order -> the immediate-offset kernel, whose patch pitch is fixed at WG_PITCH
    const int rw = dw_max - dw_min + 1;
    bool k3 = d.n_phases == 1 && ntaps == rw * (dh_max - dh_min + 1) && (ntaps == 9 || (ph >= 0 && (ntaps == 6 || ntaps == 4)));
    for (int t = 0; k3 && t < ntaps; ++t) k3 = P0.dh[t] - dh_min == t / rw && P0.dw[t] - dw_min == t % rw;
    static const char* nok3 = getenv("RD_WGRAD_NOK3");
    if (nok3) k3 = false;
    RD_CHECK_ARG(ntaps == 1 || ntaps == 9 || ntaps == 25 || k3, "wgrad: %d taps unsupported", ntaps);
    pl.rw = rw;
    pl.per_phase = 0;
    pl.w16 = 0;
    pl.w1x1 = 0;
    RD_CHECK_ARG(d.Cin % 4 == 0 && d.Cout % 4 == 0 && d.ldi % 4 == 0 && d.ldo % 4 == 0, "wgrad: channels must be multiples of 4");
    pl.shb = d.n_phases == 1;   // single phase: every tap pairs with the same dout pixel
    pl.TG = ntaps == 25 ? 5 : ntaps;
    pl.n_tg = ntaps == 25 ? 5 : 1;
    const int cmax = d.Cin > d.Cout ? d.Cin : d.Cout;
    pl.layoutA = cmax >= 64;
    pl.MF = (!pl.layoutA && cmax <= 16) ? 16 : 32;
    const int CIB = pl.layoutA ? 64 : pl.MF, COB = CIB;
    pl.n_cib = cdiv(d.Cin, CIB);
    pl.n_cob = cdiv(d.Cout, COB);
    pl.S = S;
    // tile: largest pixel count within the LDS budget, preferring full-width rows
    static const char* bud = getenv("RD_WGRAD_LDS_KB");   // diagnostics (default 78: two workgroups per CU)
    const size_t budget = (size_t)(bud ? atoi(bud) : 78) * 1024;
    pl.pitch = 0;
    pl.strip = 0;
    static const char* nostrip = getenv("RD_WGRAD_NOSTRIP");
    // (short images do not amortise the ring fill of a strip: 15-row layers measured 3 % slower than tiled)
    if (k3 && !nostrip && pl.layoutA && pl.MF == 32 && d.in_stride == 1 && (d.out_stride == 1 || ph >= 0) && P0.lh >= 24 && !tile_is_tiled(tile)) {
        // column-strip kernel: 25-pixel strips, row segments sized so that every CU gets two workgroups
        WgradStripArgs& sa = pl.sa;
        const int groups = pl.n_cib * pl.n_cob;
        static const char* wpc_env = getenv("RD_WGRAD_WG_PER_CU_X2");   // diagnostics: workgroups per CU, times two (default 4 = two per CU)
        // (the four phase launches of an UpProj layer each write their own slabs: with one workgroup per CU instead of two the
        //  slab traffic halves, which pays there -- dec1 327 -> 270 us, dec2 259 -> 251, dec3 260 -> 250, tools/sweep_wgrad_knobs.py --
        //  and nowhere else)
        int want = cdiv((wpc_env ? atoi(wpc_env) : (ph >= 0 ? 2 : 4)) * num_cus() / 2, groups);
        if (want < 1) want = 1;
        sa.n_strips = cdiv(P0.lw, 25);
        const int base_units = d.N * sa.n_strips;
        int n_segs0 = cdiv(want, base_units);
        if (n_segs0 < 1) n_segs0 = 1;
        sa.seg_rows = 2 * cdiv(cdiv(P0.lh, n_segs0), 2);
        sa.n_segs = cdiv(P0.lh, sa.seg_rows);
        sa.n_units = base_units * sa.n_segs;
        sa.units_per_split = cdiv(sa.n_units, want);
        pl.n_splits = cdiv(sa.n_units, sa.units_per_split);
        pl.slab_splits = pl.n_splits;
        pl.J = pl.n_splits < 16 ? pl.n_splits : 16;
        pl.strip = 1;
        pl.pitch = WG_PITCH;
        pl.TH = 2; pl.TW = 25; pl.tiles_h = pl.tiles_w = pl.total_tiles = pl.tiles_per_split = 0;
        pl.lds = (size_t)(6 * (WG_PITCH + 1) * 64 + 4 * 26 * 64) * 4;
        if (out) {
            WgradArgs& a = *out;
            a.N = d.N; a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.ldi = d.ldi;
            a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout; a.ldo = d.ldo; a.IS = d.in_stride; a.OS = d.out_stride;
            a.lh = P0.lh; a.lw = P0.lw; a.TH = pl.TH; a.TW = pl.TW; a.tiles_h = a.tiles_w = a.total_tiles = a.tiles_per_split = 0;
            a.n_splits = pl.n_splits;
            a.dh_min = dh_min; a.dh_max = dh_max; a.dw_min = dw_min; a.dw_max = dw_max;
            a.n_cib = pl.n_cib; a.n_cob = pl.n_cob; a.n_tg = 1; a.S = S;
            a.compact = ph >= 0; a.off_h = P0.out_off_h; a.off_w = P0.out_off_w;
            for (int gi = 0; gi < WG_MAX_GROUPS; ++gi) a.tg[gi].n = 0;
            WgTapGroup& G = a.tg[0];
            for (int t = 0; t < P0.n_taps; ++t) {
                G.dh[t] = P0.dh[t]; G.dw[t] = P0.dw[t]; G.oh[t] = 0; G.ow[t] = 0; G.widx[t] = P0.widx[t];
            }
            G.n = P0.n_taps;
        }
        return RD_OK;
    }
    const int los = ph >= 0 ? 1 : d.out_stride;      // dout pixel stride inside LDS (compact per-phase staging: 1)
    double best = -1;
    pl.TH = pl.TW = 0;
    for (int twt = 1; twt <= P0.lw; ++twt) {
        const int TW = cdiv(P0.lw, twt);
        for (int TH = 1; TH <= P0.lh; ++TH) {
            if (TH * TW > WG_MAX_PIX) break;
            const int PH = (TH - 1) * d.in_stride + (dh_max - dh_min) + 1;
            int PW = (TW - 1) * d.in_stride + (dw_max - dw_min) + 1;
            if (k3) {
                if (PW > WG_PITCH) break;
                PW = WG_PITCH;
            }
            const size_t lds = (size_t)2 * (WG_MAX_PIX + 64) * 4 + (size_t)PH * PW * CIB * 4 +
                               ((size_t)TH * los * TW * los + 1) * COB * 4;
            if (tile && (TH != tile->TH || TW != tile->TW)) continue;
            if (lds > budget) { if (tile) continue; break; }
            const double useful = (double)P0.lh * P0.lw / ((double)cdiv(P0.lh, TH) * TH * cdiv(P0.lw, TW) * TW);
            const double halo = (double)PH * PW / ((double)TH * TW * d.in_stride * d.in_stride);
            const double score = useful * (TH * TW >= 64 ? 1.0 : TH * TW / 64.0) / (1.0 + 0.15 * (halo - 1.0));
            if (score > best) {
                best = score;
                pl.TH = TH; pl.TW = TW; pl.lds = lds;
                pl.pitch = k3 ? WG_PITCH : 0;
            }
        }
        if (TW <= 4) break;
    }
    RD_CHECK_ARG(pl.TH > 0, "wgrad: no tile fits in LDS");
    pl.tiles_h = cdiv(P0.lh, pl.TH);
    pl.tiles_w = cdiv(P0.lw, pl.TW);
    pl.total_tiles = d.N * pl.tiles_h * pl.tiles_w;
    const int groups = pl.n_cib * pl.n_cob * pl.n_tg;
    static const char* wpc_env2 = getenv("RD_WGRAD_WG_PER_CU_X2");
    // (UpProj phase launches with many channel blocks -- dec1: 16 -- are slab-traffic-bound like the strip ones: 327 -> 269 us with
    //  one workgroup per CU; the 32-channel stage has one block and needs the two: 257 vs 301 us)
    int want = cdiv((wpc_env2 ? atoi(wpc_env2) : (ph >= 0 && groups >= 8 ? 2 : 4)) * num_cus() / 2, groups);
    if (want < 1) want = 1;
    if (want > pl.total_tiles) want = pl.total_tiles;
    pl.tiles_per_split = cdiv(pl.total_tiles, want);
    pl.n_splits = cdiv(pl.total_tiles, pl.tiles_per_split);
    pl.slab_splits = pl.n_splits;
    pl.J = pl.n_splits < 16 ? pl.n_splits : 16;
    if (out) {
        WgradArgs& a = *out;
        a.N = d.N; a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.ldi = d.ldi;
        a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout; a.ldo = d.ldo; a.IS = d.in_stride; a.OS = d.out_stride;
        a.lh = P0.lh; a.lw = P0.lw; a.TH = pl.TH; a.TW = pl.TW; a.tiles_h = pl.tiles_h; a.tiles_w = pl.tiles_w;
        a.total_tiles = pl.total_tiles; a.tiles_per_split = pl.tiles_per_split; a.n_splits = pl.n_splits;
        a.dh_min = dh_min; a.dh_max = dh_max; a.dw_min = dw_min; a.dw_max = dw_max;
        a.n_cib = pl.n_cib; a.n_cob = pl.n_cob; a.n_tg = pl.n_tg; a.S = S;
        a.compact = ph >= 0; a.off_h = P0.out_off_h; a.off_w = P0.out_off_w;
        // distribute taps over groups: 25 taps -> one kernel row (5 taps) per group; else a single group
        for (int gi = 0; gi < WG_MAX_GROUPS; ++gi) a.tg[gi].n = 0;
        for (int i = 0; i < d.n_phases; ++i) {
            const RdPhase& p = d.phase[i];
            for (int t = 0; t < p.n_taps; ++t) {
                const int gi = ntaps == 25 ? (p.widx[t] / 5) : 0;
                WgTapGroup& G = a.tg[gi];
                RD_CHECK_ARG(G.n < pl.TG, "wgrad: tap grouping overflow");
                G.dh[G.n] = p.dh[t]; G.dw[G.n] = p.dw[t];
                G.oh[G.n] = ph >= 0 ? 0 : (int8_t)p.out_off_h; G.ow[G.n] = ph >= 0 ? 0 : (int8_t)p.out_off_w;
                G.widx[G.n] = p.widx[t];
                ++G.n;
            }
        }
    }
    return RD_OK;
}

static inline bool tile_is_tiled(const WgradPlan* tile) { return tile != nullptr && !tile->strip; }

template <int RH, int RW>
static int launch_strip(const WgradPlan& pl, const WgradArgs& a, int grid, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = wgrad_strip_kernel<RH, RW>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), pl.lds, s, a, pl.sa);
    RD_CHECK_LAUNCH("wgrad_strip_kernel");
    return RD_OK;
}

template <int TG, int MF, bool LA, bool SHB, int PITCH = 0, int RW = 3>
static int launch_wgrad(const WgradArgs& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = wgrad_kernel<TG, MF, LA, SHB, PITCH, RW>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("wgrad_kernel");
    return RD_OK;
}

// UpProj forward form (4 output phases of a 5x5 stencil on the low-res input): one launch per phase, each a rectangular
// 3x3 / 3x2 / 2x3 / 2x2 sub-stencil whose taps share one dout pixel -> the immediate-offset kernel with a shared B fragment,
// instead of one launch of 5-tap row groups that needs a B fragment per tap.  All phases use phase 0's tile and split count,
// so that they fill disjoint slabs of one workspace.
static bool upproj_split(const RdConvDesc& d) {
    static const char* fused = getenv("RD_WGRAD_FUSEDPH");   // diagnostics: keep the single fused launch
    if (fused || d.n_phases != 4 || d.out_stride != 2 || d.in_stride != 1) return false;
    int nt = 0;
    for (int i = 0; i < 4; ++i) nt += d.phase[i].n_taps;
    return nt == 25 && d.phase[0].n_taps == 9;
}
static int plan_any_uncached(const RdConvDesc& d, WgradPlan& pl);
// plans are a pure function of the descriptor: cached (the tile search would otherwise run on every launch of every step)
static int plan_any(const RdConvDesc& d, WgradPlan& pl) {
    static std::mutex mu;
    static std::unordered_map<std::string, WgradPlan> cache;
    std::string key(reinterpret_cast<const char*>(&d), sizeof(RdConvDesc));
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) { pl = it->second; return RD_OK; }
    }
    const int rc = plan_any_uncached(d, pl);
    if (rc != RD_OK) return rc;
    std::lock_guard<std::mutex> lk(mu);
    cache.emplace(std::move(key), pl);
    return RD_OK;
}
static int plan_any_uncached(const RdConvDesc& d, WgradPlan& pl) {
    if (wgrad16_eligible(d)) {
        pl = WgradPlan{};
        int total_tiles = 0;
        wgrad16_splits(d, total_tiles, pl.n_splits);
        pl.w16 = 1; pl.S = 9; pl.slab_splits = pl.n_splits; pl.J = pl.n_splits < 16 ? pl.n_splits : 16;
        pl.total_tiles = total_tiles; pl.MF = 16;
        return RD_OK;
    }
    if (wgrad1x1_eligible(d)) {
        pl = WgradPlan{};
        long long pps = 0;
        wgrad1x1_splits(d, pl.n_splits, pps);
        pl.w1x1 = 1; pl.S = 1; pl.slab_splits = pl.n_splits; pl.J = pl.n_splits < 16 ? pl.n_splits : 16;
        pl.MF = 32; pl.TG = 1;
        return RD_OK;
    }
    if (upproj_split(d)) {
        bool ok = plan_wgrad(d, pl, nullptr, 0) == RD_OK && pl.pitch > 0;
        for (int ph = 1; ok && ph < 4; ++ph) {
            WgradPlan q;
            ok = plan_wgrad(d, q, nullptr, ph, &pl) == RD_OK && q.pitch > 0 && q.n_splits == pl.n_splits;
        }
        if (ok) { pl.per_phase = 1; return RD_OK; }
    }
    return plan_wgrad(d, pl, nullptr);
}

// immediate-offset kernels: (taps, taps per row) = (9,3) (6,3) (6,2) (4,2)
static int launch_pitch(const WgradPlan& pl, const WgradArgs& a, int grid, hipStream_t s) {
    if (pl.strip) {
        if (pl.TG == 9 && pl.rw == 3) return launch_strip<3, 3>(pl, a, grid, s);
        if (pl.TG == 6 && pl.rw == 3) return launch_strip<2, 3>(pl, a, grid, s);
        if (pl.TG == 6 && pl.rw == 2) return launch_strip<3, 2>(pl, a, grid, s);
        if (pl.TG == 4 && pl.rw == 2) return launch_strip<2, 2>(pl, a, grid, s);
        set_error("wgrad: no strip kernel for %d taps, %d per row", pl.TG, pl.rw);
        return RD_EINVAL;
    }
#define RD_WP(TG_, RW_)                                                                                              \
    if (pl.TG == TG_ && pl.rw == RW_) {                                                                              \
        if (pl.MF == 32 && pl.layoutA) return launch_wgrad<TG_, 32, true, true, WG_PITCH, RW_>(a, grid, pl.lds, s);  \
        if (pl.MF == 32) return launch_wgrad<TG_, 32, false, true, WG_PITCH, RW_>(a, grid, pl.lds, s);               \
        return launch_wgrad<TG_, 16, false, true, WG_PITCH, RW_>(a, grid, pl.lds, s);                                \
    }
    RD_WP(9, 3) RD_WP(6, 3) RD_WP(6, 2) RD_WP(4, 2)
#undef RD_WP
    set_error("wgrad: no immediate-offset kernel for %d taps, %d per row", pl.TG, pl.rw);
    return RD_EINVAL;
}

}  // namespace rd
using namespace rd;

// diagnostics: out[0..8] = TG, MF, layoutA, shb, pitch, taps per row, per_phase (4 launches), n_splits, strip kernel
// (TG == 0: wgrad16.hip; layoutA == -1: wgrad1x1.hip)
extern "C" int rd_wgrad_plan_info(const RdConvDesc* d, int32_t* out) {
    if (!d || !out) return RD_EINVAL;
    WgradPlan pl;
    if (plan_any(*d, pl) != RD_OK) return RD_EINVAL;
    const int v[9] = {pl.TG, pl.MF, pl.w1x1 ? -1 : pl.layoutA, pl.shb, pl.pitch, pl.rw, pl.per_phase, pl.n_splits, pl.strip};
    for (int i = 0; i < 9; ++i) out[i] = v[i];
    return RD_OK;
}

extern "C" int64_t rd_wgrad_workspace_floats(const RdConvDesc* d) {
    if (!d) return RD_EINVAL;
    WgradPlan pl;
    if (plan_any(*d, pl) != RD_OK) return RD_EINVAL;
    return (int64_t)(pl.slab_splits + pl.J) * pl.S * d->Cin * d->Cout;
}

extern "C" int rd_wgrad(const RdConvDesc* d, const float* in, const float* dout, float* slabs, void* stream) {
    RD_CHECK_ARG(d && in && dout && slabs, "wgrad: null argument");
    WgradPlan pl;
    WgradArgs a;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const char* dbg = getenv("RD_WGRAD_DEBUG");
    int rc = plan_any(*d, pl);
    if (rc != RD_OK) return rc;
    if (pl.w16) return launch_wgrad16(*d, in, dout, slabs, s);
    if (pl.w1x1) return launch_wgrad1x1(*d, in, dout, slabs, s);
    // the launch records (plan + kernel arguments minus the tensor pointers) are cached per descriptor like the plans
    struct Launch { WgradPlan pl; WgradArgs a; };
    static std::mutex mu;
    static std::unordered_map<std::string, std::vector<Launch>> cache;
    const std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    const std::vector<Launch>* recs = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) recs = &it->second;      // (references into an unordered_map stay valid across inserts)
    }
    if (!recs) {
        std::vector<Launch> v;
        if (pl.per_phase) {
            const WgradPlan pl0 = pl;
            for (int ph = 0; ph < 4; ++ph) {
                Launch L;
                rc = plan_wgrad(*d, L.pl, &L.a, ph, &pl0);
                if (rc != RD_OK) return rc;
                v.push_back(L);
            }
        } else {
            Launch L;
            rc = plan_wgrad(*d, L.pl, &L.a);
            if (rc != RD_OK) return rc;
            L.pl.per_phase = 0;
            v.push_back(L);
        }
        std::lock_guard<std::mutex> lk(mu);
        recs = &cache.emplace(key, std::move(v)).first->second;
    }
    if (pl.per_phase) {
        for (const Launch& L : *recs) {
            pl = L.pl; a = L.a;
            a.in = in; a.dout = dout; a.slabs = slabs; a.debug = dbg ? atoi(dbg) : 0;
            if (!pl.layoutA && pl.lds < (size_t)4 * 16 * 64 * 4) pl.lds = (size_t)4 * 16 * 64 * 4;
            rc = launch_pitch(pl, a, pl.n_cib * pl.n_cob * pl.n_splits, s);
            if (rc != RD_OK) return rc;
        }
        return RD_OK;
    }
    pl = (*recs)[0].pl; a = (*recs)[0].a;
    a.in = in; a.dout = dout; a.slabs = slabs; a.debug = dbg ? atoi(dbg) : 0;
    for (int gi = 0; gi < pl.n_tg; ++gi) RD_CHECK_ARG(a.tg[gi].n == pl.TG, "wgrad: tap group %d has %d taps, expected %d", gi, a.tg[gi].n, pl.TG);
    // layout-B epilogue reuses the head of LDS for its cross-wave reduction
    if (!pl.layoutA && pl.lds < (size_t)4 * 16 * 64 * 4) pl.lds = (size_t)4 * 16 * 64 * 4;
    const int grid = pl.n_cib * pl.n_cob * pl.n_tg * pl.n_splits;
#define RD_W(TG_, MF_, LA_, SHB_) \
    if (pl.TG == TG_ && pl.MF == MF_ && (pl.layoutA != 0) == LA_ && (pl.shb != 0) == SHB_) \
        return launch_wgrad<TG_, MF_, LA_, SHB_>(a, grid, pl.lds, s);
    if (pl.pitch > 0) return launch_pitch(pl, a, grid, s);
    RD_W(9, 32, true, true)
    RD_W(9, 32, false, true)
    RD_W(9, 16, false, true)
    RD_W(5, 32, true, false)
    RD_W(5, 32, false, false)
    RD_W(1, 32, true, true)
    RD_W(1, 32, false, true)
    RD_W(1, 16, false, true)
#undef RD_W
    set_error("wgrad: unsupported plan TG=%d MF=%d layoutA=%d shb=%d", pl.TG, pl.MF, pl.layoutA, pl.shb);
    return RD_EINVAL;
}

extern "C" int rd_wgrad_reduce(const RdConvDesc* d, const float* slabs, float* grad_oihw, int32_t O, int32_t I, int32_t KH,
                               int32_t KW, int32_t co_off, int32_t accumulate, void* stream) {
    RD_CHECK_ARG(d && slabs && grad_oihw, "wgrad_reduce: null argument");
    WgradPlan pl;
    int rc = plan_any(*d, pl);
    if (rc != RD_OK) return rc;
    RD_CHECK_ARG(KH * KW == pl.S && I == d->Cin && co_off + O <= d->Cout, "wgrad_reduce: shape mismatch");
    const int64_t E = (int64_t)pl.S * d->Cin * d->Cout;
    float* tmp = const_cast<float*>(slabs) + (int64_t)pl.slab_splits * E;
    return launch_slab_reduce(slabs, pl.slab_splits, E, tmp, grad_oihw, pl.S, d->Cin, d->Cout, O, I, co_off, accumulate,
                              static_cast<hipStream_t>(stream));
}

// ---- batched reductions (one backward segment's rd_wgrad_reduce / rd_wgrad_bf16_reduce calls as two launches)
namespace rd { bool wgrad_bf16_reduce_shape(const RdConvDesc* d, int* slab_splits, int* S); }

extern "C" int rd_wgrad_reduce_job(const RdConvDesc* d, int32_t bf16_kernel, const float* slabs, float* grad_oihw, int32_t O, int32_t I,
                                   int32_t KH, int32_t KW, int32_t co_off, int32_t accumulate, RdReduceJob* job) {
    RD_CHECK_ARG(d && slabs && grad_oihw && job, "wgrad_reduce_job: null argument");
    int slab_splits = 0, S = 0;
    if (bf16_kernel) {
        if (!wgrad_bf16_reduce_shape(d, &slab_splits, &S)) { set_error("wgrad_reduce_job: unsupported descriptor"); return RD_EINVAL; }
    } else {
        WgradPlan pl;
        int rc = plan_any(*d, pl);
        if (rc != RD_OK) return rc;
        slab_splits = pl.slab_splits; S = pl.S;
    }
    RD_CHECK_ARG(KH * KW == S && I == d->Cin && co_off + O <= d->Cout, "wgrad_reduce_job: shape mismatch");
    return make_reduce_job(slabs, slab_splits, (int64_t)S * d->Cin * d->Cout, grad_oihw, S, d->Cin, d->Cout, O, I, co_off, accumulate, job);
}

extern "C" int rd_wgrad_reduce_batched(const RdReduceJob* jobs_dev, const int32_t* block_job1_dev, int32_t n_blocks1,
                                       const int32_t* block_job2_dev, int32_t n_blocks2, void* stream) {
    RD_CHECK_ARG(jobs_dev && (n_blocks1 == 0 || block_job1_dev) && (n_blocks2 == 0 || block_job2_dev), "wgrad_reduce_batched: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_blocks1 > 0) {
        hipLaunchKernelGGL(wgrad_reduce1_batched_kernel, dim3(n_blocks1), dim3(256), 0, s, jobs_dev, block_job1_dev);
        RD_CHECK_LAUNCH("wgrad_reduce1_batched_kernel");
    }
    if (n_blocks2 > 0) {
        hipLaunchKernelGGL(wgrad_reduce2_batched_kernel, dim3(n_blocks2), dim3(256), 0, s, jobs_dev, block_job2_dev);
        RD_CHECK_LAUNCH("wgrad_reduce2_batched_kernel");
    }
    return RD_OK;
}
