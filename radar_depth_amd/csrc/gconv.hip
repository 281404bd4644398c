// Generalised NHWC fp32 convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// 64 FLOP/clk/SIMD).  See include/radar_depth_hip.h for the descriptor.  Replaces F.conv2d /
// conv_transpose2d+conv2d of the reference (model/models.py:27,96-112,203-206,652-657) and the
// input-gradient half of their autograd.
//
// Structure per workgroup (256 threads = 4 waves, WM x WN wave grid):
//   * a TH x TW tile of logical output pixels (BM = WM*MT*32 rows of the implicit GEMM) times
//     BN = WN*NT*32 output channels;
//   * the input halo patch of the tile is staged ONCE per 16/32-channel chunk in LDS and reused by every
//     tap (direct convolution: no im2col buffer, input read once from HBM/L2);
//   * weights stream through LDS in [taps][WSD/4][BN][4] slabs (rows interleaved by four, the layout
//     rd_pack_weights writes to HBM); patch and slab loads of a chunk are issued as one batch: one
//     memory round trip per chunk;
//   * each wave keeps MT x NT accumulator tiles of 32x32 in registers; a lane's A and B fragments for the
//     CKW/2 MFMAs of a step are one ds_read_b128 (b64 for CKW=4) each: padded or XOR-swizzled pixel
//     layout keeps the A reads conflict-free;
//   * epilogue: optional addend (residual-gradient merge), NHWC store through an output stride/offset
//     (UpProj phases, stride-2 dgrad), and per-tile partial BatchNorm sums for the fused statistics.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace rd {

struct GconvArgs {
    RdConvDesc d;
    const float* in;
    const float* w;
    float* out;
    const float* addend;
    float* stat;
    const float* bias;   // optional per-output-channel bias (eval mode: folded BatchNorm shift)
    int act, act_cols;   // activation applied to output channels < act_cols (RD_ACT_*), after bias and addend
    int ld_add;
    int ldw;           // packed weight row length (>= Cout)
    int TH, TW;        // tile in logical output pixels
    int PP;            // max patch pixels over phases
    int CKP;           // patch channel chunk (16 or 32)
    int tiles_total;   // tiles per image over all phases
    int n_cotiles;
    int taps_max;      // max taps over phases (sizes the LDS weight slab)
    int WSD;           // input channels per weight slab staged in LDS (multiple of CKW, divides CKP)
    int ksplit;        // split-K: number of input-channel slices (each writes its own partial output)
    long long split_stride;  // floats between consecutive partial outputs
    // per phase and tap: offset of the tap inside the LDS halo patch -- bytes for the padded layout, pixels for the swizzled
    // one.  Read through the scalar unit (s_load_dword), so the per-step address arithmetic costs no VALU issue.
    int tapoff[RD_MAX_PHASES][RD_MAX_TAPS + 1];
    // Input-parity groups of an in_stride == 2 (single-phase) descriptor.  The taps are split by the parity of their input offset;
    // group g reads the DECIMATED sub-image in[ih0 + g_oh + 2*y, iw0 + g_ow + 2*x], on which its taps are unit-stride: the patch of
    // a group has the size of an ordinary 3x3 halo patch instead of the 4x interleaved one, the padded LDS layout / scalar tap
    // offsets / pipelined chunk loop of the stride-1 path apply, and two workgroups fit a CU.  The chunk loop runs once per
    // group, all groups accumulate into the same registers.  ngroups == 0: ordinary descriptor.
    int ngroups;
    int g_ntaps[4], g_oh[4], g_ow[4], g_tbase[4];
    int g_sh_max, g_sw_max;      // largest sub-patch tap offset over all groups (sub-patch = (TH + sh_max) x (TW + sw_max))
    short g_widx[32];            // weight slab index of every tap, groups concatenated (g_tbase)
    // BatchNorm-backward sums from the epilogue (rd_gconv_bnbwd): this launch is the input gradient dy of act(s*x + t) -- the
    // BatchNorm that PRODUCED this convolution's forward input.  The epilogue then reads x at its output pixels and emits, per
    // pixel tile, sum g and sum g*(x - mean) with g = dy * act'(s*x + t) into `stat` (layout [tiles][3][Cout], slots 0 and 1): the
    // separate reduce pass over (dy, x) and its launch go away (bn_bwd_reduce_kernel, csrc/norm_act.hip).
    const float* bnb_x;
    const float* bnb_mean;
    const float* bnb_scale;
    const float* bnb_shift;
    int bnb_ld, bnb_act;
    int lds_floats;              // floats of LDS in use before the trace stamps
    unsigned long long* trace;   // diagnostics (RD_GCONV_TRACE=1): per-workgroup cycle-counter stamps, 64 per workgroup
    int debug;         // ablation bits (RD_GCONV_DEBUG env): 1 skip patch staging, 2 skip weight staging, 4 skip MFMA loop
};

// GRP: the descriptor runs as input-parity groups (GconvArgs::ngroups).  A template parameter, not a run-time flag: the large
// register tiles sit exactly at 256 registers and the group bookkeeping as run-time state made the ordinary kernels spill (67 VGPRs).
// BNB: the epilogue also emits BatchNorm-backward sums (GconvArgs::bnb_x; rd_gconv_bnbwd).  A template parameter for the same reason.
template <int MT, int NT, int WM, int WN, int CKW, bool SWZ, bool PIPE, bool GRP = false, bool BNB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MT * NT <= 6 ? 2 : 1))) void gconv_kernel(const GconvArgs a) {
    static_assert(!(GRP && SWZ), "input-parity groups use the padded patch layout");
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;

    // diagnostics: thread 0 stamps the cycle counter into LDS at the phase boundaries and dumps them when the workgroup ends
    int n_stamp = 0;
    unsigned long long* s_stamp = reinterpret_cast<unsigned long long*>(smem + a.lds_floats);   // [64], after everything else
#define RD_STAMP()                                                                                      \
    if (a.trace && tid == 0 && n_stamp < 63) s_stamp[n_stamp++] = __builtin_readcyclecounter();
    RD_STAMP()
    const unsigned long long rt0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz constant clock
    // diagnostics (RD_GCONV_DEBUG bit 8, delay in units of 6.4k cycles in bits 8..15): hold back the workgroups that sit in odd
    // thread-group slots of their CU, so that the two resident workgroups do not run their chunk loops in lockstep
    if (a.debug & 8) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: tg_id in bits 19:16
        if ((hw >> 16) & 1)
            for (int i = 0; i < ((a.debug >> 8) & 255); ++i) __builtin_amdgcn_s_sleep(100);
    }
    const int vid0 = xcd_remap(blockIdx.x, gridDim.x);
    const int ksl = vid0 % a.ksplit;              // split-K slice (slices of one tile are neighbours: same XCD, shared patch)
    const int vid = vid0 / a.ksplit;
    const int cot = vid % a.n_cotiles;
    const int pt = vid / a.n_cotiles;
    const int n = pt / a.tiles_total;
    const int tt = pt - n * a.tiles_total;
    int ph = 0;
    for (int i = 1; i < D.n_phases; ++i)
        if (tt >= D.phase[i].tile_begin) ph = i;
    const RdPhase& P = D.phase[ph];
    const int tloc = tt - P.tile_begin;
    const int tiles_w = (P.lw + a.TW - 1) / a.TW;
    const int r0 = (tloc / tiles_w) * a.TH, c0 = (tloc % tiles_w) * a.TW;
    const int th_n = min(a.TH, P.lh - r0), tw_n = min(a.TW, P.lw - c0);
    const int IS = D.in_stride, OS = D.out_stride;
    constexpr bool grouped = GRP;
    const int ISl = grouped ? 1 : IS;            // pixel step of neighbouring outputs inside the LDS patch
    const int PW = grouped ? a.TW + a.g_sw_max : (a.TW - 1) * IS + (P.dw_max - P.dw_min) + 1;
    const int PH = grouped ? th_n + a.g_sh_max : (th_n - 1) * IS + (P.dh_max - P.dh_min) + 1;
    const int ih0 = r0 * IS + P.dh_min, iw0 = c0 * IS + P.dw_min;
    const int CKP = a.CKP;
    // Patch pixel layout in LDS, two forms (float offset of channel cq of patch pixel px = paddr(px, cq)):
    //  padded   (unit input stride): pixel stride CKP+4 floats -- 16-byte aligned with an odd quad count, so the 16 lanes of
    //           a b128 read pass (consecutive pixels) hit 16 distinct bank groups;
    //  swizzled (input stride 2, whose 4x larger halo patch has no LDS to spare for padding): pixel stride CKP, quad q of
    //           pixel p stored in slot q ^ ((p >> SWS) & (QP-1)).
    const int PS = SWZ ? CKP : CKP + 4;
    const int QPM = (CKP >> 2) - 1, SWS = CKP == 16 ? 2 : 1;
    auto paddr = [&](int px, int cq) { return SWZ ? ((px * CKP + (((px >> SWS) & QPM) << 2)) ^ cq) : px * PS + cq; };
    int ntaps = grouped ? a.g_ntaps[0] : P.n_taps;   // (per group when grouped)
    int tsel = grouped ? 0 : ph;                       // row of a.tapoff in use
    int tb = 0;                                        // first entry of the group's taps in s_widx
    const int co0 = cot * BN;

    // LDS carve-up
    int* s_opix = reinterpret_cast<int*>(smem);          // [BM] output pixel index or -1
    int* s_apix = s_opix + BM;                           // [BM] patch pixel index of tap (0,0)
    int* s_widx = s_apix + BM + 32;                      // [32] weight slab index of each tap ([32] ints before it are spare)
    float* s_w = reinterpret_cast<float*>(s_widx + 32);  // [taps][WSD/4][BN][4]  (quad layout, see layout.hip)
    float* s_patch = s_w + (PIPE ? 2 : 1) * a.taps_max * a.WSD * BN;      // [PP + 2][PS] after the weight slab(s)

    for (int m = tid; m < BM; m += 256) {
        const int r = m / a.TW, c = m - r * a.TW;
        const bool ok = (r < th_n) && (c < tw_n);
        s_opix[m] = ok ? ((n * D.Ho + (r0 + r) * OS + P.out_off_h) * D.Wo + (c0 + c) * OS + P.out_off_w) : -1;
        s_apix[m] = ok ? ((r * ISl) * PW + c * ISl) : 0;
    }
    if (grouped) {
        if (tid < 32) s_widx[tid] = a.g_widx[tid];
    } else if (tid < ntaps) {
        s_widx[tid] = P.widx[tid];
    }
    rd_sync();
    RD_STAMP()

    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) abase[mt] = s_apix[(wm * MT + mt) * 32 + l31];
    const int bcol = wn * NT * 32 + l31;
    // lane-constant byte offsets of the A (padded layout) and B fragments; the per-step part is wave-uniform (SGPR)
    constexpr int KKc = CKW / 2;
    int aoffB[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aoffB[mt] = (abase[mt] * PS + hh * KKc) * 4;
    const int boffB = (((hh * KKc) >> 2) * BN + bcol) * 16 + ((hh * KKc) & 3) * 4;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int q4 = CKP >> 2;                 // float4 per patch pixel
    const int patch_elems = PH * PW * q4;
    const float* in_n = a.in + (size_t)n * D.Hi * D.Wi * D.ldi;

    // ---- one weight slab (WSD input channels of every tap) through the matrix cores; wsel: byte offset of the slab buffer
    const int WSD = a.WSD, W4 = WSD >> 2;
    auto run_slab = [&](int ks, int wsel) {
    // ---- MFMA over (k-quantum, tap) steps.  One step = CKW input channels of one tap = KK*MT*NT MFMAs.  Software
    // pipeline pinned with sched_barrier: the A/B fragments of step s+1 (and the patch offset of step s+2) are in
    // flight from LDS while step s's MFMAs issue.
    constexpr int KK = CKW / 2;
    const int nq = WSD / CKW;
    const int nsteps = (a.debug & 4) ? 0 : nq * ntaps;
    typedef float fK __attribute__((ext_vector_type(KK)));   // KK consecutive channels per lane: one LDS read
    // lane (row|col = l31, hh) feeds channel kq*CKW + hh*KK + kk to the kk-th MFMA of the step (any bijection of the
    // CKW channels onto (kk, hh) is a valid reduction order as long as A and B agree).
    //
    // Cost model (tools/micro/mfma_mix.hip): fp32 MFMAs execute on the SIMD's fp32 lanes, so a VALU instruction
    // in this loop is NOT free -- it costs ~8 clk of MFMA time -- and an LDS read costs ~10 clk when issued in a
    // block but ~3.5 clk when issued between two MFMAs.  Hence: every address = lane-constant VGPR + wave-uniform
    // SGPR term (one v_add per fragment, no multiplies), tap offsets come from the kernel arguments through the
    // scalar unit, and the LDS reads of step s+1 are interleaved with the MFMAs of step s (sched_group_barrier).
    fK ca[MT], cb_[NT], na[MT], nb[NT];
    // (tap, k-quantum) of the step whose fragments are loaded next: wave-uniform, kept in SGPRs.  tap_nxt is fetched
    // one step early so that its s_load is covered by the wait the MFMAs need anyway.
    int t_n = 0, kqA = 0, kqB = 0;
    int tap_cur = a.tapoff[tsel][0], tap_nxt = a.tapoff[tsel][ntaps > 1 ? 1 : 0];
    const char* const wbase = reinterpret_cast<const char*>(s_w) + boffB + wsel;
    const char* const pbase = reinterpret_cast<const char*>(s_patch);
#define RD_GC_LOAD(AV, BV)                                                                     \
    {                                                                                  \
        if constexpr (SWZ) {                                                           \
            const int cq = ks * WSD + (kqA >> 2) + hh * KK;                            \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                          \
                AV[mt] = *reinterpret_cast<const fK*>(s_patch + paddr(abase[mt] + tap_cur, cq)); \
        } else {                                                                       \
            const int sA = tap_cur + ks * WSD * 4 + kqA;                               \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                          \
                AV[mt] = *reinterpret_cast<const fK*>(pbase + (aoffB[mt] + sA));       \
        }                                                                              \
        const int sB = t_n * (W4 * BN * 16) + kqB;                                     \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) BV[nt] = *reinterpret_cast<const fK*>(wbase + sB + nt * 512); \
        ++t_n;                                                                         \
        if (t_n == ntaps) { t_n = 0; kqA += CKW * 4; kqB += (CKW >> 2) * BN * 16; }    \
        tap_cur = tap_nxt;                                                             \
        tap_nxt = a.tapoff[tsel][t_n + 1 == ntaps ? 0 : t_n + 1];                      \
    }
#define RD_GC_MFMA(AV, BV)                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < KK; ++kk)                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                              \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                          \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[mt][kk], BV[nt][kk], acc[mt][nt], 0, 0, 0);
    // schedule of one half-iteration: the address VALU ops, then MFMAs with one LDS read slotted after every
    // KK*NT of them (MT+NT reads in all), then the remaining MFMAs
#define RD_GC_SCHED()                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x002, SWZ ? 8 * MT : MT + 1, 0);             \
    _Pragma("unroll") for (int i = 0; i < MT + NT; ++i) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
    }                                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, KK * MT * NT - 2 * (MT + NT), 0);      \
    __builtin_amdgcn_sched_barrier(0);
    RD_GC_LOAD(ca, cb_)
    __builtin_amdgcn_sched_barrier(0);
    for (int st = 0; st < nsteps; st += 2) {
        // (the step after the last one re-reads in-bounds LDS: kq_n may reach nq, still inside the patch/slab rows
        //  because one extra quantum is reserved by the host-side LDS sizing)
        RD_GC_LOAD(na, nb)
        RD_GC_MFMA(ca, cb_)
        RD_GC_SCHED()
        if (st + 1 < nsteps) {
            RD_GC_LOAD(ca, cb_)
            RD_GC_MFMA(na, nb)
            RD_GC_SCHED()
        }
    }
#undef RD_GC_LOAD
#undef RD_GC_MFMA
#undef RD_GC_SCHED
    };

    const int cin_per = D.Cin / a.ksplit;
    const int cb_lo = ksl * cin_per, cb_hi = cb_lo + cin_per;
    float* const outp = a.out + (size_t)ksl * a.split_stride;
    const int ngr = GRP ? a.ngroups : 1;
    for (int grp = 0; grp < ngr; ++grp) {
    // sub-image origin and step of this pass (ordinary descriptors: one pass over the dense patch)
    int gih0 = ih0, giw0 = iw0, gst = 1;
    if (grouped) {
        ntaps = a.g_ntaps[grp]; tsel = grp; tb = a.g_tbase[grp];
        gih0 = ih0 + a.g_oh[grp]; giw0 = iw0 + a.g_ow[grp]; gst = 2;
        if (grp > 0) rd_sync();     // every wave is done with the previous group's patch and slabs
    }
    if constexpr (PIPE) {
        // ---- software-pipelined chunk loop (the whole patch chunk is one batch of <= UPP loads per thread and a weight slab
        // <= UWP): the weight slabs alternate between two LDS buffers and are fetched with global_load_lds one slab ahead (no
        // registers, no LDS-write pass); the next chunk's patch is loaded into registers while the last slab of the current
        // chunk is in the matrix cores.  Per-element addresses are computed once per workgroup, so a chunk issues ~50 VALU
        // instructions for its staging instead of ~900 (they would queue behind the co-resident workgroup's MFMAs).
        // (Round 3 also ran the input-parity groups through this pipeline as ONE flat (group, chunk, slab) sequence -- next group's
        //  first slab and patch fetched under the last slab of the current one.  Parity-green, no gain: the exposed restart moved
        //  from the staging stamps into the MFMA stamps, stride-2 forward 140 -> 138.5 us, UpProj input gradients 3-6 % slower:
        //  the co-resident workgroups run in lockstep, so it is the barrier / LDS-write phases that are exposed, not the fetch.)
        constexpr int UPP = MT * NT >= 4 ? 8 : 4, UWP = 7;      // (a six-load form that pipelines the 2x1 tile's 256-pixel patch measured 4-7 % slower than its plain loop: round 3)
        const int welems = ntaps * W4 * BN;
        const int slab_bytes = a.taps_max * WSD * BN * 4;
        unsigned pgo[UPP];      // byte offset of the element inside the image at channel 0; ~0u: outside -> zero
        int pdst[UPP];          // LDS float offset, -1: no such element
        // Input-parity groups: the element maps are rebuilt per group from an opaque copy of the thread id.  Without it the compiler
        // hoists every element's group-invariant (row, column, quad, tap) out of the group loop and keeps ~45 VGPRs live across
        // the whole kernel -- the 3x2 / 2x2 register tiles then spilled 40-65 VGPRs into scratch (round 2).
        int tidg = tid;
        if constexpr (GRP) asm volatile("" : "+v"(tidg));
        const int lq4 = 31 - __clz(q4), lW4 = 31 - __clz(W4);          // q4 and W4 are powers of two
#pragma unroll
        for (int u = 0; u < UPP; ++u) {
            const int e = tidg + u * 256;
            const int pix = e >> lq4, qq = e & (q4 - 1);
            const int py = pix / PW, px = pix - py * PW;
            const int ih = gih0 + gst * py, iw = giw0 + gst * px;
            pdst[u] = e < patch_elems ? paddr(pix, qq * 4) : -1;
            pgo[u] = (e < patch_elems && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi) ? (unsigned)(((ih * D.Wi + iw) * D.ldi + qq * 4) * 4) : ~0u;
        }
        unsigned woff[UWP];     // byte offset of the element inside the packed weights at input-channel quad 0; ~0u: zero
#pragma unroll
        for (int u = 0; u < UWP; ++u) {
            const int e = tidg + u * 256;
            const int j = e % BN, tk = e / BN;
            const int k4 = tk & (W4 - 1), t = min(tk >> lW4, ntaps - 1);
            woff[u] = (e < welems && co0 + j < D.Cout) ? (unsigned)((((unsigned)s_widx[tb + t] * (D.Cin >> 2) + k4) * a.ldw + co0 + j) * 16) : ~0u;
        }
        auto issue_slab = [&](int buf, int cq0) {          // cq0: first input-channel quad of the slab
            const char* src = reinterpret_cast<const char*>(a.w) + (size_t)cq0 * a.ldw * 16;
            float* dst = s_w + buf * (slab_bytes >> 2);
#pragma unroll
            for (int u = 0; u < UWP; ++u) {
                const int e = tidg + u * 256;
                if (e < welems) {
                    if (woff[u] != ~0u) glds16(reinterpret_cast<const float*>(src + woff[u]), dst + (e - (tidg & 63)) * 4);
                    else *reinterpret_cast<float4*>(dst + e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto patch_fetch = [&](int cb, float4 (&v)[UPP]) {
            const char* src = reinterpret_cast<const char*>(in_n + cb);
#pragma unroll
            for (int u = 0; u < UPP; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pgo[u] != ~0u) v[u] = *reinterpret_cast<const float4*>(src + pgo[u]);
            }
        };
        auto patch_put = [&](float4 (&v)[UPP]) {
#pragma unroll
            for (int u = 0; u < UPP; ++u)
                if (pdst[u] >= 0) *reinterpret_cast<float4*>(s_patch + pdst[u]) = v[u];
        };
        const int nsub = CKP / WSD;
        int sidx = 0;
        {
            float4 vp[UPP];
            issue_slab(0, cb_lo >> 2);
            patch_fetch(cb_lo, vp);
            patch_put(vp);
        }
        for (int cb = cb_lo; cb < cb_hi; cb += CKP) {
            for (int ks = 0; ks < nsub; ++ks, ++sidx) {
                glds_wait();
                rd_sync();          // slab sidx and the chunk's patch have landed; slab sidx-1 is fully consumed
                RD_STAMP()
                const bool last_ks = ks == nsub - 1;
                const bool more = !(last_ks && cb + CKP >= cb_hi);
                if (more) issue_slab((sidx + 1) & 1, (last_ks ? cb + CKP : cb + (ks + 1) * WSD) >> 2);
                const bool swap = last_ks && cb + CKP < cb_hi;
                float4 vp[UPP];
                if (swap) patch_fetch(cb + CKP, vp);
                run_slab(ks, (sidx & 1) * slab_bytes);
                RD_STAMP()
                if (swap) {
                    rd_sync();      // every wave is done reading this chunk's patch
                    patch_put(vp);
                }
            }
        }
    } else
    for (int cb = cb_lo; cb < cb_hi; cb += CKP) {
        rd_sync();
        // ---- stage the halo patch chunk [PH*PW][CKP] (zero outside the image / beyond Cin) and the first weight slab
        // [taps][WSD/4][BN] quads (a straight copy: the packed operand in HBM has the same quad layout).  All global loads
        // of both are issued before the first LDS write, so a chunk pays ONE memory round trip (a load-wait-store loop
        // exposes every latency; separate patch / weight batches expose two or three).
        // (batch sizes: the large register tiles run two waves per SIMD whatever the staging needs; the small ones keep
        //  their higher occupancy with shorter batches)
        constexpr int UP = MT * NT >= 4 ? 8 : 4, UW = MT * NT >= 4 ? 9 : 5;
        const int nsub = min(CKP, cb_hi - cb) / WSD;
        const int welems = ntaps * W4 * BN;
        const int cinq = D.Cin >> 2;
        const int pe = ((a.debug & 1) && cb > cb_lo) ? 0 : patch_elems;
        // The patch is copied as ROW SEGMENTS: segment = wave + 4*k is wave-uniform (scalar unit) and covers 64 consecutive
        // float4 units of one patch row, so a lane's (pixel, channel quad) are shifts of the lane id -- no division by a runtime
        // value per element (fp32 MFMAs share the SIMD's fp32 lanes: every VALU instruction of the staging costs matrix time).
        // Halo / padding units go through a raw buffer descriptor whose out-of-range offsets return zeros: no branch per unit
        // (a branch makes the compiler wait for each load before issuing the next).
        const int lq4 = 31 - __clz(q4);
        const int rowu = PW << lq4, nseg = (rowu + 63) >> 6, nsegs = pe ? PH * nseg : 0;
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(in_n + cb), 0, (unsigned)(D.Hi * D.Wi * D.ldi - cb) * 4u, 0x00020000);
        auto patch_load = [&](int k0, auto& v, auto& ld) {
            constexpr int N = sizeof(v) / sizeof(v[0]);
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const int seg = wave_u + 4 * (k0 + u);
                const int row = nseg == 1 ? seg : seg / nseg;
                const int cu = ((seg - row * nseg) << 6) + lane;
                const int px = cu >> lq4, qq = cu & (q4 - 1);
                const int ih = gih0 + gst * row, iw = giw0 + gst * px;
                const bool ok = seg < nsegs && cu < rowu;
                ld[u] = ok ? paddr(row * PW + px, qq * 4) : -1;
                const unsigned go = (ok && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi && cb + qq * 4 < D.Cin)
                                        ? (unsigned)(((ih * D.Wi + iw) * D.ldi + qq * 4) * 4) : 0x80000000u;
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)go, 0, 0));
            }
        };
        auto patch_store = [&](auto& v, auto& ld) {
            constexpr int N = sizeof(v) / sizeof(v[0]);
#pragma unroll
            for (int u = 0; u < N; ++u)
                if (ld[u] >= 0) *reinterpret_cast<float4*>(s_patch + ld[u]) = v[u];
        };
        const int lW4 = 31 - __clz(W4);
        auto w_load = [&](int base, int ks, auto& v) {
            constexpr int N = sizeof(v) / sizeof(v[0]);
            const int we = ((a.debug & 2) && (cb > cb_lo || ks > 0)) ? 0 : welems;
            const int cq0 = (cb + ks * WSD) >> 2;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const int e = base + u * 256;
                const int j = e % BN;
                const int tk = e / BN;
                const int k4 = tk & (W4 - 1), t = min(tk >> lW4, ntaps - 1);      // W4 is a power of two
                const int co = co0 + j;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < we && co < D.Cout)
                    v[u] = *reinterpret_cast<const float4*>(a.w + (((size_t)s_widx[tb + t] * cinq + cq0 + k4) * a.ldw + co) * 4);
            }
        };
        auto w_store = [&](int base, int ks, auto& v) {
            constexpr int N = sizeof(v) / sizeof(v[0]);
            const int we = ((a.debug & 2) && (cb > cb_lo || ks > 0)) ? 0 : welems;
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const int e = base + u * 256;
                if (e < we) *reinterpret_cast<float4*>(s_w + (size_t)e * 4) = v[u];   // [t][k4][BN] is linear in e
            }
        };
        const int nk = (nsegs + 3) >> 2;            // row segments per wave
        {
            float4 vp[UP], vw[UW];
            int ld[UP];
            patch_load(0, vp, ld);
            w_load(tid, 0, vw);
            patch_store(vp, ld);
            w_store(tid, 0, vw);
        }
        for (int k0 = UP; k0 < nk; k0 += UP) {
            float4 vp[UP];
            int ld[UP];
            patch_load(k0, vp, ld);
            patch_store(vp, ld);
        }
        for (int ks = 0; ks < nsub; ++ks) {
            if (ks > 0) rd_sync();
            for (int base = ks > 0 ? tid : tid + 256 * UW; base < welems; base += 256 * UW) {
                float4 vw[UW];
                w_load(base, ks, vw);
                w_store(base, ks, vw);
            }
            rd_sync();
            RD_STAMP()
            run_slab(ks, 0);
            RD_STAMP()
        }
    }
    }   // input-parity groups

    // ---- epilogue.  Row validity is uniform per half-wave and almost always true, so full M-tiles take a wave-uniform
    // fast path: no exec masking, one row pointer per accumulator row with the N-tile as an immediate offset, and the
    // residual-gradient addend gathered before its first use (a load-wait-add-store chain per element exposes every latency).
    RD_STAMP()
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    const bool has_add = a.addend != nullptr;
    const bool has_bias = a.bias != nullptr;
    // inference form (bias / activation in the epilogue, rd_gconv_fused) takes the general path below; the training
    // fast path stays free of it (adding it there cost 80 VGPRs and ~20 % on the big training kernels)
    const bool fused = has_bias || a.act != RD_ACT_NONE;
    const int cob = co0 + wn * NT * 32 + l31;
    float biasv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) biasv[nt] = (has_bias && cob + nt * 32 < D.Cout) ? a.bias[cob + nt * 32] : 0.f;
    // column validity is lane-constant (channel = cob + nt*32): the fast path keeps its unrolled, branch-free shape and lets the
    // stores / addend loads of lanes beyond Cout be masked off (the 16-channel layers fill half a 32-wide tile: they used to fall
    // into the per-element path below, whose epilogue took longer than the layer's MFMAs)
    bool cok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) cok[nt] = cob + nt * 32 < D.Cout;
    // BatchNorm-backward sums (see GconvArgs::bnb_x): a lane owns one channel per N-tile, so the coefficients are lane constants
    constexpr bool bnb = BNB;
    float bnS[NT], bnT[NT], bnM[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const bool on = bnb && cok[nt];
        bnS[nt] = on ? a.bnb_scale[cob + nt * 32] : 0.f;
        bnT[nt] = on ? a.bnb_shift[cob + nt * 32] : 0.f;
        bnM[nt] = on ? a.bnb_mean[cob + nt * 32] : 0.f;
    }
    // (one M-tile per call of a generic lambda with a compile-time index: with the BatchNorm-backward sums in the body the
    //  `#pragma unroll` form of this loop was no longer unrolled for the 3x2 tile -- the accumulators must stay statically indexed)
    auto epilogue_tile = [&](auto MTC) {
        constexpr int mt = decltype(MTC)::value;
        __builtin_amdgcn_sched_barrier(0);   // keep one M-tile's row offsets / addends live at a time
        int ro[16];
        bool rows_ok = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ro[i] = s_opix[(wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];   // -1: no such pixel
            rows_ok = rows_ok && ro[i] >= 0;
        }
        if (!fused && __all(rows_ok)) {
            // two batches of 8 accumulator rows: enough loads in flight to hide the addend latency without pushing the
            // kernel past 256 registers (arch + accumulation), which would halve the waves per SIMD
#pragma unroll
            for (int h8 = 0; h8 < 16; h8 += 8) {
                float addv[NT][8];       // the residual-gradient addend, or (bnb) the BatchNorm input x at the output pixels
                if (has_add || bnb) {
                    const float* const src = bnb ? a.bnb_x : a.addend;
                    const int lds_ = bnb ? a.bnb_ld : a.ld_add;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float* ap = src + (size_t)ro[h8 + i] * lds_ + cob;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) addv[nt][i] = cok[nt] ? ap[nt * 32] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float* rp = outp + (size_t)ro[h8 + i] * D.ldo + cob;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float v = acc[mt][nt][h8 + i];
                        if (has_add && !bnb) v += addv[nt][i];
                        if (cok[nt]) rp[nt * 32] = v;
                        if (bnb) {
                            const float xv = addv[nt][i];
                            const float g = v * act_grad_from_out(fmaf(bnS[nt], xv, bnT[nt]), a.bnb_act);
                            ssum[nt] += g;
                            ssq[nt] += g * (xv - bnM[nt]);
                        } else {
                            ssum[nt] += v;
                            ssq[nt] += v * v;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = cob + nt * 32;
                const bool cok = co < D.Cout;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (cok && ro[i] >= 0) {
                        float v = acc[mt][nt][i];
                        if (has_bias) v += biasv[nt];
                        if (has_add) v += a.addend[(size_t)ro[i] * a.ld_add + co];
                        if (co < a.act_cols) v = act_fwd(v, a.act);
                        outp[(size_t)ro[i] * D.ldo + co] = v;
                        if (bnb) {
                            const float xv = a.bnb_x[(size_t)ro[i] * a.bnb_ld + co];
                            const float g = v * act_grad_from_out(fmaf(bnS[nt], xv, bnT[nt]), a.bnb_act);
                            ssum[nt] += g;
                            ssq[nt] += g * (xv - bnM[nt]);
                        } else {
                            ssum[nt] += v;
                            ssq[nt] += v * v;
                        }
                    }
                }
            }
        }
    };
    epilogue_tile(std::integral_constant<int, 0>{});
    if constexpr (MT > 1) epilogue_tile(std::integral_constant<int, 1>{});
    if constexpr (MT > 2) epilogue_tile(std::integral_constant<int, 2>{});
    static_assert(MT <= 3, "epilogue covers up to three M-tiles");
    if (a.stat) {
        rd_sync();
        float* red = s_w;  // [WM][2][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
            const float q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
            if (hh == 0) {
                red[(wm * 2 + 0) * BN + wn * NT * 32 + nt * 32 + l31] = s;
                red[(wm * 2 + 1) * BN + wn * NT * 32 + nt * 32 + l31] = q;
            }
        }
        rd_sync();
        if (tid < 2 * BN) {
            const int which = tid / BN, j = tid - which * BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + j];
            const int co = co0 + j;
            if (co < D.Cout) a.stat[((size_t)pt * (bnb ? 3 : 2) + which) * D.Cout + co] = s;
        }
    }
    RD_STAMP()
    if (a.trace && tid == 0) {
        a.trace[(size_t)blockIdx.x * 64] = n_stamp;
        a.trace[(size_t)blockIdx.x * 64 + 62] = __builtin_amdgcn_s_memrealtime() - rt0;
        // where it ran: HW_REG_XCC_ID (id 20) and HW_REG_HW_ID (id 4); counters are only comparable within one CU
        a.trace[(size_t)blockIdx.x * 64 + 63] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                                 __builtin_amdgcn_s_getreg((31 << 11) | 4);
        for (int i = 0; i < n_stamp; ++i) a.trace[(size_t)blockIdx.x * 64 + 1 + i] = s_stamp[i];
    }
#undef RD_STAMP
}

// split-K combine: out[r][c] = sum_s part[s][r][c] (+ addend[r][c]); optional BN partial sums per row block.
// Rows are the output tensor's pixels; thread = (channel quad, row lane) like the BN reductions in norm_act.hip.
__global__ __launch_bounds__(256) void gconv_combine_kernel(const float* __restrict__ part, long long split_stride, int S,
                                                            float* __restrict__ out, int ldo, const float* __restrict__ addend,
                                                            int ld_add, long long M, int C, int RPB, float* __restrict__ stat,
                                                            const float* __restrict__ bias, int act, int act_cols) {
    extern __shared__ float sm[];
    const int Q = C >> 2, RL = 256 / Q;
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const bool active = rl < RL;
    const long long r0 = (long long)blockIdx.x * RPB, r1 = r0 + RPB < M ? r0 + RPB : M;
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    if (active) {
        const int c = q * 4;
        for (long long r = r0 + rl; r < r1; r += RL) {
            float4 v = *reinterpret_cast<const float4*>(part + r * ldo + c);
            for (int k = 1; k < S; ++k) {
                const float4 u = *reinterpret_cast<const float4*>(part + k * split_stride + r * ldo + c);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            if (bias) {
                const float4 u = *reinterpret_cast<const float4*>(bias + c);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            if (addend) {
                const float4 u = *reinterpret_cast<const float4*>(addend + r * ld_add + c);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            if (act != RD_ACT_NONE) {
                if (c + 0 < act_cols) v.x = act_fwd(v.x, act);
                if (c + 1 < act_cols) v.y = act_fwd(v.y, act);
                if (c + 2 < act_cols) v.z = act_fwd(v.z, act);
                if (c + 3 < act_cols) v.w = act_fwd(v.w, act);
            }
            *reinterpret_cast<float4*>(out + r * ldo + c) = v;
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
            s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
        }
    }
    if (stat) {
        // sm: [RL][2][C]
        if (active) {
            *reinterpret_cast<float4*>(sm + ((size_t)(rl * 2 + 0) * Q + q) * 4) = s1;
            *reinterpret_cast<float4*>(sm + ((size_t)(rl * 2 + 1) * Q + q) * 4) = s2;
        }
        rd_sync();
        for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) {
            const int w = e / C, c = e - w * C;
            float v = 0.f;
            for (int r = 0; r < RL; ++r) v += sm[(size_t)(r * 2 + w) * C + c];
            stat[((size_t)blockIdx.x * 2 + w) * C + c] = v;
        }
    }
}
static inline int combine_rows_per_block(long long M) {
    long long r = (M + 1023) / 1024;
    return (int)(r < 16 ? 16 : r);
}

// ------------------------------------------------------------------------------------------ host
struct GconvPlan {
    int MT, NT, WM, WN, CKW, CKP, TH, TW, PP, tiles_total, n_cotiles, taps_max, WSD, ksplit;
    size_t lds_bytes;
    int pipe;      // software-pipelined chunk loop (double-buffered weight slabs via global_load_lds)
    int grouped;   // input-parity groups (in_stride == 2, one phase): see GconvArgs::ngroups
    int c16;       // served by conv16.hip (16 -> 16 channels, 3x3, unit strides): 16 x 16 pixel tiles, no split, no workspace
    int wsd_half;  // pipelined loop with half-depth weight slabs (tuner-only point: less LDS -> a third / fourth resident workgroup,
                   // twice the barriers; layer4 3x3 270 -> 253 us, UpProj 128 207 -> 192 us, UpProj 256 211 -> 216 us)
};

// Input-parity decomposition of a single-phase in_stride == 2 descriptor.
struct ParityGroups {
    int n = 0;
    int ntaps[4], oh[4], ow[4], tbase[4];
    int sh[32], sw[32], widx[32];     // per tap (groups concatenated): offset inside the decimated sub-image, weight slab
    int sh_max = 0, sw_max = 0, taps_max = 0;
};
static bool build_parity_groups(const RdConvDesc& d, ParityGroups& G) {
    static const char* off = getenv("RD_GCONV_NOGROUP");      // diagnostics: keep the interleaved 4x patch
    if (off || d.in_stride != 2 || d.n_phases != 1 || d.phase[0].n_taps > 32) return false;
    const RdPhase& p = d.phase[0];
    int k = 0;
    for (int ra = 0; ra < 2; ++ra)
        for (int rb = 0; rb < 2; ++rb) {
            const int k0 = k;
            for (int t = 0; t < p.n_taps; ++t) {
                const int a = p.dh[t] - p.dh_min, b = p.dw[t] - p.dw_min;
                if ((a & 1) != ra || (b & 1) != rb) continue;
                G.sh[k] = a >> 1; G.sw[k] = b >> 1; G.widx[k] = p.widx[t];
                G.sh_max = G.sh_max > G.sh[k] ? G.sh_max : G.sh[k];
                G.sw_max = G.sw_max > G.sw[k] ? G.sw_max : G.sw[k];
                ++k;
            }
            if (k == k0) continue;
            G.ntaps[G.n] = k - k0; G.oh[G.n] = ra; G.ow[G.n] = rb; G.tbase[G.n] = k0;
            G.taps_max = G.taps_max > k - k0 ? G.taps_max : k - k0;
            ++G.n;
        }
    return G.n >= 1;
}

static int patch_pixels(const RdConvDesc& d, const RdPhase& p, int TH, int TW) {
    const int th = TH < p.lh ? TH : p.lh;
    const int PH = (th - 1) * d.in_stride + (p.dh_max - p.dh_min) + 1;
    const int PW = (TW - 1) * d.in_stride + (p.dw_max - p.dw_min) + 1;
    return PH * PW;
}

// weight-slab depth: as many input channels as fit ~40 KB (fewer barriers), at least one CKW quantum
static int pick_wsd(int taps_max, int BN, int CKW, int CKP, bool pipe = false) {
    static const char* cap = getenv("RD_GCONV_WSD_KB");   // diagnostics: slab budget in KB (default 40; two slabs when pipelined)
    const size_t budget = (size_t)(cap ? atoi(cap) : 40) * 1024 / (pipe ? 2 : 1);
    int w = CKP;
    while (w > CKW && (size_t)taps_max * w * BN * 4 > budget) w >>= 1;
    return w < CKW ? CKW : w;
}
static size_t lds_need(int BM, int BN, int CKW, int CKP, int PP, int taps_max, bool swz, bool pipe = false, int wsd = 0) {
    // + one CKW quantum of slab rows and one patch pixel row of slack: the pipelined loop prefetches one step past the end
    if (wsd == 0) wsd = pick_wsd(taps_max, BN, CKW, CKP, pipe);
    return (size_t)(2 * BM + 64) * 4 + ((size_t)taps_max * wsd * (pipe ? 2 : 1) + CKW) * BN * 4 +
            (size_t)(PP + 2) * (swz ? CKP : CKP + 4) * 4 + 64;
}

// Choose wave tiling + pixel tile for a descriptor.  Heuristic: maximise useful-MAC fraction of the
// BM x BN tile, penalise halo re-reads, prefer <= 80 KB of LDS (two workgroups per CU).
// all: when given, every feasible (score, plan) point of the search is collected (the plan tuner's candidate list); the env overrides
// RD_GCONV_FORCE / RD_GCONV_CKP / ... are diagnostics of the HEURISTIC and are not applied to a collecting search.
static bool plan_gconv(const RdConvDesc& d, GconvPlan& best, bool allow_split = false,
                       std::vector<std::pair<double, GconvPlan>>* all = nullptr) {
    // prior = measured relative efficiency of the register tile on large layers (tools/sweep_gconv.py, B=16 layer1/layer2);
    // WN=2 tilings (2,2,2,2), (4,2,2,2) never won a shape and were dropped.
    struct Cfg { int MT, NT, WM, WN; double prior; };
    static const Cfg cfgs[] = {{2, 2, 4, 1, 0.93}, {2, 1, 4, 1, 0.82}, {3, 2, 4, 1, 1.0}, {1, 2, 4, 1, 0.78}, {1, 1, 4, 1, 0.76}};
    int taps_max = 0;
    for (int i = 0; i < d.n_phases; ++i) taps_max = taps_max > d.phase[i].n_taps ? taps_max : d.phase[i].n_taps;
    ParityGroups PG;
    const bool grouped = build_parity_groups(d, PG);
    if (grouped) taps_max = PG.taps_max;          // a weight slab holds one group's taps
    const int CKW = taps_max > 9 ? 4 : 8;
    double best_score = -1;
    // reference phase for tile selection: the one with the largest logical grid
    int pr = 0;
    for (int i = 1; i < d.n_phases; ++i)
        if ((int64_t)d.phase[i].lh * d.phase[i].lw > (int64_t)d.phase[pr].lh * d.phase[pr].lw) pr = i;
    const RdPhase& P = d.phase[pr];
    static const char* force = getenv("RD_GCONV_FORCE");   // diagnostics: index into cfgs
    int cfg_i = -1;
    for (const Cfg& c : cfgs) {
        ++cfg_i;
        if (!all && force && atoi(force) != cfg_i) continue;
        const int BM = c.WM * c.MT * 32, BN = c.WN * c.NT * 32;
        const int n_cot = cdiv(d.Cout, BN);
        const double n_util = (double)d.Cout / (n_cot * BN);
        static const char* force_ckp = getenv("RD_GCONV_CKP");   // diagnostics
        for (int ckp = 32; ckp >= 16; ckp -= 16) {
            if (!all && force_ckp && atoi(force_ckp) != ckp && d.Cin >= 32) continue;
            if (ckp > d.Cin && ckp != 16) continue;
            if (d.Cin % ckp != 0 && !(ckp == 16 && d.Cin % 16 == 0)) continue;
            for (int twt = 1; twt <= cdiv(P.lw, 4); ++twt) {
                const int TW = cdiv(P.lw, twt);
                if (TW > BM) continue;
                int TH = BM / TW;
                if (TH > P.lh) TH = P.lh;
                // balance rows across tiles
                TH = cdiv(P.lh, cdiv(P.lh, TH));
                int PP = 0;
                for (int i = 0; i < d.n_phases; ++i) {
                    const int pp = grouped ? (TH + PG.sh_max) * (TW + PG.sw_max) : patch_pixels(d, d.phase[i], TH, TW);
                    PP = PP > pp ? PP : pp;
                }
                // pipelined chunk loop: the patch chunk must be one batch of loads per thread, a weight slab at most seven
                static const char* nopipe = getenv("RD_GCONV_NOPIPE");
                const int upp = c.MT * c.NT >= 4 ? 8 : 4;
                const int wsd_p = pick_wsd(taps_max, BN, CKW, ckp, true);
                static const char* gnopipe = getenv("RD_GCONV_GROUP_NOPIPE");   // diagnostics
                const bool pipe_ok = !nopipe && !(grouped && gnopipe) && PP * (ckp / 4) <= upp * 256 && taps_max * (wsd_p / 4) * BN <= 7 * 256;
                // (a collecting search also lists the plain-loop form of a pipelinable point: the tuner found it faster for some
                //  small launches)
                // (pv == 2: the pipelined loop with half-depth weight slabs -- listed for the tuner only)
                for (int pv = pipe_ok ? (all ? 2 : 1) : 0; pv >= ((all && pipe_ok) ? 0 : (pipe_ok ? 1 : 0)); --pv) {
                const bool pipe = pv != 0;
                int wsd_pt = pick_wsd(taps_max, BN, CKW, ckp, pipe);
                if (pv == 2) {
                    if (wsd_pt <= CKW) continue;
                    wsd_pt >>= 1;
                }
                const size_t lds = lds_need(BM, BN, CKW, ckp, PP, taps_max, d.in_stride == 2 && !grouped, pipe, wsd_pt);
                if (lds > 160 * 1024 - 512) continue;
                const double m_util = (double)P.lh * P.lw / ((double)cdiv(P.lh, TH) * cdiv(P.lw, TW) * BM);
                const double halo = grouped ? (double)PP / (TH * TW) : (double)PP / (TH * TW * d.in_stride * d.in_stride);
                double score = c.prior * m_util * n_util / (1.0 + 0.04 * (halo - 1.0));
                if (lds > 80 * 1024) score *= 0.85;          // one workgroup per CU only
                // half-line loads cost the large tiles ~3 %; the single-block tile on a 3x3 stencil is the other way round (the
                // small-spatial depth-branch layers: 28.8 -> 24.7 us, 34.8 -> 30.7 us with 16-channel chunks, tools/sweep_plan_layers.py)
                // (a long reduction in the plain chunk loop exposes one memory round trip per chunk: the pipelined form of the same
                //  point, where it exists, or a smaller chunk that makes it exist, measured 1.3-1.45x faster from four chunks on)
                static const char* no_rules = getenv("RD_GCONV_NO_RULES");      // diagnostics: bit 1 pipelining preference, 2 chunk size of the
                const int nr = no_rules ? atoi(no_rules) : 0;                   //              single-block tile, 4 phase balance
                if (!(nr & 1) && !pipe && d.Cin / ckp >= 4) score *= 0.8;
                const bool small3 = !(nr & 2) && c.MT * c.NT == 1 && taps_max >= 9;
                if (ckp == (small3 ? 32 : 16) && d.Cin >= 32) score *= 0.97;
                // CU load balance: the kernel is MFMA-bound, so the time is set by the CU that owns the most workgroups.
                // Small-spatial / many-channel layers do not produce enough large tiles: split the input channels over
                // KS workgroups per tile (partials combined by gconv_combine_kernel).  Only single-phase, hole-free,
                // unit-stride-output descriptors with a dense output buffer qualify.
                const double wgs1 = (double)d.N * cdiv(P.lh, TH) * cdiv(P.lw, TW) * n_cot * d.n_phases;
                const double ncu = (double)num_cus();
                static const char* nosplit = getenv("RD_GCONV_NOSPLIT");
                // (multi-phase descriptors qualify when their phases tile the whole output, e.g. the four UpProj / stride-2
                //  dgrad parity phases: every element of the partial buffers is then written before the combine reads it)
                int64_t covered = 0;
                for (int i = 0; i < d.n_phases; ++i) covered += (int64_t)d.phase[i].lh * d.phase[i].lw;
                const bool hole_free = d.n_phases == 1 ? d.out_stride == 1 : covered == (int64_t)d.Ho * d.Wo;
                const bool can_split = !nosplit && allow_split && hole_free && d.ldo == d.Cout && d.Cout <= 1024;
                double base = score;
                // A split slice must still carry enough MFMA work to pay for the combine launch (~10 us) and the partial-sum
                // round trip: the 1x1 convolutions and the small depth-branch layers ran up to 2x slower split four ways than
                // unsplit (fusion 1x1 forward 78 -> 52 us, its dgrad 104 -> 53 us), the 512-channel 3x3 layers 1.3x faster.
                static const char* minmf = getenv("RD_GCONV_SPLIT_MIN_MFLOP");      // diagnostics (default 25)
                const double min_slice_flops = (minmf ? atof(minmf) : 25.0) * 1e6;
                double taps_avg = 0;
                for (int i = 0; i < d.n_phases; ++i) taps_avg += d.phase[i].n_taps;
                taps_avg /= d.n_phases;
                const double wg_flops = 2.0 * TH * TW * BN * (double)d.Cin * taps_avg;
                double taps_big = 0;
                for (int i = 0; i < d.n_phases; ++i) taps_big = taps_big > d.phase[i].n_taps ? taps_big : d.phase[i].n_taps;
                for (int ksp = 1; ksp <= (can_split ? 4 : 1); ksp *= 2) {
                    if (d.Cin % (ksp * ckp) != 0) continue;
                    if (!all && ksp > 1 && wg_flops / ksp < min_slice_flops) continue;
                    const double wgs = wgs1 * ksp;
                    // Phases with different tap counts (UpProj 9/6/6/4, the stride-2 input gradient 4/2/2/1) make workgroups of
                    // different length: with fewer than ~4 of them per CU the long ones set the time (the per-layer sweep: stride-2
                    // dgrads 216 -> 152, 212 -> 160, 191 -> 167 us and the two small UpProj forwards 237 -> 205, 222 -> 204 us on the
                    // single-block tile instead of the 3x2 one)
                    const double phase_balance = (!(nr & 4) && d.n_phases > 1 && wgs < 4 * ncu) ? taps_avg / taps_big : 1.0;
                    // fewer than two workgroups per CU leaves staging/epilogue phases uncovered
                    static const char* kspen = getenv("RD_GCONV_KS_PEN");      // diagnostics: "p2,p4" score factors of a 2- / 4-way split
                    // (0.85 / 0.72: a split costs the partial-sum round trip and the combine launch; with the 0.95 / 0.91 of round 1 the
                    //  planner split layers at b=8 that the tuner ran 12-20 % faster unsplit on smaller tiles -- multistage b=8 312 -> 319,
                    //  latefusion b=16 741 -> 746 samples/s, tools/tune_report.py)
                    static const double p2 = kspen ? atof(kspen) : 0.85, p4 = kspen && strchr(kspen, ',') ? atof(strchr(kspen, ',') + 1) : 0.72;
                    score = base * (wgs / (ncu * ceil(wgs / ncu))) * (ksp == 1 ? 1.0 : (ksp == 2 ? p2 : p4)) *
                            (wgs < 2 * ncu && lds <= 80 * 1024 ? 0.9 : 1.0) *   // (a > 80 KB tile already paid for single residency)
                            phase_balance;
                    const GconvPlan cand{c.MT, c.NT, c.WM, c.WN, CKW, ckp, TH, TW, PP, 0, n_cot, taps_max, wsd_pt, ksp, lds,
                                         pipe ? 1 : 0, grouped ? 1 : 0, 0, pv == 2 ? 1 : 0};
                    if (all) all->emplace_back(score, cand);
                    if (score > best_score) {
                        best_score = score;
                        best = cand;
                    }
                }
                }   // pipelined / plain form
            }
        }
    }
    return best_score > 0;
}

template <int MT, int NT, int WM, int WN, int CKW, bool SWZ, bool PIPE, bool GRP = false, bool BNB = false>
static int launch_cfg(const GconvArgs& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = gconv_kernel<MT, NT, WM, WN, CKW, SWZ, PIPE, GRP, BNB>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("gconv_kernel");
    return RD_OK;
}

static int validate_desc(const RdConvDesc* d) {
    RD_CHECK_ARG(d != nullptr, "gconv: null descriptor");
    RD_CHECK_ARG(d->n_phases >= 1 && d->n_phases <= RD_MAX_PHASES, "gconv: n_phases=%d", d->n_phases);
    RD_CHECK_ARG(d->Cin % 16 == 0 && d->ldi % 4 == 0, "gconv: Cin=%d must be a multiple of 16, ldi=%d of 4", d->Cin, d->ldi);
    RD_CHECK_ARG(d->Cout % 4 == 0, "gconv: Cout=%d must be a multiple of 4", d->Cout);
    RD_CHECK_ARG((int64_t)d->Hi * d->Wi * d->ldi * 4 < (int64_t)0x80000000u, "gconv: one input image must stay below 2 GiB (32-bit buffer offsets)");
    RD_CHECK_ARG(d->in_stride >= 1 && d->in_stride <= 2 && d->out_stride >= 1 && d->out_stride <= 2, "gconv: strides");
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        RD_CHECK_ARG(p.n_taps >= 1 && p.n_taps <= RD_MAX_TAPS, "gconv: phase %d has %d taps", i, p.n_taps);
        RD_CHECK_ARG(p.lh >= 1 && p.lw >= 1, "gconv: empty phase %d", i);
        for (int t = 0; t < p.n_taps; ++t)
            RD_CHECK_ARG(p.dh[t] >= p.dh_min && p.dh[t] <= p.dh_max && p.dw[t] >= p.dw_min && p.dw[t] <= p.dw_max,
                         "gconv: tap %d of phase %d outside its declared range", t, i);
    }
    return RD_OK;
}

static void fill_tiles(RdConvDesc& d, GconvPlan& pl) {
    int tb = 0;
    for (int i = 0; i < d.n_phases; ++i) {
        d.phase[i].tile_begin = tb;
        tb += cdiv(d.phase[i].lh, pl.TH) * cdiv(d.phase[i].lw, pl.TW);
    }
    pl.tiles_total = tb;
}

}  // namespace rd

using namespace rd;

static int plan_query(const RdConvDesc* d, bool allow_split, GconvPlan& pl, RdConvDesc& dd, bool generic_only = false);

// diagnostics: out[0..9] = MT, NT, WM, WN, CKW, CKP, TH, TW, lds_bytes, workgroups
extern "C" int rd_gconv_plan_info(const RdConvDesc* d, int32_t* out) {
    GconvPlan pl;
    RdConvDesc dd;
    if (plan_query(d, true, pl, dd) != RD_OK) return RD_EINVAL;       // the cached plan: heuristic, or pinned by the tuner
    // (reports the plan used WITH a workspace; CKW slot carries pipe*10000 + ksplit*100 + CKW)
    // (CKW slot: + 1000000 when the in_stride == 2 descriptor runs as input-parity groups, i.e. on the SWZ = false instantiation)
    const int v[10] = {pl.MT, pl.NT, pl.WM, pl.WN, pl.grouped * 1000000 + pl.pipe * 10000 + pl.ksplit * 100 + pl.CKW, pl.CKP, pl.TH, pl.TW, (int)pl.lds_bytes,
                       d->N * pl.tiles_total * pl.n_cotiles * pl.ksplit};
    for (int i = 0; i < 10; ++i) out[i] = v[i];
    return RD_OK;
}

template <int MT, int NT, int WM, int WN, int CKW, bool SWZ, bool PIPE>
static int occ_cfg(size_t lds) {
    int n = -1;
    auto k = gconv_kernel<MT, NT, WM, WN, CKW, SWZ, PIPE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds) != hipSuccess) n = -1;
    return n;
}
// diagnostics: resident workgroups per CU the HIP occupancy API reports for the plan chosen for d (-1 without a GPU)
extern "C" int rd_gconv_occupancy(const RdConvDesc* d) {
    if (validate_desc(d) != RD_OK) return RD_EINVAL;
    GconvPlan pl;
    RdConvDesc dd = *d;
    if (!plan_gconv(dd, pl)) return RD_EINVAL;
#define RD_OCC2(MT_, NT_, WM_, WN_, P_)                                                                                        \
    (swz ? (pl.CKW == 8 ? occ_cfg<MT_, NT_, WM_, WN_, 8, true, P_>(pl.lds_bytes) : occ_cfg<MT_, NT_, WM_, WN_, 4, true, P_>(pl.lds_bytes)) \
         : (pl.CKW == 8 ? occ_cfg<MT_, NT_, WM_, WN_, 8, false, P_>(pl.lds_bytes) : occ_cfg<MT_, NT_, WM_, WN_, 4, false, P_>(pl.lds_bytes)))
#define RD_OCC(MT_, NT_, WM_, WN_) \
    if (pl.MT == MT_ && pl.NT == NT_ && pl.WM == WM_ && pl.WN == WN_) \
        return pl.pipe ? RD_OCC2(MT_, NT_, WM_, WN_, true) : RD_OCC2(MT_, NT_, WM_, WN_, false);
    const bool swz = d->in_stride == 2 && !pl.grouped;
    RD_OCC(2, 2, 4, 1) RD_OCC(2, 1, 4, 1) RD_OCC(3, 2, 4, 1) RD_OCC(1, 2, 4, 1) RD_OCC(1, 1, 4, 1)
#undef RD_OCC
#undef RD_OCC2
    return -1;
}

// Plans are a pure function of the descriptor: cached, because the step is issued as ~650 plain launches per iteration and the
// tile search (tens of microseconds) would otherwise be repeated on every one of them.
struct PlanEntry { GconvPlan pl; RdConvDesc dd; int tuner_owned; };   // tuner_owned == 0: handed out by a plain query -- callers may
                                                                      // have sized buffers on it, so the tuner must not replace it
static std::mutex g_plan_mu;
static std::unordered_map<std::string, PlanEntry> g_plan_cache;
static int plan_query(const RdConvDesc* d, bool allow_split, GconvPlan& pl, RdConvDesc& dd, bool generic_only) {
    std::mutex& mu = g_plan_mu;
    std::unordered_map<std::string, PlanEntry>& cache = g_plan_cache;
    typedef PlanEntry Entry;
    RD_CHECK_ARG(d != nullptr, "gconv: null descriptor");
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    key.push_back(allow_split ? 1 : 0);
    if (generic_only) key.push_back('g');      // the fused (bias / activation) form of a conv16 descriptor: 32x32 kernels
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) { pl = it->second.pl; dd = it->second.dd; return RD_OK; }
    }
    int rc = validate_desc(d);
    if (rc != RD_OK) return rc;
    dd = *d;
    if (!generic_only && conv16_eligible(*d)) {
        pl = GconvPlan{};
        pl.c16 = 1; pl.TH = pl.TW = 16; pl.n_cotiles = 1; pl.ksplit = 1; pl.MT = pl.NT = pl.WM = pl.WN = 0;
        pl.tiles_total = conv16_tiles_per_image(*d);
        dd.phase[0].tile_begin = 0;
    } else {
        if (!plan_gconv(dd, pl, allow_split)) { set_error("gconv: no feasible tiling"); return RD_EINVAL; }
        fill_tiles(dd, pl);
    }
    std::lock_guard<std::mutex> lk(mu);
    cache.emplace(std::move(key), Entry{pl, dd, 0});
    return RD_OK;
}

// ---- plan tuner (cudnn.benchmark's role): the heuristic above is fitted to the large layers; for the small-spatial, few-tap and
// multi-phase launches the best (register tile, channel chunk, split, pipelined / plain loop) point is found by timing.  The
// caller lists the candidate plans of a descriptor, times rd_gconv_ws with each one pinned, and pins the winner; every later
// call with that descriptor (workspace / statistics-tile queries included) uses the pinned plan.  Candidate = 9 ints:
// MT, NT, WM, WN, CKP, TH, TW, ksplit, pipelined.
static bool same_point(const GconvPlan& p, const int32_t* c) {
    return p.MT == c[0] && p.NT == c[1] && p.WM == c[2] && p.WN == c[3] && p.CKP == c[4] && p.TH == c[5] && p.TW == c[6] && p.ksplit == c[7] &&
           p.pipe + p.wsd_half == c[8];
}
extern "C" int rd_gconv_tune_candidates(const RdConvDesc* d, int32_t allow_split, int32_t* out, int32_t max_candidates) {
    if (validate_desc(d) != RD_OK) return RD_EINVAL;
    RD_CHECK_ARG(out && max_candidates > 0, "gconv_tune_candidates: bad arguments");
    if (conv16_eligible(*d)) return 0;          // one kernel, one tiling
    {   // a descriptor somebody already planned without the tuner keeps its plan (its buffers were sized on it): nothing to tune
        std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
        key.push_back(allow_split ? 1 : 0);
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_plan_cache.find(key);
        if (it != g_plan_cache.end() && !it->second.tuner_owned) return 0;
    }
    std::vector<std::pair<double, GconvPlan>> all;
    GconvPlan best;
    RdConvDesc dd = *d;
    if (!plan_gconv(dd, best, allow_split != 0, &all)) { set_error("gconv: no feasible tiling"); return RD_EINVAL; }
    // one point per (register tile, chunk, split, loop form): the best-scoring pixel tile of each; best score first
    std::stable_sort(all.begin(), all.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    std::vector<GconvPlan> pick;
    for (const auto& sp : all) {
        const GconvPlan& p = sp.second;
        bool dup = false;
        for (const GconvPlan& q : pick)
            dup = dup || (q.MT == p.MT && q.NT == p.NT && q.CKP == p.CKP && q.ksplit == p.ksplit && q.pipe == p.pipe && q.wsd_half == p.wsd_half);
        if (!dup) pick.push_back(p);
    }
    int n = 0;
    for (const GconvPlan& p : pick) {
        if (n == max_candidates) break;
        const int v[9] = {p.MT, p.NT, p.WM, p.WN, p.CKP, p.TH, p.TW, p.ksplit, p.pipe + p.wsd_half};
        for (int i = 0; i < 9; ++i) out[n * 9 + i] = v[i];
        ++n;
    }
    return n;
}
// cand == NULL: back to the heuristic plan
extern "C" int rd_gconv_tune_pin(const RdConvDesc* d, int32_t allow_split, const int32_t* cand) {
    if (validate_desc(d) != RD_OK) return RD_EINVAL;
    RD_CHECK_ARG(!conv16_eligible(*d), "gconv_tune_pin: descriptor is served by the conv16 kernel");
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    key.push_back(allow_split ? 1 : 0);
    GconvPlan pl;
    RdConvDesc dd = *d;
    if (!cand) {
        if (!plan_gconv(dd, pl, allow_split != 0)) { set_error("gconv: no feasible tiling"); return RD_EINVAL; }
    } else {
        std::vector<std::pair<double, GconvPlan>> all;
        GconvPlan best;
        if (!plan_gconv(dd, best, allow_split != 0, &all)) { set_error("gconv: no feasible tiling"); return RD_EINVAL; }
        bool found = false;
        for (const auto& sp : all)
            if (!found && same_point(sp.second, cand)) { pl = sp.second; found = true; }
        RD_CHECK_ARG(found, "gconv_tune_pin: not a feasible plan of this descriptor");
    }
    fill_tiles(dd, pl);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    RD_CHECK_ARG(it == g_plan_cache.end() || it->second.tuner_owned, "gconv_tune_pin: this descriptor is already planned and in use");
    g_plan_cache[key] = PlanEntry{pl, dd, 1};
    return RD_OK;
}

// A pinned plan is provisional (tuner_owned) until it is COMMITTED: callers commit right after their last pin for a descriptor and
// before they size statistics tiles / workspaces on it.  From then on the entry behaves like a heuristic plan somebody uses:
// rd_gconv_tune_candidates returns 0 for it and rd_gconv_tune_pin refuses, so a later tuner (or table lookup) in the same process
// cannot swap the plan under buffers that were sized on it.  A descriptor nobody planned yet is planned (heuristically) and committed.
extern "C" int rd_gconv_tune_commit(const RdConvDesc* d, int32_t allow_split) {
    GconvPlan pl; RdConvDesc dd;
    int rc = plan_query(d, allow_split != 0, pl, dd, false);
    if (rc != RD_OK) return rc;
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    key.push_back(allow_split ? 1 : 0);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) it->second.tuner_owned = 0;
    return RD_OK;
}
// 0: not planned yet, 1: pinned by a tuner and still replaceable, 2: in use (heuristic plan handed out, or a committed pin)
extern "C" int rd_gconv_plan_state(const RdConvDesc* d, int32_t allow_split) {
    if (!d) return RD_EINVAL;
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    key.push_back(allow_split ? 1 : 0);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    return it == g_plan_cache.end() ? 0 : it->second.tuner_owned ? 1 : 2;
}

// The plan a launch uses.  With a workspace: the split-allowed plan.  Without one: the split-allowed plan when it does not split
// (callers size the statistics tiles with the _ws query and pass no workspace when it asks for none -- the tuner may have pinned a
// plan under that key that differs from the no-split heuristic), otherwise the no-split plan.  The heuristic gives the same plan
// under both keys whenever its split-allowed choice does not split, so this only matters for pinned plans.
static int plan_lookup(const RdConvDesc* d, bool has_ws, GconvPlan& pl, RdConvDesc& dd) {
    int rc = plan_query(d, true, pl, dd);
    if (rc != RD_OK || has_ws || pl.ksplit == 1) return rc;
    return plan_query(d, false, pl, dd);
}

extern "C" int rd_gconv_stat_tiles(const RdConvDesc* d) {
    GconvPlan pl; RdConvDesc dd;
    if (plan_lookup(d, false, pl, dd) != RD_OK) return RD_EINVAL;
    return d->N * pl.tiles_total;
}

// with a workspace the planner may split the input channels over several workgroups per tile (split-K)
extern "C" int64_t rd_gconv_workspace_floats(const RdConvDesc* d) {
    GconvPlan pl; RdConvDesc dd;
    if (plan_query(d, true, pl, dd) != RD_OK) return RD_EINVAL;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->N * d->Ho * d->Wo * d->ldo : 0;
}
extern "C" int rd_gconv_stat_tiles_ws(const RdConvDesc* d) {
    GconvPlan pl; RdConvDesc dd;
    if (plan_query(d, true, pl, dd) != RD_OK) return RD_EINVAL;
    if (pl.ksplit == 1) return d->N * pl.tiles_total;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    return (int)((M + combine_rows_per_block(M) - 1) / combine_rows_per_block(M));
}

static unsigned long long* g_trace_buf = nullptr;
// diagnostics: copy the stamps of the last traced launch (64 per workgroup) to the host
extern "C" int rd_gconv_trace_read(unsigned long long* host, int n_wg) {
    if (!g_trace_buf) return RD_EINVAL;
    RD_CHECK_HIP(hipMemcpy(host, g_trace_buf, (size_t)n_wg * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RD_OK;
}

struct BnBwdArgs { const float* x; int ld; const float* mean; const float* scale; const float* shift; int act; };

static int gconv_impl(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* addend,
                      int32_t ld_add, float* stat_partial, float* ws, void* stream, const float* bias = nullptr,
                      int act = RD_ACT_NONE, int act_cols = 0, const BnBwdArgs* bnb = nullptr) {
    RD_CHECK_ARG(in && w_packed && out, "gconv: null tensor");
    GconvArgs a;
    GconvPlan pl;
    int rc = plan_lookup(d, ws != nullptr, pl, a.d);
    if (rc != RD_OK) return rc;
    a.bnb_x = nullptr; a.bnb_mean = a.bnb_scale = a.bnb_shift = nullptr; a.bnb_ld = 0; a.bnb_act = RD_ACT_NONE;
    if (bnb) {
        RD_CHECK_ARG(!pl.c16 && pl.ksplit == 1 && d->in_stride == 1 && d->out_stride == 1 && d->n_phases == 1 && pl.CKW == 8,
                     "gconv_bnbwd: needs a single-phase unit-stride descriptor of at most nine taps whose plan neither splits nor runs on the "
                     "16-channel kernel (rd_gconv_bnbwd_supported)");
        a.bnb_x = bnb->x; a.bnb_ld = bnb->ld; a.bnb_mean = bnb->mean; a.bnb_scale = bnb->scale; a.bnb_shift = bnb->shift; a.bnb_act = bnb->act;
    }
    if (pl.c16) {
        if (!bias && act == RD_ACT_NONE) return launch_conv16(*d, in, w_packed, out, addend, ld_add, stat_partial, static_cast<hipStream_t>(stream));
        rc = plan_query(d, false, pl, a.d, true);      // inference form: bias / activation live in the 32x32 kernels' epilogue
        if (rc != RD_OK) return rc;
    }
    const bool split = pl.ksplit > 1;
    a.in = in; a.w = w_packed;
    a.out = split ? ws : out;
    a.addend = split ? nullptr : addend;
    a.stat = split ? nullptr : stat_partial;
    a.bias = split ? nullptr : bias;
    a.act = split ? RD_ACT_NONE : act;
    a.act_cols = split ? 0 : act_cols;
    a.ld_add = ld_add; a.ldw = d->Cout;
    a.TH = pl.TH; a.TW = pl.TW; a.PP = pl.PP; a.CKP = pl.CKP;
    a.tiles_total = pl.tiles_total; a.n_cotiles = pl.n_cotiles; a.taps_max = pl.taps_max; a.WSD = pl.WSD;
    a.ksplit = pl.ksplit;
    a.ngroups = 0;
    if (pl.grouped) {
        ParityGroups PG;
        build_parity_groups(*d, PG);
        a.ngroups = PG.n;
        a.g_sh_max = PG.sh_max; a.g_sw_max = PG.sw_max;
        const int PWg = pl.TW + PG.sw_max, PSg = pl.CKP + 4;
        for (int t = 0; t < 32; ++t) a.g_widx[t] = 0;
        for (int g = 0; g < 4; ++g) { a.g_ntaps[g] = a.g_oh[g] = a.g_ow[g] = a.g_tbase[g] = 0; }
        for (int g = 0; g < PG.n; ++g) {
            a.g_ntaps[g] = PG.ntaps[g]; a.g_oh[g] = PG.oh[g]; a.g_ow[g] = PG.ow[g]; a.g_tbase[g] = PG.tbase[g];
            for (int t = 0; t < PG.ntaps[g]; ++t) {
                const int k = PG.tbase[g] + t;
                a.g_widx[k] = (short)PG.widx[k];
                a.tapoff[g][t] = (PG.sh[k] * PWg + PG.sw[k]) * PSg * 4;
            }
        }
    } else {
        const bool swz_ = d->in_stride == 2;
        const int PS_ = swz_ ? pl.CKP : pl.CKP + 4;
        for (int i = 0; i < d->n_phases; ++i) {
            const RdPhase& p = d->phase[i];
            const int PW_ = (pl.TW - 1) * d->in_stride + (p.dw_max - p.dw_min) + 1;
            for (int t = 0; t < p.n_taps; ++t) {
                const int pix = (p.dh[t] - p.dh_min) * PW_ + (p.dw[t] - p.dw_min);
                a.tapoff[i][t] = swz_ ? pix : pix * PS_ * 4;
            }
        }
    }
    a.split_stride = (long long)d->N * d->Ho * d->Wo * d->ldo;
    { static const char* dbg = getenv("RD_GCONV_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
    a.trace = nullptr;
    a.lds_floats = (int)((pl.lds_bytes + 7) / 8) * 2;
    {
        static const char* tr = getenv("RD_GCONV_TRACE");
        if (tr && atoi(tr)) {
            static unsigned long long* buf = nullptr;
            if (!buf) RD_CHECK_HIP(hipMalloc(&buf, (size_t)65536 * 64 * sizeof(unsigned long long)));
            g_trace_buf = buf;
            a.trace = buf;
        }
    }
    const int grid = d->N * pl.tiles_total * pl.n_cotiles * pl.ksplit;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = RD_EINVAL;
    bool launched = false;
#define RD_LAUNCH2(MT_, NT_, WM_, WN_, P_)                                                           \
    (swz ? (pl.CKW == 8 ? launch_cfg<MT_, NT_, WM_, WN_, 8, true, P_>(a, grid, lds_launch, s)             \
                        : launch_cfg<MT_, NT_, WM_, WN_, 4, true, P_>(a, grid, lds_launch, s))            \
         : pl.grouped ? (pl.CKW == 8 ? launch_cfg<MT_, NT_, WM_, WN_, 8, false, P_, true>(a, grid, lds_launch, s)   \
                                     : launch_cfg<MT_, NT_, WM_, WN_, 4, false, P_, true>(a, grid, lds_launch, s))  \
         : (pl.CKW == 8 ? launch_cfg<MT_, NT_, WM_, WN_, 8, false, P_>(a, grid, lds_launch, s)            \
                        : launch_cfg<MT_, NT_, WM_, WN_, 4, false, P_>(a, grid, lds_launch, s)))
#define RD_TRY(MT_, NT_, WM_, WN_)                                                                  \
    if (!launched && pl.MT == MT_ && pl.NT == NT_ && pl.WM == WM_ && pl.WN == WN_) {                \
        launched = true;                                                                            \
        if (a.bnb_x)                                                                                \
            rc = pl.pipe ? launch_cfg<MT_, NT_, WM_, WN_, 8, false, true, false, true>(a, grid, lds_launch, s)    \
                         : launch_cfg<MT_, NT_, WM_, WN_, 8, false, false, false, true>(a, grid, lds_launch, s);  \
        else                                                                                        \
            rc = pl.pipe ? RD_LAUNCH2(MT_, NT_, WM_, WN_, true) : RD_LAUNCH2(MT_, NT_, WM_, WN_, false);              \
    }
    const bool swz = d->in_stride == 2 && !pl.grouped;
    const size_t lds_launch = (size_t)a.lds_floats * 4 + (a.trace ? 512 : 0);
    RD_TRY(2, 2, 4, 1)
    RD_TRY(2, 1, 4, 1)
    RD_TRY(3, 2, 4, 1)
    RD_TRY(1, 2, 4, 1)
    RD_TRY(1, 1, 4, 1)
#undef RD_TRY
#undef RD_LAUNCH2
    if (!launched) { set_error("gconv: unsupported plan"); return RD_EINVAL; }
    if (rc != RD_OK || !split) return rc;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const int RPB = combine_rows_per_block(M), blocks = (int)((M + RPB - 1) / RPB);
    const int Q = d->Cout / 4, RL = 256 / Q;
    RD_CHECK_ARG(Q >= 1 && Q <= 256, "gconv: split-K combine needs Cout <= 1024");
    hipLaunchKernelGGL(gconv_combine_kernel, dim3(blocks), dim3(256), (size_t)RL * 2 * d->Cout * sizeof(float), s, ws, a.split_stride,
                       pl.ksplit, out, d->ldo, addend, ld_add, M, d->Cout, RPB, stat_partial, bias, act, act_cols);
    RD_CHECK_LAUNCH("gconv_combine_kernel");
    return RD_OK;
}

extern "C" int rd_gconv(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* addend,
                        int32_t ld_add, float* stat_partial, void* stream) {
    return gconv_impl(d, in, w_packed, out, addend, ld_add, stat_partial, nullptr, stream);
}

extern "C" int rd_gconv_ws(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* addend,
                           int32_t ld_add, float* stat_partial, float* ws, void* stream) {
    return gconv_impl(d, in, w_packed, out, addend, ld_add, stat_partial, ws, stream);
}

// Inference form: out = act_{co < act_cols}(conv(in, w) + bias[co] + addend).  With the BatchNorm scale
// folded into w (rd_pack_weights_batched with a per-channel scale) and bias = the folded shift this is conv+BN+ReLU(+residual)
// in one kernel (validate() body, main.py:564-595).
extern "C" int rd_gconv_fused(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* bias,
                              int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* ws, void* stream) {
    return gconv_impl(d, in, w_packed, out, addend, ld_add, nullptr, ws, stream, bias, act, act_cols);
}

// Input gradient of a convolution whose forward input was act(BatchNorm(x)) (models.py:96-112: conv2 of a BasicBlock behind
// bn1 + relu; :203-206: an UpProj module's 3x3 behind batchnorm1 + relu), with the BatchNorm-backward sums of THAT BatchNorm
// taken in the epilogue: red_partial[tiles][3][Cout] receives sum g (slot 0) and sum g*(x - mean) (slot 1), g = dy * act'(scale*x
// + shift), over each pixel tile (tiles = rd_gconv_stat_tiles_ws(d)); rd_bn_bwd_apply_x_t consumes it unchanged.
extern "C" int rd_gconv_bnbwd_supported(const RdConvDesc* d) {
    GconvPlan pl; RdConvDesc dd;
    if (plan_query(d, true, pl, dd) != RD_OK) return 0;
    return (!pl.c16 && pl.ksplit == 1 && d->n_phases == 1 && d->out_stride == 1 && d->in_stride == 1 && pl.CKW == 8) ? 1 : 0;
}
extern "C" int rd_gconv_bnbwd(const RdConvDesc* d, const float* dout, const float* w_packed, float* dx, const float* bn_x, int32_t ld_x,
                              const float* bn_mean, const float* bn_scale, const float* bn_shift, int32_t act, float* red_partial,
                              float* ws, void* stream) {
    RD_CHECK_ARG(bn_x && bn_mean && bn_scale && bn_shift && red_partial, "gconv_bnbwd: null argument");
    const BnBwdArgs b{bn_x, ld_x, bn_mean, bn_scale, bn_shift, act};
    return gconv_impl(d, dout, w_packed, dx, nullptr, 0, red_partial, ws, stream, nullptr, RD_ACT_NONE, 0, &b);
}
