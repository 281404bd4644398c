// fp32 convolution on the bf16 matrix cores: every fp32 operand is split into three bf16 pieces and the product is rebuilt
// from six bf16 MFMAs with fp32 accumulation.
//
//   x = x0 + x1 + x2,  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)          (exact: 3 x 8 significant bits = fp32's 24)
//   x * w = x0 w0 + (x0 w1 + x1 w0) + (x0 w2 + x1 w1 + x2 w0) + [x1 w2 + x2 w1 + x2 w2]
// The bracketed terms are below 2^-24 |x w| -- the size of the rounding of ONE fp32 product -- and are dropped; each kept product
// of two 8-bit significands is exact in fp32, a v_mfma_f32_32x32x16_bf16 sums sixteen of them and rounds once into the fp32
// accumulator (six roundings per 16 reduction elements; the fp32 MFMA of gconv.hip rounds eight times for the same sixteen).
// Measured against an fp64 convolution the results are as close as gconv.hip's (tests/test_gpu_gconv_split.py).  Cost: six
// 32-cycle instructions per 16 reduction elements instead of eight 64-cycle ones: 2.67x the fp32 MFMA rate.
//
// Same descriptor (phases x taps over an NHWC halo patch), same fp32 tensors in HBM, same epilogue as gconv.hip / gconv_bf16.hip;
// opt-in per plan (engine operands = "split"), gconv.hip stays the default and the parity reference.
//
//   workgroup : 8 waves, ONE per CU.  Waves 0-3 own the accumulators (MT x NT tiles of 32 x 32 each) and do nothing but LDS
//               fragment reads and MFMAs; waves 4-7 stage: weight pieces by global_load_lds, the next 16-channel chunk of the
//               patch through registers, where they split it into pieces (11 VALU instructions per pair of values: hidden
//               beside the other waves' MFMAs; inside one wave they would sit between MFMA bursts).
//   LDS       : patch [piece][pixel][16 + 8] bf16 (pixel pitch 48 B: the 16 lanes of a b128 read pass fall into 16 bank groups), two
//               buffers of it where they fit (PDB: the 2 x 2 tile); weights [ring of 3][piece][3 taps][2][BN] x 16 B, staged in
//               groups of three taps (all nine at once x 3 pieces x 2 buffers = 110 KB); a phase's last group is padded with
//               zero-weight taps, so every group is three MFMA steps.
//   loop      : one barrier per (16-channel chunk, tap group); it publishes the NEXT group's weights (issued a group earlier), so the
//               compute waves read that group's first fragments before the barrier and continue straight behind it.  The patch of
//               the next chunk is fetched a chunk ahead, split in its second group and stored in its last one -- into the other
//               buffer (PDB), or behind a second barrier into the only one.  Details at the loops below.
//   operand   : packed weights [piece][slab][Cin/8][ldw][8] bf16 (rd_pack_weights_batched, quad == 3).
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "slot_map.h"

namespace rd {

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned GS_OOB = 0x80000000u;
constexpr int GS_CKP = 16;                    // input channels per chunk = one MFMA step per tap
constexpr int GS_PSB = (GS_CKP + 8) * 2;      // patch pixel pitch in bytes
constexpr int GS_UPP = 6;                     // patch units (8 channels of one pixel) per loader thread
constexpr int GS_UW = 2;                      // weight units per loader thread, piece and tap group
constexpr int GS_TPS = 3;                     // taps per weight group

struct GsArgs {
    RdConvDesc d;
    const float* in;
    const unsigned short* w;      // packed operand, three piece planes
    float* out;
    const float* addend;
    const float* bias;
    float* stat;
    int act, act_cols, ld_add, ldw;
    int TH, TW, PP, tiles_total, n_cotiles, taps_max;
    int vec4;
    int pplane;                   // bytes per piece plane of the LDS patch
    long long wplane;             // bytes per piece plane of the packed operand
    int tapoff[RD_MAX_PHASES][RD_MAX_TAPS];   // byte offset of tap t inside a patch plane
    unsigned long long* trace;    // diagnostics (RD_GCONV_SPLIT_TRACE=1 compute wave 0, =2 staging wave 4): 64 stamps per workgroup
    int trace_role;
    // PRE: the activation arrives already split (rd_split_pieces / the producers' epilogues): piece planes [piece][Cin/16][pixel][16] bf16
    const unsigned short* inp;
    long long xplane;             // bytes per piece plane of the pre-split activation
    long long mpix;               // pixels per 16-channel block of a plane (= N * Hi * Wi)
    int kplane;                   // PRE: bytes per 8-channel unit plane of the LDS patch ([piece][unit][pixel] x 16 B)
    int stagger;                  // gconv_sp2_kernel: start delay of every second group of 256 blocks, in units of 64 clocks
    int dbg;                      // diagnostics (RD_GCONV_SPLIT_DEBUG, ablations for tools/ablate_gconv_split.py; results are then garbage):
                                  // 4 no weight copies, 8 no patch copies / staging, 16 no epilogue; 1 no MFMAs, 2 no fragment reads
                                  // (the last two as template instantiations of gconv_sp2_kernel<2,2> only)
    // slot map (gs_slot_pixel, computed by the host once per plan): LDS row pitch of the patch in pixels per phase (patch width + 0..3
    // pad columns) and the table [phase][BM]: (r << 16) | c of the tile pixel slot m holds, -(1 + residue of the lane) for an empty slot
    int ppitch[RD_MAX_PHASES];
    const int* slots;
    // BNB instantiations (rd_gconv_split_bnbwd / rd_gconv_split_pre_bnbwd): this launch is the input gradient of a convolution whose forward
    // input was act(BatchNorm(bnb_x)); the epilogue also emits that BatchNorm's backward sums -- sum g and sum g (x - mean), g = dx * act'
    // -- into stat as [tile][3][Cout] (gconv.hip's BNB contract: the separate reduce pass over dx and x disappears)
    const float* bnb_x;
    const float* bnb_scale;
    const float* bnb_shift;
    const float* bnb_mean;
    int bnb_ld, bnb_act;
};

// wait until at most n of this wave's vector-memory operations (global_load_lds copies included) are outstanding, n known only at
// run time (s_waitcnt takes an immediate); n beyond the table waits for a few more than necessary, which is always safe
__device__ __forceinline__ void vm_wait_upto(int n) {
#define RD_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)" ::: "memory"); break;
    switch (n < 47 ? n : 47) {
        RD_VMW(0) RD_VMW(1) RD_VMW(2) RD_VMW(3) RD_VMW(4) RD_VMW(5) RD_VMW(6) RD_VMW(7) RD_VMW(8) RD_VMW(9) RD_VMW(10) RD_VMW(11)
        RD_VMW(12) RD_VMW(13) RD_VMW(14) RD_VMW(15) RD_VMW(16) RD_VMW(17) RD_VMW(18) RD_VMW(19) RD_VMW(20) RD_VMW(21) RD_VMW(22) RD_VMW(23)
        RD_VMW(24) RD_VMW(25) RD_VMW(26) RD_VMW(27) RD_VMW(28) RD_VMW(29) RD_VMW(30) RD_VMW(31) RD_VMW(32) RD_VMW(33) RD_VMW(34) RD_VMW(35)
        RD_VMW(36) RD_VMW(37) RD_VMW(38) RD_VMW(39) RD_VMW(40) RD_VMW(41) RD_VMW(42) RD_VMW(43) RD_VMW(44) RD_VMW(45) RD_VMW(46) RD_VMW(47)
    }
#undef RD_VMW
}

// the three bf16 pieces of eight fp32 values (round to nearest even at every level; the remainders are exact), two values at a
// time so that every conversion is one v_cvt_pk_bf16_f32 and the packed result IS the piece operand: 11 VALU instructions per pair
// (element by element the compiler spends one conversion per value and packs afterwards: 17)
typedef float sf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    sf32x2 v;
    v[0] = a; v[1] = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sbf16x2));
}
__device__ __forceinline__ void split8(const float4 v0, const float4 v1, sbf16x8& p0, sbf16x8& p1, sbf16x8& p2) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    su32x4 w0, w1, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = x[2 * i], b = x[2 * i + 1];
        const unsigned u0 = cvt_pk(a, b);
        a -= __uint_as_float(u0 << 16);
        b -= __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = cvt_pk(a, b);
        a -= __uint_as_float(u1 << 16);
        b -= __uint_as_float(u1 & 0xffff0000u);
        w0[i] = u0;
        w1[i] = u1;
        w2[i] = cvt_pk(a, b);
    }
    p0 = __builtin_bit_cast(sbf16x8, w0);
    p1 = __builtin_bit_cast(sbf16x8, w1);
    p2 = __builtin_bit_cast(sbf16x8, w2);
}

// PRE: the input was split by its producer (piece planes in HBM): the staging waves issue nothing but global_load_lds copies -- patch
// rows land as [piece][8-channel unit][patch pixel] x 16 B (lane-linear: consecutive pixels in consecutive 16-byte slots, which is
// also what keeps the compute waves' ds_read_b128 passes conflict-free), out-of-image positions are zeroed once per workgroup and
// never written again (their lanes are masked in every copy).
template <int MT, int NT, bool PDB, bool PRE, bool BNB = false>
__global__ __launch_bounds__(512) void gconv_split_kernel(const GsArgs a) {
    static_assert(!PRE || PDB, "the pre-split form copies the next chunk's patch while the current one is read: two patch buffers");
    constexpr int BM = 4 * MT * 32;
    constexpr int BN = NT * 32;
    constexpr int LBN = NT == 2 ? 6 : 5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int wm = wave & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;

    if (a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 64 + 62] = __builtin_readcyclecounter();
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = vid % a.n_cotiles;
    const int pt = vid / a.n_cotiles;
    const int n = pt / a.tiles_total;
    const int tt = pt - n * a.tiles_total;
    int ph_ = 0;
    for (int i = 1; i < D.n_phases; ++i)
        if (tt >= D.phase[i].tile_begin) ph_ = i;
    const int ph = __builtin_amdgcn_readfirstlane(ph_);
    const RdPhase& P = D.phase[ph];
    const int tloc = tt - P.tile_begin;
    const int tiles_w = (P.lw + a.TW - 1) / a.TW;
    const int r0 = (tloc / tiles_w) * a.TH, c0 = (tloc % tiles_w) * a.TW;
    const int th_n = min(a.TH, P.lh - r0), tw_n = min(a.TW, P.lw - c0);
    const int IS = D.in_stride, OS = D.out_stride;
    const int PW = (a.TW - 1) * IS + (P.dw_max - P.dw_min) + 1;
    const int PWP = a.ppitch[ph];                           // LDS row pitch of the patch in pixels (>= PW, gs_slot_pixel)
    const int PH = (th_n - 1) * IS + (P.dh_max - P.dh_min) + 1;
    const int ih0 = r0 * IS + P.dh_min, iw0 = c0 * IS + P.dw_min;
    const int ntaps = __builtin_amdgcn_readfirstlane(P.n_taps);
    const int ngroups = (ntaps + GS_TPS - 1) / GS_TPS;      // the last group is padded with zero-weight taps: every group is 3 steps
    const int co0 = cot * BN;

    // LDS carve-up
    int* s_opix = reinterpret_cast<int*>(smem);          // [BM] output pixel index or -1
    int* s_apix = s_opix + BM;                           // [BM] patch pixel index of tap (0,0)
    int* s_widx = s_apix + BM;                           // [32] weight slab index of each tap
    char* s_w = reinterpret_cast<char*>(s_widx + 32);    // [3][3][GS_TPS][2][BN] x 16 B: ring of three tap groups
    constexpr int WPP = GS_TPS * 2 * BN * 16;            // bytes per piece of one weight buffer
    constexpr int SLAB = 3 * WPP;
    char* s_patch = s_w + 3 * SLAB;                      // [PDB ? 2 : 1][3][pplane]
    const int pplane = a.pplane;

    for (int m = tid; m < BM; m += 512) {
        const int sv = a.slots[ph * BM + m];
        const bool have = sv >= 0;
        const int r = sv >> 16, c = sv & 0xffff, rho = -1 - sv;
        const bool ok = have && (r < th_n) && (c < tw_n);
        s_opix[m] = ok ? ((n * D.Ho + (r0 + r) * OS + P.out_off_h) * D.Wo + (c0 + c) * OS + P.out_off_w) : -1;
        // (masked pixels of an edge tile keep their own patch address -- staged or not, their accumulator rows are never stored;
        //  empty slots read the first patch row at their lane's residue: no bank shared with the pass's live lanes)
        s_apix[m] = have ? ((r * IS) * PWP + c * IS) : rho * IS;
    }
    if (tid < ntaps) s_widx[tid] = P.widx[tid];
    if constexpr (PRE) {
        // both patch buffers start as zeros: padding pixels (and rows) of the halo are never copied
        const int n16 = (2 * 3 * pplane) >> 4;
        for (int e = tid; e < n16; e += 512) *reinterpret_cast<uint4*>(s_patch + (size_t)e * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    rd_sync();

    f32x16 acc[MT][NT];      // (zeroed in the compute branch only: live registers of the staging waves otherwise)

    const int nchunks = D.Cin / GS_CKP;
    const int total_groups = nchunks * ngroups;
    const char* in_n = reinterpret_cast<const char*>(a.in) + (size_t)n * D.Hi * D.Wi * D.ldi * 4;
    const unsigned img_bytes = (unsigned)(D.Hi * D.Wi * D.ldi) * 4u;
    // patch units: 64-unit row segments (two 8-channel units per pixel of a 16-channel chunk)
    const int rowu = PW * 2;
    const int nseg = (rowu + 63) >> 6;
    const int nsegs = PH * nseg;
    auto unit_of = [&](int seg, unsigned& goff, int& ldst) {
        const int row = nseg == 1 ? seg : seg / nseg;
        const int cu = ((seg - row * nseg) << 6) + lane;
        const int px = cu >> 1, qq = cu & 1;
        const int ih = ih0 + row, iw = iw0 + px;
        const bool ok = seg < nsegs && cu < rowu;
        ldst = ok ? (row * PWP + px) * GS_PSB + qq * 16 : -1;
        goff = (ok && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi) ? (unsigned)(((ih * D.Wi + iw) * D.ldi + qq * 8) * 4) : GS_OOB;
    };
    // The FIRST chunk's patch is staged by all eight waves (segment = wave + 8 k): the MFMA waves have nothing else to do before
    // the first barrier, and the prologue -- HBM latency + split + store of a whole patch by four waves -- was ~10 k clocks of a
    // 64-channel layer's ~65 k per workgroup.
    auto stage_first_chunk = [&]() {
        constexpr int UP8 = (GS_UPP + 1) / 2;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_n), 0, img_bytes, 0x00020000);
        float4 f0[UP8], f1[UP8];
        int dst[UP8];
#pragma unroll
        for (int u = 0; u < UP8; ++u) {
            unsigned go;
            unit_of(wave + 8 * u, go, dst[u]);
            f0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)go, 0, 0));
            f1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)go + 16, 0, 0));
        }
        const unsigned base = (unsigned)(size_t)s_patch;
#pragma unroll
        for (int u = 0; u < UP8; ++u) {
            sbf16x8 q0, q1, q2;
            split8(f0[u], f1[u], q0, q1, q2);
            if (dst[u] >= 0) {
                const su32x4 d0 = __builtin_bit_cast(su32x4, q0), d1 = __builtin_bit_cast(su32x4, q1), d2 = __builtin_bit_cast(su32x4, q2);
                const unsigned ad = base + dst[u];
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(d0) : "memory");
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad + pplane), "v"(d1) : "memory");
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad + 2 * pplane), "v"(d2) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Barrier protocol (B1 once per tap group g of chunk c, iteration `it`; every wave of the workgroup takes part):
    //   B1(it) publishes weight group it + 1 (issued behind B1(it - 1), awaited by the loaders before they arrive) and frees the
    //          ring buffer of group it - 1, into which the loaders then issue group it + 2;
    //   patch : the loaders fetch chunk c + 1 in the chunk's first group, split it in the second, and write it in the last one --
    //           PDB: into the other patch buffer, published by the next B1; else behind a second barrier B2 (the compute waves
    //           are then done with the patch) into the only one.
    // The compute waves read the first fragments of group it + 1 BEFORE B1(it + 1) (they are published), so the MFMAs continue
    // straight behind the barrier; only at a chunk boundary do they have to wait for the new patch first.
    if (loader) {
        // ------------------------------------------------------------------------------------------ staging waves
        const int ltid = tid - 256;
        const int lw = wave - 4;
        const int cin8 = D.Cin >> 3;
        // patch units of this thread in the steady state: wave lw copies the row segments lw, lw + 4, ...
        unsigned pgo[GS_UPP];     // byte offset of the unit inside the image at channel 0; GS_OOB: outside -> zeros
        int pdst[GS_UPP];         // LDS byte offset inside a patch plane, -1: no such unit
        if constexpr (!PRE) {
#pragma unroll
            for (int u = 0; u < GS_UPP; ++u) unit_of(lw + 4 * u, pgo[u], pdst[u]);
        }
        // PRE: copy the patch of 16-channel block `blk` into patch buffer pbuf: one global_load_lds per (patch row, 64-pixel segment,
        // piece, 8-channel unit), rows lw, lw + 4, ... of this wave; returns the number of copies issued (the caller's counted wait)
        auto issue_patch = [&](int pbuf, int blk) -> int {
            int cnt = 0;
            if (a.dbg & 8) return 0;
            const char* src0 = reinterpret_cast<const char*>(a.inp) + ((size_t)blk * a.mpix + (size_t)n * D.Hi * D.Wi) * 32;
            char* dst0 = s_patch + pbuf * 3 * pplane;
            const int nsg = (PW + 63) >> 6;
            for (int r = lw; r < PH; r += 4) {
                const int ih = ih0 + r;
                if (ih < 0 || ih >= D.Hi) continue;
                for (int sg = 0; sg < nsg; ++sg) {
                    const int col = (sg << 6) + lane, iw = iw0 + col;
                    const bool ok = col < PW && iw >= 0 && iw < D.Wi;
                    if (!__any(ok)) continue;
                    const char* src = src0 + ((size_t)ih * D.Wi + iw) * 32;
                    char* dst = dst0 + ((size_t)r * PWP + (sg << 6)) * 16;
                    if (ok) {
#pragma unroll
                        for (int p = 0; p < 3; ++p)
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                glds16(reinterpret_cast<const float*>(src + p * a.xplane + u * 16), reinterpret_cast<float*>(dst + p * pplane + u * a.kplane));
                    }
                    cnt += 6;
                }
            }
            return cnt;
        };
        // (returns the number of copies this wave issued: the pre-split protocol's counted waits)
        auto issue_slab = [&](int buf, int cb, int g) -> int {
            if (a.dbg & 4) return 0;
            int cnt = 0;
            const int tap0 = g * GS_TPS;
            constexpr int welems = (GS_TPS * 2) << LBN;  // 16-byte units of one piece: [tap][2][BN]
            const char* src = reinterpret_cast<const char*>(a.w) + (size_t)(cb >> 3) * a.ldw * 16;
            char* dst = s_w + buf * SLAB;
#pragma unroll
            for (int u = 0; u < GS_UW; ++u) {
                const int e = ltid + u * 256;
                if (e < welems) {
                    const int j = e & (BN - 1), tk = e >> LBN;
                    const int k8 = tk & 1, t = tap0 + (tk >> 1);
                    const bool live = t < ntaps && co0 + j < D.Cout;
                    if (__any(live)) cnt += 3;
                    if (live) {
                        const size_t go = (((size_t)s_widx[t] * cin8 + k8) * a.ldw + co0 + j) * 16;
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            glds16(reinterpret_cast<const float*>(src + p * a.wplane + go), reinterpret_cast<float*>(dst + p * WPP + (e - lane) * 16));
                    } else {                             // padding taps of the last group, output channels beyond Cout
                        // (inline assembly: a plain LDS store behind outstanding global_load_lds copies makes the compiler wait for them)
                        const su32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const unsigned ad = (unsigned)(size_t)(dst + p * WPP + e * 16);
                            asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(z) : "memory");
                        }
                    }
                }
            }
            return cnt;
        };
        sbf16x8 pc[GS_UPP][3];
        float4 v0[GS_UPP], v1[GS_UPP];
        auto fetch = [&](int cb) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_n + cb * 4), 0, img_bytes - cb * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < GS_UPP; ++u) {
                v0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)pgo[u], 0, 0));
                v1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)pgo[u] + 16, 0, 0));
            }
        };
        auto split_units = [&](auto u0_, auto u1_) {
#pragma unroll
            for (int u = decltype(u0_)::value; u < decltype(u1_)::value; ++u) {
                split8(v0[u], v1[u], pc[u][0], pc[u][1], pc[u][2]);
                __builtin_amdgcn_sched_barrier(0);      // one unit at a time: interleaved, the units' temporaries cost ~100 registers
            }
        };
        // (ds_write_b128 as inline assembly: behind outstanding global_load_lds copies the compiler makes a plain LDS store wait for
        //  the copies -- they could alias for all it knows; these stores go to the patch, the copies to the weight ring.  Completion
        //  is covered by the explicit lgkmcnt(0) in front of every barrier.)
        auto put_units = [&](int pbuf, auto u0_, auto u1_) {
            const unsigned base = (unsigned)(size_t)(s_patch + pbuf * 3 * pplane);
#pragma unroll
            for (int u = decltype(u0_)::value; u < decltype(u1_)::value; ++u)
                if (pdst[u] >= 0) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned addr = base + p * pplane + pdst[u];
                        const su32x4 dv = __builtin_bit_cast(su32x4, pc[u][p]);
                        asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(dv) : "memory");
                    }
                }
        };
        constexpr std::integral_constant<int, 0> U0{};
        constexpr std::integral_constant<int, GS_UPP> UN{};
        issue_slab(0, 0, 0);
        if (total_groups > 1) issue_slab(1, ngroups > 1 ? 0 : GS_CKP, ngroups > 1 ? 1 : 0);
        if constexpr (PRE) {
            // Lazy protocol: a copy issued behind B1(it) has until B1(it + 2) to land.  B1(it + 1) only needs what was issued BEFORE this
            // iteration (weights of group it + 1; at a chunk's end the successor's patch, issued in the chunk's first group), so the
            // wait in front of it leaves this iteration's own copies in flight (counted: vmcnt retires in order).  The barriers are raw
            // s_barriers behind an explicit LDS wait: __syncthreads would add a vmcnt(0) for the copies in flight (measured: 7.5 k clocks
            // in a chunk's first group against 4.9 k of MFMA work, profiles/r04_trace_gconv_split_pre.txt).  The compute waves therefore
            // read a group's weights only behind its own barrier (no fragment prefetch across it).
            auto lbarrier = [&]() {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            };
            issue_patch(0, 0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            int it = 0, wi = 2, c2 = 0, g2 = 2;
            while (g2 >= ngroups) { g2 -= ngroups; ++c2; }
            unsigned long long* ltr = (a.trace && a.trace_role == 2 && tid == 256) ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
            for (int c = 0; c < nchunks; ++c) {
                for (int g = 0; g < ngroups; ++g, ++it) {
                    if (ltr && it < 30) ltr[2 * it] = __builtin_readcyclecounter();
                    lbarrier();                       // B1(it)
                    if (ltr && it < 30) ltr[2 * it + 1] = __builtin_readcyclecounter();
                    int n_now = 0;
                    if (it + 2 < total_groups) n_now += issue_slab(wi, c2 * GS_CKP, g2);
                    wi = wi == 2 ? 0 : wi + 1;
                    if (++g2 == ngroups) { g2 = 0; ++c2; }
                    if (g == 0 && c + 1 < nchunks) n_now += issue_patch((c + 1) & 1, c + 1);
                    vm_wait_upto(ngroups > 1 ? n_now : 0);        // (one group per chunk: the patch issued here is needed at the next barrier)
                }
            }
        } else {
        stage_first_chunk();
        // the second chunk's patch is fetched behind the FIRST barrier when its split comes a group later (three or more tap groups:
        // issuing the loads here cost the prologue ~1.4 k clocks); else here, with a counted wait that leaves them in flight
        const bool defer_fetch = ngroups >= 3;
        if (nchunks > 1 && !defer_fetch) {
            fetch(GS_CKP);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * GS_UPP) : "memory");
        } else {
            glds_wait();
        }
        int it = 0, wi = 2;                       // wi: ring buffer of group it + 2
        int c2 = 0, g2 = 2;                       // (chunk, group) of group it + 2
        while (g2 >= ngroups) { g2 -= ngroups; ++c2; }
        // the patch of chunk c + 1 is fetched one chunk ahead (in the last group of chunk c - 1: HBM latency under load is more than a
        // tap group), split into pieces in the second group of chunk c and written in its last group
        const int gsplit = ngroups > 2 ? 1 : 0;
        unsigned long long* ltr = (a.trace && a.trace_role == 2 && tid == 256) ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            for (int g = 0; g < ngroups; ++g, ++it) {
                if (ltr && it < 30) ltr[2 * it] = __builtin_readcyclecounter();
                // B1(it) as a RAW s_barrier behind an explicit LDS wait (round 6).  rd_sync()'s __syncthreads is a workgroup-scope release fence:
                // with stores possibly in flight -- the diagnostic stamps of `ltr` above are global stores, taken or not -- hipcc puts an
                // s_waitcnt vmcnt(0) in front of the barrier, and on gfx9 that counter also holds this wave's LOADS: the next chunk's patch
                // fetch, which the counted wait at the end of the previous iteration had deliberately left in flight, was drained at every
                // tap group (ISA: `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` in this loop).  What this barrier publishes -- the weight
                // copies and the patch stores of the previous iteration -- has been waited for explicitly at that iteration's end.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (ltr && it < 30) ltr[2 * it + 1] = __builtin_readcyclecounter();
                // this group's weight copies first: the split arithmetic / the patch stores below run while they are in flight
                if (it + 2 < total_groups) issue_slab(wi, c2 * GS_CKP, g2);
                wi = wi == 2 ? 0 : wi + 1;
                if (++g2 == ngroups) { g2 = 0; ++c2; }
                const bool first_fetch = it == 0 && nchunks > 1 && defer_fetch;
                const bool refetch = (g == ngroups - 1 && c + 2 < nchunks) || first_fetch;
                if (more) {
                    // (spreading the split arithmetic over the first two groups was slower, with the stores -- 3x2 tile 139 -> 150 us on
                    //  layer3 -- and without -- 170 -> 189 us on layer1: in the first group the patch loads have not landed yet)
                    if (g == gsplit) split_units(U0, UN);
                    if (g == ngroups - 1) {
                        if constexpr (!PDB) {             // B2: the compute waves are done with this chunk's patch (raw barrier: see B1)
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                        put_units(PDB ? ((c + 1) & 1) : 0, U0, UN);
                    }
                }
                if (refetch) fetch(first_fetch ? GS_CKP : (c + 2) * GS_CKP);
                // the weights issued above must have landed before the next barrier; the patch loads issued BEHIND them in this
                // group (2 * GS_UPP buffer loads per wave, unconditionally) need not: a plain vmcnt(0) would hold the next barrier
                // back by the HBM latency (~1400 clocks of the MFMA waves per chunk, tools/trace_gconv_split.py)
                if (refetch) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * GS_UPP) : "memory");
                else glds_wait();
            }
        }
        }
    } else {
        // ------------------------------------------------------------------------------------------ compute waves
        if constexpr (!PRE) stage_first_chunk();
        // (s_setprio(3) for these waves starves the staging waves' split arithmetic: 1880 -> 4760 clocks per chunk, the MFMA waves then
        //  wait for the patch: 2x2 tile 182 -> 195 us on layer1)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        int aoffB[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            aoffB[mt] = PRE ? s_apix[(wm * MT + mt) * 32 + l31] * 16 + hh * a.kplane : s_apix[(wm * MT + mt) * 32 + l31] * GS_PSB + hh * 16;
        const int boffB = (hh * BN + l31) * 16;
        // tap offsets in the lanes of one VGPR, fetched with v_readlane (an s_load in the walk would force lgkmcnt(0) waits);
        // the padding taps of the last group read tap 0's pixels against zero weights
        const int tapv = a.tapoff[ph][lane < ntaps ? lane : 0];
        // fragment addresses: A = patch plane p, this lane's pixel of M-tile mt (+ the tap's offset, wave-uniform, added per
        // step); B = one base per ring buffer, everything else (piece, tap of the group, N-tile) is an immediate
        int abase[3][MT];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) abase[p][mt] = p * pplane + aoffB[mt];
        auto load = [&](const char* pbase, const char* wbuf, int tap, int t, sbf16x8 (&A)[3][MT], sbf16x8 (&B)[3][NT]) {
            const int ao = __builtin_amdgcn_readlane(tapv, tap);
            const char* pa = pbase + ao;
            const char* wbl = wbuf + boffB;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) A[p][mt] = *reinterpret_cast<const sbf16x8*>(pa + abase[p][mt]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) B[p][nt] = *reinterpret_cast<const sbf16x8*>(wbl + p * WPP + t * 2 * BN * 16 + nt * 512);
            }
        };
        auto loadA = [&](const char* pbase, int tap, sbf16x8 (&A)[3][MT]) {
            const char* pa = pbase + __builtin_amdgcn_readlane(tapv, tap);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) A[p][mt] = *reinterpret_cast<const sbf16x8*>(pa + abase[p][mt]);
        };
        auto loadB = [&](const char* wbuf, int t, sbf16x8 (&B)[3][NT]) {
            const char* wbl = wbuf + boffB;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) B[p][nt] = *reinterpret_cast<const sbf16x8*>(wbl + p * WPP + t * 2 * BN * 16 + nt * 512);
        };
        // issue order of one step: the 3 (MT + NT) fragment reads of the NEXT step go out between this step's first MFMAs, one
        // read (and its address add) per two MFMAs: a burst of reads in front of the MFMAs costs ~250 clocks per step in which
        // the matrix pipe is idle (an in-order wave issues nothing else while it issues them)
        auto interleave = [&]() {
#pragma unroll
            for (int i = 0; i < 3 * (MT + NT); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // 1 VALU (address)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 LDS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT * NT - 6 * (MT + NT), 0);
        };
        // the six kept terms of one accumulator back to back, smallest first (a chain on one accumulator is FASTER than rotating
        // over the MT x NT accumulators between terms: layer3 3x3 133 vs 156 us)
        auto mma = [&](const sbf16x8 (&A)[3][MT], const sbf16x8 (&B)[3][NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x16 c = acc[mt][nt];
                    RD_SPLIT_TERMS(c, A[0][mt], A[1][mt], A[2][mt], B[0][nt], B[1][nt], B[2][nt])
                    acc[mt][nt] = c;
                }
        };
        // two fragment sets; a group is three steps, so the set holding a group's first step alternates from group to group.  The
        // loop body is a PAIR of groups: set 0 is then the only one alive across the back edge (carrying both made the compiler
        // keep 120 fragment registers next to the 96 accumulators and spill).
        sbf16x8 fa[2][3][MT], fb[2][3][NT];
        unsigned long long* trc = (a.trace && a.trace_role == 1 && tid == 0) ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
        int it = 0, wi = 0, c = 0, g = 0;
        bool have = false;
        auto group = [&](auto parity) {
            constexpr int P0 = decltype(parity)::value, P1 = 1 - P0;
            if (trc && it < 30) trc[2 * it] = __builtin_readcyclecounter();
            rd_sync();                            // B1(it)
            if (trc && it < 30) trc[2 * it + 1] = __builtin_readcyclecounter();
            const char* pbase = s_patch + (PDB ? (c & 1) * 3 * pplane : 0);
            const int wn = wi == 2 ? 0 : wi + 1;
            const char* wb = s_w + wi * SLAB;
            const char* wbn = s_w + wn * SLAB;
            const int tap0 = g * GS_TPS;
            const bool pref = g + 1 < ngroups;
            if constexpr (PRE && MT == 3 && NT == 2) {      // (rotating form: one A set, the group's first B always in set 0)
                if (!have) loadA(pbase, min(tap0, ntaps - 1), fa[0]);
                loadB(wb, 0, fb[0]);
            } else if constexpr (MT == 3 && NT == 2) {      // (rotating form, weights published a group ahead: B prefetched like A)
                if (!have) { loadA(pbase, min(tap0, ntaps - 1), fa[0]); loadB(wb, 0, fb[P0]); }
            } else {
                if (!have) load(pbase, wb, min(tap0, ntaps - 1), 0, fa[P0], fb[P0]);
                else if constexpr (PRE) loadB(wb, 0, fb[P0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MT == 3 && NT == 2) {
                // The 3 x 2 tile has no registers for two full fragment sets next to its 96 accumulators (the form below reads a step's
                // 15 fragments in one burst in front of its MFMAs: ~27 clocks each with the matrix pipe idle, 4.9 k clocks per tap group
                // against 3.5 k of MFMA issue).  Rotating form: ONE set of A fragments, overwritten row by row as soon as a row's twelve
                // MFMAs have been issued, and two sets of B; the 15 reads of the next step go out two or three at a time between the
                // 6-MFMA blocks of this one.  B of a group's first step is read behind the group's barrier (lazy weight protocol).
                auto rstep = [&](sbf16x8 (&A)[3][MT], const sbf16x8 (&Bc)[3][NT], sbf16x8 (&Bn)[3][NT], int tapn, const char* wnext, int tn, bool doA, bool doB) {
                    const char* pa = pbase + __builtin_amdgcn_readlane(tapv, tapn);
                    const char* wbl = wnext + boffB;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            f32x16 c = acc[mt][nt];
                            RD_SPLIT_TERMS(c, A[0][mt], A[1][mt], A[2][mt], Bc[0][nt], Bc[1][nt], Bc[2][nt])
                            acc[mt][nt] = c;
                            __builtin_amdgcn_sched_barrier(0);
                            if (nt == 0) {
                                if (doB) {
#pragma unroll
                                    for (int q = 2 * mt; q < 2 * mt + 2; ++q)
                                        Bn[q / NT][q % NT] = *reinterpret_cast<const sbf16x8*>(wbl + (q / NT) * WPP + tn * 2 * BN * 16 + (q % NT) * 512);
                                }
                            } else if (doA) {
#pragma unroll
                                for (int pp = 0; pp < 3; ++pp) A[pp][mt] = *reinterpret_cast<const sbf16x8*>(pa + abase[pp][mt]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                };
                if constexpr (PRE) {
                    rstep(fa[0], fb[0], fb[1], min(tap0 + 1, ntaps - 1), wb, 1, true, true);
                    rstep(fa[0], fb[1], fb[0], min(tap0 + 2, ntaps - 1), wb, 2, true, true);
                    rstep(fa[0], fb[0], fb[1], min(tap0 + 3, ntaps - 1), wb, 0, pref, false);
                } else {      // (split while staging: the next group's weights were published by THIS group's barrier -- its first B is prefetched too)
                    rstep(fa[0], fb[P0], fb[P1], min(tap0 + 1, ntaps - 1), wb, 1, true, true);
                    rstep(fa[0], fb[P1], fb[P0], min(tap0 + 2, ntaps - 1), wb, 2, true, true);
                    rstep(fa[0], fb[P0], fb[P1], min(tap0 + 3, ntaps - 1), wbn, 0, pref, pref);
                }
            } else if constexpr (MT * NT < 4 || (MT * NT == 4 && PDB)) {
                load(pbase, wb, min(tap0 + 1, ntaps - 1), 1, fa[P1], fb[P1]);
                mma(fa[P0], fb[P0]);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                load(pbase, wb, min(tap0 + 2, ntaps - 1), 2, fa[P0], fb[P0]);
                mma(fa[P1], fb[P1]);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                if (pref) {
                    if constexpr (PRE) loadA(pbase, min(tap0 + 3, ntaps - 1), fa[P1]);     // (the next group's weights are published by ITS barrier)
                    else load(pbase, wbn, min(tap0 + 3, ntaps - 1), 0, fa[P1], fb[P1]);      // first step of the next group (same chunk)
                    mma(fa[P0], fb[P0]);
                    interleave();
                } else {
                    mma(fa[P0], fb[P0]);
                }
            } else {
                // the 3 x 2 tile (and the single-buffered 2 x 2) has no registers for interleaved reads (96 accumulators + two
                // fragment sets of 60 + the read addresses spill): its reads go out in front of each step's MFMAs
                load(pbase, wb, min(tap0 + 1, ntaps - 1), 1, fa[P1], fb[P1]);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa[P0], fb[P0]);
                __builtin_amdgcn_sched_barrier(0);
                load(pbase, wb, min(tap0 + 2, ntaps - 1), 2, fa[P0], fb[P0]);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa[P1], fb[P1]);
                __builtin_amdgcn_sched_barrier(0);
                if (pref) {
                    if constexpr (PRE) loadA(pbase, min(tap0 + 3, ntaps - 1), fa[P1]);
                    else load(pbase, wbn, min(tap0 + 3, ntaps - 1), 0, fa[P1], fb[P1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(fa[P0], fb[P0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            have = pref;
            wi = wn;
            if constexpr (!PDB) {
                if (!pref && c + 1 < nchunks) rd_sync();      // B2
            }
            ++it;
            if (++g == ngroups) { g = 0; ++c; }
        };
        while (it + 2 <= total_groups) {
            group(std::integral_constant<int, 0>{});
            group(std::integral_constant<int, 1>{});
        }
        if (it < total_groups) group(std::integral_constant<int, 0>{});
        if (trc) { trc[60] = __builtin_readcyclecounter(); trc[63] = (unsigned long long)total_groups; }
    }

    // ---- epilogue (gconv_bf16.hip's, fp32 tensors): the compute waves store, every wave takes part in the barriers
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    if (!loader && !(a.dbg & 16)) {
        const bool has_add = a.addend != nullptr;
        const bool has_bias = a.bias != nullptr;
        const int cob = co0 + l31;
        float biasv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) biasv[nt] = (has_bias && cob + nt * 32 < D.Cout) ? a.bias[cob + nt * 32] : 0.f;
        const bool want_stat = a.stat != nullptr;
        // 4x4 blocks (4 accumulator registers x the 4 lanes of a quad) are transposed in registers so that a lane holds four
        // consecutive channels of one pixel: 16-byte accesses (needs 4-channel alignment of every pointer and stride: a.vec4)
        const int q4l = l31 & 3, k4l = l31 >> 2;
        const bool odd1 = q4l & 1, odd2 = q4l & 2;
        float4 ssum4[NT], ssq4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ssum4[nt] = ssq4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any4 = false;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // (rows of masked or empty slots -- the slot map scatters an edge tile's masked pixels over the M tiles -- are skipped row
            //  by row: their loads go to row 0, their stores and statistics are predicated)
            int ro4[4];
            bool rok4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ro4[g] = s_opix[(wm * MT + mt) * 32 + q4l + 8 * g + 4 * hh];
                rok4[g] = ro4[g] >= 0;
                ro4[g] = rok4[g] ? ro4[g] : 0;
            }
            if (a.vec4) {
                any4 = true;
                const int cq = co0 + 4 * k4l;
                float4 addv[NT][4];      // the residual addend, or (BNB) the BatchNorm input x at the output pixels
                float4 bS[NT], bT[NT], bM[NT];
                if constexpr (BNB) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool on = cq + nt * 32 < D.Cout;
                        bS[nt] = on ? ld4(a.bnb_scale + cq + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                        bT[nt] = on ? ld4(a.bnb_shift + cq + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                        bM[nt] = on ? ld4(a.bnb_mean + cq + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (has_add || BNB) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float* ap = BNB ? a.bnb_x + (size_t)ro4[g] * a.bnb_ld + cq : a.addend + (size_t)ro4[g] * a.ld_add + cq;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) addv[nt][g] = (cq + nt * 32 < D.Cout) ? ld4(ap + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bool cok4 = cq + nt * 32 < D.Cout;
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (has_bias && cok4) b4 = *reinterpret_cast<const float4*>(a.bias + cq + nt * 32);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float e0 = acc[mt][nt][4 * g], e1 = acc[mt][nt][4 * g + 1], e2 = acc[mt][nt][4 * g + 2], e3 = acc[mt][nt][4 * g + 3];
                        quad_transpose(e0, e1, e2, e3, odd1, odd2);
                        float4 v = make_float4(e0 + b4.x, e1 + b4.y, e2 + b4.z, e3 + b4.w);
                        if (has_add && !BNB) { v.x += addv[nt][g].x; v.y += addv[nt][g].y; v.z += addv[nt][g].z; v.w += addv[nt][g].w; }
                        const int cc = cq + nt * 32;
                        if (cc < a.act_cols) {
                            v.x = act_fwd(v.x, a.act); v.y = act_fwd(v.y, a.act); v.z = act_fwd(v.z, a.act); v.w = act_fwd(v.w, a.act);
                        }
                        if (cok4 && rok4[g]) st4(a.out + (size_t)ro4[g] * D.ldo + cc, v);
                        if (want_stat && rok4[g]) {
                            if constexpr (BNB) {
                                const float4 xv = addv[nt][g];
                                const float gx = v.x * act_grad_from_out(fmaf(bS[nt].x, xv.x, bT[nt].x), a.bnb_act);
                                const float gy = v.y * act_grad_from_out(fmaf(bS[nt].y, xv.y, bT[nt].y), a.bnb_act);
                                const float gz = v.z * act_grad_from_out(fmaf(bS[nt].z, xv.z, bT[nt].z), a.bnb_act);
                                const float gw = v.w * act_grad_from_out(fmaf(bS[nt].w, xv.w, bT[nt].w), a.bnb_act);
                                ssum4[nt].x += gx; ssum4[nt].y += gy; ssum4[nt].z += gz; ssum4[nt].w += gw;
                                ssq4[nt].x += gx * (xv.x - bM[nt].x); ssq4[nt].y += gy * (xv.y - bM[nt].y);
                                ssq4[nt].z += gz * (xv.z - bM[nt].z); ssq4[nt].w += gw * (xv.w - bM[nt].w);
                            } else {
                                ssum4[nt].x += v.x; ssum4[nt].y += v.y; ssum4[nt].z += v.z; ssum4[nt].w += v.w;
                                ssq4[nt].x += v.x * v.x; ssq4[nt].y += v.y * v.y; ssq4[nt].z += v.z * v.z; ssq4[nt].w += v.w * v.w;
                            }
                        }
                    }
                }
            } else {
                int ro[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) ro[i] = s_opix[(wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = cob + nt * 32;
                    const bool cok = co < D.Cout;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (cok && ro[i] >= 0) {
                            float v = acc[mt][nt][i] + biasv[nt];
                            if (has_add) v += a.addend[(size_t)ro[i] * a.ld_add + co];
                            if (co < a.act_cols) v = act_fwd(v, a.act);
                            a.out[(size_t)ro[i] * D.ldo + co] = v;
                            ssum[nt] += v;
                            ssq[nt] += v * v;
                        }
                    }
                }
            }
        }
        if (want_stat && __any(any4)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4 s4 = ssum4[nt], q4 = ssq4[nt];
                s4.x += dpp_xor1(s4.x); s4.y += dpp_xor1(s4.y); s4.z += dpp_xor1(s4.z); s4.w += dpp_xor1(s4.w);
                q4.x += dpp_xor1(q4.x); q4.y += dpp_xor1(q4.y); q4.z += dpp_xor1(q4.z); q4.w += dpp_xor1(q4.w);
                s4.x += dpp_xor2(s4.x); s4.y += dpp_xor2(s4.y); s4.z += dpp_xor2(s4.z); s4.w += dpp_xor2(s4.w);
                q4.x += dpp_xor2(q4.x); q4.y += dpp_xor2(q4.y); q4.z += dpp_xor2(q4.z); q4.w += dpp_xor2(q4.w);
                ssum[nt] += odd2 ? (odd1 ? s4.w : s4.z) : (odd1 ? s4.y : s4.x);
                ssq[nt] += odd2 ? (odd1 ? q4.w : q4.z) : (odd1 ? q4.y : q4.x);
            }
        }
    }
    if (a.stat) {
        rd_sync();
        float* red = reinterpret_cast<float*>(s_w);  // [4][2][BN]
        if (!loader) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
                const float q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
                if (hh == 0) {
                    red[(wm * 2 + 0) * BN + nt * 32 + l31] = s;
                    red[(wm * 2 + 1) * BN + nt * 32 + l31] = q;
                }
            }
        }
        rd_sync();
        if (tid < 2 * BN) {
            const int which = tid / BN, j = tid - which * BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * BN + j];
            const int co = co0 + j;
            if (co < D.Cout) a.stat[((size_t)pt * (BNB ? 3 : 2) + which) * D.Cout + co] = s;
        }
    }
    if (a.trace && a.trace_role == 1 && tid == 0) a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_readcyclecounter();
}

// ================================================================================================================================
// gconv_sp2_kernel: the pre-split form with TWO workgroups per CU (round 4).
//
// Measured on the 8-wave kernel above (tools/ablate_gconv_split.py, tools/trace_gconv_split.py --pre, profiles/r04_*): with the split
// arithmetic gone from the staging waves the kernel did not get faster -- the MFMA waves' own stream (fragment-read bursts, 4.9 k clocks
// per tap group against 3.5 k of MFMA issue), the barrier waits behind copies that take 2-4 k clocks to land when every CU of the chip
// asks for the same weight slab at once, and an exposed prologue / VALU-heavy epilogue (a quarter of a 64-channel layer's lifetime)
// all sit on ONE workgroup's critical path, because one workgroup per CU has nobody to yield the matrix pipe to.  This kernel is the
// same arithmetic laid out the way the fp32 kernel of gconv.hip hides the same things: four waves per workgroup, no role split (staging
// is a handful of global_load_lds issues per tap group), at most 80 KB of LDS and 256 registers, so that two workgroups share a CU
// and one's copies-in-flight, barriers and epilogue run under the other's MFMAs.
//
//   tile      : BM = 4 waves x MT x 32 output pixels (MT <= 2), BN = NT x 32 channels.
//   LDS       : patch [piece][8-channel unit][patch pixel] x 16 B, ONE buffer (the next chunk's copy is issued behind a barrier at the
//               chunk's end and lands under the other workgroup's work); weights [2][piece][3 taps][2][BN] x 16 B, the next tap group's
//               copy issued at the start of each group.
//   loop      : per tap group: wait for the own copies, barrier, issue the next group's weights, three MFMA steps with the next step's
//               fragment reads interleaved; per chunk one more barrier before the patch is overwritten.
template <int MT, int NT, int DBG = 0, bool BNB = false>      // DBG (diagnostics, tools/ablate_gconv_split.py): 1 no MFMAs, 2 no fragment reads -- compile-time: a
                                            // run-time test inside the step splits the basic block the read / MFMA interleaving lives in
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gconv_sp2_kernel(const GsArgs a) {
    constexpr int BM = 4 * MT * 32;
    constexpr int BN = NT * 32;
    constexpr int LBN = NT == 2 ? 6 : 5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;

    if (a.trace && tid == 0) {
        a.trace[(size_t)blockIdx.x * 64 + 62] = __builtin_readcyclecounter();
        a.trace[(size_t)blockIdx.x * 64 + 59] = __builtin_amdgcn_s_memrealtime();                                   // 100 MHz, chip-wide
        a.trace[(size_t)blockIdx.x * 64 + 57] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |      // XCC_ID
                                                 (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                // HW_ID
    }
    // The two workgroups of a CU start together and would run in lockstep: the second "wave" of 256 blocks the dispatcher hands out
    // (blocks 256..511, 768..1023, ...: the co-residents of the first) starts a.stagger x 64 clocks late, so that one's chunk
    // boundaries (a barrier + the patch copy's latency) fall into the other's MFMA phases
    if (a.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = vid % a.n_cotiles;
    const int pt = vid / a.n_cotiles;
    const int n = pt / a.tiles_total;
    const int tt = pt - n * a.tiles_total;
    int ph_ = 0;
    for (int i = 1; i < D.n_phases; ++i)
        if (tt >= D.phase[i].tile_begin) ph_ = i;
    const int ph = __builtin_amdgcn_readfirstlane(ph_);
    const RdPhase& P = D.phase[ph];
    const int tloc = tt - P.tile_begin;
    const int tiles_w = (P.lw + a.TW - 1) / a.TW;
    const int r0 = (tloc / tiles_w) * a.TH, c0 = (tloc % tiles_w) * a.TW;
    const int th_n = min(a.TH, P.lh - r0), tw_n = min(a.TW, P.lw - c0);
    const int IS = D.in_stride, OS = D.out_stride;
    const int PW = (a.TW - 1) * IS + (P.dw_max - P.dw_min) + 1;
    const int PWP = a.ppitch[ph];                           // LDS row pitch of the patch in pixels (>= PW, gs_slot_pixel)
    const int PH = (th_n - 1) * IS + (P.dh_max - P.dh_min) + 1;
    const int ih0 = r0 * IS + P.dh_min, iw0 = c0 * IS + P.dw_min;
    const int ntaps = __builtin_amdgcn_readfirstlane(P.n_taps);
    const int ngroups = (ntaps + GS_TPS - 1) / GS_TPS;
    const int co0 = cot * BN;

    int* s_opix = reinterpret_cast<int*>(smem);          // [BM] output pixel index or -1
    int* s_apix = s_opix + BM;                           // [BM] patch pixel index of tap (0,0)
    int* s_widx = s_apix + BM;                           // [32] weight slab index of each tap
    char* s_w = reinterpret_cast<char*>(s_widx + 32);    // [2][3][GS_TPS][2][BN] x 16 B
    constexpr int WPP = GS_TPS * 2 * BN * 16;
    constexpr int SLAB = 3 * WPP;
    char* s_patch = s_w + 2 * SLAB;                      // [3][pplane]
    const int pplane = a.pplane;

    for (int m = tid; m < BM; m += 256) {
        const int sv = a.slots[ph * BM + m];
        const bool have = sv >= 0;
        const int r = sv >> 16, c = sv & 0xffff, rho = -1 - sv;
        const bool ok = have && (r < th_n) && (c < tw_n);
        s_opix[m] = ok ? ((n * D.Ho + (r0 + r) * OS + P.out_off_h) * D.Wo + (c0 + c) * OS + P.out_off_w) : -1;
        // (masked pixels of an edge tile keep their own patch address -- staged or not, their accumulator rows are never stored;
        //  empty slots read the first patch row at their lane's residue: no bank shared with the pass's live lanes)
        s_apix[m] = have ? ((r * IS) * PWP + c * IS) : rho * IS;
    }
    if (tid < ntaps) s_widx[tid] = P.widx[tid];
    {   // the patch starts as zeros: padding pixels / rows of the halo are never copied (their lanes are masked in every copy)
        const int n16 = (3 * pplane) >> 4;
        for (int e = tid; e < n16; e += 256) *reinterpret_cast<uint4*>(s_patch + (size_t)e * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    rd_sync();

    const int nchunks = D.Cin / GS_CKP;
    const int total_groups = nchunks * ngroups;
    const int cin8 = D.Cin >> 3;
    // copy the patch of 16-channel block blk: one global_load_lds per (patch row, 64-pixel segment, piece, 8-channel unit); rows
    // wm, wm + 4, ... of this wave
    auto issue_patch = [&](int blk) {
        if (a.dbg & 8) return;
        const char* src0 = reinterpret_cast<const char*>(a.inp) + ((size_t)blk * a.mpix + (size_t)n * D.Hi * D.Wi) * 32;
        const int nsg = (PW + 63) >> 6;
        for (int r = wm; r < PH; r += 4) {
            const int ih = ih0 + r;
            if (ih < 0 || ih >= D.Hi) continue;
            for (int sg = 0; sg < nsg; ++sg) {
                const int col = (sg << 6) + lane, iw = iw0 + col;
                const bool ok = col < PW && iw >= 0 && iw < D.Wi;
                const char* src = src0 + ((size_t)ih * D.Wi + iw) * 32;
                char* dst = s_patch + ((size_t)r * PWP + (sg << 6)) * 16;
                if (ok) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            glds16(reinterpret_cast<const float*>(src + p * a.xplane + u * 16), reinterpret_cast<float*>(dst + p * pplane + u * a.kplane));
                }
            }
        }
    };
    auto issue_slab = [&](int buf, int cb, int g) {
        if (a.dbg & 4) return;
        const int tap0 = g * GS_TPS;
        constexpr int welems = (GS_TPS * 2) << LBN;      // 16-byte units of one piece: [tap][2][BN]
        const char* src = reinterpret_cast<const char*>(a.w) + (size_t)(cb >> 3) * a.ldw * 16;
        char* dst = s_w + buf * SLAB;
#pragma unroll
        for (int u = 0; u < GS_UW; ++u) {
            const int e = tid + u * 256;
            if (e < welems) {
                const int j = e & (BN - 1), tk = e >> LBN;
                const int k8 = tk & 1, t = tap0 + (tk >> 1);
                if (t < ntaps && co0 + j < D.Cout) {
                    const size_t go = (((size_t)s_widx[t] * cin8 + k8) * a.ldw + co0 + j) * 16;
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        glds16(reinterpret_cast<const float*>(src + p * a.wplane + go), reinterpret_cast<float*>(dst + p * WPP + (e - lane) * 16));
                } else {
                    // (inline assembly: a plain LDS store behind outstanding global_load_lds copies makes the compiler wait for the copies)
                    const su32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned ad = (unsigned)(size_t)(dst + p * WPP + e * 16);
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(z) : "memory");
                    }
                }
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
    int aoffB[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aoffB[mt] = s_apix[(wm * MT + mt) * 32 + l31] * 16 + hh * a.kplane;
    const int boffB = (hh * BN + l31) * 16;
    const int tapv = a.tapoff[ph][lane < ntaps ? lane : 0];
    auto loadA = [&](int tap, sbf16x8 (&A)[3][MT]) {
        if constexpr (DBG & 2) return;
        const char* pa = s_patch + __builtin_amdgcn_readlane(tapv, tap);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) A[p][mt] = *reinterpret_cast<const sbf16x8*>(pa + p * pplane + aoffB[mt]);
    };
    auto loadB = [&](const char* wbuf, int t, sbf16x8 (&B)[3][NT]) {
        if constexpr (DBG & 2) return;
        const char* wbl = wbuf + boffB;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) B[p][nt] = *reinterpret_cast<const sbf16x8*>(wbl + p * WPP + t * 2 * BN * 16 + nt * 512);
    };
    // the next step's 3 (MT + NT) fragment reads go out between this step's first MFMAs, one read per two MFMAs
    auto interleave = [&](int) {
#pragma unroll
        for (int i = 0; i < 3 * (MT + NT); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // 1 VALU (address)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 LDS read
        }
        if constexpr (6 * MT * NT - 6 * (MT + NT) > 0) __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT * NT - 6 * (MT + NT), 0);
    };
    auto mma = [&](const sbf16x8 (&A)[3][MT], const sbf16x8 (&B)[3][NT]) {
        if constexpr (DBG & 1) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x16 c = acc[mt][nt];
                RD_SPLIT_TERMS(c, A[0][mt], A[1][mt], A[2][mt], B[0][nt], B[1][nt], B[2][nt])
                acc[mt][nt] = c;
            }
    };
    sbf16x8 fa[2][3][MT], fb[2][3][NT];
    unsigned long long* trc = (a.trace && a.trace_role == 1 && tid == 0) ? a.trace + (size_t)blockIdx.x * 64 : nullptr;

    // chunk order: workgroup-dependent rotation -- at any moment the workgroups of the chip then ask the L2 for DIFFERENT weight slabs
    // instead of all for the same one (a copy took 2-4 k clocks to land when they did); the order is a function of the tile alone,
    // so every output is still summed in one fixed order
    const int rot = nchunks > 1 ? (int)((unsigned)pt % (unsigned)nchunks) : 0;
    auto chunk_of = [&](int k) { const int q = k + rot; return q >= nchunks ? q - nchunks : q; };
    issue_patch(chunk_of(0));
    issue_slab(0, chunk_of(0) * GS_CKP, 0);
    int it = 0, c = 0, g = 0;
    int nc = ngroups > 1 ? 0 : 1, ng = ngroups > 1 ? 1 : 0;      // (chunk, group) of tap group it + 1
    bool haveA = false;
    // a group is three steps, so the fragment set holding a group's first step alternates from group to group: the loop body is a
    // PAIR of groups (compile-time set indices)
    auto group = [&](auto parity) {
        constexpr int P0 = decltype(parity)::value, P1 = 1 - P0;
        if (trc && it < 30) trc[2 * it] = __builtin_readcyclecounter();
        glds_wait();                          // this wave's copies: weights of group it (and the chunk's patch when g == 0)
        rd_sync();                            // B(it): they are published; every wave is done with group it - 1
        if (trc && it < 30) trc[2 * it + 1] = __builtin_readcyclecounter();
        if (it + 1 < total_groups) issue_slab((it + 1) & 1, chunk_of(nc) * GS_CKP, ng);
        if (++ng == ngroups) { ng = 0; ++nc; }
        const char* wb = s_w + (it & 1) * SLAB;
        const int tap0 = g * GS_TPS;
        const bool pref = g + 1 < ngroups;    // the next group reads the same patch: its first A fragments are prefetched
        if (!haveA) loadA(min(tap0, ntaps - 1), fa[P0]);
        loadB(wb, 0, fb[P0]);
        __builtin_amdgcn_sched_barrier(0);
        loadA(min(tap0 + 1, ntaps - 1), fa[P1]);
        loadB(wb, 1, fb[P1]);
        mma(fa[P0], fb[P0]);
        interleave(0);
        __builtin_amdgcn_sched_barrier(0);
        loadA(min(tap0 + 2, ntaps - 1), fa[P0]);
        loadB(wb, 2, fb[P0]);
        mma(fa[P1], fb[P1]);
        interleave(0);
        __builtin_amdgcn_sched_barrier(0);
        if (pref) loadA(min(tap0 + 3, ntaps - 1), fa[P1]);
        mma(fa[P0], fb[P0]);
        __builtin_amdgcn_sched_barrier(0);
        haveA = pref;
        ++it;
        if (++g == ngroups) {
            g = 0; ++c;
            if (c < nchunks) {
                rd_sync();                    // every wave is done with this chunk's patch
                issue_patch(chunk_of(c));
            }
        }
    };
    while (it + 2 <= total_groups) {
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
    }
    if (it < total_groups) group(std::integral_constant<int, 0>{});
    if (trc) { trc[60] = __builtin_readcyclecounter(); trc[63] = (unsigned long long)total_groups; }

    // ---- epilogue (the 8-wave kernel's, every wave a compute wave)
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    if (!(a.dbg & 16)) {
        const bool has_add = a.addend != nullptr;
        const bool has_bias = a.bias != nullptr;
        const int cob = co0 + l31;
        float biasv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) biasv[nt] = (has_bias && cob + nt * 32 < D.Cout) ? a.bias[cob + nt * 32] : 0.f;
        const bool want_stat = a.stat != nullptr;
        const int q4l = l31 & 3, k4l = l31 >> 2;
        const bool odd1 = q4l & 1, odd2 = q4l & 2;
        float4 ssum4[NT], ssq4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ssum4[nt] = ssq4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any4 = false;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // (rows of masked or empty slots -- the slot map scatters an edge tile's masked pixels over the M tiles -- are skipped row
            //  by row: their loads go to row 0, their stores and statistics are predicated)
            int ro4[4];
            bool rok4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ro4[q] = s_opix[(wm * MT + mt) * 32 + q4l + 8 * q + 4 * hh];
                rok4[q] = ro4[q] >= 0;
                ro4[q] = rok4[q] ? ro4[q] : 0;
            }
            if (a.vec4) {
                any4 = true;
                const int cq = co0 + 4 * k4l;
                float4 addv[NT][4];      // the residual addend, or (BNB) the BatchNorm input x at the output pixels
                float4 bS[NT], bT[NT], bM[NT];
                if constexpr (BNB) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bool on = cq + nt * 32 < D.Cout;
                        bS[nt] = on ? ld4(a.bnb_scale + cq + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                        bT[nt] = on ? ld4(a.bnb_shift + cq + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                        bM[nt] = on ? ld4(a.bnb_mean + cq + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (has_add || BNB) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float* ap = BNB ? a.bnb_x + (size_t)ro4[q] * a.bnb_ld + cq : a.addend + (size_t)ro4[q] * a.ld_add + cq;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) addv[nt][q] = (cq + nt * 32 < D.Cout) ? ld4(ap + nt * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bool cok4 = cq + nt * 32 < D.Cout;
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (has_bias && cok4) b4 = *reinterpret_cast<const float4*>(a.bias + cq + nt * 32);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float e0 = acc[mt][nt][4 * q], e1 = acc[mt][nt][4 * q + 1], e2 = acc[mt][nt][4 * q + 2], e3 = acc[mt][nt][4 * q + 3];
                        quad_transpose(e0, e1, e2, e3, odd1, odd2);
                        float4 v = make_float4(e0 + b4.x, e1 + b4.y, e2 + b4.z, e3 + b4.w);
                        if (has_add && !BNB) { v.x += addv[nt][q].x; v.y += addv[nt][q].y; v.z += addv[nt][q].z; v.w += addv[nt][q].w; }
                        const int cc = cq + nt * 32;
                        if (cc < a.act_cols) {
                            v.x = act_fwd(v.x, a.act); v.y = act_fwd(v.y, a.act); v.z = act_fwd(v.z, a.act); v.w = act_fwd(v.w, a.act);
                        }
                        if (cok4 && rok4[q]) st4(a.out + (size_t)ro4[q] * D.ldo + cc, v);
                        if (want_stat && rok4[q]) {
                            if constexpr (BNB) {
                                const float4 xv = addv[nt][q];
                                const float gx = v.x * act_grad_from_out(fmaf(bS[nt].x, xv.x, bT[nt].x), a.bnb_act);
                                const float gy = v.y * act_grad_from_out(fmaf(bS[nt].y, xv.y, bT[nt].y), a.bnb_act);
                                const float gz = v.z * act_grad_from_out(fmaf(bS[nt].z, xv.z, bT[nt].z), a.bnb_act);
                                const float gw = v.w * act_grad_from_out(fmaf(bS[nt].w, xv.w, bT[nt].w), a.bnb_act);
                                ssum4[nt].x += gx; ssum4[nt].y += gy; ssum4[nt].z += gz; ssum4[nt].w += gw;
                                ssq4[nt].x += gx * (xv.x - bM[nt].x); ssq4[nt].y += gy * (xv.y - bM[nt].y);
                                ssq4[nt].z += gz * (xv.z - bM[nt].z); ssq4[nt].w += gw * (xv.w - bM[nt].w);
                            } else {
                                ssum4[nt].x += v.x; ssum4[nt].y += v.y; ssum4[nt].z += v.z; ssum4[nt].w += v.w;
                                ssq4[nt].x += v.x * v.x; ssq4[nt].y += v.y * v.y; ssq4[nt].z += v.z * v.z; ssq4[nt].w += v.w * v.w;
                            }
                        }
                    }
                }
            } else {
                int ro[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) ro[i] = s_opix[(wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = cob + nt * 32;
                    const bool cok = co < D.Cout;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (cok && ro[i] >= 0) {
                            float v = acc[mt][nt][i] + biasv[nt];
                            if (has_add) v += a.addend[(size_t)ro[i] * a.ld_add + co];
                            if (co < a.act_cols) v = act_fwd(v, a.act);
                            a.out[(size_t)ro[i] * D.ldo + co] = v;
                            ssum[nt] += v;
                            ssq[nt] += v * v;
                        }
                    }
                }
            }
        }
        if (want_stat && __any(any4)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4 s4 = ssum4[nt], q4 = ssq4[nt];
                s4.x += dpp_xor1(s4.x); s4.y += dpp_xor1(s4.y); s4.z += dpp_xor1(s4.z); s4.w += dpp_xor1(s4.w);
                q4.x += dpp_xor1(q4.x); q4.y += dpp_xor1(q4.y); q4.z += dpp_xor1(q4.z); q4.w += dpp_xor1(q4.w);
                s4.x += dpp_xor2(s4.x); s4.y += dpp_xor2(s4.y); s4.z += dpp_xor2(s4.z); s4.w += dpp_xor2(s4.w);
                q4.x += dpp_xor2(q4.x); q4.y += dpp_xor2(q4.y); q4.z += dpp_xor2(q4.z); q4.w += dpp_xor2(q4.w);
                ssum[nt] += odd2 ? (odd1 ? s4.w : s4.z) : (odd1 ? s4.y : s4.x);
                ssq[nt] += odd2 ? (odd1 ? q4.w : q4.z) : (odd1 ? q4.y : q4.x);
            }
        }
    }
    if (a.stat) {
        rd_sync();
        float* red = reinterpret_cast<float*>(s_w);  // [4][2][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float s = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
            const float q = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
            if (hh == 0) {
                red[(wm * 2 + 0) * BN + nt * 32 + l31] = s;
                red[(wm * 2 + 1) * BN + nt * 32 + l31] = q;
            }
        }
        rd_sync();
        if (tid < 2 * BN) {
            const int which = tid / BN, j = tid - which * BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * BN + j];
            const int co = co0 + j;
            if (co < D.Cout) a.stat[((size_t)pt * (BNB ? 3 : 2) + which) * D.Cout + co] = s;
        }
    }
    if (a.trace && a.trace_role == 1 && tid == 0) {
        a.trace[(size_t)blockIdx.x * 64 + 61] = __builtin_readcyclecounter();
        a.trace[(size_t)blockIdx.x * 64 + 58] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------ host
struct GsPlan {
    int MT, NT, TH, TW, PP, tiles_total, n_cotiles, taps_max, pplane;
    size_t lds_bytes;
    int pdb;        // two patch buffers: the next chunk's patch is written while the current one is read (no second barrier)
    int sp2;        // pre-split input on gconv_sp2_kernel (four waves, two workgroups per CU)
    int ppitch[RD_MAX_PHASES];          // LDS row pitch of the patch per phase, pixels (gs_slot_pixel)
    unsigned nres[RD_MAX_PHASES][4];    // tile pixels per residue class, bytes
};

// row pitch of a phase's LDS patch: the patch width plus the 0..3 pad columns that leave the fewest pixels outside the conflict-free
// passes of the slot map (a 15 x 25 tile at pitch 27: none; 5 x 50 at 52: 2 of 250, at 54: none)
static int gs_pick_pitch(const RdConvDesc& d, const RdPhase& p, int TH, int TW, int G, unsigned (*nres)[4]) {
    const int PW = (TW - 1) * d.in_stride + (p.dw_max - p.dw_min) + 1;
    int best = PW, best_over = -1, n[16];
    for (int pad = 0; pad < 4; ++pad) {
        const int over = gs_residues(TH, TW, PW + pad, G, n);
        if (best_over < 0 || over < best_over) { best_over = over; best = PW + pad; }
        if (over == 0) break;
    }
    if (nres) {
        gs_residues(TH, TW, best, G, n);
        for (int w = 0; w < 4; ++w) {
            (*nres)[w] = 0;
            for (int b = 0; b < 4; ++b) (*nres)[w] |= (unsigned)(n[4 * w + b] > 255 ? 255 : n[4 * w + b]) << (8 * b);
        }
    }
    return best;
}

// patch rows x LDS pitch of a phase (the full tile: edge tiles use the same layout)
static int gs_patch_pixels(const RdConvDesc& d, const RdPhase& p, int TH, int TW, int G, int* rows, int* cols) {
    const int th = TH < p.lh ? TH : p.lh;
    const int PH = (th - 1) * d.in_stride + (p.dh_max - p.dh_min) + 1;
    const int PW = (TW - 1) * d.in_stride + (p.dw_max - p.dw_min) + 1;
    if (rows) *rows = PH;
    if (cols) *cols = PW;
    return PH * gs_pick_pitch(d, p, TH, TW, G, nullptr);
}

static void gs_fill_slot_map(const RdConvDesc& d, GsPlan& pl) {
    const int G = 4 * pl.MT * 32 / 16;
    for (int i = 0; i < RD_MAX_PHASES; ++i) {
        pl.ppitch[i] = 0;
        for (int w = 0; w < 4; ++w) pl.nres[i][w] = 0;
    }
    for (int i = 0; i < d.n_phases; ++i) pl.ppitch[i] = gs_pick_pitch(d, d.phase[i], pl.TH, pl.TW, G, &pl.nres[i]);
}

static bool plan_gconv_split(const RdConvDesc& d, GsPlan& best, bool pre) {
    struct Cfg { int MT, NT; };
    static const Cfg cfgs[] = {{3, 2}, {2, 2}, {1, 2}, {2, 1}, {1, 1}};
    int taps_max = 0;
    for (int i = 0; i < d.n_phases; ++i) {
        taps_max = taps_max > d.phase[i].n_taps ? taps_max : d.phase[i].n_taps;
    }
    int pr = 0;
    for (int i = 1; i < d.n_phases; ++i)
        if ((int64_t)d.phase[i].lh * d.phase[i].lw > (int64_t)d.phase[pr].lh * d.phase[pr].lw) pr = i;
    const RdPhase& P = d.phase[pr];
    double best_cost = -1;
    static const char* force = getenv("RD_GCONV_SPLIT_FORCE");   // diagnostics: index into cfgs
    int cfg_i = -1;
    for (const Cfg& c : cfgs) {
        ++cfg_i;
        if (force && atoi(force) != cfg_i) continue;
        const int BM = 4 * c.MT * 32, BN = c.NT * 32;
        if (BN > 32 && d.Cout <= 32) continue;
        const int n_cot = cdiv(d.Cout, BN);
        for (int twt = 1; twt <= cdiv(P.lw, 4); ++twt) {
            const int TW = cdiv(P.lw, twt);
            if (TW > BM) continue;
            int TH = BM / TW;
            if (TH > P.lh) TH = P.lh;
            TH = cdiv(P.lh, cdiv(P.lh, TH));
            int PP = 0, segs = 0;
            for (int i = 0; i < d.n_phases; ++i) {
                int rows, cols;
                const int pp = gs_patch_pixels(d, d.phase[i], TH, TW, BM / 16, &rows, &cols);
                PP = PP > pp ? PP : pp;
                const int sg = rows * cdiv(cols * 2, 64);
                segs = segs > sg ? segs : sg;
            }
            if (!pre && segs > 4 * GS_UPP) continue;         // the next chunk's patch is one register batch of the staging waves
            if (pre && 6 * segs > 4 * 44) continue;          // pre-split: at most 44 copies per staging wave and chunk in flight (vm_wait_upto)
            // pre-split: [unit][pixel] x 16 B per piece, no padding (consecutive pixels = consecutive 16-byte slots)
            const int pplane = pre ? 2 * (((PP * 16) + 63) & ~63) : ((((PP + 1) * GS_PSB) + 15) & ~15);
            static const char* nopdb = getenv("RD_GCONV_SPLIT_NOPDB");       // diagnostics
            for (int pdb = (nopdb && !pre) ? 0 : 1; pdb >= (pre ? 1 : 0); --pdb) {
                const size_t lds = (size_t)(2 * BM + 32) * 4 + (size_t)3 * 3 * GS_TPS * 2 * BN * 16 + (size_t)(pdb ? 2 : 1) * 3 * pplane + 64;
                if (lds > 160 * 1024 - 512) continue;
                // matrix-core bound, one workgroup per CU: rounds of workgroups x (clocks of one: three-step tap groups of
                // MT x NT x 6 MFMAs of ~36 clocks, a barrier per group, an exposed patch write per chunk unless double-buffered,
                // a fixed prologue / epilogue)
                double groups = 0, wgs = 0;
                for (int i = 0; i < d.n_phases; ++i) {
                    const double t = (double)cdiv(d.phase[i].lh, TH) * cdiv(d.phase[i].lw, TW);
                    groups += t * cdiv(d.phase[i].n_taps, GS_TPS);
                    wgs += t;
                }
                const double groups_per_wg = groups / wgs;
                wgs *= (double)d.N * n_cot;
                // calibrated on layer1..4 at b = 16 (tools/bench_split_one.py, tools/trace_gconv_split.py), all in: a tap group costs
                // ~42 clocks per MFMA (the MFMA waves' own stream runs at ~40) + ~1300 (barrier, the wait for the staging waves,
                // which share the SIMDs' issue slots): 3x2 tile 5870 measured, 2x2 4000, 1x1 2060
                const double mfma_clk = 3.0 * c.MT * c.NT * 6 * 42;
                const double lds_clk = 3.0 * 3 * (c.MT + c.NT) * 4 * 4 / 0.7;         // four compute waves x 4 LDS clocks per b128 read
                const double per_group = (mfma_clk > lds_clk ? mfma_clk : lds_clk) + 1300.0;
                const double per_chunk = groups_per_wg * per_group;
                const double per_wg = (double)(d.Cin / GS_CKP) * per_chunk + 7000.0 + 16.0 * c.MT * c.NT * 60;
                const double rounds = ceil(wgs / (double)num_cus());
                const double cost = rounds * per_wg;
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best = GsPlan{c.MT, c.NT, TH, TW, PP, 0, n_cot, taps_max, pplane, lds, pdb, 0, {}, {}};
                }
                break;      // the double-buffered form of a tile is never worse than its single-buffered one
            }
        }
    }
    return best_cost > 0;
}

// gconv_sp2_kernel: tile for two workgroups per CU (<= 80 KB of LDS each)
static bool plan_sp2(const RdConvDesc& d, GsPlan& best) {
    struct Cfg { int MT, NT; };
    static const Cfg cfgs[] = {{2, 2}, {1, 2}, {2, 1}, {1, 1}, {3, 1}};
    int taps_max = 0;
    for (int i = 0; i < d.n_phases; ++i) taps_max = taps_max > d.phase[i].n_taps ? taps_max : d.phase[i].n_taps;
    int pr = 0;
    for (int i = 1; i < d.n_phases; ++i)
        if ((int64_t)d.phase[i].lh * d.phase[i].lw > (int64_t)d.phase[pr].lh * d.phase[pr].lw) pr = i;
    const RdPhase& P = d.phase[pr];
    double best_cost = -1;
    static const char* force = getenv("RD_GCONV_SP2_FORCE");     // diagnostics: index into cfgs
    int cfg_i = -1;
    for (const Cfg& c : cfgs) {
        ++cfg_i;
        if (force && atoi(force) != cfg_i) continue;
        const int BM = 4 * c.MT * 32, BN = c.NT * 32;
        if (BN > 32 && d.Cout <= 32) continue;
        const int n_cot = cdiv(d.Cout, BN);
        for (int twt = 1; twt <= cdiv(P.lw, 4); ++twt) {
            const int TW = cdiv(P.lw, twt);
            if (TW > BM) continue;
            int TH = BM / TW;
            if (TH > P.lh) TH = P.lh;
            TH = cdiv(P.lh, cdiv(P.lh, TH));
            int PP = 0;
            double copies = 0;
            for (int i = 0; i < d.n_phases; ++i) {
                int rows, cols;
                const int pp = gs_patch_pixels(d, d.phase[i], TH, TW, BM / 16, &rows, &cols);
                PP = PP > pp ? PP : pp;
                copies = copies > 6.0 * rows * cdiv(cols, 64) ? copies : 6.0 * rows * cdiv(cols, 64);
            }
            const int pplane = 2 * (((PP * 16) + 63) & ~63);
            const size_t lds = (size_t)(2 * BM + 32) * 4 + (size_t)2 * 3 * GS_TPS * 2 * BN * 16 + (size_t)3 * pplane + 64;
            if (lds > 80 * 1024 - 256) continue;
            double groups = 0, wgs = 0;
            for (int i = 0; i < d.n_phases; ++i) {
                const double t = (double)cdiv(d.phase[i].lh, TH) * cdiv(d.phase[i].lw, TW);
                groups += t * cdiv(d.phase[i].n_taps, GS_TPS);
                wgs += t;
            }
            const double groups_per_wg = groups / wgs;
            wgs *= (double)d.N * n_cot;
            // two workgroups share a CU's matrix pipe: a CU's time is the MFMA issue time of the workgroups it gets (34 clocks per MFMA
            // incl. what the co-resident workgroup does not hide) plus per-group / per-chunk / per-workgroup costs that are only partly
            // hidden; copies cost issue slots (~40 clocks each, a quarter per wave)
            const double per_group = 3.0 * c.MT * c.NT * 6 * 34 + 250.0;
            const double per_chunk = groups_per_wg * per_group + 600.0 + copies / 4 * 40.0;
            const double per_wg = (double)(d.Cin / GS_CKP) * per_chunk + 3000.0 + 16.0 * c.MT * c.NT * 50;
            // a CU runs its workgroups two at a time; an odd one out runs alone and hides nothing (measured ~0.6 of the shared rate)
            const double per_cu = ceil(wgs / (double)num_cus());
            const double cost = per_cu * per_wg + (((int)per_cu & 1) ? 0.5 * per_wg : 0.0);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = GsPlan{c.MT, c.NT, TH, TW, PP, 0, n_cot, taps_max, pplane, lds, 0, 1, {}, {}};
            }
        }
    }
    return best_cost > 0;
}

template <int MT, int NT, int DBG = 0, bool BNB = false>
static int launch_sp2(const GsArgs& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = gconv_sp2_kernel<MT, NT, DBG, BNB>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("gconv_sp2_kernel");
    return RD_OK;
}

template <int MT, int NT, bool PDB, bool PRE, bool BNB = false>
static int launch_gs(const GsArgs& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = gconv_split_kernel<MT, NT, PDB, PRE, BNB>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
    RD_CHECK_LAUNCH("gconv_split_kernel");
    return RD_OK;
}

static std::mutex g_gs_mu;
static int g_gs_all = -1;         // -1: not set yet (the environment decides at first use)
static unsigned g_gs_epoch = 0;   // bumped by rd_gconv_split_plan_all: plans cached under another setting are dropped
static bool gs_plan_all() {
    std::lock_guard<std::mutex> lk(g_gs_mu);
    if (g_gs_all < 0) g_gs_all = getenv("RD_GCONV_SPLIT_ALL") ? 1 : 0;
    return g_gs_all == 1;
}

static bool gs_shape_ok(const RdConvDesc* d, bool sp2 = false) {
    if (!d || d->n_phases < 1 || d->n_phases > RD_MAX_PHASES) return false;
    if (d->Cin % 16 != 0 || d->ldi % 4 != 0 || d->Cin < 32 || d->Cout < 32) return false;      // (the 16-channel layers stay on conv16.hip)
    if (d->in_stride < 1 || d->in_stride > 2 || d->out_stride < 1 || d->out_stride > 2) return false;
    // (rd_gconv_split_plan_all(1), tests: plan every shape the kernel can run; RD_GCONV_SPLIT_ALL=1 sets the initial value)
    if (!gs_plan_all()) {
        // measured slower than (or level with) gconv.hip at the bench geometry (tools/bench_split.py, profiles/r03_bench_split.txt):
        //   * one-tap layers (0.4-0.8x: a chunk is 36 MFMAs, nothing to hide the staging behind);
        //   * fewer than 64 channels on a side (0.8-1.0x on the 32-channel decoder layers: the 32-wide output tile halves the reuse
        //     of every staged patch);
        //   * (the UpProj input gradient -- stride-2 input, four phases -- gains 1.1-1.4x; stride-2 3x3: see below)
        int taps_max = 0;
        for (int i = 0; i < d->n_phases; ++i) taps_max = taps_max > d->phase[i].n_taps ? taps_max : d->phase[i].n_taps;
        if (taps_max < 4) return false;
        // (gconv_sp2_kernel has 32-wide output tiles at two workgroups per CU: the 32-channel decoder / depth-encoder layers pay there)
        if (!sp2 && (d->Cin < 64 || d->Cout < 64)) return false;
        // stride-2 3x3 in either direction: only on pre-split operands (gconv_sp2: 1.28-1.56x forward, 1.05-1.25x for the zero-filled input
        // gradients against the fp32 kernel; the producers' extra piece planes leave +0.9 % on the step, round 5 -- in round 4, with the 1x1
        // layers still on the fp32 kernel, it was level); split while staging it is 0.75-0.95x: the patch holds four times the pixels a tap
        // touches.  RD_GCONV_S2_PRE=0 / 1 (A/B): fp32 kernels / forward only
        static const int s2pre = getenv("RD_GCONV_S2_PRE") ? atoi(getenv("RD_GCONV_S2_PRE")) : 2;      // 0: fp32 kernels, 1: forward only, 2: + input gradient
        if (d->in_stride == 2 && taps_max <= 9 && !(sp2 && s2pre >= 1)) return false;
        if (d->out_stride == 2 && taps_max <= 4 && !(sp2 && s2pre >= 2)) return false;      // stride-2 input gradient (phases of 1 / 2 / 2 / 4 taps)
    }
    if ((int64_t)d->Hi * d->Wi * d->ldi * 4 >= (int64_t)GS_OOB) return false;
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        if (p.n_taps < 1 || p.n_taps > RD_MAX_TAPS || p.lh < 1 || p.lw < 1) return false;
        for (int t = 0; t < p.n_taps; ++t)
            if (p.dh[t] < p.dh_min || p.dh[t] > p.dh_max || p.dw[t] < p.dw_min || p.dw[t] > p.dw_max) return false;
    }
    return true;
}

// 1: planned, 0: no plan (shape outside the kernel's domain or no tiling fits the LDS)
static int gs_plan_query(const RdConvDesc* d, GsPlan& pl, RdConvDesc& dd, bool pre = false) {
    struct Entry { int ok; GsPlan pl; RdConvDesc dd; };
    static std::mutex mu;
    static std::unordered_map<std::string, Entry> cache;
    static unsigned cache_epoch = 0;
    if (!d) return 0;
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    key.push_back(pre ? 'P' : 'R');
    {
        unsigned ep;
        { std::lock_guard<std::mutex> lk(g_gs_mu); ep = g_gs_epoch; }
        std::lock_guard<std::mutex> lk(mu);
        if (ep != cache_epoch) { cache.clear(); cache_epoch = ep; }
        auto it = cache.find(key);
        if (it != cache.end()) { pl = it->second.pl; dd = it->second.dd; return it->second.ok; }
    }
    Entry e{};
    e.dd = *d;
    static const char* no_sp2 = getenv("RD_GCONV_SP2");          // RD_GCONV_SP2=0: pre-split input on the 8-wave kernel (diagnostics)
    const bool sp2 = pre && !(no_sp2 && atoi(no_sp2) == 0);
    e.ok = gs_shape_ok(d, sp2) && (sp2 ? plan_sp2(e.dd, e.pl) : plan_gconv_split(e.dd, e.pl, pre)) ? 1 : 0;
    if (e.ok) {
        int tb = 0;
        for (int i = 0; i < e.dd.n_phases; ++i) {
            e.dd.phase[i].tile_begin = tb;
            tb += cdiv(e.dd.phase[i].lh, e.pl.TH) * cdiv(e.dd.phase[i].lw, e.pl.TW);
        }
        e.pl.tiles_total = tb;
        gs_fill_slot_map(e.dd, e.pl);
    }
    pl = e.pl; dd = e.dd;
    std::lock_guard<std::mutex> lk(mu);
    cache.emplace(std::move(key), e);
    return e.ok;
}

// slot table of a plan on the current device, [phase][BM] ints (GsArgs::slots); built and uploaded at the first launch of the plan
// (one synchronous 1.5-6 KB copy: the training / inference plans run one un-captured step before any graph capture).  Tables are
// never freed (a few KB per distinct descriptor); RD_GCONV_SPLIT_NATURAL=1 (diagnostics, read once): the round-4 row-major map.
static const int* gs_slot_table(const RdConvDesc* d, bool pre, const GsPlan& pl) {
    static std::mutex mu;
    static std::unordered_map<std::string, int*> tables;
    static const bool natural = getenv("RD_GCONV_SPLIT_NATURAL") && atoi(getenv("RD_GCONV_SPLIT_NATURAL"));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    unsigned ep;
    { std::lock_guard<std::mutex> lk(g_gs_mu); ep = g_gs_epoch; }
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    key.push_back(pre ? 'P' : 'R');
    key.append(reinterpret_cast<const char*>(&dev), sizeof(dev));
    key.append(reinterpret_cast<const char*>(&ep), sizeof(ep));
    std::lock_guard<std::mutex> lk(mu);
    auto it = tables.find(key);
    if (it != tables.end()) return it->second;
    const int BM = 4 * pl.MT * 32;
    std::vector<int> host((size_t)RD_MAX_PHASES * BM, -1);
    for (int i = 0; i < d->n_phases; ++i)
        for (int m = 0; m < BM; ++m) {
            int r, c, rho;
            const bool have = gs_slot_pixel(m, pl.TH, pl.TW, pl.ppitch[i], BM / 16, pl.nres[i], natural, r, c, rho);
            host[(size_t)i * BM + m] = have ? ((r << 16) | c) : -1 - rho;
        }
    int* devp = nullptr;
    if (hipMalloc(&devp, host.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(devp, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(devp); return nullptr; }
    tables.emplace(std::move(key), devp);
    return devp;
}

}  // namespace rd

using namespace rd;

static unsigned long long* g_gs_trace = nullptr;
// diagnostics: copy the stamps of the last traced launch (64 slots per workgroup: before / after barrier B1 of the first 30 tap
// groups; [60] end of the MFMA loop, [61] end of the epilogue, [62] kernel entry, [63] tap groups)
extern "C" int rd_gconv_split_trace_read(unsigned long long* host, int n_wg) {
    if (!g_gs_trace) return RD_EINVAL;
    RD_CHECK_HIP(hipMemcpy(host, g_gs_trace, (size_t)n_wg * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RD_OK;
}

// tests / sweeps: on != 0 plans every shape the kernel can run, also those the library leaves to rd_gconv because they measured
// slower there.  Returns the previous setting.  Plans cached under the other setting are dropped; callers that sized buffers on a
// plan (statistics tiles) must not toggle this between the query and the launch.
extern "C" int rd_gconv_split_plan_all(int on) {
    gs_plan_all();
    std::lock_guard<std::mutex> lk(g_gs_mu);
    const int prev = g_gs_all;
    if ((on != 0) != (prev == 1)) { g_gs_all = on ? 1 : 0; ++g_gs_epoch; }
    return prev;
}

// (one-tap descriptors -- 1x1 convolutions and their input gradients -- go to gemm1_split.hip's channel-grouped kernel: same operands,
//  same arithmetic, same epilogue; the three queries below and rd_gconv_split answer for whichever kernel serves the descriptor)
extern "C" int rd_gconv_split_supported(const RdConvDesc* d) {
    GsPlan pl; RdConvDesc dd;
    if (gemm1_split_supported(d)) return 1;
    return gs_plan_query(d, pl, dd);
}

// diagnostics: out[0..7] = MT, NT, TH, TW, PP, LDS bytes, workgroups, tap groups of the largest phase
extern "C" int rd_gconv_split_plan_info(const RdConvDesc* d, int32_t* out) {
    GsPlan pl; RdConvDesc dd;
    if (out && gemm1_split_supported(d)) return gemm1_split_plan_info(d, out);        // (out[7] = 1000 marks the 1x1 kernel)
    if (!out || gs_plan_query(d, pl, dd) != 1) return RD_EINVAL;
    const int v[8] = {pl.MT, pl.NT, pl.TH, pl.TW, pl.PP, (int)pl.lds_bytes, d->N * pl.tiles_total * pl.n_cotiles, (pl.taps_max + GS_TPS - 1) / GS_TPS + 100 * pl.pdb};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RD_OK;
}

// diagnostics / tests: the slot map of phase `phase` of the plan (pre != 0: the pre-split plan): out[0..3] = BM, TH, TW, row pitch of
// the LDS patch in pixels; slots[m] = (r << 16) | c of the tile pixel slot m holds, -1 for an empty slot (n_slots >= BM entries)
extern "C" int rd_gconv_split_slot_map(const RdConvDesc* d, int32_t pre, int32_t phase, int32_t* out, int32_t* slots, int32_t n_slots) {
    GsPlan pl; RdConvDesc dd;
    if (!out || !slots || gs_plan_query(d, pl, dd, pre != 0) != 1 || phase < 0 || phase >= d->n_phases) return RD_EINVAL;
    const int BM = 4 * pl.MT * 32;
    if (n_slots < BM) return RD_EINVAL;
    out[0] = BM; out[1] = pl.TH; out[2] = pl.TW; out[3] = pl.ppitch[phase];
    for (int m = 0; m < BM; ++m) {
        int r, c, rho;
        slots[m] = gs_slot_pixel(m, pl.TH, pl.TW, pl.ppitch[phase], BM / 16, pl.nres[phase], false, r, c, rho) ? ((r << 16) | c) : -1;
    }
    return RD_OK;
}

extern "C" int rd_gconv_split_stat_tiles(const RdConvDesc* d) {
    GsPlan pl; RdConvDesc dd;
    if (gemm1_split_supported(d)) return gemm1_split_stat_tiles(d);
    if (gs_plan_query(d, pl, dd) != 1) return RD_EINVAL;
    return d->N * pl.tiles_total;
}

struct GsBnb {      // rd_gconv_split[_pre]_bnbwd: the BatchNorm whose backward sums this input-gradient launch also emits (x == nullptr: none)
    const float* x = nullptr;
    int ld = 0;
    const float* mean = nullptr;
    const float* scale = nullptr;
    const float* shift = nullptr;
    int act = 0;
};

static int gs_launch(const RdConvDesc* d, const float* in, const void* in_pieces, int64_t in_piece_elems, const void* w_split, int64_t piece_elems,
                     float* out, const float* bias, int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, void* stream,
                     const GsBnb& bnb = GsBnb()) {
    const bool pre = in_pieces != nullptr;
    RD_CHECK_ARG(d && (in || in_pieces) && w_split && out, "gconv_split: null argument");
    if (!pre && gemm1_split_supported(d)) {
        RD_CHECK_ARG(!bnb.x, "gconv_split_bnbwd: one-tap descriptors are not served (rd_gconv_split_bnbwd_supported)");
        return launch_gemm1_split(d, in, w_split, piece_elems, out, bias, act, act_cols, addend, ld_add, stat_partial, static_cast<hipStream_t>(stream));
    }
    GsArgs a;
    GsPlan pl;
    if (gs_plan_query(d, pl, a.d, pre) != 1) { set_error("gconv_split: descriptor not supported (rd_gconv_split_supported)"); return RD_EINVAL; }
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(pre ? in_pieces : (const void*)in) % 16 == 0 && reinterpret_cast<uintptr_t>(w_split) % 16 == 0, "gconv_split: unaligned tensor");
    a.in = in; a.w = static_cast<const unsigned short*>(w_split); a.out = out;
    a.addend = addend; a.bias = bias; a.stat = stat_partial;
    a.act = act; a.act_cols = act_cols; a.ld_add = ld_add; a.ldw = d->Cout;
    a.TH = pl.TH; a.TW = pl.TW; a.PP = pl.PP;
    a.tiles_total = pl.tiles_total; a.n_cotiles = pl.n_cotiles; a.taps_max = pl.taps_max;
    a.pplane = pl.pplane;
    a.inp = static_cast<const unsigned short*>(in_pieces);
    a.mpix = (long long)d->N * d->Hi * d->Wi;
    a.xplane = (long long)in_piece_elems * 2;
    a.kplane = pl.pplane / 2;
    if (pre) RD_CHECK_ARG(in_piece_elems >= (int64_t)(d->Cin / 16) * a.mpix * 16 && in_piece_elems % 8 == 0, "gconv_split: activation piece stride %lld too small for %d x %lld",
                          (long long)in_piece_elems, d->Cin, a.mpix);
    int S = 0;
    for (int i = 0; i < d->n_phases; ++i)
        for (int t = 0; t < d->phase[i].n_taps; ++t) S = S > d->phase[i].widx[t] + 1 ? S : d->phase[i].widx[t] + 1;
    RD_CHECK_ARG(piece_elems >= (int64_t)S * d->Cin * d->Cout && piece_elems % 8 == 0, "gconv_split: piece stride %lld too small for %d slabs of %d x %d",
                 (long long)piece_elems, S, d->Cin, d->Cout);
    a.wplane = (long long)piece_elems * 2;
    a.vec4 = d->Cout % 4 == 0 && d->ldo % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && act_cols % 4 == 0 &&
             (!addend || (ld_add % 4 == 0 && reinterpret_cast<uintptr_t>(addend) % 16 == 0)) &&
             (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        const int PW_ = pl.ppitch[i];
        for (int t = 0; t < p.n_taps; ++t) a.tapoff[i][t] = ((p.dh[t] - p.dh_min) * PW_ + (p.dw[t] - p.dw_min)) * (pre ? 16 : GS_PSB);
    }
    for (int i = 0; i < RD_MAX_PHASES; ++i) a.ppitch[i] = pl.ppitch[i];
    a.bnb_x = bnb.x; a.bnb_ld = bnb.ld; a.bnb_mean = bnb.mean; a.bnb_scale = bnb.scale; a.bnb_shift = bnb.shift; a.bnb_act = bnb.act;
    if (bnb.x) {
        RD_CHECK_ARG(bnb.mean && bnb.scale && bnb.shift && stat_partial && !addend && !bias && act == 0, "gconv_split_bnbwd: bad arguments");
        RD_CHECK_ARG(a.vec4 && bnb.ld % 4 == 0 && reinterpret_cast<uintptr_t>(bnb.x) % 16 == 0 && reinterpret_cast<uintptr_t>(bnb.mean) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(bnb.scale) % 16 == 0 && reinterpret_cast<uintptr_t>(bnb.shift) % 16 == 0,
                     "gconv_split_bnbwd: four-channel alignment of every tensor and stride required");
    }
    a.slots = gs_slot_table(d, pre, pl);
    if (!a.slots) { set_error("gconv_split: cannot allocate the slot table of the plan"); return RD_ELAUNCH; }
    const int grid = d->N * pl.tiles_total * pl.n_cotiles;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.trace = nullptr;
    a.trace_role = 0;
    {
        const char* dbg = getenv("RD_GCONV_SPLIT_DEBUG");     // (read at every launch: the ablation tool toggles it between launches)
        a.dbg = dbg ? atoi(dbg) : 0;
        const char* stg = getenv("RD_GCONV_SP2_STAGGER");     // (diagnostics / sweep)
        a.stagger = stg ? atoi(stg) : 0;
    }
    {
        static const char* tr = getenv("RD_GCONV_SPLIT_TRACE");
        if (tr && atoi(tr)) {
            if (!g_gs_trace) RD_CHECK_HIP(hipMalloc(&g_gs_trace, (size_t)65536 * 64 * sizeof(unsigned long long)));
            a.trace = g_gs_trace;
            a.trace_role = atoi(tr);
        }
    }
    if (bnb.x) {
        // the BNB instantiations (epilogue also emits the BatchNorm-backward sums)
        if (pl.sp2) {
            if (pl.MT == 2 && pl.NT == 2) return launch_sp2<2, 2, 0, true>(a, grid, pl.lds_bytes, s);
            if (pl.MT == 1 && pl.NT == 2) return launch_sp2<1, 2, 0, true>(a, grid, pl.lds_bytes, s);
            if (pl.MT == 2 && pl.NT == 1) return launch_sp2<2, 1, 0, true>(a, grid, pl.lds_bytes, s);
            if (pl.MT == 1 && pl.NT == 1) return launch_sp2<1, 1, 0, true>(a, grid, pl.lds_bytes, s);
            if (pl.MT == 3 && pl.NT == 1) return launch_sp2<3, 1, 0, true>(a, grid, pl.lds_bytes, s);
        }
#define RD_GSB(MT_, NT_)                                                                                                              \
    if (pl.MT == MT_ && pl.NT == NT_)                                                                                                 \
        return pre ? launch_gs<MT_, NT_, true, true, true>(a, grid, pl.lds_bytes, s)                                                  \
                   : pl.pdb ? launch_gs<MT_, NT_, true, false, true>(a, grid, pl.lds_bytes, s) : launch_gs<MT_, NT_, false, false, true>(a, grid, pl.lds_bytes, s);
        RD_GSB(3, 2) RD_GSB(2, 2) RD_GSB(1, 2) RD_GSB(2, 1) RD_GSB(1, 1)
#undef RD_GSB
        set_error("gconv_split_bnbwd: no kernel for tile %dx%d", pl.MT, pl.NT);
        return RD_EINVAL;
    }
    if (pl.sp2) {
        if (pl.MT == 2 && pl.NT == 2 && (a.dbg & 3) == 1) return launch_sp2<2, 2, 1>(a, grid, pl.lds_bytes, s);      // (ablations of the main tile only)
        if (pl.MT == 2 && pl.NT == 2 && (a.dbg & 3) == 2) return launch_sp2<2, 2, 2>(a, grid, pl.lds_bytes, s);
        if (pl.MT == 2 && pl.NT == 2 && (a.dbg & 3) == 3) return launch_sp2<2, 2, 3>(a, grid, pl.lds_bytes, s);
        if (pl.MT == 2 && pl.NT == 2) return launch_sp2<2, 2>(a, grid, pl.lds_bytes, s);
        if (pl.MT == 1 && pl.NT == 2) return launch_sp2<1, 2>(a, grid, pl.lds_bytes, s);
        if (pl.MT == 2 && pl.NT == 1) return launch_sp2<2, 1>(a, grid, pl.lds_bytes, s);
        if (pl.MT == 1 && pl.NT == 1) return launch_sp2<1, 1>(a, grid, pl.lds_bytes, s);
        if (pl.MT == 3 && pl.NT == 1) return launch_sp2<3, 1>(a, grid, pl.lds_bytes, s);
    }
#define RD_GS(MT_, NT_)                                                                                                         \
    if (pl.MT == MT_ && pl.NT == NT_)                                                                                           \
        return pre ? launch_gs<MT_, NT_, true, true>(a, grid, pl.lds_bytes, s)                                                  \
                   : pl.pdb ? launch_gs<MT_, NT_, true, false>(a, grid, pl.lds_bytes, s) : launch_gs<MT_, NT_, false, false>(a, grid, pl.lds_bytes, s);
    RD_GS(3, 2) RD_GS(2, 2) RD_GS(1, 2) RD_GS(2, 1) RD_GS(1, 1)
#undef RD_GS
    set_error("gconv_split: no kernel for tile %dx%d", pl.MT, pl.NT);
    return RD_EINVAL;
}

extern "C" int rd_gconv_split(const RdConvDesc* d, const float* in, const void* w_split, int64_t piece_elems, float* out, const float* bias,
                              int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, void* stream) {
    RD_CHECK_ARG(in != nullptr, "gconv_split: null argument");
    return gs_launch(d, in, nullptr, 0, w_split, piece_elems, out, bias, act, act_cols, addend, ld_add, stat_partial, stream);
}

// The same convolution with the activation ALREADY split by its producer (rd_split_pieces, or the piece outputs of rd_bn_act_p and
// friends): in_pieces = three planes [piece][Cin/16][N*Hi*Wi][16] bf16, in_piece_elems elements apart.  The staging waves then do
// nothing but global_load_lds copies.  The plan (tile, statistics tiles) differs from rd_gconv_split's: query with the _pre forms.
extern "C" int rd_gconv_split_pre(const RdConvDesc* d, const void* in_pieces, int64_t in_piece_elems, const void* w_split, int64_t piece_elems, float* out,
                                  const float* bias, int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, void* stream) {
    RD_CHECK_ARG(in_pieces != nullptr, "gconv_split_pre: null argument");
    return gs_launch(d, nullptr, in_pieces, in_piece_elems, w_split, piece_elems, out, bias, act, act_cols, addend, ld_add, stat_partial, stream);
}

// Input gradient + the BatchNorm-backward sums of the BatchNorm in front of the convolution, in one launch (rd_gconv_bnbwd's contract on the
// split kernels): d is the input-gradient descriptor, out = dx (no addend), bn_x = the BatchNorm's input (the raw output of the previous
// convolution, [N,Ho,Wo,Cout] with channel stride bn_ld), mean / scale / shift its per-channel coefficients, act the activation behind it;
// red_partial [rd_gconv_split[_pre]_stat_tiles(d)][3][Cout]: slot 0 = sum g, slot 1 = sum g (x - mean), g = dx * act'(scale x + shift)
// -- what rd_bn_bwd_reduce_x_t computes in a pass of its own over dx and x.  Not for one-tap descriptors.
extern "C" int rd_gconv_split_bnbwd_supported(const RdConvDesc* d, int32_t pre) {
    GsPlan pl; RdConvDesc dd;
    if (!d || d->Cout % 4 != 0 || d->ldo % 4 != 0) return 0;
    if (!pre && gemm1_split_supported(d)) return 0;
    return gs_plan_query(d, pl, dd, pre != 0);
}
extern "C" int rd_gconv_split_bnbwd(const RdConvDesc* d, const float* in, const void* w_split, int64_t piece_elems, float* out, const float* bn_x,
                                    int32_t bn_ld, const float* mean, const float* scale, const float* shift, int32_t bn_act, float* red_partial,
                                    void* stream) {
    RD_CHECK_ARG(in && bn_x, "gconv_split_bnbwd: null argument");
    GsBnb b;
    b.x = bn_x; b.ld = bn_ld; b.mean = mean; b.scale = scale; b.shift = shift; b.act = bn_act;
    return gs_launch(d, in, nullptr, 0, w_split, piece_elems, out, nullptr, 0, 0, nullptr, 0, red_partial, stream, b);
}
extern "C" int rd_gconv_split_pre_bnbwd(const RdConvDesc* d, const void* in_pieces, int64_t in_piece_elems, const void* w_split, int64_t piece_elems,
                                        float* out, const float* bn_x, int32_t bn_ld, const float* mean, const float* scale, const float* shift,
                                        int32_t bn_act, float* red_partial, void* stream) {
    RD_CHECK_ARG(in_pieces && bn_x, "gconv_split_pre_bnbwd: null argument");
    GsBnb b;
    b.x = bn_x; b.ld = bn_ld; b.mean = mean; b.scale = scale; b.shift = shift; b.act = bn_act;
    return gs_launch(d, nullptr, in_pieces, in_piece_elems, w_split, piece_elems, out, nullptr, 0, 0, nullptr, 0, red_partial, stream, b);
}

extern "C" int rd_gconv_split_pre_supported(const RdConvDesc* d) {
    GsPlan pl; RdConvDesc dd;
    return gs_plan_query(d, pl, dd, true);
}

// 1 when the pre-split form is expected to beat BOTH rd_gconv_split (split while staging) and rd_gconv on d by more than the extra
// 6 bytes per element its producer has to write -- measured at b = 16, 450 x 800 (profiles/r04_bench_split_pre.txt): the shapes
// rd_gconv_split does not serve at all (32-channel layers: gconv_sp2_kernel's 32-wide tiles at two workgroups per CU), and the
// 64-channel-input layers with >= 2.5 workgroups per CU on the 2 x 2 tile (layer1: 178 -> 142 us; UpProj 64: 134 -> 123 us), where the
// prologue / epilogue of one-workgroup-per-CU launches is a quarter of the lifetime.  Callers (engine.py) use the plain forms otherwise.
extern "C" int rd_gconv_split_pre_preferred(const RdConvDesc* d) {
    GsPlan pl; RdConvDesc dd;
    if (gs_plan_query(d, pl, dd, true) != 1 || !pl.sp2) return 0;
    static const char* force = getenv("RD_GCONV_PRE_PREFER");    // diagnostics: "all" / "none"
    if (force) return force[0] == 'a';
    GsPlan p8; RdConvDesc d8;
    if (gs_plan_query(d, p8, d8, false) != 1) return 1;
    const int wgs = d->N * pl.tiles_total * pl.n_cotiles;
    // (>= 2.5 workgroups per CU: b = 8 of BASELINE config 4 -- 736 workgroups on layer1 -- gains like b = 16 does, 389.6 -> 393.6 -> 398.5
    //  samples/s with none / the b = 16 rule / all pre-split, profiles/r04_*)
    if (pl.MT == 2 && pl.NT == 2 && 2 * wgs >= 5 * num_cus() && d->Cin <= 64) return 1;
    // small launches the 8-wave tiling cannot spread over the chip (depth encoder layer3, 64 channels at 29 x 50: 240 workgroups of one per CU,
    // 23.8 us; 480 four-wave workgroups at two per CU: 19.1 us -- profiles/r04_bench_split_pre.txt with RD_GCONV_SPLIT_ALL=1)
    static const bool small_rule = !(getenv("RD_GCONV_PRE_SMALL") && atoi(getenv("RD_GCONV_PRE_SMALL")) == 0);      // (A/B switch)
    const int wgs8 = d->N * p8.tiles_total * p8.n_cotiles;
    return small_rule && wgs8 < num_cus() && 2 * wgs >= 3 * wgs8 ? 1 : 0;
}

extern "C" int rd_gconv_split_pre_stat_tiles(const RdConvDesc* d) {
    GsPlan pl; RdConvDesc dd;
    if (gs_plan_query(d, pl, dd, true) != 1) return RD_EINVAL;
    return d->N * pl.tiles_total;
}

// diagnostics: rd_gconv_split_plan_info for the pre-split plan
extern "C" int rd_gconv_split_pre_plan_info(const RdConvDesc* d, int32_t* out) {
    GsPlan pl; RdConvDesc dd;
    if (!out || gs_plan_query(d, pl, dd, true) != 1) return RD_EINVAL;
    const int v[8] = {pl.MT, pl.NT, pl.TH, pl.TW, pl.PP, (int)pl.lds_bytes, d->N * pl.tiles_total * pl.n_cotiles, (pl.taps_max + GS_TPS - 1) / GS_TPS + 100 * pl.pdb};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RD_OK;
}

// ---- fp32 NHWC tensor -> three bf16 piece planes [piece][C/16][M][16] (x = p0 + p1 + p2 exactly); the stand-alone form of what the
// BatchNorm / activation kernels emit from their epilogues (norm_act.hip)
namespace rd {
__global__ __launch_bounds__(256) void split_pieces_kernel(const float* __restrict__ x, int ldx, int64_t M, int C, unsigned short* __restrict__ pc, int64_t plane) {
    const int Q = C >> 2;
    const int lq = quad_log2(Q);
    const int64_t total = M * Q;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r;
        int c;
        split_quad(e, Q, lq, r, c);
        store_pieces4(pc, plane, M, r, c, ld4(x + r * ldx + c));
    }
}
}  // namespace rd

extern "C" int rd_split_pieces(const float* x, int32_t ldx, int64_t M, int32_t C, void* pieces, int64_t piece_elems, void* stream) {
    RD_CHECK_ARG(x && pieces && M > 0 && C >= 16 && C % 16 == 0 && ldx % 4 == 0, "split_pieces: bad arguments (C must be a multiple of 16)");
    RD_CHECK_ARG(piece_elems >= (int64_t)C * M && piece_elems % 8 == 0, "split_pieces: piece stride too small");
    const int64_t total = M * (C / 4);
    int64_t g = cdiv64(total, 256);
    const int64_t cap = (int64_t)num_cus() * 8;
    hipLaunchKernelGGL(split_pieces_kernel, dim3((unsigned)(g < cap ? g : cap)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, M, C,
                       static_cast<unsigned short*>(pieces), piece_elems);
    RD_CHECK_LAUNCH("split_pieces_kernel");
    return RD_OK;
}
