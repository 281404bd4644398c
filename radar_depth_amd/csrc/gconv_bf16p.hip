// Persistent, software-pipelined bf16-storage convolution (round 5; BASELINE.json configs 3 / 5): the unit-stride-input descriptors of
// rd_gconv_bf16_t with bf16 tensors in HBM -- 3x3 forward and input gradient, the four-phase UpProj forward -- on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same contract, descriptor, packed weights, epilogue and statistics layout as
// gconv_bf16.hip, which stays the kernel for everything else (stride-2 inputs, the 25-tap UpProj input gradient, 1x1, fp32 storage).
//
// Why another kernel.  gconv_bf16_kernel stages a chunk (weights + patch), waits, and only then issues the chunk's MFMAs; with one
// bf16 MFMA per fragment pair a chunk's MFMAs are a few hundred clocks while its copies take one to two thousand to land, and the two
// or three co-resident workgroups that are supposed to hide that start together and stay in step: 0.13 of the HBM rate and 0.15 of the
// matrix peak on the 64-channel layers for three rounds (profiles/r04_bench_c3.json).  What the hardware can do
// (tools/micro/fill_rate.hip, profiles/r05_fill_rate.txt): a CU ingests 47-57 B/clock of L2-resident data (the weights) into the
// LDS with four waves keeping 4-8 copies in flight each, 10-14 B/clock from HBM; the old loop saw ~10 in total because its copies
// go out in one burst per chunk and are waited for at once.
//
//   * persistent workgroups: 2 per CU (4 waves, <= 80 KB of LDS, <= 256 registers), each walks the tile list job = wg, wg + G, ...
//     The copies of the NEXT stage -- also across the tile boundary: the next tile's first stage is requested before the current
//     tile's last MFMAs and lands under them and the epilogue -- are issued right behind the barrier that frees their ring slot: the
//     weights (L2 hits) one stage (36 MFMAs per wave on the 3x3 layers) ahead, the patch (HBM) two stages ahead; every second
//     workgroup of a CU starts half a tile late (measured: the start offset changes nothing, profiles/r05_ablate_bf16p.txt).
//   * stage = one 16-channel chunk of the patch x all taps of the phase: weights [tap][2][BN] x 16 B copied with global_load_lds from
//     the packed operand (already in this layout), patch rows copied with buffer_load_dwordx4 ... lds straight from the bf16 NHWC
//     tensor: one instruction per patch row (lanes 0-31: channels 0-7 of 32 columns, lanes 32-63: channels 8-15), out-of-image
//     lanes read an out-of-range buffer offset, for which the hardware writes ZEROS into the LDS (tools/micro/buf_lds_oob.hip): no
//     masks, no pre-zeroed buffer.  No staging registers, no conversion, no LDS store instruction anywhere.
//   * LDS patch image [row][unit][32 columns] x 16 B with a row pitch of 64 + p slots (p = 0..15 chosen by the planner): a tap is a
//     wave-uniform offset (dh rows + dw slots), and the slot map of gconv_split.hip (slot_map.h) places the 16 lanes of every
//     ds_read_b128 pass on 16 different slots mod 16 -- the pad p is what balances the residue classes r * pitch + c of a tile
//     whose width is not a multiple of 16.  Two ring slots for the weights, three for the patch; one raw s_barrier per stage.
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

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));

constexpr unsigned GP_OOB = 0x80000000u;
constexpr int GP_MAXROWS = 3;      // patch rows per wave and stage (4 waves: at most 12 rows)

struct GpArgs {
    RdConvDesc d;
    const bf16s* in;
    const unsigned short* w;      // packed bf16 operand [slab][Cin/8][ldw][8]
    bf16s* out;
    const bf16s* addend;
    const float* bias;
    float* stat;
    int act, act_cols, ld_add, ldw;
    int TH, TW, tiles_total, n_cotiles, njobs, taps_max;
    int vec4;
    int wslot, pslot;             // bytes per ring slot (weights / patch)
    int rowb;                     // bytes per patch row in the LDS: (64 + pad) x 16
    int stagger;                  // start delay of every second group of 256 workgroups, in units of 64 clocks
    int dbg;                      // diagnostics (RD_GCONV_BF16P_DEBUG; results are then garbage): 1 no MFMAs, 4 no weight copies, 8 no patch copies, 16 no epilogue
    int tapoff[RD_MAX_PHASES][RD_MAX_TAPS];   // byte offset of tap t inside the patch image
    const int* slots;             // slot map [BM]: (r << 16) | c, or -(1 + column residue) for an empty slot
};

// raw workgroup barrier behind this wave's own LDS traffic only: __syncthreads() would also wait (vmcnt(0)) for the copies in flight
__device__ __forceinline__ void gp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int MT, int NT, int DBG = 0>      // DBG (diagnostics, tools/ablate_bf16p.py): 1 no MFMAs -- compile-time: a run-time test inside the
                                            // step splits the basic block and the fragment reads lose their counted waits
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gconv_bf16p_kernel(const GpArgs a) {
    constexpr int BM = 4 * MT * 32;
    constexpr int BN = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;
    const int OS = D.out_stride;

    // LDS carve-up
    int* s_rc = reinterpret_cast<int*>(smem);                    // [BM] (r << 16) | c of the slot's tile pixel, -1: empty
    float* s_red = reinterpret_cast<float*>(s_rc + BM);          // [4][2][BN] statistics scratch
    int* s_tap = reinterpret_cast<int*>(s_red + 8 * BN);         // [phases][32] tap offsets (read per tile with an LDS load: a vector
                                                                 //  load from the kernel arguments would wait on vmcnt, i.e. on copies)
    char* s_w = reinterpret_cast<char*>(s_tap + RD_MAX_PHASES * 32);   // [2][wslot]
    char* s_p = s_w + 2 * a.wslot;                               // [3][pslot]

    int aoff[MT];
    for (int m = tid; m < BM; m += 256) {
        const int sv = a.slots[m];
        s_rc[m] = sv >= 0 ? sv : -1;
    }
    if (tid < RD_MAX_PHASES * 32) s_tap[tid] = a.tapoff[tid >> 5][(tid & 31) < RD_MAX_TAPS ? (tid & 31) : 0];
    const int rowb = a.rowb;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int sv = a.slots[(wm * MT + mt) * 32 + l31];
        const int r = sv >= 0 ? (sv >> 16) : 0, c = sv >= 0 ? (sv & 0xffff) : (-1 - sv);
        aoff[mt] = r * rowb + hh * 512 + c * 16;
    }
    rd_sync();
    const int boff = (hh * BN + l31) * 16;
    const int cin8 = D.Cin >> 3;
    const int S = D.Cin >> 4;                                    // stages per tile
    const int G = gridDim.x;
    if (a.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }

    // ---- job decoding (everything wave-uniform)
    struct Job { int ph, n, r0, c0, th_n, tw_n, co0, pt, ntaps, PW, PH, ih0, iw0; };
    auto decode = [&](int j) {
        Job J;
        const int cot = j % a.n_cotiles;
        J.pt = j / a.n_cotiles;
        J.n = J.pt / a.tiles_total;
        const int tt = J.pt - J.n * a.tiles_total;
        int ph = 0;
        for (int i = 1; i < D.n_phases; ++i)
            if (tt >= D.phase[i].tile_begin) ph = i;
        J.ph = ph;
        const RdPhase& P = D.phase[ph];
        const int tloc = tt - P.tile_begin;
        const int tiles_w = (P.lw + a.TW - 1) / a.TW;
        J.r0 = (tloc / tiles_w) * a.TH;
        J.c0 = (tloc % tiles_w) * a.TW;
        J.th_n = min(a.TH, P.lh - J.r0);
        J.tw_n = min(a.TW, P.lw - J.c0);
        J.co0 = cot * BN;
        J.ntaps = P.n_taps;
        J.PW = a.TW + (P.dw_max - P.dw_min);
        J.PH = J.th_n + (P.dh_max - P.dh_min);
        J.ih0 = J.r0 + P.dh_min;
        J.iw0 = J.c0 + P.dw_min;
        return J;
    };
    // per-job copy state of this wave: byte offsets of its patch rows inside the image at channel 0 (GP_OOB: zeros), the image base
    unsigned pvo[GP_MAXROWS];
    bool pact = false;
    const char* img = nullptr;
    const unsigned img_bytes = (unsigned)(D.Hi * D.Wi * D.ldi) * 2u;
    auto job_rows = [&](const Job& J) {
        const int c = l31;
        pact = c < J.PW;
        const int iw = J.iw0 + c;
#pragma unroll
        for (int k = 0; k < GP_MAXROWS; ++k) {
            const int r = wm + 4 * k;
            const int ih = J.ih0 + r;
            const bool in = r < J.PH && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi;
            pvo[k] = in ? (unsigned)(((ih * D.Wi + iw) * D.ldi + hh * 8) * 2) : GP_OOB;
        }
        img = reinterpret_cast<const char*>(a.in) + (size_t)J.n * D.Hi * D.Wi * D.ldi * 2;
    };
    // weights of stage (J, chunk cb) into weight slot `slot`: [tap][2][BN] x 16 B (copy e = (tap, k8) for BN = 64, e = tap for BN = 32)
    auto issue_w = [&](const Job& J, int cb, int slot) {
        if (a.dbg & 4) return;
        const RdPhase& P = D.phase[J.ph];
        char* wdst = s_w + slot * a.wslot;
        const char* wsrc = reinterpret_cast<const char*>(a.w) + ((size_t)(cb >> 3) * a.ldw + J.co0) * 16;
        if constexpr (NT == 2) {
            const int ne = J.ntaps * 2;
            for (int e = wm; e < ne; e += 4) {
                const int t = e >> 1, k8 = e & 1;
                const unsigned go = (((unsigned)P.widx[t] * cin8 + k8) * a.ldw + lane) * 16u;
                glds16(reinterpret_cast<const float*>(wsrc + go), reinterpret_cast<float*>(wdst + e * 1024));
            }
        } else {
            for (int t = wm; t < J.ntaps; t += 4) {
                const unsigned go = (((unsigned)P.widx[t] * cin8 + hh) * a.ldw + l31) * 16u;
                glds16(reinterpret_cast<const float*>(wsrc + go), reinterpret_cast<float*>(wdst + t * 1024));
            }
        }
    };
    // this wave's patch rows of chunk cb of the job job_rows() was last called for, into patch slot `slot`; returns the number of copies
    // issued (the counted wait of the next barrier leaves exactly these in flight)
    auto issue_p = [&](int PH, int cb, int slot) -> int {
        if (a.dbg & 8) return 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img + cb * 2), 0, img_bytes - cb * 2, 0x00020000);
        char* pdst = s_p + slot * a.pslot;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < GP_MAXROWS; ++k) {
            const int r = wm + 4 * k;
            if (r < PH) {
                ++cnt;
                if (pact)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, reinterpret_cast<__attribute__((address_space(3))) void*>((unsigned)(size_t)(pdst + r * rowb)),
                                                             16, (int)pvo[k], 0, 0, 0);
            }
        }
        return cnt;
    };

    // ---- the pipeline.  Stages are numbered g = 0, 1, ... over all tiles of this workgroup.  At stage g, behind its barrier:
    //        weights of stage g + 1 -> weight slot (g + 1) & 1       (L2 hits: one stage of lead)
    //        patch   of stage g + 2 -> patch slot (g + 2) % 3        (HBM: two stages of lead; measured with one stage of lead the
    //                                                                 MFMAs of every stage waited ~2.7 k clocks for their patch)
    //      both slots were last read in stage g - 1, which every wave has left when it passes the barrier of stage g.  The wait in
    //      front of that barrier leaves this wave's newest patch copies (stage g + 1's, issued in stage g - 1) in flight: counted,
    //      loads retire in order.  Behind a tile's epilogue the count would include its stores, which need not retire in order with
    //      the loads: the first stage of every tile waits for everything.
    struct Cursor { Job J; int j, s; bool live; };
    auto advance = [&](Cursor& c) {
        if (!c.live) return;
        if (c.s + 1 < S) { ++c.s; return; }
        c.j += G;
        c.s = 0;
        c.live = c.j < a.njobs;
        if (c.live) c.J = decode(c.j);
    };
    f32x16 acc[MT][NT];
    if ((int)blockIdx.x >= a.njobs) return;
    Cursor cc{decode(blockIdx.x), (int)blockIdx.x, 0, true};      // compute cursor
    Cursor cw = cc, cp = cc;                                       // weight / patch copy cursors (one / two stages ahead)
    job_rows(cp.J);
    int ppend = 0;                                                 // patch copies this wave issued in the previous stage
    issue_w(cw.J, 0, 0);
    advance(cw);
    issue_p(cp.J.PH, 0, 0);
    advance(cp);
    if (cp.live) {
        if (cp.s == 0) job_rows(cp.J);
        ppend = issue_p(cp.J.PH, cp.s * 16, 1);
        advance(cp);
        if (cp.live && cp.s == 0) job_rows(cp.J);
    }
    int g = 0, pslot_c = 0, pslot_i = 2;                           // patch slot of the stage being computed / being filled
    bool after_store = true;                                       // (first stage: wait for everything)
    for (;;) {
        const Job& J = cc.J;
        const int tapv = s_tap[J.ph * 32 + (lane & 31)];
        const int ntaps = __builtin_amdgcn_readfirstlane(J.ntaps);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        for (int s = 0; s < S; ++s, ++g) {
            if (after_store || ppend >= GP_MAXROWS + 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (ppend == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (ppend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (ppend == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            after_store = false;
            gp_barrier();                        // stage g's weights and patch are published; every wave has left stage g - 1
            if (cw.live) issue_w(cw.J, cw.s * 16, (g + 1) & 1);
            advance(cw);
            ppend = 0;
            if (cp.live) {
                ppend = issue_p(cp.J.PH, cp.s * 16, pslot_i);
                advance(cp);
                if (cp.live && cp.s == 0) job_rows(cp.J);          // (its first copies go out in the next stage)
            }
            pslot_i = pslot_i == 2 ? 0 : pslot_i + 1;
            // ---- the stage's MFMAs: one 16-channel step per tap; fragment ring of three steps, reads two steps ahead (gconv_bf16.hip)
            const char* wb = s_w + (g & 1) * a.wslot + boff;
            const char* pb = s_p + pslot_c * a.pslot;
            pslot_c = pslot_c == 2 ? 0 : pslot_c + 1;
            // Fragment reads as inline assembly with hand-counted waits.  Left to the compiler the ring of three steps is waited for with
            // lgkmcnt(0) -- all reads, including the ones issued an instruction earlier -- once per three steps (straight-line code) or
            // once per trip (loop): ~150 clocks of exposed LDS latency each time, and two code paths (an unrolled 9-tap one next to the
            // loop) made it copy the 64 accumulator registers between them.  Here every step issues the reads of step t + 2, waits until
            // at most the 2 (MT + NT) newest reads are outstanding (LDS reads retire in order), and issues its MFMAs.
            auto load = [&](int t, pbf16x8 (&A)[MT], pbf16x8 (&B)[NT]) {
                const int ao = __builtin_amdgcn_readlane(tapv, t);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned ad = (unsigned)(size_t)(pb + aoff[mt] + ao);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(A[mt]) : "v"(ad) : "memory");
                }
                const unsigned bd = (unsigned)(size_t)(wb + t * (2 * BN * 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[nt]) : "v"(bd), "n"(nt * 512) : "memory");
            };
            auto mma = [&](pbf16x8 (&A)[MT], pbf16x8 (&B)[NT], auto pending) {
                // wait until at most `pending` newer fragment sets are in flight; the fragments are operands of the wait so that the
                // MFMAs below cannot be scheduled above it
                constexpr int cnt = decltype(pending)::value * (MT + NT);
                if constexpr (MT == 2 && NT == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0]), "+v"(B[1]) : "n"(cnt));
                else if constexpr (MT == 2 && NT == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0]) : "n"(cnt));
                else if constexpr (MT == 1 && NT == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(A[0]), "+v"(B[0]), "+v"(B[1]) : "n"(cnt));
                else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(A[0]), "+v"(B[0]) : "n"(cnt));
                if constexpr (DBG & 1) return;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt], B[nt], acc[mt][nt], 0, 0, 0);
            };
            constexpr int DEPTH = 3;
            constexpr std::integral_constant<int, 2> P2{};
            constexpr std::integral_constant<int, 1> P1{};
            constexpr std::integral_constant<int, 0> P0{};
            pbf16x8 fa[DEPTH][MT], fb[DEPTH][NT];
            const int last = ntaps - 1;
            load(0, fa[0], fb[0]);
            load(min(1, last), fa[1], fb[1]);
            int t0 = 0;
            for (; t0 + DEPTH <= ntaps; t0 += DEPTH) {
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) {
                    load(min(t0 + q + 2, last), fa[(q + 2) % DEPTH], fb[(q + 2) % DEPTH]);      // (past the end: re-reads the last step)
                    mma(fa[q], fb[q], P2);
                }
            }
            if (t0 < ntaps) mma(fa[0], fb[0], P1);
            if (t0 + 1 < ntaps) mma(fa[1], fb[1], P0);
            // (every fragment read has been waited for or is a redundant re-read: the barrier's lgkmcnt(0) retires those)
        }
        after_store = true;

        // ---- epilogue (gconv_bf16.hip's: 4x4 register transposition, 8-byte bf16 accesses; rows are masked one by one)
        if (!(a.dbg & 16)) {
            const RdPhase& P = D.phase[J.ph];
            const bool has_add = a.addend != nullptr;
            const bool has_bias = a.bias != nullptr;
            const bool want_stat = a.stat != nullptr;
            const int co0 = J.co0;
            const int q4l = l31 & 3, k4l = l31 >> 2;
            const bool odd1 = q4l & 1, odd2 = q4l & 2;
            const int obase = (J.n * D.Ho + J.r0 * OS + P.out_off_h) * D.Wo + J.c0 * OS + P.out_off_w;
            float ssum[NT], ssq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
            if (a.vec4) {
                float4 ssum4[NT], ssq4[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) ssum4[nt] = ssq4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int ro4[4];
                    bool rok4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rc = s_rc[(wm * MT + mt) * 32 + q4l + 8 * q + 4 * hh];
                        const int r = rc >> 16, c = rc & 0xffff;
                        rok4[q] = rc >= 0 && r < J.th_n && c < J.tw_n;
                        ro4[q] = rok4[q] ? obase + (r * OS) * D.Wo + c * OS : 0;
                    }
                    const int cq = co0 + 4 * k4l;
                    float4 addv[NT][4];
                    if (has_add) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bf16s* ap = a.addend + (size_t)ro4[q] * a.ld_add + cq;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) addv[nt][q] = ld4(ap + nt * 32);
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (has_bias) b4 = *reinterpret_cast<const float4*>(a.bias + cq + nt * 32);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float e0 = acc[mt][nt][4 * q], e1 = acc[mt][nt][4 * q + 1], e2 = acc[mt][nt][4 * q + 2], e3 = acc[mt][nt][4 * q + 3];
                            quad_transpose(e0, e1, e2, e3, odd1, odd2);
                            float4 v = make_float4(e0 + b4.x, e1 + b4.y, e2 + b4.z, e3 + b4.w);
                            if (has_add) { v.x += addv[nt][q].x; v.y += addv[nt][q].y; v.z += addv[nt][q].z; v.w += addv[nt][q].w; }
                            const int cc = cq + nt * 32;
                            if (cc < a.act_cols) {
                                v.x = act_fwd(v.x, a.act); v.y = act_fwd(v.y, a.act); v.z = act_fwd(v.z, a.act); v.w = act_fwd(v.w, a.act);
                            }
                            if (rok4[q]) st4(a.out + (size_t)ro4[q] * D.ldo + cc, v);
                            if (want_stat && rok4[q]) {
                                ssum4[nt].x += v.x; ssum4[nt].y += v.y; ssum4[nt].z += v.z; ssum4[nt].w += v.w;
                                ssq4[nt].x += v.x * v.x; ssq4[nt].y += v.y * v.y; ssq4[nt].z += v.z * v.z; ssq4[nt].w += v.w * v.w;
                            }
                        }
                    }
                }
                if (want_stat) {
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
            } else {
                const int cob = co0 + l31;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rc = s_rc[(wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
                        const int r = rc >> 16, c = rc & 0xffff;
                        if (rc >= 0 && r < J.th_n && c < J.tw_n) {
                            const int ro = obase + (r * OS) * D.Wo + c * OS;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const int co = cob + nt * 32;
                                float v = acc[mt][nt][i] + (has_bias ? a.bias[co] : 0.f);
                                if (has_add) v += ld1(a.addend + (size_t)ro * a.ld_add + co);
                                if (co < a.act_cols) v = act_fwd(v, a.act);
                                st1(a.out + (size_t)ro * D.ldo + co, v);
                                ssum[nt] += v;
                                ssq[nt] += v * v;
                            }
                        }
                    }
                }
            }
            if (want_stat) {
                // (the scratch is free: its readers of the previous tile finished before that tile's successor passed a stage barrier)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float sv = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
                    const float qv = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
                    if (hh == 0) {
                        s_red[(wm * 2 + 0) * BN + nt * 32 + l31] = sv;
                        s_red[(wm * 2 + 1) * BN + nt * 32 + l31] = qv;
                    }
                }
                gp_barrier();
                if (tid < 2 * BN) {
                    const int which = tid / BN, jj = tid - which * BN;
                    float sv = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) sv += s_red[(w * 2 + which) * BN + jj];
                    a.stat[((size_t)J.pt * 2 + which) * D.Cout + co0 + jj] = sv;
                }
            }
        }
        cc.j += G;
        if (cc.j >= a.njobs) break;
        cc.J = decode(cc.j);
    }
}

// ------------------------------------------------------------------------------------------ host
struct GpPlan {
    int MT, NT, TH, TW, tiles_total, n_cotiles, taps_max, wslot, pslot, rowp;
    size_t lds_bytes;
    unsigned nres[4];
};

static bool gp_shape_ok(const RdConvDesc* d) {
    if (!d || d->n_phases < 1 || d->n_phases > RD_MAX_PHASES) return false;
    if (d->in_stride != 1 || d->out_stride < 1 || d->out_stride > 2) return false;
    if (d->Cin % 16 != 0 || d->Cin < 32 || d->ldi % 8 != 0 || d->Cout % 32 != 0) return false;
    if ((int64_t)d->Hi * d->Wi * d->ldi * 2 >= (int64_t)GP_OOB) return false;
    int taps_max = 0;
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        if (p.n_taps < 1 || p.n_taps > 9 || p.lh < 1 || p.lw < 1) return false;
        taps_max = taps_max > p.n_taps ? taps_max : p.n_taps;
        for (int t = 0; t < p.n_taps; ++t)
            if (p.dh[t] < p.dh_min || p.dh[t] > p.dh_max || p.dw[t] < p.dw_min || p.dw[t] > p.dw_max) return false;
    }
    if (taps_max < 4) return false;          // 1x1 layers: a stage would be four MFMAs
    return true;
}

static bool plan_bf16p(const RdConvDesc& d, GpPlan& pl) {
    int taps_max = 0, halo_w = 0, halo_h = 0, pr = 0;
    for (int i = 0; i < d.n_phases; ++i) {
        const RdPhase& p = d.phase[i];
        taps_max = taps_max > p.n_taps ? taps_max : p.n_taps;
        halo_w = halo_w > p.dw_max - p.dw_min ? halo_w : p.dw_max - p.dw_min;
        halo_h = halo_h > p.dh_max - p.dh_min ? halo_h : p.dh_max - p.dh_min;
        if ((int64_t)p.lh * p.lw > (int64_t)d.phase[pr].lh * d.phase[pr].lw) pr = i;
    }
    const RdPhase& P = d.phase[pr];
    const int NT = d.Cout % 64 == 0 ? 2 : 1, BN = NT * 32;
    const int n_cot = d.Cout / BN;
    static const char* force_mt = getenv("RD_GCONV_BF16P_MT");      // diagnostics
    double best = -1;
    for (int MT = 2; MT >= 1; --MT) {
        if (force_mt && atoi(force_mt) != MT) continue;
        const int BM = 4 * MT * 32;
        const int tw_max = 32 - halo_w;
        for (int twt = cdiv(P.lw, tw_max); twt <= cdiv(P.lw, 4); ++twt) {
            const int TW = cdiv(P.lw, twt);
            if (TW > tw_max) continue;
            int TH = BM / TW;
            if (TH + halo_h > 4 * GP_MAXROWS) TH = 4 * GP_MAXROWS - halo_h;
            if (TH > P.lh) TH = P.lh;
            if (TH < 1) continue;
            TH = cdiv(P.lh, cdiv(P.lh, TH));
            // LDS row pitch 64 + p slots: the pad with the fewest tile pixels outside the conflict-free passes of the slot map
            int n16[16], rowp = 64, over = -1;
            for (int pad = 0; pad < 16; ++pad) {
                const int o = gs_residues(TH, TW, 64 + pad, BM / 16, n16);
                if (over < 0 || o < over) { over = o; rowp = 64 + pad; }
                if (o == 0) break;
            }
            gs_residues(TH, TW, rowp, BM / 16, n16);
            const int wslot = taps_max * 2 * BN * 16, pslot = (TH + halo_h) * rowp * 16;
            const size_t lds = (size_t)BM * 4 + 8 * BN * 4 + RD_MAX_PHASES * 32 * 4 + 2 * (size_t)wslot + 3 * (size_t)pslot + 64;
            if (lds > 80 * 1024 - 256) continue;
            double tiles = 0, slots_used = 0;
            for (int i = 0; i < d.n_phases; ++i) {
                tiles += (double)cdiv(d.phase[i].lh, TH) * cdiv(d.phase[i].lw, TW);
                slots_used += (double)d.phase[i].lh * d.phase[i].lw;
            }
            const double jobs = tiles * d.N * n_cot;
            const double G = 2.0 * num_cus();
            // MFMA work issued (tiles x BM slots) with the persistent grid's tail (jobs per workgroup rounded up), a per-tile
            // overhead for the epilogue / barriers, and the halo's extra fill per tile
            const double rounds = ceil(jobs / G);
            const double per_tile = (double)BM * (1.0 + 2.0 / (d.Cin / 16.0)) + 0.15 * (TH + halo_h) * 32.0 + 2.0 * over;
            const double cost = rounds * per_tile * (jobs < G ? G / jobs : 1.0);
            (void)slots_used;
            if (best < 0 || cost < best) {
                best = cost;
                pl = GpPlan{MT, NT, TH, TW, 0, n_cot, taps_max, wslot, pslot, rowp, lds, {0, 0, 0, 0}};
                for (int w = 0; w < 4; ++w)
                    for (int b = 0; b < 4; ++b) pl.nres[w] |= (unsigned)n16[4 * w + b] << (8 * b);
            }
        }
    }
    return best > 0;
}

static std::mutex g_gp_mu;
static int g_gp_all = -1;          // -1: not set yet (RD_GCONV_BF16P=all decides at first use)
static unsigned g_gp_epoch = 0;    // bumped by gconv_bf16p_plan_all: plans cached under the other setting are dropped

static int gp_plan_query(const RdConvDesc* d, GpPlan& pl, RdConvDesc& dd) {
    struct Entry { int ok; GpPlan pl; RdConvDesc dd; };
    static std::mutex mu;
    static std::unordered_map<std::string, Entry> cache;
    static unsigned cache_epoch = 0;
    if (!d) return 0;
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    bool all;
    {
        unsigned ep;
        {
            std::lock_guard<std::mutex> lk(g_gp_mu);
            if (g_gp_all < 0) g_gp_all = (getenv("RD_GCONV_BF16P") && getenv("RD_GCONV_BF16P")[0] == 'a') ? 1 : 0;
            all = g_gp_all == 1;
            ep = g_gp_epoch;
        }
        std::lock_guard<std::mutex> lk(mu);
        if (ep != cache_epoch) { cache.clear(); cache_epoch = ep; }
        auto it = cache.find(key);
        if (it != cache.end()) { pl = it->second.pl; dd = it->second.dd; return it->second.ok; }
    }
    Entry e{};
    e.dd = *d;
    static const bool off = getenv("RD_GCONV_BF16P") && atoi(getenv("RD_GCONV_BF16P")) == 0;      // A/B switch: everything on gconv_bf16_kernel
    e.ok = !off && gp_shape_ok(d) && plan_bf16p(e.dd, e.pl) ? 1 : 0;
    if (e.ok) {
        // Planner rule from the per-layer A/B at b = 16, 450 x 800 (tools/bench_bf16_storage_ops.py, profiles/r05_bf16_storage_ops_*.txt):
        // the persistent kernel wins where a workgroup walks several 256-pixel tiles -- layer1 53 vs 57 us, the 32-channel decoder
        // layers 25.6 vs 31.9 and 79 vs 93 us, the stride-2 layers' input gradient 46 vs 52 us -- and loses on the 128-pixel tiles it
        // needs to fill the chip on the small-spatial layers (layer2-4: 62 / 74 / 88 vs 44 / 55 / 58 us: twice the weight copies and
        // three fragment reads per two MFMAs); rd_gconv_bf16p_plan_all(1) / RD_GCONV_BF16P=all serve every shape the kernel can run.
        int tb0 = 0;
        for (int i = 0; i < e.dd.n_phases; ++i) tb0 += cdiv(e.dd.phase[i].lh, e.pl.TH) * cdiv(e.dd.phase[i].lw, e.pl.TW);
        if (!all && !(e.pl.MT == 2 && (int64_t)d->N * tb0 * e.pl.n_cotiles >= 4 * (int64_t)num_cus())) e.ok = 0;
    }
    if (e.ok) {
        int tb = 0;
        for (int i = 0; i < e.dd.n_phases; ++i) {
            e.dd.phase[i].tile_begin = tb;
            tb += cdiv(e.dd.phase[i].lh, e.pl.TH) * cdiv(e.dd.phase[i].lw, e.pl.TW);
        }
        e.pl.tiles_total = tb;
    }
    pl = e.pl; dd = e.dd;
    std::lock_guard<std::mutex> lk(mu);
    cache.emplace(std::move(key), e);
    return e.ok;
}

// slot table of a plan on the current device (see gs_slot_table in gconv_split.hip); keyed by the tile alone
static const int* gp_slot_table(const GpPlan& pl) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, int*> tables;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const uint64_t key = ((uint64_t)dev << 56) | ((uint64_t)pl.MT << 48) | ((uint64_t)pl.rowp << 32) | ((uint64_t)pl.TH << 16) | (uint64_t)pl.TW;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tables.find(key);
    if (it != tables.end()) return it->second;
    const int BM = 4 * pl.MT * 32;
    std::vector<int> host(BM);
    for (int m = 0; m < BM; ++m) {
        int r, c, rho;
        host[m] = gs_slot_pixel(m, pl.TH, pl.TW, pl.rowp, BM / 16, pl.nres, false, r, c, rho) ? ((r << 16) | c) : -1 - rho;
    }
    int* devp = nullptr;
    if (hipMalloc(&devp, host.size() * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(devp, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(devp); return nullptr; }
    tables.emplace(key, devp);
    return devp;
}

template <int MT, int NT, int DBG = 0>
static int launch_gp(const GpArgs& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = gconv_bf16p_kernel<MT, NT, DBG>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("gconv_bf16p_kernel");
    return RD_OK;
}

// tests / sweeps: on != 0 serves every shape the kernel can run, also those the planner rule leaves to gconv_bf16_kernel; returns the
// previous setting.  Do not toggle between sizing a statistics buffer on a plan and launching it.
int gconv_bf16p_plan_all(int on) {
    std::lock_guard<std::mutex> lk(g_gp_mu);
    if (g_gp_all < 0) g_gp_all = (getenv("RD_GCONV_BF16P") && getenv("RD_GCONV_BF16P")[0] == 'a') ? 1 : 0;
    const int prev = g_gp_all;
    if ((on != 0) != (prev == 1)) { g_gp_all = on ? 1 : 0; ++g_gp_epoch; }
    return prev;
}

// 1 when the persistent kernel serves d with bf16 tensors (rd_gconv_bf16_t dispatches to it; its statistics tiling is its own)
int gconv_bf16p_supported(const RdConvDesc* d) {
    GpPlan pl; RdConvDesc dd;
    return gp_plan_query(d, pl, dd);
}

int gconv_bf16p_stat_tiles(const RdConvDesc* d) {
    GpPlan pl; RdConvDesc dd;
    if (gp_plan_query(d, pl, dd) != 1) return RD_EINVAL;
    return d->N * pl.tiles_total;
}

int gconv_bf16p_plan_info(const RdConvDesc* d, int32_t* out) {
    GpPlan pl; RdConvDesc dd;
    if (!out || gp_plan_query(d, pl, dd) != 1) return RD_EINVAL;
    const int njobs = d->N * pl.tiles_total * pl.n_cotiles;
    const int v[8] = {pl.MT, pl.NT, 16, pl.TH, pl.TW, 0, (int)pl.lds_bytes, njobs};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RD_OK;
}

int launch_gconv_bf16p(const RdConvDesc* d, const void* in, const void* w_packed_bf16, void* out, const float* bias, int32_t act, int32_t act_cols,
                       const void* addend, int32_t ld_add, float* stat_partial, hipStream_t s) {
    GpArgs a;
    GpPlan pl;
    if (gp_plan_query(d, pl, a.d) != 1) { set_error("gconv_bf16p: descriptor not supported"); return RD_EINVAL; }
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(w_packed_bf16) % 16 == 0, "gconv_bf16p: unaligned tensor");
    a.in = static_cast<const bf16s*>(in); a.w = static_cast<const unsigned short*>(w_packed_bf16); a.out = static_cast<bf16s*>(out);
    a.addend = static_cast<const bf16s*>(addend); a.bias = bias; a.stat = stat_partial;
    a.act = act; a.act_cols = act_cols; a.ld_add = ld_add; a.ldw = d->Cout;
    a.TH = pl.TH; a.TW = pl.TW; a.tiles_total = pl.tiles_total; a.n_cotiles = pl.n_cotiles; a.taps_max = pl.taps_max;
    a.njobs = d->N * pl.tiles_total * pl.n_cotiles;
    a.wslot = pl.wslot; a.pslot = pl.pslot; a.rowb = pl.rowp * 16;
    a.vec4 = d->ldo % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0 && act_cols % 4 == 0 &&
             (!addend || (ld_add % 4 == 0 && reinterpret_cast<uintptr_t>(addend) % 8 == 0)) && (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        for (int t = 0; t < p.n_taps; ++t) a.tapoff[i][t] = (p.dh[t] - p.dh_min) * a.rowb + (p.dw[t] - p.dw_min) * 16;
    }
    a.slots = gp_slot_table(pl);
    if (!a.slots) { set_error("gconv_bf16p: cannot allocate the slot table of the plan"); return RD_ELAUNCH; }
    {
        // every second workgroup of a CU starts half a tile late (64-clock units): S stages x taps x MT x NT MFMAs of 32 clocks, shared
        const char* dbg = getenv("RD_GCONV_BF16P_DEBUG");      // (read at every launch: the ablation tool toggles it)
        a.dbg = dbg ? atoi(dbg) : 0;
        const char* stg = getenv("RD_GCONV_BF16P_STAGGER");
        a.stagger = stg ? atoi(stg) : (int)((d->Cin / 16) * pl.taps_max * pl.MT * pl.NT * 32 / 64);
    }
    int cap = 2 * num_cus();
    {
        const char* gc = getenv("RD_GCONV_BF16P_GRID");      // tests / diagnostics (read at every launch): cap the persistent grid, so that
        if (gc && atoi(gc) > 0) cap = atoi(gc);              // small problems walk many tiles per workgroup
    }
    const int grid = a.njobs < cap ? a.njobs : cap;
    if (pl.MT == 2 && pl.NT == 2 && (a.dbg & 1)) return launch_gp<2, 2, 1>(a, grid, pl.lds_bytes, s);      // (ablation of the main tile only)
    if (pl.MT == 2 && pl.NT == 2) return launch_gp<2, 2>(a, grid, pl.lds_bytes, s);
    if (pl.MT == 2 && pl.NT == 1) return launch_gp<2, 1>(a, grid, pl.lds_bytes, s);
    if (pl.MT == 1 && pl.NT == 2) return launch_gp<1, 2>(a, grid, pl.lds_bytes, s);
    if (pl.MT == 1 && pl.NT == 1) return launch_gp<1, 1>(a, grid, pl.lds_bytes, s);
    set_error("gconv_bf16p: no kernel for tile %dx%d", pl.MT, pl.NT);
    return RD_EINVAL;
}

}  // namespace rd
