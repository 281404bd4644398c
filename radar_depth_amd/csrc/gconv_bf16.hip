// bf16-operand form of the generalised convolution (gconv.hip): the same descriptor (phases x taps over an NHWC halo patch),
// the same fp32 tensors in HBM and the same fp32 accumulation / epilogue, but the LDS patch and the weight operand are bf16
// and the reduction runs on v_mfma_f32_32x32x16_bf16 (16 input channels per instruction, 16x the fp32 MFMA rate).
// Configs 3 / 5 of BASELINE.json (bf16) -- opt-in, the fp32 kernels of gconv.hip stay the default and the parity reference.
//
//   * A operand: LDS patch [pixel][CKP + 8] bf16 (pixel pitch 80 / 144 B: the 16 lanes of one b128 read pass fall into 16
//     different bank groups); activations are converted fp32 -> bf16 (round to nearest even) while they are staged.
//   * B operand: packed weights [slab][Cin/8][ldw][8] bf16 (rd_pack_weights_batched, quad == 2), staged as a straight copy;
//     a lane's 8 consecutive input channels of one output channel are one 16-byte LDS read.
//   * lane (l31, hh) feeds channels kstep*16 + hh*8 .. +8 of pixel / output channel l31: one b128 read per fragment.
//   * C/D layout is that of the 32x32 fp32 MFMA, so the epilogue is gconv.hip's.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>

#include "common.h"

namespace rd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 32 bytes (8 fp32 channels) through a raw buffer descriptor: offsets at or beyond num_records return zeros, so padding / halo
// units need no branch (a branch per unit makes the compiler wait for each load before it issues the next)
constexpr unsigned RD_OOB = 0x80000000u;
__device__ __forceinline__ void buf_load8(__amdgpu_buffer_rsrc_t r, unsigned off, float4& v0, float4& v1) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off + 16, 0, 0);
    v0 = __builtin_bit_cast(float4, a);
    v1 = __builtin_bit_cast(float4, b);
}

struct GconvBfArgs {
    RdConvDesc d;
    const void* in;               // NHWC activations: fp32, or bf16 when the kernel is instantiated with IO16
    const unsigned short* w;      // packed bf16 operand
    void* out;
    const void* addend;
    const float* bias;
    float* stat;
    int act, act_cols, ld_add, ldw;
    int TH, TW, PP, CKP, tiles_total, n_cotiles, taps_max;
    int vec4;                     // out / addend / bias allow 4-channel accesses (alignment, strides, Cout, act_cols all multiples of 4)
    int tapoff[RD_MAX_PHASES][RD_MAX_TAPS];   // byte offset of tap t inside the patch
    unsigned long long* trace;                // diagnostics (RD_GCONV_BF16_TRACE=1): 32 cycle-counter stamps per workgroup
};

__device__ __forceinline__ bf16x4 cvt4(const float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}

// IO16: the activation tensors (in, out, addend) are stored as bf16 in HBM (bf16-storage plans): the patch is a straight 16-byte
// copy per 8-channel unit (half the bytes, no conversion), the epilogue rounds to nearest-even on store; the BatchNorm partial
// sums are still taken from the fp32 accumulators.
template <int MT, int NT, bool PIPE, bool IO16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gconv_bf16_kernel(const GconvBfArgs a) {
    typedef typename std::conditional<IO16, bf16s, float>::type io_t;
    constexpr int ESZ = IO16 ? 2 : 4;
    constexpr int WM = 4;
    constexpr int BM = WM * MT * 32;
    constexpr int BN = NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;

    int n_stamp = 0;
#define RD_STAMP() \
    if (a.trace && tid == 0 && n_stamp < 30) a.trace[(size_t)blockIdx.x * 32 + 1 + n_stamp++] = __builtin_readcyclecounter();
    RD_STAMP()
    const unsigned long long rt0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;   // 100 MHz constant clock
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = vid % a.n_cotiles;
    const int pt = vid / a.n_cotiles;
    const int n = pt / a.tiles_total;
    const int tt = pt - n * a.tiles_total;
    int ph_ = 0;
    for (int i = 1; i < D.n_phases; ++i)
        if (tt >= D.phase[i].tile_begin) ph_ = i;
    // (wave-uniform by construction; said explicitly so that everything indexed by it -- tap offsets, tap counts, loop bounds --
    //  goes through the scalar unit: a vector load of a tap offset would put a vmcnt(0) wait, i.e. a wait for the prefetched
    //  next chunk, in front of every MFMA walk)
    const int ph = __builtin_amdgcn_readfirstlane(ph_);
    const RdPhase& P = D.phase[ph];
    const int tloc = tt - P.tile_begin;
    const int tiles_w = (P.lw + a.TW - 1) / a.TW;
    const int r0 = (tloc / tiles_w) * a.TH, c0 = (tloc % tiles_w) * a.TW;
    const int th_n = min(a.TH, P.lh - r0), tw_n = min(a.TW, P.lw - c0);
    const int IS = D.in_stride, OS = D.out_stride;
    const int PW = (a.TW - 1) * IS + (P.dw_max - P.dw_min) + 1;
    const int PH = (th_n - 1) * IS + (P.dh_max - P.dh_min) + 1;
    const int ih0 = r0 * IS + P.dh_min, iw0 = c0 * IS + P.dw_min;
    const int CKP = a.CKP;
    const int PSB = (CKP + 8) * 2;           // patch pixel pitch in bytes
    const int ntaps = __builtin_amdgcn_readfirstlane(P.n_taps);
    const int co0 = cot * BN;

    // LDS carve-up
    int* s_opix = reinterpret_cast<int*>(smem);          // [BM] output pixel index or -1
    int* s_apix = s_opix + BM;                           // [BM] patch pixel index of tap (0,0)
    int* s_widx = s_apix + BM;                           // [32] weight slab index of each tap
    char* s_w = reinterpret_cast<char*>(s_widx + 32);    // [taps][CKP/8][BN] x 16 B (two of them when pipelined)
    const int slab_bytes = a.taps_max * (CKP >> 3) * BN * 16;
    char* s_patch = s_w + (PIPE ? 2 : 1) * slab_bytes;   // [PP][PSB]

    for (int m = tid; m < BM; m += 256) {
        const int r = m / a.TW, c = m - r * a.TW;
        const bool ok = (r < th_n) && (c < tw_n);
        s_opix[m] = ok ? ((n * D.Ho + (r0 + r) * OS + P.out_off_h) * D.Wo + (c0 + c) * OS + P.out_off_w) : -1;
        s_apix[m] = ok ? ((r * IS) * PW + c * IS) : 0;
    }
    if (tid < ntaps) s_widx[tid] = P.widx[tid];
    rd_sync();
    RD_STAMP()

    int aoffB[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aoffB[mt] = s_apix[(wm * MT + mt) * 32 + l31] * PSB + hh * 16;
    const int boffB = (hh * BN + l31) * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    const int q8 = CKP >> 3;                  // 8-channel (32-byte fp32 / 16-byte bf16) units per patch pixel: 2, 4 or 8
    const int lq8 = 31 - __clz(q8);
    const int welems = ntaps << (lq8 + (NT == 2 ? 6 : 5));   // 16-byte units of the weight slab [tap][q8][BN]
    const int cin8 = D.Cin >> 3;
    const char* in_n = static_cast<const char*>(a.in) + (size_t)n * D.Hi * D.Wi * D.ldi * ESZ;
    const int lks = lq8 - 1;                  // log2 of the 16-channel MFMA steps per tap
    const int nsteps = ntaps << lks;

    // ---- (tap, 16-channel step) walk over one staged chunk.  Two fragment sets alternate: the reads of step s+1 are issued
    // before the MFMAs of step s.  The tap offsets live in the lanes of one VGPR and are fetched with v_readlane (an s_load
    // in this loop would force lgkmcnt(0) waits, i.e. serialize the LDS reads behind it).
    const int tapv = a.tapoff[ph][min(lane, RD_MAX_TAPS - 1)];
    auto run_chunk = [&](const char* wbuf) {
        auto load = [&](int s, bf16x8 (&A)[MT], bf16x8 (&B)[NT]) {
            const int t = s >> lks, k = s & ((1 << lks) - 1);
            const int ao = __builtin_amdgcn_readlane(tapv, t) + k * 32;
            const int bo = (((t << lq8) + k * 2) * BN) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) A[mt] = *reinterpret_cast<const bf16x8*>(s_patch + aoffB[mt] + ao);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) B[nt] = *reinterpret_cast<const bf16x8*>(wbuf + boffB + bo + nt * 512);
        };
        auto mma = [&](const bf16x8 (&A)[MT], const bf16x8 (&B)[NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt], B[nt], acc[mt][nt], 0, 0, 0);
        };
        // Fragment ring of DEPTH steps, statically indexed (the step loop is unrolled by DEPTH): the reads of step s + DEPTH - 1
        // are issued before the MFMAs of step s, unconditionally (past the end they re-read the last step), so the body is
        // straight-line code and the compiler can wait for exactly the oldest outstanding read instead of lgkmcnt(0).
        // The main loop has NO conditional inside and its order is pinned: with an `if (s < nsteps)` around the MFMAs the wait-count
        // pass saw a merge of differently loaded paths and emitted s_waitcnt lgkmcnt(0) -- a full drain of the reads in flight --
        // once per trip; without the scheduling barriers the scheduler sinks the reads below the next step's MFMAs to save
        // registers and waits for them immediately.  Now every MFMA group waits with lgkmcnt(2 steps of reads).  The remainder
        // (nsteps mod DEPTH steps, already loaded) follows the loop.
        constexpr int DEPTH = 3;                 // (four sets measured the same; the 3x2 tile has 254 registers with three)
        bf16x8 fa[DEPTH][MT], fb[DEPTH][NT];
        const int last = nsteps - 1;
#pragma unroll
        for (int j = 0; j < DEPTH - 1; ++j) load(min(j, last), fa[j], fb[j]);
        int s0 = 0;
        for (; s0 + DEPTH <= nsteps; s0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                load(min(s0 + j + DEPTH - 1, last), fa[(j + DEPTH - 1) % DEPTH], fb[(j + DEPTH - 1) % DEPTH]);
                __builtin_amdgcn_sched_barrier(0);
                mma(fa[j], fb[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s0 < nsteps) mma(fa[0], fb[0]);
        if (s0 + 1 < nsteps) mma(fa[1], fb[1]);
    };

    // ---- staging.  These layers are bound by memory and by the integer work of the staging itself (the bf16 MFMAs of a chunk
    // take a few hundred clocks), so the element -> address maps avoid every division by a runtime value:
    //   patch  : a wave copies whole patch ROW SEGMENTS (64 consecutive units of one row); segment = wave + 4*k is wave-uniform
    //            (scalar unit), a lane's pixel / channel-unit inside it are shifts of the lane id;
    //   weights: unit e of the slab [tap][q8][BN] decomposes by shifts (BN and q8 are powers of two) and lands in LDS by
    //            global_load_lds -- the packed operand in HBM already has the LDS layout.
    const int wave_u = __builtin_amdgcn_readfirstlane(wm);
    const int rowu = PW << lq8;               // units per patch row
    const int nseg = (rowu + 63) >> 6;        // 64-unit segments per row
    const int nsegs = PH * nseg;
    auto unit_of = [&](int k, unsigned& goff, int& ldst) {
        const int seg = wave_u + 4 * k;
        const int row = nseg == 1 ? seg : seg / nseg;
        const int cu = ((seg - row * nseg) << 6) + lane;
        const int px = cu >> lq8, qq = cu & (q8 - 1);
        const int ih = ih0 + row, iw = iw0 + px;
        const bool ok = seg < nsegs && cu < rowu;
        ldst = ok ? (row * PW + px) * PSB + qq * 16 : -1;
        goff = (ok && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi) ? (unsigned)(((ih * D.Wi + iw) * D.ldi + qq * 8) * ESZ) : RD_OOB;
    };
    const unsigned img_bytes = (unsigned)(D.Hi * D.Wi * D.ldi) * (unsigned)ESZ;
    auto chunk_rsrc = [&](int cb) {           // image n from channel cb on
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in_n + cb * ESZ), 0, img_bytes - cb * ESZ, 0x00020000);
    };
    // one 8-channel unit in flight: two float4 (fp32 storage) or 16 raw bytes in v0 (bf16 storage; v1 unused and dropped)
    auto load_unit = [&](__amdgpu_buffer_rsrc_t r, unsigned off, float4& v0, float4& v1) {
        if constexpr (IO16) v0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
        else buf_load8(r, off, v0, v1);
    };
    auto weight_off = [&](int e) -> unsigned {      // byte offset of slab unit e inside the packed weights at input-channel group 0
        const int j = e & (BN - 1), tk = e >> (NT == 2 ? 6 : 5);
        const int k8 = tk & (q8 - 1), t = tk >> lq8;
        return (e < welems && co0 + j < D.Cout) ? (unsigned)((((unsigned)s_widx[min(t, ntaps - 1)] * cin8 + k8) * a.ldw + co0 + j) * 16) : ~0u;
    };
    auto put_unit = [&](int ldst, const float4 v0, const float4 v1) {
        if constexpr (IO16) {
            if (ldst >= 0) *reinterpret_cast<float4*>(s_patch + ldst) = v0;
        } else if (ldst >= 0) {
            bf16x8 r;
            const bf16x4 lo = cvt4(v0), hi = cvt4(v1);
            r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
            r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
            *reinterpret_cast<bf16x8*>(s_patch + ldst) = r;
        }
    };

    if constexpr (PIPE) {
        // ---- software-pipelined chunk loop (host guarantees: the patch is at most 4*UPP row segments, a weight slab at most UWP
        // units per thread).  The weight slabs alternate between two LDS buffers and arrive one chunk ahead; the next chunk's
        // activations are fetched into registers before the current chunk's MFMAs and converted / written to LDS after them.
        // Per-unit addresses are computed once per workgroup.
        constexpr int UPP = MT * NT >= 6 ? 4 : 8, UWP = 9;   // (the 3x2 register tile has no room for eight staged units)
        unsigned pgo[UPP];      // byte offset of the unit inside the image at channel 0; RD_OOB: outside -> zero
        int pdst[UPP];          // LDS byte offset inside the patch, -1: no such unit
#pragma unroll
        for (int u = 0; u < UPP; ++u) unit_of(u, pgo[u], pdst[u]);
        unsigned woff[UWP];
#pragma unroll
        for (int u = 0; u < UWP; ++u) woff[u] = weight_off(tid + u * 256);
        auto issue_slab = [&](int buf, int cb) {
            const char* src = reinterpret_cast<const char*>(a.w) + (size_t)(cb >> 3) * a.ldw * 16;
            char* dst = s_w + buf * slab_bytes;
#pragma unroll
            for (int u = 0; u < UWP; ++u) {
                const int e = tid + u * 256;
                if (e < welems) {
                    if (woff[u] != ~0u) glds16(reinterpret_cast<const float*>(src + woff[u]), reinterpret_cast<float*>(dst + (e - lane) * 16));
                    else *reinterpret_cast<uint4*>(dst + e * 16) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        };
        auto patch_fetch = [&](int cb, float4 (&v0)[UPP], float4 (&v1)[UPP]) {
            const __amdgpu_buffer_rsrc_t r = chunk_rsrc(cb);
#pragma unroll
            for (int u = 0; u < UPP; ++u) load_unit(r, pgo[u], v0[u], v1[u]);
        };
        {
            float4 v0[UPP], v1[UPP];
            issue_slab(0, 0);
            patch_fetch(0, v0, v1);
#pragma unroll
            for (int u = 0; u < UPP; ++u) put_unit(pdst[u], v0[u], v1[u]);
        }
        int idx = 0;
        for (int cb = 0; cb < D.Cin; cb += CKP, ++idx) {
            glds_wait();
            rd_sync();          // slab idx and the chunk's patch have landed; slab idx-1 is fully consumed
            RD_STAMP()
            const bool more = cb + CKP < D.Cin;
            float4 v0[UPP], v1[UPP];
            if (more) {
                issue_slab((idx + 1) & 1, cb + CKP);
                patch_fetch(cb + CKP, v0, v1);
            }
            run_chunk(s_w + (idx & 1) * slab_bytes);
            RD_STAMP()
            if (more) {
                rd_sync();      // every wave is done reading this chunk's patch
#pragma unroll
                for (int u = 0; u < UPP; ++u) put_unit(pdst[u], v0[u], v1[u]);
            }
        }
    } else {
        // ---- plain chunk loop: weights by global_load_lds, then the patch in batches of UP row segments per wave (all loads of
        // a batch are issued before its first conversion), one barrier pair per chunk.  Workgroups of the same CU overlap each
        // other's phases.
        constexpr int UP = 8;
        const int nk = (nsegs + 3) >> 2;            // row segments per wave
        for (int cb = 0; cb < D.Cin; cb += CKP) {
            rd_sync();
            {
                const char* src = reinterpret_cast<const char*>(a.w) + (size_t)(cb >> 3) * a.ldw * 16;
                for (int e = tid; e < welems; e += 256) {
                    const unsigned wo = weight_off(e);
                    if (wo != ~0u) glds16(reinterpret_cast<const float*>(src + wo), reinterpret_cast<float*>(s_w + (e - lane) * 16));
                    else *reinterpret_cast<uint4*>(s_w + e * 16) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            const __amdgpu_buffer_rsrc_t r = chunk_rsrc(cb);
            for (int k0 = 0; k0 < nk; k0 += UP) {
                float4 v0[UP], v1[UP];
                int ld[UP];
#pragma unroll
                for (int u = 0; u < UP; ++u) {
                    unsigned go;
                    unit_of(k0 + u, go, ld[u]);
                    load_unit(r, go, v0[u], v1[u]);
                }
#pragma unroll
                for (int u = 0; u < UP; ++u) put_unit(ld[u], v0[u], v1[u]);
            }
            glds_wait();
            rd_sync();
            RD_STAMP()
            run_chunk(s_w);
            RD_STAMP()
        }
    }

    // ---- epilogue (gconv.hip's: C/D layout of the 32x32 MFMA)
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    const bool has_add = a.addend != nullptr;
    const bool has_bias = a.bias != nullptr;
    const int cob = co0 + l31;
    float biasv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) biasv[nt] = (has_bias && cob + nt * 32 < D.Cout) ? a.bias[cob + nt * 32] : 0.f;
    const bool want_stat = a.stat != nullptr;
    // Row-major stores: in the MFMA's C/D layout a lane holds ONE output channel of 16 pixels, so a plain epilogue stores (and
    // reads the addend) one element per lane and instruction -- 96 two-byte stores per lane for the 3x2 tile, which took 25 % of a
    // workgroup's lifetime on the 64-channel layers (13.6 k of 54 k clocks, tools/trace_gconv_bf16.py).  Each 4x4 block
    // (4 accumulator registers x the 4 lanes of a quad = 4 pixels x 4 channels) is transposed in registers with two DPP
    // exchanges, after which a lane holds FOUR consecutive channels of one pixel: 8-byte (bf16) / 16-byte (fp32) accesses, a
    // quarter of the instructions.  Needs 4-channel alignment of every pointer and stride (a.vec4, checked by the host).
    const int q4l = l31 & 3, k4l = l31 >> 2;
    const bool odd1 = q4l & 1, odd2 = q4l & 2;
    float4 ssum4[NT], ssq4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum4[nt] = ssq4[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any4 = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int ro4[4];
        bool rows_ok = true;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ro4[g] = s_opix[(wm * MT + mt) * 32 + q4l + 8 * g + 4 * hh];   // this lane's pixel of register group g after the transposition
            rows_ok = rows_ok && ro4[g] >= 0;
        }
        if (a.vec4 && __all(rows_ok)) {
            // full M-tile (almost all of them): no exec masking; the addends are gathered before their first use
            any4 = true;
            const int cq = co0 + 4 * k4l;
            float4 addv[NT][4];
            if (has_add) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const io_t* ap = static_cast<const io_t*>(a.addend) + (size_t)ro4[g] * a.ld_add + cq;
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
                    if (has_add) { v.x += addv[nt][g].x; v.y += addv[nt][g].y; v.z += addv[nt][g].z; v.w += addv[nt][g].w; }
                    const int cc = cq + nt * 32;
                    if (cc < a.act_cols) {          // (act_cols is a multiple of 4 whenever vec4 is set)
                        v.x = act_fwd(v.x, a.act); v.y = act_fwd(v.y, a.act); v.z = act_fwd(v.z, a.act); v.w = act_fwd(v.w, a.act);
                    }
                    if (cok4) st4(static_cast<io_t*>(a.out) + (size_t)ro4[g] * D.ldo + cc, v);
                    if (want_stat) {
                        ssum4[nt].x += v.x; ssum4[nt].y += v.y; ssum4[nt].z += v.z; ssum4[nt].w += v.w;
                        ssq4[nt].x += v.x * v.x; ssq4[nt].y += v.y * v.y; ssq4[nt].z += v.z * v.z; ssq4[nt].w += v.w * v.w;
                    }
                }
            }
        } else {
            int ro[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) ro[i] = s_opix[(wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];   // -1: no such pixel
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = cob + nt * 32;
                const bool cok = co < D.Cout;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (cok && ro[i] >= 0) {
                        float v = acc[mt][nt][i] + biasv[nt];
                        if (has_add) v += ld1(static_cast<const io_t*>(a.addend) + (size_t)ro[i] * a.ld_add + co);
                        if (co < a.act_cols) v = act_fwd(v, a.act);
                        st1(static_cast<io_t*>(a.out) + (size_t)ro[i] * D.ldo + co, v);
                        ssum[nt] += v;
                        ssq[nt] += v * v;
                    }
                }
            }
        }
    }
    if (want_stat && __any(any4)) {
        // back to one channel per lane: sum the four pixels of the quad (every lane of it then holds the quad's four channel sums),
        // lane q keeps channel q
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
    if (a.stat) {
        rd_sync();
        float* red = reinterpret_cast<float*>(s_w);  // [WM][2][BN]
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
            for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + j];
            const int co = co0 + j;
            if (co < D.Cout) a.stat[((size_t)pt * 2 + which) * D.Cout + co] = s;
        }
    }
    RD_STAMP()
    if (a.trace && tid == 0) {
        a.trace[(size_t)blockIdx.x * 32] = n_stamp;
        a.trace[(size_t)blockIdx.x * 32 + 31] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#undef RD_STAMP
}

// ------------------------------------------------------------------------------------------ host
struct GconvBfPlan {
    int MT, NT, CKP, TH, TW, PP, tiles_total, n_cotiles, taps_max;
    size_t lds_bytes;
    int pipe;
};

static int bf_patch_pixels(const RdConvDesc& d, const RdPhase& p, int TH, int TW) {
    const int th = TH < p.lh ? TH : p.lh;
    const int PH = (th - 1) * d.in_stride + (p.dh_max - p.dh_min) + 1;
    const int PW = (TW - 1) * d.in_stride + (p.dw_max - p.dw_min) + 1;
    return PH * PW;
}

static bool plan_gconv_bf16(const RdConvDesc& d, GconvBfPlan& best) {
    struct Cfg { int MT, NT; };
    static const Cfg cfgs[] = {{2, 2}, {2, 1}, {3, 2}, {1, 2}, {1, 1}};
    int taps_max = 0;
    for (int i = 0; i < d.n_phases; ++i) taps_max = taps_max > d.phase[i].n_taps ? taps_max : d.phase[i].n_taps;
    int pr = 0;
    for (int i = 1; i < d.n_phases; ++i)
        if ((int64_t)d.phase[i].lh * d.phase[i].lw > (int64_t)d.phase[pr].lh * d.phase[pr].lw) pr = i;
    const RdPhase& P = d.phase[pr];
    double best_score = -1;
    static const char* force = getenv("RD_GCONV_BF16_FORCE");   // diagnostics: index into cfgs
    static const char* force_ckp = getenv("RD_GCONV_BF16_CKP");
    int cfg_i = -1;
    for (const Cfg& c : cfgs) {
        ++cfg_i;
        if (force && atoi(force) != cfg_i) continue;
        const int BM = 4 * c.MT * 32, BN = c.NT * 32;
        const int n_cot = cdiv(d.Cout, BN);
        const double n_util = (double)d.Cout / (n_cot * BN);
        for (int ckp = 64; ckp >= 16; ckp >>= 1) {
            if (d.Cin % ckp != 0) continue;
            if (force_ckp && atoi(force_ckp) != ckp && d.Cin % atoi(force_ckp) == 0) continue;
            const size_t wbytes1 = (size_t)taps_max * ckp * BN * 2;
            if (wbytes1 > 72 * 1024) continue;
            for (int twt = 1; twt <= cdiv(P.lw, 4); ++twt) {
                const int TW = cdiv(P.lw, twt);
                if (TW > BM) continue;
                int TH = BM / TW;
                if (TH > P.lh) TH = P.lh;
                TH = cdiv(P.lh, cdiv(P.lh, TH));
                int PP = 0;
                for (int i = 0; i < d.n_phases; ++i) {
                    const int pp = bf_patch_pixels(d, d.phase[i], TH, TW);
                    PP = PP > pp ? PP : pp;
                }
                // pipelined chunk loop: the patch chunk must be one batch of <= 8 loads per thread, a weight slab at most nine
                static const char* nopipe = getenv("RD_GCONV_BF16_NOPIPE");
                int segs = 0;       // 64-unit row segments of the largest patch (what one wave quarter copies)
                for (int i = 0; i < d.n_phases; ++i) {
                    const RdPhase& q = d.phase[i];
                    const int th = TH < q.lh ? TH : q.lh;
                    const int ph_ = (th - 1) * d.in_stride + (q.dh_max - q.dh_min) + 1, pw_ = (TW - 1) * d.in_stride + (q.dw_max - q.dw_min) + 1;
                    const int sg = ph_ * cdiv(pw_ * (ckp / 8), 64);
                    segs = segs > sg ? segs : sg;
                }
                // the pipelined loop pays a longer prologue and more registers: worth it for long reductions (tools/sweep_gconv_bf16.py)
                const bool pipe = !nopipe && d.Cin >= 320 && segs <= 4 * (c.MT * c.NT >= 6 ? 4 : 8) && taps_max * (ckp / 8) * BN <= 9 * 256;
                const size_t wbytes = wbytes1 * (pipe ? 2 : 1);
                const size_t lds = (size_t)(2 * BM + 32) * 4 + wbytes + (size_t)(PP + 1) * (ckp + 8) * 2 + 64;
                if (lds > 160 * 1024 - 512) continue;
                // Cost model (fits the sweeps to ~15 %): with bf16 MFMAs every layer of this network is bound by the bytes its
                // workgroups pull through L2 -- each one reads its halo patch over all input channels plus the weight slab of
                // its output-channel tile -- at ~6 TB/s aggregate, plus a fixed per-workgroup cost (prologue, first-load latency,
                // epilogue drain ~ 48 KB worth of transfer time).  Fewer than two workgroups per CU leave latencies uncovered.
                double taps_avg = 0;
                for (int i = 0; i < d.n_phases; ++i) taps_avg += d.phase[i].n_taps;
                taps_avg /= d.n_phases;
                const double wgs = (double)d.N * cdiv(P.lh, TH) * cdiv(P.lw, TW) * n_cot * d.n_phases;
                const double ncu = (double)num_cus();
                const double per_wg = (double)PP * d.Cin * 4.0 * (ckp == 16 && d.Cin >= 32 ? 1.08 : 1.0) + taps_avg * d.Cin * BN * 2.0 + 48.0 * 1024;
                double cost = wgs * per_wg + (double)d.N * d.Ho * d.Wo * d.Cout * 4.0;
                const int occ_regs = pipe ? 2 : (c.MT * c.NT >= 4 ? 2 : (c.MT * c.NT == 2 ? 3 : 4));
                int occ = (int)((160 * 1024) / lds);
                occ = occ < 1 ? 1 : (occ > occ_regs ? occ_regs : occ);
                const double slots = ncu * occ;
                cost *= ceil(wgs / slots) * slots / wgs;             // tail imbalance
                if (wgs < 2 * ncu) cost *= 2 * ncu / wgs;            // latency-bound: too few workgroups in flight
                cost *= 1.0 + 0.5 / occ;                             // more resident workgroups hide more of each other's phases
                cost /= sqrt(n_util);                                // output-channel tiles wider than Cout: wasted LDS reads / epilogue
                if (!pipe && d.Cin >= 320) cost *= 1.3;              // serialized chunk loop over a long reduction
                const double score = 1e12 / cost;
                if (score > best_score) {
                    best_score = score;
                    best = GconvBfPlan{c.MT, c.NT, ckp, TH, TW, PP, 0, n_cot, taps_max, lds, pipe ? 1 : 0};
                }
            }
        }
    }
    return best_score > 0;
}

template <int MT, int NT, bool PIPE, bool IO16>
static int launch_bf(const GconvBfArgs& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = gconv_bf16_kernel<MT, NT, PIPE, IO16>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("gconv_bf16_kernel");
    return RD_OK;
}

static int bf_plan_query(const RdConvDesc* d, GconvBfPlan& pl, RdConvDesc& dd) {
    struct Entry { GconvBfPlan pl; RdConvDesc dd; };
    static std::mutex mu;
    static std::unordered_map<std::string, Entry> cache;
    RD_CHECK_ARG(d != nullptr, "gconv_bf16: null descriptor");
    std::string key(reinterpret_cast<const char*>(d), sizeof(RdConvDesc));
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) { pl = it->second.pl; dd = it->second.dd; return RD_OK; }
    }
    RD_CHECK_ARG(d->n_phases >= 1 && d->n_phases <= RD_MAX_PHASES, "gconv_bf16: n_phases=%d", d->n_phases);
    RD_CHECK_ARG(d->Cin % 16 == 0 && d->ldi % 4 == 0, "gconv_bf16: Cin=%d must be a multiple of 16, ldi=%d of 4", d->Cin, d->ldi);
    RD_CHECK_ARG(d->in_stride >= 1 && d->in_stride <= 2 && d->out_stride >= 1 && d->out_stride <= 2, "gconv_bf16: strides");
    RD_CHECK_ARG((int64_t)d->Hi * d->Wi * d->ldi * 4 < (int64_t)RD_OOB, "gconv_bf16: one input image must stay below 2 GiB (32-bit buffer offsets)");
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        RD_CHECK_ARG(p.n_taps >= 1 && p.n_taps <= RD_MAX_TAPS, "gconv_bf16: phase %d has %d taps", i, p.n_taps);
        RD_CHECK_ARG(p.lh >= 1 && p.lw >= 1, "gconv_bf16: empty phase %d", i);
        for (int t = 0; t < p.n_taps; ++t)
            RD_CHECK_ARG(p.dh[t] >= p.dh_min && p.dh[t] <= p.dh_max && p.dw[t] >= p.dw_min && p.dw[t] <= p.dw_max,
                         "gconv_bf16: tap %d of phase %d outside its declared range", t, i);
    }
    dd = *d;
    if (!plan_gconv_bf16(dd, pl)) { set_error("gconv_bf16: no feasible tiling"); return RD_EINVAL; }
    int tb = 0;
    for (int i = 0; i < dd.n_phases; ++i) {
        dd.phase[i].tile_begin = tb;
        tb += cdiv(dd.phase[i].lh, pl.TH) * cdiv(dd.phase[i].lw, pl.TW);
    }
    pl.tiles_total = tb;
    std::lock_guard<std::mutex> lk(mu);
    cache.emplace(std::move(key), Entry{pl, dd});
    return RD_OK;
}

}  // namespace rd

using namespace rd;

static unsigned long long* g_bf_trace = nullptr;
// diagnostics: copy the stamps of the last traced launch (32 slots per workgroup: count, then cycle-counter values)
extern "C" int rd_gconv_bf16_trace_read(unsigned long long* host, int n_wg) {
    if (!g_bf_trace) return RD_EINVAL;
    RD_CHECK_HIP(hipMemcpy(host, g_bf_trace, (size_t)n_wg * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RD_OK;
}

extern "C" int rd_gconv_bf16_plan_info(const RdConvDesc* d, int32_t* out) {
    GconvBfPlan pl; RdConvDesc dd;
    int rc = bf_plan_query(d, pl, dd);
    if (rc != RD_OK) return rc;
    const int v[8] = {pl.MT, pl.NT, pl.pipe * 1000 + pl.CKP, pl.TH, pl.TW, pl.PP, (int)pl.lds_bytes, d->N * pl.tiles_total * pl.n_cotiles};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RD_OK;
}

extern "C" int rd_gconv_bf16_stat_tiles(const RdConvDesc* d) {
    GconvBfPlan pl; RdConvDesc dd;
    if (bf_plan_query(d, pl, dd) != RD_OK) return RD_EINVAL;
    return d->N * pl.tiles_total;
}

// storage-typed forms of the two queries: with bf16 tensors the descriptor may be served by gconv_bf16p.hip, whose tiling differs
extern "C" int rd_gconv_bf16_stat_tiles_t(int32_t dtype, const RdConvDesc* d) {
    if (dtype == RD_DTYPE_BF16 && gconv_bf16p_supported(d) == 1) return gconv_bf16p_stat_tiles(d);
    return rd_gconv_bf16_stat_tiles(d);
}

extern "C" int rd_gconv_bf16p_plan_all(int32_t on) { return gconv_bf16p_plan_all(on); }

extern "C" int rd_gconv_bf16_plan_info_t(int32_t dtype, const RdConvDesc* d, int32_t* out) {
    if (dtype == RD_DTYPE_BF16 && gconv_bf16p_supported(d) == 1) {
        const int rc = gconv_bf16p_plan_info(d, out);
        if (rc == RD_OK) out[2] += 2000;          // (marks the persistent kernel in the plan strings of the tools)
        return rc;
    }
    return rd_gconv_bf16_plan_info(d, out);
}

static int gconv_bf16_impl(bool io16, const RdConvDesc* d, const void* in, const void* w_packed_bf16, void* out, const float* bias,
                           int32_t act, int32_t act_cols, const void* addend, int32_t ld_add, float* stat_partial, void* stream) {
    RD_CHECK_ARG(in && w_packed_bf16 && out, "gconv_bf16: null tensor");
    RD_CHECK_ARG(!io16 || (d && d->ldi % 8 == 0), "gconv_bf16: bf16 storage needs the input channel stride to be a multiple of 8");
    // bf16 storage, unit-stride input, 4..9 taps per phase: the persistent pipelined kernel (gconv_bf16p.hip; its statistics tiling is
    // its own: rd_gconv_bf16_stat_tiles_t)
    if (io16 && gconv_bf16p_supported(d) == 1)
        return launch_gconv_bf16p(d, in, w_packed_bf16, out, bias, act, act_cols, addend, ld_add, stat_partial, static_cast<hipStream_t>(stream));
    GconvBfArgs a;
    GconvBfPlan pl;
    int rc = bf_plan_query(d, pl, a.d);
    if (rc != RD_OK) return rc;
    a.in = in; a.w = static_cast<const unsigned short*>(w_packed_bf16); a.out = out;
    a.addend = addend; a.bias = bias; a.stat = stat_partial;
    a.act = act; a.act_cols = act_cols; a.ld_add = ld_add; a.ldw = d->Cout;
    a.TH = pl.TH; a.TW = pl.TW; a.PP = pl.PP; a.CKP = pl.CKP;
    a.tiles_total = pl.tiles_total; a.n_cotiles = pl.n_cotiles; a.taps_max = pl.taps_max;
    {
        const uintptr_t al = io16 ? 8 : 16;     // four channels
        static const char* novec = getenv("RD_GCONV_BF16_NOVEC4");   // diagnostics: element-wise epilogue
        a.vec4 = !novec && d->Cout % 4 == 0 && d->ldo % 4 == 0 && reinterpret_cast<uintptr_t>(out) % al == 0 && act_cols % 4 == 0 &&
                 (!addend || (ld_add % 4 == 0 && reinterpret_cast<uintptr_t>(addend) % al == 0)) &&
                 (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    }
    const int PSB = (pl.CKP + 8) * 2;
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        const int PW_ = (pl.TW - 1) * d->in_stride + (p.dw_max - p.dw_min) + 1;
        for (int t = 0; t < p.n_taps; ++t) a.tapoff[i][t] = ((p.dh[t] - p.dh_min) * PW_ + (p.dw[t] - p.dw_min)) * PSB;
    }
    const int grid = d->N * pl.tiles_total * pl.n_cotiles;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.trace = nullptr;
    {
        static const char* tr = getenv("RD_GCONV_BF16_TRACE");
        if (tr && atoi(tr)) {
            if (!g_bf_trace) RD_CHECK_HIP(hipMalloc(&g_bf_trace, (size_t)65536 * 32 * sizeof(unsigned long long)));
            a.trace = g_bf_trace;
        }
    }
#define RD_BF(MT_, NT_)                                                                                                        \
    if (pl.MT == MT_ && pl.NT == NT_)                                                                                          \
        return io16 ? (pl.pipe ? launch_bf<MT_, NT_, true, true>(a, grid, pl.lds_bytes, s)                                     \
                               : launch_bf<MT_, NT_, false, true>(a, grid, pl.lds_bytes, s))                                   \
                    : (pl.pipe ? launch_bf<MT_, NT_, true, false>(a, grid, pl.lds_bytes, s)                                    \
                               : launch_bf<MT_, NT_, false, false>(a, grid, pl.lds_bytes, s));
    RD_BF(2, 2) RD_BF(2, 1) RD_BF(3, 2) RD_BF(1, 2) RD_BF(1, 1)
#undef RD_BF
    set_error("gconv_bf16: no kernel for tile %dx%d", pl.MT, pl.NT);
    return RD_EINVAL;
}

extern "C" int rd_gconv_bf16(const RdConvDesc* d, const float* in, const void* w_packed_bf16, float* out, const float* bias,
                             int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial,
                             void* stream) {
    return gconv_bf16_impl(false, d, in, w_packed_bf16, out, bias, act, act_cols, addend, ld_add, stat_partial, stream);
}
// storage-typed form: dtype = RD_DTYPE_BF16 -> in / out / addend are bf16 NHWC tensors (strides in elements)
extern "C" int rd_gconv_bf16_t(int32_t dtype, const RdConvDesc* d, const void* in, const void* w_packed_bf16, void* out,
                               const float* bias, int32_t act, int32_t act_cols, const void* addend, int32_t ld_add,
                               float* stat_partial, void* stream) {
    RD_CHECK_ARG(dtype == RD_DTYPE_F32 || dtype == RD_DTYPE_BF16, "gconv_bf16_t: bad dtype %d", dtype);
    return gconv_bf16_impl(dtype == RD_DTYPE_BF16, d, in, w_packed_bf16, out, bias, act, act_cols, addend, ld_add, stat_partial, stream);
}
