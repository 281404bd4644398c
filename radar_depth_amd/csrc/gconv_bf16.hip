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
#include <unordered_map>

#include "common.h"

namespace rd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct GconvBfArgs {
    RdConvDesc d;
    const float* in;
    const unsigned short* w;      // packed bf16 operand
    float* out;
    const float* addend;
    const float* bias;
    float* stat;
    int act, act_cols, ld_add, ldw;
    int TH, TW, PP, CKP, tiles_total, n_cotiles, taps_max;
    int tapoff[RD_MAX_PHASES][RD_MAX_TAPS];   // byte offset of tap t inside the patch
};

__device__ __forceinline__ bf16x4 cvt4(const float4 v) {
    bf16x4 r;
    r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    return r;
}

template <int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gconv_bf16_kernel(const GconvBfArgs a) {
    constexpr int WM = 4;
    constexpr int BM = WM * MT * 32;
    constexpr int BN = NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
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
    const int PW = (a.TW - 1) * IS + (P.dw_max - P.dw_min) + 1;
    const int PH = (th_n - 1) * IS + (P.dh_max - P.dh_min) + 1;
    const int ih0 = r0 * IS + P.dh_min, iw0 = c0 * IS + P.dw_min;
    const int CKP = a.CKP;
    const int PSB = (CKP + 8) * 2;           // patch pixel pitch in bytes
    const int ntaps = P.n_taps;
    const int co0 = cot * BN;

    // LDS carve-up
    int* s_opix = reinterpret_cast<int*>(smem);          // [BM] output pixel index or -1
    int* s_apix = s_opix + BM;                           // [BM] patch pixel index of tap (0,0)
    int* s_widx = s_apix + BM;                           // [32] weight slab index of each tap
    char* s_w = reinterpret_cast<char*>(s_widx + 32);    // [taps][CKP/8][BN] x 16 B
    char* s_patch = s_w + (size_t)a.taps_max * (CKP >> 3) * BN * 16;   // [PP][PSB]

    for (int m = tid; m < BM; m += 256) {
        const int r = m / a.TW, c = m - r * a.TW;
        const bool ok = (r < th_n) && (c < tw_n);
        s_opix[m] = ok ? ((n * D.Ho + (r0 + r) * OS + P.out_off_h) * D.Wo + (c0 + c) * OS + P.out_off_w) : -1;
        s_apix[m] = ok ? ((r * IS) * PW + c * IS) : 0;
    }
    if (tid < ntaps) s_widx[tid] = P.widx[tid];
    __syncthreads();

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

    const int q8 = CKP >> 3;                  // 8-channel units per patch pixel
    const int patch_elems = PH * PW * q8;
    const int welems = ntaps * q8 * BN;       // 16-byte units of the weight slab
    const int cin8 = D.Cin >> 3;
    const float* in_n = a.in + (size_t)n * D.Hi * D.Wi * D.ldi;
    const int ksteps = CKP >> 4;
    const int nsteps = ntaps * ksteps;

    for (int cb = 0; cb < D.Cin; cb += CKP) {
        __syncthreads();
        // ---- stage the halo patch chunk [PH*PW][CKP] as bf16 (zero outside the image) and the chunk's weight slab.  Batches of
        // loads are issued before their LDS writes so that a batch pays one memory round trip.
        constexpr int UP = 4, UW = 4;
        for (int base = tid; base < patch_elems; base += UP * 256) {
            float4 v0[UP], v1[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int e = base + u * 256;
                const int pix = e / q8, qq = e - pix * q8;
                const int py = pix / PW, px = pix - py * PW;
                const int ih = ih0 + py, iw = iw0 + px;
                v0[u] = v1[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < patch_elems && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi) {
                    const float* p = in_n + ((size_t)ih * D.Wi + iw) * D.ldi + cb + qq * 8;
                    v0[u] = *reinterpret_cast<const float4*>(p);
                    v1[u] = *reinterpret_cast<const float4*>(p + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int e = base + u * 256;
                if (e < patch_elems) {
                    const int pix = e / q8, qq = e - pix * q8;
                    bf16x8 r;
                    const bf16x4 lo = cvt4(v0[u]), hi = cvt4(v1[u]);
                    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
                    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
                    *reinterpret_cast<bf16x8*>(s_patch + pix * PSB + qq * 16) = r;
                }
            }
        }
        for (int base = tid; base < welems; base += UW * 256) {
            uint4 v[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int e = base + u * 256;
                const int j = e % BN, tk = e / BN;
                const int k8 = tk % q8, t = tk / q8;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (e < welems && co0 + j < D.Cout)
                    v[u] = *reinterpret_cast<const uint4*>(a.w + (((size_t)s_widx[t] * cin8 + (cb >> 3) + k8) * a.ldw + co0 + j) * 8);
            }
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int e = base + u * 256;
                if (e < welems) *reinterpret_cast<uint4*>(s_w + (size_t)e * 16) = v[u];
            }
        }
        __syncthreads();

        // ---- (tap, 16-channel step) walk; fragments of step s+1 are read while step s's MFMAs issue
        bf16x8 ca[MT], cbv[NT], na[MT], nb[NT];
        auto load = [&](int s, bf16x8 (&A)[MT], bf16x8 (&B)[NT]) {
            const int t = s / ksteps, k = s - t * ksteps;
            const int ao = a.tapoff[ph][t] + k * 32;
            const int bo = ((t * q8 + k * 2) * BN) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) A[mt] = *reinterpret_cast<const bf16x8*>(s_patch + aoffB[mt] + ao);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) B[nt] = *reinterpret_cast<const bf16x8*>(s_w + boffB + bo + nt * 512);
        };
        load(0, ca, cbv);
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) load(s + 1, na, nb);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[mt], cbv[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ca[mt] = na[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) cbv[nt] = nb[nt];
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
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
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
                    if (has_add) v += a.addend[(size_t)ro[i] * a.ld_add + co];
                    if (co < a.act_cols) v = act_fwd(v, a.act);
                    a.out[(size_t)ro[i] * D.ldo + co] = v;
                    ssum[nt] += v;
                    ssq[nt] += v * v;
                }
            }
        }
    }
    if (a.stat) {
        __syncthreads();
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
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, j = tid - which * BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + j];
            const int co = co0 + j;
            if (co < D.Cout) a.stat[((size_t)pt * 2 + which) * D.Cout + co] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------ host
struct GconvBfPlan {
    int MT, NT, CKP, TH, TW, PP, tiles_total, n_cotiles, taps_max;
    size_t lds_bytes;
};

static int bf_patch_pixels(const RdConvDesc& d, const RdPhase& p, int TH, int TW) {
    const int th = TH < p.lh ? TH : p.lh;
    const int PH = (th - 1) * d.in_stride + (p.dh_max - p.dh_min) + 1;
    const int PW = (TW - 1) * d.in_stride + (p.dw_max - p.dw_min) + 1;
    return PH * PW;
}

static bool plan_gconv_bf16(const RdConvDesc& d, GconvBfPlan& best) {
    struct Cfg { int MT, NT; double prior; };
    static const Cfg cfgs[] = {{2, 2, 0.95}, {2, 1, 0.8}, {3, 2, 1.0}, {1, 2, 0.75}, {1, 1, 0.7}};
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
            const size_t wbytes = (size_t)taps_max * ckp * BN * 2;
            if (wbytes > 72 * 1024) continue;
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
                const size_t lds = (size_t)(2 * BM + 32) * 4 + wbytes + (size_t)(PP + 1) * (ckp + 8) * 2 + 64;
                if (lds > 160 * 1024 - 512) continue;
                const double m_util = (double)P.lh * P.lw / ((double)cdiv(P.lh, TH) * cdiv(P.lw, TW) * BM);
                const double halo = (double)PP / (TH * TW * d.in_stride * d.in_stride);
                double score = c.prior * m_util * n_util / (1.0 + 0.15 * (halo - 1.0));   // HBM-side cost weighs more than in fp32
                if (lds > 80 * 1024) score *= 0.8;
                if (ckp == 16 && d.Cin >= 32) score *= 0.9;
                if (ckp == 32 && d.Cin >= 64) score *= 0.97;
                const double wgs = (double)d.N * cdiv(P.lh, TH) * cdiv(P.lw, TW) * n_cot * d.n_phases;
                const double ncu = (double)num_cus();
                score *= wgs / (ncu * ceil(wgs / ncu));
                if (score > best_score) {
                    best_score = score;
                    best = GconvBfPlan{c.MT, c.NT, ckp, TH, TW, PP, 0, n_cot, taps_max, lds};
                }
            }
        }
    }
    return best_score > 0;
}

template <int MT, int NT>
static int launch_bf(const GconvBfArgs& a, int grid, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    auto k = gconv_bf16_kernel<MT, NT>;
    if (!attr_set) {
        RD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
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

extern "C" int rd_gconv_bf16_plan_info(const RdConvDesc* d, int32_t* out) {
    GconvBfPlan pl; RdConvDesc dd;
    int rc = bf_plan_query(d, pl, dd);
    if (rc != RD_OK) return rc;
    const int v[8] = {pl.MT, pl.NT, pl.CKP, pl.TH, pl.TW, pl.PP, (int)pl.lds_bytes, d->N * pl.tiles_total * pl.n_cotiles};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RD_OK;
}

extern "C" int rd_gconv_bf16_stat_tiles(const RdConvDesc* d) {
    GconvBfPlan pl; RdConvDesc dd;
    if (bf_plan_query(d, pl, dd) != RD_OK) return RD_EINVAL;
    return d->N * pl.tiles_total;
}

extern "C" int rd_gconv_bf16(const RdConvDesc* d, const float* in, const void* w_packed_bf16, float* out, const float* bias,
                             int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial,
                             void* stream) {
    RD_CHECK_ARG(in && w_packed_bf16 && out, "gconv_bf16: null tensor");
    GconvBfArgs a;
    GconvBfPlan pl;
    int rc = bf_plan_query(d, pl, a.d);
    if (rc != RD_OK) return rc;
    a.in = in; a.w = static_cast<const unsigned short*>(w_packed_bf16); a.out = out;
    a.addend = addend; a.bias = bias; a.stat = stat_partial;
    a.act = act; a.act_cols = act_cols; a.ld_add = ld_add; a.ldw = d->Cout;
    a.TH = pl.TH; a.TW = pl.TW; a.PP = pl.PP; a.CKP = pl.CKP;
    a.tiles_total = pl.tiles_total; a.n_cotiles = pl.n_cotiles; a.taps_max = pl.taps_max;
    const int PSB = (pl.CKP + 8) * 2;
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        const int PW_ = (pl.TW - 1) * d->in_stride + (p.dw_max - p.dw_min) + 1;
        for (int t = 0; t < p.n_taps; ++t) a.tapoff[i][t] = ((p.dh[t] - p.dh_min) * PW_ + (p.dw[t] - p.dw_min)) * PSB;
    }
    const int grid = d->N * pl.tiles_total * pl.n_cotiles;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RD_BF(MT_, NT_) if (pl.MT == MT_ && pl.NT == NT_) return launch_bf<MT_, NT_>(a, grid, pl.lds_bytes, s);
    RD_BF(2, 2) RD_BF(2, 1) RD_BF(3, 2) RD_BF(1, 2) RD_BF(1, 1)
#undef RD_BF
    set_error("gconv_bf16: no kernel for tile %dx%d", pl.MT, pl.NT);
    return RD_EINVAL;
}
