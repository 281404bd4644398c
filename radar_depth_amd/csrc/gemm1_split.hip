// 1x1 convolutions (forward and input gradient, stride 1 and 2) as a GEMM with gconv_split.hip's arithmetic: fp32 operands as three bf16
// pieces, six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation.  Round 5: the 1x1 layers of the default plan -- fusion 640 -> 512,
// conv2 512 -> 256, the three downsample convolutions and their input gradients (/root/reference/model/models.py:559-569,600-625,652-657) --
// ran on gconv.hip's fp32 MFMA kernel at 40-70 TFLOP/s: the tap-group structure of gconv_split_kernel pads a one-tap phase to three MFMA
// steps (two thirds of the matrix work wasted), so rd_gconv_split did not serve them.  Here the reduction is grouped by CHANNELS instead:
//
//   out[m][co] = sum over ci of x[pixel(m)][ci] * w[ci][co]        m = a logical output pixel of the descriptor's phase, flattened over
//                                                                   (image, row, column); pixel(m) / the output position by the strides
//   workgroup : 4 waves, two per CU; BM = 4 x MT x 32 rows, BN = NT x 32 output channels; no halo, no patch: a row is one pixel.
//   stage     : 32 input channels = two MFMA steps.  The weights of stage s + 1 (three piece planes [4 units][BN] x 16 B, already in this
//               layout in the packed operand) are copied by global_load_lds while stage s computes; the activations of stage s + 1 are
//               loaded into registers in front of stage s's MFMAs, split into pieces behind them (v_cvt_pk_bf16_f32 pairs, 11 VALU
//               per pair) and stored into the single A image [piece][unit][row] x 16 B between two barriers.  The co-resident
//               workgroup's MFMAs run under that.
//   reads     : lane (l31, hh) reads row l31 of its M tile, unit 2 step + hh: consecutive lanes -> consecutive 16-byte slots (no bank
//               conflicts for the ds_read_b128 passes: rows are contiguous, there is no tile-row break); B likewise.
//   epilogue  : gconv_split.hip's (4x4 register transposition, 16-byte accesses, bias / addend / activation, BatchNorm partial sums
//               per row tile, rows masked one by one).
// Dispatched from rd_gconv_split / rd_gconv_split_supported / rd_gconv_split_stat_tiles (gconv_split.hip): callers see one entry point.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace rd {

typedef __bf16 g1bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int g1u32x4 __attribute__((ext_vector_type(4)));
typedef float g1f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g1bf16x2 __attribute__((ext_vector_type(2)));

constexpr unsigned G1_OOB = 0x80000000u;
constexpr int G1_CK = 32;             // input channels per stage
constexpr int G1_UPAD = 64;           // unit stride of the activation image = rows x 16 B + 64 B: the four units of a store pass sit 16 banks apart

struct G1Args {
    RdConvDesc d;
    const float* in;
    const unsigned short* w;          // packed operand, three piece planes [slab][Cin/8][ldw][8] bf16
    float* out;
    const float* addend;
    const float* bias;
    float* stat;
    int act, act_cols, ld_add, ldw;
    int vec4, n_cotiles;
    long long wplane;                 // bytes per piece plane of the packed operand
    int tile_begin[RD_MAX_PHASES + 1];        // first row tile of each phase; [n_phases] = total
};

__device__ __forceinline__ unsigned g1_cvt_pk(float a, float b) {
    g1f32x2 v;
    v[0] = a; v[1] = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, g1bf16x2));
}
// three bf16 pieces of eight fp32 values (gconv_split.hip: split8)
__device__ __forceinline__ void g1_split8(const float4 v0, const float4 v1, g1u32x4& w0, g1u32x4& w1, g1u32x4& w2) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = x[2 * i], b = x[2 * i + 1];
        const unsigned u0 = g1_cvt_pk(a, b);
        a -= __uint_as_float(u0 << 16);
        b -= __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = g1_cvt_pk(a, b);
        a -= __uint_as_float(u1 << 16);
        b -= __uint_as_float(u1 & 0xffff0000u);
        w0[i] = u0;
        w1[i] = u1;
        w2[i] = g1_cvt_pk(a, b);
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm1_split_kernel(const G1Args a) {
    constexpr int BM = 4 * MT * 32;
    constexpr int BN = NT * 32;
    constexpr int UPT = BM * 4 / 256;                 // (row, 8-channel unit) items of a stage per thread: 4 (BM = 256) or 2 (BM = 128)
    constexpr int AU = BM * 16 + G1_UPAD;             // bytes per unit of the A image: rows x 16 B, padded (see split_put)
    constexpr int APL = 4 * AU;                       // bytes per piece of the A image [unit][row] x 16 B
    constexpr int BPL = 4 * BN * 16;                  // bytes per piece of a weight buffer [unit][BN] x 16 B
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const RdConvDesc& D = a.d;

    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = vid % a.n_cotiles;
    const int pt = vid / a.n_cotiles;
    int ph_ = 0;
    for (int i = 1; i < D.n_phases; ++i)
        if (pt >= a.tile_begin[i]) ph_ = i;
    const int ph = __builtin_amdgcn_readfirstlane(ph_);
    const RdPhase& P = D.phase[ph];
    const int m0 = (pt - a.tile_begin[ph]) * BM;
    const int Mph = D.N * P.lh * P.lw;
    const int co0 = cot * BN;

    int* s_opix = reinterpret_cast<int*>(smem);                  // [BM] output pixel index or -1
    unsigned* s_in = reinterpret_cast<unsigned*>(s_opix + BM);   // [BM] byte offset of the row's pixel in `in` at channel 0 / 2 (see below), G1_OOB: zeros
    float* s_red = reinterpret_cast<float*>(s_in + BM);          // [4][2][BN]
    char* s_a = reinterpret_cast<char*>(s_red + 8 * BN);         // [3][4][BM] x 16 B
    char* s_b = s_a + 3 * APL;                                   // [2][3][4][BN] x 16 B

    for (int m = tid; m < BM; m += 256) {
        const int gm = m0 + m;
        const bool ok = gm < Mph;
        const int n = gm / (P.lh * P.lw), rem = gm - n * (P.lh * P.lw);
        const int r = rem / P.lw, c = rem - r * P.lw;
        const int ih = r * D.in_stride + P.dh[0], iw = c * D.in_stride + P.dw[0];
        const bool in_ok = ok && ih >= 0 && ih < D.Hi && iw >= 0 && iw < D.Wi;
        s_opix[m] = ok ? ((n * D.Ho + r * D.out_stride + P.out_off_h) * D.Wo + c * D.out_stride + P.out_off_w) : -1;
        // (pixel offsets in units of 8 bytes: a whole tensor of up to 16 GiB stays below the out-of-range marker)
        s_in[m] = in_ok ? (unsigned)((((size_t)n * D.Hi + ih) * D.Wi + iw) * D.ldi >> 1) : G1_OOB;
    }
    rd_sync();

    // staging map of this thread: item = it * 256 + tid -> (row = item / 4, unit = item % 4): four lanes read the 128 contiguous bytes of
    // a row's 32 channels (a wave's load touches 16 lines, each by four lanes), and store 16 bytes per piece at [unit][row]: the 16 lanes
    // of a ds_write_b128 pass are 4 rows x 4 units = 4 x 16 B x (4 units 64 B apart mod 256): all 64 banks once.
    const int su = tid & 3;
    const float* srcp[UPT];
    bool sok[UPT];
#pragma unroll
    for (int it = 0; it < UPT; ++it) {
        const unsigned srow = s_in[it * 64 + (tid >> 2)];
        sok[it] = srow != G1_OOB;
        srcp[it] = a.in + (sok[it] ? ((size_t)srow << 1) : 0) + su * 8;       // (rows outside the input: a valid address, the value is dropped)
    }
    const int cin8 = D.Cin >> 3;
    const int S = D.Cin / G1_CK;
    const size_t wslab = (size_t)P.widx[0] * cin8 * a.ldw;      // units in front of this tap's slab

    float4 v0[UPT], v1[UPT];
    auto fetch = [&](int s) {
#pragma unroll
        for (int it = 0; it < UPT; ++it) {
            const float* p = srcp[it] + s * G1_CK;
            v0[it] = *reinterpret_cast<const float4*>(p);
            v1[it] = *reinterpret_cast<const float4*>(p + 4);
        }
    };
    auto split_put = [&]() {
        char* base = s_a + su * AU + (tid >> 2) * 16;
#pragma unroll
        for (int it = 0; it < UPT; ++it) {
            g1u32x4 w0, w1, w2;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            g1_split8(sok[it] ? v0[it] : z, sok[it] ? v1[it] : z, w0, w1, w2);
            char* ad = base + it * 64 * 16;
            *reinterpret_cast<g1u32x4*>(ad) = w0;
            *reinterpret_cast<g1u32x4*>(ad + APL) = w1;
            *reinterpret_cast<g1u32x4*>(ad + 2 * APL) = w2;
        }
    };
    // weights of stage s into buffer buf: 3 pieces x 4 units x BN columns of 16 bytes; copy e = (piece, unit[, column half]) per wave
    auto issue_w = [&](int s, int buf) {
        char* dst = s_b + buf * 3 * BPL;
        const char* src = reinterpret_cast<const char*>(a.w) + (wslab + (size_t)(s * (G1_CK / 8)) * a.ldw + co0) * 16;
        if constexpr (NT == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = wm + 4 * i;
                const int p = e >> 2, k8 = e & 3;
                glds16(reinterpret_cast<const float*>(src + p * a.wplane + ((size_t)k8 * a.ldw + lane) * 16), reinterpret_cast<float*>(dst + p * BPL + k8 * BN * 16));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = wm + 4 * i;
                if (e >= 6) break;
                const int p = e >> 1, k8 = (e & 1) * 2 + hh;
                glds16(reinterpret_cast<const float*>(src + p * a.wplane + ((size_t)k8 * a.ldw + l31) * 16), reinterpret_cast<float*>(dst + p * BPL + (e & 1) * 2 * BN * 16));
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
    const char* abase = s_a + hh * AU + (wm * MT * 32 + l31) * 16;
    const char* bbase = s_b + (hh * BN + l31) * 16;

    issue_w(0, 0);
    fetch(0);
    split_put();
    for (int s = 0; s < S; ++s) {
        glds_wait();                              // this wave's weight copies of stage s and its A stores
        rd_sync();                                // stage s is published; every wave has left stage s - 1
        const bool more = s + 1 < S;
        if (more) {
            issue_w(s + 1, (s + 1) & 1);
            fetch(s + 1);
        }
        const char* bb = bbase + (s & 1) * 3 * BPL;
        g1bf16x8 A[2][3][MT], B[2][3][NT];
        auto load = [&](int ks, g1bf16x8 (&Af)[3][MT], g1bf16x8 (&Bf)[3][NT]) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    Af[p][mt] = *reinterpret_cast<const g1bf16x8*>(abase + p * APL + ks * 2 * AU + mt * 512);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    Bf[p][nt] = *reinterpret_cast<const g1bf16x8*>(bb + p * BPL + ks * 2 * BN * 16 + nt * 512);
            }
        };
        auto mma = [&](const g1bf16x8 (&Af)[3][MT], const g1bf16x8 (&Bf)[3][NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x16 c = acc[mt][nt];
                    RD_SPLIT_TERMS(c, Af[0][mt], Af[1][mt], Af[2][mt], Bf[0][nt], Bf[1][nt], Bf[2][nt])
                    acc[mt][nt] = c;
                }
        };
        load(0, A[0], B[0]);
        load(1, A[1], B[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(A[0], B[0]);
        mma(A[1], B[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            rd_sync();                            // every wave is done reading the A image of stage s
            split_put();
        }
    }

    // ---- epilogue (gconv_split.hip's)
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    {
        const bool has_add = a.addend != nullptr;
        const bool has_bias = a.bias != nullptr;
        const bool want_stat = a.stat != nullptr;
        const int cob = co0 + l31;
        const int q4l = l31 & 3, k4l = l31 >> 2;
        const bool odd1 = q4l & 1, odd2 = q4l & 2;
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
                    ro4[q] = s_opix[(wm * MT + mt) * 32 + q4l + 8 * q + 4 * hh];
                    rok4[q] = ro4[q] >= 0;
                    ro4[q] = rok4[q] ? ro4[q] : 0;
                }
                const int cq = co0 + 4 * k4l;
                float4 addv[NT][4];
                if (has_add) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float* ap = a.addend + (size_t)ro4[q] * a.ld_add + cq;
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
                        if (has_add) { v.x += addv[nt][q].x; v.y += addv[nt][q].y; v.z += addv[nt][q].z; v.w += addv[nt][q].w; }
                        const int cc = cq + nt * 32;
                        if (cc < a.act_cols) {
                            v.x = act_fwd(v.x, a.act); v.y = act_fwd(v.y, a.act); v.z = act_fwd(v.z, a.act); v.w = act_fwd(v.w, a.act);
                        }
                        if (cok4 && rok4[q]) st4(a.out + (size_t)ro4[q] * D.ldo + cc, v);
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
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = s_opix[(wm * MT + mt) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh];
                    if (ro < 0) continue;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = cob + nt * 32;
                        if (co >= D.Cout) continue;
                        float v = acc[mt][nt][i] + (has_bias ? a.bias[co] : 0.f);
                        if (has_add) v += a.addend[(size_t)ro * a.ld_add + co];
                        if (co < a.act_cols) v = act_fwd(v, a.act);
                        a.out[(size_t)ro * D.ldo + co] = v;
                        ssum[nt] += v;
                        ssq[nt] += v * v;
                    }
                }
            }
        }
    }
    if (a.stat) {
        rd_sync();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float sv = ssum[nt] + __shfl_xor(ssum[nt], 32, 64);
            const float qv = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
            if (hh == 0) {
                s_red[(wm * 2 + 0) * BN + nt * 32 + l31] = sv;
                s_red[(wm * 2 + 1) * BN + nt * 32 + l31] = qv;
            }
        }
        rd_sync();
        if (tid < 2 * BN) {
            const int which = tid / BN, j = tid - which * BN;
            float sv = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sv += s_red[(w * 2 + which) * BN + j];
            const int co = co0 + j;
            if (co < D.Cout) a.stat[((size_t)pt * 2 + which) * D.Cout + co] = sv;
        }
    }
}

// ------------------------------------------------------------------------------------------ host
struct G1Plan { int MT, NT, n_cotiles, tiles; int tile_begin[RD_MAX_PHASES + 1]; size_t lds; };

static bool g1_plan(const RdConvDesc* d, G1Plan& pl) {
    static const bool off = getenv("RD_GEMM1_SPLIT") && atoi(getenv("RD_GEMM1_SPLIT")) == 0;      // A/B switch: 1x1 layers stay on rd_gconv
    if (off || !d || d->n_phases < 1 || d->n_phases > RD_MAX_PHASES) return false;
    if (d->in_stride < 1 || d->in_stride > 2 || d->out_stride < 1 || d->out_stride > 2) return false;
    if (d->Cin % G1_CK != 0 || d->Cin < 32 || d->Cout < 32 || d->Cout % 8 != 0 || d->ldi % 4 != 0) return false;
    if ((int64_t)d->N * d->Hi * d->Wi * d->ldi / 2 >= (int64_t)G1_OOB) return false;
    int64_t rows = 0;
    for (int i = 0; i < d->n_phases; ++i) {
        const RdPhase& p = d->phase[i];
        if (p.n_taps != 1 || p.lh < 1 || p.lw < 1 || p.widx[0] < 0) return false;
        rows += (int64_t)d->N * p.lh * p.lw;
    }
    pl.NT = d->Cout > 32 ? 2 : 1;
    if (d->Cout % (pl.NT * 32) != 0) return false;        // (the weight copies read whole column tiles)
    pl.n_cotiles = d->Cout / (pl.NT * 32);
    // 256-row tiles when they still give every CU its two workgroups, else 128-row tiles
    pl.MT = (rows / 256) * pl.n_cotiles >= 2 * (int64_t)num_cus() ? 2 : 1;
    const int BM = 4 * pl.MT * 32, BN = pl.NT * 32;
    int tb = 0;
    for (int i = 0; i < d->n_phases; ++i) {
        pl.tile_begin[i] = tb;
        tb += (int)cdiv64((int64_t)d->N * d->phase[i].lh * d->phase[i].lw, BM);
    }
    for (int i = d->n_phases; i <= RD_MAX_PHASES; ++i) pl.tile_begin[i] = tb;
    pl.tiles = tb;
    pl.lds = (size_t)BM * 8 + 8 * BN * 4 + 3 * (size_t)4 * (BM * 16 + G1_UPAD) + 2 * 3 * (size_t)4 * BN * 16;
    return pl.lds <= 80 * 1024 - 256;
}

template <int MT, int NT>
static int launch_g1(const G1Args& a, int grid, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    auto k = gemm1_split_kernel<MT, NT>;
    RD_SET_ATTR_ONCE(attr_set, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("gemm1_split_kernel");
    return RD_OK;
}

int gemm1_split_supported(const RdConvDesc* d) {
    G1Plan pl;
    return g1_plan(d, pl) ? 1 : 0;
}

int gemm1_split_stat_tiles(const RdConvDesc* d) {
    G1Plan pl;
    return g1_plan(d, pl) ? pl.tiles : RD_EINVAL;
}

int gemm1_split_plan_info(const RdConvDesc* d, int32_t* out) {
    G1Plan pl;
    if (!out || !g1_plan(d, pl)) return RD_EINVAL;
    const int v[8] = {pl.MT, pl.NT, 4 * pl.MT * 32, 1, 0, (int)pl.lds, pl.tiles * pl.n_cotiles, 1000};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RD_OK;
}

int launch_gemm1_split(const RdConvDesc* d, const float* in, const void* w_split, int64_t piece_elems, float* out, const float* bias, int32_t act,
                       int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, hipStream_t s) {
    G1Args a;
    G1Plan pl;
    if (!g1_plan(d, pl)) { set_error("gemm1_split: descriptor not supported"); return RD_EINVAL; }
    RD_CHECK_ARG(in && w_split && out, "gemm1_split: null argument");
    RD_CHECK_ARG(reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(w_split) % 16 == 0, "gemm1_split: unaligned tensor");
    int S = 0;
    for (int i = 0; i < d->n_phases; ++i) S = S > d->phase[i].widx[0] + 1 ? S : d->phase[i].widx[0] + 1;
    RD_CHECK_ARG(piece_elems >= (int64_t)S * d->Cin * d->Cout && piece_elems % 8 == 0, "gemm1_split: piece stride %lld too small", (long long)piece_elems);
    a.d = *d;
    a.in = in; a.w = static_cast<const unsigned short*>(w_split); a.out = out; a.addend = addend; a.bias = bias; a.stat = stat_partial;
    a.act = act; a.act_cols = act_cols; a.ld_add = ld_add; a.ldw = d->Cout;
    a.n_cotiles = pl.n_cotiles;
    a.wplane = (long long)piece_elems * 2;
    a.vec4 = d->Cout % 4 == 0 && d->ldo % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && act_cols % 4 == 0 &&
             (!addend || (ld_add % 4 == 0 && reinterpret_cast<uintptr_t>(addend) % 16 == 0)) && (!bias || reinterpret_cast<uintptr_t>(bias) % 16 == 0);
    for (int i = 0; i <= RD_MAX_PHASES; ++i) a.tile_begin[i] = pl.tile_begin[i];
    const int grid = pl.tiles * pl.n_cotiles;
    if (pl.MT == 2 && pl.NT == 2) return launch_g1<2, 2>(a, grid, pl.lds, s);
    if (pl.MT == 1 && pl.NT == 2) return launch_g1<1, 2>(a, grid, pl.lds, s);
    if (pl.MT == 2 && pl.NT == 1) return launch_g1<2, 1>(a, grid, pl.lds, s);
    return launch_g1<1, 1>(a, grid, pl.lds, s);
}

}  // namespace rd
