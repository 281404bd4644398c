// Forward of the 16-channel depth stem (conv1_depth: 7x7, stride 2, pad 3, one input plane -- two in stage 2 of the multistage net;
// model/models.py:633,643, multistage_model.py:236-241) on v_mfma_f32_16x16x4_f32.
// stem.hip's forward kernel is built for the RGB stem: 32-wide N tiles (half of every MFMA is empty at 16 output channels) and
// persistent workgroups that keep the 38 KB weight operand resident.  Here the operand is 3-6 KB, so nothing is worth keeping:
// one workgroup per 8 x 32 output tile, as many resident as fit (their staging overlaps), N = 16 exactly.
//   A[m = pixel][k] : LDS patch planes [ci][21][69] through a k -> offset table (k = tap * Cin + ci), lane (m = l & 15, k = l >> 4)
//   B[k][n = co]    : LDS [K][16]
//   C/D             : lane (co = l & 15) holds pixels 4 (l >> 4) + i of each 16-pixel M-tile: 16 lanes store 64 contiguous bytes
#include "common.h"

namespace rd {

struct Stem16Args {
    const float* plane[2];
    long long stride[2];
    const float* w;       // packed [49][Cin][Cout]
    void* out;            // NHWC [N,Ho,Wo,Cout], fp32 or (io16) bf16
    float* stat;          // [tiles][2][Cout] partial (sum, sum of squares), or null
    int io16, Cin, N, H, W, Ho, Wo, Cout, tiles_h, tiles_w;
};

constexpr int S16_TH = 8, S16_TW = 32, S16_PH = 2 * S16_TH + 5, S16_PW = 2 * S16_TW + 5;      // patch 21 x 69
constexpr int S16_PLANE = S16_PH * S16_PW;

template <int CIN>
__global__ __launch_bounds__(256) void stem16_fwd_kernel(const Stem16Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kb = lane >> 4;
    constexpr int K = 49 * CIN, Kp = (K + 3) & ~3;
    int* s_koff = reinterpret_cast<int*>(smem);        // [Kp]
    float* s_w = smem + Kp;                            // [Kp][16]
    float* s_patch = s_w + Kp * 16;                    // [Cin][21][69]
    float* s_red = s_patch + a.Cin * S16_PLANE;        // [4][2][16]

    const int bid = blockIdx.x;
    const int n = bid / (a.tiles_h * a.tiles_w);
    const int trem = bid - n * (a.tiles_h * a.tiles_w);
    const int r0 = (trem / a.tiles_w) * S16_TH, c0 = (trem % a.tiles_w) * S16_TW;
    const int ih0 = 2 * r0 - 3, iw0 = 2 * c0 - 3;

    // patch: all loads of a thread in flight before the first LDS write; out-of-image -> 0 through the buffer descriptor
    constexpr int UP = (S16_PLANE + 255) / 256;      // 6
    float v[2][UP];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
        if (ci < a.Cin) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.plane[ci] + (size_t)n * a.stride[ci]), 0,
                                                                              (unsigned)(a.H * a.W) * 4u, 0x00020000);
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int e = tid + u * 256;
                const int py = e / S16_PW, px = e - py * S16_PW;
                const int ih = ih0 + py, iw = iw0 + px;
                const unsigned off = (e < S16_PLANE && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? (unsigned)(ih * a.W + iw) * 4u : 0x80000000u;
                v[ci][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
            }
        }
    for (int k = tid; k < Kp; k += 256) {
        int off = 0;
        if (k < K) {
            const int t = k / a.Cin, ci = k - t * a.Cin;
            off = ci * S16_PLANE + (t / 7) * S16_PW + (t % 7);
        }
        s_koff[k] = off;
    }
    for (int e = tid; e < Kp * 16; e += 256) {
        const int k = e >> 4, j = e & 15;
        s_w[e] = (k < K && j < a.Cout) ? a.w[(size_t)k * a.Cout + j] : 0.f;
    }
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
        if (ci < a.Cin) {
#pragma unroll
            for (int u = 0; u < UP; ++u)
                if (tid + u * 256 < S16_PLANE) s_patch[ci * S16_PLANE + tid + u * 256] = v[ci][u];
        }
    rd_sync();

    // wave w owns output rows 2w, 2w + 1 of the tile: four 16-pixel M-tiles (row, half row)
    int abase[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) abase[mt] = (2 * (2 * wave + (mt >> 1))) * S16_PW + 2 * ((mt & 1) * 16 + l15);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the k -> offset entries and the B column of every step first (they depend on nothing but the lane), then the walk: its
    // patch reads no longer wait for a table read of their own step
    constexpr int NST = Kp >> 2;
    int ko[NST];
    float bw[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        ko[st] = s_koff[4 * st + kb];
        bw[st] = s_w[(4 * st + kb) * 16 + l15];
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_patch[abase[mt] + ko[st]], bw[st], acc[mt], 0, 0, 0);
    }

    // epilogue: lane (co = l15) holds pixels 4 kb + i of each M-tile.  Each 4 x 4 block (4 registers x the 4 lanes of a quad = 4
    // pixels x 4 channels) is transposed with two DPP exchanges (common.h): a lane then holds four consecutive channels of ONE
    // pixel, a 16-lane group stores 256 contiguous bytes and the wave 1 KB -- a quarter of the store instructions
    float ssum = 0.f, ssq = 0.f;
    if ((a.Cout & 3) == 0) {
        const int q4l = l15 & 3, k4l = l15 >> 2;
        const bool odd1 = q4l & 1, odd2 = q4l & 2;
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oh = r0 + 2 * wave + (mt >> 1);
            const int ow = c0 + (mt & 1) * 16 + 4 * kb + q4l;
            float e0 = acc[mt][0], e1 = acc[mt][1], e2 = acc[mt][2], e3 = acc[mt][3];
            quad_transpose(e0, e1, e2, e3, odd1, odd2);
            if (oh < a.Ho && ow < a.Wo && 4 * k4l < a.Cout) {
                const float4 x = make_float4(e0, e1, e2, e3);
                const size_t o = (((size_t)n * a.Ho + oh) * a.Wo + ow) * a.Cout + 4 * k4l;
                if (a.io16) st4(static_cast<bf16s*>(a.out) + o, x);
                else st4(static_cast<float*>(a.out) + o, x);
                s4.x += x.x; s4.y += x.y; s4.z += x.z; s4.w += x.w;
                q4.x += x.x * x.x; q4.y += x.y * x.y; q4.z += x.z * x.z; q4.w += x.w * x.w;
            }
        }
        // back to one channel per lane: sum the quad's four pixels, lane q keeps channel q
        s4.x += dpp_xor1(s4.x); s4.y += dpp_xor1(s4.y); s4.z += dpp_xor1(s4.z); s4.w += dpp_xor1(s4.w);
        q4.x += dpp_xor1(q4.x); q4.y += dpp_xor1(q4.y); q4.z += dpp_xor1(q4.z); q4.w += dpp_xor1(q4.w);
        s4.x += dpp_xor2(s4.x); s4.y += dpp_xor2(s4.y); s4.z += dpp_xor2(s4.z); s4.w += dpp_xor2(s4.w);
        q4.x += dpp_xor2(q4.x); q4.y += dpp_xor2(q4.y); q4.z += dpp_xor2(q4.z); q4.w += dpp_xor2(q4.w);
        ssum = odd2 ? (odd1 ? s4.w : s4.z) : (odd1 ? s4.y : s4.x);
        ssq = odd2 ? (odd1 ? q4.w : q4.z) : (odd1 ? q4.y : q4.x);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oh = r0 + 2 * wave + (mt >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ow = c0 + (mt & 1) * 16 + 4 * kb + i;
                if (oh < a.Ho && ow < a.Wo && l15 < a.Cout) {
                    const float x = acc[mt][i];
                    const size_t o = (((size_t)n * a.Ho + oh) * a.Wo + ow) * a.Cout + l15;
                    if (a.io16) st1(static_cast<bf16s*>(a.out) + o, x);
                    else static_cast<float*>(a.out)[o] = x;
                    ssum += x;
                    ssq += x * x;
                }
            }
        }
    }
    if (a.stat) {
        ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
        ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16) {
            s_red[(wave * 2 + 0) * 16 + l15] = ssum;
            s_red[(wave * 2 + 1) * 16 + l15] = ssq;
        }
        rd_sync();
        if (tid < 32) {
            const int which = tid >> 4, j = tid & 15;
            const float s = s_red[(0 * 2 + which) * 16 + j] + s_red[(1 * 2 + which) * 16 + j] + s_red[(2 * 2 + which) * 16 + j] + s_red[(3 * 2 + which) * 16 + j];
            if (j < a.Cout) a.stat[((size_t)bid * 2 + which) * a.Cout + j] = s;
        }
    }
}

bool stem16_eligible(int Cin, int Cout) {
    static const char* off = getenv("RD_STEM_NO16");      // diagnostics: keep the depth stem on stem.hip's 32-wide kernel
    return !off && Cin >= 1 && Cin <= 2 && Cout >= 1 && Cout <= 16;
}

// same tile grid (8 x 32 output pixels) and statistics layout as stem.hip's forward kernel: rd_stem_stat_tiles is unchanged
int launch_stem16_fwd(int io16, const float* const* planes, const int64_t* strides, int Cin, int N, int H, int W, const float* w_packed, int Cout,
                      void* out, float* stat_partial, hipStream_t s) {
    Stem16Args a;
    a.io16 = io16; a.Cin = Cin; a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.Cout = Cout;
    for (int i = 0; i < 2; ++i) { a.plane[i] = i < Cin ? planes[i] : nullptr; a.stride[i] = i < Cin ? strides[i] : 0; }
    a.w = w_packed; a.out = out; a.stat = stat_partial;
    a.tiles_h = cdiv(a.Ho, S16_TH); a.tiles_w = cdiv(a.Wo, S16_TW);
    const int Kp = (49 * Cin + 3) & ~3;
    const size_t lds = ((size_t)Kp * 17 + (size_t)Cin * S16_PLANE + 128) * 4;
    if (Cin == 1) hipLaunchKernelGGL(stem16_fwd_kernel<1>, dim3(N * a.tiles_h * a.tiles_w), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(stem16_fwd_kernel<2>, dim3(N * a.tiles_h * a.tiles_w), dim3(256), lds, s, a);
    RD_CHECK_LAUNCH("stem16_fwd_kernel");
    return RD_OK;
}

}  // namespace rd
