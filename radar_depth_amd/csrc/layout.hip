// Layout glue: weight packing for gconv, NCHW<->NHWC transposes (module boundary), fills.
#include "common.h"

namespace rd {

// Logical packed[slab][row][col]: transpose==0 -> row=i (I rows), col=co_off+o ; transpose!=0 -> row=co_off+o, col=i.
// gconv reads its operand with the reduction rows interleaved by four ("quad" layout): element (slab, row, col) lives at
// ((slab * R/4 + row/4) * ldc + col) * 4 + row%4, so that one 16-byte LDS read feeds four consecutive MFMAs.
__device__ __forceinline__ int64_t packed_index(int quad, int64_t t, int64_t rows, int64_t row, int64_t ldc, int64_t col) {
    return quad ? ((t * (rows >> 2) + (row >> 2)) * ldc + col) * 4 + (row & 3) : (t * rows + row) * ldc + col;
}
// bf16 operand of gconv_bf16.hip (quad == 2): eight reduction rows per 16-byte unit, element (slab, row, col) at
// ((slab * R/8 + row/8) * ldc + col) * 8 + row%8 (bf16 units)
__device__ __forceinline__ int64_t packed_index_bf16(int64_t t, int64_t rows, int64_t row, int64_t ldc, int64_t col) {
    return ((t * (rows >> 3) + (row >> 3)) * ldc + col) * 8 + (row & 7);
}
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int O, int I, int T,
                                    int ldc, int off, int rows_total, int transpose) {
    const int64_t total = (int64_t)O * I * T;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        // destination-major enumeration keeps the writes coalesced
        int t, o, i;
        if (!transpose) {
            o = (int)(e % O);
            const int64_t r = e / O;
            i = (int)(r % I);
            t = (int)(r / I);
            packed[packed_index(1, t, I, i, ldc, off + o)] = w[((int64_t)o * I + i) * T + t];
        } else {
            i = (int)(e % I);
            const int64_t r = e / I;
            o = (int)(r % O);
            t = (int)(r / O);
            packed[packed_index(1, t, rows_total, off + o, ldc, i)] = w[((int64_t)o * I + i) * T + t];
        }
    }
}

__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ packed, int O, int I, int T,
                                         int ldc, int off, int rows_total, int transpose) {
    const int64_t total = (int64_t)O * I * T;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(e % O);
        const int64_t r = e / O;
        const int i = (int)(r % I), t = (int)(r / I);
        const float v = w[((int64_t)o * I + i) * T + t];
        if (!transpose) packed[packed_index_bf16(t, I, i, ldc, off + o)] = (__bf16)v;
        else packed[packed_index_bf16(t, rows_total, off + o, ldc, i)] = (__bf16)v;
    }
}

// One launch for every weight tensor of the network: block b works on chunk (b - job.first_block) of job block_job[b].
// Same index mapping as pack_weights_kernel.
struct PackJob {
    const float* src;
    float* dst;
    const float* scale;   // optional per-output-channel factor (eval mode: folded BatchNorm scale), may be null
    int O, I, T, ldc, off, rows_total, transpose, first_block;
    int quad, pad_;       // 1: gconv operand layout (rows interleaved by four); 2: bf16 operand (by eight); 0: plain [slab][row][col] (stem kernels)
                          // 3: three bf16 piece planes of the bf16 layout (gconv_split.hip), plane stride T * rows * ldc elements
};
// the three bf16 pieces of an fp32 value (gconv_split.hip): v = p0 + p1 + p2 exactly
__device__ __forceinline__ void store_split3(__bf16* dst, int64_t idx, int64_t plane, float v) {
    const __bf16 h = (__bf16)v;
    float r = v - (float)h;
    const __bf16 m = (__bf16)r;
    r -= (float)m;
    dst[idx] = h;
    dst[idx + plane] = m;
    dst[idx + 2 * plane] = (__bf16)r;
}
constexpr int PACK_CHUNK = 2048;
// bf16 operands (quad 2: one plane; quad 3: the three piece planes of gconv_split.hip): a thread owns ONE 16-byte unit position -- eight
// consecutive reduction rows of one column -- for ALL taps: it reads its 8 x T source values (OIHW: the T taps of an (o, i) pair are
// contiguous) and writes T whole units per plane, consecutive threads consecutive columns.  (Round 4: element by element this kernel
// wrote 2-byte values 16 bytes apart -- 264 us for 1.05 GB of HBM traffic per step in the split plan, at the start of the step where
// nothing overlaps it.)
__device__ __forceinline__ void pack_units_bf16(const PackJob& j, int64_t u) {
    const int R = j.transpose ? j.O : j.I;              // reduction rows this job contributes
    const int Cn = j.transpose ? j.I : j.O;             // columns
    const int64_t units = (int64_t)(R >> 3) * Cn;
    if (u >= units) return;
    const int col = (int)(u % Cn), r8 = (int)(u / Cn);
    const int rows_total = j.transpose ? j.rows_total : j.I;
    const int row0 = (j.transpose ? j.off : 0) + r8 * 8, colp = (j.transpose ? 0 : j.off) + col;
    __bf16* dst = reinterpret_cast<__bf16*>(j.dst);
    const int64_t plane = (int64_t)j.T * rows_total * j.ldc;
    for (int t = 0; t < j.T; ++t) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int o = j.transpose ? r8 * 8 + r : col, i = j.transpose ? col : r8 * 8 + r;
            v[r] = j.src[((int64_t)o * j.I + i) * j.T + t];
            if (j.scale) v[r] *= j.scale[o];
        }
        const int64_t idx = packed_index_bf16(t, rows_total, row0, j.ldc, colp);      // (row0 % 8 == 0: the unit's first element)
        unsigned w0[4], w1[4], w2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = v[2 * q], b = v[2 * q + 1];
            w0[q] = cvt_pk_bf16(a, b);
            a -= __uint_as_float(w0[q] << 16); b -= __uint_as_float(w0[q] & 0xffff0000u);
            w1[q] = cvt_pk_bf16(a, b);
            a -= __uint_as_float(w1[q] << 16); b -= __uint_as_float(w1[q] & 0xffff0000u);
            w2[q] = cvt_pk_bf16(a, b);
        }
        *reinterpret_cast<uint4*>(dst + idx) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
        if (j.quad == 3) {
            *reinterpret_cast<uint4*>(dst + idx + plane) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
            *reinterpret_cast<uint4*>(dst + idx + 2 * plane) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
        }
    }
}
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackJob* __restrict__ jobs, const int* __restrict__ block_job) {
    const PackJob j = jobs[block_job[blockIdx.x]];
    if (j.quad >= 2 && ((j.transpose ? (j.O | j.off | j.rows_total) : j.I) & 7) == 0) {
        // (the job owns ceil(O I T / PACK_CHUNK) blocks, sized for the element-wise form: a block of this form covers 256 units = 2048 T
        //  elements, so the surplus blocks find nothing to do)
        pack_units_bf16(j, (int64_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x);
        return;
    }
    const int64_t total = (int64_t)j.O * j.I * j.T;
    const int64_t base = (int64_t)(blockIdx.x - j.first_block) * PACK_CHUNK;
#pragma unroll
    for (int u = 0; u < PACK_CHUNK / 256; ++u) {
        const int64_t e64 = base + u * 256 + threadIdx.x;
        if (e64 >= total) break;
        const unsigned e = (unsigned)e64;      // one weight tensor has fewer than 2^31 elements (engine.py asserts it when it builds the job table): 32-bit divisions
        if (!j.transpose) {
            const unsigned r = e / (unsigned)j.O;
            const int o = (int)(e - r * (unsigned)j.O);
            const int t = (int)(r / (unsigned)j.I), i = (int)(r - (r / (unsigned)j.I) * (unsigned)j.I);
            float v = j.src[((int64_t)o * j.I + i) * j.T + t];
            if (j.scale) v *= j.scale[o];
            if (j.quad == 3) store_split3(reinterpret_cast<__bf16*>(j.dst), packed_index_bf16(t, j.I, i, j.ldc, j.off + o), (int64_t)j.T * j.I * j.ldc, v);
            else if (j.quad == 2) reinterpret_cast<__bf16*>(j.dst)[packed_index_bf16(t, j.I, i, j.ldc, j.off + o)] = (__bf16)v;
            else j.dst[packed_index(j.quad, t, j.I, i, j.ldc, j.off + o)] = v;
        } else {
            const unsigned r = e / (unsigned)j.I;
            const int i = (int)(e - r * (unsigned)j.I);
            const int t = (int)(r / (unsigned)j.O), o = (int)(r - (r / (unsigned)j.O) * (unsigned)j.O);
            float v = j.src[((int64_t)o * j.I + i) * j.T + t];
            if (j.scale) v *= j.scale[o];
            if (j.quad == 3) store_split3(reinterpret_cast<__bf16*>(j.dst), packed_index_bf16(t, j.rows_total, j.off + o, j.ldc, i), (int64_t)j.T * j.rows_total * j.ldc, v);
            else if (j.quad == 2) reinterpret_cast<__bf16*>(j.dst)[packed_index_bf16(t, j.rows_total, j.off + o, j.ldc, i)] = (__bf16)v;
            else j.dst[packed_index(j.quad, t, j.rows_total, j.off + o, j.ldc, i)] = v;
        }
    }
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) p[e] = v;
}

// 32x32 LDS-tiled transpose between [C][HW] and [HW][C] per image
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    src += (size_t)n * rows * cols;
    dst += (size_t)n * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = src[(size_t)r * cols + c];
    }
    rd_sync();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

static int grid_for(int64_t n, int block) {
    int64_t g = cdiv64(n, block);
    const int64_t cap = (int64_t)num_cus() * 8;
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace rd
using namespace rd;

extern "C" int rd_pack_weights(const float* w_oihw, float* packed, int32_t O, int32_t I, int32_t KH, int32_t KW,
                               int32_t ldc, int32_t co_off, int32_t rows_total, int32_t transpose, void* stream) {
    RD_CHECK_ARG(w_oihw && packed && O > 0 && I > 0 && KH > 0 && KW > 0, "pack_weights: bad arguments");
    RD_CHECK_ARG((transpose ? rows_total : I) % 4 == 0, "pack_weights: the reduction dimension must be a multiple of 4");
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w_oihw, packed, O, I, KH * KW, ldc, co_off, rows_total, transpose);
    RD_CHECK_LAUNCH("pack_weights_kernel");
    return RD_OK;
}

extern "C" int rd_pack_weights_bf16(const float* w_oihw, void* packed_bf16, int32_t O, int32_t I, int32_t KH, int32_t KW,
                                    int32_t ldc, int32_t co_off, int32_t rows_total, int32_t transpose, void* stream) {
    RD_CHECK_ARG(w_oihw && packed_bf16 && O > 0 && I > 0 && KH > 0 && KW > 0, "pack_weights_bf16: bad arguments");
    RD_CHECK_ARG((transpose ? rows_total : I) % 8 == 0, "pack_weights_bf16: the reduction dimension must be a multiple of 8");
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w_oihw, static_cast<__bf16*>(packed_bf16), O, I, KH * KW, ldc, co_off, rows_total, transpose);
    RD_CHECK_LAUNCH("pack_weights_bf16_kernel");
    return RD_OK;
}

extern "C" int rd_pack_chunk(void) { return PACK_CHUNK; }

extern "C" int rd_pack_weights_batched(const void* jobs_dev, const int32_t* block_job_dev, int32_t n_blocks, void* stream) {
    RD_CHECK_ARG(jobs_dev && block_job_dev && n_blocks > 0, "pack_weights_batched: bad arguments");
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PackJob*>(jobs_dev), block_job_dev);
    RD_CHECK_LAUNCH("pack_weights_batched_kernel");
    return RD_OK;
}

extern "C" int rd_fill(float* p, int64_t n, float v, void* stream) {
    if (n <= 0) return RD_OK;
    RD_CHECK_ARG(p != nullptr, "fill: null pointer");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p, n, v);
    RD_CHECK_LAUNCH("fill_kernel");
    return RD_OK;
}

extern "C" int rd_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    RD_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
    const int rows = C, cols = H * W;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), N), dim3(32, 8), 0,
                       static_cast<hipStream_t>(stream), src, dst, rows, cols);
    RD_CHECK_LAUNCH("transpose_kernel");
    return RD_OK;
}

extern "C" int rd_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream) {
    RD_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad arguments");
    const int rows = H * W, cols = C;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), N), dim3(32, 8), 0,
                       static_cast<hipStream_t>(stream), src, dst, rows, cols);
    RD_CHECK_LAUNCH("transpose_kernel");
    return RD_OK;
}
