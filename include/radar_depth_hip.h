/*
 * radar_depth_hip.h -- C ABI of libradardepth_hip.so (gfx950 / MI355X).
 *
 * The reference (brade31919/radar_depth) has no FFI: its hot path is torch.nn layers that
 * dispatch to ATen/cuDNN.  This library is the MI355X-native replacement for those
 * dispatches, underneath the reference's nn.Module plugin surface.  Every entry point cites
 * the reference call it replaces (paths relative to /root/reference).
 *
 * Conventions
 *  - plain C: raw device pointers + sizes, no torch types.  Pointers are BORROWED for the
 *    duration of the call; the library never allocates or frees device memory.
 *  - activations are NHWC fp32 with an explicit channel stride `ld` (>= channels) so a
 *    tensor may be a channel slice of a wider buffer (the late-fusion concat is free).
 *  - every call is asynchronous on the caller's hipStream_t (passed as void*), never
 *    synchronises, and is re-entrant per stream.
 *  - returns 0 on success, a negative RD_E* code otherwise; rd_last_error() gives the text.
 */
#ifndef RADAR_DEPTH_HIP_H
#define RADAR_DEPTH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RD_OK 0
#define RD_EINVAL (-1)   /* bad argument / unsupported shape */
#define RD_ELAUNCH (-2)  /* HIP launch / runtime error */

#define RD_MAX_TAPS 25
#define RD_MAX_PHASES 4

#define RD_ACT_NONE 0
#define RD_ACT_RELU 1
#define RD_ACT_LEAKY02 2 /* LeakyReLU(0.2), model/models.py:564 */

const char* rd_last_error(void);
int rd_abi_version(void);
/* number of CUs / name of device 0 as the library sees it (diagnostics) */
int rd_device_info(int* n_cu, char* name, int name_len);

/* ---------------------------------------------------------------------------------------
 * Generalised convolution ("gconv").  One descriptor expresses every conv on the path:
 *
 *   out[n, oh*out_stride + out_off_h, ow*out_stride + out_off_w, co] =
 *       sum_{t < n_taps} sum_{ci}  in[n, oh*in_stride + dh[t], ow*in_stride + dw[t], ci]
 *                                   * w[widx[t]][ci][co]            (out-of-range input = 0)
 *
 * for (oh, ow) over the phase's logical grid lh x lw.  Phases write disjoint output pixels.
 *   - 3x3 / 1x1 forward, stride 1 or 2        : 1 phase          (nn.Conv2d, models.py:96-112)
 *   - Unpool + 5x5 conv of an UpProj module    : 4 phases, 9/6/6/4 taps on the LOW-RES input,
 *     both branches fused along co             (models.py:13-27,181-209; zero-skipping identity)
 *   - every input-gradient (dgrad) of the above: same form with transposed weights
 * Weights are the packed layout produced by rd_pack_weights: logical [slab][cin][cout], stored with the
 * reduction rows interleaved by four: element (slab, ci, co) at ((slab*Cin/4 + ci/4)*ld + co)*4 + ci%4.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_taps;
    int32_t out_off_h, out_off_w;
    int32_t lh, lw;                  /* logical output grid of this phase */
    int32_t dh_min, dh_max, dw_min, dw_max;
    int32_t tile_begin;              /* filled by the library */
    int8_t dh[RD_MAX_TAPS];
    int8_t dw[RD_MAX_TAPS];
    int16_t widx[RD_MAX_TAPS];
} RdPhase;

typedef struct {
    int32_t N;
    int32_t Hi, Wi, Cin, ldi;
    int32_t Ho, Wo, Cout, ldo;
    int32_t in_stride, out_stride;
    int32_t n_phases;
    RdPhase phase[RD_MAX_PHASES];
} RdConvDesc;

/* Forward / dgrad convolution.  addend (optional, may be NULL): out += addend[pixel][co]
 * (residual-gradient merge).  stat_partial (optional): per-pixel-tile partial (sum, sum of
 * squares) per output channel for training-mode BatchNorm, layout [n_stat_tiles][2][Cout];
 * the tile count is returned by rd_gconv_stat_tiles.  Replaces F.conv2d /
 * conv_transpose2d+conv2d (models.py:27,203-206) and their autograd input-gradients. */
int rd_gconv(const RdConvDesc* d, const float* in, const float* w_packed, float* out,
             const float* addend, int32_t ld_add, float* stat_partial, void* stream);
int rd_gconv_stat_tiles(const RdConvDesc* d);
/* Same convolution with a caller-provided workspace: lets the library split the input channels of small-spatial /
 * many-channel layers over several workgroups per tile (split-K; partials in ws are combined, the addend added and
 * the BN partial sums produced by a second kernel).  ws: rd_gconv_workspace_floats(d) floats (0 -> no split, ws may
 * be NULL); the stat buffer then holds rd_gconv_stat_tiles_ws(d) tiles. */
int64_t rd_gconv_workspace_floats(const RdConvDesc* d);
int rd_gconv_stat_tiles_ws(const RdConvDesc* d);
int rd_gconv_ws(const RdConvDesc* d, const float* in, const float* w_packed, float* out,
                const float* addend, int32_t ld_add, float* stat_partial, float* ws, void* stream);
/* Input gradient with the BatchNorm-backward sums in its epilogue.  For the chains conv -> BatchNorm -> ReLU -> conv
 * (models.py:96-112 BasicBlock conv1/bn1/relu/conv2; :203-206 UpProj conv1/batchnorm1/relu/conv2) the gradient dy that the second
 * convolution's dgrad produces is exactly what the first BatchNorm's backward reduces: g = dy * act'(scale*x + shift),
 * red_partial[tile][0][c] = sum g, red_partial[tile][1][c] = sum g * (x - mean) over the launch's pixel tiles
 * (rd_gconv_stat_tiles_ws(d) tiles, layout [tiles][3][Cout], slot 2 untouched) -- the layout rd_bn_bwd_apply_x_t reads.  Saves the
 * rd_bn_bwd_reduce_x_t pass over (dy, x) and its launch.  rd_gconv_bnbwd_supported: 1 when the descriptor's plan can do it
 * (single phase, unit output stride, no split-K, not the 16-channel kernel). */
int rd_gconv_bnbwd_supported(const RdConvDesc* d);
int rd_gconv_bnbwd(const RdConvDesc* d, const float* dout, const float* w_packed, float* dx, const float* bn_x, int32_t ld_x,
                   const float* bn_mean, const float* bn_scale, const float* bn_shift, int32_t act, float* red_partial, float* ws,
                   void* stream);
/* Inference form (validate() body main.py:564-595):
 *   out = act_{co < act_cols}( conv(in, w) + bias[co] + addend )
 * with the eval-mode BatchNorm folded in: scale into the packed weights, shift = bias.  bias / addend / ws may be NULL. */
int rd_gconv_fused(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* bias,
                   int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* ws, void* stream);
/* bf16-operand form of rd_gconv / rd_gconv_fused (BASELINE.json configs 3/5; opt-in, the fp32 entry points above stay the
 * parity path): in / out / bias / addend are fp32 tensors exactly as above, the activations are rounded to bf16 (nearest even)
 * while they are staged, w_packed_bf16 is the bf16 operand written by rd_pack_weights_batched with quad == 2 -- element
 * (slab, ci, co) at ((slab*Cin/8 + ci/8)*ld + co)*8 + ci%8 -- and the reduction accumulates in fp32 on
 * v_mfma_f32_32x32x16_bf16.  stat_partial as in rd_gconv (rd_gconv_bf16_stat_tiles rows), may be NULL.  Replaces the
 * same F.conv2d call sites under torch.autocast(bfloat16) semantics (operands bf16, accumulate fp32). */
int rd_gconv_bf16(const RdConvDesc* d, const float* in, const void* w_packed_bf16, float* out, const float* bias,
                  int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, void* stream);
int rd_gconv_bf16_stat_tiles(const RdConvDesc* d);
/* diagnostics: out[0..7] = MT, NT, pipelined*1000 + CKP, TH, TW, patch pixels, lds_bytes, workgroups */
int rd_gconv_bf16_plan_info(const RdConvDesc* d, int32_t* out);
/* storage-typed forms: with dtype = RD_DTYPE_BF16 the unit-stride-input descriptors of 4..9 taps per phase (3x3 forward / input
 * gradient, the UpProj forward) run on the persistent pipelined kernel of csrc/gconv_bf16p.hip, whose tiling -- hence the number of
 * statistics rows rd_gconv_bf16_t writes -- differs from rd_gconv_bf16's: size the statistics buffer with this query when the
 * tensors are bf16.  plan_info_t: out[2] >= 2000 marks that kernel. */
int rd_gconv_bf16_stat_tiles_t(int32_t dtype, const RdConvDesc* d);
int rd_gconv_bf16_plan_info_t(int32_t dtype, const RdConvDesc* d, int32_t* out);
/* tests / sweeps: on != 0 makes the persistent kernel serve every shape it can run (also those a planner rule leaves to rd_gconv_bf16's
 * kernel because they measured slower); returns the previous setting.  Do not toggle between sizing buffers on a plan and launching. */
int rd_gconv_bf16p_plan_all(int32_t on);
/* diagnostics: with RD_GCONV_BF16_TRACE=1 every workgroup records cycle-counter stamps at its phase boundaries (32 slots per
 * workgroup: count, stamps); copies the last traced launch to the host (tools/trace_gconv_bf16.py) */
int rd_gconv_bf16_trace_read(unsigned long long* host, int n_wg);
/* fp32 convolution on the bf16 matrix cores (csrc/gconv_split.hip; opt-in, rd_gconv stays the default and the parity reference):
 * same tensors, descriptor and epilogue as rd_gconv_fused, but every fp32 operand is split into three bf16 pieces
 * (x = x0 + x1 + x2 exactly) and the product is rebuilt from the six bf16 MFMAs whose terms are not below 2^-24 of it, with fp32
 * accumulation: results as close to an fp64 convolution as rd_gconv's, at 2.67x the fp32 MFMA rate.  w_split: operand written by
 * rd_pack_weights_batched with quad == 3 -- three planes of the bf16 layout of rd_gconv_bf16, piece_elems elements apart.
 * rd_gconv_split_supported: 1 when the library has a plan for d (>= 32 channels on both sides, Cin a multiple of 16, a tile whose
 * three-piece patch fits the LDS), else 0 -- callers keep rd_gconv for those.  Replaces the F.conv2d / conv_transpose2d call sites
 * of models.py:27,203-206 and their autograd input gradients, like rd_gconv. */
/* Range: exact for every finite fp32 operand whose third piece is a normal bf16 number (|x| >= 2^-110; below that x2 falls into
 * bf16's subnormals and the product keeps ~16 significant bits, far below any activation of these networks).  Non-finite inputs:
 * x = +-inf gives x - bf16(x) = NaN, so an output that rd_gconv reports as +-inf is NaN here; NaN stays NaN.  Outputs that do not
 * touch the non-finite element are unaffected (tests/test_gpu_gconv_split.py::test_split_dynamic_range_and_non_finite). */
int rd_gconv_split_supported(const RdConvDesc* d);
/* tests / sweeps: on != 0 makes the planner accept every shape the kernel can run (also those it leaves to rd_gconv because they
 * measured slower); returns the previous setting.  Do not toggle between sizing buffers on a plan and launching it. */
int rd_gconv_split_plan_all(int on);
int rd_gconv_split(const RdConvDesc* d, const float* in, const void* w_split, int64_t piece_elems, float* out, const float* bias,
                   int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, void* stream);
int rd_gconv_split_stat_tiles(const RdConvDesc* d);
/* diagnostics: out[0..7] = MT, NT, TH, TW, patch pixels, lds_bytes, workgroups, tap groups of the largest phase */
int rd_gconv_split_plan_info(const RdConvDesc* d, int32_t* out);
/* Pre-split activations: an fp32 NHWC tensor x[M][C] (row stride ldx) as three bf16 piece planes, each [C/16][M][16], piece_elems
 * elements apart (x = p0 + p1 + p2 exactly).  rd_split_pieces is the stand-alone producer; in the training plan the BatchNorm /
 * activation kernels write the planes from their epilogues (rd_bn_act_p, rd_bn_bwd_apply*_p, rd_bnact_maxpool_fwd_p), so the split
 * arithmetic leaves the convolutions: rd_gconv_split_pre is rd_gconv_split with `in` replaced by such planes -- its staging waves
 * issue nothing but global_load_lds copies.  Same result bit for bit as rd_gconv_split on the tensor the planes were made from when
 * both run the same tile; the plan (hence rd_gconv_split_pre_stat_tiles) is its own. */
int rd_split_pieces(const float* x, int32_t ldx, int64_t M, int32_t C, void* pieces, int64_t piece_elems, void* stream);
/* rd_gconv_bnbwd's contract on the split kernels (round 6): the input-gradient launch of a conv -> BatchNorm -> act -> conv chain also
 * emits that BatchNorm's backward sums (red_partial [rd_gconv_split[_pre]_stat_tiles(d)][3][Cout]: slot 0 = sum g, slot 1 = sum g (x - mean),
 * g = dx * act'(scale x + shift)); no addend, four-channel alignment, not for one-tap descriptors.  pre != 0 asks about the pre-split form. */
int rd_gconv_split_bnbwd_supported(const RdConvDesc* d, int32_t pre);
int rd_gconv_split_bnbwd(const RdConvDesc* d, const float* in, const void* w_split, int64_t piece_elems, float* out, const float* bn_x,
                         int32_t bn_ld, const float* mean, const float* scale, const float* shift, int32_t bn_act, float* red_partial, void* stream);
int rd_gconv_split_pre_bnbwd(const RdConvDesc* d, const void* in_pieces, int64_t in_piece_elems, const void* w_split, int64_t piece_elems,
                             float* out, const float* bn_x, int32_t bn_ld, const float* mean, const float* scale, const float* shift,
                             int32_t bn_act, float* red_partial, void* stream);
int rd_gconv_split_pre_supported(const RdConvDesc* d);
/* 1 when the pre-split form is expected to be the fastest plan for d including its producer's extra piece pass (planner rule from the
 * measurements in profiles/r04_*): what engine.py asks before it routes a convolution through rd_gconv_split_pre */
int rd_gconv_split_pre_preferred(const RdConvDesc* d);
int rd_gconv_split_pre(const RdConvDesc* d, const void* in_pieces, int64_t in_piece_elems, const void* w_split, int64_t piece_elems, float* out,
                       const float* bias, int32_t act, int32_t act_cols, const float* addend, int32_t ld_add, float* stat_partial, void* stream);
int rd_gconv_split_pre_stat_tiles(const RdConvDesc* d);
int rd_gconv_split_pre_plan_info(const RdConvDesc* d, int32_t* out);
/* diagnostics / tests (host only, no GPU needed): the slot map of the split kernels -- which tile pixel each lane of an A fragment
 * reads.  The map places the 16 lanes of every ds_read_b128 pass on 16 different 16-byte LDS slots (csrc/gconv_split.hip,
 * gs_slot_pixel; tests/test_slot_map.py checks it against the lane groups of MI355X_MICROARCH.md).  pre != 0: the pre-split plan.
 * out[0..3] = slots per tile (BM), tile rows, tile columns, LDS row pitch of the patch in pixels; slots[m] = (r << 16) | c of the
 * tile pixel in slot m, -1 for an empty slot; n_slots >= BM. */
int rd_gconv_split_slot_map(const RdConvDesc* d, int32_t pre, int32_t phase, int32_t* out, int32_t* slots, int32_t n_slots);
/* diagnostics: with RD_GCONV_SPLIT_TRACE=1 (an MFMA wave) / =2 (a staging wave) every workgroup records cycle-counter stamps around
 * the barrier of its first 30 tap groups (64 slots per workgroup); copies the last traced launch to the host (tools/trace_gconv_split.py) */
int rd_gconv_split_trace_read(unsigned long long* host, int n_wg);
/* fp32 weight gradient on the bf16 matrix cores (csrc/wgrad_split.hip; opt-in like rd_gconv_split, rd_wgrad stays the default and the
 * parity reference): same tensors, slab layout and deterministic reduction as rd_wgrad / rd_wgrad_reduce; both operands are split
 * into three bf16 pieces while they are staged and every product is rebuilt from six bf16 MFMAs with fp32 accumulation.
 * rd_wgrad_split_supported: 1 for the full 3x3 / stride-1 descriptors with >= 64 channels on both sides (multiples of 8), else 0.
 * slabs: rd_wgrad_split_workspace_floats(d) floats.  Replaces the autograd weight gradients of the same F.conv2d call sites. */
int rd_wgrad_split_supported(const RdConvDesc* d);
int64_t rd_wgrad_split_workspace_floats(const RdConvDesc* d);
int rd_wgrad_split(const RdConvDesc* d, const float* in, const float* dout, float* slabs, void* stream);
int rd_wgrad_split_reduce(const RdConvDesc* d, const float* slabs, float* grad_oihw, int32_t O, int32_t I, int32_t KH, int32_t KW,
                          int32_t co_off, int32_t accumulate, void* stream);
/* diagnostics: out[0..3] = splits, tiles per split, workgroups, pixel tiles */
int rd_wgrad_split_plan_info(const RdConvDesc* d, int32_t* out);
/* rd_wgrad_split with both operands already split by their producers (piece planes as for rd_gconv_split_pre): same plan, slabs and
 * reduction (rd_wgrad_split_workspace_floats / rd_wgrad_split_reduce); channel counts multiples of 16. */
int rd_wgrad_split_pre_supported(const RdConvDesc* d);
int rd_wgrad_split_pre(const RdConvDesc* d, const void* x_pieces, int64_t x_piece_elems, const void* dy_pieces, int64_t dy_piece_elems,
                       float* slabs, void* stream);
/* diagnostics: out[0..9] = MT, NT, WM, WN, pipelined*10000+ksplit*100+CKW, CKP, TH, TW, lds_bytes, workgroups chosen for d
 * (workspace plan) */
int rd_gconv_plan_info(const RdConvDesc* d, int32_t* out);
/* diagnostics: workgroups per CU the HIP occupancy API reports for that plan (-1 without a GPU) */
int rd_gconv_occupancy(const RdConvDesc* d);
/* diagnostics: with RD_GCONV_TRACE=1 every workgroup of rd_gconv records cycle-counter stamps at its phase boundaries;
 * copies the 64 slots per workgroup of the last traced launch to the host (tools/trace_gconv.py). */
int rd_gconv_trace_read(unsigned long long* host, int n_wg);

/* Weight gradient of the same descriptor: dw[slab][ci][co] = sum_pixels in(...) * dout(...).
 * `d` is the FORWARD descriptor (in = forward input, "out" geometry = dout).  slabs is a
 * workspace of rd_wgrad_workspace_floats(d) floats; the result is reduced deterministically
 * into OIHW gradient tensors by rd_wgrad_reduce.  Replaces the weight half of
 * convolution_backward (autograd of models.py:96-112,203-206; 50% of the reference CPU step). */
int64_t rd_wgrad_workspace_floats(const RdConvDesc* d);
int rd_wgrad(const RdConvDesc* d, const float* in, const float* dout, float* slabs, void* stream);
/* grad_oihw[o][i][kh][kw] (+)= sum over slabs; o covers packed columns [co_off, co_off+O).
 * slab index of (kh,kw) is kh*KW+kw. */
int rd_wgrad_reduce(const RdConvDesc* d, const float* slabs, float* grad_oihw, int32_t O, int32_t I,
                    int32_t KH, int32_t KW, int32_t co_off, int32_t accumulate, void* stream);
/* diagnostics: out[0..8] = taps per group, MFMA size, layoutA, shared-B, LDS pitch, taps per row, per-phase launches, splits,
 * column-strip kernel */
int rd_wgrad_plan_info(const RdConvDesc* d, int32_t* out);

/* OIHW -> packed logical [slab][I][ldc] at column offset co_off (forward operand), or, with
 * transpose != 0, -> packed [slab][O.. as rows][I as columns] (dgrad operand: rows are the
 * forward output channels at row offset co_off, ldc >= I).  flip != 0 reverses the slab order
 * (kh,kw -> KH-1-kh, KW-1-kw).  The physical order interleaves the rows by four (see the
 * descriptor comment above); the row count must be a multiple of 4.  Replaces nothing in the
 * reference (layout glue). */
int rd_pack_weights(const float* w_oihw, float* packed, int32_t O, int32_t I, int32_t KH, int32_t KW,
                    int32_t ldc, int32_t co_off, int32_t rows_total, int32_t transpose, void* stream);

/* Batched form of rd_wgrad_reduce / rd_wgrad_bf16_reduce: the slab reductions of MANY weight-gradient launches (all of one
 * backward segment on one stream: the gradients are not needed before the segment's bucket boundary) in two launches instead
 * of one or two per weight tensor (98 launches of ~10 us per step at b=16 450x800).  rd_wgrad_reduce_job fills the HOST
 * record of one reduction (same arguments as rd_wgrad_reduce; bf16_kernel != 0: the slabs came from rd_wgrad_bf16); the caller
 * assigns consecutive block ranges (first_block1 / first_block2; n_blocks1 may be 0: no first stage), uploads the job array
 * and the two block -> job tables, and issues rd_wgrad_reduce_batched.  Column ranges of one slab set (the two 5x5
 * convolutions of an UpProj module) must appear in increasing co_off order, as with rd_wgrad_reduce.  Same summation order
 * as the per-tensor entry points: bit-identical results. */
typedef struct {
    const float* slabs;
    float* tmp;
    float* grad;
    int64_t E;                      /* floats per slab */
    int32_t n_splits, J, S, Cin, Cout, O, I, co_off, accumulate;
    int32_t first_block1, n_blocks1, first_block2, n_blocks2;
    int32_t pad_;
} RdReduceJob;
int rd_wgrad_reduce_job(const RdConvDesc* d, int32_t bf16_kernel, const float* slabs, float* grad_oihw, int32_t O, int32_t I,
                        int32_t KH, int32_t KW, int32_t co_off, int32_t accumulate, RdReduceJob* job);
int rd_wgrad_reduce_batched(const RdReduceJob* jobs_dev, const int32_t* block_job1_dev, int32_t n_blocks1,
                            const int32_t* block_job2_dev, int32_t n_blocks2, void* stream);

/* bf16-operand form of rd_wgrad / rd_wgrad_reduce for descriptors that decompose into at most four stride-1 3x3-shaped
 * passes over decimated tensors (3x3 / 1x1 at stride 1 or 2, the UpProj phases; channel counts multiples of 16;
 * rd_wgrad_bf16_supported says whether a descriptor qualifies -- everything else stays on rd_wgrad): in / dout are the fp32 tensors of rd_wgrad, rounded to bf16 (nearest even) while they are staged, accumulated
 * in fp32 on v_mfma_f32_32x32x16_bf16.  slabs: rd_wgrad_bf16_workspace_floats(d) floats; rd_wgrad_bf16_reduce sums them in a
 * fixed order into OIHW gradients exactly like rd_wgrad_reduce. */
int rd_wgrad_bf16_supported(const RdConvDesc* d);
int64_t rd_wgrad_bf16_workspace_floats(const RdConvDesc* d);
int rd_wgrad_bf16(const RdConvDesc* d, const float* in, const float* dout, float* slabs, void* stream);
int rd_wgrad_bf16_reduce(const RdConvDesc* d, const float* slabs, float* grad_oihw, int32_t O, int32_t I, int32_t KH,
                         int32_t KW, int32_t co_off, int32_t accumulate, void* stream);
/* diagnostics: out[0..6] = ci tiles per block, co tiles per block, channel blocks, pixel splits, slabs, lds bytes, passes */
int rd_wgrad_bf16_plan_info(const RdConvDesc* d, int32_t* out);
/* diagnostics: RD_WGRAD_BF16_TRACE=1 records cycle-counter stamps per workgroup (32 slots: count, stamps at: start, then per
 * tile: buffer ready, next tile's loads issued, MFMAs done, next tile written to LDS; end) -- tools/trace_wgrad_bf16.py */
int rd_wgrad_bf16_trace_read(unsigned long long* host, int n_wg);

/* bf16 operand of rd_gconv_bf16: same arguments, element (slab, row, col) at ((slab*R/8 + row/8)*ldc + col)*8 + row%8,
 * rounded to nearest even; the reduction dimension must be a multiple of 8. */
int rd_pack_weights_bf16(const float* w_oihw, void* packed_bf16, int32_t O, int32_t I, int32_t KH, int32_t KW,
                         int32_t ldc, int32_t co_off, int32_t rows_total, int32_t transpose, void* stream);

/* All weight tensors of a network in ONE launch.  jobs_dev: device array of
 *   struct { const float* src; float* dst; const float* scale; int32_t O, I, T(=KH*KW), ldc, off, rows_total,
 *            transpose, first_block, quad, pad; }
 *   scale (nullable): per-output-channel factor = folded BatchNorm scale (eval mode); quad 1: row-interleaved gconv
 *   operand layout, 2: bf16 operand of rd_gconv_bf16, 0: plain [slab][row][col] (the 7x7 stem kernels)
 * (same meaning as rd_pack_weights' arguments); block_job_dev[b] = job index of block b, where job j owns blocks
 * [first_block, first_block + ceil(O*I*T / rd_pack_chunk())).  Both arrays are built once by the host plan. */
int rd_pack_chunk(void);
int rd_pack_weights_batched(const void* jobs_dev, const int32_t* block_job_dev, int32_t n_blocks, void* stream);

/* ---------------------------------------------------------------------------------------
 * Stem convolutions: 7x7 stride 2 pad 3 read straight from the network's NCHW input
 * (models.py:539,559,633,643; multistage_model.py:163-164,236-241).  The Cin (1..3) input
 * channels are given as separate planes: planes[i] points at image 0 of channel i ([H,W] each),
 * strides[i] is the element distance between consecutive images (Ctot*H*W for a channel of the
 * network input, H*W for a stand-alone map such as the stage-1 prediction).  Both arrays live in
 * host memory and are copied at call time.  Weights are packed [49][Cin][Cout]
 * (rd_pack_weights).  Output NHWC [N,Ho,Wo,Cout], Ho = (H-1)/2+1.
 * ------------------------------------------------------------------------------------- */
int rd_stem_fwd(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                int32_t W, const float* w_packed, int32_t Cout, float* out, float* stat_partial,
                void* stream);
int rd_stem_stat_tiles(int32_t N, int32_t H, int32_t W);
/* bf16-operand form of rd_stem_fwd (opt-in with rd_gconv_bf16): identical arguments, tile geometry and stat_partial layout;
 * input planes and the packed fp32 weights are rounded to bf16 while they are staged, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16 (K ordered (plane, kernel row, kernel column padded to 8): no im2col buffer). */
int rd_stem_fwd_bf16(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                     int32_t W, const float* w_packed, int32_t Cout, float* out, float* stat_partial,
                     void* stream);
/* fp32 form of the same kernel for the split plans (rd_gconv_split's arithmetic): input planes and weights are split into three bf16
 * pieces while they are staged (x = x0 + x1 + x2 exactly), every product is rebuilt from six bf16 MFMAs with fp32 accumulation.  Same
 * arguments, tensors, tile geometry and stat_partial layout as rd_stem_fwd; as close to an fp64 convolution as rd_stem_fwd is
 * (tests/test_gpu_stem.py); range / non-finite behaviour as documented at rd_gconv_split. */
int rd_stem_fwd_split(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                      int32_t W, const float* w_packed, int32_t Cout, float* out, float* stat_partial,
                      void* stream);
/* The 16 -> 16 channel 3x3 unit-stride layers (depth encoder layer1, the last decoder stage's conv2; inside rd_gconv they run on the
 * 16x16x4 fp32 MFMA kernel of csrc/conv16.hip) with three-piece bf16 operands, six v_mfma_f32_16x16x32_bf16 per product, fp32
 * accumulation -- the split plans' form.  rd_gconv's contract for those descriptors: fp32 tensors, the fp32 quad-packed weight operand
 * of rd_pack_weights (forward or transposed), optional residual addend, optional BatchNorm partial sums with rd_gconv_stat_tiles_ws(d)
 * rows.  rd_conv16_split_supported: 1 for exactly those descriptors. */
int rd_conv16_split_supported(const RdConvDesc* d);
int rd_conv16_split(const RdConvDesc* d, const float* in, const float* w_packed, float* out, const float* addend, int32_t ld_add,
                    float* stat_partial, void* stream);
/* weight gradient (OIHW, overwritten) of the stem; ws needs rd_stem_wgrad_workspace_floats */
int64_t rd_stem_wgrad_workspace_floats(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int rd_stem_wgrad(const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                  int32_t W, const float* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream);
/* The same weight gradient on the bf16 matrix cores -- the split plans' (and the bf16-storage plans') form: the input planes and an
 * fp32 dout are split into three bf16 pieces while they are staged, six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation
 * (rd_gconv_split's arithmetic; a bf16 dout is its own single piece).  rd_stem_wgrad_t's contract: dtype is the element type of dout,
 * grad_oihw [Cout,Cin,7,7] is overwritten, ws needs rd_stem_wgrad_workspace_floats floats.  rd_stem_wgrad_split_supported: 1 for the
 * stems of the path (3 -> 64, 1 -> 16, 2 -> 16).  As close to an fp64 gradient as the fp32-MFMA kernel (tests/test_gpu_stem.py).
 * (reference: loss.backward() through models.py:627-631 conv1 / conv1_depth, main.py:440) */
int rd_stem_wgrad_split_supported(int32_t Cin, int32_t Cout);
int rd_stem_wgrad_split_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                          int32_t W, const void* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream);
/* The same with the stem BatchNorm's backward apply pass folded into its staging waves -- for a stem nobody asks an input gradient of
 * (the RGB stem; the depth stem outside the multistage network's second stage), whose BatchNorm input gradient only this kernel would
 * read.  g: gradient at the BatchNorm OUTPUT [N,Ho,Wo,Cout] (what rd_bnact_maxpool_bwd_stats_t stores); x: the stem's raw output;
 * red_partial / n_tiles: that call's partial sums; coef_ws: 3*Cout floats.  Equivalent to rd_bn_bwd_apply_t(g, x, ..., which = 1, dx)
 * followed by rd_stem_wgrad_split_t(dx): same dgamma / dbeta, same weight-gradient bits, no dx tensor (3 -> 64 and 1 -> 16 stems). */
int rd_stem_wgrad_split_bn_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                             int32_t W, const void* g, const void* x, const float* red_partial, int32_t n_tiles, const float* gamma,
                             const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, int32_t Cout,
                             float* grad_oihw, float* ws, void* stream);
/* input gradient w.r.t. ONE input channel `ci` of the stem (stage-2 dense-depth channel,
 * multistage_model.py:75 -- the stage-1 prediction is not detached).  dx is [N,H,W] (overwritten). */
int rd_stem_dgrad_channel(const float* dout, const float* w_packed, int32_t N, int32_t H, int32_t W,
                          int32_t Cin, int32_t ci, int32_t Cout, float* dx, void* stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm2d (training mode) + activation + residual join
 * (torch.nn.BatchNorm2d defaults eps=1e-5 momentum=0.1; models.py:101-110,203-208).
 * ------------------------------------------------------------------------------------- */
/* Reduce conv-epilogue partials [n_tiles][2][ld] over channels [c0, c0+C) -> mean / invstd, fused
 * scale = gamma*invstd and shift = beta - mean*scale; update running stats (unbiased var) and
 * num_batches_tracked (running_* / nbt may be NULL). */
int rd_bn_finalize(const float* stat_partial, int32_t n_tiles, int32_t ld, int32_t c0, int32_t C, int64_t count,
                   const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running stats */
int rd_bn_eval_coeffs(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, void* stream);
/* the same for many BatchNorm layers in one launch.  jobs_dev: device array of
 *   struct { const float *gamma, *beta, *running_mean, *running_var; float *scale, *shift; int32_t C, pad; } */
int rd_bn_eval_coeffs_batched(const void* jobs_dev, int32_t n_jobs, float eps, void* stream);
/* partial stats of an existing tensor (used where the producer has no fused epilogue) */
int rd_bn_stats(const float* x, int64_t M, int32_t C, int32_t ldx, float* stat_partial, int32_t* n_tiles,
                void* stream);
int rd_bn_stats_tiles(int64_t M);
/* y = act(scale1*x1 + shift1 [+ (scale2*x2 + shift2 | x2)]) ; scale2 == NULL -> identity residual */
int rd_bn_act(const float* x1, int32_t ldx1, const float* scale1, const float* shift1,
              const float* x2, int32_t ldx2, const float* scale2, const float* shift2,
              float* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* stream);
/* backward pass 1: g = dy * act'(y) (written to g, ldg); partial sums of g, g*(x1-mean1) [, g*(x2-mean2)]
 * -> red_partial [n_tiles][3][C]. */
int rd_bn_bwd_reduce(const float* dy, int32_t lddy, const float* y, int32_t ldy,
                     const float* x1, int32_t ldx1, const float* mean1,
                     const float* x2, int32_t ldx2, const float* mean2,
                     float* g, int32_t ldg, int64_t M, int32_t C, int32_t act,
                     float* red_partial, void* stream);
/* Same for out = act(scale1 * x1 + shift1) with no second operand: the activation's sign is recomputed from x1 (with the
 * forward kernel's own fmaf), so the activation tensor is not read -- one HBM pass less. */
int rd_bn_bwd_reduce_x(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* mean1,
                       const float* scale1, const float* shift1, float* g, int32_t ldg, int64_t M, int32_t C,
                       int32_t act, float* red_partial, void* stream);
/* row blocks (= rows of red_partial) the backward reductions of an [M, C] tensor produce */
int rd_bn_bwd_tiles(int64_t M, int32_t C);
/* backward pass 2 (per BN): finishes the reduction, writes dgamma/dbeta (overwrite) and
 * dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).  which = 1 or 2 selects the x1 / x2 sums. */
int rd_bn_bwd_apply(const float* g, int32_t ldg, const float* x, int32_t ldx, const float* red_partial,
                    int32_t n_tiles, int32_t which, const float* gamma, const float* mean,
                    const float* invstd, float* dgamma, float* dbeta, float* coef_ws /* 3*C floats */,
                    float* dx, int32_t lddx, int64_t M, int32_t C, void* stream);
/* Apply pass paired with rd_bn_bwd_reduce_x(g = NULL): dy is the raw output gradient of act(scale*x + shift); the activation
 * factor is recomputed from x here, so the masked gradient is never written to HBM. */
int rd_bn_bwd_apply_x(const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles,
                      const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift,
                      int32_t act, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C,
                      void* stream);
/* out = act(bn1(x1) + bn2(x2)) (down-sampling blocks, UpProj joins): the same two passes for both operands at once -- the
 * activation output is not read, the masked gradient is not written, dy is read once per pass.  red_partial as for
 * rd_bn_bwd_reduce ([tiles][3][C]); coef_ws6: 6*C floats. */
int rd_bn_bwd_reduce_x2(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* mean1, const float* scale1,
                        const float* shift1, const float* x2, int32_t ldx2, const float* mean2, const float* scale2,
                        const float* shift2, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream);
int rd_bn_bwd_apply_x2(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* x2, int32_t ldx2,
                       const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1,
                       const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2,
                       const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2,
                       float* dbeta2, float* coef_ws6, float* dx1, int32_t lddx1, float* dx2, int32_t lddx2, int64_t M,
                       int32_t C, void* stream);

/* MaxPool2d(3,2,1) fused with the stem's BN affine + activation (models.py:634-636,644-646):
 * y = maxpool(act(scale*x+shift)); idx = argmax position 0..8 in the window. */
int rd_bnact_maxpool_fwd(const float* x, const float* scale, const float* shift, int32_t act,
                         int32_t N, int32_t H, int32_t W, int32_t C, float* y, int32_t ldy,
                         uint8_t* idx, void* stream);
/* gradient w.r.t. the pre-activation BN output: g[n,h,w,c] = act'(scale*x+shift) *
 * sum of dy over the pooling windows whose argmax is (h,w). */
int rd_bnact_maxpool_bwd(const float* dy, int32_t lddy, const uint8_t* idx, const float* x,
                         const float* scale, const float* shift, int32_t act, int32_t N, int32_t H,
                         int32_t W, int32_t C, float* g, void* stream);
/* Same pass, plus the stem BatchNorm's backward sums (sum g, sum g*(x - mean)) as per-block partials
 * [rd_bnact_maxpool_bwd_tiles(N,H,W,C)][3][C] for rd_bn_bwd_apply(which = 1): g and x are in registers here, so the separate
 * rd_bn_bwd_reduce pass over the two largest tensors of the network is not needed.  C/4 must divide 256. */
int rd_bnact_maxpool_bwd_tiles(int32_t N, int32_t H, int32_t W, int32_t C);
int rd_bnact_maxpool_bwd_stats(const float* dy, int32_t lddy, const uint8_t* idx, const float* x, const float* scale,
                               const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, float* g,
                               const float* mean, float* red_partial, void* stream);

/* ---------------------------------------------------------------------------------------
 * Head: conv3 (3x3, C->1, models.py:587,661) and bilinear align_corners=True resize
 * (models.py:588,662).  d is NHWC-1 / NCHW-1 (same memory) [N,Hs,Ws]; out is [N,1,Ho,Wo].
 * ------------------------------------------------------------------------------------- */
int rd_head_conv_fwd(const float* x, int32_t ldx, const float* w_oihw, int32_t N, int32_t H, int32_t W,
                     int32_t C, float* d, void* stream);
int rd_head_conv_bwd(const float* x, int32_t ldx, const float* w_oihw, const float* dd, int32_t N,
                     int32_t H, int32_t W, int32_t C, float* dx, int32_t lddx, float* dw_oihw,
                     float* ws, void* stream);
int64_t rd_head_conv_bwd_workspace_floats(int32_t N, int32_t H, int32_t W, int32_t C);
int rd_bilinear_fwd(const float* d, int32_t N, int32_t Hs, int32_t Ws, float* out, int32_t Ho, int32_t Wo,
                    void* stream);
int rd_bilinear_bwd(const float* dout, int32_t N, int32_t Ho, int32_t Wo, float* dd, int32_t Hs, int32_t Ws,
                    void* stream);

/* ---------------------------------------------------------------------------------------
 * Losses (evaluation/criteria_new.py) and the step tail (main.py:416-445).
 * ------------------------------------------------------------------------------------- */
/* MaskedL1Loss (:44-54): sums[0] = sum |t-p| over t>0, sums[1] = count.  ws: 2*rd_loss_tiles(n) DOUBLES
 * (8-byte aligned), passed as float*. */
int rd_masked_l1_sums(const float* pred, const float* target, int64_t n, float* ws, double* sums, void* stream);
int rd_loss_tiles(int64_t n);
/* dpred (+)= coef_dev[0] * (-sign(t-p)) / count on valid pixels; coef read on device */
int rd_masked_l1_bwd(const float* pred, const float* target, int64_t n, const double* sums,
                     const float* coef, float* dpred, int32_t accumulate, void* stream);
/* MaskedMSELoss (:31-41, `-c l2`, main.py:294-305): sums[0] = sum (t-p)^2 over t>0, sums[1] = count; same workspace.
 * dpred (+)= coef_dev[0] * 2 (p-t) / count on valid pixels. */
int rd_masked_l2_sums(const float* pred, const float* target, int64_t n, float* ws, double* sums, void* stream);
int rd_masked_l2_bwd(const float* pred, const float* target, int64_t n, const double* sums,
                     const float* coef, float* dpred, int32_t accumulate, void* stream);
/* Result.evaluate (evaluation/metrics.py:34-58) as ONE masked reduction instead of ~12 blocking float() syncs:
 * sums[10] = count, sum d^2, sum |d|, sum |log10 o - log10 t|, sum |d|/t, #(r<1.25), #(r<1.25^2), #(r<1.25^3),
 * sum (1/o-1/t)^2, sum |1/o-1/t| over pixels with target > 0.  ws: 10*rd_loss_tiles(n) doubles. */
int rd_depth_metrics(const float* output, const float* target, int64_t n, float* ws, double* sums, void* stream);
/* SmoothnessLoss (:8-28) on pred [N,1,H,W] and image [N,C,H,W] (NCHW).  out[0] = loss.
 * ws: rd_smooth_workspace_floats(N,H,W) floats, 8-byte aligned; rd_smooth_bwd reuses what fwd left there. */
int64_t rd_smooth_workspace_floats(int32_t N, int32_t H, int32_t W);
int rd_smooth_fwd(const float* pred, const float* image, int32_t N, int32_t C, int32_t H, int32_t W,
                  float* ws, double* out, void* stream);
int rd_smooth_bwd(int32_t N, int32_t H, int32_t W, const float* ws, const float* coef, float* dpred,
                  int32_t accumulate, void* stream);
/* Filter_layer (multistage_model.py:87-119): kept = sparse*mask, mask = |dense-sparse| <= 5*3.6^(dense/100).
 * sparse is channel `c` of the NCHW input x [N,Ctot,H,W]. */
int rd_radar_filter(const float* x, int32_t N, int32_t Ctot, int32_t c, int64_t hw, const float* dense,
                    float* kept, float* mask, void* stream);
/* uncertainty-weighted total (main.py:423-429).  in: l1 sums of both stages, smoothness, w1, w2.
 * out_host-visible device scalars: loss[0..3] = d1, d2, smooth, total; coefs[0..2] = e^-w1, 0.1*e^-w1,
 * e^-w2 (the factors the loss backward kernels read); dw[0..1] = gradients of w1, w2. */
int rd_uncertainty_total(const double* sums1, const double* sums2, const double* smooth, const float* w1,
                         const float* w2, float w_smooth, float* loss4, float* coefs3, float* dw1,
                         float* dw2, void* stream);
/* loss[0] = sums[0]/sums[1]; coef[0] = 1 (plain MaskedL1 step, main.py:440-441) */
int rd_l1_total(const double* sums, float* loss, float* coef, void* stream);

/* torch.optim.SGD step (main.py:285-290,445): g += wd*p; buf = first ? g : mom*buf + g; p -= lr*buf,
 * over a flat arena.  grad_scale multiplies g first (1/world for data-parallel averaging). */
int rd_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd,
                float grad_scale, int32_t first_step, void* stream);
int rd_fill(float* p, int64_t n, float v, void* stream);
/* Input staging from the exported .h5 frames.  Replaces the depth decompression of
 * nuscenes_dataset_torch.get_data (dataset/nuscenes_dataset_torch_new.py:191-195) and the CenterCrop -> /255 -> ToTensor ->
 * max-depth clamp -> cat of transform_val (same file :415-455, :503-512; CenterCrop dataset/transforms.py:332-385):
 *   inputs [B,4,H,W] fp32 NCHW = (rgb/255, radar/256 with values > max_depth zeroed), labels [B,1,H,W] = lidar/256,
 * both cropped at (i0, j0) out of the H0 x W0 frames.  rgb: uint8 [B,H0,W0,3]; lidar, radar: int16 [B,H0,W0] (depth*256).
 * Bit-exact with the reference arithmetic.  max_depth = +inf disables the clamp (main.py:71). */
int rd_stage_frames(const uint8_t* rgb_hwc, const int16_t* lidar, const int16_t* radar, int32_t B, int32_t H0, int32_t W0,
                    int32_t i0, int32_t j0, int32_t H, int32_t W, float max_depth, float* inputs_nchw4, float* labels,
                    void* stream);

/* NCHW [N,C,H,W] (channel c0..c0+C of Ctot) <-> NHWC helpers for module-level tests */
int rd_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int rd_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) form of the 3x3 / stride-1 / pad-1 convolutions with >= 64 channels a side, on the split-bf16 pipeline
 * (csrc/wino_split.hip): 16 instead of 36 products per 2x2 output tile, each product rebuilt from six bf16 MFMA terms like
 * rd_gconv_split.  Replaces F.conv2d / its input gradient for the BasicBlock and UpProj conv2 layers
 * (/root/reference/model/models.py:96-112,203-206; cuDNN's own Winograd under cudnn.benchmark, /root/reference/main.py:11,47).
 *   rd_wino_pack      : U = G g G^T of an OIHW [O][I][3][3] tensor (fp64, rounded once to fp32, three bf16 pieces) in the kernel's
 *                       copy layout; flip = 1 packs the input-gradient operand (channels transposed, taps rotated by 180 degrees).
 *                       u_packed: rd_wino_packed_bytes(O, I, flip) bytes.
 *   rd_wino_conv3x3   : out[N,H,W,Cout (stride ldo)] = conv3x3(in[N,H,W,Cin (stride ldi)]) (+ addend); stat_partial (may be NULL):
 *                       [rd_wino_stat_tiles(N,H,W)][2][Cout] per-tile sum / sum of squares of the stored values, the BatchNorm
 *                       statistics contract of rd_gconv.  fp32 NHWC tensors; Cin % 16 == 0, Cout % 64 == 0, both >= 64. */
int64_t rd_wino_packed_bytes(int32_t O, int32_t I, int32_t flip);
int rd_wino_pack(const float* w_oihw, int32_t O, int32_t I, int32_t flip, void* u_packed, void* stream);
int rd_wino_supported(int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ldi, int32_t ldo);
/* planner rule from the kernel-level measurements (1: the Winograd form is expected to beat rd_gconv_split[_pre] for this layer) */
int rd_wino_preferred(int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ldi, int32_t ldo);
/* every Winograd operand of a plan in one launch: jobs = device array of 32-byte records { const float* w; void* u; int32 O, I, flip,
 * first_block }, block_job[b] = the job of block b, a job takes rd_wino_pack_blocks(O, I, flip) consecutive blocks from first_block */
int rd_wino_pack_blocks(int32_t O, int32_t I, int32_t flip);
int rd_wino_pack_batched(const void* jobs, const int32_t* block_job, int32_t n_blocks, void* stream);
int rd_wino_stat_tiles(int32_t N, int32_t H, int32_t W);
int rd_wino_conv3x3(const float* in, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldi, const void* u_packed, float* out,
                    int32_t Cout, int32_t ldo, const float* addend, int32_t ld_add, float* stat_partial, void* stream);
/* the input gradient of a Winograd layer + the backward sums of the BatchNorm in front of the convolution in one launch: rd_gconv_bnbwd's
 * contract (bn_x = that BatchNorm's input [N,H,W,Cout], channel stride bn_ld; red_partial [rd_wino_stat_tiles][3][Cout]: sum g,
 * sum g (x - mean), g = dx * act'(scale x + shift)).  Replaces the rd_bn_bwd_reduce_x_t pass of a conv -> BN -> act -> conv chain. */
int rd_wino_conv3x3_bnbwd(const float* in, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t ldi, const void* u_packed, float* out,
                          int32_t Cout, int32_t ldo, const float* bn_x, int32_t bn_ld, const float* mean, const float* scale, const float* shift,
                          int32_t bn_act, float* red_partial, void* stream);

/* Diagnostics (bench.py `roofline.shader_clock_mhz`): n_blocks (<= 64) one-wave workgroups each sleep for duration_us and write
 * out[4 * block + 0..3] = { shader clocks elapsed (s_memtime), 10-ns ticks elapsed (s_memrealtime), XCC_ID, start tick }:
 * clocks / ticks * 100 = the effective shader clock in MHz of that workgroup's XCD over the window.  Meant to run on a stream of
 * its own beside the training step (the sleeping wave takes no issue slots).  `out`: device memory, 32 * n_blocks bytes.
 * No counterpart in the reference (measurement only). */
int rd_clock_probe(unsigned long long* out, int32_t n_blocks, int32_t duration_us, void* stream);

/* Fork/join between the caller's streams (the plan runs the depth encoder and the weight-gradient chains on side streams);
 * record/wait pairs are captured as graph edges when the main stream is being captured. */
int rd_event_create(void** event);
int rd_event_destroy(void* event);
int rd_event_record(void* event, void* stream);
int rd_stream_wait_event(void* stream, void* event);

/* ---------------------------------------------------------------------------------------
 * Data-parallel exchange step (the reference itself is single-process, main.py:285-447, so these replace
 * what a DistributedDataParallel wrapper around its model would do): one process per GPU, RCCL over xGMI.
 *   rank 0:      rd_comm_unique_id(token)            -> ship the 128-byte token to every rank (any out-of-band channel)
 *   every rank:  hipSetDevice; rd_comm_init(token, rank, world)                       (collective)
 *   per step:    after the last backward kernel of a gradient bucket: record an event on the compute stream, make the
 *                communication stream wait for it, rd_allreduce_bucket(ptr, count, RD_DTYPE_F32, comm_stream); before the
 *                optimizer step the compute stream waits for an event recorded behind the last bucket.  In-place sum; the
 *                1/world average is folded into rd_sgd_step's grad_scale.  Nothing synchronises the host.
 * RCCL is loaded at run time (dlopen): a process that already holds one (PyTorch's) shares it.
 * ------------------------------------------------------------------------------------- */
#define RD_DTYPE_F32 0
#define RD_DTYPE_BF16 1
int rd_comm_unique_id(void* out128);
int rd_comm_init(const void* unique_id128, int32_t rank, int32_t world);
int rd_comm_world(void); /* 0 before rd_comm_init */
int rd_comm_rank(void);
int rd_allreduce_bucket(void* ptr, int64_t count, int32_t dtype, void* stream);
int rd_broadcast(void* ptr, int64_t count, int32_t dtype, int32_t root, void* stream);
int rd_comm_destroy(void);

/* ---------------------------------------------------------------------------------------
 * Storage-typed forms (bf16-storage plans, BASELINE.json configs 3 / 5): the same operations with the NHWC activation /
 * gradient tensors stored as fp32 (dtype = RD_DTYPE_F32) or bf16 (RD_DTYPE_BF16) in HBM.  Arithmetic is fp32 in both cases
 * (bf16 values are widened on load, results rounded to nearest-even on store); per-channel statistics, coefficients, weights'
 * gradients and every reduction stay fp32 / fp64.  Strides are in elements.  The argument lists are those of the fp32 entry
 * points above with `dtype` in front and the tensors as void pointers.
 * ------------------------------------------------------------------------------------- */
int rd_bn_stats_t(int32_t dtype, const void* x, int64_t M, int32_t C, int32_t ldx, float* stat_partial, int32_t* n_tiles, void* stream);
int rd_bn_act_t(int32_t dtype, const void* x1, int32_t ldx1, const float* scale1, const float* shift1, const void* x2, int32_t ldx2, const float* scale2, const float* shift2, void* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* stream);
int rd_bn_bwd_reduce_t(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x1, int32_t ldx1, const float* mean1, const void* x2, int32_t ldx2, const float* mean2, void* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream);
int rd_bn_bwd_reduce_x_t(int32_t dtype, const void* dy, int32_t lddy, const void* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, void* g, int32_t ldg, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream);
int rd_bn_bwd_reduce_x2_t(int32_t dtype, const void* dy, int32_t lddy, const void* x1, int32_t ldx1, const float* mean1, const float* scale1, const float* shift1, const void* x2, int32_t ldx2, const float* mean2, const float* scale2, const float* shift2, int64_t M, int32_t C, int32_t act, float* red_partial, void* stream);
int rd_bn_bwd_apply_t(int32_t dtype, const void* g, int32_t ldg, const void* x, int32_t ldx, const float* red_partial, int32_t n_tiles, int32_t which, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, void* dx, int32_t lddx, int64_t M, int32_t C, void* stream);
int rd_bn_bwd_apply_x2_t(int32_t dtype, const void* dy, int32_t lddy, const void* x1, int32_t ldx1, const void* x2, int32_t ldx2, const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1, const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2, const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* coef_ws6, void* dx1, int32_t lddx1, void* dx2, int32_t lddx2, int64_t M, int32_t C, void* stream);
int rd_bn_bwd_apply_x_t(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift, int32_t act, float* dgamma, float* dbeta, float* coef_ws, void* dx, int32_t lddx, int64_t M, int32_t C, void* stream);
int rd_bnact_maxpool_fwd_t(int32_t dtype, const void* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, void* y, int32_t ldy, uint8_t* idx, void* stream);
/* fp32 forms that ALSO write the result as pre-split piece planes (see rd_split_pieces: [piece][C/16][M][16] bf16, piece_elems elements
 * apart; pieces may be NULL = the plain kernel) -- the producers of the operands of rd_gconv_split_pre / rd_wgrad_split_pre: the split
 * arithmetic runs in these HBM-bound passes, where the VALU is idle, instead of in the staging waves of the MFMA-bound convolutions.
 * Same arguments as rd_bn_act / rd_bn_bwd_apply / rd_bn_bwd_apply_x / rd_bn_bwd_apply_x2 / rd_bnact_maxpool_fwd otherwise
 * (models.py:96-112,203-208,633-650 and their backward). */
int rd_bn_act_p(const float* x1, int32_t ldx1, const float* scale1, const float* shift1, const float* x2, int32_t ldx2, const float* scale2, const float* shift2, float* y, int32_t ldy, int64_t M, int32_t C, int32_t act, void* pieces, int64_t piece_elems, void* stream);
int rd_bn_bwd_apply_p(const float* g, int32_t ldg, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles, int32_t which, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C, void* pieces, int64_t piece_elems, void* stream);
int rd_bn_bwd_apply_x_p(const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* red_partial, int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift, int32_t act, float* dgamma, float* dbeta, float* coef_ws, float* dx, int32_t lddx, int64_t M, int32_t C, void* pieces, int64_t piece_elems, void* stream);
int rd_bn_bwd_apply_x2_p(const float* dy, int32_t lddy, const float* x1, int32_t ldx1, const float* x2, int32_t ldx2, const float* red_partial, int32_t n_tiles, const float* gamma1, const float* mean1, const float* invstd1, const float* scale1, const float* shift1, const float* gamma2, const float* mean2, const float* invstd2, const float* scale2, const float* shift2, int32_t act, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, float* coef_ws6, float* dx1, int32_t lddx1, float* dx2, int32_t lddx2, int64_t M, int32_t C, void* pieces1, int64_t piece_elems1, void* pieces2, int64_t piece_elems2, void* stream);
int rd_bnact_maxpool_fwd_p(const float* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, float* y, int32_t ldy, uint8_t* idx, void* pieces, int64_t piece_elems, void* stream);
int rd_bnact_maxpool_bwd_t(int32_t dtype, const void* dy, int32_t lddy, const uint8_t* idx, const void* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, void* g, void* stream);
int rd_bnact_maxpool_bwd_stats_t(int32_t dtype, const void* dy, int32_t lddy, const uint8_t* idx, const void* x, const float* scale, const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, void* g, const float* mean, float* red_partial, void* stream);
/* Two-pass form of the stem's pool + BatchNorm backward that never materialises the full-resolution gradient g (the largest tensor
 * of the network; models.py:633-650 backward): pass 1 = rd_bnact_maxpool_bwd_stats_t with g == NULL (sums only), pass 2 = this call:
 * finishes the sums into dgamma / dbeta / the dx coefficients and repeats the pool gather, storing dx = A*g + B*(x - mean) + K.
 * Saves one write and one read of [N,H,W,C] against one more read of the quarter-size pooled gradient and its argmax bytes. */
int rd_bnact_maxpool_bwd_apply_t(int32_t dtype, const void* dy, int32_t lddy, const uint8_t* idx, const void* x, const float* scale,
                                 const float* shift, int32_t act, int32_t N, int32_t H, int32_t W, int32_t C, const float* red_partial,
                                 int32_t n_tiles, const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                 float* coef_ws, void* dx, void* stream);
int rd_gconv_bf16_t(int32_t dtype, const RdConvDesc* d, const void* in, const void* w_packed_bf16, void* out, const float* bias,
                    int32_t act, int32_t act_cols, const void* addend, int32_t ld_add, float* stat_partial, void* stream);
int rd_wgrad_bf16_t(int32_t dtype, const RdConvDesc* d, const void* in, const void* dout, float* slabs, void* stream);
int rd_stem_fwd_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                  const float* w_packed, int32_t Cout, void* out, float* stat_partial, void* stream);
int rd_stem_fwd_bf16_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H,
                       int32_t W, const float* w_packed, int32_t Cout, void* out, float* stat_partial, void* stream);
int rd_stem_wgrad_t(int32_t dtype, const float* const* planes, const int64_t* strides, int32_t Cin, int32_t N, int32_t H, int32_t W,
                    const void* dout, int32_t Cout, float* grad_oihw, float* ws, void* stream);
int rd_stem_dgrad_channel_t(int32_t dtype, const void* dout, const float* w_packed, int32_t N, int32_t H, int32_t W, int32_t Cin,
                            int32_t ci, int32_t Cout, float* dx, void* stream);
int rd_head_conv_fwd_t(int32_t dtype, const void* x, int32_t ldx, const float* w_oihw, int32_t N, int32_t H, int32_t W, int32_t C,
                       float* d, void* stream);
int rd_head_conv_bwd_t(int32_t dtype, const void* x, int32_t ldx, const float* w_oihw, const float* dd, int32_t N, int32_t H,
                       int32_t W, int32_t C, void* dx, int32_t lddx, float* dw_oihw, float* ws, void* stream);
/* the two halves of rd_head_conv_bwd_t as separate calls (same kernels): the input gradient is what the rest of the backward waits for,
 * the weight gradient (+ its slab reduction; ws as for rd_head_conv_bwd) may run on another stream */
int rd_head_conv_dgrad_t(int32_t dtype, const float* w_oihw, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C, void* dx,
                         int32_t lddx, void* stream);
int rd_head_conv_wgrad_t(int32_t dtype, const void* x, int32_t ldx, const float* dd, int32_t N, int32_t H, int32_t W, int32_t C,
                         float* dw_oihw, float* ws, void* stream);

/* Plan tuner for rd_gconv / rd_gconv_ws (the role cudnn.benchmark plays for the reference's convolutions): list the candidate
 * execution plans of a descriptor (9 ints each: MT, NT, WM, WN, CKP, TH, TW, ksplit, loop form -- 0 plain, 1 software-pipelined,
 * 2 pipelined with half-depth weight slabs (more resident workgroups, more barriers); best heuristic score first; returns
 * the count), pin one -- every later call with that descriptor, workspace / statistics-tile queries included, uses it -- or pin
 * NULL to return to the heuristic.  The caller times rd_gconv_ws under each pinned candidate (radar_depth_amd/autotune.py).
 * allow_split: 1 for the rd_gconv_ws form (with workspace), 0 for rd_gconv. */
int rd_gconv_tune_candidates(const RdConvDesc* d, int32_t allow_split, int32_t* out, int32_t max_candidates);
int rd_gconv_tune_pin(const RdConvDesc* d, int32_t allow_split, const int32_t* candidate);
/* rd_gconv_tune_commit: the pinned (or, if none, the heuristic) plan of d is final -- call it after the last pin and BEFORE sizing
 * statistics tiles / workspaces on the plan; afterwards rd_gconv_tune_candidates returns 0 and rd_gconv_tune_pin refuses for d.
 * rd_gconv_plan_state: 0 not planned, 1 pinned and still replaceable, 2 in use. */
int rd_gconv_tune_commit(const RdConvDesc* d, int32_t allow_split);
int rd_gconv_plan_state(const RdConvDesc* d, int32_t allow_split);

/* diagnostics: fill every CU's LDS with NaN bit patterns (LDS is not cleared between kernels): a kernel that consumes an LDS
 * word it never wrote then yields NaN instead of depending on its predecessor's leftovers (tools/fuzz_conv.py --poison) */
int rd_debug_poison_lds(void* stream);

/* hipGraph capture helpers so a whole step replays without host launch cost */
int rd_graph_begin(void* stream);
int rd_graph_end(void* stream, void** graph_exec);
int rd_graph_launch(void* graph_exec, void* stream);
int rd_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------------------------------
 * Op-table replay: the step body of main.py:416-445 (forward, loss, zero_grad, backward, [gradient exchange], SGD) is a STATIC
 * list of the entry points above over fixed buffers.  A table holds that list pre-marshalled -- per op the entry point's name
 * and its arguments as 64-bit words (pointers / integers by value, a float as its IEEE bit pattern in the low 32 bits) -- and
 * rd_optable_run issues ops [begin, end) with ONE call from the host language: no per-launch FFI marshalling, nothing for
 * eight per-GPU ranks to contend for on the host.  Arguments that are streams are marked with a slot index
 * (stream_slots[i] >= 0, else -1) and take streams[slot] of each run, so a table survives stream rebinding and can be
 * replayed under rd_graph_begin / rd_graph_end.  Descriptor / table arguments are pointers into caller-owned memory that must
 * outlive the table.  There is no second implementation of any op: a run calls exactly the functions named.
 *   rd_optable_add      -> index of the new op (>= 0) or a negative error (unknown entry point, wrong argument count)
 *   rd_optable_entry_args(name) -> that entry point's argument count, RD_EINVAL if it is not replayable
 *   rd_optable_run      -> 0, or the failing op's return code with *failed_op set (ops behind it are not issued)
 * A table is not thread-safe: one table, one issuing thread.
 * ------------------------------------------------------------------------------------- */
int rd_optable_create(void** table);
int rd_optable_destroy(void* table);
int rd_optable_entry_args(const char* entry);
int rd_optable_add(void* table, const char* entry, int32_t nargs, const uint64_t* words, const int32_t* stream_slots);
int rd_optable_size(const void* table);
int rd_optable_set_word(void* table, int32_t op, int32_t arg, uint64_t word);
int rd_optable_run(void* table, int32_t begin, int32_t end, void* const* streams, int32_t n_streams, int32_t* failed_op);
/* The same replay from n_lanes host threads (1..8): the ops of stream slot s are issued, in table order, by lane s % n_lanes; an
 * rd_stream_wait_event is issued only after the rd_event_record it pairs with (the last record of that event before it in the table) has
 * been issued by its lane.  Lanes 1.. are persistent threads of the library.  Not under stream capture.  (The step is 480-960 launches on
 * three streams; the HIP launch path costs 10-16 us of host time per op.) */
int rd_optable_run_mt(void* table, int32_t begin, int32_t end, void* const* streams, int32_t n_streams, int32_t n_lanes, int32_t* failed_op);

#ifdef __cplusplus
}
#endif
#endif
