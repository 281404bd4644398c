"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/radar_depth_hip.h).
Tensors must live on the GPU; all calls are asynchronous on torch's current HIP stream."""
import ctypes as C
import os as _os

import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), "expected contiguous CUDA float32 tensor"
    return t


def pack_weights(w_oihw, transpose=False, out=None, ldc=None, off=0, rows_total=None):
    """OIHW -> [slab][I][ldc] (forward operand) or [slab][rows_total][ldc>=I] (dgrad operand)."""
    o, i, kh, kw = w_oihw.shape
    if not transpose:
        ldc = ldc or o
        rows_total = i
        shape = (kh * kw, i, ldc)
    else:
        ldc = ldc or i
        rows_total = rows_total or o
        shape = (kh * kw, rows_total, ldc)
    if out is None:
        out = torch.zeros(shape, dtype=torch.float32, device=w_oihw.device)
    check(lib().rd_pack_weights(ptr(_f32(w_oihw)), ptr(out), o, i, kh, kw, ldc, off, rows_total, int(transpose),
                                current_stream()), "rd_pack_weights")
    return out


def pack_weights_bf16(w_oihw, transpose=False, ldc=None, off=0, rows_total=None):
    """bf16 operand of gconv_bf16 (same logical shapes as pack_weights)."""
    o, i, kh, kw = w_oihw.shape
    if not transpose:
        ldc, rows_total = ldc or o, i
    else:
        ldc, rows_total = ldc or i, rows_total or o
    out = torch.zeros((kh * kw, rows_total, ldc), dtype=torch.bfloat16, device=w_oihw.device)
    check(lib().rd_pack_weights_bf16(ptr(_f32(w_oihw)), ptr(out), o, i, kh, kw, ldc, off, rows_total, int(transpose),
                                     current_stream()), "rd_pack_weights_bf16")
    return out


def gconv_bf16(desc, x, w_packed_bf16, out, bias=None, act=0, act_cols=0, addend=None, ld_add=0, stat=None):
    """bf16-operand convolution (fp32 tensors, fp32 accumulation): rd_gconv_bf16."""
    _poison()
    assert w_packed_bf16.dtype == torch.bfloat16
    check(lib().rd_gconv_bf16(C.byref(desc), ptr(_f32(x)), ptr(w_packed_bf16), ptr(_f32(out)), ptr(bias), act, act_cols,
                              ptr(addend), ld_add, ptr(stat), current_stream()), "rd_gconv_bf16")
    return out


def pack_weights_split(w_oihw, transpose=False, ldc=None, off=0, rows_total=None):
    """Three-piece bf16 operand of gconv_split: [piece][slab][rows][ldc], w = p0 + p1 + p2 exactly (each piece packed like
    pack_weights_bf16; the pieces are bf16-representable, so packing them is exact)."""
    h = w_oihw.to(torch.bfloat16).float()
    r = w_oihw - h
    m = r.to(torch.bfloat16).float()
    lo = (r - m).to(torch.bfloat16).float()
    return torch.stack([pack_weights_bf16(p.contiguous(), transpose, ldc, off, rows_total) for p in (h, m, lo)]).contiguous()


def gconv_split_supported(desc):
    return lib().rd_gconv_split_supported(C.byref(desc)) == 1


def gconv_split(desc, x, w_split, out, bias=None, act=0, act_cols=0, addend=None, ld_add=0, stat=None):
    """fp32 convolution rebuilt from six bf16 MFMAs per product (three-piece operands): rd_gconv_split."""
    _poison()
    assert w_split.dtype == torch.bfloat16 and w_split.shape[0] == 3
    check(lib().rd_gconv_split(C.byref(desc), ptr(_f32(x)), ptr(w_split), C.c_int64(w_split[0].numel()), ptr(_f32(out)), ptr(bias), act,
                               act_cols, ptr(addend), ld_add, ptr(stat), current_stream()), "rd_gconv_split")
    return out


def split_pieces(x_nhwc):
    """fp32 NHWC tensor -> three bf16 piece planes [3][C/16][M][16] (rd_split_pieces; x = p0 + p1 + p2 exactly)."""
    import torch
    n, h, w, c = x_nhwc.shape
    m = n * h * w
    pc = torch.empty(3, c // 16, m, 16, dtype=torch.bfloat16, device=x_nhwc.device)
    check(lib().rd_split_pieces(ptr(_f32(x_nhwc)), c, C.c_int64(m), c, ptr(pc), C.c_int64(pc[0].numel()), current_stream()), "rd_split_pieces")
    return pc


def gconv_split_pre_supported(desc):
    return lib().rd_gconv_split_pre_supported(C.byref(desc)) == 1


def gconv_split_pre(desc, x_pieces, w_split, out, bias=None, act=0, act_cols=0, addend=None, ld_add=0, stat=None):
    """rd_gconv_split with the activation already split by its producer (ops.split_pieces)."""
    import torch
    _poison()
    assert x_pieces.dtype == torch.bfloat16 and x_pieces.shape[0] == 3 and w_split.dtype == torch.bfloat16 and w_split.shape[0] == 3
    check(lib().rd_gconv_split_pre(C.byref(desc), ptr(x_pieces), C.c_int64(x_pieces[0].numel()), ptr(w_split), C.c_int64(w_split[0].numel()),
                                   ptr(_f32(out)), ptr(bias), act, act_cols, ptr(addend), ld_add, ptr(stat), current_stream()), "rd_gconv_split_pre")
    return out


def gconv_split_pre_stat_tiles(desc):
    n = lib().rd_gconv_split_pre_stat_tiles(C.byref(desc))
    if n < 0:
        check(n, "rd_gconv_split_pre_stat_tiles")
    return n


def gconv_split_stat_tiles(desc):
    n = lib().rd_gconv_split_stat_tiles(C.byref(desc))
    if n < 0:
        check(n, "rd_gconv_split_stat_tiles")
    return n


def gconv_stat_tiles(desc):
    n = lib().rd_gconv_stat_tiles_ws(C.byref(desc))
    if n < 0:
        check(n, "rd_gconv_stat_tiles_ws")
    return n


_WS = {}


def _poison():
    """RD_POISON_LDS=1: NaN-fill every CU's LDS before the launch (a kernel that consumes unwritten LDS then yields NaN)."""
    if _os.environ.get("RD_POISON_LDS") == "1":
        check(lib().rd_debug_poison_lds(current_stream()), "rd_debug_poison_lds")


def gconv(desc, x, w_packed, out, addend=None, ld_add=0, stat=None):
    """Runs through rd_gconv_ws (split-K allowed); the workspace is cached per size on the tensor's device."""
    _poison()
    n = int(lib().rd_gconv_workspace_floats(C.byref(desc)))
    if n < 0:
        check(n, "rd_gconv_workspace_floats")
    ws = None
    if n > 0:
        ws = _WS.get((n, out.device))
        if ws is None:
            ws = _WS[(n, out.device)] = torch.empty(n, dtype=torch.float32, device=out.device)
    check(lib().rd_gconv_ws(C.byref(desc), ptr(x), ptr(w_packed), ptr(out), ptr(addend), ld_add, ptr(stat), ptr(ws),
                            current_stream()), "rd_gconv_ws")
    return out


def gconv_bnbwd(desc, dout, w_packed, dx, bn_x, ld_x, mean, scale, shift, act, red):
    """Input gradient + the BatchNorm-backward sums of the BatchNorm behind the convolution's forward input (rd_gconv_bnbwd)."""
    _poison()
    check(lib().rd_gconv_bnbwd(C.byref(desc), ptr(dout), ptr(w_packed), ptr(dx), ptr(bn_x), ld_x, ptr(mean), ptr(scale), ptr(shift),
                               act, ptr(red), None, current_stream()), "rd_gconv_bnbwd")
    return dx


def fill(t, v):
    check(lib().rd_fill(ptr(t), C.c_int64(t.numel()), C.c_float(v), current_stream()), "rd_fill")
    return t


def nchw_to_nhwc(x):
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    check(lib().rd_nchw_to_nhwc(ptr(_f32(x)), ptr(out), n, c, h, w, current_stream()), "rd_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(lib().rd_nhwc_to_nchw(ptr(_f32(x)), ptr(out), n, c, h, w, current_stream()), "rd_nhwc_to_nchw")
    return out


def wgrad_workspace_floats(desc):
    n = lib().rd_wgrad_workspace_floats(C.byref(desc))
    if n < 0:
        check(int(n), "rd_wgrad_workspace_floats")
    return int(n)


def wgrad(desc, x, dout, slabs):
    _poison()
    check(lib().rd_wgrad(C.byref(desc), ptr(x), ptr(dout), ptr(slabs), current_stream()), "rd_wgrad")
    return slabs


def wgrad_reduce(desc, slabs, grad_oihw, co_off=0, accumulate=False):
    o, i, kh, kw = grad_oihw.shape
    check(lib().rd_wgrad_reduce(C.byref(desc), ptr(slabs), ptr(_f32(grad_oihw)), o, i, kh, kw, co_off, int(accumulate),
                                current_stream()), "rd_wgrad_reduce")
    return grad_oihw


def wgrad_reduce_batched(jobs):
    """jobs: [(desc, slabs, grad_oihw, co_off)] (fp32 rd_wgrad slabs) reduced by ONE rd_wgrad_reduce_batched call (two launches)."""
    import numpy as np
    from ._lib import RdReduceJob
    arr = (RdReduceJob * len(jobs))()
    bj1, bj2 = [], []
    for q, (desc, slabs, grad, co_off) in enumerate(jobs):
        o, i, kh, kw = grad.shape
        check(lib().rd_wgrad_reduce_job(C.byref(desc), 0, ptr(slabs), ptr(_f32(grad)), o, i, kh, kw, co_off, 0, C.byref(arr[q])), "rd_wgrad_reduce_job")
        arr[q].first_block1, arr[q].first_block2 = len(bj1), len(bj2)
        bj1 += [q] * arr[q].n_blocks1
        bj2 += [q] * arr[q].n_blocks2
    dev = jobs[0][1].device
    table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
    t1 = torch.tensor(bj1 or [0], dtype=torch.int32, device=dev)
    t2 = torch.tensor(bj2, dtype=torch.int32, device=dev)
    check(lib().rd_wgrad_reduce_batched(ptr(table), ptr(t1), len(bj1), ptr(t2), len(bj2), current_stream()), "rd_wgrad_reduce_batched")
    torch.cuda.current_stream().synchronize()          # the tables are temporaries
    return len(bj1), len(bj2)


# ---- BatchNorm / activation helpers (used by the unit tests; the engine calls the C ABI directly)
def bn_stats(x2d, C_):
    """x2d: [M, ld] view (ld >= C_).  Returns (partial [tiles,2,C], tiles)."""
    M = x2d.shape[0]
    tiles = lib().rd_bn_stats_tiles(C.c_int64(M))
    part = torch.zeros(tiles, 2, C_, device=x2d.device)
    nt = C.c_int32(0)
    check(lib().rd_bn_stats(ptr(x2d), C.c_int64(M), C_, x2d.stride(0), ptr(part), C.byref(nt), current_stream()), "rd_bn_stats")
    assert nt.value == tiles
    return part, tiles


def wgrad_bf16(desc, x, dout, grad_oihw):
    """bf16-operand weight gradient of a stride-1 3x3 convolution (rd_wgrad_bf16 + rd_wgrad_bf16_reduce) into grad_oihw."""
    _poison()
    n = int(lib().rd_wgrad_bf16_workspace_floats(C.byref(desc)))
    if n < 0:
        check(n, "rd_wgrad_bf16_workspace_floats")
    slabs = torch.empty(n, dtype=torch.float32, device=x.device)
    check(lib().rd_wgrad_bf16(C.byref(desc), ptr(_f32(x)), ptr(_f32(dout)), ptr(slabs), current_stream()), "rd_wgrad_bf16")
    o, i, kh, kw = grad_oihw.shape
    check(lib().rd_wgrad_bf16_reduce(C.byref(desc), ptr(slabs), ptr(_f32(grad_oihw)), o, i, kh, kw, 0, 0, current_stream()),
          "rd_wgrad_bf16_reduce")
    return grad_oihw


def wgrad_split_supported(desc):
    return lib().rd_wgrad_split_supported(C.byref(desc)) == 1


def wgrad_split_workspace_floats(desc):
    n = int(lib().rd_wgrad_split_workspace_floats(C.byref(desc)))
    if n < 0:
        check(n, "rd_wgrad_split_workspace_floats")
    return n


def wgrad_split(desc, x, dout, slabs):
    """fp32 weight gradient rebuilt from six bf16 MFMAs per product (three-piece operands): rd_wgrad_split."""
    _poison()
    check(lib().rd_wgrad_split(C.byref(desc), ptr(_f32(x)), ptr(_f32(dout)), ptr(slabs), current_stream()), "rd_wgrad_split")


def wgrad_split_pre_supported(desc):
    return lib().rd_wgrad_split_pre_supported(C.byref(desc)) == 1


def wgrad_split_pre(desc, x_pieces, dy_pieces, slabs):
    """rd_wgrad_split with both operands already split by their producers (ops.split_pieces)."""
    _poison()
    check(lib().rd_wgrad_split_pre(C.byref(desc), ptr(x_pieces), C.c_int64(x_pieces[0].numel()), ptr(dy_pieces), C.c_int64(dy_pieces[0].numel()),
                                   ptr(slabs), current_stream()), "rd_wgrad_split_pre")


def wgrad_split_reduce(desc, slabs, grad, co_off=0, accumulate=False):
    o, i, kh, kw = grad.shape
    check(lib().rd_wgrad_split_reduce(C.byref(desc), ptr(slabs), ptr(grad), o, i, kh, kw, co_off, int(accumulate), current_stream()),
          "rd_wgrad_split_reduce")


# ------------------------------------------------------------------ Winograd F(2x2, 3x3) on the split pipeline (csrc/wino_split.hip)
def wino_supported(H, W, cin, cout, ldi=None, ldo=None):
    return lib().rd_wino_supported(H, W, cin, cout, cin if ldi is None else ldi, cout if ldo is None else ldo) == 1


def wino_pack(w_oihw, flip=False):
    """G g G^T of an [O,I,3,3] weight tensor as the three-piece bf16 operand of rd_wino_conv3x3 (flip: the input-gradient operand --
    channels transposed, taps rotated by 180 degrees)."""
    o, i, kh, kw = w_oihw.shape
    assert kh == 3 and kw == 3
    lib().rd_wino_packed_bytes.restype = C.c_int64
    nbytes = int(lib().rd_wino_packed_bytes(o, i, int(flip)))
    u = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_oihw.device)
    check(lib().rd_wino_pack(ptr(_f32(w_oihw.contiguous())), o, i, int(flip), ptr(u), current_stream()), "rd_wino_pack")
    return u


def wino_conv3x3(x_nhwc, u_packed, out_nhwc, addend=None, stat=None):
    """3x3 / stride 1 / pad 1 convolution of an NHWC fp32 tensor (channel slices allowed: strides are taken from the tensors)."""
    n, h, w, cin = x_nhwc.shape
    cout = out_nhwc.shape[3]
    for t in (x_nhwc, out_nhwc) + ((addend,) if addend is not None else ()):
        assert t.dtype == torch.float32 and t.is_cuda and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2) and t.stride(0) == t.shape[1] * t.stride(1), \
            "expected an NHWC CUDA float32 tensor (a channel slice of a wider buffer is fine)"
    check(lib().rd_wino_conv3x3(ptr(x_nhwc), n, h, w, cin, x_nhwc.stride(2), ptr(u_packed), ptr(out_nhwc), cout, out_nhwc.stride(2),
                                ptr(addend), 0 if addend is None else addend.stride(2), ptr(stat), current_stream()), "rd_wino_conv3x3")
    return out_nhwc


def wino_stat_tiles(n, h, w):
    return int(lib().rd_wino_stat_tiles(n, h, w))
