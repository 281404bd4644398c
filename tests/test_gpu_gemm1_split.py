"""GPU parity of the one-tap kernel behind rd_gconv_split (csrc/gemm1_split.hip: 1x1 convolutions and their input gradients as a GEMM over
three-piece bf16 operands) against torch CPU fp32 convolutions, through the C ABI, at the tolerance of the other fp32 convolution tests
(2e-5 of the output's max magnitude).  The layers: conv_fusion 640 -> 512, conv2 512 -> 256, the ResNet downsample convolutions
(/root/reference/model/models.py:559-569,600-625,652-657)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _served_by_gemm1(d):
    from radar_depth_amd._lib import lib
    info = (C.c_int32 * 8)()
    return lib().rd_gconv_split_plan_info(C.byref(d), info) == 0 and info[7] == 1000


FWD = [
    # n, cin, cout, stride, h, w
    (2, 640, 512, 1, 15, 25),     # conv_fusion
    (16, 512, 256, 1, 15, 25),    # decoder conv2 at the bench batch
    (2, 64, 128, 2, 113, 200),    # layer2 downsample
    (2, 128, 256, 2, 57, 100),
    (2, 256, 512, 2, 29, 50),
    (3, 32, 32, 1, 9, 7),         # one ragged tile, one column tile of 32
    (1, 96, 64, 1, 1, 1),         # a single pixel
    (2, 64, 192, 2, 7, 5),        # odd sizes under stride 2
    (5, 160, 64, 1, 31, 17),
]


@pytest.mark.parametrize("cfg", FWD)
def test_gemm1_split_forward(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, s, h, w = cfg
    g = torch.Generator().manual_seed(10)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5
    y = F.conv2d(x, wt, stride=s)
    d = cd.conv_fwd(n, h, w, ci, co, 1, s, 0)
    assert ops.gconv_split_supported(d) and _served_by_gemm1(d)
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
    stat = torch.zeros(ops.gconv_split_stat_tiles(d), 2, co, device="cuda")
    ops.gconv_split(d, ops.nchw_to_nhwc(x.cuda()), ops.pack_weights_split(wt.cuda()), out, stat=stat)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, y) < 2e-5, (_rel(got, y), cfg)
    s_ = stat.sum(0).cpu().double()
    ref_s = y.double().sum((0, 2, 3))
    ref_q = (y.double() ** 2).sum((0, 2, 3))
    assert ((s_[0] - ref_s).abs().max() / ref_q.sqrt().max()).item() < 1e-4
    assert _rel(s_[1], ref_q) < 1e-4


@pytest.mark.parametrize("cfg", [
    (2, 64, 128, 2, 113, 200),
    (2, 256, 512, 2, 29, 50),
    (16, 640, 512, 1, 15, 25),
    (2, 512, 256, 1, 15, 25),
    (3, 32, 32, 1, 9, 7),
    (2, 64, 192, 2, 7, 5),
])
def test_gemm1_split_dgrad(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, s, h, w = cfg
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
    wt = torch.randn(co, ci, 1, 1, generator=g) * (2.0 / co) ** 0.5
    y = F.conv2d(x, wt, stride=s)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d, zero_fill = cd.conv_dgrad(n, h, w, ci, co, 1, s, 0)
    assert ops.gconv_split_supported(d) and _served_by_gemm1(d)
    assert zero_fill == (s == 2)
    wp = ops.pack_weights_split(wt.cuda(), transpose=True)
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    if zero_fill:
        ops.fill(dx, 0.0)
    add = torch.randn(n, h, w, ci, generator=g)
    ops.gconv_split(d, ops.nchw_to_nhwc(gy.cuda()), wp, dx, addend=add.cuda() if not zero_fill else None, ld_add=ci)
    torch.cuda.synchronize()
    want = x.grad + (add.permute(0, 3, 1, 2) if not zero_fill else 0)
    got = dx.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, want) < 2e-5


@pytest.mark.parametrize("act_cols", [64, 30])
def test_gemm1_split_fused_epilogue_and_strided_tensors(act_cols):
    """bias + addend + activation on the first act_cols channels (act_cols = 30: the scalar epilogue); the input a channel slice of a
    wider tensor (ldi > Cin: the late-fusion concatenation buffer), the output a slice of a wider tensor (ldo > Cout)."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 2, 96, 128, 21, 30
    g = torch.Generator().manual_seed(12)
    xw = torch.randn(n, h, w, ci + 32, generator=g)
    x = xw[..., 32:].permute(0, 3, 1, 2).contiguous()
    wt = torch.randn(co, ci, 1, 1, generator=g) * 0.1
    bias = torch.randn(co, generator=g)
    add = torch.randn(n, co, h, w, generator=g)
    y = F.conv2d(x, wt) + bias[None, :, None, None] + add
    y[:, :act_cols] = F.relu(y[:, :act_cols])
    d = cd.conv_fwd(n, h, w, ci, co, 1, 1, 0, ldi=ci + 32)
    d.ldo = co + 64
    assert _served_by_gemm1(d)
    xg = xw.cuda()
    outw = torch.full((n, h, w, co + 64), float("nan"), device="cuda")
    # (ops.gconv_split takes whole tensors; the channel slices go through the C ABI directly, the way the engine passes them)
    from radar_depth_amd._lib import current_stream, lib
    wp, bg, ag = ops.pack_weights_split(wt.cuda()), bias.cuda(), ops.nchw_to_nhwc(add.cuda())
    rc = lib().rd_gconv_split(C.byref(d), C.c_void_p(xg[..., 32:].data_ptr()), C.c_void_p(wp.data_ptr()), C.c_int64(wp[0].numel()),
                              C.c_void_p(outw[..., 64:].data_ptr()), C.c_void_p(bg.data_ptr()), 1, act_cols, C.c_void_p(ag.data_ptr()), co, None,
                              current_stream())
    assert rc == 0, lib().rd_last_error().decode()
    torch.cuda.synchronize()
    assert torch.isnan(outw[..., :64]).all()          # nothing outside the slice is written
    assert _rel(outw[..., 64:].permute(0, 3, 1, 2).cpu(), y) < 2e-5


def test_gemm1_split_is_bitwise_reproducible_and_as_close_to_fp64_as_fp32_mfma():
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 16, 640, 512, 15, 25
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5
    y64 = F.conv2d(x.double(), wt.double())
    d = cd.conv_fwd(n, h, w, ci, co, 1, 1, 0)
    xg = ops.nchw_to_nhwc(x.cuda())
    wp = ops.pack_weights_split(wt.cuda())
    outs = []
    for _ in range(3):
        out = torch.empty(n, h, w, co, device="cuda")
        ops.gconv_split(d, xg, wp, out)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.empty(n, h, w, co, device="cuda")
    ops.gconv(d, xg, ops.pack_weights(wt.cuda()), ref)
    torch.cuda.synchronize()
    e_split = _rel(outs[0].permute(0, 3, 1, 2).cpu().double(), y64)
    e_fp32 = _rel(ref.permute(0, 3, 1, 2).cpu().double(), y64)
    assert e_split < 2.0 * e_fp32 + 1e-7, (e_split, e_fp32)


def test_gemm1_split_dynamic_range_and_non_finite():
    """Operands from 2^-60 to 2^60 keep fp32 accuracy; an Inf / NaN input reaches exactly the output pixels that read it."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 1, 64, 64, 8, 8
    g = torch.Generator().manual_seed(14)
    for ex in (-60, 0, 60):
        x = torch.randn(n, ci, h, w, generator=g) * 2.0 ** ex
        wt = torch.randn(co, ci, 1, 1, generator=g) * 2.0 ** (-ex / 2)
        y = F.conv2d(x.double(), wt.double()).float()
        d = cd.conv_fwd(n, h, w, ci, co, 1, 1, 0)
        out = torch.empty(n, h, w, co, device="cuda")
        ops.gconv_split(d, ops.nchw_to_nhwc(x.cuda()), ops.pack_weights_split(wt.cuda()), out)
        assert _rel(out.permute(0, 3, 1, 2).cpu(), y) < 2e-5, ex
    x = torch.randn(n, ci, h, w, generator=g)
    x[0, 3, 2, 5] = float("inf")
    x[0, 7, 6, 1] = float("nan")
    out = torch.empty(n, h, w, co, device="cuda")
    ops.gconv_split(d, ops.nchw_to_nhwc(x.cuda()), ops.pack_weights_split(wt.cuda()), out)
    bad = ~torch.isfinite(out.cpu())
    want = torch.zeros(n, h, w, co, dtype=torch.bool)
    want[0, 2, 5] = True
    want[0, 6, 1] = True
    assert torch.equal(bad, want)
