"""GPU parity of the 7x7 / stride-2 stem forward kernels (rd_stem_fwd: csrc/stem.hip for the RGB stem, csrc/stem16.hip for the 16-channel
depth stem with one or two input planes) against torch CPU fp32 conv2d, through the C ABI, incl. the BatchNorm partial sums.
Tolerance 2e-5 of the output's max magnitude (exact-fp32 MFMA chain; only the summation order differs)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    (2, 3, 64, 97, 161),      # RGB stem, ragged tiles
    (2, 1, 16, 97, 161),      # depth stem
    (2, 2, 16, 97, 161),      # stage 2 of the multistage net: two depth planes
    (3, 1, 16, 450, 800),     # bench geometry
    (1, 1, 16, 15, 63),       # a single ragged tile row
    (2, 2, 16, 64, 64),
])
def test_stem_forward(cfg):
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    n, cin, cout, h, w = cfg
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin + 1, h, w, generator=g)          # the stem reads planes 1.. of a wider NCHW tensor (strided images)
    wt = torch.randn(cout, cin, 7, 7, generator=g) * 0.1
    y = F.conv2d(x[:, 1:], wt, stride=2, padding=3)
    xg = x.cuda()
    hw = h * w
    planes = (C.c_void_p * 3)(*[xg.data_ptr() + 4 * hw * (1 + c) if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[(cin + 1) * hw if c < cin else 0 for c in range(3)])
    wp = wt.permute(2, 3, 1, 0).reshape(49, cin, cout).contiguous().cuda()
    ho, wo = y.shape[2], y.shape[3]
    out = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
    tiles = L.rd_stem_stat_tiles(n, h, w)
    stat = torch.zeros(tiles, 2, cout, device="cuda")
    check(L.rd_stem_fwd(planes, strides, cin, n, h, w, ptr(wp), cout, ptr(out), ptr(stat), current_stream()), "rd_stem_fwd")
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert ((got - y).abs().max() / y.abs().max()).item() < 2e-5, cfg
    s_ = stat.sum(0).cpu().double()
    ref_s, ref_q = y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))
    assert ((s_[0] - ref_s).abs().max() / ref_q.sqrt().max()).item() < 1e-4
    assert ((s_[1] - ref_q).abs().max() / ref_q.max()).item() < 1e-4
