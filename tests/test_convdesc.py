"""Host logic (no GPU): every descriptor builder reproduces the torch convolution it stands for,
including odd sizes (113, 57, 29, 15), stride-2 parity phases and the UpProj zero-skipping identity."""
import pytest
import torch
import torch.nn.functional as F

from radar_depth_amd import convdesc as cd
from tests.desc_emulator import pack_dgrad, pack_fwd, run_desc, run_wgrad

torch.manual_seed(0)
DT = torch.float64


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("k,s,p,h,w", [(3, 1, 1, 7, 9), (3, 2, 1, 7, 9), (3, 2, 1, 8, 10), (1, 2, 0, 7, 9),
                                        (1, 1, 0, 5, 6), (3, 2, 1, 113, 20), (1, 2, 0, 29, 5)])
def test_conv_fwd_dgrad_wgrad(k, s, p, h, w):
    n, ci, co = 2, 16, 8
    x = torch.randn(n, ci, h, w, dtype=DT, requires_grad=True)
    wt = torch.randn(co, ci, k, k, dtype=DT, requires_grad=True)
    y = F.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn_like(y)
    y.backward(gy)
    d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
    assert (d.Ho, d.Wo) == tuple(y.shape[2:])
    got = run_desc(d, nhwc(x.detach()), pack_fwd(wt.detach()))
    assert torch.allclose(got, nhwc(y.detach()), atol=1e-10)
    dd, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
    gx = run_desc(dd, nhwc(gy), pack_dgrad(wt.detach()))
    if zero_fill:
        gx = torch.nan_to_num(gx, nan=0.0)
    assert not torch.isnan(gx).any()
    assert torch.allclose(gx, nhwc(x.grad), atol=1e-10)
    dw = run_wgrad(d, nhwc(x.detach()), nhwc(gy), k * k)
    assert torch.allclose(dw, pack_fwd(wt.grad), atol=1e-9)


@pytest.mark.parametrize("h,w", [(7, 9), (15, 25), (4, 3)])
def test_upproj_identity(h, w):
    n, ci, co = 2, 16, 12
    x = torch.randn(n, ci, h, w, dtype=DT, requires_grad=True)
    wt = torch.randn(co, ci, 5, 5, dtype=DT, requires_grad=True)
    u = torch.zeros(n, ci, 2 * h, 2 * w, dtype=DT)
    u = u.clone()
    up = F.conv_transpose2d(x, torch.ones(ci, 1, 1, 1, dtype=DT), stride=2, groups=ci, output_padding=1)
    y = F.conv2d(up, wt, padding=2)
    gy = torch.randn_like(y)
    y.backward(gy)
    d = cd.upproj_fwd(n, h, w, ci, co)
    taps = [d.phase[i].n_taps for i in range(4)]
    assert taps == [9, 6, 6, 4]
    got = run_desc(d, nhwc(x.detach()), pack_fwd(wt.detach()))
    assert torch.allclose(got, nhwc(y.detach()), atol=1e-10)
    dd = cd.upproj_dgrad(n, h, w, ci, co)
    gx = run_desc(dd, nhwc(gy), pack_dgrad(wt.detach()))
    assert torch.allclose(gx, nhwc(x.grad), atol=1e-10)
    dw = run_wgrad(d, nhwc(x.detach()), nhwc(gy), 25)
    assert torch.allclose(dw, pack_fwd(wt.grad), atol=1e-9)
