"""GPU parity of rd_conv16_split (csrc/conv16_split.hip: the 16 -> 16 channel 3x3 layers with three-piece bf16 operands, six
v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation) against an fp64 convolution and against rd_gconv's 16x16x4 fp32-MFMA kernel for the
same descriptors: forward and input gradient, residual addend, BatchNorm partial sums, ragged tiles, the bench geometries, a 2^-20 .. 2^20
input range.  Bar: 2e-6 of the output's max magnitude and no worse than twice the fp32-MFMA kernel's own error."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(2, 33, 47), (1, 16, 16), (3, 15, 63), (16, 113, 200), (16, 240, 400)])
@pytest.mark.parametrize("direction", ["fwd", "dgrad"])
def test_conv16_split(cfg, direction):
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    b, h, w = cfg
    dev = "cuda"
    g = torch.Generator().manual_seed(21)
    x = torch.randn(b, h, w, 16, generator=g).to(dev)
    x[:, : h // 3] *= 2.0 ** 20
    x[:, 2 * h // 3:] *= 2.0 ** -20
    wt = (torch.randn(16, 16, 3, 3, generator=g) * 0.2).to(dev)
    add = torch.randn(b, h, w, 16, generator=g).to(dev)
    if direction == "fwd":
        d = cd.conv_fwd(b, h, w, 16, 16, 3, 1, 1)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1).contiguous()
        tr = False
    else:
        d, zf = cd.conv_dgrad(b, h, w, 16, 16, 3, 1, 1)
        assert not zf
        ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1).contiguous()
        tr = True
    assert L.rd_conv16_split_supported(C.byref(d)) == 1
    wp = ops.pack_weights(wt, transpose=tr)
    tiles = L.rd_gconv_stat_tiles_ws(C.byref(d))
    out32, st32 = torch.empty(b, h, w, 16, device=dev), torch.zeros(tiles, 2, 16, device=dev)
    ops.gconv(d, x, wp, out32, stat=st32)
    out, st = torch.full((b, h, w, 16), float("nan"), device=dev), torch.zeros(tiles, 2, 16, device=dev)
    check(L.rd_conv16_split(C.byref(d), ptr(x), ptr(wp), ptr(out), None, 0, ptr(st), current_stream()), "rd_conv16_split")
    outa = torch.full((b, h, w, 16), float("nan"), device=dev)
    check(L.rd_conv16_split(C.byref(d), ptr(x), ptr(wp), ptr(outa), ptr(add), 16, None, current_stream()), "rd_conv16_split(addend)")
    torch.cuda.synchronize()
    assert not torch.isnan(out).any() and not torch.isnan(outa).any()

    def err(o, r):          # per third of the image (each third has its own magnitude)
        e = []
        for lo, hi in ((0, h // 3 - 1), (h // 3 + 1, 2 * h // 3 - 1), (2 * h // 3 + 1, h)):
            if hi > lo:
                e.append(((o[:, lo:hi].double() - r[:, lo:hi]).abs().max() / r[:, lo:hi].abs().max()).item())
        return max(e)
    e_sp, e_32 = err(out, ref), err(out32, ref)
    assert e_sp < 2e-6 and e_sp < 2.0 * e_32 + 1e-7, (cfg, direction, e_sp, e_32)
    assert err(outa, ref + add.double()) < 2e-6
    s_, f_ = st.sum(0).double(), st32.sum(0).double()
    assert ((s_ - f_).abs().max() / f_.abs().max()).item() < 1e-5
    # per-tile rows: same tiling, so the rows agree row by row as well
    assert ((st.double() - st32.double()).abs().max() / st32.double().abs().max()).item() < 1e-5
