"""Randomised parity sweep of rd_gconv (forward, dgrad) and rd_wgrad against torch CPU fp32 over many small random geometries:
   python tools/fuzz_conv.py [n_cases] [seed]"""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops

POISON = "--poison" in sys.argv          # NaN-fill every CU's LDS before each launch: catches reads of unwritten LDS
sys.argv = [a for a in sys.argv if a != "--poison"]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
from radar_depth_amd._lib import current_stream, lib


def poison():
    if POISON:
        assert lib().rd_debug_poison_lds(current_stream()) == 0


rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


bad = 0
for case in range(n_cases):
    k, s = [(3, 1), (3, 2), (1, 1), (1, 2)][rng.randint(4)]
    p = k // 2
    ci = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 160, 256]))
    co = int(rng.choice([4, 16, 24, 32, 48, 64, 96, 128, 192, 256]))
    n = int(rng.randint(1, 5))
    big = rng.rand() < 0.25          # a quarter of the cases are tall/wide enough for the column-strip kernel and row segments
    h, w = (int(rng.randint(24, 130)), int(rng.randint(20, 210))) if big else (int(rng.randint(1, 70)), int(rng.randint(1, 90)))
    if big:
        ci, co = min(ci, 128), min(co, 128)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    tag = "n%d ci%d co%d k%d s%d %dx%d" % (n, ci, co, k, s, h, w)
    try:
        d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
        xs, gys = ops.nchw_to_nhwc(x.detach().cuda()), ops.nchw_to_nhwc(gy.cuda())
        out = torch.empty(n, d.Ho, d.Wo, co, device="cuda")
        wp_ = ops.pack_weights(wt.detach().cuda())
        poison()
        ops.gconv(d, xs, wp_, out)
        e_f = rel(ops.nhwc_to_nchw(out).cpu(), y.detach())
        # the same launch with the fused BatchNorm partial sums (what every training-plan forward conv requests)
        stat = torch.full((ops.gconv_stat_tiles(d), 2, co), float("nan"), device="cuda")
        out2 = torch.full_like(out, float("nan"))
        poison()
        ops.gconv(d, xs, wp_, out2, stat=stat)
        e_f = max(e_f, rel(ops.nhwc_to_nchw(out2).cpu(), y.detach()))
        yd = y.detach().double()
        s_ = stat.double().sum(0).cpu()
        e_s = max(((s_[0] - yd.sum((0, 2, 3))).abs().max() / (yd ** 2).sum((0, 2, 3)).sqrt().max().clamp_min(1e-30)).item(),
                  rel(s_[1], (yd ** 2).sum((0, 2, 3))))
        if not (e_s < 1e-4):
            e_f = max(e_f, 1.0)
            print("  stat mismatch %.2e" % e_s, tag)
        slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
        poison()
        ops.wgrad(d, xs, gys, slabs)
        gw = torch.full((co, ci, k, k), float("nan"), device="cuda")
        ops.wgrad_reduce(d, slabs, gw)
        e_w = rel(gw.cpu(), wt.grad)
        e_d = 0.0
        if co % 16 == 0:        # dgrad: the reduction dimension (forward Cout) must be a multiple of 16
            dd, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
            dx = torch.zeros(n, h, w, ci, device="cuda") if zero_fill else torch.empty(n, h, w, ci, device="cuda")
            wd_ = ops.pack_weights(wt.detach().cuda(), transpose=True)
            poison()
            ops.gconv(dd, gys, wd_, dx)
            e_d = rel(ops.nhwc_to_nchw(dx).cpu(), x.grad)
        torch.cuda.synchronize()
        ok = e_f < 5e-5 and e_w < 1e-4 and e_d < 5e-5
    except Exception as ex:     # noqa: BLE001
        ok, e_f, e_w, e_d = False, -1, -1, -1
        print("EXC", tag, repr(ex)[:200])
    if not ok:
        bad += 1
        print("FAIL", tag, "fwd %.2e wgrad %.2e dgrad %.2e" % (e_f, e_w, e_d))
print("%d cases, %d failures" % (n_cases, bad))
