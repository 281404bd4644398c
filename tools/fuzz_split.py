"""Randomised parity sweep of the split kernels (rd_gconv_split forward with statistics and input gradient, rd_wgrad_split + reduction,
UpProj forms) against torch CPU fp32 over many random geometries, every shape the kernels accept (RD_GCONV_SPLIT_ALL=1):
   python tools/fuzz_split.py [n_cases] [seed] [--poison]"""
import os
import sys

os.environ["RD_GCONV_SPLIT_ALL"] = "1"
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import current_stream, lib

POISON = "--poison" in sys.argv
sys.argv = [a for a in sys.argv if a != "--poison"]
if POISON:
    os.environ["RD_POISON_LDS"] = "1"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


bad = 0
ran = {"fwd": 0, "dgrad": 0, "wgrad": 0, "up_fwd": 0, "up_dgrad": 0, "up_wgrad": 0}
for case in range(n_cases):
    upproj = rng.rand() < 0.25
    g = torch.Generator().manual_seed(case)
    n = int(rng.randint(1, 5))
    try:
        if upproj:
            c = int(rng.choice([32, 64, 96, 128]))
            h, w = int(rng.randint(1, 40)), int(rng.randint(1, 60))
            tag = "upproj n%d c%d %dx%d" % (n, c, h, w)
            x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
            wcat = (torch.randn(c, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5).requires_grad_(True)
            u = torch.zeros(n, c, 2 * h, 2 * w)
            u[:, :, ::2, ::2] = x.detach()
            u.requires_grad_(True)
            y = F.conv2d(u, wcat, padding=2)
            gy = torch.randn(y.shape, generator=g)
            y.backward(gy)
            d, dd = cd.upproj_fwd(n, h, w, c, c), cd.upproj_dgrad(n, h, w, c, c)
            xs, gys = ops.nchw_to_nhwc(x.detach().cuda()), ops.nchw_to_nhwc(gy.cuda())
            errs = {}
            if ops.gconv_split_supported(d):
                out = torch.full((n, 2 * h, 2 * w, c), float("nan"), device="cuda")
                ops.gconv_split(d, xs, ops.pack_weights_split(wcat.detach().cuda()), out)
                errs["up_fwd"] = rel(out.permute(0, 3, 1, 2).cpu(), y.detach())
            if ops.gconv_split_supported(dd):
                dx = torch.full((n, h, w, c), float("nan"), device="cuda")
                ops.gconv_split(dd, gys, ops.pack_weights_split(wcat.detach().cuda(), transpose=True), dx)
                errs["up_dgrad"] = rel(dx.permute(0, 3, 1, 2).cpu(), u.grad[:, :, ::2, ::2])
            if ops.wgrad_split_supported(d):
                sl = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
                ops.wgrad_split(d, xs, gys, sl)
                g0 = torch.full((c // 2, c, 5, 5), float("nan"), device="cuda")
                g1 = torch.full((c // 2, c, 5, 5), float("nan"), device="cuda")
                ops.wgrad_split_reduce(d, sl, g0, co_off=0)
                ops.wgrad_split_reduce(d, sl, g1, co_off=c // 2)
                errs["up_wgrad"] = max(rel(g0.cpu(), wcat.grad[:c // 2]), rel(g1.cpu(), wcat.grad[c // 2:]))
        else:
            k, s = [(3, 1), (3, 1), (3, 2), (1, 1)][rng.randint(4)]
            p = k // 2
            ci = int(rng.choice([32, 48, 64, 80, 96, 128, 160, 256]))
            co = int(rng.choice([32, 40, 64, 72, 96, 128, 192, 256]))
            big = rng.rand() < 0.25
            h, w = (int(rng.randint(24, 130)), int(rng.randint(20, 210))) if big else (int(rng.randint(1, 70)), int(rng.randint(1, 90)))
            if big:
                ci, co = min(ci, 128), min(co, 128)
            tag = "n%d ci%d co%d k%d s%d %dx%d" % (n, ci, co, k, s, h, w)
            x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
            wt = (torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5).requires_grad_(True)
            y = F.conv2d(x, wt, stride=s, padding=p)
            gy = torch.randn(y.shape, generator=g)
            y.backward(gy)
            d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
            dd, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
            xs, gys = ops.nchw_to_nhwc(x.detach().cuda()), ops.nchw_to_nhwc(gy.cuda())
            errs = {}
            if ops.gconv_split_supported(d):
                out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
                st = torch.zeros(ops.gconv_split_stat_tiles(d), 2, co, device="cuda")
                ops.gconv_split(d, xs, ops.pack_weights_split(wt.detach().cuda()), out, stat=st)
                errs["fwd"] = rel(out.permute(0, 3, 1, 2).cpu(), y.detach())
                ssum = st.sum(0).cpu().double()
                q = (y.detach().double() ** 2).sum((0, 2, 3))
                errs["fwd"] = max(errs["fwd"], ((ssum[1] - q).abs().max() / q.max()).item() * 0.2)
            if ops.gconv_split_supported(dd):
                dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
                if zero_fill:
                    ops.fill(dx, 0.0)
                ops.gconv_split(dd, gys, ops.pack_weights_split(wt.detach().cuda(), transpose=True), dx)
                errs["dgrad"] = rel(dx.permute(0, 3, 1, 2).cpu(), x.grad)
            if ops.wgrad_split_supported(d):
                sl = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
                ops.wgrad_split(d, xs, gys, sl)
                gw = torch.full((co, ci, k, k), float("nan"), device="cuda")
                ops.wgrad_split_reduce(d, sl, gw)
                errs["wgrad"] = rel(gw.cpu(), wt.grad)
        torch.cuda.synchronize()
        for key, e in errs.items():
            ran[key] += 1
            tol = 5e-5 if "wgrad" in key else 2e-5
            if not (e < tol):
                bad += 1
                print("FAIL %s %s rel %.3e" % (tag, key, e))
    except Exception as ex:      # a launch failure is a finding too
        bad += 1
        print("ERROR %s: %s" % (tag, ex))
print("cases", n_cases, "kernel runs", ran, "failures", bad, "(LDS poisoned)" if POISON else "")
