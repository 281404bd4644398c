"""GPU unit parity of the BatchNorm / activation / pooling / head / loss / SGD kernels against torch CPU autograd."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _act(z, act):
    return {0: z, 1: torch.relu(z), 2: F.leaky_relu(z, 0.2)}[act]


@pytest.mark.parametrize("M,Cc,act,res", [(192, 128, 1, "bn"), (48, 512, 1, "id"), (768, 64, 1, "bn"), (5000, 16, 1, "id"),
                                          (100, 4, 0, None), (77, 640, 0, None), (22600, 64, 2, None), (3, 32, 1, "id")])
def test_bn_forward_backward(M, Cc, act, res):
    from radar_depth_amd import ops
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    g = torch.Generator().manual_seed(5)
    x1 = (torch.randn(M, Cc, generator=g) * 2 + 0.5).requires_grad_(True)
    x2 = torch.randn(M, Cc, generator=g).requires_grad_(True) if res else None
    gam1, bet1 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    gam2, bet2 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    gam1.requires_grad_(True); bet1.requires_grad_(True); gam2.requires_grad_(True); bet2.requires_grad_(True)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)

    def bn(x, gm, bt, rmean=None, rvar=None):
        return F.batch_norm(x.t().reshape(1, Cc, M).permute(1, 0, 2).reshape(1, Cc, M) if False else x.t().unsqueeze(0), rmean, rvar,
                            gm, bt, True, 0.1, 1e-5).squeeze(0).t()

    z = bn(x1, gam1, bet1, rm, rv)
    if res == "bn":
        z = z + bn(x2, gam2, bet2)
    elif res == "id":
        z = z + x2
    y = _act(z, act)
    dy = torch.randn(M, Cc, generator=g)
    y.backward(dy)

    dev = "cuda"
    X1, X2 = x1.detach().to(dev), (x2.detach().to(dev) if res else None)

    def coeffs(X, gm, bt, rmean, rvar):
        part, tiles = ops.bn_stats(X, Cc)
        out = [torch.empty(Cc, device=dev) for _ in range(4)]
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        check(L.rd_bn_finalize(ptr(part), tiles, Cc, 0, Cc, C.c_int64(M), ptr(gm), ptr(bt), C.c_float(1e-5), C.c_float(0.1), ptr(rmean),
                               ptr(rvar), ptr(nbt), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), current_stream()), "finalize")
        return out, nbt

    G1, B1 = gam1.detach().to(dev), bet1.detach().to(dev)
    G2, B2 = gam2.detach().to(dev), bet2.detach().to(dev)
    RM, RV = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    (mean1, inv1, sc1, sh1), nbt = coeffs(X1, G1, B1, RM, RV)
    if res == "bn":
        (mean2, inv2, sc2, sh2), _ = coeffs(X2, G2, B2, None, None)
    Y = torch.empty(M, Cc, device=dev)
    check(L.rd_bn_act(ptr(X1), Cc, ptr(sc1), ptr(sh1), ptr(X2), Cc if res else 0, ptr(sc2) if res == "bn" else None,
                      ptr(sh2) if res == "bn" else None, ptr(Y), Cc, C.c_int64(M), Cc, act, current_stream()), "bn_act")
    torch.cuda.synchronize()
    assert _rel(Y.cpu(), y.detach()) < 2e-5
    assert _rel(RM.cpu(), rm) < 1e-5 and _rel(RV.cpu(), rv) < 1e-5 and int(nbt) == 1
    # backward
    DY = dy.to(dev)
    tiles = L.rd_bn_bwd_tiles(C.c_int64(M), Cc)
    red = torch.zeros(tiles, 3, Cc, device=dev)
    Gt = torch.empty(M, Cc, device=dev)
    check(L.rd_bn_bwd_reduce(ptr(DY), Cc, ptr(Y), Cc, ptr(X1), Cc, ptr(mean1), ptr(X2) if res == "bn" else None, Cc if res == "bn" else 0,
                             ptr(mean2) if res == "bn" else None, ptr(Gt), Cc, C.c_int64(M), Cc, act, ptr(red), current_stream()), "reduce")
    if not res and act != 0:
        # lone act(bn(x1)): the variant that recomputes the activation's sign from x1 must agree bit for bit
        red_x, Gx = torch.zeros(tiles, 3, Cc, device=dev), torch.empty(M, Cc, device=dev)
        check(L.rd_bn_bwd_reduce_x(ptr(DY), Cc, ptr(X1), Cc, ptr(mean1), ptr(sc1), ptr(sh1), ptr(Gx), Cc, C.c_int64(M), Cc, act,
                                   ptr(red_x), current_stream()), "reduce_x")
        torch.cuda.synchronize()
        assert torch.equal(Gx, Gt) and torch.equal(red_x[:, :2], red[:, :2])
        # ... and the pair that never materialises the masked gradient (g = NULL, apply_x) must give the same dx, dgamma, dbeta
        red_n = torch.zeros(tiles, 3, Cc, device=dev)
        check(L.rd_bn_bwd_reduce_x(ptr(DY), Cc, ptr(X1), Cc, ptr(mean1), ptr(sc1), ptr(sh1), None, 0, C.c_int64(M), Cc, act,
                                   ptr(red_n), current_stream()), "reduce_x(g=NULL)")
        dGx, dBx, coefx, DXx = (torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(3 * Cc, device=dev),
                                torch.empty(M, Cc, device=dev))
        check(L.rd_bn_bwd_apply_x(ptr(DY), Cc, ptr(X1), Cc, ptr(red_n), tiles, ptr(G1), ptr(mean1), ptr(inv1), ptr(sc1), ptr(sh1), act,
                                  ptr(dGx), ptr(dBx), ptr(coefx), ptr(DXx), Cc, C.c_int64(M), Cc, current_stream()), "apply_x")
        torch.cuda.synchronize()
        assert _rel(dBx.cpu(), bet1.grad) < 2e-5 and _rel(dGx.cpu(), gam1.grad) < 5e-5 and _rel(DXx.cpu(), x1.grad) < 5e-5
    dG, dB, coef, DX = (torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(3 * Cc, device=dev), torch.empty(M, Cc, device=dev))
    check(L.rd_bn_bwd_apply(ptr(Gt), Cc, ptr(X1), Cc, ptr(red), tiles, 1, ptr(G1), ptr(mean1), ptr(inv1), ptr(dG), ptr(dB), ptr(coef), ptr(DX),
                            Cc, C.c_int64(M), Cc, current_stream()), "apply1")
    torch.cuda.synchronize()
    assert _rel(dB.cpu(), bet1.grad) < 2e-5, "dbeta"
    assert _rel(dG.cpu(), gam1.grad) < 5e-5, "dgamma"
    assert _rel(DX.cpu(), x1.grad) < 5e-5, "dx1"
    if res == "bn":
        check(L.rd_bn_bwd_apply(ptr(Gt), Cc, ptr(X2), Cc, ptr(red), tiles, 2, ptr(G2), ptr(mean2), ptr(inv2), ptr(dG), ptr(dB), ptr(coef),
                                ptr(DX), Cc, C.c_int64(M), Cc, current_stream()), "apply2")
        torch.cuda.synchronize()
        assert _rel(dG.cpu(), gam2.grad) < 5e-5 and _rel(DX.cpu(), x2.grad) < 5e-5
        # the two-operand pair that reads neither y nor a materialised g: same sums bit for bit, same gradients
        red2 = torch.zeros(tiles, 3, Cc, device=dev)
        check(L.rd_bn_bwd_reduce_x2(ptr(DY), Cc, ptr(X1), Cc, ptr(mean1), ptr(sc1), ptr(sh1), ptr(X2), Cc, ptr(mean2), ptr(sc2), ptr(sh2),
                                    C.c_int64(M), Cc, act, ptr(red2), current_stream()), "reduce_x2")
        dg1, db1, dg2, db2 = (torch.empty(Cc, device=dev) for _ in range(4))
        coef6, DX1, DX2 = torch.empty(6 * Cc, device=dev), torch.empty(M, Cc, device=dev), torch.empty(M, Cc, device=dev)
        check(L.rd_bn_bwd_apply_x2(ptr(DY), Cc, ptr(X1), Cc, ptr(X2), Cc, ptr(red2), tiles, ptr(G1), ptr(mean1), ptr(inv1), ptr(sc1), ptr(sh1),
                                   ptr(G2), ptr(mean2), ptr(inv2), ptr(sc2), ptr(sh2), act, ptr(dg1), ptr(db1), ptr(dg2), ptr(db2), ptr(coef6),
                                   ptr(DX1), Cc, ptr(DX2), Cc, C.c_int64(M), Cc, current_stream()), "apply_x2")
        torch.cuda.synchronize()
        assert torch.equal(red2, red)
        assert _rel(DX1.cpu(), x1.grad) < 5e-5 and _rel(DX2.cpu(), x2.grad) < 5e-5
        assert _rel(dg1.cpu(), gam1.grad) < 5e-5 and _rel(dg2.cpu(), gam2.grad) < 5e-5 and _rel(db1.cpu(), bet1.grad) < 2e-5
    elif res == "id":
        assert _rel(Gt.cpu(), x2.grad) < 1e-6


@pytest.mark.parametrize("N,H,W,Cc,act", [(2, 49, 81, 64, 1), (2, 50, 80, 16, 2), (1, 7, 9, 16, 2), (3, 8, 9, 16, 1), (1, 9, 8, 64, 2), (2, 1, 5, 16, 1)])
def test_bnact_maxpool(N, H, W, Cc, act):
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(N, Cc, H, W, generator=g, requires_grad=True)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    z = _act(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), act)
    y = F.max_pool2d(z, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dev = "cuda"
    X = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    Ho, Wo = y.shape[2:]
    SC, SH = sc.to(dev), sh.to(dev)          # keep the device copies alive across the asynchronous launches
    Y = torch.empty(N, Ho, Wo, Cc, device=dev)
    idx = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device=dev)
    check(L.rd_bnact_maxpool_fwd(ptr(X), ptr(SC), ptr(SH), act, N, H, W, Cc, ptr(Y), Cc, ptr(idx), current_stream()), "pool")
    G = torch.empty(N, H, W, Cc, device=dev)
    DY = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    check(L.rd_bnact_maxpool_bwd(ptr(DY), Cc, ptr(idx), ptr(X), ptr(SC), ptr(SH), act, N, H, W, Cc, ptr(G), current_stream()), "poolb")
    torch.cuda.synchronize()
    assert _rel(Y.permute(0, 3, 1, 2).cpu(), y.detach()) < 1e-6
    want_g = x.grad / sc.view(1, -1, 1, 1)          # kernel returns the gradient w.r.t. the BN output
    assert _rel(G.permute(0, 3, 1, 2).cpu(), want_g) < 1e-5
    # the variant that also produces the stem BatchNorm's backward sums: same g bit for bit, sums as the separate reduce pass
    mean = torch.randn(Cc, generator=g).to(dev)
    tiles = L.rd_bnact_maxpool_bwd_tiles(N, H, W, Cc)
    red = torch.zeros(tiles, 3, Cc, device=dev)
    G2 = torch.empty(N, H, W, Cc, device=dev)
    check(L.rd_bnact_maxpool_bwd_stats(ptr(DY), Cc, ptr(idx), ptr(X), ptr(SC), ptr(SH), act, N, H, W, Cc, ptr(G2), ptr(mean), ptr(red),
                                       current_stream()), "poolb_stats")
    torch.cuda.synchronize()
    assert torch.equal(G2, G)
    g64, x64 = G.double().reshape(-1, Cc), X.double().reshape(-1, Cc)
    s0, s1 = g64.sum(0), (g64 * (x64 - mean.double())).sum(0)
    got = red.double().sum(0)
    scale_ = g64.abs().sum(0).max().item() + 1e-30
    assert (got[0] - s0).abs().max().item() / scale_ < 1e-6 and (got[1] - s1).abs().max().item() / scale_ < 1e-5


@pytest.mark.parametrize("N,H,W,Ho,Wo", [(2, 64, 96, 97, 161), (1, 240, 400, 450, 800), (2, 5, 7, 5, 7)])
def test_head(N, H, W, Ho, Wo):
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, 16, H, W, generator=g, requires_grad=True)
    w = (torch.randn(1, 16, 3, 3, generator=g) * 0.1).requires_grad_(True)
    d = F.conv2d(x, w, padding=1)
    out = F.interpolate(d, size=(Ho, Wo), mode="bilinear", align_corners=True)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    dev = "cuda"
    X = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    Wt = w.detach().to(dev)
    D = torch.empty(N, H, W, device=dev)
    O = torch.empty(N, 1, Ho, Wo, device=dev)
    check(L.rd_head_conv_fwd(ptr(X), 16, ptr(Wt), N, H, W, 16, ptr(D), current_stream()), "head_fwd")
    check(L.rd_bilinear_fwd(ptr(D), N, H, W, ptr(O), Ho, Wo, current_stream()), "bil_fwd")
    DD = torch.empty(N, H, W, device=dev)
    GO = go.to(dev)
    check(L.rd_bilinear_bwd(ptr(GO), N, Ho, Wo, ptr(DD), H, W, current_stream()), "bil_bwd")
    DX = torch.empty(N, H, W, 16, device=dev)
    DW = torch.empty(1, 16, 3, 3, device=dev)
    ws = torch.empty(int(L.rd_head_conv_bwd_workspace_floats(N, H, W, 16)), device=dev)
    check(L.rd_head_conv_bwd(ptr(X), 16, ptr(Wt), ptr(DD), N, H, W, 16, ptr(DX), 16, ptr(DW), ptr(ws), current_stream()), "head_bwd")
    torch.cuda.synchronize()
    assert _rel(O.cpu(), out.detach()) < 5e-5   # lerp weights are fp32 on both sides; evaluation order differs
    assert _rel(DX.permute(0, 3, 1, 2).cpu(), x.grad) < 2e-5
    assert _rel(DW.cpu(), w.grad) < 5e-5


def test_losses_filter_sgd(golden_dir):
    import os

    import numpy as np

    from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss, SmoothnessLoss
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    u = np.load(os.path.join(golden_dir, "units.npz"))
    dev = "cuda"
    pred = torch.tensor(u["l1/pred"], device=dev, requires_grad=True)
    loss = MaskedL1Loss()(pred, torch.tensor(u["l1/target"], device=dev))
    loss.backward()
    assert abs(loss.item() - u["l1/loss"][0]) < 1e-5 * abs(u["l1/loss"][0])
    assert _rel(pred.grad.cpu(), torch.tensor(u["l1/grad"])) < 1e-6
    assert torch.isnan(MaskedL1Loss()(pred.detach(), torch.zeros_like(pred))).item()
    p = torch.tensor(u["smooth/pred"], device=dev, requires_grad=True)
    s = SmoothnessLoss()(p, torch.tensor(u["smooth/image"], device=dev))
    s.backward()
    assert abs(s.item() - u["smooth/loss"][0]) < 1e-5 * abs(u["smooth/loss"][0])
    assert _rel(p.grad.cpu(), torch.tensor(u["smooth/grad"])) < 1e-4
    # Filter_layer
    sparse, dense = torch.tensor(u["filter/sparse"]), torch.tensor(u["filter/dense"])
    n, _, h, w = sparse.shape
    x = torch.cat((torch.zeros(n, 3, h, w), sparse), 1).to(dev)
    kept, mask = torch.empty(n, 1, h, w, device=dev), torch.empty(n, 1, h, w, device=dev)
    DENSE = dense.to(dev)
    check(lib().rd_radar_filter(ptr(x), n, 4, 3, C.c_int64(h * w), ptr(DENSE), ptr(kept), ptr(mask), current_stream()), "filter")
    torch.cuda.synchronize()
    assert (mask.cpu().numpy() == u["filter/mask"]).all() and _rel(kept.cpu(), torch.tensor(u["filter/kept"])) < 1e-7
    # SGD: two steps against torch.optim.SGD
    g = torch.Generator().manual_seed(8)
    pw = torch.randn(1003, generator=g)
    ref = torch.nn.Parameter(pw.clone())
    opt = torch.optim.SGD([ref], 0.01, momentum=0.9, weight_decay=1e-4)
    P, Buf = torch.zeros(1004, device=dev), torch.zeros(1004, device=dev)
    P[:1003] = pw.to(dev)
    for it in range(2):
        gr = torch.randn(1003, generator=g)
        ref.grad = gr.clone()
        opt.step()
        Gd = torch.zeros(1004, device=dev)
        Gd[:1003] = gr.to(dev)
        check(lib().rd_sgd_step(ptr(P), ptr(Gd), ptr(Buf), C.c_int64(1003), C.c_float(0.01), C.c_float(0.9), C.c_float(1e-4), C.c_float(1.0), 0,
                                current_stream()), "sgd")
    torch.cuda.synchronize()
    assert _rel(P[:1003].cpu(), ref.detach()) < 1e-6


def test_depth_metrics(golden_dir):
    """Result.evaluate / AverageMeter (evaluation/metrics.py:34-58,179-216) against vectors from the real reference."""
    import os

    import numpy as np

    from radar_depth_amd.evaluation.metrics import AverageMeter, Result
    want = np.load(os.path.join(golden_dir, "metrics.npz"))
    out, tgt = torch.tensor(want["out"]).cuda(), torch.tensor(want["target"]).cuda()
    names = [str(n) for n in want["names"]]
    r1, r2 = Result(), Result()
    r1.evaluate(out, tgt)
    r2.evaluate(out * 1.1 + 0.3, tgt)
    for r, key in ((r1, "r1"), (r2, "r2")):
        got = np.array([getattr(r, n) for n in names])
        assert np.abs(got - want[key]).max() / np.abs(want[key]).max() < 2e-5, (key, got, want[key])
    m = AverageMeter()
    m.update(r1, 0.5, 0.1, 2)
    m.update(r2, 0.7, 0.2, 3)
    a = m.average()
    got = np.array([getattr(a, n) for n in names] + [a.gpu_time, a.data_time])
    assert np.abs(got - want["avg"]).max() / np.abs(want["avg"]).max() < 2e-5
