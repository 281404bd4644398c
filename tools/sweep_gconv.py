"""Sweep the wave-tiling configs of gconv per layer shape (RD_GCONV_FORCE) -> time per config; run as
   for c in 0..4: RD_GCONV_FORCE=$c python tools/sweep_gconv.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import lib
B, dev = 16, "cuda"
L = lib()
SH = [("layer1", "c", 64, 64, 3, 1, 113, 200), ("l2.0c1", "c", 64, 128, 3, 2, 113, 200), ("layer2", "c", 128, 128, 3, 1, 57, 100),
      ("layer3", "c", 256, 256, 3, 1, 29, 50), ("layer4", "c", 512, 512, 3, 1, 15, 25), ("d.l1", "c", 16, 16, 3, 1, 113, 200),
      ("d.l2", "c", 32, 32, 3, 1, 57, 100), ("dec4c2", "c", 16, 16, 3, 1, 240, 400), ("dec3c2", "c", 32, 32, 3, 1, 120, 200),
      ("fusion", "c", 640, 512, 1, 1, 15, 25), ("up256", "u", 256, 256, 5, 1, 15, 25), ("up64", "u", 64, 64, 5, 1, 60, 100),
      ("up32", "u", 32, 32, 5, 1, 120, 200), ("dup256", "du", 256, 256, 5, 1, 15, 25), ("dup64", "du", 64, 64, 5, 1, 60, 100)]
out = []
for name, kind, ci, co, k, s, h, w in SH:
    if kind == "c":
        d = cd.conv_fwd(B, h, w, ci, co, k, s, k // 2); xs = (B, h, w, ci); S = k * k
    elif kind == "u":
        d = cd.upproj_fwd(B, h, w, ci, co); xs = (B, h, w, ci); S = 25
    else:
        d = cd.upproj_dgrad(B, h, w, ci, co); xs = (B, 2 * h, 2 * w, co); S = 25
    info = (C.c_int32 * 10)()
    if L.rd_gconv_plan_info(C.byref(d), info) != 0:
        out.append("%s:-" % name); continue
    x = torch.randn(*xs, device=dev); wp = torch.randn(S, d.Cin, d.Cout, device=dev); y = torch.empty(B, d.Ho, d.Wo, d.Cout, device=dev)
    import time
    w0 = time.perf_counter()
    while time.perf_counter() - w0 < 0.05:      # let the device clock ramp (cold launches run ~13 % slower)
        for _ in range(20): ops.gconv(d, x, wp, y)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gconv(d, x, wp, y)
    e1.record(); torch.cuda.synchronize()
    out.append("%s:%.0f(%dx%d,%d)" % (name, e0.elapsed_time(e1) * 100, info[6], info[7], info[5]))
print("cfg", os.environ.get("RD_GCONV_FORCE", "auto"), " ".join(out))
