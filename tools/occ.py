import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from radar_depth_amd import convdesc as cd
from radar_depth_amd._lib import lib
L = lib(); torch.zeros(1, device="cuda")
for (ci, co, k, s, p, h, w) in [(64, 64, 3, 1, 1, 113, 200), (256, 256, 3, 1, 1, 29, 50), (512, 512, 3, 1, 1, 15, 25), (16, 16, 3, 1, 1, 240, 400)]:
    d = cd.conv_fwd(16, h, w, ci, co, k, s, p)
    info = (C.c_int32 * 10)(); L.rd_gconv_plan_info(C.byref(d), info)
    print((ci, co, h, w), list(info), "occupancy blocks/CU:", L.rd_gconv_occupancy(C.byref(d)))
