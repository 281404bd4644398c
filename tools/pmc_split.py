"""Tiny driver for rocprofv3 --pmc runs: a few launches of the split kernels on the dominant shapes (forward, weight gradient, B=16).
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \\
              SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o p -- python tools/pmc_split.py
    python tools/pmc_mfma.py $OUT        (a second pass with SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE for the LDS conflicts)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radar_depth_amd import convdesc as cd, ops  # noqa: E402

B = 16
for (ci, co, h, w) in [(64, 64, 113, 200), (128, 128, 57, 100), (256, 256, 29, 50), (512, 512, 15, 25)]:
    d = cd.conv_fwd(B, h, w, ci, co, 3, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda")
    ws = ops.pack_weights_split(wt)
    y = torch.empty(B, h, w, co, device="cuda")
    dy = torch.randn(B, h, w, co, device="cuda")
    slabs = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    xp = ops.split_pieces(x)
    u = ops.wino_pack(wt)
    for _ in range(3):
        ops.gconv_split(d, x, ws, y)
        ops.gconv_split_pre(d, xp, ws, y)          # gconv_sp2_kernel (pre-split input, two workgroups per CU)
        ops.wgrad_split(d, x, dy, slabs)
        ops.wino_conv3x3(x, u, y)                  # wino_split_kernel (round 6)
torch.cuda.synchronize()
