"""Host logic of the bf16-operand kernels, no GPU: the planners behind rd_gconv_bf16 / rd_wgrad_bf16 are pure functions of the
descriptor, so their decisions can be pinned on CPU -- every convolution of the network gets a feasible plan inside the LDS
budget, and the weight-gradient decomposition into passes is what csrc/wgrad_bf16.hip documents."""
import ctypes as C

import pytest

from radar_depth_amd import convdesc as cd
from radar_depth_amd._lib import lib

# (Cin, Cout, k, stride, pad, H, W) of every gconv-lowered convolution of resnet18_latefusion at 450x800
CONVS = [(64, 64, 3, 1, 1, 113, 200), (64, 128, 3, 2, 1, 113, 200), (128, 128, 3, 1, 1, 57, 100), (64, 128, 1, 2, 0, 113, 200),
         (128, 256, 3, 2, 1, 57, 100), (256, 256, 3, 1, 1, 29, 50), (256, 512, 3, 2, 1, 29, 50), (512, 512, 3, 1, 1, 15, 25),
         (16, 16, 3, 1, 1, 113, 200), (16, 32, 3, 2, 1, 113, 200), (32, 32, 3, 1, 1, 57, 100), (64, 64, 3, 1, 1, 29, 50),
         (128, 128, 3, 1, 1, 15, 25), (640, 512, 1, 1, 0, 15, 25), (512, 256, 1, 1, 0, 15, 25), (128, 128, 3, 1, 1, 30, 50),
         (64, 64, 3, 1, 1, 60, 100), (32, 32, 3, 1, 1, 120, 200), (16, 16, 3, 1, 1, 240, 400)]
UPPROJ = [(256, 15, 25), (128, 30, 50), (64, 60, 100), (32, 120, 200)]


def _descs(batch):
    for ci, co, k, s, p, h, w in CONVS:
        yield "conv%dx%d s%d %d->%d" % (k, k, s, ci, co), cd.conv_fwd(batch, h, w, ci, co, k, s, p)
        yield "dgrad%dx%d s%d %d->%d" % (k, k, s, ci, co), cd.conv_dgrad(batch, h, w, ci, co, k, s, p)[0]
    for c, h, w in UPPROJ:
        yield "upproj %d" % c, cd.upproj_fwd(batch, h, w, c, c)
        yield "upproj dgrad %d" % c, cd.upproj_dgrad(batch, h, w, c, c)


@pytest.mark.parametrize("batch", [1, 16])
def test_every_layer_has_a_bf16_plan_inside_the_lds_budget(batch):
    L = lib()
    for name, d in _descs(batch):
        out = (C.c_int32 * 8)()
        assert L.rd_gconv_bf16_plan_info(C.byref(d), out) == 0, name
        mt, nt, ckp, th, tw, pp, lds, wgs = out[0], out[1], out[2] % 1000, out[3], out[4], out[5], out[6], out[7]
        assert (mt, nt) in {(3, 2), (2, 2), (2, 1), (1, 2), (1, 1)} and ckp in (16, 32, 64), name
        assert d.Cin % ckp == 0 and th * tw <= 128 * mt and 0 < lds <= 160 * 1024 - 512 and wgs > 0, (name, list(out))
        if out[2] >= 1000:        # the software-pipelined loop is reserved for long reductions
            assert d.Cin >= 320, name
        assert L.rd_gconv_bf16_stat_tiles(C.byref(d)) > 0


def test_wgrad_bf16_pass_decomposition():
    L = lib()
    info = (C.c_int32 * 8)()

    def passes(d):
        assert L.rd_wgrad_bf16_supported(C.byref(d)) == 1
        assert L.rd_wgrad_bf16_plan_info(C.byref(d), info) == 0
        assert 0 < info[5] <= 160 * 1024          # two LDS tile buffers
        return info[6]
    assert passes(cd.conv_fwd(16, 113, 200, 64, 64, 3, 1, 1)) == 1       # stride-1 3x3: one pass, nine taps
    assert passes(cd.conv_fwd(16, 113, 200, 64, 128, 3, 2, 1)) == 4      # stride 2: four input-parity passes (4/2/2/1 taps)
    assert passes(cd.conv_fwd(16, 113, 200, 64, 128, 1, 2, 0)) == 1      # 1x1
    assert passes(cd.conv_fwd(16, 15, 25, 640, 512, 1, 1, 0)) == 1
    assert passes(cd.upproj_fwd(16, 60, 100, 64, 64)) == 4               # UpProj: its four parity phases (9/6/6/4 taps)
    # slab count = splits x (4 / tile pairs of a channel block): 64x64 channels -> 4 pairs -> one slab per split
    assert L.rd_wgrad_bf16_plan_info(C.byref(cd.conv_fwd(16, 113, 200, 64, 64, 3, 1, 1)), info) == 0 and info[4] == info[3]
    assert L.rd_wgrad_bf16_plan_info(C.byref(cd.conv_fwd(16, 57, 100, 32, 32, 3, 1, 1)), info) == 0 and info[4] == 4 * info[3]
    # what does not decompose stays on the fp32 kernel
    for d in (cd.conv_fwd(2, 32, 32, 24, 32, 3, 1, 1), cd.conv_fwd(2, 32, 32, 32, 64, 5, 1, 2)):
        assert L.rd_wgrad_bf16_supported(C.byref(d)) == 0
        assert L.rd_wgrad_bf16_workspace_floats(C.byref(d)) < 0
    # workspace = (slabs + 16 reduction rows) x taps x Cin x Cout
    d = cd.upproj_fwd(2, 16, 16, 64, 64)
    assert L.rd_wgrad_bf16_plan_info(C.byref(d), info) == 0
    assert L.rd_wgrad_bf16_workspace_floats(C.byref(d)) == (info[4] + 16) * 25 * 64 * 64


def test_bf16_plan_flag_reaches_the_plan(tmp_path):
    """Dry-run plans (no launches): operands='bf16' routes convolutions to the bf16 entry points, the 16-channel weight gradients
    and the depth stem stay on the fp32 kernels."""
    import torch
    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.model.models import ResNet_latefusion
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [97, 161], 4, False)
    plan = LateFusionPlan(m, 2, 97, 161, train=True, dry_run=True, bf16=True)
    L = lib()

    def fn_names(ops):
        return [getattr(fn, "__name__", str(fn)) for _, fn, _ in ops]
    fwd, bwd = fn_names(plan.fwd), fn_names(plan.bwd)
    assert "rd_gconv_bf16_t" in fwd and "rd_gconv_ws" not in fwd and "rd_gconv_ws" not in bwd
    assert fwd.count("rd_stem_fwd_bf16_t") == 1 and fwd.count("rd_stem_fwd_t") == 1      # RGB stem bf16, depth stem fp32
    assert "rd_wgrad_bf16_t" in bwd and "rd_wgrad" in bwd                                 # >= 32 channels vs 16-channel layers
    fam = {k: v[0] for k, v in plan.meta.items()}
    assert fam["layer1.0.conv1.wgrad"] == "wgrad_bf16" and fam["layer1_depth.0.conv1.wgrad"] == "wgrad"
    assert plan.dt == 0 and plan.cat.t.dtype == torch.float32                             # bf16 OPERANDS: tensors stay fp32
    plan32 = LateFusionPlan(m, 2, 97, 161, train=True, dry_run=True)
    assert not any("bf16" in n for n in fn_names(plan32.fwd) + fn_names(plan32.bwd))
    # bf16 STORAGE: every NHWC tensor is bf16, every weight gradient runs on the bf16 kernel (nothing else reads bf16 tensors),
    # every storage-typed op carries dtype 1 as its first argument
    plan16 = LateFusionPlan(m, 2, 97, 161, train=True, dry_run=True, storage="bf16")
    assert plan16.bf16 and plan16.dt == 1 and plan16.cat.t.dtype == torch.bfloat16 and plan16.z.t.dtype == torch.bfloat16
    assert plan16.pred.dtype == torch.float32 and plan16.x_in.dtype == torch.float32
    bwd16 = fn_names(plan16.bwd)
    assert "rd_wgrad" not in bwd16 and "rd_wgrad_reduce" not in bwd16 and bwd16.count("rd_wgrad_bf16_t") == bwd.count("rd_wgrad_bf16_t") + bwd.count("rd_wgrad")
    for _, fn, args in plan16.fwd + plan16.bwd:
        if getattr(fn, "__name__", "").endswith("_t"):
            assert args[0] == 1
    assert plan16.taps["layer1.0"].ptr.value - plan16.taps["layer1.0"].t.data_ptr() == 0
    assert plan16.cat.chan(512, 128).ptr.value - plan16.cat.t.data_ptr() == 2 * 512
