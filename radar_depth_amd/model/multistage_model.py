"""MI355X-native counterpart of the reference's model/multistage_model.py:

  ResNet_multistage  <- /root/reference/model/multistage_model.py:22-83
  Filter_layer       <- :87-119
  ResNet_latefusion2 <- :123-276

Same constructor signatures, attribute names, output dict and state_dict keys.  The two stages run as two static
HIP plans that share the network input; the radar filter is one HIP kernel; the stage-2 loss back-propagates into
stage 1 through the dense-depth input channel of stage 2's depth stem exactly as in the reference (the stage-1
prediction is not detached, :75).

Deliberate differences: the inner stages never try to download ImageNet weights
(the reference hard-codes pretrained=True for torchvision, :29-30); Filter_layer's constants are plain floats.
"""
import ctypes as C
import os
import weakref

import torch
import torch.nn as nn

from .._lib import check, lib, ptr
from .models import ArenaOwner, ResNet_latefusion, _DEPTHS, _PlanOnly


class ResNet_latefusion2(ResNet_latefusion):
    """Late fusion whose depth stem takes in_channels-3 inputs (:163-164)."""

    def _depth_inputs(self):
        return self.in_channels - 3

    def forward(self, x):
        if self.in_channels != 4:
            raise RuntimeError("a 5-channel ResNet_latefusion2 runs as stage 2 of ResNet_multistage (its depth stem reads the "
                               "filtered radar map and the stage-1 prediction in place)")
        return super().forward(x)


class Filter_layer(_PlanOnly):
    """mask = |dense - sparse| <= 5 * 3.6^(dense/100); returns (sparse*mask, mask).  Executed by rd_radar_filter."""
    ALPHA, BETA, K = 5.0, 18.0, 100.0

    def forward(self, sparse_depth, dense_depth):
        if not sparse_depth.is_cuda:
            raise RuntimeError("radar_depth_amd modules run on MI355X only (HIP kernels)")
        sparse = sparse_depth.contiguous().float()
        dense = dense_depth.contiguous().float()
        n, c, h, w = sparse.shape
        assert c == 1 and dense.shape == sparse.shape
        kept, mask = torch.empty_like(sparse), torch.empty_like(sparse)
        check(lib().rd_radar_filter(ptr(sparse), n, 1, 0, C.c_int64(h * w), ptr(dense), ptr(kept), ptr(mask),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rd_radar_filter")
        return kept, mask


class ResNet_multistage(ArenaOwner, nn.Module):
    def __init__(self, layers, decoder, output_size, pretrained=True, project_root="YOUR_PATH/radar_depth"):
        if layers not in _DEPTHS:
            raise RuntimeError("Only 18, 34, 50, 101, and 152 layer model are defined for ResNet. Got {}".format(layers))
        super().__init__()
        self.stage1 = ResNet_latefusion2(layers, decoder, output_size, in_channels=4, pretrained=False)
        self.stage2 = ResNet_latefusion2(layers, decoder, output_size, in_channels=5, pretrained=False)
        self.filter_layer = Filter_layer()
        for child in (self.stage1, self.stage2):
            child.__dict__["_arena_owner_ref"] = weakref.ref(self)
        self.output_size = output_size
        if pretrained is True:
            path = os.path.join(project_root, "pretrained/resnet18_latefusion.pth.tar")
            if not os.path.exists(path):
                raise ValueError("[Error] Can't find pretrained latefusion model. "
                                 "Please follow the instructions in README.md to download the weights!")
            from ..utils import load_checkpoint      # reference .pth.tar: pickled Namespace + Result next to the state_dict
            weights = load_checkpoint(path)["model_state_dict"]
            self.stage1.load_state_dict(weights)
            self.stage2.load_state_dict(self.filter_state_dict(weights, self.stage2.state_dict()), strict=False)
        self.__dict__["_ms_plans"] = {}
        self._adopt_arena_children()

    @staticmethod
    def filter_state_dict(pretrain_dict, target_dict):
        return {k: v for k, v in pretrain_dict.items() if target_dict[k].shape == v.shape}

    # ------------------------------------------------------------------ HIP execution
    def _plans(self, batch, height, width, train, bf16=False, storage="fp32", segment_joins=True, autotune=None, split=False):
        from ..engine import LateFusionPlan
        assert list(self.output_size) == [height, width], "the multistage net feeds its stage-1 output back as an input map: " \
            "output_size must equal the input size"
        st = self._ensure_arenas()
        key = (batch, height, width, bool(train), st["version"], bool(bf16), storage, bool(segment_joins), bool(split))
        cache = self.__dict__.setdefault("_ms_plans", {})
        if key in cache:
            cache[key] = cache.pop(key)
        else:
            from .models import _evict_plans
            _evict_plans(cache, st["version"])
            p1 = LateFusionPlan(self.stage1, batch, height, width, train=train, bf16=bf16, storage=storage, segment_joins=segment_joins,
                                autotune=autotune, split=split)
            dev = p1.dev
            kept = torch.empty(batch, 1, height, width, device=dev)
            mask = torch.empty(batch, 1, height, width, device=dev)
            p2 = LateFusionPlan(self.stage2, batch, height, width, train=train, depth_planes=[kept, p1.pred], x_source=p1.x_in,
                                dense_grad_dst=p1.dpred if train else None, bf16=bf16, storage=storage, segment_joins=segment_joins,
                                autotune=autotune, split=split)
            cache[key] = MultistagePlan(p1, p2, kept, mask)
        return cache[key]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("radar_depth_amd modules run on MI355X only (HIP kernels); got a %s tensor" % x.device.type)
        assert x.dim() == 4 and x.shape[1] >= 4
        x = x.contiguous().float()
        from .models import eager_operands
        mp = self._plans(x.shape[0], x.shape[2], x.shape[3], self.training, split=self.training and eager_operands(self) == "split")
        if self.training and torch.is_grad_enabled():
            params = [p for p in self._ensure_arenas()["params"]]
            d1, d2 = _MultistageFunction.apply(mp, x, *params)
        else:
            mp.run_forward(x)
            d1, d2 = mp.p1.pred.clone(), mp.p2.pred.clone()
        return {"stage1": d1, "stage2": d2, "mask": mp.mask.clone(), "radar_filtered": mp.kept.clone()}


class MultistagePlan:
    """Two LateFusionPlans + the radar filter between them (forward :63-83; backward couples stage 2 into stage 1)."""

    def __init__(self, p1, p2, kept, mask):
        self.p1, self.p2, self.kept, self.mask = p1, p2, kept, mask
        self.L = lib()

    def filter_args(self):
        """The radar filter between the stages as a plan op (name, C-ABI function, arguments) on stage 1's main stream."""
        p1 = self.p1
        hw = p1.H * p1.W
        return ("radar_filter", self.L.rd_radar_filter, (ptr(p1.x_in), p1.N, p1.x_in.shape[1], 3, C.c_int64(hw), ptr(p1.pred), ptr(self.kept),
                                                         ptr(self.mask), p1.streams[0]))

    def filter_op(self):
        name, fn, args = self.filter_args()
        check(fn(*args), name)

    def run_forward(self, x=None):
        self.p1.run_forward(x)
        self.p2.set_stream()
        self.filter_op()
        self.p2.run_forward(None)

    def run_backward(self, g1, g2):
        """g1, g2: gradients w.r.t. the stage-1 / stage-2 predictions (either may be None)."""
        p1, p2 = self.p1, self.p2
        p2.set_stream()
        p1.set_stream()
        if g2 is not None:
            p2.dpred.copy_(g2)
        else:
            p2.dpred.zero_()
        p2.run_list("bwd")                   # writes d(loss)/d(stage-1 prediction as stage-2 input) into p1.dpred
        if g1 is not None:
            p1.dpred.add_(g1.view_as(p1.dpred))
        p1.run_list("bwd")


class _MultistageFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mp, x, *params):
        from .models import hold_plan
        ctx.mp, ctx.params = hold_plan(mp, ctx), params
        mp.run_forward(x)
        ctx.generation = (mp.p1.generation, mp.p2.generation)
        return mp.p1.pred.clone(), mp.p2.pred.clone()

    @staticmethod
    def backward(ctx, g1, g2):
        mp = ctx.mp
        if (mp.p1.generation, mp.p2.generation) != ctx.generation:
            raise RuntimeError("these plans ran another forward since the one being differentiated (static activation buffers): "
                               "call backward() before the next training-mode forward of the same (batch, size)")
        mp.run_backward(None if g1 is None else g1.contiguous(), None if g2 is None else g2.contiguous())
        root = mp.p1.m._arena_root()
        staged = {id(p) for p in mp.p1.m.parameters()} | {id(p) for p in mp.p2.m.parameters()}
        gv = root._ensure_arenas()["gviews"]
        grads = [gv[id(p)] if id(p) in staged else None for p in ctx.params]
        return (None, None) + tuple(grads)
