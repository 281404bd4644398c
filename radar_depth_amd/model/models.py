"""MI355X-native counterpart of the reference's model/models.py (hot-path subset).

Same public surface -- Unpool, weights_init*, BasicBlock, Decoder, UpProj, choose_decoder,
ResNet_latefusion with the reference's constructor signatures, attribute names and state_dict keys
(/root/reference/model/models.py:13-27,30-72,75-133,178-230,519-664) -- but the arithmetic runs in
hand-written HIP kernels through the C ABI (include/radar_depth_hip.h):

  * the child nn.Conv2d / nn.BatchNorm2d objects are parameter containers (OIHW weights, BN affine and
    running statistics) so that checkpoints, `weights_init`-style initialisers and optimizers see exactly
    the reference's tensors;
  * `ResNet_latefusion.forward` runs a static plan (radar_depth_amd/engine.py) of HIP kernels and is
    differentiable through torch.autograd (`loss.backward()` fills `.grad` of every parameter);
  * there is NO CPU / PyTorch fallback: calling forward on a CPU tensor or without the built library raises.

Out of scope here (other --arch/--decoder choices of the reference): ResNet, ResNet_pnp, ResNet2,
ResNet_multifusion, DeConv, UpConv; `choose_decoder` rejects them.
"""
import math
import os
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

_DEPTHS = (18, 34, 50, 101, 152)


def _conv(cin, cout, k, stride=1, pad=None):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k // 2 if pad is None else pad), bias=False)


class _PlanOnly(nn.Module):
    """Container whose arithmetic is executed by the parent network's HIP plan."""

    def forward(self, *args, **kwargs):
        raise RuntimeError("%s is a parameter container: its arithmetic runs inside the parent network's HIP plan "
                           "(call the ResNet_latefusion / ResNet_multistage module)" % type(self).__name__)


class Unpool(nn.Module):
    """Zero-stuffing x2 upsample (models.py:13-27).  Holds no tensor: inside a network the HIP path folds it into the following
    5x5 convolution (four-phase zero-skipping form), so there is nothing to move with .cuda()/.to().  Called on its own it is pure
    data movement -- the input lands on every stride-th pixel of a zero buffer (what the reference's conv_transpose2d with a one-hot
    kernel computes) -- with no arithmetic and therefore no kernel of its own; autograd differentiates the strided copy."""

    def __init__(self, num_channels, stride=2):
        super().__init__()
        self.num_channels, self.stride = num_channels, stride

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("radar_depth_amd modules run on MI355X only (HIP kernels); got a %s tensor" % x.device.type)
        assert x.dim() == 4 and x.shape[1] == self.num_channels
        s = self.stride
        out = x.new_zeros(x.shape[0], x.shape[1], x.shape[2] * s, x.shape[3] * s)
        out[:, :, ::s, ::s] = x
        return out


def weights_init(m):
    if isinstance(m, nn.Conv2d):
        fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        m.weight.data.normal_(0, math.sqrt(2.0 / fan))
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


def _kaiming(m, nonlin):
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity=nonlin)
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)


def weights_init_kaiming(m):
    _kaiming(m, "relu")


def weights_init_kaiming_leaky(m):
    _kaiming(m, "leaky_relu")


PLAN_CACHE_SIZE = int(os.environ.get("RD_PLAN_CACHE", "3"))


def eager_operands(module):
    """Arithmetic of the convolutions when a module is CALLED (`pred = model(x)`, the reference's main.py:440) rather than stepped by
    HipTrainStep: "split" (default since round 6 -- fp32 arithmetic on the bf16 matrix cores, the plan the headline is measured on) or
    "fp32" (every convolution on the fp32 MFMA).  Per module: `model.operands = "fp32"` (looked up on the arena root, so it can be
    set on the network and holds for its sub-modules); process-wide default: RD_EAGER_OPERANDS."""
    root = module._arena_root() if hasattr(module, "_arena_root") else module
    mode = getattr(module, "operands", None) or getattr(root, "operands", None) or os.environ.get("RD_EAGER_OPERANDS", "split")
    if mode not in ("split", "fp32"):
        raise ValueError("operands must be 'split' or 'fp32', got %r" % (mode,))
    return mode


def _close_plan(plan):
    """Free a plan's ~11 GB and its hipEvents NOW (instead of whenever the garbage collector gets to an object that sits in a reference
    cycle).  The plan's side streams are not known to torch's caching allocator: nothing of the plan may still be running when its
    buffers go back to the allocator and the next plan zeroes / uploads into the same blocks."""
    for pl in ([plan.p1, plan.p2] if hasattr(plan, "p1") else [plan]):
        if hasattr(pl, "close"):
            for st in (getattr(pl, "_side", None) or []):
                st.synchronize()
            if not getattr(pl, "dry_run", True):
                torch.cuda.current_stream().synchronize()
            pl.close()
            pl.keep = []


def hold_plan(plan, holder=None):
    """Explicit holder count of a cached plan (a LateFusionPlan or a MultistagePlan): a HipTrainStep / HipInference holds its plan from
    construction to close(), an autograd node from its forward until the node is collected (`holder`: the object whose lifetime is the
    hold -- released by a finalizer).  A plan the cache has evicted is closed by its LAST holder's release, never under a live one
    (ADVICE r4 / VERDICT r5 #13: this replaces a sys.getrefcount() test)."""
    plan.__dict__["holders"] = plan.__dict__.get("holders", 0) + 1
    if holder is not None:
        weakref.finalize(holder, release_plan, plan)
    return plan


def release_plan(plan):
    n = plan.__dict__.get("holders", 0) - 1
    plan.__dict__["holders"] = max(n, 0)
    if n <= 0 and plan.__dict__.get("evicted"):
        try:
            _close_plan(plan)
        except Exception:                 # (interpreter shutdown: the runtime may already be gone)
            pass


def _evict_plans(cache, version, room_for=1):
    """A plan owns every activation / gradient / workspace buffer of one (batch, size, mode) key -- about 11 GB at b=16
    450x800 training.  The cache is a small LRU (the steady batch, a ragged last batch, validate()'s batch 1): plans of a
    rebuilt parameter arena go first, then the least recently used.  An evicted plan nobody holds (hold_plan) is closed at once --
    ragged last batches plus validate() at 900x1600 could otherwise briefly keep more than PLAN_CACHE_SIZE plans alive; one that a
    HipTrainStep / HipInference / autograd node still holds goes with its last holder (release_plan)."""
    stale = [k for k in cache if k[4] != version]
    for k in stale + [k for k in cache if k not in stale][:max(0, len(cache) - len(stale) + room_for - PLAN_CACHE_SIZE)]:
        plan = cache.pop(k)
        plan.__dict__["evicted"] = True
        if plan.__dict__.get("holders", 0) <= 0:
            _close_plan(plan)


# ------------------------------------------------------------------------------------------------
class ArenaOwner:
    """Mixin: keeps every parameter of the (top-level) network in one flat fp32 arena, with matching flat
    gradient and momentum arenas.  Parameters remain ordinary nn.Parameters (views into the arena), so
    state_dict / load_state_dict / external optimizers work unchanged, while the HIP SGD kernel and the RCCL
    gradient all-reduce see one contiguous buffer each."""

    REHOMES = [0]      # bumped whenever ANY root (re)builds its arenas: a HipTrainStep re-checks every parameter pointer when it moved

    def _adopt_arena_children(self):
        """The network is the arena owner of every ArenaOwner below it (BasicBlock, UpProj, UpProjModule; the stages of the multistage
        net): a stand-alone call of a child -- model.layer1[0](x), model.decoder(x) -- then runs on the parent's arenas instead of
        re-homing the child's parameters into an arena of its own behind a live HipTrainStep's back."""
        for mod in self.modules():
            if mod is not self and isinstance(mod, ArenaOwner) and "_arena_owner_ref" not in mod.__dict__:
                mod.__dict__["_arena_owner_ref"] = weakref.ref(self)

    def _arena_root(self):
        ref = getattr(self, "_arena_owner_ref", None)
        owner = ref() if ref is not None else None
        return owner._arena_root() if owner is not None else self

    def _ensure_arenas(self):
        root = self._arena_root()
        params = [p for _, p in root.named_parameters()]
        st = root.__dict__.get("_arena_state")
        if st is not None and len(st["ptrs"]) == len(params) and all(p.data_ptr() == q for p, q in zip(params, st["ptrs"])):
            return st
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4           # 16-byte aligned slots
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        grads = torch.zeros(total, dtype=torch.float32, device=dev)
        mom = torch.zeros(total, dtype=torch.float32, device=dev)
        gviews = {}
        for p, off in zip(params, offs):
            v = arena[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            gviews[id(p)] = grads[off:off + p.numel()].view(p.shape)
        st = dict(arena=arena, grads=grads, mom=mom, gviews=gviews, ptrs=[p.data_ptr() for p in params], params=params,
                  version=(root.__dict__.get("_arena_state") or {}).get("version", 0) + 1, total=total)
        root.__dict__["_arena_state"] = st
        ArenaOwner.REHOMES[0] += 1
        return st

    def _grad_view(self, param):
        return self._arena_root()._ensure_arenas()["gviews"][id(param)]


class _ModuleFunction(torch.autograd.Function):
    """Bridges a ModulePlan (one sub-module run stand-alone) into torch.autograd, like _PlanFunction does for a network."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.plan, ctx.params = plan, params
        out = plan.forward_nchw(x)
        ctx.generation = plan.generation
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("this module ran another forward since the one being differentiated: its saved activations are static "
                               "buffers, so call backward() before the next training-mode forward of the same input shape")
        dx = plan.backward_nchw(gout.contiguous().float())
        gv = plan.m._arena_root()._ensure_arenas()["gviews"]      # (looked up ONCE: _grad_view walks the module tree on every call)
        return (None, dx) + tuple(gv[id(p)] for p in ctx.params)


class _StandaloneForward:
    """Mixin of the contract's sub-modules (BasicBlock, UpProj.UpProjModule, UpProj): inside a network they are parameter containers
    whose arithmetic runs in the parent's plan; called on their own -- the reference's are ordinary callable nn.Modules,
    models.py:96-112,199-216 -- they build (and cache per input shape / mode) a one-module plan from the same op builders
    (engine.ModulePlan) and run it: differentiable in training mode (batch statistics, running statistics updated), folded-BN
    inference in eval mode.  The module becomes the root of its own parameter arena for that; a parent network that runs afterwards
    notices and re-homes its parameters (ArenaOwner._ensure_arenas)."""
    _plan_kind = None

    def forward(self, x):
        from ..engine import ModulePlan
        if not x.is_cuda:
            raise RuntimeError("radar_depth_amd modules run on MI355X only (HIP kernels); got a %s tensor" % x.device.type)
        assert x.dim() == 4
        x = x.contiguous().float()
        st = self._ensure_arenas()          # (the arenas of the network this module belongs to, or its own when it was built alone)
        train = bool(self.training)
        split = train and eager_operands(self) == "split"      # (eval plans fold BatchNorm into fp32 weights: the fp32 kernels)
        key = (tuple(x.shape), train, st["version"], split)
        plans = self.__dict__.setdefault("_module_plans", {})
        if key not in plans:
            for k in [k for k in plans if k[2] != st["version"]] + list(plans)[:max(0, len(plans) + 1 - PLAN_CACHE_SIZE)]:
                plans.pop(k, None)
            plans[key] = ModulePlan(self, self, self._plan_kind, x.shape[0], x.shape[2], x.shape[3], x.shape[1], train=train, split=split)
        else:
            plans[key] = plans.pop(key)
        plan = plans[key]
        if train and torch.is_grad_enabled():
            return _ModuleFunction.apply(plan, x, *list(self.parameters()))
        return plan.forward_nchw(x)


class BasicBlock(_StandaloneForward, ArenaOwner, _PlanOnly):
    expansion = 1
    _plan_kind = "block"

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = _conv(inplanes, planes, 3, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(planes, planes, 3)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


def _make_layer(inplanes, planes, blocks, stride, init=None):
    down = None
    if stride != 1 or inplanes != planes:
        down = nn.Sequential(_conv(inplanes, planes, 1, stride, pad=0), nn.BatchNorm2d(planes))
    seq = nn.Sequential(BasicBlock(inplanes, planes, stride, down), *[BasicBlock(planes, planes) for _ in range(1, blocks)])
    if init is not None:
        for mod in seq.modules():
            init(mod)
    return seq


class Decoder(_PlanOnly):
    names = ["deconv2", "deconv3", "upconv", "upproj"]

    def __init__(self):
        super().__init__()
        self.layer1 = self.layer2 = self.layer3 = self.layer4 = None


class UpProj(_StandaloneForward, ArenaOwner, Decoder):
    _plan_kind = "decoder"

    class UpProjModule(_StandaloneForward, ArenaOwner, _PlanOnly):
        _plan_kind = "upproj"

        def __init__(self, in_channels):
            super().__init__()
            half = in_channels // 2
            self.unpool = Unpool(in_channels)
            self.upper_branch = nn.Sequential(OrderedDict([
                ("conv1", _conv(in_channels, half, 5)),
                ("batchnorm1", nn.BatchNorm2d(half)),
                ("relu", nn.ReLU()),
                ("conv2", _conv(half, half, 3)),
                ("batchnorm2", nn.BatchNorm2d(half)),
            ]))
            self.bottom_branch = nn.Sequential(OrderedDict([
                ("conv", _conv(in_channels, half, 5)),
                ("batchnorm", nn.BatchNorm2d(half)),
            ]))
            self.relu = nn.ReLU()

    def __init__(self, in_channels):
        super().__init__()
        self.layer1 = self.UpProjModule(in_channels)
        self.layer2 = self.UpProjModule(in_channels // 2)
        self.layer3 = self.UpProjModule(in_channels // 4)
        self.layer4 = self.UpProjModule(in_channels // 8)


def choose_decoder(decoder, in_channels):
    if decoder == "upproj":
        return UpProj(in_channels)
    assert False, "invalid option for decoder: {}".format(decoder)


class ResNet_latefusion(ArenaOwner, nn.Module):
    def __init__(self, layers, decoder, output_size, in_channels=4, pretrained=True):
        if layers not in _DEPTHS:
            raise RuntimeError("Only 18, 34, 50, 101, and 152 layer model are defined for ResNet. Got {}".format(layers))
        super().__init__()
        if layers != 18:
            raise NotImplementedError("the MI355X hot path covers the resnet18 variants (layers=18)")
        assert in_channels > 3
        self.in_channels = in_channels
        self._norm_layer = nn.BatchNorm2d
        self.output_size = output_size

        self.conv1 = _conv(3, 64, 7, 2)
        self.bn1 = nn.BatchNorm2d(64)
        weights_init(self.conv1)
        weights_init(self.bn1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = _make_layer(64, 64, 2, 1, weights_init_kaiming)
        self.layer2 = _make_layer(64, 128, 2, 2, weights_init_kaiming)
        self.layer3 = _make_layer(128, 256, 2, 2, weights_init_kaiming)
        self.layer4 = _make_layer(256, 512, 2, 2, weights_init_kaiming)
        if pretrained:
            self._load_imagenet_encoder()

        self.conv1_depth = _conv(self._depth_inputs(), 16, 7, 2)
        self.bn1_depth = nn.BatchNorm2d(16)
        weights_init_kaiming_leaky(self.conv1)   # sic: the reference re-initialises the RGB stem here (models.py:561-562)
        weights_init_kaiming(self.bn1)
        self.relu_depth = nn.LeakyReLU(0.2, inplace=True)
        self.maxpool_depth = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1_depth = _make_layer(16, 16, 2, 1, weights_init_kaiming)
        self.layer2_depth = _make_layer(16, 32, 2, 2, weights_init_kaiming)
        self.layer3_depth = _make_layer(32, 64, 2, 2, weights_init_kaiming)
        self.layer4_depth = _make_layer(64, 128, 2, 2, weights_init_kaiming)

        self.conv_fusion = _conv(512 + 128, 512, 1, pad=0)
        self.bn_fusion = nn.BatchNorm2d(512)
        self.conv2 = _conv(512, 256, 1, pad=0)
        self.bn2 = nn.BatchNorm2d(256)
        self.decoder = choose_decoder(decoder, 256)
        self.conv3 = _conv(16, 1, 3)
        self.bilinear = nn.Upsample(size=self.output_size, mode="bilinear", align_corners=True)

        self.conv2.apply(weights_init)
        self.bn2.apply(weights_init)
        self.decoder.apply(weights_init)
        self.conv3.apply(weights_init)
        self.__dict__["_plans"] = {}
        self._adopt_arena_children()

    # the reference takes the encoder from torchvision.models.resnet18(pretrained=True) (models.py:526,546-551); there is
    # no torchvision / network here, so ImageNet weights come from a local torchvision-format state_dict instead.
    def _load_imagenet_encoder(self):
        path = os.environ.get("RADAR_DEPTH_RESNET18_WEIGHTS", "")
        if not os.path.exists(path):
            raise RuntimeError("pretrained=True needs ImageNet ResNet-18 weights: set RADAR_DEPTH_RESNET18_WEIGHTS to a "
                               "torchvision resnet18 state_dict file, or construct with pretrained=False (--no-pretrain)")
        sd = torch.load(path, map_location="cpu", weights_only=True)      # a plain torchvision state_dict: tensors only
        own = self.state_dict()
        picked = {k: v for k, v in sd.items() if k.split(".")[0] in ("layer1", "layer2", "layer3", "layer4") and k in own}
        self.load_state_dict(picked, strict=False)

    def _depth_inputs(self):
        return 1

    # ------------------------------------------------------------------ HIP execution
    def _plan(self, batch, height, width, train, depth_planes=None, bf16=False, storage="fp32", segment_joins=True, autotune=None, split=False):
        from ..engine import LateFusionPlan
        st = self._ensure_arenas()
        key = (batch, height, width, bool(train), st["version"], None if depth_planes is None else tuple(t.data_ptr() for t in depth_planes),
               bool(bf16), storage, bool(segment_joins), bool(split))
        plans = self.__dict__.setdefault("_plans", {})
        if key not in plans:
            _evict_plans(plans, st["version"])
            plans[key] = LateFusionPlan(self, batch, height, width, train=train, depth_planes=depth_planes, bf16=bf16, storage=storage,
                                        segment_joins=segment_joins, autotune=autotune, split=split)
        else:
            plans[key] = plans.pop(key)          # most recently used last
        return plans[key]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("radar_depth_amd modules run on MI355X only (HIP kernels); got a %s tensor" % x.device.type)
        assert x.dim() == 4 and x.shape[1] >= 4
        x = x.contiguous().float()
        # (eval plans fold BatchNorm into the packed fp32 weights and run the fp32 kernels; training plans follow eager_operands())
        plan = self._plan(x.shape[0], x.shape[2], x.shape[3], self.training, split=self.training and eager_operands(self) == "split")
        if self.training and torch.is_grad_enabled():
            return _PlanFunction.apply(plan, x, *self._arena_root()._ensure_arenas()["params"])
        return plan.run_forward(x).clone()


class _PlanFunction(torch.autograd.Function):
    """Bridges a LateFusionPlan into torch.autograd: backward returns the plan's gradient-arena views."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.plan = hold_plan(plan, ctx)          # (held until this node is collected: the cache never closes a plan under it)
        ctx.n_params = len(params)
        ctx.params = params
        out = plan.run_forward(x).clone()
        ctx.generation = plan.generation
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("this plan ran another forward since the one being differentiated: its saved activations are "
                               "static buffers, so call backward() before the next training-mode forward of the same "
                               "(batch, size) -- e.g. accumulate gradients one batch at a time")
        plan.run_backward(gout.contiguous())
        own = {id(p) for p in plan.m.parameters()}
        # (the arena state is looked up ONCE: _grad_view re-validates the arenas -- a walk over every parameter -- on each call, which made
        #  this return statement 163 module-tree walks, ~10 ms of host time per backward at the headline geometry: bench.py alt_eager)
        gv = plan.m._arena_root()._ensure_arenas()["gviews"]
        grads = [gv[id(p)] if id(p) in own else None for p in ctx.params]
        return (None, None) + tuple(grads)
