"""Counterpart of the hot-path parts of the reference's main.py:

  create_model  <- main.py:118-186  (--arch / --decoder / --modality plugin dispatch; unknown archs raise the same
                   ValueError; archs outside the MI355X hot path raise NotImplementedError naming the scope)
  HipTrainStep  <- the training-step body main.py:416-445 (forward, MaskedL1 / uncertainty-weighted loss,
                   zero_grad, backward, SGD step) fused on the device: no host synchronisation inside the step,
                   optional hipGraph replay, and -- when torch.distributed is initialised -- data-parallel
                   gradient averaging with bucketed RCCL all-reduces overlapped with the rest of backward.
"""
import ctypes as C

import torch
import torch.nn as nn

from ._lib import check, lib, ptr
from .model.models import ResNet_latefusion

HOT_PATH_ARCHS = ("resnet18_latefusion", "resnet18_multistage", "resnet18_multistage_uncertainty_fixs")


def create_model(args, output_size):
    print(f"[Info] Creating Model ({args.arch}-{args.decoder}) ...")
    in_channels = len(args.modality)
    if args.arch == "resnet18_latefusion":
        model = ResNet_latefusion(layers=18, decoder=args.decoder, output_size=output_size, in_channels=in_channels,
                                  pretrained=args.pretrained)
    elif args.arch == "resnet18_multistage":
        from .model.multistage_model import ResNet_multistage
        model = ResNet_multistage(layers=18, decoder=args.decoder, output_size=output_size, pretrained=args.pretrained)
    elif args.arch == "resnet18_multistage_uncertainty_fixs":
        from .model.multistage_model import ResNet_multistage
        model = ResNet_multistage(layers=18, decoder=args.decoder, output_size=output_size, pretrained=args.pretrained)
        w_stage1 = nn.Parameter(torch.tensor(1., dtype=torch.float32), requires_grad=True)
        w_stage2 = nn.Parameter(torch.tensor(1., dtype=torch.float32), requires_grad=True)
        model.register_parameter("w_stage1", w_stage1)
        model.register_parameter("w_stage2", w_stage2)
        return model, {"w_stage1": w_stage1, "w_stage2": w_stage2, "w_smooth": 0.1}
    elif args.arch in ("resnet50", "resnet18", "resnet34", "resnet18_new", "resnet18_multistage_uncertainty"):
        raise NotImplementedError("--arch %s is outside the MI355X hot path (%s)" % (args.arch, ", ".join(HOT_PATH_ARCHS)))
    else:
        raise ValueError("[Error] Unknown model!!")
    print("[Info] model created.")
    return model


# backward-op name prefixes whose gradients are complete once the last op carrying the prefix has run, in the order
# backward finishes them; each bucket is one contiguous slice of the flat gradient arena (named_parameters order)
_BUCKETS = (("conv_fusion", "bn_fusion", "conv2", "bn2", "decoder", "conv3"), ("layer4",), ("layer3",),
            ("conv1", "bn1", "layer1", "layer2"),
            ("conv1_depth", "bn1_depth", "layer1_depth", "layer2_depth", "layer3_depth", "layer4_depth"))


class HipTrainStep:
    """One reference training step (main.py:440-445 for resnet18_latefusion) entirely on the device."""

    def __init__(self, model, batch, height, width, lr=0.01, momentum=0.9, weight_decay=1e-4, use_graph=True):
        assert isinstance(model, ResNet_latefusion)
        self.model = model
        self.L = lib()
        model.train()
        self.plan = model._plan(batch, height, width, True)
        self.st = model._ensure_arenas()
        dev = self.plan.dev
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        n = batch * self.plan.Ho * self.plan.Wo
        self.n_out = n
        self.target = torch.zeros(batch, 1, self.plan.Ho, self.plan.Wo, device=dev)
        tiles = self.L.rd_loss_tiles(C.c_int64(n))
        self.l1_ws = torch.zeros(2 * tiles, dtype=torch.float64, device=dev)
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        self.loss = torch.zeros(1, device=dev)
        self.coef = torch.zeros(1, device=dev)
        self.use_graph = use_graph
        self.graphs = None
        # hipGraph capture is illegal on the legacy default stream: the step owns a stream and fences it against the caller's
        self.side = torch.cuda.Stream(device=dev)
        self.steps = 0
        self._segments = self._split_backward()

    # gradient buckets: (last backward-op index, arena slice)
    def _split_backward(self):
        names = [n for n, _ in self.model.named_parameters()]
        params = self.st["params"]
        offs, off = {}, 0
        for nme, p in zip(names, params):
            offs[nme] = (off, off + p.numel())
            off += (p.numel() + 3) // 4 * 4
        segs, start = [], 0
        ops = self.plan.bwd
        for prefixes in _BUCKETS:
            idx = [i for i, (nm, _, _) in enumerate(ops) if nm.split(".")[0] in prefixes]
            sl = [offs[nme] for nme in names if nme.split(".")[0] in prefixes]
            if not idx or not sl:
                continue
            lo, hi = min(s[0] for s in sl), max(s[1] for s in sl)
            segs.append((start, max(idx) + 1, lo, (hi + 3) // 4 * 4))
            start = max(idx) + 1
        if start < len(ops):
            s0, _, lo, hi = segs[-1]
            segs[-1] = (s0, len(ops), lo, hi)
        assert [s[0] for s in segs[1:]] == [s[1] for s in segs[:-1]], "backward ops are not ordered by bucket"
        return segs

    def set_lr(self, lr):
        if lr != self.lr:
            self.lr = lr
            self._drop_graphs()

    def _drop_graphs(self):
        if self.graphs:
            for g in self.graphs:
                self.L.rd_graph_destroy(g)
        self.graphs = None

    # ---- the pieces of one step, each a sequence of C-ABI launches on the current stream
    def _fwd_loss(self):
        p, s = self.plan, self.plan.stream
        p._run(p.prep)
        p._run(p.fwd)
        check(self.L.rd_masked_l1_sums(ptr(p.pred), ptr(self.target), C.c_int64(self.n_out), ptr(self.l1_ws), ptr(self.sums), s), "l1_sums")
        check(self.L.rd_l1_total(ptr(self.sums), ptr(self.loss), ptr(self.coef), s), "l1_total")
        check(self.L.rd_masked_l1_bwd(ptr(p.pred), ptr(self.target), C.c_int64(self.n_out), ptr(self.sums), ptr(self.coef), ptr(p.dpred), 0, s),
              "l1_bwd")

    def _bwd_segment(self, k):
        a, b, _, _ = self._segments[k]
        self.plan._run(self.plan.bwd[a:b])

    def _sgd(self):
        st = self.st
        check(self.L.rd_sgd_step(ptr(st["arena"]), ptr(st["grads"]), ptr(st["mom"]), C.c_int64(st["total"]), C.c_float(self.lr),
                                 C.c_float(self.momentum), C.c_float(self.wd), C.c_float(1.0 / self.world), 0, self.plan.stream), "sgd_step")

    def _pieces(self):
        """Graph-capturable pieces: with one GPU the whole step is one piece; with DP the all-reduces sit between them."""
        nseg = len(self._segments)
        if self.world == 1:
            return [lambda: (self._fwd_loss(), [self._bwd_segment(k) for k in range(nseg)], self._sgd())]
        pieces = [lambda: (self._fwd_loss(), self._bwd_segment(0))]
        pieces += [(lambda k=k: self._bwd_segment(k)) for k in range(1, nseg)]
        pieces.append(self._sgd)
        return pieces

    def step(self, inputs, target):
        """inputs [B,4,H,W], target [B,1,Ho,Wo] CUDA fp32.  Returns (loss[1], pred) device tensors (no sync)."""
        p = self.plan
        caller = torch.cuda.current_stream()
        self.side.wait_stream(caller)
        with torch.cuda.stream(self.side):
            self._step_on_side(inputs, target)
        caller.wait_stream(self.side)
        return self.loss, p.pred

    def _step_on_side(self, inputs, target):
        p = self.plan
        p.set_stream()
        p.x_in.copy_(inputs[:, :p.x_in.shape[1]])
        self.target.copy_(target)
        pieces = self._pieces()
        capture = self.use_graph and self.steps >= 1 and self.graphs is None
        if capture:
            self.graphs = []
            for piece in pieces:
                check(self.L.rd_graph_begin(p.stream), "graph_begin")
                piece()
                g = C.c_void_p(0)
                check(self.L.rd_graph_end(p.stream, C.byref(g)), "graph_end")
                self.graphs.append(g)
        works = []
        for i, piece in enumerate(pieces):
            if self.graphs is not None:
                check(self.L.rd_graph_launch(self.graphs[i], p.stream), "graph_launch")
            else:
                piece()
            if self.world > 1 and i < len(self._segments):
                _, _, lo, hi = self._segments[i]
                works.append(torch.distributed.all_reduce(self.st["grads"][lo:hi], async_op=True))
                if i == len(self._segments) - 1:
                    for w in works:
                        w.wait()
        self.steps += 1
