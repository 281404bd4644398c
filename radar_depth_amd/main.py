"""Counterpart of the hot-path parts of the reference's main.py:

  create_model  <- main.py:118-186  (--arch / --decoder / --modality plugin dispatch; unknown archs raise the same
                   ValueError; archs outside the MI355X hot path raise NotImplementedError naming the scope)
  HipTrainStep  <- the training-step body main.py:416-445 (forward, MaskedL1 / uncertainty-weighted loss,
                   zero_grad, backward, SGD step) fused on the device: no host synchronisation inside the step,
                   optional hipGraph replay, and -- when torch.distributed is initialised -- data-parallel
                   gradient averaging with bucketed RCCL all-reduces overlapped with the rest of backward.
"""
import ctypes as C
import os

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


def _param_offsets(root):
    """name -> (begin, end) element offsets in the flat arenas (16-byte aligned slots, named_parameters order)."""
    offs, off = {}, 0
    for name, p in root.named_parameters():
        offs[name] = (off, off + p.numel())
        off += (p.numel() + 3) // 4 * 4
    return offs


def bucket_segments(plan, param_offsets, prefix=""):
    """Gradient buckets of a LateFusionPlan: its backward is built as segments that end with all streams joined
    (engine.LateFusionPlan._build_backward).  Returns [(op_begin, op_end, [(lo, hi), ...])]: after ops [op_begin, op_end)
    the listed arena slices (parameters whose top-level name is in the segment's prefix list) are final."""
    out = []
    for begin, end, names in plan.bwd_segments:
        sl = sorted((v[0], (v[1] + 3) // 4 * 4) for k, v in param_offsets.items()
                    if k.startswith(prefix) and k[len(prefix):].split(".")[0] in names)
        merged = []
        for lo, hi in sl:
            if merged and lo == merged[-1][1]:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        out.append((begin, end, [tuple(m) for m in merged]))
    return out


def reduce_gradient_buckets(grads, buckets):
    """Sum-all-reduce the gradient buckets (each a list of arena slices [(lo, hi), ...]) of the flat gradient tensor
    across the default process group.  Generator form: yields before each bucket so the caller can enqueue the backward
    segment that completes it; all collectives are asynchronous and waited for at the end (RCCL over xGMI on the GPU,
    gloo in the CPU tests)."""
    works = []
    for slices in buckets:
        yield slices
        for lo, hi in slices:
            works.append(torch.distributed.all_reduce(grads[lo:hi], async_op=True))
    for wk in works:
        wk.wait()


class HipTrainStep:
    """One reference training step entirely on the device (no host synchronisation inside):

      resnet18_latefusion                    main.py:440-445   loss = MaskedL1(pred, target)
      resnet18_multistage                    main.py:431-438   loss = d1 + d2
      resnet18_multistage_uncertainty_fixs   main.py:416-429   loss = e^-w1 (d1 + 0.1 smooth) + e^-w2 d2 + w1 + w2

    followed by zero_grad / backward / SGD(momentum, weight decay) (main.py:443-445).  With torch.distributed initialised
    (one process per GPU, backend nccl = RCCL) gradients are averaged with bucketed all-reduces launched as soon as each
    bucket's last backward kernel has been enqueued, overlapping the rest of backward.

    use_graph: replay the step as hipGraphs from the second call on.  Off by default: the step keeps ~650 kernels on three
    streams in flight from a few ms of host time, and on this stack the graph executor's handling of the cross-stream edges is
    slower than the plain stream launches (661 vs 679 samples/s for latefusion b=16, 272 vs 288 for multistage b=8, and the
    data-parallel path loses 2.5 % as five graphs but nothing as plain launches).

    operands: "fp32" (default, the parity path) or "bf16": forward, input-gradient and stride-1 3x3 weight-gradient convolutions
    with bf16 operands on the bf16 matrix cores (csrc/gconv_bf16.hip, csrc/wgrad_bf16.hip; fp32 tensors, fp32 accumulation, fp32
    BatchNorm / loss / SGD and the remaining weight gradients) -- the
    torch.autocast(bfloat16) analogue for BASELINE.json configs 2/4; tolerances in tests/test_gpu_bf16.py."""

    def __init__(self, model, batch, height, width, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=None, use_graph=False,
                 operands="fp32", criterion="l1", comm="auto", storage="fp32", autotune=None):
        """criterion: "l1" (MaskedL1Loss, the default of utils.parse_command) or "l2" (MaskedMSELoss, `-c l2`, main.py:294-305).
        comm: "rccl" = the C ABI's own communicator (radar_depth_amd.comm, rd_allreduce_bucket on a dedicated communication
        stream, event-chained behind each backward segment); "torch" = torch.distributed.all_reduce (the cross-check);
        "auto" = rccl when radar_depth_amd.comm is initialised, else torch when torch.distributed is, else single-GPU."""
        from .model.multistage_model import ResNet_multistage
        assert criterion in ("l1", "l2"), criterion
        # storage="bf16": NHWC activations and their gradients live in HBM as bf16 (implies bf16 conv operands); BatchNorm
        # statistics, losses, parameters, their gradients and the optimizer stay fp32 -- BASELINE.json configs 3 / 5
        assert storage in ("fp32", "bf16"), storage
        if storage == "bf16":
            operands = "bf16"
        self.operands, self.storage = operands, storage
        self.model = model
        # The backward is built in bucket-aligned segments.  Segment-granular hipGraphs and torch.distributed's all_reduce (which
        # orders itself behind the CURRENT stream) need every segment to end with all plan streams joined; plain launches with the
        # native communicator -- and single-GPU runs -- do not: the communication stream waits for per-stream events instead and
        # the main chain keeps running ahead of the weight-gradient stream (+1 % fp32, +2 % bf16 storage).
        from . import comm as _comm0
        tdist0 = torch.distributed.is_available() and torch.distributed.is_initialized()
        native = comm == "rccl" or (comm == "auto" and (_comm0.initialised() or not tdist0))
        joins = bool(use_graph) or not native
        self.L = lib()
        self._f_sums, self._f_bwd = ((self.L.rd_masked_l1_sums, self.L.rd_masked_l1_bwd) if criterion == "l1" else
                                     (self.L.rd_masked_l2_sums, self.L.rd_masked_l2_bwd))
        model.train()
        self.multistage = isinstance(model, ResNet_multistage)
        if self.multistage:
            self.mp = model._plans(batch, height, width, True, bf16=operands == "bf16", storage=storage, segment_joins=joins, autotune=autotune)
            self.plans = [self.mp.p1, self.mp.p2]
        else:
            assert isinstance(model, ResNet_latefusion)
            self.mp = None
            self.plans = [model._plan(batch, height, width, True, bf16=operands == "bf16", storage=storage, segment_joins=joins,
                                      autotune=autotune)]
        self.plan = self.plans[0]
        self.st = model._ensure_arenas()
        self._arena_version = self.st["version"]
        dev = self.plan.dev
        self.batch, self.height, self.width = batch, height, width
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        from . import comm as _comm
        tdist = torch.distributed.is_available() and torch.distributed.is_initialized()
        assert comm in ("auto", "rccl", "torch"), comm
        if comm == "auto":
            comm = "rccl" if _comm.initialised() else "torch"
        if comm == "rccl" and not _comm.initialised():
            raise RuntimeError("comm='rccl' needs radar_depth_amd.comm.init_from_torch_distributed() / init_from_file() first")
        self.comm = comm
        self.world = _comm.world() if comm == "rccl" else (torch.distributed.get_world_size() if tdist else 1)
        self.comm_stream = torch.cuda.Stream(device=dev) if comm == "rccl" else None
        if self.world > 1:
            self.sync_state_from_rank0()
        # RD_FORCE_DP=1 runs the data-parallel code path (segmented graphs + bucketed all-reduce) even with one rank (tests)
        self.dp = self.world > 1 or (os.environ.get("RD_FORCE_DP") == "1" and (comm == "rccl" or tdist))
        p = self.plan
        self.n_out = batch * p.Ho * p.Wo
        self.target = torch.zeros(batch, 1, p.Ho, p.Wo, device=dev)
        tiles = self.L.rd_loss_tiles(C.c_int64(self.n_out))
        self.l1_ws = torch.zeros(2 * tiles, dtype=torch.float64, device=dev)
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        self.sums2 = torch.zeros(2, dtype=torch.float64, device=dev)
        self.loss = torch.zeros(1, device=dev)
        self.coef = torch.zeros(1, device=dev)
        if self.multistage:
            self.uncertainty = loss_weights is not None
            self.w_smooth = float(loss_weights["w_smooth"]) if self.uncertainty else 0.0
            self.loss4 = torch.zeros(4, device=dev)           # d1, d2, smooth, total
            self.coefs3 = torch.zeros(3, device=dev)          # e^-w1, w_smooth e^-w1, e^-w2
            self.smooth_out = torch.zeros(1, dtype=torch.float64, device=dev)
            nfl = int(self.L.rd_smooth_workspace_floats(batch, height, width))
            self.smooth_ws = torch.zeros((nfl + 1) // 2, dtype=torch.float64, device=dev)
            if self.uncertainty:
                self.w1, self.w2 = model.w_stage1, model.w_stage2
                self.dw1, self.dw2 = model._grad_view(self.w1), model._grad_view(self.w2)
            else:                                             # loss = d1 + d2: the same kernel with w1 = w2 = 0
                self.w1, self.w2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
                self.dw1, self.dw2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            self.loss = self.loss4[3:4]
        self.use_graph = use_graph
        self.graphs = None
        self.steps = 0
        # hipGraph capture is illegal on the legacy default stream: the step owns a stream and fences it against the caller's
        self.side = torch.cuda.Stream(device=dev)   # default priority: raising any stream's priority measured 17-29 % slower
        offs = _param_offsets(model)
        if self.multistage:
            # eight buckets in backward-completion order: the four segments of stage 2, then the four of stage 1; the scalar
            # w_stage1/2 (final right after the loss) ride with the last one
            self._seg2 = bucket_segments(self.mp.p2, offs, "stage2.")
            self._seg1 = bucket_segments(self.mp.p1, offs, "stage1.")
            tops = sorted((v[0], (v[1] + 3) // 4 * 4) for k, v in offs.items() if not k.startswith(("stage1.", "stage2.")))
            self._buckets = [sl for _, _, sl in self._seg2] + [sl for _, _, sl in self._seg1]
            self._buckets[-1] = list(self._buckets[-1]) + tops
            self._bucket_side_events = list(self.mp.p2.segment_events) + list(self.mp.p1.segment_events)
        else:
            self._segments = bucket_segments(self.plan, offs)
            self._buckets = [sl for _, _, sl in self._segments]
            self._bucket_side_events = list(self.plan.segment_events)

    def sync_state_from_rank0(self):
        """Data-parallel replicas must start from one state (DDP broadcasts at construction): parameters, momentum and the
        BatchNorm buffers of rank 0 replace every other rank's, so a checkpoint loaded on rank 0 only -- or different seeds --
        cannot silently train divergent replicas whose gradients are still averaged."""
        if self.comm == "rccl":
            from . import comm as _comm
            cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for tns in [self.st["arena"], self.st["mom"]] + list(self.model.buffers()):
                _comm.broadcast_(tns, cur, 0)
            return
        dist = torch.distributed
        dist.broadcast(self.st["arena"], 0)
        dist.broadcast(self.st["mom"], 0)
        for b in self.model.buffers():
            dist.broadcast(b, 0)

    # ---- optimizer state in torch.optim.SGD's layout (the reference checkpoints `optimizer.state_dict()`, main.py:358-374)
    def state_dict(self):
        params = self.st["params"]
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        state = {}
        if self.steps > 0:
            state = {i: {"momentum_buffer": self.st["mom"][o:o + p.numel()].view(p.shape).clone()}
                     for i, (p, o) in enumerate(zip(params, offs))}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.wd, "nesterov": False,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.set_lr(g["lr"])
        if g["momentum"] != self.momentum or g["weight_decay"] != self.wd:
            self.momentum, self.wd = g["momentum"], g["weight_decay"]
            self._drop_graphs()
        params = self.st["params"]
        off = 0
        loaded = 0
        for i, p in enumerate(params):
            ent = sd["state"].get(i)
            if ent is not None and ent.get("momentum_buffer") is not None:
                self.st["mom"][off:off + p.numel()].view(p.shape).copy_(ent["momentum_buffer"])
                loaded += 1
            off += (p.numel() + 3) // 4 * 4
        # torch.optim.SGD starts a missing buffer as buf = grad; rd_sgd_step's first step does the same on a zero buffer
        # (0.9 * 0 + g), so only a restored state switches the "first step" behaviour off
        if loaded:
            self.steps = max(self.steps, 1)

    def set_lr(self, lr):
        if lr != self.lr:
            self.lr = lr
            self._drop_graphs()

    def _drop_graphs(self):
        if self.graphs:
            for g in self.graphs:
                self.L.rd_graph_destroy(g)
        self.graphs = None

    # ---- pieces of one step: each is a sequence of C-ABI launches on the step's stream, separately graph-capturable;
    # ---- after piece i (i < len(buckets)) the gradient bucket i is final
    def _l1(self, pred, sums):
        check(self._f_sums(ptr(pred), ptr(self.target), C.c_int64(self.n_out), ptr(self.l1_ws), ptr(sums), self.plan.stream), "masked_sums")

    def _l1_bwd(self, pred, sums, coef, dpred, accumulate):
        check(self._f_bwd(ptr(pred), ptr(self.target), C.c_int64(self.n_out), ptr(sums), coef, ptr(dpred), accumulate,
                          self.plan.stream), "masked_bwd")

    def _latefusion_pieces(self):
        p = self.plan

        def head():
            p._run(p.prep)
            p._run(p.fwd)
            self._l1(p.pred, self.sums)
            check(self.L.rd_l1_total(ptr(self.sums), ptr(self.loss), ptr(self.coef), p.stream), "l1_total")
            self._l1_bwd(p.pred, self.sums, ptr(self.coef), p.dpred, 0)
        segs = self._segments
        pieces = [lambda: (head(), p._run(p.bwd[segs[0][0]:segs[0][1]]))]
        pieces += [(lambda k=k: p._run(p.bwd[segs[k][0]:segs[k][1]])) for k in range(1, len(segs))]
        return pieces

    def _multistage_pieces(self):
        mp, p1, p2, s = self.mp, self.mp.p1, self.mp.p2, self.plan.stream
        fptr = lambda t, i: C.c_void_p(t.data_ptr() + 4 * i)

        seg2, seg1 = self._seg2, self._seg1

        def stage2_head():
            p1._run(p1.prep)
            p1._run(p1.fwd)
            mp.filter_op()
            p2._run(p2.prep)
            p2._run(p2.fwd)
            self._l1(p1.pred, self.sums)
            self._l1(p2.pred, self.sums2)
            if self.uncertainty:
                check(self.L.rd_smooth_fwd(ptr(p1.pred), ptr(p1.x_in), p1.N, p1.x_in.shape[1], p1.H, p1.W, ptr(self.smooth_ws),
                                           ptr(self.smooth_out), s), "smooth_fwd")
            check(self.L.rd_uncertainty_total(ptr(self.sums), ptr(self.sums2), ptr(self.smooth_out), ptr(self.w1), ptr(self.w2),
                                              C.c_float(self.w_smooth), ptr(self.loss4), ptr(self.coefs3), ptr(self.dw1), ptr(self.dw2), s),
                  "uncertainty_total")
            self._l1_bwd(p2.pred, self.sums2, fptr(self.coefs3, 2), p2.dpred, 0)
            p2._run(p2.bwd[seg2[0][0]:seg2[0][1]])

        def stage1_head():
            # the last segment of stage 2 has written d(loss)/d(stage-1 prediction) into p1.dpred (multistage_model.py:75)
            self._l1_bwd(p1.pred, self.sums, fptr(self.coefs3, 0), p1.dpred, 1)
            if self.uncertainty:
                check(self.L.rd_smooth_bwd(p1.N, p1.H, p1.W, ptr(self.smooth_ws), fptr(self.coefs3, 1), ptr(p1.dpred), 1, s), "smooth_bwd")
            p1._run(p1.bwd[seg1[0][0]:seg1[0][1]])
        pieces = [stage2_head] + [(lambda k=k: p2._run(p2.bwd[seg2[k][0]:seg2[k][1]])) for k in range(1, len(seg2))]
        pieces += [stage1_head] + [(lambda k=k: p1._run(p1.bwd[seg1[k][0]:seg1[k][1]])) for k in range(1, len(seg1))]
        return pieces

    def _sgd(self):
        st = self.st
        check(self.L.rd_sgd_step(ptr(st["arena"]), ptr(st["grads"]), ptr(st["mom"]), C.c_int64(st["total"]), C.c_float(self.lr),
                                 C.c_float(self.momentum), C.c_float(self.wd), C.c_float(1.0 / self.world), 0, self.plan.stream), "sgd_step")

    def _pieces(self):
        parts = self._multistage_pieces() if self.multistage else self._latefusion_pieces()
        if not self.dp:                                       # single GPU: the whole step is one graph
            return [lambda: ([f() for f in parts], self._sgd())]
        return parts + [self._sgd]

    def step(self, inputs, target):
        """inputs [B,4,H,W], target [B,1,Ho,Wo] CUDA fp32.  Returns (loss[1], pred) device tensors (no sync)."""
        caller = torch.cuda.current_stream()
        self.side.wait_stream(caller)
        with torch.cuda.stream(self.side):
            self._step_on_side(inputs, target)
        caller.wait_stream(self.side)
        return self.loss, self.plans[-1].pred

    def _step_on_side(self, inputs, target):
        p = self.plan
        # the step is bound to one batch geometry (static buffers): never rely on copy_ broadcasting a ragged last batch
        want_in, want_t = (self.batch, p.x_in.shape[1], self.height, self.width), tuple(self.target.shape)
        if tuple(inputs.shape[:1]) + tuple(inputs.shape[2:]) != (want_in[0],) + want_in[2:] or inputs.shape[1] < want_in[1]:
            raise ValueError("HipTrainStep was built for inputs [%d,>=%d,%d,%d], got %s (build another HipTrainStep for a ragged "
                             "last batch or use drop_last=True)" % (want_in + (tuple(inputs.shape),)))
        if tuple(target.shape) != want_t:
            raise ValueError("HipTrainStep was built for target %s, got %s" % (want_t, tuple(target.shape)))
        st = self.st                    # cheap per-step check (no module traversal): first / last parameter still view the arena
        if (self.model._arena_root().__dict__.get("_arena_state") is not st or st["params"][0].data_ptr() != st["ptrs"][0]
                or st["params"][-1].data_ptr() != st["ptrs"][-1]):
            raise RuntimeError("the model's parameter arena was rebuilt after this HipTrainStep was created (model.to() / "
                               "reassigned parameter .data): build a new HipTrainStep")
        for pl in self.plans:
            pl.set_stream()
        p.x_in.copy_(inputs[:, :p.x_in.shape[1]])
        self.target.copy_(target)
        pieces = self._pieces()
        if self.use_graph and self.steps >= 1 and self.graphs is None:
            self.graphs = []
            for piece in pieces:
                check(self.L.rd_graph_begin(p.stream), "graph_begin")
                piece()
                g = C.c_void_p(0)
                check(self.L.rd_graph_end(p.stream, C.byref(g)), "graph_end")
                self.graphs.append(g)
        def launch(i):
            if self.graphs is not None:
                check(self.L.rd_graph_launch(self.graphs[i], p.stream), "graph_launch")
            else:
                pieces[i]()
        if not self.dp:
            launch(0)
        elif self.comm == "rccl":
            # piece i completes gradient bucket i: an event behind it on the step's stream releases the bucket's all-reduce on
            # the communication stream, which runs under pieces i+1..; the SGD kernel waits for the last bucket's event
            from . import comm as _comm
            cs = C.c_void_p(self.comm_stream.cuda_stream)
            grads = self.st["grads"]
            if not hasattr(self, "_bucket_events"):
                self._bucket_events = [torch.cuda.Event() for _ in range(len(self._buckets) + 1)]
            for i, slices in enumerate(self._buckets):
                launch(i)
                self._bucket_events[i].record(self.side)
                self.comm_stream.wait_event(self._bucket_events[i])
                for ev in self._bucket_side_events[i]:          # the segment's tails on the depth / weight-gradient streams
                    check(self.L.rd_stream_wait_event(cs, ev), "stream_wait_event")
                for lo, hi in slices:
                    _comm.allreduce_(grads, cs, lo, hi)
            self._bucket_events[-1].record(self.comm_stream)
            self.side.wait_event(self._bucket_events[-1])
            launch(len(self._buckets))
        else:
            # piece i completes gradient bucket i; its all-reduce is enqueued right behind it and overlaps pieces i+1..
            for i, _ in enumerate(reduce_gradient_buckets(self.st["grads"], self._buckets)):
                launch(i)
            launch(len(self._buckets))                        # SGD, after every bucket has been waited for
        self.steps += 1


class HipInference:
    """Eval-mode forward of the validate() body (main.py:564-595: model.eval(), no_grad, batch 1) as one hipGraph:
    BatchNorm folded into the convolutions (scale in the packed weights, shift/ReLU/residual in the conv epilogue)."""

    def __init__(self, model, batch, height, width, use_graph=True, operands="fp32", storage="fp32"):
        """operands: "fp32" (default; the parity path, within 1e-3 of the reference) or "bf16" (conv operands rounded to bf16,
        fp32 accumulation and tensors: csrc/gconv_bf16.hip; tolerance stated in tests/test_gpu_bf16.py)."""
        from .model.multistage_model import ResNet_multistage
        assert operands in ("fp32", "bf16") and storage in ("fp32", "bf16")
        bf16 = operands == "bf16" or storage == "bf16"
        self.L = lib()
        model.eval()
        self.multistage = isinstance(model, ResNet_multistage)
        if self.multistage:
            self.mp = model._plans(batch, height, width, False, bf16=bf16, storage=storage)
            self.plans = [self.mp.p1, self.mp.p2]
        else:
            self.mp = None
            self.plans = [model._plan(batch, height, width, False, bf16=bf16, storage=storage)]
        self.use_graph = use_graph
        self.graph = None
        self.calls = 0
        self.side = torch.cuda.Stream(device=self.plans[0].dev)

    def _run(self):
        p1 = self.plans[0]
        p1._run(p1.prep)
        p1._run(p1.fwd)
        if self.multistage:
            self.mp.filter_op()
            p2 = self.plans[1]
            p2._run(p2.prep)
            p2._run(p2.fwd)

    def __call__(self, x):
        """x [B,>=4,H,W] CUDA fp32 -> prediction(s) (plan-owned tensors, valid until the next call)."""
        caller = torch.cuda.current_stream()
        self.side.wait_stream(caller)
        with torch.cuda.stream(self.side):
            p = self.plans[0]
            for pl in self.plans:
                pl.set_stream()
            p.x_in.copy_(x[:, :p.x_in.shape[1]])
            if self.use_graph and self.calls >= 1 and self.graph is None:
                check(self.L.rd_graph_begin(p.streams[0]), "graph_begin")
                self._run()
                g = C.c_void_p(0)
                check(self.L.rd_graph_end(p.streams[0], C.byref(g)), "graph_end")
                self.graph = g
            if self.graph is not None:
                check(self.L.rd_graph_launch(self.graph, p.streams[0]), "graph_launch")
            else:
                self._run()
            self.calls += 1
        caller.wait_stream(self.side)
        if self.multistage:
            return {"stage1": self.mp.p1.pred, "stage2": self.mp.p2.pred, "mask": self.mp.mask, "radar_filtered": self.mp.kept}
        return self.plans[0].pred
