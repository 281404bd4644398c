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


def pack_buffers(bufs):
    """Every module buffer (BatchNorm running_mean / running_var fp32, num_batches_tracked int64 scalars) as ONE flat array of
    32-bit words, bit-preserving: what the data-parallel state broadcast ships instead of one collective per buffer."""
    return torch.cat([b.detach().contiguous().view(-1).view(torch.float32) for b in bufs]) if bufs else None


def unpack_buffers(bufs, flat):
    """Inverse of pack_buffers: writes the words back INTO the buffers (in place, whatever their dtype / rank)."""
    off = 0
    for b in bufs:
        n = b.numel() * b.element_size() // 4
        b.view(-1).view(torch.float32).copy_(flat[off:off + n])
        off += n
    assert flat is None or off == flat.numel()


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

    operands: None (default) = "split" for fp32 storage -- fp32 arithmetic on the bf16 matrix cores wherever the library has a plan
    for it, pinned against the CPU oracle at BASELINE.json's own batch sizes (tests/test_gpu_configs.py, both operand modes at the
    same bars); "fp32" = every convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32; the plan the eager autograd forward uses);
    "bf16": forward, input-gradient and stride-1 3x3 weight-gradient convolutions
    with bf16 operands on the bf16 matrix cores (csrc/gconv_bf16.hip, csrc/wgrad_bf16.hip; fp32 tensors, fp32 accumulation, fp32
    BatchNorm / loss / SGD and the remaining weight gradients) -- the
    torch.autocast(bfloat16) analogue for BASELINE.json configs 2/4; tolerances in tests/test_gpu_bf16.py.
    "split": fp32 arithmetic on the bf16 matrix cores -- the forward and input-gradient convolutions the library has a plan for
    (csrc/gconv_split.hip: >= 32 channels, 3x3 / 5x5 shapes) split every fp32 operand into three bf16 pieces and rebuild the
    product from six bf16 MFMAs with fp32 accumulation; results as close to fp64 as the fp32 MFMA path's (tests/test_gpu_gconv_split.py),
    fp32 tolerances unchanged; everything else is the fp32 path."""

    def __init__(self, model, batch, height, width, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=None, use_graph=False,
                 operands=None, criterion="l1", comm="auto", storage="fp32", autotune=None):
        """criterion: "l1" (MaskedL1Loss, the default of utils.parse_command) or "l2" (MaskedMSELoss, `-c l2`, main.py:294-305).
        comm: "rccl" = the C ABI's own communicator (radar_depth_amd.comm, rd_allreduce_bucket on a dedicated communication
        stream, event-chained behind each backward segment); "torch" = torch.distributed.all_reduce (the cross-check);
        "auto" = rccl when radar_depth_amd.comm is initialised, else torch when torch.distributed is, else single-GPU."""
        from .model.multistage_model import ResNet_multistage
        assert criterion in ("l1", "l2"), criterion
        # storage="bf16": NHWC activations and their gradients live in HBM as bf16 (implies bf16 conv operands); BatchNorm
        # statistics, losses, parameters, their gradients and the optimizer stay fp32 -- BASELINE.json configs 3 / 5
        assert storage in ("fp32", "bf16"), storage
        if operands is None:
            operands = "split"
        assert operands in ("fp32", "bf16", "split"), operands
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
            self.mp = model._plans(batch, height, width, True, bf16=operands == "bf16", storage=storage, segment_joins=joins, autotune=autotune,
                                   split=operands == "split")
            self.plans = [self.mp.p1, self.mp.p2]
        else:
            assert isinstance(model, ResNet_latefusion)
            self.mp = None
            self.plans = [model._plan(batch, height, width, True, bf16=operands == "bf16", storage=storage, segment_joins=joins,
                                      autotune=autotune, split=operands == "split")]
        self.plan = self.plans[0]
        from .model.models import ArenaOwner, hold_plan
        self._held = hold_plan(self.mp if self.multistage else self.plan)      # released by close(): the plan cache never closes it under this step
        self.st = model._ensure_arenas()
        self._arena_version = self.st["version"]
        self._rehomes = ArenaOwner.REHOMES[0]
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
        # RD_FORCE_DP=1 runs the data-parallel code path (state broadcast, segmented graphs, bucketed all-reduce) even with one rank (tests)
        self.dp = self.world > 1 or (os.environ.get("RD_FORCE_DP") == "1" and (comm == "rccl" or tdist))
        if self.dp:
            self.sync_state_from_rank0()
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
        self.steps = 0                 # optimizer steps (state_dict: a momentum state exists once > 0)
        self._warm = 0                 # plain launches issued by THIS object: hipGraph capture only after one warm, un-captured step
        self._table, self._ops, self._ranges, self._bucket_events, self._cs = None, None, None, None, None
        self._bind_sites, self._bound = None, None          # where the step's ops take the input / target pointer, and what they point at now
        self._bound_refs = None                             # the caller's tensors the plan currently points at (kept alive while bound)
        self._zero_copy = os.environ.get("RD_ZERO_COPY_INPUT", "1") == "1"
        self._issue_threads = max(1, int(os.environ.get("RD_ISSUE_THREADS", "1")))
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
        # three collectives in all (parameter arena, momentum arena, every BatchNorm buffer packed into one flat word array): the
        # ~160 per-buffer broadcasts this used to issue were 160 chances for a rank-order mismatch
        bufs = [b for b in self.model.buffers()]
        flat = pack_buffers(bufs)
        if self.comm == "rccl":
            from . import comm as _comm
            cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for tns in [self.st["arena"], self.st["mom"]] + ([flat] if flat is not None else []):
                _comm.broadcast_(tns, cur, 0)
        else:
            dist = torch.distributed
            for tns in [self.st["arena"], self.st["mom"]] + ([flat] if flat is not None else []):
                dist.broadcast(tns, 0)
        unpack_buffers(bufs, flat)

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
        self.st["mom"].zero_()          # parameters missing from sd["state"] start from an empty buffer, as in torch.optim.SGD
        for i, p in enumerate(params):
            ent = sd["state"].get(i)
            if ent is not None and ent.get("momentum_buffer") is not None:
                self.st["mom"][off:off + p.numel()].view(p.shape).copy_(ent["momentum_buffer"])
                loaded += 1
            off += (p.numel() + 3) // 4 * 4
        # torch.optim.SGD starts a missing buffer as buf = grad; rd_sgd_step's first step does the same on a zero buffer
        # (0.9 * 0 + g), so only a restored state switches the "first step" behaviour off
        if loaded:
            self.steps = max(self.steps, 1)     # (graph capture is gated by self._warm, not by this counter)

    def set_lr(self, lr):
        if lr != self.lr:
            self.lr = lr
            self._drop_graphs()

    def _drop_graphs(self):
        """Hyper-parameters are baked into the marshalled step (and its graphs): drop both, the next step rebuilds them."""
        for g in (self.graphs or {}).values():
            self.L.rd_graph_destroy(g)
        self.graphs = None
        tb, self._table = getattr(self, "_table", None), None
        if tb is not None:
            tb.close()
        if getattr(self, "_bound", None) is not None:       # the rebuilt op list takes the static buffers again
            for pl in self.plans:
                pl.bind_input(self.plan.x_in.data_ptr(), self.plan.x_in.shape[1])
        self._bind_sites, self._bound = None, None

    # ---- pieces of one step: each is a list of (name, C-ABI function, arguments) on the step's streams, separately
    # ---- graph-capturable; after piece i (i < len(buckets)) the gradient bucket i is final
    def _l1_ops(self, tag, pred, sums, s):
        return [(tag + ".sums", self._f_sums, (ptr(pred), ptr(self.target), C.c_int64(self.n_out), ptr(self.l1_ws), ptr(sums), s))]

    def _l1_bwd_ops(self, tag, pred, sums, coef, dpred, accumulate, s):
        return [(tag + ".bwd", self._f_bwd, (ptr(pred), ptr(self.target), C.c_int64(self.n_out), ptr(sums), coef, ptr(dpred), accumulate, s))]

    def _latefusion_pieces(self):
        p = self.plan
        s = p.streams[0]
        segs = self._segments
        head = list(p.prep) + list(p.fwd) + self._l1_ops("loss", p.pred, self.sums, s)
        head.append(("loss.total", self.L.rd_l1_total, (ptr(self.sums), ptr(self.loss), ptr(self.coef), s)))
        head += self._l1_bwd_ops("loss", p.pred, self.sums, ptr(self.coef), p.dpred, 0, s)
        pieces = [head + p.bwd[segs[0][0]:segs[0][1]]]
        pieces += [p.bwd[segs[k][0]:segs[k][1]] for k in range(1, len(segs))]
        return pieces

    def _multistage_pieces(self):
        mp, p1, p2 = self.mp, self.mp.p1, self.mp.p2
        s = p1.streams[0]
        fptr = lambda t, i: C.c_void_p(t.data_ptr() + 4 * i)
        seg2, seg1 = self._seg2, self._seg1
        head2 = list(p1.prep) + list(p1.fwd) + [mp.filter_args()] + list(p2.prep) + list(p2.fwd)
        head2 += self._l1_ops("loss1", p1.pred, self.sums, s) + self._l1_ops("loss2", p2.pred, self.sums2, s)
        if self.uncertainty:
            head2.append(("smooth.fwd", self.L.rd_smooth_fwd, (ptr(p1.pred), ptr(p1.x_in), p1.N, p1.x_in.shape[1], p1.H, p1.W,
                                                               ptr(self.smooth_ws), ptr(self.smooth_out), s)))
        head2.append(("uncertainty_total", self.L.rd_uncertainty_total,
                      (ptr(self.sums), ptr(self.sums2), ptr(self.smooth_out), ptr(self.w1), ptr(self.w2), C.c_float(self.w_smooth),
                       ptr(self.loss4), ptr(self.coefs3), ptr(self.dw1), ptr(self.dw2), s)))
        head2 += self._l1_bwd_ops("loss2", p2.pred, self.sums2, fptr(self.coefs3, 2), p2.dpred, 0, s)
        # the last segment of stage 2 writes d(loss)/d(stage-1 prediction) into p1.dpred (multistage_model.py:75)
        head1 = self._l1_bwd_ops("loss1", p1.pred, self.sums, fptr(self.coefs3, 0), p1.dpred, 1, s)
        if self.uncertainty:
            head1.append(("smooth.bwd", self.L.rd_smooth_bwd, (p1.N, p1.H, p1.W, ptr(self.smooth_ws), fptr(self.coefs3, 1), ptr(p1.dpred), 1, s)))
        pieces = [head2 + p2.bwd[seg2[0][0]:seg2[0][1]]] + [p2.bwd[seg2[k][0]:seg2[k][1]] for k in range(1, len(seg2))]
        pieces += [head1 + p1.bwd[seg1[0][0]:seg1[0][1]]] + [p1.bwd[seg1[k][0]:seg1[k][1]] for k in range(1, len(seg1))]
        return pieces

    def _sgd_ops(self):
        st = self.st
        return [("sgd_step", self.L.rd_sgd_step, (ptr(st["arena"]), ptr(st["grads"]), ptr(st["mom"]), C.c_int64(st["total"]), C.c_float(self.lr),
                                                   C.c_float(self.momentum), C.c_float(self.wd), C.c_float(1.0 / self.world), 0,
                                                   self.plan.streams[0]))]

    def _comm_ops(self, i):
        """Native exchange behind piece i: an event behind the piece on the step's stream (plus the segment's tails on the depth /
        weight-gradient streams) releases bucket i's in-place sums on the communication stream; they run under pieces i+1.."""
        s, cs = self.plan.streams[0], self._cs
        ops = [("bucket%d.record" % i, self.L.rd_event_record, (self._bucket_events[i], s)),
               ("bucket%d.wait" % i, self.L.rd_stream_wait_event, (cs, self._bucket_events[i]))]
        ops += [("bucket%d.wait_side" % i, self.L.rd_stream_wait_event, (cs, ev)) for ev in self._bucket_side_events[i]]
        g = self.st["grads"]
        ops += [("bucket%d.allreduce" % i, self.L.rd_allreduce_bucket, (C.c_void_p(g.data_ptr() + 4 * lo), C.c_int64(hi - lo), 0, cs))
                for lo, hi in self._buckets[i]]
        return ops

    def _build_table(self):
        """Marshal the whole step once (optable.py).  self._ranges: [(kind, begin, end)], kind "piece" (graph-capturable compute),
        "comm" (native exchange between pieces) -- issued in order; the torch.distributed cross-check path interleaves its
        all_reduce calls from Python after every "piece" but the last (SGD)."""
        from .optable import OpTable
        pieces = self._multistage_pieces() if self.multistage else self._latefusion_pieces()
        streams = [q for pl in self.plans for q in pl.streams]
        ops, ranges = [], []

        def add(kind, lst):
            ranges.append((kind, len(ops), len(ops) + len(lst)))
            ops.extend(lst)
        if not self.dp:                                       # single GPU: the whole step is one piece (one graph)
            add("piece", [op for pc in pieces for op in pc] + self._sgd_ops())
        elif self.comm == "rccl":
            if self._bucket_events is None:
                self._bucket_events = []
                for _ in range(len(self._buckets) + 1):
                    ev = C.c_void_p(0)
                    check(self.L.rd_event_create(C.byref(ev)), "rd_event_create")
                    self._bucket_events.append(ev)
            self._cs = C.c_void_p(self.comm_stream.cuda_stream)
            streams.append(self._cs)
            for i, pc in enumerate(pieces):
                add("piece", pc)
                add("comm", self._comm_ops(i))
            s = self.plan.streams[0]
            add("comm", [("exchange.done", self.L.rd_event_record, (self._bucket_events[-1], self._cs)),
                         ("exchange.join", self.L.rd_stream_wait_event, (s, self._bucket_events[-1]))])
            add("piece", self._sgd_ops())
        else:
            for pc in pieces:
                add("piece", pc)
            add("piece", self._sgd_ops())
        self._ops, self._ranges = ops, ranges
        self._table = OpTable(self.L, ops, streams)

    def _bind_batch(self, inputs, target):
        """The step reads the caller's batch in place when it can (the reference hands its batch tensors straight to the model too,
        main.py:416): contiguous fp32 tensors of exactly the plan's shapes -- the stems' plane tables and the few ops that take the input /
        target pointer are re-pointed (no launch; two 92 + 23 MB copies at the head of the step, where nothing overlaps them, otherwise).
        Anything else, hipGraph replay (addresses are baked into the graph) and RD_ZERO_COPY_INPUT=0 copy into the plan's static buffers.
        The tensors are only read; step() fences the caller's stream behind the step's, so the caller may reuse them as usual."""
        p = self.plan
        if self._bind_sites is None:
            xin, tgt = p.x_in.data_ptr(), self.target.data_ptr()
            self._bind_sites = [(i, k, a, 0 if a.value == xin else 1) for i, (_, _, args) in enumerate(self._ops) for k, a in enumerate(args)
                                if isinstance(a, C.c_void_p) and a.value in (xin, tgt)]
            # bind sites are found by pointer VALUE: an op that took the input at an offset (x_in + k) would keep reading the plan's static
            # buffer after a re-binding, silently.  No op does -- the stems read x_in through their plane tables -- and this keeps it so.
            spans = ((xin, p.x_in.numel() * p.x_in.element_size()), (tgt, self.target.numel() * self.target.element_size()))
            inside = [(name, k) for name, _, args in self._ops for k, a in enumerate(args)
                      if isinstance(a, C.c_void_p) and a.value and any(b < a.value < b + n for b, n in spans)]
            if inside:
                raise RuntimeError("ops take the input / target buffer at an interior offset, which zero-copy binding cannot re-point: %s" % inside[:4])
            self._bound = [xin, tgt]
        zc = (self._zero_copy and not self.use_graph and inputs.dtype == torch.float32 and target.dtype == torch.float32
              and inputs.is_contiguous() and target.is_contiguous() and inputs.shape[1] == p.x_in.shape[1]
              and inputs.device == p.x_in.device and target.device == p.x_in.device and inputs.data_ptr() % 16 == 0 and target.data_ptr() % 16 == 0)
        if zc:
            want = [inputs.data_ptr(), target.data_ptr()]
            inputs.record_stream(self.side)
            target.record_stream(self.side)
            # the plan's stem plane tables and the step's op arguments keep pointing at these tensors after step() returns (the cached plan
            # is shared with the eager forward and other steps, which re-bind only when THEY run): hold them until the next binding replaces
            # them, so that nothing of the plan ever points at freed memory
            self._bound_refs = (inputs, target)
        else:
            self._bound_refs = None
            p.x_in.copy_(inputs[:, :p.x_in.shape[1]])
            self.target.copy_(target)
            want = [p.x_in.data_ptr(), self.target.data_ptr()]
        if want != self._bound:
            for i, k, a, which in self._bind_sites:
                a.value = want[which]                                             # (the Python-loop diagnostics path reads the argument objects)
                check(self.L.rd_optable_set_word(self._table.h, i, k, C.c_uint64(want[which])), "rd_optable_set_word")
            self._bound = want
        for pl in self.plans:                                   # (plan state, shared with the eager forward of a cached plan: checked every step)
            if pl._x_bound != want[0]:
                pl.bind_input(want[0], p.x_in.shape[1])

    def synchronize_comm(self):
        """Block the host until this step's own communicator (comm="rccl": a second RCCL communicator on a private stream next to
        torch.distributed's) has drained.  Call it -- or torch.cuda.synchronize() -- BEFORE issuing a torch.distributed collective
        (metric averaging, barrier) after step(): collectives of two communicators enqueued without a device sync in between can
        interleave in a different order on different ranks, which deadlocks RCCL/NCCL."""
        if self.comm_stream is not None:
            self.comm_stream.synchronize()
        self.side.synchronize()

    def close(self):
        self._drop_graphs()
        tb, self._table = getattr(self, "_table", None), None
        if tb is not None:
            tb.close()
        evs, self._bucket_events = getattr(self, "_bucket_events", None), None
        for ev in evs or []:
            if ev.value:
                self.L.rd_event_destroy(ev)
        held, self._held = getattr(self, "_held", None), None
        if held is not None:
            from .model.models import release_plan
            release_plan(held)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, inputs, target):
        """inputs [B,4,H,W], target [B,1,Ho,Wo] CUDA fp32.  Returns (loss[1], pred) device tensors (no sync).
        With comm="rccl" see synchronize_comm() before mixing in torch.distributed collectives."""
        caller = torch.cuda.current_stream()
        self.side.wait_stream(caller)
        with torch.cuda.stream(self.side):
            self._step_on_side(inputs, target)
        caller.wait_stream(self.side)
        return self.loss, self.plans[-1].pred

    def _issue(self, begin, end, capturing=False):
        if self.plan._diagnostic_loop():                       # RD_POISON_LDS / RD_TRACE_OPS: host hook between ops
            return self.plan._run(self._ops[begin:end])
        # RD_ISSUE_THREADS=n: the step's ops are issued from n host threads, one per stream (rd_optable_run_mt)
        self._table.run(begin, end, lanes=1 if capturing else self._issue_threads)

    def _step_on_side(self, inputs, target):
        p = self.plan
        # the step is bound to one batch geometry (static buffers): never rely on copy_ broadcasting a ragged last batch
        want_in, want_t = (self.batch, p.x_in.shape[1], self.height, self.width), tuple(self.target.shape)
        if tuple(inputs.shape[:1]) + tuple(inputs.shape[2:]) != (want_in[0],) + want_in[2:] or inputs.shape[1] < want_in[1]:
            raise ValueError("HipTrainStep was built for inputs [%d,>=%d,%d,%d], got %s (build another HipTrainStep for a ragged "
                             "last batch or use drop_last=True)" % (want_in + (tuple(inputs.shape),)))
        if tuple(target.shape) != want_t:
            raise ValueError("HipTrainStep was built for target %s, got %s" % (want_t, tuple(target.shape)))
        st = self.st                    # cheap per-step check (no module traversal): first / last parameter still view the arena;
        from .model.models import ArenaOwner      # EVERY parameter whenever any module anywhere (re)built an arena since the last step
        moved = st["params"][0].data_ptr() != st["ptrs"][0] or st["params"][-1].data_ptr() != st["ptrs"][-1]
        if not moved and self._rehomes != ArenaOwner.REHOMES[0]:
            moved = any(q.data_ptr() != a for q, a in zip(st["params"], st["ptrs"]))
            self._rehomes = ArenaOwner.REHOMES[0]
        if self.model._arena_root().__dict__.get("_arena_state") is not st or moved:
            raise RuntimeError("the model's parameter arena was rebuilt after this HipTrainStep was created (model.to() / "
                               "reassigned parameter .data): build a new HipTrainStep")
        for pl in self.plans:
            pl.set_stream()
            pl.generation += 1          # the step overwrites the plan's saved activations: autograd nodes of an earlier eager forward go stale
        if self._table is None:
            self._build_table()
        self._bind_batch(inputs, target)
        ranges = self._ranges
        if self.use_graph and self._warm >= 1 and self.graphs is None:
            self.graphs = {}
            for k, (kind, b, e) in enumerate(ranges):
                if kind != "piece":
                    continue
                check(self.L.rd_graph_begin(p.streams[0]), "graph_begin")
                self._issue(b, e, capturing=True)
                g = C.c_void_p(0)
                check(self.L.rd_graph_end(p.streams[0], C.byref(g)), "graph_end")
                self.graphs[k] = g
        if self.graphs is None and (not self.dp or self.comm == "rccl"):
            self._issue(0, len(self._ops))                     # ONE C-ABI call: every launch, event and collective of the step
        else:
            works, n_piece = [], sum(1 for kind, _, _ in ranges if kind == "piece")
            seen = 0
            for k, (kind, b, e) in enumerate(ranges):
                if kind == "piece" and self.dp and self.comm == "torch" and seen == n_piece - 1:
                    for wk in works:                           # SGD, after every bucket has been waited for
                        wk.wait()
                if kind == "piece" and self.graphs is not None:
                    check(self.L.rd_graph_launch(self.graphs[k], p.streams[0]), "graph_launch")
                else:
                    self._issue(b, e)
                if kind == "piece":
                    if self.dp and self.comm == "torch" and seen < len(self._buckets):
                        # piece i completes gradient bucket i; its all-reduce is enqueued right behind it and overlaps pieces i+1..
                        works += [torch.distributed.all_reduce(self.st["grads"][lo:hi], async_op=True) for lo, hi in self._buckets[seen]]
                    seen += 1
        self.steps += 1
        self._warm += 1


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
        from .model.models import hold_plan
        self._held = hold_plan(self.mp if self.multistage else self.plans[0], self)     # (released when this object is collected)
        self.use_graph = use_graph
        self.graph = None
        self.calls = 0
        self._table = self._ops = None
        self.side = torch.cuda.Stream(device=self.plans[0].dev)

    def _run(self):
        p1 = self.plans[0]
        if self._table is None:
            from .optable import OpTable
            self._ops = list(p1.prep) + list(p1.fwd)
            if self.multistage:
                p2 = self.plans[1]
                self._ops += [self.mp.filter_args()] + list(p2.prep) + list(p2.fwd)
            self._table = OpTable(self.L, self._ops, [q for pl in self.plans for q in pl.streams])
        if p1._diagnostic_loop():
            return p1._run(self._ops)
        self._table.run()

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
