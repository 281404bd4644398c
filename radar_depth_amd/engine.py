"""Static execution plan of the late-fusion network on the HIP C ABI.

A LateFusionPlan is built once per (module, batch, input size, train/eval): it allocates every activation /
gradient / workspace buffer up front (addresses never change, so a whole step can be captured in a hipGraph)
and records the forward and backward passes as flat lists of C-ABI calls.  Running a pass is a loop over
prebuilt ctypes argument tuples; nothing is allocated and nothing synchronises.

Layout in HBM: activations NHWC fp32 with a channel stride, so the RGB and depth encoders write straight into
one 640-channel buffer (the torch.cat of model/models.py:652 costs nothing) and an UpProj module's two 5x5
convolutions write one [N,2H,2W,C] buffer (upper | bottom halves).  Parameters stay OIHW torch tensors (the
reference's state_dict contract); packed copies for the kernels are refreshed by rd_pack_weights each forward.

What is saved for backward: every conv input, every raw conv output, BN (mean, invstd), the activated
outputs (ReLU masks are re-derived from them) and the max-pool argmax bytes.

Forward structure follows model/models.py:627-664 (ResNet_latefusion.forward); the training-mode BatchNorm,
ReLU / LeakyReLU, residual joins and max-pool follow models.py:96-112,203-208,633-650.
"""
import contextlib
import ctypes as C
import os

import torch

from . import convdesc as cd
from ._lib import ACT_LEAKY02, ACT_NONE, ACT_RELU, check, lib

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Act:
    """A channel slice [c0, c0+C) of an NHWC buffer t[N,H,W,ld] (fp32, or bf16 in bf16-storage plans)."""
    __slots__ = ("t", "c0", "C")

    def __init__(self, t, c0=0, C_=None):
        self.t, self.c0 = t, c0
        self.C = t.shape[3] - c0 if C_ is None else C_

    @property
    def N(self):
        return self.t.shape[0]

    @property
    def H(self):
        return self.t.shape[1]

    @property
    def W(self):
        return self.t.shape[2]

    @property
    def ld(self):
        return self.t.shape[3]

    @property
    def M(self):
        return self.t.shape[0] * self.t.shape[1] * self.t.shape[2]

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr() + self.t.element_size() * self.c0)

    def chan(self, c0, C_):
        return Act(self.t, self.c0 + c0, C_)

    def view(self):
        return self.t[..., self.c0:self.c0 + self.C]


def _strided_zero_check(t, dd):
    """-> callable: True while the pixels of the NHWC buffer t that the strided input-gradient descriptor dd never writes are still zero."""
    def chk():
        m = torch.zeros(t.shape[1], t.shape[2], dtype=torch.bool, device=t.device)
        for i in range(dd.n_phases):
            ph = dd.phase[i]
            m[ph.out_off_h::dd.out_stride, ph.out_off_w::dd.out_stride][:ph.lh, :ph.lw] = True
        return bool((t[:, ~m] == 0).all().item())
    return chk


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


SPLIT_BNB_DEFAULT = "0"


class LateFusionPlan:
    def __init__(self, module, batch, height, width, train=True, depth_planes=None, x_source=None, dense_grad_dst=None,
                 dry_run=False, bf16=False, storage="fp32", segment_joins=True, autotune=None, split=False):
        """module: a radar_depth_amd ResNet_latefusion(2); bf16: run the gconv-lowered convolutions with bf16 operands on
        v_mfma_f32_32x32x16_bf16 (fp32 tensors, fp32 accumulation -- BASELINE.json configs 3/5, opt-in; in train plans the
        forward and input-gradient convolutions and the weight gradients of the >= 32-channel layers -- wgrad_bf16.hip; the
        16-channel layers, stems and head keep their fp32 weight-gradient kernels); depth_planes: None (depth stem reads channel(s) 3.. of the
        network input) or, for stage 2 of the multistage net, a list of stand-alone [N,H,W] maps; x_source: share another
        plan's static input buffer (stage 2 reads the RGB planes of stage 1's); dense_grad_dst: [N,H,W]-sized buffer that
        receives the gradient w.r.t. the second depth plane (stage-1 prediction, multistage_model.py:75)."""
        self.m = module
        self.N, self.H, self.W = batch, height, width
        self.train = train
        # segment_joins=False: the backward's bucket boundaries do not join the side streams into the main one (the main chain
        # keeps running ahead of the weight-gradient stream); each boundary instead records one event per side stream in
        # self.segment_events, which a data-parallel caller makes its communication stream wait for.  Only the end of backward joins.
        # autotune: time the candidate execution plans of every fp32 gconv descriptor once and pin the fastest (autotune.py)
        from . import autotune as _at
        self.autotune = _at.enabled_by_default() if autotune is None else bool(autotune)
        self.segment_joins = bool(segment_joins)
        self.segment_events = []
        # storage: element type of the NHWC activation / gradient tensors in HBM.  "bf16" (BASELINE.json configs 3 / 5) halves the
        # bytes of every HBM-bound kernel; it implies bf16 conv operands.  Statistics, parameters, gradients of parameters stay fp32.
        assert storage in ("fp32", "bf16")
        self.storage = storage
        self.adt = torch.bfloat16 if storage == "bf16" else torch.float32
        self.dt = 1 if storage == "bf16" else 0            # RD_DTYPE_BF16 / RD_DTYPE_F32
        self.bf16 = bool(bf16) or storage == "bf16"
        # split: fp32 arithmetic on the bf16 matrix cores for the forward / input-gradient convolutions the library plans that way
        # (csrc/gconv_split.hip: three bf16 pieces per operand, six MFMAs per product, fp32 accumulation); the rest is the fp32 plan
        self.split = bool(split) and not self.bf16
        # pre: the operands of the split convolutions / weight gradients are split into their three bf16 pieces by the kernels that
        # PRODUCE them (BatchNorm / activation / pooling passes: HBM-bound, idle VALU) and reach the matrix kernels as piece planes
        # [piece][C/16][pixels][16] -- rd_gconv_split_pre / rd_wgrad_split_pre stage with global_load_lds only (RD_SPLIT_PRE=0: split
        # while staging, as in round 3)
        self.pre = self.split and storage == "fp32" and os.environ.get("RD_SPLIT_PRE", "1") == "1"
        self._pc = {}          # data_ptr of an NHWC buffer -> dict(t=buffer, pc=piece planes or None, plane=c_int64, prod=[(slot, c0, C)])
        self.dev = next(module.parameters()).device
        # dry_run: record the op lists against host buffers without ever launching (CPU tests of the host logic)
        assert dry_run or self.dev.type == "cuda", "the HIP path needs the module on a GPU"
        self.dry_run = dry_run
        self.L = lib()
        # three streams: 0 = main chain (caller's stream), 1 = depth encoder, 2 = weight-gradient chains.  self._s is the
        # stream object the op builders attach to the ops they create (see on()).
        self.streams = [C.c_void_p(0), C.c_void_p(0), C.c_void_p(0)]
        self._s = self.streams[0]
        self._side = None            # torch streams backing streams[1:], created on first run
        self.events = []
        self.multi_stream = os.environ.get("RD_SINGLE_STREAM") != "1"
        # diagnostics: RD_STREAM_MASK bit0 = depth encoder on stream 1, bit1 = weight-gradient chains on stream 2 (default 3)
        self.stream_mask = int(os.environ.get("RD_STREAM_MASK", "3"))
        self.fwd, self.bwd = [], []
        self.prep = []
        self.probes = []       # (name, stream index, torch timing event): RD_TAIL_EVENTS=1
        self.evalcoef_jobs = []   # (C, bn, scale ptr, shift ptr) of the folded BatchNorms of an inference plan
        self.pack_jobs = []    # (src, dst, O, I, T, ldc, off, rows_total, transpose, scale, quad): packed in ONE launch per forward
        self.wino_jobs = []    # (weight OIHW, packed operand, O, I, flip): G g G^T of every Winograd layer in ONE launch per forward
        self.taps = {}         # name -> Act of intermediate tensors (tests / debugging)
        self.meta = {}         # op name -> (kernel family, descriptor) for the conv launches (bench roofline accounting)
        self.keep = []         # keep ctypes descriptors and tensors alive
        self.depth_planes = depth_planes
        self.x_source = x_source
        self.dense_grad_dst = dense_grad_dst
        self.Ho, self.Wo = getattr(module, "output_size", (height, width))
        self.generation = 0    # bumped by every forward: autograd nodes of an earlier forward must not read this plan's buffers
        self._optable = None   # prep + fwd + bwd marshalled once for rd_optable_run (see run_list)
        # RD_WGRAD_REDUCE_BATCH=n: at most n slab reductions per rd_wgrad_reduce_batched launch pair (flushed earlier at every bucket
        # boundary); 0: one rd_wgrad_reduce per weight tensor right behind its rd_wgrad, as in round 2
        # Default: unbatched for fp32 storage -- measured at b=16 450x800 (profiles/r03_reduce_batch.txt): 759.6 samples/s unbatched,
        # 757.1 / 753.6 / 753.7 with batches of 4 / 8 / a whole segment: the per-tensor reductions run hidden beside the MFMA-bound
        # kernels of the other streams, a batch at the end of a segment is a serial tail
        self.reduce_batch_max = int(os.environ.get("RD_WGRAD_REDUCE_BATCH", "0"))
        self.batch_reduces = self.reduce_batch_max > 0
        self._pending_reduces, self.reduce_batches = {}, []
        self.persistent = []   # buffers whose contents are plan state laid down at build time (not per-step scratch): tools/poison_global.py skips them
        self.table_pins = 0    # descriptors whose plan came from the offline-tuned table
        # RD_FUSE_BN_BWD=0 (diagnostics): every BatchNorm backward runs its own reduce pass, as in round 2
        self.fuse_bn_bwd = os.environ.get("RD_FUSE_BN_BWD", "1") == "1"
        self.bnb_out = None
        self.persistent_zero_checks = []   # callables: the never-written pixels of the persistent buffers still hold zeros (tests)
        self._build()
        if self.persistent and not self.dry_run:
            # the build-time zeros were laid down on torch's current stream; the plan may first run on any stream: order them here
            torch.cuda.current_stream().synchronize()

    def close(self):
        """Destroy the plan's hipEvents (its buffers are torch tensors and go with the object).  Called when the LAST holder lets go
        (__del__): the plan cache only drops its reference, because a HipTrainStep / HipInference / autograd node may still hold the plan."""
        tb, self._optable = getattr(self, "_optable", None), None
        if tb is not None:
            tb.close()
        evs, self.events = getattr(self, "events", []), []
        if not getattr(self, "dry_run", True):
            for ev in evs:
                if ev.value:
                    self.L.rd_event_destroy(ev)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ small helpers
    def buf(self, *shape, dtype=torch.float32):
        t = torch.empty(shape, dtype=dtype, device=self.dev)
        self.keep.append(t)
        return t

    def act(self, N, H, W, Cc):
        return Act(self.buf(N, H, W, Cc, dtype=self.adt))

    def op(self, lst, name, fn, *args):
        lst.append((name, fn, args))

    # ------------------------------------------------------------------ pre-split piece planes of an activation / gradient tensor
    def _pc_entry(self, a):
        return self._pc.setdefault(a.t.data_ptr(), dict(t=a.t, pc=None, plane=C.c_int64(0), prod=[]))

    def _pc_bind(self, ent):
        base, t = ent["pc"].data_ptr(), ent["t"]
        m = t.shape[0] * t.shape[1] * t.shape[2]
        ent["plane"].value = t.shape[3] * m
        for slot, c0, _ in ent["prod"]:
            slot.value = base + (c0 // 16) * m * 32

    def pc_out(self, a):
        """For the kernel that PRODUCES Act a: (piece pointer, plane stride) arguments -- NULL until a consumer asks for the planes
        (pc_in), in which case the producer's op writes them next to the fp32 tensor."""
        if not self.pre or a.C % 16 or a.c0 % 16 or a.ld % 16 or a.t.dtype != torch.float32:
            return C.c_void_p(0), C.c_int64(0)
        ent = self._pc_entry(a)
        slot = C.c_void_p(0)
        ent["prod"].append((slot, a.c0, a.C))
        if ent["pc"] is not None:
            self._pc_bind(ent)
        return slot, ent["plane"]

    def pc_in(self, a, lst):
        """For a kernel that CONSUMES Act a as piece planes: (pointer to a's first 16-channel block, plane stride).  The planes are
        allocated on first request; channels no producer op covers are split by a stand-alone rd_split_pieces pass emitted here."""
        assert self.pre and a.C % 16 == 0 and a.c0 % 16 == 0 and a.ld % 16 == 0
        ent = self._pc_entry(a)
        if ent["pc"] is None:
            t = ent["t"]
            ent["pc"] = self.buf(3, t.shape[3] // 16, t.shape[0] * t.shape[1] * t.shape[2], 16, dtype=torch.bfloat16)
            self._pc_bind(ent)
        m = a.M
        ptr_ = C.c_void_p(ent["pc"].data_ptr() + (a.c0 // 16) * m * 32)
        covered = sorted((c0, c0 + cc) for _, c0, cc in ent["prod"])
        pos = a.c0
        for lo, hi in covered:
            if lo <= pos < hi:
                pos = hi
        key = "standalone_%d_%d" % (a.c0, a.C)
        here = self.streams.index(self._s)
        if pos < a.c0 + a.C and key not in ent:
            ent[key] = (here, id(lst))
            ent["prod"].append((C.c_void_p(0), a.c0, a.C))
            self.op(lst, "split_pieces@%x+%d" % (a.t.data_ptr(), a.c0), self.L.rd_split_pieces, a.ptr, a.ld, C.c_int64(m), a.C, ptr_, ent["plane"], self.stream)
        elif key in ent and ent[key][0] != here:
            # a later consumer of the same stand-alone planes on ANOTHER stream: order it behind the stream the split pass was issued on
            # (the shipped models have no such consumer; a new one must not race the pass)
            if ent[key][1] != id(lst):
                raise NotImplementedError("piece planes of %x are produced by a stand-alone pass in another op list than their consumer on stream %d"
                                          % (a.t.data_ptr(), here))
            self.edge(lst, "split_pieces_join@%x+%d" % (a.t.data_ptr(), a.c0), ent[key][0], here)
        return ptr_, ent["plane"]

    def _pre_ok(self, a):
        return self.pre and a.C % 16 == 0 and a.c0 % 16 == 0 and a.ld % 16 == 0 and a.t.dtype == torch.float32

    @property
    def stream(self):
        """Stream object attached to the ops being built (the C ABI reads its value at call time)."""
        return self._s

    @contextlib.contextmanager
    def on(self, k):
        prev = self._s
        if not self.multi_stream or not (self.stream_mask >> (k - 1)) & 1 if k else False:
            k = 0
        self._s = self.streams[k]
        try:
            yield
        finally:
            self._s = prev

    def _event(self):
        ev = C.c_void_p(0)
        if not self.dry_run:
            check(self.L.rd_event_create(C.byref(ev)), "rd_event_create")
        self.events.append(ev)
        return ev

    def probe(self, lst, name, k):
        """RD_TAIL_EVENTS=1 (tools/tail_probe.py): a timing event on stream k at this point of the op list -- where the chains of the three
        streams really are at a fork / join of an un-profiled step (under rocprofv3 the host falls behind and reorders the streams)."""
        if self.dry_run or os.environ.get("RD_TAIL_EVENTS") != "1":
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                       # (materialises the hipEvent_t)
        self.probes.append((name, k, ev))
        self.op(lst, "probe." + name, self.L.rd_event_record, C.c_void_p(ev.cuda_event), self.streams[k])

    def edge(self, lst, name, src, dst):
        """Make stream `dst` wait for everything enqueued so far on stream `src` (fork or join)."""
        for q in (1, 2):
            if not (self.stream_mask >> (q - 1)) & 1:
                src, dst = (0 if src == q else src), (0 if dst == q else dst)
        if not self.multi_stream or src == dst:
            return
        ev = self._event()
        self.op(lst, name + ".record", self.L.rd_event_record, ev, self.streams[src])
        self.op(lst, name + ".wait", self.L.rd_stream_wait_event, self.streams[dst], ev)

    def grad_of(self, param):
        """fp32 gradient buffer of a parameter (the module's flat gradient arena view)."""
        return self.m._grad_view(param)

    # ------------------------------------------------------------------ convolution (gconv family)
    def _tune(self, d):
        """fp32 gconv descriptors only (the bf16 kernels have their own planner); must run before the descriptor's statistics
        tiles / workspace are sized, because both depend on the plan."""
        if self.dry_run or self.bf16:
            return
        from . import autotune as _at
        if self.autotune:
            _at.tune_gconv(self.L, d, self.dev)
        else:
            # offline-tuned plan table (radar_depth_amd/tuned_plans.json): a deterministic lookup, identical on every rank
            self.table_pins += 1 if _at.pin_from_table(self.L, d) else 0
        # whatever was pinned (or the heuristic) is final from here on: the statistics tiles / workspace sized next depend on it, and a
        # later tuner or table lookup in this process must not swap it
        check(self.L.rd_gconv_tune_commit(C.byref(d), 1), "rd_gconv_tune_commit")

    def _gconv_ws(self, d, name):
        """Split-K workspace of a descriptor (None when the library's plan does not split)."""
        self._tune(d)
        n = self.L.rd_gconv_workspace_floats(C.byref(d))
        if n < 0:
            check(int(n), "rd_gconv_workspace_floats(%s)" % name)
        return self.buf(int(n)) if n > 0 else None

    def conv_fwd(self, name, x, weights, k, stride, pad, out=None, upproj=False, lst=None):
        """weights: list of (param OIHW, column offset).  Returns (raw Act, ctx)."""
        lst = self.fwd if lst is None else lst
        N, H, W = x.N, x.H, x.W
        cout = sum(w.shape[0] for w, _ in weights)
        cin = x.C
        if upproj:
            d = cd.upproj_fwd(N, H, W, cin, cout, ldi=x.ld)
        else:
            d = cd.conv_fwd(N, H, W, cin, cout, k, stride, pad, ldi=x.ld)
        if out is None:
            out = self.act(N, d.Ho, d.Wo, cout)
        d.ldo = out.ld
        S = k * k
        wdt = torch.bfloat16 if self.bf16 else torch.float32
        quad = 2 if self.bf16 else 1
        # split plans: each of the two operands (forward, input gradient) is packed as three bf16 piece planes when the library has a
        # split plan for that descriptor, and stays the fp32 operand of rd_gconv otherwise
        sp_f = sp_d = pre_f = pre_d = c16 = False
        # Winograd F(2x2,3x3) on the split pipeline (csrc/wino_split.hip) for the 3x3 / stride-1 layers where the kernel-level gate measured
        # it ahead of the direct split kernels (rd_wino_preferred: 512-channel layers, small maps with <= 128 channels); forward and input
        # gradient decided separately (the input gradient is the same kernel on the flipped operand, channels swapped)
        wino_f = wino_d = False
        if self.split and self.train and not upproj and k == 3 and stride == 1 and pad == 1 and len(weights) == 1 and weights[0][1] == 0:
            wino_f = self.L.rd_wino_preferred(H, W, cin, cout, x.ld, out.ld if out is not None else cout) == 1
            wino_d = self.L.rd_wino_preferred(H, W, cout, cin, cout, cin) == 1
        if self.split:
            sp_f = self.L.rd_gconv_split_supported(C.byref(d)) == 1
            dd0 = (cd.upproj_dgrad(N, H, W, cin, cout) if upproj else cd.conv_dgrad(N, H, W, cin, cout, k, stride, pad)[0])
            sp_d = self.L.rd_gconv_split_supported(C.byref(dd0)) == 1
            # the pre-split form (operand split by its producer, gconv_sp2_kernel) where the library expects it to win even after the
            # producer's extra piece pass -- incl. 32-channel layers rd_gconv_split does not serve
            pre_f = (self._pre_ok(x) and self.L.rd_gconv_split_pre_supported(C.byref(d)) == 1
                     and self.L.rd_gconv_split_pre_preferred(C.byref(d)) == 1)
            sp_f = sp_f or pre_f
            # (the input gradient's operand is a [N,Ho,Wo,cout] tensor the BatchNorm backward produces: same alignment rule)
            pre_d = (self.pre and cout % 16 == 0 and self.L.rd_gconv_split_pre_supported(C.byref(dd0)) == 1
                     and self.L.rd_gconv_split_pre_preferred(C.byref(dd0)) == 1)
            sp_d = sp_d or pre_d
            if wino_f:
                sp_f = pre_f = False
            if wino_d:
                sp_d = pre_d = False
        self.L.rd_wino_packed_bytes.restype = C.c_int64
        uf = ud = None
        if wino_f:
            wp = None
            uf = self.buf(int(self.L.rd_wino_packed_bytes(cout, cin, 0)) // 2, dtype=torch.bfloat16)
            self.wino_jobs.append((weights[0][0], uf, cout, cin, 0))
        else:
            wp = self.buf(3, S, cin, cout, dtype=torch.bfloat16) if sp_f else self.buf(S, cin, cout, dtype=wdt)
        if wino_d:
            wd = None
            ud = self.buf(int(self.L.rd_wino_packed_bytes(cout, cin, 1)) // 2, dtype=torch.bfloat16)
            self.wino_jobs.append((weights[0][0], ud, cout, cin, 1))
        else:
            wd = self.buf(3, S, cout, cin, dtype=torch.bfloat16) if sp_d else self.buf(S, cout, cin, dtype=wdt)
        for w, off in weights:
            o, i, kh, kw = w.shape
            if not wino_f:
                self.pack_jobs.append((w, wp, o, i, kh * kw, cout, off, i, 0, None, 3 if sp_f else quad))
            if not wino_d:
                self.pack_jobs.append((w, wd, o, i, kh * kw, cin, off, cout, 1, None, 3 if sp_d else quad))
        if not sp_f and not wino_f:
            self._tune(d)
        if wino_f:
            tiles = self.L.rd_wino_stat_tiles(N, H, W)
        elif self.bf16 and not sp_f:
            tiles = self.L.rd_gconv_bf16_stat_tiles_t(self.dt, C.byref(d))      # (bf16 storage: the persistent kernel's own tiling)
        else:
            tiles = (self.L.rd_gconv_split_pre_stat_tiles if pre_f else self.L.rd_gconv_split_stat_tiles if sp_f else
                     self.L.rd_gconv_stat_tiles_ws)(C.byref(d))
        if tiles < 0:
            check(tiles, "rd_gconv_stat_tiles(%s)" % name)
        stat = self.buf(tiles, 2, cout) if self.train else None
        self.keep.append(d)
        if wino_f:
            if self.L.rd_wino_supported(H, W, cin, cout, x.ld, out.ld) != 1:
                raise RuntimeError("%s: planned as a Winograd layer, but the library does not serve %dx%d %d->%d with strides %d / %d" % (name, H, W, cin, cout, x.ld, out.ld))
            self.op(lst, name, self.L.rd_wino_conv3x3, x.ptr, N, H, W, cin, x.ld, _p(uf), out.ptr, cout, out.ld, C.c_void_p(0), 0, _p(stat), self.stream)
        elif pre_f:
            xp, xplane = self.pc_in(x, lst)
            self.op(lst, name, self.L.rd_gconv_split_pre, C.byref(d), xp, xplane, _p(wp), C.c_int64(S * cin * cout), out.ptr, C.c_void_p(0), 0, 0,
                    C.c_void_p(0), 0, _p(stat), self.stream)
        elif sp_f:
            self.op(lst, name, self.L.rd_gconv_split, C.byref(d), x.ptr, _p(wp), C.c_int64(S * cin * cout), out.ptr, C.c_void_p(0), 0, 0,
                    C.c_void_p(0), 0, _p(stat), self.stream)
        elif self.bf16:
            self.op(lst, name, self.L.rd_gconv_bf16_t, self.dt, C.byref(d), x.ptr, _p(wp), out.ptr, C.c_void_p(0), 0, 0, C.c_void_p(0), 0,
                    _p(stat), self.stream)
        elif self._c16_split(d):
            # split plans: the 16 -> 16 channel 3x3 layers (depth encoder layer1, dec4 conv2) with three-piece operands too
            # (csrc/conv16_split.hip: conv16.hip's contract, tiling and fp32 weight operand)
            c16 = True
            self.op(lst, name, self.L.rd_conv16_split, C.byref(d), x.ptr, _p(wp), out.ptr, C.c_void_p(0), 0, _p(stat), self.stream)
        else:
            ws = self._gconv_ws(d, name)
            self.op(lst, name, self.L.rd_gconv_ws, C.byref(d), x.ptr, _p(wp), out.ptr, C.c_void_p(0), 0, _p(stat), _p(ws), self.stream)
        self.taps[name] = out
        self.meta[name] = ("wino" if wino_f else "gconv_split_pre" if pre_f else "gconv_split" if sp_f else "gconv_bf16" if self.bf16 else "conv16_split" if c16 else "gconv", d)
        ctx = dict(name=name, d=d, x=x, out=out, weights=weights, wd=wd, k=k, stride=stride, pad=pad, upproj=upproj,
                   stat=stat, tiles=tiles, cin=cin, cout=cout, split_dgrad=sp_d, pre_dgrad=self.split and pre_d, wino_dgrad=wino_d, ud=ud)
        return out, ctx

    def _c16_split(self, d):
        return (self.split and not self.bf16 and os.environ.get("RD_CONV16_SPLIT", "1") == "1"
                and self.L.rd_conv16_split_supported(C.byref(d)) == 1)

    def conv_bwd(self, ctx, dout, need_dx=True, addend=None, dx=None, bnb=None):
        """Appends wgrad (+ reduce into the parameters' gradient views) and, optionally, dgrad.  Returns dx Act.
        bnb = dict(x=raw BatchNorm input Act, co=its coefficient dict, act=activation): this convolution's forward input was
        act(BatchNorm(x)); when the library can (fp32, unsplit plan) the dgrad launch also emits that BatchNorm's backward sums
        (rd_gconv_bnbwd) and self.bnb_out = (red, tiles) tells the caller to skip its reduce pass; else self.bnb_out = None."""
        self.bnb_out = None
        name, d, x = ctx["name"], ctx["d"], ctx["x"]
        N, H, W, cin, cout, k = x.N, x.H, x.W, ctx["cin"], ctx["cout"], ctx["k"]
        # wgrad uses the forward descriptor with dout's stride
        dwd = type(d)()
        C.memmove(C.byref(dwd), C.byref(d), C.sizeof(d))
        dwd.ldo = dout.ld
        self.keep.append(dwd)
        # bf16 plans: the weight gradients run on the bf16 matrix cores as well (3x3 / 1x1 at both strides, UpProj 5x5); what the
        # bf16 kernel cannot decompose keeps the fp32 kernel
        # (32-wide MFMA tiles: the 16-channel layers are faster on the fp32 kernel -- 42 vs 55 us for the depth encoder's layer1)
        wg_bf16 = (self.bf16 and os.environ.get("RD_WGRAD_BF16", "1") == "1" and min(cin, cout) >= int(os.environ.get("RD_WGRAD_BF16_MINC", "32"))
                   and self.L.rd_wgrad_bf16_supported(C.byref(dwd)) == 1)
        if self.storage == "bf16":
            # bf16 tensors: only the bf16 weight-gradient kernel reads them (the 16-channel layers included: a half-empty
            # 32-wide tile, but half the bytes and no conversion in its staging waves)
            wg_bf16 = True
            if self.L.rd_wgrad_bf16_supported(C.byref(dwd)) != 1:
                raise NotImplementedError("bf16 storage: no bf16 weight-gradient decomposition for %s" % name)
        # split plans: the 3x3 / stride-1 weight gradients with >= 64 channels run on the bf16 matrix cores as well (csrc/wgrad_split.hip:
        # both operands split into three bf16 pieces while they are staged); same slabs, same deterministic reduction
        wg_split = (self.split and not self.batch_reduces and os.environ.get("RD_WGRAD_SPLIT", "1") == "1"
                    and self.L.rd_wgrad_split_supported(C.byref(dwd)) == 1)
        f_ws, f_wgrad, f_reduce, fam = ((self.L.rd_wgrad_bf16_workspace_floats, self.L.rd_wgrad_bf16, self.L.rd_wgrad_bf16_reduce, "wgrad_bf16")
                                        if wg_bf16 else
                                        (self.L.rd_wgrad_split_workspace_floats, self.L.rd_wgrad_split, self.L.rd_wgrad_split_reduce, "wgrad_split")
                                        if wg_split else (self.L.rd_wgrad_workspace_floats, self.L.rd_wgrad, self.L.rd_wgrad_reduce, "wgrad"))
        nws = f_ws(C.byref(dwd))
        if nws < 0:
            check(int(nws), "rd_wgrad_workspace_floats(%s)" % name)
        ws = self.buf(int(nws))
        # pre-split operands: both piece-plane sets are requested HERE, on the current stream in front of the dgrad / wgrad fork, so a
        # stand-alone split pass (an operand without a piece-writing producer) is ordered before both consumers
        # (the weight gradient keeps splitting while it stages: its pre-split form measured 0.8-0.9x, profiles/r04_bench_wgrad_split.txt;
        #  RD_WGRAD_PRE=1 opts in)
        wg_pre = (wg_split and os.environ.get("RD_WGRAD_PRE", "0") == "1" and self._pre_ok(x) and self._pre_ok(dout)
                  and self.L.rd_wgrad_split_pre_supported(C.byref(dwd)) == 1)
        sp_d0 = bool(ctx.get("split_dgrad"))
        dg_pre = need_dx and sp_d0 and bool(ctx.get("pre_dgrad")) and self._pre_ok(dout)
        yp = yplane = xp = xplane = None
        if wg_pre or dg_pre:
            yp, yplane = self.pc_in(dout, self.bwd)
        if wg_pre:
            xp, xplane = self.pc_in(x, self.bwd)
        # the weight-gradient chain (wgrad + its slab reductions) gates nothing until the bucket boundary: it runs on the
        # wgrad stream, concurrently with the dgrad / BatchNorm chain that continues on the current stream
        cur = self.streams.index(self._s) if self._s in self.streams else 0
        wst = 2 if cur == 0 else cur

        def launch_wgrad():
            self.edge(self.bwd, name + ".fork_wgrad", cur, wst)
            with self.on(wst):
                if wg_bf16:
                    self.op(self.bwd, name + ".wgrad", self.L.rd_wgrad_bf16_t, self.dt, C.byref(dwd), x.ptr, dout.ptr, _p(ws), self.stream)
                elif wg_pre:
                    self.op(self.bwd, name + ".wgrad", self.L.rd_wgrad_split_pre, C.byref(dwd), xp, xplane, yp, yplane, _p(ws), self.stream)
                else:
                    self.op(self.bwd, name + ".wgrad", f_wgrad, C.byref(dwd), x.ptr, dout.ptr, _p(ws), self.stream)
                self.meta[name + ".wgrad"] = ("wgrad_split_pre" if wg_pre else fam, dwd)
                for w, off in sorted(ctx["weights"], key=lambda t: t[1]):
                    o, i, kh, kw = w.shape
                    if self.batch_reduces:
                        # the slab reductions are not needed before the segment's bucket boundary: they are collected per
                        # stream and issued as ONE batched launch pair at the end of the segment (_flush_reduces)
                        self._pending_reduces.setdefault(self.streams.index(self._s), []).append(
                            (name, dwd, 1 if wg_bf16 else 0, ws, self.grad_of(w), o, i, kh, kw, off))
                    else:
                        self.op(self.bwd, name + ".wreduce", f_reduce, C.byref(dwd), _p(ws), _p(self.grad_of(w)), o, i, kh, kw,
                                off, 0, self.stream)
                if self.batch_reduces and len(self._pending_reduces.get(self.streams.index(self._s), [])) >= self.reduce_batch_max:
                    self._flush_reduces(only=self.streams.index(self._s))

        # the weight-gradient chain is forked BEHIND the dgrad launch rather than beside it: both are MFMA-bound and gain nothing
        # from running together, whereas behind the dgrad the wgrad overlaps the memory-bound BatchNorm kernels that follow
        # (+1.4 % on the step; RD_WGRAD_AFTER_DGRAD=0 restores the side-by-side fork)
        late = os.environ.get("RD_WGRAD_AFTER_DGRAD", "1") == "1" and need_dx
        if not late:
            launch_wgrad()
        if not need_dx:
            return None
        dx_owned = dx is None
        if dx is None:
            dx = self.act(N, H, W, cin)
        if ctx["upproj"]:
            dd, zero_fill = cd.upproj_dgrad(N, H, W, cin, cout, ld_dy=dout.ld, ld_dx=dx.ld), False
        else:
            dd, zero_fill = cd.conv_dgrad(N, H, W, cin, cout, k, ctx["stride"], ctx["pad"], ld_dy=dout.ld, ld_dx=dx.ld)
        self.keep.append(dd)
        if zero_fill:
            assert dx.C == dx.ld, "zero-filled dgrad target must be a whole buffer"
            if addend is not None:
                raise NotImplementedError("addend with a zero-filled stride-2 dgrad")
            if dx_owned:
                # the launch writes only the pixels the stride touches; the buffer is this plan's own and nothing else ever writes it, so the
                # zeros in between are laid down ONCE, here (it was an rd_fill in front of every such launch: 6 per step on the dependent
                # chain of the backward, 8 us each alone and a kernel boundary beside the other streams' work)
                dx.t.zero_()
                self.persistent.append(dx.t)
                self.persistent_zero_checks.append(_strided_zero_check(dx.t, dd))
            else:
                self.op(self.bwd, name + ".zero", self.L.rd_fill, dx.ptr, C.c_int64(dx.t.numel() * dx.t.element_size() // 4), C.c_float(0.0),
                        self.stream)
        sp_d = bool(ctx.get("split_dgrad"))
        # split plans (round 6): reduce-in-epilogue for the conv -> BN -> act -> conv chains, like rd_gconv_bnbwd on the fp32 plan (RD_SPLIT_BNB=0: off)
        # RD_SPLIT_BNB: "0" none, "1" all, or a comma list of kernel families: wino, pre (gconv_sp2), split (8-wave).  Measured on the step
        # (profiles/r06_split_bnb_ab.txt): see the default below.
        bnb_fam = "wino" if ctx.get("wino_dgrad") else "pre" if dg_pre else "split"
        want = os.environ.get("RD_SPLIT_BNB", SPLIT_BNB_DEFAULT)
        split_bnb = (bnb is not None and addend is None and not zero_fill and self.fuse_bn_bwd and self.split
                     and (want == "1" or bnb_fam in want.split(",")) and dx.C % 4 == 0 and dx.ld % 4 == 0)
        if ctx.get("wino_dgrad"):
            # the input gradient of a Winograd layer: the same kernel on the flipped operand (channels swapped, taps rotated by 180 degrees)
            if zero_fill or self.L.rd_wino_supported(H, W, cout, cin, dout.ld, dx.ld) != 1:
                raise RuntimeError("%s: planned as a Winograd input gradient, but the library does not serve %dx%d %d->%d with strides %d / %d"
                                   % (name, H, W, cout, cin, dout.ld, dx.ld))
            self.meta[name + ".dgrad"] = ("wino", dd)
            if split_bnb:
                # the BatchNorm behind this convolution's forward input takes its backward sums from this launch's epilogue (one
                # rd_bn_bwd_reduce_x_t pass over dx and x less per conv -> BN -> ReLU -> conv chain; VERDICT r5 item 2b, second half)
                tiles = self.L.rd_wino_stat_tiles(N, H, W)
                red = self.buf(tiles, 3, dx.C)
                xb, co = bnb["x"], bnb["co"]
                assert xb.M == dx.M and xb.C == dx.C
                self.op(self.bwd, name + ".dgrad", self.L.rd_wino_conv3x3_bnbwd, dout.ptr, N, H, W, cout, dout.ld, _p(ctx["ud"]), dx.ptr, cin, dx.ld,
                        xb.ptr, xb.ld, _p(co["mean"]), _p(co["scale"]), _p(co["shift"]), bnb["act"], _p(red), self.stream)
                self.bnb_out = (red, tiles)
            else:
                self.op(self.bwd, name + ".dgrad", self.L.rd_wino_conv3x3, dout.ptr, N, H, W, cout, dout.ld, _p(ctx["ud"]), dx.ptr, cin, dx.ld,
                        addend.ptr if addend is not None else C.c_void_p(0), addend.ld if addend is not None else 0, C.c_void_p(0), self.stream)
            if late:
                launch_wgrad()
            return dx
        dg_pre = dg_pre and sp_d and self.L.rd_gconv_split_pre_supported(C.byref(dd)) == 1
        if sp_d and not dg_pre and self.L.rd_gconv_split_supported(C.byref(dd)) != 1:
            raise RuntimeError("%s: the input-gradient operand was packed as three bf16 pieces, but the library has neither a split plan for "
                               "the final descriptor nor piece planes for its output-gradient tensor" % name)
        self.meta[name + ".dgrad"] = ("gconv_split_pre" if dg_pre else "gconv_split" if sp_d else "gconv_bf16" if self.bf16 else "gconv", dd)
        # (bf16 plans keep the separate BatchNorm-backward reduce pass: the same fusion in gconv_bf16's epilogue -- parity-green in
        #  round 3 -- made the bf16-storage step 5 % SLOWER, 1790 -> 1697 samples/s: that kernel's epilogue is already its longest
        #  phase, and the extra x loads sit on it)
        if (dg_pre or sp_d) and split_bnb and self.L.rd_gconv_split_bnbwd_supported(C.byref(dd), 1 if dg_pre else 0) == 1:
            tiles = (self.L.rd_gconv_split_pre_stat_tiles if dg_pre else self.L.rd_gconv_split_stat_tiles)(C.byref(dd))
            red = self.buf(tiles, 3, dx.C)
            xb, co = bnb["x"], bnb["co"]
            assert xb.M == dx.M and xb.C == dx.C
            if dg_pre:
                self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_split_pre_bnbwd, C.byref(dd), yp, yplane, _p(ctx["wd"]), C.c_int64(k * k * cin * cout),
                        dx.ptr, xb.ptr, xb.ld, _p(co["mean"]), _p(co["scale"]), _p(co["shift"]), bnb["act"], _p(red), self.stream)
            else:
                self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_split_bnbwd, C.byref(dd), dout.ptr, _p(ctx["wd"]), C.c_int64(k * k * cin * cout),
                        dx.ptr, xb.ptr, xb.ld, _p(co["mean"]), _p(co["scale"]), _p(co["shift"]), bnb["act"], _p(red), self.stream)
            self.bnb_out = (red, tiles)
        elif dg_pre:
            self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_split_pre, C.byref(dd), yp, yplane, _p(ctx["wd"]), C.c_int64(k * k * cin * cout), dx.ptr,
                    C.c_void_p(0), 0, 0, addend.ptr if addend is not None else C.c_void_p(0), addend.ld if addend is not None else 0,
                    C.c_void_p(0), self.stream)
        elif sp_d:
            # (no BatchNorm-backward sums from this epilogue: the caller keeps its reduce pass, as in the bf16 plans)
            self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_split, C.byref(dd), dout.ptr, _p(ctx["wd"]), C.c_int64(k * k * cin * cout), dx.ptr,
                    C.c_void_p(0), 0, 0, addend.ptr if addend is not None else C.c_void_p(0), addend.ld if addend is not None else 0,
                    C.c_void_p(0), self.stream)
        elif self.bf16:
            self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_bf16_t, self.dt, C.byref(dd), dout.ptr, _p(ctx["wd"]), dx.ptr, C.c_void_p(0), 0, 0,
                    addend.ptr if addend is not None else C.c_void_p(0), addend.ld if addend is not None else 0,
                    C.c_void_p(0), self.stream)
        elif self._c16_split(dd):
            self.op(self.bwd, name + ".dgrad", self.L.rd_conv16_split, C.byref(dd), dout.ptr, _p(ctx["wd"]), dx.ptr,
                    addend.ptr if addend is not None else C.c_void_p(0), addend.ld if addend is not None else 0, C.c_void_p(0), self.stream)
            self.meta[name + ".dgrad"] = ("conv16_split", dd)
        else:
            ws_d = self._gconv_ws(dd, name + ".dgrad")      # (plans the descriptor: table pin / tuner first)
            fuse = (bnb is not None and addend is None and not zero_fill and self.fuse_bn_bwd
                    and self.L.rd_gconv_bnbwd_supported(C.byref(dd)) == 1)
            if fuse:
                # the BatchNorm behind this convolution's forward input takes its backward sums from this launch's epilogue:
                # one pass over (dy, x) and one launch less per conv -> BN -> ReLU -> conv chain
                tiles = self.L.rd_gconv_stat_tiles_ws(C.byref(dd))
                red = self.buf(tiles, 3, dx.C)
                xb, co = bnb["x"], bnb["co"]
                assert xb.M == dx.M and xb.C == dx.C
                self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_bnbwd, C.byref(dd), dout.ptr, _p(ctx["wd"]), dx.ptr, xb.ptr, xb.ld,
                        _p(co["mean"]), _p(co["scale"]), _p(co["shift"]), bnb["act"], _p(red), _p(ws_d), self.stream)
                self.bnb_out = (red, tiles)
                self.meta[name + ".dgrad"] = ("gconv_bnb", dd)      # (the BNB instantiation of the kernel: its own name in a trace)
            else:
                self.op(self.bwd, name + ".dgrad", self.L.rd_gconv_ws, C.byref(dd), dout.ptr, _p(ctx["wd"]), dx.ptr,
                        addend.ptr if addend is not None else C.c_void_p(0), addend.ld if addend is not None else 0,
                        C.c_void_p(0), _p(ws_d), self.stream)
        if late:
            launch_wgrad()
        return dx

    def _flush_reduces(self, only=None):
        """Emit the pending weight-gradient slab reductions of every stream as one rd_wgrad_reduce_batched op each (two launches:
        the first stage of the many-split layers, then every OIHW gradient).  Called at bucket boundaries, before the streams'
        segment-end events / joins."""
        import numpy as np
        from ._lib import RdReduceJob
        for k in sorted(self._pending_reduces):
            jobs = self._pending_reduces[k]
            if not jobs or (only is not None and k != only):
                continue
            self._pending_reduces[k] = []
            arr = (RdReduceJob * len(jobs))()
            bj1, bj2 = [], []
            for q, (name, dwd, is_bf16, ws, grad, o, i, kh, kw, off) in enumerate(jobs):
                check(self.L.rd_wgrad_reduce_job(C.byref(dwd), is_bf16, _p(ws), _p(grad), o, i, kh, kw, off, 0, C.byref(arr[q])),
                      "rd_wgrad_reduce_job(%s)" % name)
                arr[q].first_block1, arr[q].first_block2 = len(bj1), len(bj2)
                bj1 += [q] * arr[q].n_blocks1
                bj2 += [q] * arr[q].n_blocks2
            table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.dev)
            t1 = torch.tensor(bj1 or [0], dtype=torch.int32, device=self.dev)
            t2 = torch.tensor(bj2, dtype=torch.int32, device=self.dev)
            self.keep += [table, t1, t2]
            with self.on(k):
                self.op(self.bwd, "segment%d.s%d.b%d.wreduce_all" % (len(self.bwd_segments) if hasattr(self, "bwd_segments") else 0, k, len(self.reduce_batches)),
                        self.L.rd_wgrad_reduce_batched, _p(table), _p(t1), len(bj1), _p(t2), len(bj2), self.stream)
            self.reduce_batches.append((len(self.bwd_segments) if hasattr(self, "bwd_segments") else 0, k, [j[0] for j in jobs]))

    # ------------------------------------------------------------------ inference: conv + folded BatchNorm (+ReLU, +residual)
    def conv_bn_eval(self, name, x, parts, k, stride, pad, act, act_cols=None, addend=None, out=None, upproj=False):
        """Eval-mode conv+BN(+act)(+residual) as ONE kernel: the BatchNorm scale gamma/sqrt(var+eps)
        is folded into the packed weights, its shift is the epilogue bias.  parts: [(conv weight, column offset, bn)]."""
        N, H, W, cin = x.N, x.H, x.W, x.C
        cout = sum(w.shape[0] for w, _, _ in parts)
        d = cd.upproj_fwd(N, H, W, cin, cout, ldi=x.ld) if upproj else cd.conv_fwd(N, H, W, cin, cout, k, stride, pad, ldi=x.ld)
        if out is None:
            out = self.act(N, d.Ho, d.Wo, cout)
        d.ldo = out.ld
        wp = self.buf(k * k, cin, cout, dtype=torch.bfloat16 if self.bf16 else torch.float32)
        bias = self.buf(cout)
        scale = self.buf(cout)
        for w, off, bn in parts:
            o, i, kh, kw = w.shape
            sc = C.c_void_p(scale.data_ptr() + 4 * off)
            sh = C.c_void_p(bias.data_ptr() + 4 * off)
            self.evalcoef_jobs.append((o, bn, sc, sh))       # all folded BatchNorms: ONE launch (see _finish_pack_jobs)
            self.pack_jobs.append((w, wp, o, i, kh * kw, cout, off, i, 0, scale[off:off + o], 2 if self.bf16 else 1))
        self.keep += [d, scale]
        if self.bf16:
            self.op(self.fwd, name, self.L.rd_gconv_bf16_t, self.dt, C.byref(d), x.ptr, _p(wp), out.ptr, _p(bias), act,
                    cout if act_cols is None else act_cols, addend.ptr if addend is not None else C.c_void_p(0),
                    addend.ld if addend is not None else 0, C.c_void_p(0), self.stream)
            self.meta[name] = ("gconv_bf16", d)
            return out
        ws = self._gconv_ws(d, name)
        self.op(self.fwd, name, self.L.rd_gconv_fused, C.byref(d), x.ptr, _p(wp), out.ptr, _p(bias), act,
                cout if act_cols is None else act_cols, addend.ptr if addend is not None else C.c_void_p(0),
                addend.ld if addend is not None else 0, _p(ws), self.stream)
        self.meta[name] = ("gconv", d)
        return out

    # ------------------------------------------------------------------ batch norm
    def bn_coeffs(self, name, bn, stat, tiles, ld, c0, count, lst=None):
        """Per-channel (mean, invstd, scale, shift) of a BatchNorm2d over channels [c0, c0+C) of fused stats."""
        lst = self.fwd if lst is None else lst
        Cc = bn.weight.shape[0]
        co = dict(name=name, bn=bn, C=Cc, count=count, mean=self.buf(Cc), invstd=self.buf(Cc), scale=self.buf(Cc), shift=self.buf(Cc))
        if self.train:
            self.op(lst, name + ".finalize", self.L.rd_bn_finalize, _p(stat), tiles, ld, c0, Cc, C.c_int64(count), _p(bn.weight),
                    _p(bn.bias), C.c_float(BN_EPS), C.c_float(BN_MOMENTUM), _p(bn.running_mean), _p(bn.running_var),
                    _p(bn.num_batches_tracked), _p(co["mean"]), _p(co["invstd"]), _p(co["scale"]), _p(co["shift"]), self.stream)
        else:
            self.op(lst, name + ".evalcoef", self.L.rd_bn_eval_coeffs, Cc, _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
                    _p(bn.running_var), C.c_float(BN_EPS), _p(co["scale"]), _p(co["shift"]), self.stream)
        return co

    def bn_act(self, name, x1, co1, act, x2=None, co2=None, out=None):
        """out = act(bn1(x1) [+ bn2(x2) | + x2])."""
        if out is None:
            out = self.act(x1.N, x1.H, x1.W, x1.C)
        if self._pre_ok(out):
            pcp, pcs = self.pc_out(out)
            self.op(self.fwd, name, self.L.rd_bn_act_p, x1.ptr, x1.ld, _p(co1["scale"]), _p(co1["shift"]),
                    x2.ptr if x2 is not None else C.c_void_p(0), x2.ld if x2 is not None else 0,
                    _p(co2["scale"]) if co2 is not None else C.c_void_p(0), _p(co2["shift"]) if co2 is not None else C.c_void_p(0),
                    out.ptr, out.ld, C.c_int64(x1.M), x1.C, act, pcp, pcs, self.stream)
            self.taps[name] = out
            return out
        self.op(self.fwd, name, self.L.rd_bn_act_t, self.dt, x1.ptr, x1.ld, _p(co1["scale"]), _p(co1["shift"]),
                x2.ptr if x2 is not None else C.c_void_p(0), x2.ld if x2 is not None else 0,
                _p(co2["scale"]) if co2 is not None else C.c_void_p(0), _p(co2["shift"]) if co2 is not None else C.c_void_p(0),
                out.ptr, out.ld, C.c_int64(x1.M), x1.C, act, self.stream)
        self.taps[name] = out
        return out

    def bn_join_bwd(self, name, dy, y, act, x1, co1, x2=None, co2=None, dx2=None, lone=False, pre=None):
        """Backward of out = act(bn1(x1) [+ bn2(x2) | + x2]).  Returns (dx1, dx2 or g-if-identity-residual).
        lone: out is act(bn1(x1)) with nothing added (callers with an identity residual pass x2=None too, hence the flag)."""
        M, Cc = x1.M, x1.C
        if lone and x2 is None and act != ACT_NONE and pre is not None:
            # the sums came out of the dgrad launch that produced dy (conv_bwd(..., bnb=...)): no reduce pass
            red, tiles = pre
            dx1 = self.act(x1.N, x1.H, x1.W, Cc)
            self._bn_apply_x(name + ".bn1", dy, x1, red, tiles, co1, act, dx1)
            return dx1, None
        tiles = self.L.rd_bn_bwd_tiles(C.c_int64(M), Cc)
        red = self.buf(tiles, 3, Cc)
        if lone and x2 is None and act != ACT_NONE:
            # lone act(bn(x1)): the activation's sign is recomputed from x1 in both passes -- y is not read and the masked
            # gradient is never materialised
            self.op(self.bwd, name + ".bwd_reduce", self.L.rd_bn_bwd_reduce_x_t, self.dt, dy.ptr, dy.ld, x1.ptr, x1.ld, _p(co1["mean"]),
                    _p(co1["scale"]), _p(co1["shift"]), C.c_void_p(0), 0, C.c_int64(M), Cc, act, _p(red), self.stream)
            dx1 = self.act(x1.N, x1.H, x1.W, Cc)
            self._bn_apply_x(name + ".bn1", dy, x1, red, tiles, co1, act, dx1)
            return dx1, None
        if co2 is not None and act != ACT_NONE:
            # act(bn1(x1) + bn2(x2)): same idea for both operands at once (y not read, g not materialised, dy read once per pass)
            self.op(self.bwd, name + ".bwd_reduce", self.L.rd_bn_bwd_reduce_x2_t, self.dt, dy.ptr, dy.ld, x1.ptr, x1.ld, _p(co1["mean"]),
                    _p(co1["scale"]), _p(co1["shift"]), x2.ptr, x2.ld, _p(co2["mean"]), _p(co2["scale"]), _p(co2["shift"]),
                    C.c_int64(M), Cc, act, _p(red), self.stream)
            dx1 = self.act(x1.N, x1.H, x1.W, Cc)
            if dx2 is None:
                dx2 = self.act(x2.N, x2.H, x2.W, Cc)
            b1, b2 = co1["bn"], co2["bn"]
            coef = self.buf(6 * Cc)
            if self._pre_ok(dx1) and self._pre_ok(dx2):
                p1, s1_ = self.pc_out(dx1)
                p2, s2_ = self.pc_out(dx2)
                self.op(self.bwd, name + ".bwd_apply", self.L.rd_bn_bwd_apply_x2_p, dy.ptr, dy.ld, x1.ptr, x1.ld, x2.ptr, x2.ld, _p(red), tiles,
                        _p(b1.weight), _p(co1["mean"]), _p(co1["invstd"]), _p(co1["scale"]), _p(co1["shift"]),
                        _p(b2.weight), _p(co2["mean"]), _p(co2["invstd"]), _p(co2["scale"]), _p(co2["shift"]), act,
                        _p(self.grad_of(b1.weight)), _p(self.grad_of(b1.bias)), _p(self.grad_of(b2.weight)), _p(self.grad_of(b2.bias)),
                        _p(coef), dx1.ptr, dx1.ld, dx2.ptr, dx2.ld, C.c_int64(M), Cc, p1, s1_, p2, s2_, self.stream)
                return dx1, dx2
            self.op(self.bwd, name + ".bwd_apply", self.L.rd_bn_bwd_apply_x2_t, self.dt, dy.ptr, dy.ld, x1.ptr, x1.ld, x2.ptr, x2.ld, _p(red), tiles,
                    _p(b1.weight), _p(co1["mean"]), _p(co1["invstd"]), _p(co1["scale"]), _p(co1["shift"]),
                    _p(b2.weight), _p(co2["mean"]), _p(co2["invstd"]), _p(co2["scale"]), _p(co2["shift"]), act,
                    _p(self.grad_of(b1.weight)), _p(self.grad_of(b1.bias)), _p(self.grad_of(b2.weight)), _p(self.grad_of(b2.bias)),
                    _p(coef), dx1.ptr, dx1.ld, dx2.ptr, dx2.ld, C.c_int64(M), Cc, self.stream)
            return dx1, dx2
        # with no activation g == dy: skip the copy and let the apply pass read dy directly
        g = self.act(x1.N, x1.H, x1.W, Cc) if act != ACT_NONE else dy
        if True:
            self.op(self.bwd, name + ".bwd_reduce", self.L.rd_bn_bwd_reduce_t, self.dt, dy.ptr, dy.ld, y.ptr if y is not None else C.c_void_p(0),
                    y.ld if y is not None else 0, x1.ptr, x1.ld, _p(co1["mean"]),
                    x2.ptr if co2 is not None else C.c_void_p(0), x2.ld if co2 is not None else 0,
                    _p(co2["mean"]) if co2 is not None else C.c_void_p(0), g.ptr if act != ACT_NONE else C.c_void_p(0), g.ld,
                    C.c_int64(M), Cc, act, _p(red), self.stream)
        dx1 = self.act(x1.N, x1.H, x1.W, Cc)
        self._bn_apply(name + ".bn1", g, x1, red, tiles, 1, co1, dx1)
        if co2 is not None:
            if dx2 is None:
                dx2 = self.act(x2.N, x2.H, x2.W, Cc)
            self._bn_apply(name + ".bn2", g, x2, red, tiles, 2, co2, dx2)
            return dx1, dx2
        return dx1, g

    def _bn_apply(self, name, g, x, red, tiles, which, co, dx):
        bn = co["bn"]
        coef = self.buf(3 * co["C"])
        if self._pre_ok(dx):
            pcp, pcs = self.pc_out(dx)
            self.op(self.bwd, name + ".bwd_apply", self.L.rd_bn_bwd_apply_p, g.ptr, g.ld, x.ptr, x.ld, _p(red), tiles, which,
                    _p(bn.weight), _p(co["mean"]), _p(co["invstd"]), _p(self.grad_of(bn.weight)), _p(self.grad_of(bn.bias)),
                    _p(coef), dx.ptr, dx.ld, C.c_int64(x.M), co["C"], pcp, pcs, self.stream)
            return
        self.op(self.bwd, name + ".bwd_apply", self.L.rd_bn_bwd_apply_t, self.dt, g.ptr, g.ld, x.ptr, x.ld, _p(red), tiles, which,
                _p(bn.weight), _p(co["mean"]), _p(co["invstd"]), _p(self.grad_of(bn.weight)), _p(self.grad_of(bn.bias)),
                _p(coef), dx.ptr, dx.ld, C.c_int64(x.M), co["C"], self.stream)

    def _bn_apply_x(self, name, dy, x, red, tiles, co, act, dx):
        bn = co["bn"]
        coef = self.buf(3 * co["C"])
        if self._pre_ok(dx):
            pcp, pcs = self.pc_out(dx)
            self.op(self.bwd, name + ".bwd_apply", self.L.rd_bn_bwd_apply_x_p, dy.ptr, dy.ld, x.ptr, x.ld, _p(red), tiles,
                    _p(bn.weight), _p(co["mean"]), _p(co["invstd"]), _p(co["scale"]), _p(co["shift"]), act,
                    _p(self.grad_of(bn.weight)), _p(self.grad_of(bn.bias)), _p(coef), dx.ptr, dx.ld, C.c_int64(x.M), co["C"], pcp, pcs, self.stream)
            return
        self.op(self.bwd, name + ".bwd_apply", self.L.rd_bn_bwd_apply_x_t, self.dt, dy.ptr, dy.ld, x.ptr, x.ld, _p(red), tiles,
                _p(bn.weight), _p(co["mean"]), _p(co["invstd"]), _p(co["scale"]), _p(co["shift"]), act,
                _p(self.grad_of(bn.weight)), _p(self.grad_of(bn.bias)), _p(coef), dx.ptr, dx.ld, C.c_int64(x.M), co["C"], self.stream)

    # ------------------------------------------------------------------ network pieces
    def _stem(self, name, planes, strides, conv, bn, act, out_name):
        """7x7/2 conv straight from input planes -> BN -> act -> maxpool(3,2,1)."""
        N, H, W = self.N, self.H, self.W
        cin, cout = conv.weight.shape[1], conv.weight.shape[0]
        Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        raw = self.act(N, Hc, Wc, cout)
        wp = self.buf(49, cin, cout)
        self.pack_jobs.append((conv.weight, wp, cout, cin, 49, cout, 0, cin, 0, None, 0))   # stem kernels read the plain layout
        tiles = self.L.rd_stem_stat_tiles(N, H, W)
        stat = self.buf(tiles, 2, cout) if self.train else None
        pl = (C.c_void_p * 3)(*([p for p in planes] + [None] * (3 - len(planes))))
        st = (C.c_int64 * 3)(*(list(strides) + [0] * (3 - len(strides))))
        self.keep += [pl, st]
        # bf16 plans: the 64-channel RGB stem forward runs on the bf16 matrix cores too (220 vs 574 us at b=16; its weight
        # gradient stays fp32); the 16-channel depth stem is faster on the fp32 kernel (175 vs 330 us)
        # split plans: the 64-channel RGB stem on the bf16 matrix cores with three-piece operands (fp32 arithmetic, csrc/stem_bf16.hip NP = 3)
        if self.split and not self.bf16 and cout >= 64 and os.environ.get("RD_STEM_SPLIT", "1") == "1":
            self.op(self.fwd, name, self.L.rd_stem_fwd_split, pl, st, cin, N, H, W, _p(wp), cout, raw.ptr, _p(stat), self.stream)
        else:
            self.op(self.fwd, name, self.L.rd_stem_fwd_bf16_t if (self.bf16 and cout >= 64) else self.L.rd_stem_fwd_t, self.dt, pl, st, cin, N, H, W,
                    _p(wp), cout, raw.ptr, _p(stat), self.stream)
        co = self.bn_coeffs(name + ".bn", bn, stat, tiles, cout, 0, N * Hc * Wc)
        Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
        pooled = self.act(N, Hp, Wp, cout)
        idx = self.buf(N, Hp, Wp, cout, dtype=torch.uint8)
        if self._pre_ok(pooled):
            pcp, pcs = self.pc_out(pooled)
            self.op(self.fwd, name + ".bnact_pool", self.L.rd_bnact_maxpool_fwd_p, raw.ptr, _p(co["scale"]), _p(co["shift"]), act, N, Hc, Wc,
                    cout, pooled.ptr, pooled.ld, _p(idx), pcp, pcs, self.stream)
        else:
            self.op(self.fwd, name + ".bnact_pool", self.L.rd_bnact_maxpool_fwd_t, self.dt, raw.ptr, _p(co["scale"]), _p(co["shift"]), act, N, Hc, Wc,
                    cout, pooled.ptr, pooled.ld, _p(idx), self.stream)
        self.taps[out_name] = pooled
        return pooled, dict(name=name, pl=pl, st=st, cin=cin, cout=cout, raw=raw, wp=wp, co=co, idx=idx, act=act, conv=conv,
                            pooled=pooled, Hc=Hc, Wc=Wc)

    def _stem_bwd(self, ctx, dpooled, dgrad_channel=None):
        N, H, W = self.N, self.H, self.W
        raw, co, cout, cin = ctx["raw"], ctx["co"], ctx["cout"], ctx["cin"]
        # the pooling gather applies the activation derivative and, in the same pass, produces the BatchNorm-backward sums
        # (g and x are in registers there): no separate reduce pass over the two largest tensors of the network
        tiles = self.L.rd_bnact_maxpool_bwd_tiles(N, ctx["Hc"], ctx["Wc"], cout)
        red = self.buf(tiles, 3, cout)
        # Nobody but the weight gradient reads this BatchNorm's input gradient unless an input gradient is asked of the stem: on the plans
        # whose stem weight gradient runs on the bf16 matrix cores the apply pass (a read of g and raw and a write of dx -- 1.1 GB for the
        # RGB stem at b = 16 -- as the last kernel of the main chain) is folded into that kernel's staging waves (rd_stem_wgrad_split_bn_t:
        # same bits, tests/test_gpu_stem.py; RD_STEM_WGRAD_BN=0 restores the separate pass).  (Folded into the fp32-MFMA kernel it was
        # level: no registers left there to keep a second operand's loads in flight.  profiles/r05_stem_tail.txt)
        wg_split = (self.split or self.storage == "bf16") and self.L.rd_stem_wgrad_split_supported(cin, cout) == 1
        can_fuse = wg_split and dgrad_channel is None and cin != 2 and os.environ.get("RD_STEM_WGRAD_BN", "1") == "1"
        # where the apply pass stays: two passes over the pooled gradient (sums, then the BatchNorm input gradient stored directly; the
        # full-resolution g is never materialised) -- level on fp32 tensors, +0.6 % on bf16 storage since the 2 x 2 gather (1877 vs 1866
        # and 278.1 vs 276.0 samples/s, configs 3 / 5, one job; profiles/r05_stem_tail.txt)
        two_pass = os.environ.get("RD_STEM_BWD_TWO_PASS", "1" if self.storage == "bf16" and not can_fuse else "0") == "1"
        fuse_apply = can_fuse and not two_pass
        dx = None if fuse_apply else self.act(raw.N, raw.H, raw.W, cout)
        if two_pass:
            # the first pass takes the sums only, the second repeats the (quarter-size) gather and stores the BatchNorm input gradient
            # directly.  0.6 GB/step less HBM traffic on fp32 tensors, bit-identical results
            self.op(self.bwd, ctx["name"] + ".pool_bwd", self.L.rd_bnact_maxpool_bwd_stats_t, self.dt, dpooled.ptr, dpooled.ld, _p(ctx["idx"]), raw.ptr,
                    _p(co["scale"]), _p(co["shift"]), ctx["act"], N, ctx["Hc"], ctx["Wc"], cout, C.c_void_p(0), _p(co["mean"]), _p(red), self.stream)
            bn = co["bn"]
            coef = self.buf(3 * cout)
            self.op(self.bwd, ctx["name"] + ".bn.bn1.bwd_apply", self.L.rd_bnact_maxpool_bwd_apply_t, self.dt, dpooled.ptr, dpooled.ld, _p(ctx["idx"]),
                    raw.ptr, _p(co["scale"]), _p(co["shift"]), ctx["act"], N, ctx["Hc"], ctx["Wc"], cout, _p(red), tiles, _p(bn.weight),
                    _p(co["mean"]), _p(co["invstd"]), _p(self.grad_of(bn.weight)), _p(self.grad_of(bn.bias)), _p(coef), dx.ptr, self.stream)
        else:
            g = self.act(N, ctx["Hc"], ctx["Wc"], cout)
            self.op(self.bwd, ctx["name"] + ".pool_bwd", self.L.rd_bnact_maxpool_bwd_stats_t, self.dt, dpooled.ptr, dpooled.ld, _p(ctx["idx"]), raw.ptr,
                    _p(co["scale"]), _p(co["shift"]), ctx["act"], N, ctx["Hc"], ctx["Wc"], cout, g.ptr, _p(co["mean"]), _p(red), self.stream)
            if not fuse_apply:
                self._bn_apply(ctx["name"] + ".bn.bn1", g, raw, red, tiles, 1, co, dx)
        nws = self.L.rd_stem_wgrad_workspace_floats(N, H, W, cin, cout)
        ws = self.buf(int(nws))
        cur = self.streams.index(self._s) if self._s in self.streams else 0
        wst = 2 if cur == 0 else cur
        self.edge(self.bwd, ctx["name"] + ".fork_wgrad", cur, wst)
        with self.on(wst):
            # split and bf16-storage plans: the stem weight gradients on the bf16 matrix cores too (csrc/stem_wgrad_split.hip: three-piece
            # operands, fp32 arithmetic; the RGB one is the last kernel of the step, alone on the chip)
            if fuse_apply:
                bn = co["bn"]
                coef = self.buf(3 * cout)
                # (the op keeps the apply pass's name: it is this BatchNorm's one gradient writer)
                self.op(self.bwd, ctx["name"] + ".bn.bn1.bwd_apply", self.L.rd_stem_wgrad_split_bn_t, self.dt, ctx["pl"], ctx["st"], cin, N, H, W,
                        g.ptr, raw.ptr, _p(red), tiles, _p(bn.weight), _p(co["mean"]), _p(co["invstd"]), _p(self.grad_of(bn.weight)),
                        _p(self.grad_of(bn.bias)), _p(coef), cout, _p(self.grad_of(ctx["conv"].weight)), _p(ws), self.stream)
            else:
                f_wg = self.L.rd_stem_wgrad_split_t if wg_split else self.L.rd_stem_wgrad_t
                self.op(self.bwd, ctx["name"] + ".wgrad", f_wg, self.dt, ctx["pl"], ctx["st"], cin, N, H, W, dx.ptr, cout,
                        _p(self.grad_of(ctx["conv"].weight)), _p(ws), self.stream)
        if dgrad_channel is not None:
            ci, dst = dgrad_channel
            self.op(self.bwd, ctx["name"] + ".dgrad_ch", self.L.rd_stem_dgrad_channel_t, self.dt, dx.ptr, _p(ctx["wp"]), N, H, W, cin, ci, cout,
                    _p(dst), self.stream)

    def _block(self, name, blk, x, out=None):
        """BasicBlock forward (models.py:96-112)."""
        stride = blk.stride
        if not self.train:
            y1 = self.conv_bn_eval(name + ".conv1", x, [(blk.conv1.weight, 0, blk.bn1)], 3, stride, 1, ACT_RELU)
            skip = x
            if blk.downsample is not None:
                skip = self.conv_bn_eval(name + ".downsample.0", x, [(blk.downsample[0].weight, 0, blk.downsample[1])], 1, stride, 0,
                                         ACT_NONE)
            y = self.conv_bn_eval(name + ".conv2", y1, [(blk.conv2.weight, 0, blk.bn2)], 3, 1, 1, ACT_RELU, addend=skip, out=out)
            self.taps[name] = y
            return y, None
        r1, c1 = self.conv_fwd(name + ".conv1", x, [(blk.conv1.weight, 0)], 3, stride, 1)
        co1 = self.bn_coeffs(name + ".bn1", blk.bn1, c1["stat"], c1["tiles"], r1.C, 0, r1.M)
        y1 = self.bn_act(name + ".relu1", r1, co1, ACT_RELU)
        r2, c2 = self.conv_fwd(name + ".conv2", y1, [(blk.conv2.weight, 0)], 3, 1, 1)
        co2 = self.bn_coeffs(name + ".bn2", blk.bn2, c2["stat"], c2["tiles"], r2.C, 0, r2.M)
        ctx = dict(name=name, x=x, r1=r1, c1=c1, co1=co1, y1=y1, r2=r2, c2=c2, co2=co2, ds=None)
        if blk.downsample is not None:
            rd_, cds = self.conv_fwd(name + ".downsample.0", x, [(blk.downsample[0].weight, 0)], 1, stride, 0)
            cods = self.bn_coeffs(name + ".downsample.1", blk.downsample[1], cds["stat"], cds["tiles"], rd_.C, 0, rd_.M)
            y = self.bn_act(name, r2, co2, ACT_RELU, x2=rd_, co2=cods, out=out)
            ctx.update(ds=dict(rd=rd_, c=cds, co=cods))
        else:
            y = self.bn_act(name, r2, co2, ACT_RELU, x2=x, out=out)
        ctx["y"] = y
        return y, ctx

    def _block_bwd(self, ctx, dy):
        """Returns the gradient w.r.t. the block input."""
        name = ctx["name"]
        if ctx["ds"] is not None:
            ds = ctx["ds"]
            dr2, drd = self.bn_join_bwd(name, dy, ctx["y"], ACT_RELU, ctx["r2"], ctx["co2"], x2=ds["rd"], co2=ds["co"])
        else:
            dr2, g = self.bn_join_bwd(name, dy, ctx["y"], ACT_RELU, ctx["r2"], ctx["co2"])
        dy1 = self.conv_bwd(ctx["c2"], dr2, bnb=dict(x=ctx["r1"], co=ctx["co1"], act=ACT_RELU))
        dr1, _ = self.bn_join_bwd(name + ".relu1", dy1, ctx["y1"], ACT_RELU, ctx["r1"], ctx["co1"], lone=True, pre=self.bnb_out)
        self.taps["grad_out:" + name] = dy
        if ctx["ds"] is not None:
            dx_part = self.conv_bwd(ctx["ds"]["c"], drd)          # 1x1 stride-2 dgrad (zero-filled odd pixels)
            dx = self.conv_bwd(ctx["c1"], dr1, addend=dx_part)
        else:
            dx = self.conv_bwd(ctx["c1"], dr1, addend=g)
        self.taps["grad_in:" + name] = dx
        return dx

    def _upproj(self, name, mod, x):
        """UpProjModule forward (models.py:181-209) with the unpool folded into a 4-phase convolution."""
        Cc = x.C
        half = Cc // 2
        ub, bb = mod.upper_branch, mod.bottom_branch
        if not self.train:
            R = self.conv_bn_eval(name + ".conv5x5", x, [(ub.conv1.weight, 0, ub.batchnorm1), (bb.conv.weight, half, bb.batchnorm)], 5, 1, 2,
                                  ACT_RELU, act_cols=half, upproj=True)
            y = self.conv_bn_eval(name + ".upper_branch.conv2", R.chan(0, half), [(ub.conv2.weight, 0, ub.batchnorm2)], 3, 1, 1, ACT_RELU,
                                  addend=R.chan(half, half))
            self.taps[name] = y
            return y, None
        R, cR = self.conv_fwd(name + ".conv5x5", x, [(ub.conv1.weight, 0), (bb.conv.weight, half)], 5, 1, 2, upproj=True)
        M = R.M
        co_u1 = self.bn_coeffs(name + ".upper_branch.batchnorm1", ub.batchnorm1, cR["stat"], cR["tiles"], Cc, 0, M)
        co_b = self.bn_coeffs(name + ".bottom_branch.batchnorm", bb.batchnorm, cR["stat"], cR["tiles"], Cc, half, M)
        y1 = self.bn_act(name + ".upper_branch.relu", R.chan(0, half), co_u1, ACT_RELU)
        r2, c2 = self.conv_fwd(name + ".upper_branch.conv2", y1, [(ub.conv2.weight, 0)], 3, 1, 1)
        co_u2 = self.bn_coeffs(name + ".upper_branch.batchnorm2", ub.batchnorm2, c2["stat"], c2["tiles"], half, 0, M)
        y = self.bn_act(name, r2, co_u2, ACT_RELU, x2=R.chan(half, half), co2=co_b)
        return y, dict(name=name, x=x, R=R, cR=cR, co_u1=co_u1, co_b=co_b, y1=y1, r2=r2, c2=c2, co_u2=co_u2, y=y, half=half)

    def _upproj_bwd(self, ctx, dy):
        name, half, R = ctx["name"], ctx["half"], ctx["R"]
        dR = self.act(R.N, R.H, R.W, R.C)
        dr2, _ = self.bn_join_bwd(name, dy, ctx["y"], ACT_RELU, ctx["r2"], ctx["co_u2"], x2=R.chan(half, half), co2=ctx["co_b"],
                                  dx2=dR.chan(half, half))
        x1, co = R.chan(0, half), ctx["co_u1"]
        dy1 = self.conv_bwd(ctx["c2"], dr2, bnb=dict(x=x1, co=co, act=ACT_RELU))
        if self.bnb_out is not None:
            red, tiles = self.bnb_out
        else:
            M, tiles = R.M, self.L.rd_bn_bwd_tiles(C.c_int64(R.M), half)
            red = self.buf(tiles, 3, half)
            self.op(self.bwd, name + ".bn1.bwd_reduce", self.L.rd_bn_bwd_reduce_x_t, self.dt, dy1.ptr, dy1.ld, x1.ptr, x1.ld, _p(co["mean"]),
                    _p(co["scale"]), _p(co["shift"]), C.c_void_p(0), 0, C.c_int64(M), half, ACT_RELU, _p(red), self.stream)
        self._bn_apply_x(name + ".bn1", dy1, x1, red, tiles, co, ACT_RELU, dR.chan(0, half))
        dx = self.conv_bwd(ctx["cR"], dR)
        self.taps["grad_out:" + name] = dy
        self.taps["grad_in:" + name] = dx
        self.taps["grad_R:" + name] = dR
        self.taps["grad_y1:" + name] = dy1
        return dx

    # ------------------------------------------------------------------ whole network
    def _build(self):
        m, N, H, W = self.m, self.N, self.H, self.W
        hw = H * W
        ndep_in = m.conv1_depth.weight.shape[1]
        if self.x_source is not None:
            self.x_in = self.x_source
        else:
            self.x_in = self.buf(N, 3 + (ndep_in if self.depth_planes is None else 1), H, W)
        ctot = self.x_in.shape[1]
        xp = self.x_in.data_ptr()
        rgb_planes = [xp + 4 * hw * c for c in range(3)]
        rgb_strides = [ctot * hw] * 3
        if self.depth_planes is None:
            ndep = m.conv1_depth.weight.shape[1]
            dep_planes = [xp + 4 * hw * (3 + c) for c in range(ndep)]
            dep_strides = [ctot * hw] * ndep
        else:
            dep_planes = [t.data_ptr() for t in self.depth_planes]
            dep_strides = [hw] * len(dep_planes)
        # encoders: RGB on the main stream, the (small-channel, low-occupancy) depth encoder concurrently on stream 1
        self.probe(self.fwd, "fwd_begin", 0)
        self.edge(self.fwd, "fork_depth", 0, 1)
        a, self.c_stem_rgb = self._stem("conv1", rgb_planes, rgb_strides, m.conv1, m.bn1, ACT_RELU, "maxpool")
        with self.on(1):
            d_, self.c_stem_d = self._stem("conv1_depth", dep_planes, dep_strides, m.conv1_depth, m.bn1_depth, ACT_LEAKY02, "maxpool_depth")
        # the stems read the network input through small HOST arrays of plane pointers / image strides (the ops hold the arrays' addresses):
        # a caller may point them at its own [N,C,H,W] batch instead of copying it into x_in (bind_input; main.HipTrainStep does)
        self.x_bind = [(self.c_stem_rgb["pl"], self.c_stem_rgb["st"], 0, 3)]
        if self.depth_planes is None:
            self.x_bind.append((self.c_stem_d["pl"], self.c_stem_d["st"], 3, ndep))
        self._x_bound = xp           # base pointer the stems' plane tables point at (bind_input)
        # The weight pack of everything but the stems runs on the (then idle) weight-gradient stream beside the stems and their pooling
        # (_finish_pack_jobs): at the head of the step nothing else could overlap its ~0.2 ms.  Both encoder chains pick it up here.
        self.pack_overlap = self.train and self.multi_stream and os.environ.get("RD_PACK_OVERLAP", "1") == "1"
        if self.pack_overlap:
            self.edge(self.fwd, "pack_join", 2, 0)
            with self.on(1):
                self.edge(self.fwd, "pack_join_depth", 2, 1)
        self.blocks_rgb, self.blocks_d = [], []
        layers_rgb = [("layer1", m.layer1), ("layer2", m.layer2), ("layer3", m.layer3), ("layer4", m.layer4)]
        layers_d = [("layer1_depth", m.layer1_depth), ("layer2_depth", m.layer2_depth), ("layer3_depth", m.layer3_depth),
                    ("layer4_depth", m.layer4_depth)]
        # size of the fused 640-channel buffer: spatial size after three stride-2 blocks
        hh, ww = a.H, a.W
        for _ in range(3):
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        c_rgb = m.layer4[1].conv2.weight.shape[0]
        c_dep = m.layer4_depth[1].conv2.weight.shape[0]
        self.cat = self.act(N, hh, ww, c_rgb + c_dep)
        x, xd = a, d_
        for li in range(4):                      # emission interleaved per stage (host order only matters in eager mode)
            lname, layer = layers_rgb[li]
            for bi, blk in enumerate(layer):
                last = li == 3 and bi == len(layer) - 1
                x, ctx = self._block("%s.%d" % (lname, bi), blk, x, out=self.cat.chan(0, c_rgb) if last else None)
                self.blocks_rgb.append(ctx)
            lname, layer = layers_d[li]
            with self.on(1):
                for bi, blk in enumerate(layer):
                    last = li == 3 and bi == len(layer) - 1
                    xd, ctx = self._block("%s.%d" % (lname, bi), blk, xd, out=self.cat.chan(c_rgb, c_dep) if last else None)
                    self.blocks_d.append(ctx)
        self.probe(self.fwd, "fwd_rgb_encoder_end", 0)
        self.probe(self.fwd, "fwd_depth_encoder_end", 1)
        self.edge(self.fwd, "join_depth", 1, 0)
        # fusion 1x1 convs (no activation, models.py:652-657)
        if not self.train:
            yf = self.conv_bn_eval("conv_fusion", self.cat, [(m.conv_fusion.weight, 0, m.bn_fusion)], 1, 1, 0, ACT_NONE)
            z = self.conv_bn_eval("conv2", yf, [(m.conv2.weight, 0, m.bn2)], 1, 1, 0, ACT_NONE)
            self.taps["bn_fusion"], self.taps["bn2"] = yf, z
            for i, mod in enumerate((m.decoder.layer1, m.decoder.layer2, m.decoder.layer3, m.decoder.layer4), 1):
                z, _ = self._upproj("decoder.layer%d" % i, mod, z)
            self._head(z)
            self._finish_pack_jobs()
            return
        rf, self.c_fus = self.conv_fwd("conv_fusion", self.cat, [(m.conv_fusion.weight, 0)], 1, 1, 0)
        self.co_fus = self.bn_coeffs("bn_fusion", m.bn_fusion, self.c_fus["stat"], self.c_fus["tiles"], rf.C, 0, rf.M)
        self.yf = self.bn_act("bn_fusion", rf, self.co_fus, ACT_NONE)
        r2, self.c_c2 = self.conv_fwd("conv2", self.yf, [(m.conv2.weight, 0)], 1, 1, 0)
        self.co_c2 = self.bn_coeffs("bn2", m.bn2, self.c_c2["stat"], self.c_c2["tiles"], r2.C, 0, r2.M)
        z = self.bn_act("bn2", r2, self.co_c2, ACT_NONE)
        self.rf, self.r2 = rf, r2
        # decoder
        self.ups = []
        for i, mod in enumerate((m.decoder.layer1, m.decoder.layer2, m.decoder.layer3, m.decoder.layer4), 1):
            z, ctx = self._upproj("decoder.layer%d" % i, mod, z)
            self.ups.append(ctx)
        self._head(z)
        self._finish_pack_jobs()
        if self.train:
            self._build_backward()

    def _head(self, z):
        m, N = self.m, self.N
        self.z = z
        self.dmap = self.buf(N, z.H, z.W)
        self.op(self.fwd, "conv3", self.L.rd_head_conv_fwd_t, self.dt, z.ptr, z.ld, _p(m.conv3.weight), N, z.H, z.W, z.C, _p(self.dmap), self.stream)
        self.pred = self.buf(N, 1, self.Ho, self.Wo)
        self.op(self.fwd, "bilinear", self.L.rd_bilinear_fwd, _p(self.dmap), N, z.H, z.W, _p(self.pred), self.Ho, self.Wo, self.stream)

    def _finish_pack_jobs(self):
        """Upload the job table of rd_pack_weights_batched: one launch refreshes every packed weight copy."""
        import numpy as np

        class Job(C.Structure):
            _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("scale", C.c_void_p), ("O", C.c_int32), ("I", C.c_int32), ("T", C.c_int32),
                        ("ldc", C.c_int32), ("off", C.c_int32), ("rows_total", C.c_int32), ("transpose", C.c_int32),
                        ("first_block", C.c_int32), ("quad", C.c_int32), ("pad_", C.c_int32)]
        if self.evalcoef_jobs:
            class EJob(C.Structure):
                _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("rm", C.c_void_p), ("rv", C.c_void_p), ("scale", C.c_void_p),
                            ("shift", C.c_void_p), ("C", C.c_int32), ("pad_", C.c_int32)]
            ej = (EJob * len(self.evalcoef_jobs))()
            for k, (o, bn, sc, sh) in enumerate(self.evalcoef_jobs):
                ej[k] = EJob(bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), sc.value,
                             sh.value, o, 0)
            self.evalcoef_table = torch.from_numpy(np.frombuffer(bytes(ej), dtype=np.uint8).copy()).to(self.dev)
            self.keep.append(self.evalcoef_table)
            self.op(self.prep, "evalcoef_all", self.L.rd_bn_eval_coeffs_batched, _p(self.evalcoef_table), len(self.evalcoef_jobs),
                    C.c_float(BN_EPS), self.streams[0])
        chunk = self.L.rd_pack_chunk()

        def emit(name, job_list):
            if not job_list:          # (a module whose every convolution is a Winograd layer: nothing for rd_pack_weights_batched)
                return None, None
            jobs = (Job * len(job_list))()
            block_job, nb = [], 0
            for k, (src, dst, o, i, t, ldc, off, rows, tr, scale, quad) in enumerate(job_list):
                assert o * i * t < 2 ** 31, "pack_weights_batched indexes one weight tensor with 32-bit arithmetic"
                n = -(-(o * i * t) // chunk)
                jobs[k] = Job(src.data_ptr(), dst.data_ptr(), scale.data_ptr() if scale is not None else None, o, i, t, ldc, off, rows, tr, nb, quad, 0)
                block_job += [k] * n
                nb += n
            raw = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
            table = torch.from_numpy(raw).to(self.dev)
            blocks = torch.tensor(block_job, dtype=torch.int32, device=self.dev)
            self.keep += [table, blocks]
            self.op(self.prep, name, self.L.rd_pack_weights_batched, _p(table), _p(blocks), nb, self.stream)
            return table, blocks

        def emit_wino():
            if not self.wino_jobs:
                return

            class WJob(C.Structure):
                _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("O", C.c_int32), ("I", C.c_int32), ("flip", C.c_int32), ("first_block", C.c_int32)]
            wj = (WJob * len(self.wino_jobs))()
            block_job, nb = [], 0
            for q, (w, u, o, i, flip) in enumerate(self.wino_jobs):
                n = self.L.rd_wino_pack_blocks(o, i, flip)
                wj[q] = WJob(w.data_ptr(), u.data_ptr(), o, i, flip, nb)
                block_job += [q] * n
                nb += n
            table = torch.from_numpy(np.frombuffer(bytes(wj), dtype=np.uint8).copy()).to(self.dev)
            blocks = torch.tensor(block_job, dtype=torch.int32, device=self.dev)
            self.keep += [table, blocks]
            self.op(self.prep, "pack_wino", self.L.rd_wino_pack_batched, _p(table), _p(blocks), nb, self.stream)

        stems = [j for j in self.pack_jobs if j[4] == 49 and j[10] == 0]          # (7x7 taps, plain layout: the two stem convolutions)
        rest = [j for j in self.pack_jobs if not (j[4] == 49 and j[10] == 0)]
        if getattr(self, "pack_overlap", False) and stems and rest:
            emit("pack_stems", stems)
            self.edge(self.prep, "pack_fork", 0, 2)          # (behind the previous step's SGD update, which ends on stream 0)
            with self.on(2):
                self.pack_table, self.pack_blocks = emit("pack_all", rest)
                emit_wino()
        else:
            self.pack_table, self.pack_blocks = emit("pack_all", self.pack_jobs)
            emit_wino()

    def _build_backward(self):
        """Backward as four bucket-aligned segments (self.bwd_segments): every segment ends with all streams joined, so
        it can be captured as its own hipGraph and its gradient bucket all-reduced while the next segment runs:
          0: head, decoder, bn2/conv2, bn_fusion/conv_fusion      1: layer4 || layer4_depth
          2: layer3 || layer3_depth                               3: layer2, layer1, stem || their depth counterparts
        Inside a segment the RGB chain runs on stream 0, the depth chain on stream 1 and every weight-gradient chain of
        the main stream on stream 2."""
        m, N = self.m, self.N
        z = self.z
        self.dpred = self.buf(N, 1, self.Ho, self.Wo)
        self.dx_dense = None
        self.bwd_segments = []

        def end_segment(prefixes, last=False):
            evs = []
            self._flush_reduces()
            if last:
                for q in (0, 1, 2):
                    self.probe(self.bwd, "bwd_end_stream%d" % q, q)
            if last or self.segment_joins:
                self.edge(self.bwd, "join1", 1, 0)
                self.edge(self.bwd, "join2", 2, 0)
            elif self.multi_stream:
                for q in (1, 2):
                    if (self.stream_mask >> (q - 1)) & 1:
                        ev = self._event()
                        self.op(self.bwd, "segment_end%d.record" % q, self.L.rd_event_record, ev, self.streams[q])
                        evs.append(ev)
            self.segment_events.append(evs)
            begin = self.bwd_segments[-1][1] if self.bwd_segments else 0
            self.bwd_segments.append((begin, len(self.bwd), prefixes))

        ddm = self.buf(N, z.H, z.W)
        self.op(self.bwd, "bilinear.bwd", self.L.rd_bilinear_bwd, _p(self.dpred), N, self.Ho, self.Wo, _p(ddm), z.H, z.W, self.stream)
        dz = self.act(N, z.H, z.W, z.C)
        ws = self.buf(int(self.L.rd_head_conv_bwd_workspace_floats(N, z.H, z.W, z.C)))
        # input gradient on the main chain, weight gradient (+ slab reduction) on the weight-gradient stream beside the decoder's backward
        self.op(self.bwd, "conv3.dgrad", self.L.rd_head_conv_dgrad_t, self.dt, _p(m.conv3.weight), _p(ddm), N, z.H, z.W, z.C, dz.ptr, dz.ld,
                self.stream)
        fork = os.environ.get("RD_HEAD_WGRAD_FORK", "1") == "1"
        if fork:
            self.edge(self.bwd, "conv3.fork_wgrad", 0, 2)
        with self.on(2 if fork else 0):
            self.op(self.bwd, "conv3.wgrad", self.L.rd_head_conv_wgrad_t, self.dt, z.ptr, z.ld, _p(ddm), N, z.H, z.W, z.C,
                    _p(self.grad_of(m.conv3.weight)), _p(ws), self.stream)
        for ctx in reversed(self.ups):
            dz = self._upproj_bwd(ctx, dz)
        dr2, _ = self.bn_join_bwd("bn2", dz, None, ACT_NONE, self.r2, self.co_c2)
        dyf = self.conv_bwd(self.c_c2, dr2)
        drf, _ = self.bn_join_bwd("bn_fusion", dyf, None, ACT_NONE, self.rf, self.co_fus)
        dcat = self.act(N, self.cat.H, self.cat.W, self.cat.C)
        self.conv_bwd(self.c_fus, drf, dx=dcat)
        end_segment(("conv_fusion", "bn_fusion", "conv2", "bn2", "decoder", "conv3"))

        self.probe(self.bwd, "bwd_encoders_begin", 0)
        c_rgb = self.blocks_rgb[-1]["y"].C
        g = dcat.chan(0, c_rgb)
        gd = dcat.chan(c_rgb, dcat.C - c_rgb)
        dense = None
        if self.depth_planes is not None and len(self.depth_planes) == 2:
            self.dx_dense = self.dense_grad_dst if self.dense_grad_dst is not None else self.buf(N, self.H, self.W)
            dense = (1, self.dx_dense)
        # blocks come in pairs per ResNet stage: indices (6,7)=layer4, (4,5)=layer3, (2,3)=layer2, (0,1)=layer1
        # The depth chain needs the main stream once: for dcat (the fusion layer's input gradient).  Where the segments end in joins
        # (hipGraph capture, torch.distributed's all_reduce) every segment forks it again; without joins a second fork would only make
        # depth layer3 wait for RGB layer4 and depth layer2 / layer1 / stem for RGB layer3 -- the depth chain then trails the RGB chain
        # at the end of the step with the chip to itself (RD_DEPTH_FORK_ONCE=0 restores the per-stage forks).
        fork_once = not self.segment_joins and os.environ.get("RD_DEPTH_FORK_ONCE", "1") == "1"
        for stage in (3, 2):
            if stage == 3 or not fork_once:
                self.edge(self.bwd, "fork_depth", 0, 1)
            for ctx in reversed(self.blocks_rgb[2 * stage:2 * stage + 2]):
                g = self._block_bwd(ctx, g)
            with self.on(1):
                for ctx in reversed(self.blocks_d[2 * stage:2 * stage + 2]):
                    gd = self._block_bwd(ctx, gd)
            end_segment(("layer%d" % (stage + 1), "layer%d_depth" % (stage + 1)))
        self.probe(self.bwd, "bwd_rgb_layer3_end", 0)
        self.probe(self.bwd, "bwd_depth_layer3_end", 1)
        if not fork_once:
            self.edge(self.bwd, "fork_depth", 0, 1)
        for ctx in reversed(self.blocks_rgb[0:4]):
            g = self._block_bwd(ctx, g)
        # (Making the depth chain's last blocks wait for the RGB chain to reach its stem -- to run them beside the stem's two HBM-bound
        #  passes instead of beside the MFMA-bound layers -- costs 1-3.5 %: that chain is ~0.8 ms of dependent small launches and then
        #  trails the step.  profiles/r05_stem_tail.txt)
        self._stem_bwd(self.c_stem_rgb, g)
        with self.on(1):
            for ctx in reversed(self.blocks_d[0:4]):
                gd = self._block_bwd(ctx, gd)
            self._stem_bwd(self.c_stem_d, gd, dgrad_channel=dense)
        end_segment(("conv1", "bn1", "layer1", "layer2", "conv1_depth", "bn1_depth", "layer1_depth", "layer2_depth"), last=True)

    # ------------------------------------------------------------------ execution
    def _diagnostic_loop(self):
        """RD_POISON_LDS / RD_TRACE_OPS need a host hook between ops: they keep the Python loop (RD_PY_LOOP=1 forces it, to
        measure what the op table saves: tools/host_time.py)."""
        return ((os.environ.get("RD_POISON_LDS") == "1" and not self.multi_stream) or os.environ.get("RD_TRACE_OPS") == "1"
                or os.environ.get("RD_PY_LOOP") == "1")

    def run_list(self, key, begin=0, end=None):
        """Issue ops [begin, end) of self.prep / self.fwd / self.bwd (key = "prep" | "fwd" | "bwd") with ONE C-ABI call: the three
        lists are marshalled once into an op table (optable.py, rd_optable_run)."""
        lst = getattr(self, key)
        end = len(lst) if end is None else end
        if self.dry_run:
            raise RuntimeError("dry-run plans cannot execute: the HIP path has no CPU fallback")
        if self._diagnostic_loop():
            return self._run(lst[begin:end])
        if self._optable is None:
            from .optable import OpTable
            self._table_base, ops = {}, []
            for k in ("prep", "fwd", "bwd"):
                self._table_base[k] = (len(ops), len(getattr(self, k)))
                ops += getattr(self, k)
            self._optable = OpTable(self.L, ops, self.streams)
        base, n = self._table_base[key]
        assert n == len(lst), "the plan's op lists changed after its op table was built"
        self._optable.run(base + begin, base + end)

    def _run(self, ops):
        if self.dry_run:
            raise RuntimeError("dry-run plans cannot execute: the HIP path has no CPU fallback")
        # diagnostics: RD_POISON_LDS=1 (with RD_SINGLE_STREAM=1) NaN-fills every CU's LDS before each op, so that a kernel
        # consuming LDS it never wrote shows up as NaN in the parity tests instead of depending on its predecessor's leftovers
        poison = os.environ.get("RD_POISON_LDS") == "1" and not self.multi_stream
        trace_ops = os.environ.get("RD_TRACE_OPS") == "1"       # diagnostics: name every op and synchronise behind it
        for name, fn, args in ops:
            if poison:
                self.L.rd_debug_poison_lds(self.streams[0])
            if trace_ops:
                print("[op]", name, getattr(fn, "__name__", fn), flush=True)
            rc = fn(*args)
            if rc != 0:
                check(rc, name)
            if trace_ops:
                torch.cuda.synchronize()

    def set_stream(self, serialize=False):
        """Bind stream 0 to torch's current stream and streams 1/2 to the plan's side streams (serialize=True binds all
        three to the current stream: used by the instrumented timing pass of bench.py)."""
        cur = torch.cuda.current_stream().cuda_stream
        self.streams[0].value = cur
        if serialize or not self.multi_stream:
            self.streams[1].value = self.streams[2].value = cur
            return
        if self._side is None:
            # (default priorities: raising a stream's priority measured 17-29 % slower in round 1; RD_SIDE_PRIO="p1,p2" for experiments)
            pr = [int(v) for v in os.environ.get("RD_SIDE_PRIO", "0,0").split(",")]
            self._side = [torch.cuda.Stream(device=self.dev, priority=pr[0]), torch.cuda.Stream(device=self.dev, priority=pr[1])]
        self.streams[1].value = self._side[0].cuda_stream
        self.streams[2].value = self._side[1].cuda_stream

    def bind_input(self, base_ptr, channels):
        """Point the stems' input planes at an fp32 [N,channels,H,W] tensor at base_ptr (contiguous; channels >= the planes the network reads).
        bind_input(self.x_in.data_ptr(), self.x_in.shape[1]) restores the plan's own buffer."""
        hw = self.H * self.W
        for pl, st, c0, n in self.x_bind:
            for c in range(n):
                pl[c] = base_ptr + 4 * hw * (c0 + c)
                st[c] = channels * hw
        self._x_bound = base_ptr

    def bind_own_input(self):
        """The plan's own entry points (run_forward and the stand-alone module plans) read x_in: undo a caller's bind_input (a fused step and
        the eager forward may share one cached plan)."""
        if getattr(self, "_x_bound", self.x_in.data_ptr()) != self.x_in.data_ptr():
            self.bind_input(self.x_in.data_ptr(), self.x_in.shape[1])

    def run_forward(self, x=None):
        """x: [N,>=4,H,W] fp32 CUDA tensor (copied into the plan's static input buffer) or None if already there."""
        self.set_stream()
        self.generation += 1
        self.bind_own_input()
        if x is not None and self.x_source is None:
            if tuple(x.shape[:1]) + tuple(x.shape[2:]) != (self.N, self.H, self.W):
                raise ValueError("plan built for [%d,*,%d,%d], got %s" % (self.N, self.H, self.W, tuple(x.shape)))
            self.x_in.copy_(x[:, :self.x_in.shape[1]])
        self.run_list("prep")
        self.run_list("fwd")
        return self.pred

    def run_backward(self, dpred=None):
        self.set_stream()
        if dpred is not None:
            self.dpred.copy_(dpred)
        self.run_list("bwd")


class ModulePlan(LateFusionPlan):
    """ONE BasicBlock (models.py:75-112), UpProjModule (models.py:181-209) or whole UpProj decoder (models.py:210-216: four modules in a
    row) run through the very op builders the network plan uses (_block / _upproj and their backward), on a stand-alone NHWC input.
    It backs the stand-alone `forward` of those modules (radar_depth_amd/model/models.py: the reference's sub-modules are callable on
    their own) and the layer-level parity tests against the reference-generated fixtures (tests/golden/upproj_module.npz,
    basic_block.npz).  owner: the ArenaOwner nn.Module whose gradient arena holds `mod`'s parameters (the module itself when it is
    called stand-alone).  train=False: eval-mode forward only (BatchNorm folded into the convolutions from the running statistics)."""

    def __init__(self, owner, mod, kind, batch, height, width, cin, bf16=False, train=True, split=False):
        assert kind in ("block", "upproj", "decoder")
        self._mod, self._kind, self._cin = mod, kind, cin
        owner._ensure_arenas()      # parameters move into the flat arena BEFORE any op captures their addresses
        super().__init__(owner, batch, height, width, train=train, bf16=bf16, split=split)

    def _build(self):
        self.x = self.act(self.N, self.H, self.W, self._cin)
        if self._kind == "decoder":
            mods = [("layer%d" % i, getattr(self._mod, "layer%d" % i)) for i in (1, 2, 3, 4)]
            build, back = self._upproj, self._upproj_bwd
        else:
            mods = [("m", self._mod)]
            build, back = (self._upproj, self._upproj_bwd) if self._kind == "upproj" else (self._block, self._block_bwd)
        z, ctxs = self.x, []
        for name, mod in mods:
            z, ctx = build(name, mod, z)
            ctxs.append(ctx)
        self.y = z
        self._finish_pack_jobs()
        if not self.train:
            return
        self.dy = self.act(self.y.N, self.y.H, self.y.W, self.y.C)
        g = self.dy
        for ctx in reversed(ctxs):
            g = back(ctx, g)
        self.dx = g
        self._flush_reduces()
        self.edge(self.bwd, "join1", 1, 0)
        self.edge(self.bwd, "join2", 2, 0)

    def forward_nchw(self, x_nchw):
        """x [N,C,H,W] CUDA fp32 -> y [N,C',H',W'] (a fresh NCHW tensor); no host synchronisation."""
        if tuple(x_nchw.shape) != (self.N, self._cin, self.H, self.W):
            raise ValueError("plan built for [%d,%d,%d,%d], got %s" % (self.N, self._cin, self.H, self.W, tuple(x_nchw.shape)))
        self.set_stream()
        self.generation += 1
        self.x.t.copy_(x_nchw.permute(0, 2, 3, 1))
        self.run_list("prep")
        self.run_list("fwd")
        return self.y.view().permute(0, 3, 1, 2).contiguous()

    def backward_nchw(self, dy_nchw):
        """dy [N,C',H',W'] -> dx [N,C,H,W]; parameter gradients land in the owner's gradient arena."""
        self.set_stream()
        self.dy.t.copy_(dy_nchw.permute(0, 2, 3, 1))
        self.run_list("bwd")
        return self.dx.view().permute(0, 3, 1, 2).contiguous()

    def run(self, x_nchw, dy_nchw):
        """x [N,C,H,W], dy [N,C',H',W'] CUDA fp32 -> (y, dx) as NCHW tensors; parameter gradients land in the owner's arena."""
        y = self.forward_nchw(x_nchw)
        dx = self.backward_nchw(dy_nchw)
        torch.cuda.synchronize()
        return y, dx
