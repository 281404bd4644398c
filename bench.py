#!/usr/bin/env python3
"""Headline benchmark: training samples/sec of resnet18_latefusion (upproj, rgbd), b=16 per GPU, 450x800, fp32,
full step = forward + MaskedL1 + zero_grad + backward + SGD(momentum .9, wd 1e-4)  (reference main.py:400-447).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; inputs are resident in HBM before the timed region (synthetic batch, SURVEY.md 8d recipe,
seed = 1234 + 1000*rank).  Timed region: K steps bracketed by barrier + synchronize, max over ranks.  Rank 0
prints ONE JSON line.  Extra objects:
  roofline     -- the dominant kernel (largest total time among the conv kernel instantiations) measured with HIP
                  events in an instrumented eager pass on the step's own stream: algorithmic FLOPs per launch /
                  average launch duration.  The default plan ("split", config.arith) computes every fp32 product from six
                  bf16 MFMA terms: `frac` prices the bf16 MFMA FLOPs it issues against the dense bf16 peak (2500 TFLOP/s;
                  equivalently algorithmic fp32 FLOPs against 2500/6), and `frac_of_fp32_mfma_peak` the algorithmic
                  rate against the fp32 MFMA/vector peak (157.3 TFLOP/s); `--operands fp32` runs every
                  convolution on the fp32 MFMA and is priced against 157.3 only;
  alt_fp32_mfma-- the same workload on the plain fp32-MFMA plan, timed the same way in the same process;
  cpu_baseline -- the CPU oracle (plain PyTorch restatement of the reference) timed on the host cores of this
                  box on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector == f32-MFMA peak (guides/MI355X_MICROARCH.md)
METRIC = "training samples/sec, resnet18_latefusion b=16 450x800 rgbd, 1/2/4/8 MI355X"


def desc_flops(d):
    f = 0
    for i in range(d.n_phases):
        p = d.phase[i]
        f += p.lh * p.lw * p.n_taps
    return 2.0 * f * d.N * d.Cin * d.Cout


def desc_bytes(kind, d, esz):
    """Algorithmic HBM bytes of one launch: every operand tensor read once, the result written once (weights and halos ignored,
    SURVEY.md 8d / appendix B).  Convolution: non-zero input + output; weight gradient: input + output gradient."""
    return float(esz) * d.N * (d.Hi * d.Wi * d.Cin + d.Ho * d.Wo * d.Cout)


def kernel_identity(L, kind, d, io16=False):
    """Kernel name as rocprofv3 prints it (spaces removed), from the library's own plan for the descriptor."""
    tb_ = lambda v: "true" if v else "false"
    if kind == "gconv_bf16":
        info = (C.c_int32 * 8)()
        L.rd_gconv_bf16_plan_info(C.byref(d), info)           # MT, NT, pipe*1000 + CKP, ...
        return "gconv_bf16_kernel<%d,%d,%s,%s>" % (info[0], info[1], tb_(info[2] >= 1000), tb_(io16))
    if kind == "conv16_split":
        return "conv16_split_kernel<stat|add>"
    if kind == "wino":
        return "wino_split_kernel<false>"
    if kind == "gconv_split":
        info = (C.c_int32 * 8)()
        L.rd_gconv_split_plan_info(C.byref(d), info)          # MT, NT, TH, TW, PP, lds, workgroups, tap groups + 100 * double-buffered patch
        return "gconv_split_kernel<%d,%d,%s,false>" % (info[0], info[1], tb_(info[7] >= 100))
    if kind == "gconv_split_pre":
        info = (C.c_int32 * 8)()
        L.rd_gconv_split_pre_plan_info(C.byref(d), info)
        return "gconv_sp2_kernel<%d,%d,0>" % (info[0], info[1])
    if kind in ("wgrad_split", "wgrad_split_pre"):
        info = (C.c_int32 * 4)()
        L.rd_wgrad_split_plan_info(C.byref(d), info)           # splits, tiles per split, workgroups, tiles | tall geometry << 30
        geo = "WsGeo<4,16>" if info[3] >> 30 else "WsGeo<2,32>"
        pre = tb_(kind == "wgrad_split_pre")
        if d.n_phases == 1:                                    # 3x3: one launch per op -- the name rocprofv3 prints
            return "wgrad_split_kernel<3,3,%s,%s>" % (geo, pre)
        return "wgrad_split_kernel<3,3|3,2|2,3|2,2,%s,%s> (4 UpProj phase launches)" % (geo, pre)
    if kind == "wgrad_bf16":
        info = (C.c_int32 * 8)()
        L.rd_wgrad_bf16_plan_info(C.byref(d), info)           # cpi, cpo, ...
        full = d.n_phases == 1 and d.in_stride == 1 and d.phase[0].n_taps == 9
        return "wgrad_bf16_kernel<%s,%d,%s>" % (tb_(full), 4 // (info[0] * info[1]), tb_(io16))
    if kind in ("gconv", "gconv_bnb"):
        info = (C.c_int32 * 10)()
        L.rd_gconv_plan_info(C.byref(d), info)                 # info[4] = grouped*1000000 + pipelined*10000 + ksplit*100 + CKW
        grouped = info[4] >= 1000000                          # in_stride == 2 run as input-parity groups: SWZ = false instantiation
        if info[0] == 0:                                      # 16 -> 16 channel 3x3 layers: csrc/conv16.hip (16x16x4 MFMA)
            return "conv16_kernel<stat|add>"
        # template parameters: MT, NT, WM, WN, CKW, SWZ, PIPE, GRP, BNB (BNB: the dgrad launches that also emit BatchNorm-backward sums)
        return "gconv_kernel<%d,%d,%d,%d,%d,%s,%s,%s,%s>" % (info[0], info[1], info[2], info[3], info[4] % 100,
                                                             "true" if (d.in_stride == 2 and not grouped) else "false",
                                                             "true" if info[4] % 1000000 >= 10000 else "false", "true" if grouped else "false",
                                                             "true" if kind == "gconv_bnb" else "false")
    w = (C.c_int32 * 9)()
    L.rd_wgrad_plan_info(C.byref(d), w)
    if w[0] == 0:      # 16 -> 16 channel 3x3 layers: csrc/wgrad16.hip (16x16x4 MFMA)
        return "wgrad16_kernel"
    if w[2] == -1:     # 1x1 layers with >= 64 channels each side: csrc/wgrad1x1.hip (pixel-reduction GEMM, 64x64 tiles per wave)
        return "wgrad1x1_kernel<%d,%d>" % (2 if d.Cin >= 128 else 1, 2 if d.Cout >= 128 else 1)
    tb = lambda v: "true" if v else "false"
    if w[8]:       # column-strip kernel (one per UpProj phase when w[6])
        return "wgrad_strip_kernel<%s>%s" % ("3,3|2,3|3,2|2,2" if w[6] else "%d,%d" % (w[0] // w[5], w[5]), " (4 UpProj phase launches)" if w[6] else "")
    if w[6]:       # UpProj: four launches (9/6/6/4-tap sub-stencils) inside one op
        return "wgrad_kernel<9|6|6|4,%d,%s,true,%d,3|2> (4 UpProj phase launches)" % (w[1], tb(w[2]), w[4])
    return "wgrad_kernel<%d,%d,%s,%s,%d,%d>" % (w[0], w[1], tb(w[2]), tb(w[3]), w[4], 3 if w[4] == 0 else w[5])


def instrumented_pass(ts):
    """One eager step with every op bracketed by events on the step's stream (all plan streams bound to it).
    Returns {kernel: [total_ms, launches, total_flops, total_bytes]} and the per-step op breakdown by family.  Walks the step's own
    marshalled op list (HipTrainStep._ops: both stages of the multistage net, the losses and the SGD update included)."""
    L = ts.L
    if ts._ops is None:
        ts._build_table()
    meta = {}
    for plan in ts.plans:
        for name, _, a in plan.prep + plan.fwd + plan.bwd:
            if name in plan.meta:
                meta[id(a)] = plan.meta[name]
    io16 = ts.plan.storage == "bf16"
    agg, fam = {}, {}
    with torch.cuda.stream(ts.side):
        for plan in ts.plans:
            plan.set_stream(serialize=True)      # all plan streams -> this stream, so per-op event pairs bracket the op
        recs = []
        # three un-instrumented passes first, back to back with the measured one: the device is then at its sustained
        # (power-managed) clock like in the timed region, not at the boost clock it reaches after an idle gap
        for _ in range(3):
            for name, fn, args in ts._ops:
                assert fn(*args) == 0, name
        for name, fn, args in ts._ops:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            assert rc == 0, name
            recs.append((name, id(args), e0, e1))
        torch.cuda.synchronize()
        for name, key, e0, e1 in recs:
            ms = e0.elapsed_time(e1)
            tag = name.rsplit(".", 1)[-1]
            if tag in ("record", "wait"):
                continue
            family = {"wgrad": "wgrad", "dgrad": "dgrad", "wreduce": "wgrad_reduce", "wreduce_all": "wgrad_reduce", "pack_all": "pack"}.get(tag)
            if family is None:
                family = "conv_fwd" if key in meta else "bn/act/pool/head/loss/sgd"
            fam[family] = fam.get(family, 0.0) + ms
            if key in meta:
                kind, d = meta[key]
                k = kernel_identity(L, kind, d, io16)
                a = agg.setdefault(k, [0.0, 0, 0.0, 0.0])
                a[0] += ms
                a[1] += 1
                a[2] += desc_flops(d)
                a[3] += desc_bytes(kind, d, 2 if io16 else 4)
    return agg, fam


def cpu_baseline(arch, batch, height, width):
    """Oracle (CPU restatement of the reference, pinned to it by golden vectors) full SGD steps on this box's host cores, per
    SURVEY.md 8(d): the configuration's own shape (and, for the headline, b=2 as well), warm-up + timed steps, best and median;
    the thread count is chosen by a quick sweep at b=2 (oversubscribing the host makes oneDNN slower, not faster).  The sample
    is bounded to about 30 s of CPU work: the number of timed steps shrinks with the step time."""
    import statistics
    import types

    from oracle import train as otrain
    from radar_depth_amd.synthetic import make_batch
    torch.manual_seed(0)
    made = otrain.create_model(types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False), [height, width])
    m, lw = made if isinstance(made, tuple) else (made, None)
    m.train()
    opt = torch.optim.SGD(m.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    crit = otrain.make_criterion(arch)

    def steps(x, t, n):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            otrain.train_step(arch, m, crit, opt, x, t, lw)
            out.append(time.perf_counter() - t0)
        return out
    ncpu = os.cpu_count() or 1
    big = height * width > 450 * 800
    x2, t2 = make_batch(1 if big else 2, height, width, 1234)
    steps(x2, t2, 1)                                        # oneDNN primitive creation
    sweep = {}
    for nt in sorted({max(1, ncpu // 32), max(1, ncpu // 16), max(1, ncpu // 8), max(1, ncpu // 4)}):
        torch.set_num_threads(nt)
        sweep[nt] = min(steps(x2, t2, 1 if big else 2))
    nt = min(sweep, key=sweep.get)
    torch.set_num_threads(nt)
    res, plan_txt = {}, []
    for b in sorted({x2.shape[0], batch}):
        x, t = (x2, t2) if b == x2.shape[0] else make_batch(b, height, width, 1234)
        first = steps(x, t, 1)[0]
        n_warm = 1 if first > 6 else 2                        # 2 warm-up + 5 timed steps (fewer only when a step takes > 5 s)
        n_timed = max(1, min(5, int(36.0 / max(first, 1e-3)) - n_warm))
        tt = steps(x, t, n_warm - 1 + n_timed)[n_warm - 1:]
        res[b] = (b / min(tt), b / statistics.median(tt), min(tt))
        plan_txt.append("b=%d: %d warm-up + %d timed steps" % (b, n_warm, n_timed))
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        cpu = "unknown"
    bs = x2.shape[0]
    return {"value": round(res[batch][0], 3), "median": round(res[batch][1], 3), "unit": "samples/s", "cores": nt, "kind": "port",
            "b%d_value" % bs: round(res[bs][0], 3), "b%d_median" % bs: round(res[bs][1], 3),
            "thread_sweep_b%d_step_s" % bs: {str(k): round(v, 3) for k, v in sweep.items()}, "cpu": cpu, "host_cpus": ncpu,
            "sample": "oracle (PyTorch CPU restatement pinned to the reference by golden vectors), %s full SGD step, %dx%d fp32 "
                      "(the reference has no reduced-precision path: the fp32 CPU step is the baseline of every configuration): %s, "
                      "best (value) and median; best b=%d step %.2f s; %d threads (fastest of the sweep) on %d host cpus (%s)"
                      % (arch, height, width, "; ".join(plan_txt), batch, res[batch][2], nt, ncpu, cpu)}


def eager_loop(model, x, t, warmup, steps, lr=0.01, momentum=0.9, weight_decay=1e-4):
    """EXACTLY the reference's loop body (main.py:440-445) through the drop-in surface -- `pred = model(input); loss = criterion(pred,
    target); optimizer.zero_grad(); loss.backward(); optimizer.step()` with torch.optim.SGD -- timed like the fused step.  This is what
    a maintainer gets who only swaps the three imports of INTEGRATION.md section 1."""
    from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss
    model.train()
    crit = MaskedL1Loss()
    opt = torch.optim.SGD(model.parameters(), lr, momentum=momentum, weight_decay=weight_decay)
    loss = None

    def one():
        pred = model(x)
        loss = crit(pred, t)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        loss = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        loss = one()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, float(loss.item()), host


class PowerReader:
    """Average socket power over a window, from whatever the box exposes (amdsmi python binding, hwmon sysfs, rocm-smi); None when
    nothing is readable.  Sampled from a thread during the clock pass -- never inside the timed region."""

    def __init__(self, device_index=0):
        import glob
        self.kind, self._h, self._path = None, None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self._amdsmi, self._h = amdsmi, hs[min(device_index, len(hs) - 1)]
            if self._read_amdsmi() is not None:
                self.kind = "amdsmi"
        except Exception:                                   # noqa: BLE001
            self._h = None
        if self.kind is None:
            for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
                for f in sorted(glob.glob(pat)):
                    try:
                        if float(open(f).read()) > 0:
                            self.kind, self._path = "sysfs:" + f, f
                            break
                    except (OSError, ValueError):
                        pass
                if self.kind:
                    break
        self.samples = []
        self._stop = None

    def _read_amdsmi(self):
        try:
            info = self._amdsmi.amdsmi_get_power_info(self._h)
            for key in ("current_socket_power", "average_socket_power", "socket_power"):
                v = info.get(key)
                if isinstance(v, (int, float)) and v > 0:
                    return float(v)
        except Exception:                                   # noqa: BLE001
            pass
        return None

    def read(self):
        if self.kind == "amdsmi":
            return self._read_amdsmi()
        if self._path:
            try:
                return float(open(self._path).read()) / 1e6      # microwatts
            except (OSError, ValueError):
                return None
        return None

    def start(self, period=0.01):
        import threading
        if self.kind is None:
            return
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                v = self.read()
                if v is not None:
                    self.samples.append(v)
                self._stop.wait(period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._stop is not None:
            self._stop.set()
            self._thr.join()
        if not self.samples:
            return None
        return {"source": self.kind, "avg_w": round(sum(self.samples) / len(self.samples), 1), "max_w": round(max(self.samples), 1), "samples": len(self.samples)}


def clock_pass(L, step_fn, steps, ms_per_step, main_stream):
    """Effective shader clock WHILE the steps execute: rd_clock_probe windows of 1 ms back to back on a stream of
    their own for the duration of `steps` more steps issued right behind them (an extra pass OUTSIDE the timed region), eight one-wave
    workgroups each (one per XCD); two marker probes on the steps' stream delimit the steps on the device's real-time counter, only
    windows inside them are kept.  Returns MHz statistics and the average socket power over the same pass."""
    nblk, win_us = 8, 1000
    n_win = int(steps * ms_per_step * 1.25) + 4
    buf = torch.zeros((n_win + 2) * nblk * 4, dtype=torch.int64, device="cuda")
    base = buf.data_ptr()
    probe = torch.cuda.Stream()
    ps = C.c_void_p(probe.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(3):
        step_fn()                                            # the device is at its in-step clock when the windows start
    power = PowerReader(torch.cuda.current_device())
    power.start()
    assert L.rd_clock_probe(C.c_void_p(base + n_win * nblk * 32), 1, 1, main_stream()) == 0            # start marker (behind the 3 steps)
    probe.wait_stream(torch.cuda.current_stream())
    for k in range(n_win):
        assert L.rd_clock_probe(C.c_void_p(base + k * nblk * 32), nblk, win_us, ps) == 0
    for _ in range(steps):
        step_fn()
    assert L.rd_clock_probe(C.c_void_p(base + (n_win + 1) * nblk * 32), 1, 1, main_stream()) == 0      # end marker
    torch.cuda.synchronize()
    pw = power.stop()
    v = buf.cpu().view(-1, nblk, 4)
    t_begin, t_end = int(v[n_win, 0, 3]), int(v[n_win + 1, 0, 3])
    mhz, per_xcd = [], {}
    for k in range(n_win):
        for b in range(nblk):
            clk, ticks, xcc, r0 = (int(q) for q in v[k, b])
            if ticks > 0 and r0 >= t_begin and r0 + ticks <= t_end:
                f = 100.0 * clk / ticks
                mhz.append(f)
                per_xcd.setdefault(xcc & 0xf, []).append(f)
    if not mhz:
        return None, pw
    mhz.sort()
    return {"mean": round(sum(mhz) / len(mhz), 1), "min": round(mhz[0], 1), "p10": round(mhz[len(mhz) // 10], 1), "median": round(mhz[len(mhz) // 2], 1),
            "max": round(mhz[-1], 1), "windows": len(mhz), "window_us": win_us,
            "per_xcd_mean": {str(k): round(sum(a) / len(a), 1) for k, a in sorted(per_xcd.items())},
            "how": "rd_clock_probe: s_memtime / s_memrealtime over 1-ms windows on a side stream, one wave per XCD, concurrent with %d "
                   "steps of an extra pass outside the timed region; windows inside the steps only" % steps}, pw


# BASELINE.json configs (index as in the file): (arch, per-GPU batch, height, width, storage)
CONFIGS = {
    2: ("resnet18_latefusion", 16, 450, 800, "fp32"),
    3: ("resnet18_latefusion", 16, 450, 800, "bf16"),
    4: ("resnet18_multistage_uncertainty_fixs", 8, 450, 800, "fp32"),
    5: ("resnet18_multistage_uncertainty_fixs", 8, 900, 1600, "bf16"),
}
# SURVEY.md 8(d) / BASELINE.md section 4: sum-over-layers roofline bound in samples/s per GPU and algorithmic work per sample;
# bf16 bounds at 6.29 TB/s measured copy rate / 8 TB/s spec.  Other sizes scale with the pixel count.
BOUNDS = {
    ("resnet18_latefusion", 450, 800): {"fp32": 1494.0, "bf16": (15181.0, 17527.0), "gflop": 104.57, "gb": (0.658, 0.329)},
    ("resnet18_latefusion", 900, 1600): {"fp32": 381.0, "bf16": (3861.0, 4461.0), "gflop": 410.42, "gb": (2.593, 1.296)},
    ("resnet18_multistage_uncertainty_fixs", 450, 800): {"fp32": 746.0, "bf16": (7545.0, 8715.0), "gflop": 209.57, "gb": (1.326, 0.663)},
    ("resnet18_multistage_uncertainty_fixs", 900, 1600): {"fp32": 190.0, "bf16": (1919.0, 2218.0), "gflop": 822.53, "gb": (5.226, 2.613)},
}


def bound_for(arch, height, width):
    b = BOUNDS.get((arch, height, width))
    if b is not None:
        return b
    r = 360000.0 / (height * width)
    b0 = BOUNDS[(arch, 450, 800)]
    return {"fp32": b0["fp32"] * r, "bf16": (b0["bf16"][0] * r, b0["bf16"][1] * r), "gflop": b0["gflop"] / r,
            "gb": (b0["gb"][0] / r, b0["gb"][1] / r)}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...` (one rank
    per GPU, rendezvous on 127.0.0.1).  Rank 0 of the relaunched job prints the one JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvp(cmd[0], cmd)


def dry_run(args, world, rank):
    """No GPU: the multi-rank plumbing of this script (rendezvous, per-rank seeds, bucketed gradient exchange of the plan's own
    bucket map, barrier + max-over-ranks timing, the single JSON line) over gloo on CPU.  Nothing is launched or measured."""
    import types

    import torch.distributed as dist

    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.main import _param_offsets, bucket_segments, create_model, reduce_gradient_buckets
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    made = create_model(types.SimpleNamespace(arch="resnet18_latefusion", decoder="upproj", modality="rgbd", pretrained=False), [64, 96])
    model = made[0] if isinstance(made, tuple) else made
    plan = LateFusionPlan(model, 1, 64, 96, train=True, dry_run=True, segment_joins=False)
    offs = _param_offsets(model)
    buckets = [sl for _, _, sl in bucket_segments(plan, offs)]
    total = max(v[1] for v in offs.values())
    grads = torch.full(((total + 3) // 4 * 4,), float(rank + 1))
    t0 = time.perf_counter()
    if world > 1:
        for _ in reduce_gradient_buckets(grads, buckets):
            pass
        dist.barrier()
    dt = time.perf_counter() - t0
    want = world * (world + 1) / 2.0
    covered = sum(hi - lo for bk in buckets for lo, hi in bk)
    ok = bool((grads == want).all().item()) and covered == grads.numel()
    per_rank = [dt]
    # the self-verification fields of the real line: identical parameters on every rank (same seed -> same checksum), per-rank values
    checksum = int(sum(int(p.detach().view(torch.int32).to(torch.int64).sum().item()) for p in model.parameters()))
    sums = [checksum]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, (dt, checksum))
        per_rank = [b[0] for b in box]
        sums = [b[1] for b in box]
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "samples/s", "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True,
                          "comm": "gloo" if world > 1 else "none", "rccl_world_size": None, "params_identical_across_ranks": len(set(sums)) == 1,
                          "final_loss_per_rank": None, "final_loss_spread": None, "gradient_buckets": len(buckets), "exchange_ok": ok,
                          "exchange_s_per_rank": [round(v, 4) for v in per_rank], "plan_ops": len(plan.prep + plan.fwd + plan.bwd)}), flush=True)
    if not ok:
        raise SystemExit("dry run: the bucketed exchange did not produce the all-rank sum over the whole arena")


def eager_block(model, x, t, args, operands, fused_value=None):
    model.operands = operands
    dt, final_loss, host = eager_loop(model, x, t, args.warmup, args.steps)
    v = args.batch * args.steps / dt
    blk = {"operands": operands, "value": round(v, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / args.steps, 3),
           "host_issue_ms_per_step": round(1e3 * host / args.steps, 3), "steps": args.steps, "warmup": args.warmup, "final_loss": round(final_loss, 5),
           "loop": "pred = model(x); loss = MaskedL1Loss()(pred, target); optimizer.zero_grad(); loss.backward(); optimizer.step() -- "
                   "radar_depth_amd drop-in modules + torch.optim.SGD(lr .01, momentum .9, wd 1e-4): the reference's main.py:440-445 verbatim"}
    if fused_value:
        blk["frac_of_fused_step"] = round(v / fused_value, 4)
    return blk


def eager_main(args, ts, model, x, t):
    """`--mode eager`: the line's value IS the reference's loop body through the drop-in modules (no HipTrainStep anywhere)."""
    blk = eager_block(model, x, t, args, args.operands)
    out = {"metric": METRIC, "value": blk["value"], "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": blk["ms_per_step"], "host_issue_ms_per_step": blk["host_issue_ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "eager",
           "config": {"workload": "%s --decoder upproj --modality rgbd, b=%d/GPU %dx%d fp32, full step through the nn.Module surface: %s"
                                  % (args.arch, args.batch, args.height, args.width, blk["loop"]),
                      "baseline_config": args.config, "global_batch": args.batch, "parallelism": "single", "final_loss": blk["final_loss"],
                      "rd_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("RD_")}},
           "arith": "fp32 via 3 x bf16 split, 6 MFMA terms, fp32 accumulate" if args.operands == "split" else "fp32 MFMA"}
    if (args.batch, args.height, args.width) != (16, 450, 800):
        out["metric"] = "training samples/sec, %s b=%d %dx%d rgbd" % (args.arch, args.batch, args.height, args.width)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.arch, args.batch, args.height, args.width)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json configs index: 2 = latefusion b=16 450x800 fp32 (the metric's configuration, the default), "
                         "3 = the same with bf16 storage, 4 = multistage_uncertainty_fixs b=8 450x800 fp32, 5 = multistage b=8/GPU "
                         "900x1600 bf16 storage; sets --arch/--batch/--height/--width/--storage")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default 16, or the --config's")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--arch", default=None,
                    choices=["resnet18_latefusion", "resnet18_multistage_uncertainty_fixs"],
                    help="headline = resnet18_latefusion (BASELINE configs[1]); the multistage arch is configs[3] (use --batch 8)")
    ap.add_argument("--graph", action="store_true", help="replay the step as hipGraphs (slower than plain stream launches here)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra `alt_fp32_mfma` measurement (the same workload on the plain fp32-MFMA plan; also skipped by "
                                                        "--no-roofline / --no-cpu-baseline, i.e. by the profiling command lines)")
    ap.add_argument("--operands", default=None, choices=["fp32", "bf16", "split"],
                    help="arithmetic of the convolutions.  split (default for fp32 storage; config.arith) = fp32 arithmetic on the bf16 matrix cores: "
                         "every fp32 operand as three bf16 pieces, six MFMA terms per product, fp32 accumulation (csrc/gconv_split.hip, wgrad_split.hip; "
                         "fp32 tensors; pinned against the CPU oracle at the fp32 tolerances at BASELINE's own batch sizes, tests/test_gpu_configs.py); "
                         "fp32 = every convolution on v_mfma_f32_32x32x2_f32; bf16 (BASELINE configs 3/5) rounds the operands to bf16 and is reported "
                         "with dtype \"bf16\" under its own metric name -- never as the fp32 headline")
    ap.add_argument("--storage", default=None, choices=["fp32", "bf16"],
                    help="element type of the NHWC activation / gradient tensors in HBM.  bf16 (BASELINE.json configs 3 / 5) implies "
                         "--operands bf16; statistics, parameters, their gradients and the optimizer stay fp32")
    ap.add_argument("--autotune", action="store_true",
                    help="time the candidate plans of every fp32 convolution descriptor once while the plan is built, before "
                         "the warm-up steps, and pin the fastest (radar_depth_amd/autotune.py; cudnn.benchmark's role).  Off by "
                         "default: at the headline geometry the gain is inside the run-to-run spread (+0..1.3 percent), and heuristic plans "
                         "keep the bench line, the kernel trace and the counter passes on identical launches -- and identical on "
                         "every data-parallel rank (timing-dependent plans would give ranks different summation orders)")
    ap.add_argument("--no-autotune", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="gradient exchange for --gpus > 1: rccl = the C ABI's own communicator (rd_allreduce_bucket on a communication "
                         "stream, event-chained per backward segment); torch = torch.distributed.all_reduce (cross-check)")
    ap.add_argument("--mode", default="fused", choices=["fused", "eager"],
                    help="fused (default): HipTrainStep, the whole step as one rd_optable_run call.  eager: `value` is the reference's own loop "
                         "body through the drop-in modules -- pred = model(x); loss = criterion(pred, target); optimizer.zero_grad(); "
                         "loss.backward(); optimizer.step() with torch.optim.SGD (main.py:440-445); latefusion / fp32 storage only.  The default "
                         "line carries the same measurement as its `alt_eager` block")
    ap.add_argument("--no-clock", action="store_true", help="skip the in-step shader clock / socket power pass (roofline.shader_clock_mhz)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: exercise the multi-rank plumbing (self-launch, rendezvous, bucketed exchange, JSON line) over gloo")
    args = ap.parse_args()
    cfg = CONFIGS[args.config if args.config is not None else 2]
    args.arch = args.arch or cfg[0]
    args.batch = args.batch or (cfg[1] if args.config is not None or args.arch == cfg[0] else 8)
    args.height = args.height or cfg[2]
    args.width = args.width or cfg[3]
    args.storage = args.storage or (cfg[4] if args.config is not None else "fp32")
    if args.storage == "bf16":
        args.operands = "bf16"
    if args.operands is None:
        args.operands = "split"

    # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a %d-rank job (WORLD_SIZE): launch with torch.distributed.run "
                         "--nproc-per-node %d, or run it bare and it launches its ranks itself" % (args.gpus, world, args.gpus))
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    # RD_FORCE_DP=1 with one rank: run the data-parallel code path (segmented graphs + bucketed RCCL all-reduce) on a 1-rank
    # group, to measure its host/launch overhead against the single-graph step on the same box
    if world > 1 or os.environ.get("RD_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    comm_used = "none"
    if torch.distributed.is_initialized():
        comm_used = "torch"
        if args.comm == "rccl":
            # torch.distributed only carries the 128-byte RCCL token (and the barrier / max-over-ranks of the timing)
            try:
                from radar_depth_amd import comm as rd_comm
                rd_comm.init_from_torch_distributed()
                comm_used = "rccl"
            except Exception as ex:                          # noqa: BLE001 -- never lose the scaling run to the bootstrap
                print("[bench] native RCCL communicator unavailable (%s); falling back to torch.distributed" % ex, file=sys.stderr)

    import types

    from radar_depth_amd.main import HipTrainStep, create_model
    from radar_depth_amd.synthetic import make_batch

    torch.manual_seed(0)                                     # identical random init on every rank
    made = create_model(types.SimpleNamespace(arch=args.arch, decoder="upproj", modality="rgbd", pretrained=False),
                        [args.height, args.width])
    model, loss_weights = made if isinstance(made, tuple) else (made, None)
    model = model.cuda()
    if args.mode == "eager" and (world > 1 or args.arch != "resnet18_latefusion" or args.storage != "fp32" or args.operands not in ("split", "fp32")):
        raise SystemExit("bench.py --mode eager: the reference's loop body main.py:440-445 (latefusion, fp32 storage, split or fp32 operands, one GPU)")
    ts = None if args.mode == "eager" else HipTrainStep(
        model, args.batch, args.height, args.width, lr=0.01, momentum=0.9, weight_decay=1e-4,
        loss_weights=loss_weights, use_graph=args.graph, operands=args.operands,
        comm=comm_used if comm_used != "none" else "auto", storage=args.storage, autotune=bool(args.autotune))
    x, t = make_batch(args.batch, args.height, args.width, 1234 + 1000 * rank)
    x, t = x.cuda(), t.cuda()
    if args.mode == "eager":
        return eager_main(args, ts, model, x, t)

    def sync():
        # drain this rank's own work first: the step's all-reduces (own communicator, communication stream) are then complete
        # before the barrier's collective (torch's communicator) is enqueued -- collectives of two communicators never interleave
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step(x, t)
    sync()
    # per-step events on the caller's stream (step() fences it behind the step's own streams): median / min without any
    # synchronisation inside the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_issue = 0.0
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        h0 = time.perf_counter()
        loss, _ = ts.step(x, t)
        host_issue += time.perf_counter() - h0
        marks[k + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    per_rank_ms = [1e3 * dt / args.steps]
    if world > 1:
        box = [torch.zeros(1, device="cuda", dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(box, torch.tensor([dt], device="cuda", dtype=torch.float64))
        per_rank_ms = [1e3 * b.item() / args.steps for b in box]
        dt = max(b.item() for b in box)
    final_loss = float(loss.item())
    # self-verification of the data-parallel run (outside the timed region; one 16-byte all-gather): every rank must hold the SAME
    # parameters after the timed steps (checksum = integer sum of the arena's bit patterns, exact), the per-rank losses differ only
    # through the rank-distinct data, and RCCL's own communicator reports the world size the line claims
    checksum = int(ts.st["arena"].view(torch.int32).to(torch.int64).sum().item())
    rank_losses, rank_sums = [final_loss], [checksum]
    if world > 1:
        box = [torch.zeros(2, device="cuda", dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(box, torch.tensor([final_loss, float(checksum % (1 << 52))], device="cuda", dtype=torch.float64))
        rank_losses = [b[0].item() for b in box]
        rank_sums = [int(b[1].item()) for b in box]
    rccl_world = None
    if comm_used == "rccl":
        from radar_depth_amd import comm as rd_comm
        rccl_world = rd_comm.world()

    multistage = args.arch != "resnet18_latefusion"
    bf16 = args.operands == "bf16"
    bnd = bound_for(args.arch, args.height, args.width)
    env_knobs = {k: v for k, v in sorted(os.environ.items()) if k.startswith("RD_")}
    out = {
        "metric": METRIC, "value": round(world * args.batch * args.steps / dt, 2), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
        "ms_per_step_rank_min": round(min(per_rank_ms), 3), "ms_per_step_rank_max": round(max(per_rank_ms), 3),
        "host_issue_ms_per_step": round(1e3 * host_issue / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s --decoder upproj --modality rgbd, b=%d/GPU %dx%d fp32, full step "
                               "(fwd + loss + bwd + SGD momentum .9 wd 1e-4), random init" % (args.arch, args.batch, args.height, args.width),
                   "baseline_config": args.config if args.config is not None else (2 if (args.arch, args.batch, args.height, args.width, args.storage) == CONFIGS[2] and args.operands in ("fp32", "split") else None),
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world if world > 1 else ("dp1 (forced data-parallel code path)" if os.environ.get("RD_FORCE_DP") == "1" else "single"),
                   "hipgraph": args.graph, "comm": comm_used, "rccl_world_size": rccl_world,
                   "params_identical_across_ranks": len(set(rank_sums)) == 1, "final_loss_per_rank": [round(v, 5) for v in rank_losses],
                   "final_loss_spread": round(max(rank_losses) - min(rank_losses), 6),
                   "autotuned_plans": bool(args.autotune), "final_loss": round(final_loss, 5),
                   "step_issue": "one rd_optable_run call per step (%d ops)" % len(ts._ops),
                   "rd_env": env_knobs},
    }
    if multistage or args.batch != 16 or (args.height, args.width) != (450, 800):
        out["metric"] = "training samples/sec, %s b=%d %dx%d rgbd" % (args.arch, args.batch, args.height, args.width)
    if bf16:
        out["dtype"] = "bf16"
        out["metric"] += " [bf16 conv operands on v_mfma_f32_32x32x16_bf16: forward, input gradients, weight gradients of the >=32-channel layers; stems/head/16-channel weight gradients fp32; fp32 tensors/accumulation]"
        out["config"]["workload"] = out["config"]["workload"].replace(" fp32,", " bf16-operand convs,")
        if args.storage == "bf16":
            out["metric"] = out["metric"].split(" [")[0] + " [bf16 storage: NHWC activations and gradients bf16 in HBM, bf16 MFMA convolutions incl. every weight gradient, fp32 accumulation / BatchNorm statistics / loss / parameters / SGD]"
            out["config"]["workload"] = out["config"]["workload"].replace(" bf16-operand convs,", " bf16 storage + bf16 convs,")
    split = args.operands == "split"
    if split:
        kinds0 = [k for pl in ts.plans for k, _ in pl.meta.values()]
        out["config"]["arith"] = ("fp32 tensors and fp32 results; the convolutions the library has a split plan for (%d forward / input-gradient and %d "
                                  "weight-gradient launches per step, plus the RGB stem's forward and both stems' weight gradients) compute every fp32 product on the bf16 matrix cores from three bf16 pieces per "
                                  "operand (x = x0 + x1 + x2 exactly), six v_mfma_f32_32x32x16_bf16 terms per product (the three dropped terms are below "
                                  "2^-24 of it), fp32 accumulation; error vs fp64 at the level of the fp32 MFMA kernels, pinned against the CPU oracle at "
                                  "this batch size at the fp32 bars (tests/test_gpu_configs.py, tests/test_gpu_gconv_split.py, tests/test_gpu_wgrad_split.py, tests/test_gpu_stem.py); "
                                  "all other kernels fp32 (v_mfma_f32_32x32x2_f32 / VALU)" % (kinds0.count("gconv_split") + kinds0.count("gconv_split_pre"), kinds0.count("wgrad_split") + kinds0.count("wgrad_split_pre")))
    elif not bf16:
        out["config"]["arith"] = "fp32 everywhere: every convolution on v_mfma_f32_32x32x2_f32 (the alternate plan of the default line)"
    # short arithmetic tag at the top level (the metric string is BASELINE.json's, the arithmetic must still be visible in one glance)
    out["arith"] = ("fp32 via 3 x bf16 split, 6 MFMA terms, fp32 accumulate" if split else "bf16 storage + bf16 MFMA" if args.storage == "bf16"
                    else "bf16 MFMA operands, fp32 tensors" if bf16 else "fp32 MFMA")
    per_gpu = out["value"] / world
    if rank == 0 and not args.no_roofline:
        agg, fam = instrumented_pass(ts)
        name, (ms, n, flops, nbytes) = max(agg.items(), key=lambda kv: kv[1][0])
        suffix = {("fp32", "fp32"): "_fp32_mfma", ("bf16", "bf16"): "_bf16_storage", ("split", "fp32"): ""}.get((args.operands, args.storage))
        tfile = None
        if suffix is not None:
            import glob
            cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc_traffic%s.json" % suffix)))
            tfile = os.path.basename(cands[-1]) if cands else None       # the newest round's counter passes of THIS command line
        traffic, traffic_source = None, None
        if tfile and not multistage and (args.batch, args.height, args.width) == (16, 450, 800) and os.path.exists(os.path.join(REPO, "profiles", tfile)):
            # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes of THIS command line (FETCH_SIZE x2 +
            # WRITE_SIZE, tools/pmc_traffic.py); counters cannot be read from inside the run, so the figure is labelled with its source
            try:
                tj = json.load(open(os.path.join(REPO, "profiles", tfile)))
                key = name.replace(" ", "")
                # exact instantiation, or -- a kernel family the plan runs in several instantiations (wgrad_split_kernel<...>) -- the
                # dispatch-weighted mean over the instantiations whose name starts with the family's
                hits = [v for k, v in tj["kernels"].items() if k == key] or [v for k, v in tj["kernels"].items() if k.startswith(key + "<")]
                if hits:
                    nd = sum(v["dispatches"] for v in hits)
                    traffic = int(sum((v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]) * v["dispatches"] for v in hits) / max(nd, 1))
                    traffic_source = "profiles/%s@%s" % (tfile, tj.get("collected_at", "unknown"))
            except (OSError, KeyError, ValueError):
                pass
        common = {"kernel": name, "launches_per_step": n, "avg_launch_us": round(1e3 * ms / n, 2), "traffic": traffic,
                  "traffic_source": traffic_source}
        if name.startswith("wgrad_split_kernel<3,3,"):
            # (a rocprofv3 summary averages this instantiation over MORE launches: the first phase of every UpProj weight gradient is a
            #  <3,3> launch too -- smaller layers, counted here under the "(4 UpProj phase launches)" ops)
            common["launch_scope"] = "single-phase 3x3 weight gradients only; rocprofv3's average for the name also includes the UpProj ops' <3,3> phase launches"
        if split and name.startswith(("gconv_split_kernel", "wgrad_split_kernel", "gconv_sp2_kernel")):
            # the dominant kernel runs six bf16 MFMAs per fp32 multiply-add: priced against the dense bf16 peak on the MFMA FLOPs it
            # actually issues (6 x algorithmic); `fp32_equivalent_tflops` is the algorithmic rate next to the fp32 MFMA peak
            achieved = 6.0 * flops / (ms * 1e-3) / 1e12
            step_flops = bnd["gflop"] * 1e9 * args.batch
            out["roofline"] = dict(common, bound="mfma", achieved=round(achieved, 1), peak=2500.0, unit="TFLOP/s", frac=round(achieved / 2500.0, 4),
                                   note="bf16 MFMA FLOPs issued (6 per algorithmic fp32 FLOP) against the dense bf16 peak, i.e. algorithmic fp32 FLOPs "
                                        "against 2500/6 = 416.7 TFLOP/s; random-data sustained rate of this device is ~1850 TFLOP/s "
                                        "(tools/micro/mfma_agpr.hip: the clock drops to 1.85 GHz)",
                                   # what the device SUSTAINS on a pure v_mfma_f32_32x32x16_bf16 stream with random operands (the clock falls to
                                   # 1.73-1.85 GHz under that load: tools/micro/mfma_bf16_peak.hip, profiles/r03_mfma_bf16_peak.txt); `frac` above stays
                                   # priced against the nominal dense peak
                                   sustained_bf16_peak=1750.0, frac_of_sustained=round(achieved / 1750.0, 4),
                                   fp32_equivalent_tflops=round(flops / (ms * 1e-3) / 1e12, 1), fp32_equivalent_peak=round(2500.0 / 6, 1),
                                   fp32_mfma_peak=PEAK_FP32_TFLOPS, frac_of_fp32_mfma_peak=round(flops / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
                                   step_frac_of_split_bound=round(step_flops * args.steps / dt / 1e12 / (2500.0 / 6), 4),
                                   algorithmic_gflop_per_sample=round(bnd["gflop"], 2),
                                   step_conv_tflops=round(step_flops * args.steps / dt / 1e12, 2),
                                   bound_samples_per_s_per_gpu=round(bnd["fp32"], 1), step_frac_of_fp32_mfma_bound=round(per_gpu / bnd["fp32"], 4))
            by_kernel = {k: [round(v[0], 3), v[1], round(v[2] / (v[0] * 1e-3) / 1e12, 1)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        elif not bf16:
            achieved = flops / (ms * 1e-3) / 1e12
            step_flops = bnd["gflop"] * 1e9 * args.batch
            out["roofline"] = dict(common, bound="mfma", achieved=round(achieved, 2), peak=PEAK_FP32_TFLOPS, unit="TFLOP/s",
                                   frac=round(achieved / PEAK_FP32_TFLOPS, 4),
                                   algorithmic_gflop_per_sample=round(bnd["gflop"], 2),
                                   step_conv_tflops=round(step_flops * args.steps / dt / 1e12, 2),
                                   step_frac_of_peak=round(step_flops * args.steps / dt / 1e12 / PEAK_FP32_TFLOPS, 4),
                                   bound_samples_per_s_per_gpu=round(bnd["fp32"], 1), step_frac_of_bound=round(per_gpu / bnd["fp32"], 4))
            by_kernel = {k: [round(v[0], 3), v[1], round(v[2] / (v[0] * 1e-3) / 1e12, 1)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        else:
            # bf16 configurations are HBM-bound (arithmetic intensity ~ the ridge): the roofline of the dominant kernel is
            # algorithmic bytes per launch (operand tensors once in, result once out) / its average duration against 8 TB/s
            achieved = nbytes / (ms * 1e-3) / 1e9
            out["roofline"] = dict(common, bound="hbm", achieved=round(achieved, 1), peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 4),
                                   algorithmic_bytes_per_launch=int(nbytes / n), kernel_tflops=round(flops / (ms * 1e-3) / 1e12, 1),
                                   algorithmic_gb_per_sample=bnd["gb"][1 if args.storage == "bf16" else 0],
                                   bound_samples_per_s_per_gpu=[round(bnd["bf16"][0], 1), round(bnd["bf16"][1], 1)],
                                   bound_note="SURVEY 8(d) bf16 bound at 6.29 TB/s measured copy rate / 8 TB/s spec",
                                   step_frac_of_bound=round(per_gpu / bnd["bf16"][0], 4), step_frac_of_bound_8tbs=round(per_gpu / bnd["bf16"][1], 4))
            by_kernel = {k: [round(v[0], 3), v[1], round(v[3] / (v[0] * 1e-3) / 1e9, 0)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        out["roofline"]["eager_ms_by_family"] = {k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
        out["roofline"]["eager_ms_by_kernel"] = by_kernel
        if not args.no_clock and world == 1:
            # the clock the chip runs at INSIDE the step and the socket power over the same pass: what turns "the bf16
            # matrix plan is power-limited" from an inference into a measurement, and what makes lines of different boxes comparable
            try:
                with torch.cuda.stream(ts.side):
                    mhz, pw = clock_pass(ts.L, lambda: ts.step(x, t), args.steps, 1e3 * dt / args.steps, lambda: C.c_void_p(ts.side.cuda_stream))
                out["roofline"]["shader_clock_mhz"] = mhz
                out["roofline"]["socket_power_w"] = pw
                if mhz and "sustained_bf16_peak" in out["roofline"]:
                    # dense bf16 MFMA rate at the MEASURED in-step clock (2500 TFLOP/s is the figure at the 2400 MHz boost clock)
                    sus = 2500.0 * mhz["mean"] / 2400.0
                    out["roofline"]["sustained_bf16_peak"] = round(sus, 1)
                    out["roofline"]["sustained_bf16_peak_source"] = "2500 TFLOP/s x measured in-step shader clock (mean %.0f MHz) / 2400 MHz" % mhz["mean"]
                    out["roofline"]["frac_of_sustained"] = round(out["roofline"]["achieved"] / sus, 4)
            except Exception as e:                          # noqa: BLE001 -- diagnostics never take the metric down
                out["roofline"]["shader_clock_mhz"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if (rank == 0 and world == 1 and args.operands == "split" and args.storage == "fp32" and not args.no_alt and not args.graph
            and not args.no_roofline and not args.no_cpu_baseline and os.environ.get("RD_FORCE_DP") != "1"):
        # the same workload once more on the plain fp32-MFMA plan (every convolution on v_mfma_f32_32x32x2_f32), timed the same way in
        # this process, reported NEXT TO the metric -- `value` above is the default (split) plan and is not affected
        try:
            ts.close()
            torch.manual_seed(0)
            made2 = create_model(types.SimpleNamespace(arch=args.arch, decoder="upproj", modality="rgbd", pretrained=False), [args.height, args.width])
            model2, lw2 = made2 if isinstance(made2, tuple) else (made2, None)
            ts2 = HipTrainStep(model2.cuda(), args.batch, args.height, args.width, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=lw2,
                               operands="fp32", comm="auto")
            for _ in range(args.warmup):
                ts2.step(x, t)
            torch.cuda.synchronize()
            a0 = time.perf_counter()
            for _ in range(args.steps):
                loss2, _ = ts2.step(x, t)
            torch.cuda.synchronize()
            adt = time.perf_counter() - a0
            mhz2, pw2 = (None, None)
            if not args.no_clock:
                try:
                    with torch.cuda.stream(ts2.side):
                        mhz2, pw2 = clock_pass(ts2.L, lambda: ts2.step(x, t), args.steps, 1e3 * adt / args.steps, lambda: C.c_void_p(ts2.side.cuda_stream))
                except Exception as e:                      # noqa: BLE001
                    mhz2 = {"error": "%s: %s" % (type(e).__name__, e)}
            out["alt_fp32_mfma"] = {"operands": "fp32", "value": round(args.batch * args.steps / adt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * adt / args.steps, 3),
                                    "shader_clock_mhz": mhz2, "socket_power_w": pw2,
                                    "steps": args.steps, "warmup": args.warmup, "final_loss": round(float(loss2.item()), 5),
                                    "step_frac_of_bound": round(args.batch * args.steps / adt / bnd["fp32"], 4),
                                    "note": "same workload, same process, every convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32; rounds 1-3's headline "
                                            "plan); NOT the metric's value"}
            ts2.close()
        except Exception as e:      # the alternative line must never take the metric down with it
            out["alt_fp32_mfma"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if (rank == 0 and world == 1 and args.arch == "resnet18_latefusion" and args.storage == "fp32" and args.operands in ("split", "fp32")
            and not args.no_alt and not args.graph and not args.no_roofline and os.environ.get("RD_FORCE_DP") != "1"):
        # the drop-in route of INTEGRATION.md section 1, measured: the reference's loop body verbatim on the same model
        try:
            ts.close()
            out["alt_eager"] = eager_block(model, x, t, args, args.operands, fused_value=out["value"])
        except Exception as e:                              # noqa: BLE001
            out["alt_eager"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        torch.distributed.barrier()
    # communicator teardown BEFORE the result line, and C stdio flushed around it: RCCL writes its version banner through C
    # stdio, which would otherwise land after the JSON line when the buffers drain at exit
    if comm_used == "rccl":
        from radar_depth_amd import comm as rd_comm
        ts.synchronize_comm()
        rd_comm.destroy()
    C.CDLL(None).fflush(None)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.arch, args.batch, args.height, args.width)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
