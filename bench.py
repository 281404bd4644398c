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
                  average launch duration against the fp32 MFMA/vector peak (157.3 TFLOP/s);
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
    if kind == "wgrad_bf16":
        info = (C.c_int32 * 8)()
        L.rd_wgrad_bf16_plan_info(C.byref(d), info)           # cpi, cpo, ...
        full = d.n_phases == 1 and d.in_stride == 1 and d.phase[0].n_taps == 9
        return "wgrad_bf16_kernel<%s,%d,%s>" % (tb_(full), 4 // (info[0] * info[1]), tb_(io16))
    if kind == "gconv":
        info = (C.c_int32 * 10)()
        L.rd_gconv_plan_info(C.byref(d), info)                 # info[4] = grouped*1000000 + pipelined*10000 + ksplit*100 + CKW
        grouped = info[4] >= 1000000                          # in_stride == 2 run as input-parity groups: SWZ = false instantiation
        if info[0] == 0:                                      # 16 -> 16 channel 3x3 layers: csrc/conv16.hip (16x16x4 MFMA)
            return "conv16_kernel<stat|add>"
        return "gconv_kernel<%d,%d,%d,%d,%d,%s,%s,%s>" % (info[0], info[1], info[2], info[3], info[4] % 100,
                                                          "true" if (d.in_stride == 2 and not grouped) else "false",
                                                          "true" if info[4] % 1000000 >= 10000 else "false", "true" if grouped else "false")
    w = (C.c_int32 * 9)()
    L.rd_wgrad_plan_info(C.byref(d), w)
    if w[0] == 0:      # 16 -> 16 channel 3x3 layers: csrc/wgrad16.hip (16x16x4 MFMA)
        return "wgrad16_kernel"
    tb = lambda v: "true" if v else "false"
    if w[8]:       # column-strip kernel (one per UpProj phase when w[6])
        return "wgrad_strip_kernel<%s>%s" % ("3,3|2,3|3,2|2,2" if w[6] else "%d,%d" % (w[0] // w[5], w[5]), " (4 UpProj phase launches)" if w[6] else "")
    if w[6]:       # UpProj: four launches (9/6/6/4-tap sub-stencils) inside one op
        return "wgrad_kernel<9|6|6|4,%d,%s,true,%d,3|2> (4 UpProj phase launches)" % (w[1], tb(w[2]), w[4])
    return "wgrad_kernel<%d,%d,%s,%s,%d,%d>" % (w[0], w[1], tb(w[2]), tb(w[3]), w[4], 3 if w[4] == 0 else w[5])


def instrumented_pass(ts):
    """One eager step with every conv launch bracketed by events on the step's stream.
    Returns {kernel: [total_ms, launches, total_flops]} and the per-step op breakdown by family."""
    plan = ts.plan
    L = ts.L
    agg, fam = {}, {}
    with torch.cuda.stream(ts.side):
        plan.set_stream(serialize=True)      # all plan streams -> this stream, so per-op event pairs bracket the op
        recs = []
        # three un-instrumented passes first, back to back with the measured one: the device is then at its sustained
        # (power-managed) clock like in the timed region, not at the boost clock it reaches after an idle gap
        for _ in range(3):
            for lst in (plan.prep, plan.fwd, plan.bwd):
                for name, fn, args in lst:
                    assert fn(*args) == 0, name
        for lst in (plan.prep, plan.fwd, plan.bwd):
            for name, fn, args in lst:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args)
                e1.record()
                assert rc == 0, name
                recs.append((name, e0, e1))
        torch.cuda.synchronize()
        for name, e0, e1 in recs:
            ms = e0.elapsed_time(e1)
            tag = name.rsplit(".", 1)[-1]
            if tag in ("record", "wait"):
                continue
            family = {"wgrad": "wgrad", "dgrad": "dgrad", "wreduce": "wgrad_reduce", "pack": "pack", "packT": "pack"}.get(tag)
            if family is None:
                family = "conv_fwd" if name in plan.meta else ("bn/act/pool/head" if True else "other")
            fam[family] = fam.get(family, 0.0) + ms
            if name in plan.meta:
                kind, d = plan.meta[name]
                k = kernel_identity(L, kind, d, plan.storage == "bf16")
                a = agg.setdefault(k, [0.0, 0, 0.0, 0.0])
                a[0] += ms
                a[1] += 1
                a[2] += desc_flops(d)
                a[3] += desc_bytes(kind, d, 2 if plan.storage == "bf16" else 4)
    return agg, fam


def cpu_baseline(height, width):
    """Oracle (CPU restatement of the reference, pinned to it by golden vectors) full SGD steps on this box's host cores, per
    SURVEY.md 8(d): the C2 shape (b=16) and b=2, 2 warm-up + 5 timed steps each, best and median; the thread count is chosen by
    a quick sweep at b=2 (oversubscribing the host makes oneDNN slower, not faster)."""
    import statistics

    from oracle.criteria import MaskedL1Loss
    from oracle.models import ResNet_latefusion
    from radar_depth_amd.synthetic import make_batch
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [height, width], 4, False).train()
    opt = torch.optim.SGD(m.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    crit = MaskedL1Loss()

    def steps(x, t, n):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            loss = crit(m(x), t)
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append(time.perf_counter() - t0)
        return out
    ncpu = os.cpu_count() or 1
    x2, t2 = make_batch(2, height, width, 1234)
    steps(x2, t2, 1)                                        # oneDNN primitive creation
    sweep = {}
    for nt in sorted({max(1, ncpu // 32), max(1, ncpu // 16), max(1, ncpu // 8), max(1, ncpu // 4)}):
        torch.set_num_threads(nt)
        sweep[nt] = min(steps(x2, t2, 2))
    nt = min(sweep, key=sweep.get)
    torch.set_num_threads(nt)
    res = {}
    for b in (2, 16):
        x, t = (x2, t2) if b == 2 else make_batch(b, height, width, 1234)
        tt = steps(x, t, 7)[2:]
        res[b] = (b / min(tt), b / statistics.median(tt), min(tt))
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        cpu = "unknown"
    return {"value": round(res[16][0], 3), "median": round(res[16][1], 3), "unit": "samples/s", "cores": nt, "kind": "port",
            "b2_value": round(res[2][0], 3), "b2_median": round(res[2][1], 3),
            "thread_sweep_b2_step_s": {str(k): round(v, 3) for k, v in sweep.items()}, "cpu": cpu, "host_cpus": ncpu,
            "sample": "oracle (PyTorch CPU restatement pinned to the reference by golden vectors), resnet18_latefusion full SGD "
                      "step, %dx%d fp32: b=16 (the C2 shape) and b=2, 2 warm-up + 5 timed steps each, best (value) and median; "
                      "best b=16 step %.2f s; %d threads (fastest of the sweep) on %d host cpus (%s)"
                      % (height, width, res[16][2], nt, ncpu, cpu)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (weak scaling)")
    ap.add_argument("--height", type=int, default=450)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--arch", default="resnet18_latefusion",
                    choices=["resnet18_latefusion", "resnet18_multistage_uncertainty_fixs"],
                    help="headline = resnet18_latefusion (BASELINE configs[1]); the multistage arch is configs[3] (use --batch 8)")
    ap.add_argument("--graph", action="store_true", help="replay the step as hipGraphs (slower than plain stream launches here)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--operands", default="fp32", choices=["fp32", "bf16"],
                    help="conv operand precision.  fp32 (default) is the BASELINE.json configs[1] measurement; bf16 (configs 2/4) runs the "
                         "forward / input-gradient / weight-gradient convolutions on bf16 MFMA with fp32 tensors + accumulation and is reported with "
                         "dtype \"bf16\" and its own metric name -- never as the fp32 headline")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"],
                    help="element type of the NHWC activation / gradient tensors in HBM.  bf16 (BASELINE.json configs 3 / 5) implies "
                         "--operands bf16; statistics, parameters, their gradients and the optimizer stay fp32")
    ap.add_argument("--autotune", action="store_true",
                    help="time the candidate plans of every fp32 convolution descriptor once while the plan is built, before "
                         "the warm-up steps, and pin the fastest (radar_depth_amd/autotune.py; cudnn.benchmark's role).  Off by "
                         "default: at the headline geometry the gain is inside the run-to-run spread (+0..1.3 percent), and heuristic plans "
                         "keep the bench line, the kernel trace and the counter passes on identical launches")
    ap.add_argument("--no-autotune", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="gradient exchange for --gpus > 1: rccl = the C ABI's own communicator (rd_allreduce_bucket on a communication "
                         "stream, event-chained per backward segment); torch = torch.distributed.all_reduce (cross-check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    if args.storage == "bf16":
        args.operands = "bf16"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    # RD_FORCE_DP=1 with one rank: run the data-parallel code path (segmented graphs + bucketed RCCL all-reduce) on a 1-rank
    # group, to measure its host/launch overhead against the single-graph step on the same box
    if world > 1 or os.environ.get("RD_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
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
    ts = HipTrainStep(model, args.batch, args.height, args.width, lr=0.01, momentum=0.9, weight_decay=1e-4,
                      loss_weights=loss_weights, use_graph=args.graph, operands=args.operands,
                      comm=comm_used if comm_used != "none" else "auto", storage=args.storage, autotune=bool(args.autotune))
    x, t = make_batch(args.batch, args.height, args.width, 1234 + 1000 * rank)
    x, t = x.cuda(), t.cuda()

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
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        loss, _ = ts.step(x, t)
        marks[k + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    final_loss = float(loss.item())

    out = {
        "metric": METRIC, "value": round(world * args.batch * args.steps / dt, 2), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s --decoder upproj --modality rgbd, b=%d/GPU %dx%d fp32, full step "
                               "(fwd + loss + bwd + SGD momentum .9 wd 1e-4), random init" % (args.arch, args.batch, args.height, args.width),
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world if world > 1 else ("dp1 (forced data-parallel code path)" if os.environ.get("RD_FORCE_DP") == "1" else "single"),
                   "hipgraph": args.graph, "comm": comm_used, "autotuned_plans": bool(args.autotune), "final_loss": round(final_loss, 5)},
    }
    multistage = args.arch != "resnet18_latefusion"
    bf16 = args.operands == "bf16"
    if bf16:
        out["dtype"] = "bf16"
        out["metric"] = METRIC + " [bf16 conv operands on v_mfma_f32_32x32x16_bf16: forward, input gradients, weight gradients of the >=32-channel layers; stems/head/16-channel weight gradients fp32; fp32 tensors/accumulation]"
        out["config"]["workload"] = out["config"]["workload"].replace(" fp32,", " bf16-operand convs,")
        if args.storage == "bf16":
            out["metric"] = METRIC + " [bf16 storage: NHWC activations and gradients bf16 in HBM, bf16 MFMA convolutions incl. every weight gradient, fp32 accumulation / BatchNorm statistics / loss / parameters / SGD]"
            out["config"]["workload"] = out["config"]["workload"].replace(" bf16-operand convs,", " bf16 storage + bf16 convs,")
            # HBM-bound configuration: fraction of SURVEY 8(d)'s bf16 bound (0.329 GB/sample at 6.29 TB/s measured copy rate and 8 TB/s spec)
            out["roofline_note"] = "bf16 bound (SURVEY 8d): 15181 samples/s @6.29 TB/s, 17527 @8 TB/s per GPU at 450x800; this run: %.1f%% / %.1f%%" % (
                100 * out["value"] / world / (15181 * 360000.0 / (args.height * args.width)), 100 * out["value"] / world / (17527 * 360000.0 / (args.height * args.width)))
    if rank == 0 and not args.no_roofline and not multistage and not bf16:
        agg, fam = instrumented_pass(ts)
        name, (ms, n, flops, _) = max(agg.items(), key=lambda kv: kv[1][0])
        achieved = flops / (ms * 1e-3) / 1e12
        # algorithmic work of one training sample (UpProj zero-skipped, stem dgrads omitted): 104.57 GFLOP at 450x800
        alg_gflop = 104.57 * (args.height * args.width) / (450.0 * 800.0)
        step_flops = alg_gflop * 1e9 * args.batch
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE,
        # profiles/r01_pmc_traffic.json; collected with tools described in DESIGN.md, not re-measured on every run)
        traffic = None
        try:
            pmc = "r02_pmc_traffic.json" if os.path.exists(os.path.join(REPO, "profiles", "r02_pmc_traffic.json")) else "r01_pmc_traffic.json"
            tj = json.load(open(os.path.join(REPO, "profiles", pmc)))["kernels"]
            key = name.replace(" ", "")
            if key in tj:
                traffic = tj[key]["hbm_read_bytes_per_launch"] + tj[key]["hbm_write_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out["roofline"] = {"bound": "mfma", "kernel": name, "launches_per_step": n, "avg_launch_us": round(1e3 * ms / n, 2),
                           "achieved": round(achieved, 2), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                           "algorithmic_gflop_per_sample": round(alg_gflop, 2),
                           "step_conv_tflops": round(step_flops * args.steps / dt / 1e12, 2),
                           "step_frac_of_peak": round(step_flops * args.steps / dt / 1e12 / PEAK_FP32_TFLOPS, 4),
                           "eager_ms_by_family": {k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
                           "eager_ms_by_kernel": {k: [round(v[0], 3), v[1], round(v[2] / (v[0] * 1e-3) / 1e12, 1)]
                                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}}
    if rank == 0 and not args.no_roofline and not multistage and bf16:
        # bf16 configurations are HBM-bound (arithmetic intensity ~ the ridge): the roofline of the dominant kernel is
        # algorithmic bytes per launch (operand tensors once in, result once out) / its average duration against 8 TB/s
        agg, fam = instrumented_pass(ts)
        name, (ms, n, flops, nbytes) = max(agg.items(), key=lambda kv: kv[1][0])
        achieved = nbytes / (ms * 1e-3) / 1e9
        traffic = None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "r02_pmc_traffic_bf16_storage.json" if args.storage == "bf16"
                                             else "r02_pmc_traffic_bf16_operands.json")))["kernels"]
            if name in tj:
                traffic = tj[name]["hbm_read_bytes_per_launch"] + tj[name]["hbm_write_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out["roofline"] = {"bound": "hbm", "kernel": name, "launches_per_step": n, "avg_launch_us": round(1e3 * ms / n, 2),
                           "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                           "traffic": traffic, "algorithmic_bytes_per_launch": int(nbytes / n),
                           "kernel_tflops": round(flops / (ms * 1e-3) / 1e12, 1),
                           "eager_ms_by_family": {k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
                           "eager_ms_by_kernel": {k: [round(v[0], 3), v[1], round(v[3] / (v[0] * 1e-3) / 1e9, 0)]
                                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}}
    if world > 1:
        torch.distributed.barrier()
    # communicator teardown BEFORE the result line, and C stdio flushed around it: RCCL writes its version banner through C
    # stdio, which would otherwise land after the JSON line when the buffers drain at exit
    if comm_used == "rccl":
        from radar_depth_amd import comm as rd_comm
        rd_comm.destroy()
    C.CDLL(None).fflush(None)
    if rank == 0:
        if multistage:
            out["metric"] = "training samples/sec, %s b=%d %dx%d rgbd" % (args.arch, args.batch, args.height, args.width)
            out["roofline_note"] = "algorithmic work 209.57 GFLOP/sample at 450x800 (SURVEY 8d): %.1f%% of the fp32 peak" % (
                100 * 209.57e9 * (args.height * args.width / 360000.0) * out["value"] / 157.3e12)
        if world == 1 and not args.no_cpu_baseline and not multistage and not bf16:
            out["cpu_baseline"] = cpu_baseline(args.height, args.width)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
