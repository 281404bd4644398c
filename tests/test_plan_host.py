"""Host logic of the execution plan and of the data-parallel path, without a GPU:
 * a dry-run LateFusionPlan records the op lists on host buffers: every parameter gets exactly one gradient writer, the
   backward ops are ordered so that the gradient buckets complete in arena-contiguous slices;
 * world_size-2 gloo run of reduce_gradient_buckets == gradient averaging over the flat arena."""
import collections
import os

import pytest
import torch
import torch.multiprocessing as mp


def _model(h=97, w=161):
    from radar_depth_amd.model.models import ResNet_latefusion
    torch.manual_seed(0)
    return ResNet_latefusion(18, "upproj", [h, w], 4, False)


def test_dry_run_plan_structure():
    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.main import _param_offsets, bucket_segments
    m = _model()
    plan = LateFusionPlan(m, 2, 97, 161, train=True, dry_run=True)
    names = [n for n, _, _ in plan.bwd]
    fwd_convs = [n for n, _, _ in plan.fwd if n in plan.meta]
    assert len(fwd_convs) == 55 - 3 - 4            # 55 convs - (2 stems + conv3: own kernels) - 4 (each UpProj 5x5 pair is one fused launch)
    with pytest.raises(RuntimeError):
        plan.run_forward()
    offs = _param_offsets(m)
    segs = bucket_segments(plan, offs)
    assert len(segs) == 4 and segs[0][0] == 0 and segs[-1][1] == len(names)
    assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
    # buckets tile the whole arena exactly once
    cover = sorted(sl for _, _, sls in segs for sl in sls)
    assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert cover[-1][1] == m._ensure_arenas()["total"]
    # every segment ends with the side streams joined back (hipGraph capture requirement), and an op that writes a
    # gradient of bucket k never appears after segment k
    for begin, end, sls in segs:
        assert names[end - 1].endswith(".wait")
    seg_of = {}
    for k, (begin, end, sls) in enumerate(segs):
        for n in names[begin:end]:
            seg_of.setdefault(n.split(".")[0] + "|" + n, k)
    for k, (begin, end, prefixes) in enumerate(plan.bwd_segments):
        for n in names[begin:end]:
            if n.endswith((".wreduce", ".bwd_apply")) or n.endswith(".wgrad"):
                assert n.split(".")[0] in prefixes, (n, prefixes)
    # every conv weight has a wgrad reduce and every BN a backward apply
    n_w = sum(1 for n, p in m.named_parameters() if p.dim() == 4)
    n_reduce = sum(1 for n in names if n.endswith(".wreduce")) + 2 + 1      # + two stems + conv3 (own kernels)
    assert n_reduce == n_w
    n_bn = sum(1 for n, p in m.named_parameters() if n.endswith(".bias"))
    # one apply launch per BatchNorm, except the ten two-operand joins (4 UpProj + 3 + 3 down-sampling blocks), whose single
    # launch produces both input gradients
    assert sum(1 for n in names if n.endswith(".bwd_apply")) == n_bn - 10



def test_batched_slab_reductions(monkeypatch):
    """RD_WGRAD_REDUCE_BATCH=n: the slab reductions become rd_wgrad_reduce_batched ops of at most n jobs per stream, each issued
    inside the backward segment whose gradient bucket its tensors belong to (never behind the bucket boundary)."""
    from radar_depth_amd.engine import LateFusionPlan
    monkeypatch.setenv("RD_WGRAD_REDUCE_BATCH", "4")
    m = _model()
    plan = LateFusionPlan(m, 2, 97, 161, train=True, dry_run=True)
    names = [n for n, _, _ in plan.bwd]
    n_w = sum(1 for n, p in m.named_parameters() if p.dim() == 4)
    assert not any(n.endswith(".wreduce") for n in names)
    assert sum(len(jobs) for _, _, jobs in plan.reduce_batches) + 2 + 1 == n_w      # + two stems + conv3 (own kernels)
    assert sum(1 for n in names if n.endswith(".wreduce_all")) == len(plan.reduce_batches) < n_w // 2
    for b, (seg, stream, jobs) in enumerate(plan.reduce_batches):
        prefixes = plan.bwd_segments[seg][2]
        assert all(j.split(".")[0] in prefixes for j in jobs) and 1 <= len(jobs) <= plan.reduce_batch_max + 1, (seg, jobs)      # (an UpProj pair adds two jobs at once)
        op = "segment%d.s%d.b%d.wreduce_all" % (seg, stream, b)
        assert plan.bwd_segments[seg][0] <= names.index(op) < plan.bwd_segments[seg][1]


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radar_depth_amd.engine import LateFusionPlan
        from radar_depth_amd.main import _param_offsets, bucket_segments, reduce_gradient_buckets
        m = _model()
        plan = LateFusionPlan(m, 2, 97, 161, train=True, dry_run=True)
        st = m._ensure_arenas()
        segs = bucket_segments(plan, _param_offsets(m))
        g = torch.Generator().manual_seed(100 + rank)
        st["grads"].copy_(torch.randn(st["total"], generator=g))
        mine = st["grads"].clone()
        order = []
        for slices in reduce_gradient_buckets(st["grads"], [sl for _, _, sl in segs]):
            order.append(slices)
        other = torch.randn(st["total"], generator=torch.Generator().manual_seed(100 + (1 - rank)))
        ok = torch.allclose(st["grads"], mine + other, atol=1e-6) and len(order) == len(segs)
        # the per-parameter views see the reduced values (what the SGD kernel reads, scaled by 1/world)
        ok = ok and torch.equal(m._grad_view(m.conv3.weight).flatten(), st["grads"][-m.conv3.weight.numel() - (-m.conv3.weight.numel()) % 4:][:m.conv3.weight.numel()])
        q.put((rank, bool(ok)))
    finally:
        torch.distributed.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_multistage_buckets_partition_the_arena():
    """Data-parallel multistage step: eight gradient buckets (four backward segments of stage 2, then four of stage 1; the
    scalar w_stage1/2 ride with the last) that cover the flat gradient arena exactly once."""
    import types

    import torch

    from radar_depth_amd import main as hmain
    from radar_depth_amd.engine import LateFusionPlan
    args = types.SimpleNamespace(arch="resnet18_multistage_uncertainty_fixs", decoder="upproj", modality="rgbd", pretrained=False)
    model, _ = hmain.create_model(args, [64, 96])
    offs = hmain._param_offsets(model)
    p1 = LateFusionPlan(model.stage1, 1, 64, 96, train=True, dry_run=True)
    kept = torch.empty(1, 1, 64, 96)
    p2 = LateFusionPlan(model.stage2, 1, 64, 96, train=True, depth_planes=[kept, p1.pred], x_source=p1.x_in, dense_grad_dst=p1.dpred,
                        dry_run=True)
    seg2 = hmain.bucket_segments(p2, offs, "stage2.")
    seg1 = hmain.bucket_segments(p1, offs, "stage1.")
    assert len(seg2) == 4 and len(seg1) == 4
    tops = sorted((v[0], (v[1] + 3) // 4 * 4) for k, v in offs.items() if not k.startswith(("stage1.", "stage2.")))
    assert [k for k in offs if not k.startswith(("stage1.", "stage2."))] == ["w_stage1", "w_stage2"]
    covered = sorted(sl for _, _, bk in seg2 + seg1 for sl in bk) + tops
    covered.sort()
    total = max(v[1] for v in offs.values())
    assert covered[0][0] == 0 and covered[-1][1] == (total + 3) // 4 * 4
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def test_plan_cache_is_a_small_lru():
    """A plan owns ~11 GB of buffers at b=16 450x800: the per-module cache keeps the PLAN_CACHE_SIZE most recently used keys and
    drops plans of a rebuilt parameter arena first.  Eviction only drops the cache's REFERENCE: a HipTrainStep / HipInference /
    autograd node that still holds an evicted plan keeps a usable plan (events alive); it is closed when its last holder goes."""
    import gc

    from radar_depth_amd.model import models

    closed = []

    class FakePlan:
        def __init__(self, i):
            self.i = i

        def close(self):                                      # idempotent, like LateFusionPlan.close(): eviction closes an un-held plan
            if self.i not in closed:                          # eagerly, its __del__ then finds nothing left to do
                closed.append(self.i)

        def __del__(self):
            self.close()
    cap = models.PLAN_CACHE_SIZE
    cache = {}
    held = models.hold_plan(FakePlan(0))                 # what a HipTrainStep does: fetch once, hold, keep using
    cache[(0, 450, 800, True, 1, None, False, "fp32", True)] = held
    for i in range(1, cap + 2):
        models._evict_plans(cache, version=1)
        cache[(i, 450, 800, True, 1, None, False, "fp32", True)] = FakePlan(i)
        assert len(cache) <= cap
    gc.collect()
    assert sorted(k[0] for k in cache) == list(range(2, cap + 2))                # oldest keys went first
    assert closed == [1]                                  # the un-held evicted plan is gone; the held one was NOT closed under its holder
    # a rebuilt arena (new version): every plan of the old version leaves the cache, whatever its age
    models._evict_plans(cache, version=2)
    gc.collect()
    assert not cache and sorted(closed) == list(range(1, cap + 2))
    assert 0 not in closed and held.__dict__["evicted"] and held.__dict__["holders"] == 1
    models.release_plan(held)                            # HipTrainStep.close(): the LAST holder of an evicted plan closes it -- by count,
    assert 0 in closed                                   # not by sys.getrefcount (another name bound to the plan changes nothing)
    # an autograd node holds through a finalizer on the node object
    class Node:
        pass
    node, p9 = Node(), FakePlan(9)
    models.hold_plan(p9, node)
    cache[(9, 1, 1, True, 2, None)] = p9
    models._evict_plans(cache, version=3)
    assert 9 not in closed
    del node
    gc.collect()
    assert 9 in closed


def test_segment_events_replace_joins():
    """segment_joins=False (native communicator / single GPU): the three inner bucket boundaries record one event per side
    stream instead of joining the streams; only the end of backward joins.  The op lists are otherwise identical."""
    import torch

    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.model.models import ResNet_latefusion
    m = ResNet_latefusion(18, "upproj", [64, 96], 4, False)
    pj = LateFusionPlan(m, 1, 64, 96, train=True, dry_run=True, segment_joins=True)
    pe = LateFusionPlan(m, 1, 64, 96, train=True, dry_run=True, segment_joins=False)
    assert [len(e) for e in pj.segment_events] == [0, 0, 0, 0]
    assert [len(e) for e in pe.segment_events] == [2, 2, 2, 0]
    names = lambda plan: [n for n, _, _ in plan.bwd]
    strip = lambda ns: [n for n in ns if not n.startswith(("join", "segment_end", "fork_depth"))]
    assert strip(names(pj)) == strip(names(pe))
    assert names(pj).count("join1.record") == 4 and names(pe).count("join1.record") == 1
    # without joins the depth chain is forked from the main stream ONCE (it needs the fusion layer's input gradient and nothing else)
    assert names(pj).count("fork_depth.record") == 3 and names(pe).count("fork_depth.record") == 1
    assert [s[2] for s in pj.bwd_segments] == [s[2] for s in pe.bwd_segments]


def test_op_table_marshals_every_plan_op():
    """rd_optable_* (the one-call replay of a step, include/radar_depth_hip.h): every op of the training and inference plans of
    both storage types names a replayable entry point and carries exactly the argument count of its C prototype; stream
    arguments are recognised by identity.  Marshalling needs no GPU (nothing is launched)."""
    import ctypes as C

    import pytest

    from radar_depth_amd._lib import RadarDepthHipError, lib
    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.optable import OpTable, _word
    L = lib()
    m = ResNet_latefusion(18, "upproj", [64, 96], 4, False)
    for kw in (dict(train=True), dict(train=True, storage="bf16"), dict(train=False), dict(train=True, bf16=True)):
        plan = LateFusionPlan(m, 1, 64, 96, dry_run=True, **kw)
        ops = plan.prep + plan.fwd + plan.bwd
        tb = OpTable(L, ops, plan.streams)
        assert len(tb) == len(ops) == L.rd_optable_size(tb.h)
        for name, fn, args in ops:
            assert L.rd_optable_entry_args(fn.__name__.encode()) == len(args), (name, fn.__name__)
            assert sum(1 for a in args if any(a is s for s in plan.streams)) >= 1, name      # every op is bound to a plan stream
        tb.close()
    # floats travel as their bit pattern, negative ints as two's complement, byref as the address
    assert _word(C.c_float(1.0)) == 0x3F800000 and _word(-1) == 2 ** 64 - 1
    d = C.c_int32(5)
    assert _word(C.byref(d)) == C.addressof(d)
    with pytest.raises(RadarDepthHipError):
        OpTable(L, [("bad", L.rd_last_error, ())], [])                     # not a replayable entry point
    with pytest.raises(RadarDepthHipError):
        OpTable(L, [("short", L.rd_fill, (C.c_void_p(0), C.c_int64(1)))], [])     # wrong argument count


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts the N=1 run) must start its two ranks
    itself -- torch.distributed.run on 127.0.0.1 -- and rank 0 prints ONE JSON line.  --dry-run swaps the GPU step for the
    plan's bucketed gradient exchange over gloo, so the whole multi-rank plumbing of the script runs here on CPU."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["comm"] == "gloo"
    assert out["exchange_ok"] is True and out["gradient_buckets"] == 4 and len(out["exchange_s_per_rank"]) == 2
    # the self-verification fields of the multi-rank line (identical parameter checksums on every rank; RCCL's own world size and the
    # per-rank losses are filled in by the GPU run)
    assert out["params_identical_across_ranks"] is True and "rccl_world_size" in out and "final_loss_spread" in out
    # a rank count that contradicts the surrounding job is refused, not silently run
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="3", RANK="0"), timeout=120)
    assert r.returncode != 0 and "3-rank job" in (r.stderr + r.stdout)


def test_state_broadcast_packing_round_trip():
    """pack_buffers / unpack_buffers (the flat word array the data-parallel state broadcast ships): bit-preserving for the fp32
    running statistics AND the int64 0-dim num_batches_tracked counters, written back in place."""
    import torch

    from radar_depth_amd.main import pack_buffers, unpack_buffers
    from radar_depth_amd.model.models import ResNet_latefusion
    m = ResNet_latefusion(18, "upproj", [64, 96], 4, False)
    bufs = list(m.buffers())
    assert any(b.dtype == torch.int64 and b.dim() == 0 for b in bufs) and any(b.dtype == torch.float32 for b in bufs)
    g = torch.Generator().manual_seed(3)
    for b in bufs:                                  # "rank 0" state
        if b.dtype == torch.int64:
            b.fill_(int(torch.randint(1, 2 ** 40, (1,), generator=g)))
        else:
            b.copy_(torch.randn(b.shape, generator=g))
    want = [b.clone() for b in bufs]
    flat = pack_buffers(bufs)
    assert flat.dtype == torch.float32 and flat.numel() == sum(b.numel() * b.element_size() // 4 for b in bufs)
    ptrs = [b.data_ptr() for b in bufs]
    for b in bufs:                                  # "rank 1" state before the broadcast
        b.zero_()
    unpack_buffers(bufs, flat)
    assert all(torch.equal(a, b) for a, b in zip(bufs, want)) and ptrs == [b.data_ptr() for b in bufs]


def test_split_plan_routes_supported_convolutions(monkeypatch):
    """operands="split": every forward / input-gradient convolution the library has a split plan for goes to rd_gconv_split with a
    three-piece bf16 operand (pack quad 3); the rest keeps rd_gconv with the fp32 operand; the 3x3 / stride-1 weight gradients go to
    rd_wgrad_split (same slabs and reduction order), the others stay on rd_wgrad.  RD_SPLIT_BNB=1 (round 6; off by default: it measured
    -0.7 ... -1 % on the step, profiles/r06_split_bnb_ab.txt): the input gradients of the conv -> BN -> ReLU -> conv chains also emit that
    BatchNorm's backward sums (rd_gconv_split[_pre]_bnbwd / rd_wino_conv3x3_bnbwd) and their rd_bn_bwd_reduce_x_t passes are gone."""
    import ctypes as C
    from radar_depth_amd.engine import LateFusionPlan
    m = _model(450, 800)
    default = LateFusionPlan(m, 16, 450, 800, train=True, dry_run=True, split=True)
    assert not any(getattr(f, "__name__", "").endswith("_bnbwd") for _, f, _ in default.bwd)
    monkeypatch.setenv("RD_SPLIT_BNB", "1")
    plan = LateFusionPlan(m, 16, 450, 800, train=True, dry_run=True, split=True)
    ref = LateFusionPlan(m, 16, 450, 800, train=True, dry_run=True)
    kinds = collections.Counter(k for k, _ in plan.meta.values())
    n_wgrad = collections.Counter(k for k, _ in ref.meta.values())["wgrad"]
    assert kinds["gconv_split"] > 30 and kinds["gconv"] > 0
    assert kinds["wgrad_split"] >= 20 and kinds["wgrad_split"] + kinds["wgrad"] == n_wgrad      # 3x3 / stride-1 layers with >= 64 channels
    L = plan.L
    for name, (kind, d) in plan.meta.items():
        if kind == "gconv_split":
            assert L.rd_gconv_split_supported(C.byref(d)) == 1, name
            assert min(d.Cin, d.Cout) >= 32
    quads = collections.Counter(j[-1] for j in plan.pack_jobs)
    assert quads[3] > 0 and quads[1] > 0 and quads[2] == 0
    for j in plan.pack_jobs:
        if j[-1] == 3:
            assert j[1].dtype == torch.bfloat16 and j[1].shape[0] == 3
    names = [n for n, _, _ in plan.fwd + plan.bwd]
    assert any(n.endswith(".dgrad") for n in names)
    fns = collections.Counter(f.__name__ if hasattr(f, "__name__") else str(f) for _, f, _ in plan.fwd + plan.bwd)
    assert fns["rd_gconv_split"] + fns["rd_gconv_split_bnbwd"] == kinds["gconv_split"]
    assert fns["rd_gconv_split_pre"] + fns["rd_gconv_split_pre_bnbwd"] == kinds["gconv_split_pre"]
    assert fns["rd_wino_conv3x3"] + fns["rd_wino_conv3x3_bnbwd"] == kinds["wino"] >= 16
    n_bnb = fns["rd_gconv_split_bnbwd"] + fns["rd_gconv_split_pre_bnbwd"] + fns["rd_wino_conv3x3_bnbwd"]
    ref_reduce = collections.Counter(f.__name__ for _, f, _ in ref.bwd if hasattr(f, "__name__"))["rd_bn_bwd_reduce_x_t"]
    assert n_bnb >= 14 and fns["rd_bn_bwd_reduce_x_t"] <= 8, (n_bnb, fns["rd_bn_bwd_reduce_x_t"], ref_reduce)
    assert fns["rd_wgrad_split"] == kinds["wgrad_split"] and fns["rd_wgrad_split_reduce"] >= kinds["wgrad_split"]


def test_bind_input_repoints_the_stem_plane_tables():
    """LateFusionPlan.bind_input (what the fused step uses to read the caller's batch in place): the stems' host-side plane tables follow
    the given base pointer and channel count -- RGB planes 0..2, the depth plane 3 -- and bind_own_input restores the plan's buffer."""
    from radar_depth_amd.engine import LateFusionPlan
    m = _model(64, 96)
    plan = LateFusionPlan(m, 2, 64, 96, train=True, dry_run=True)
    hw, base0 = 64 * 96, plan.x_in.data_ptr()
    (pl_rgb, st_rgb, c0, n), (pl_d, st_d, c0d, nd) = plan.x_bind
    assert (c0, n, c0d, nd) == (0, 3, 3, 1)
    assert [pl_rgb[c] for c in range(3)] == [base0 + 4 * hw * c for c in range(3)] and pl_d[0] == base0 + 4 * hw * 3
    plan.bind_input(1 << 40, 6)                          # a [N,6,H,W] batch somewhere else
    assert [pl_rgb[c] for c in range(3)] == [(1 << 40) + 4 * hw * c for c in range(3)] and pl_d[0] == (1 << 40) + 4 * hw * 3
    assert list(st_rgb)[:3] == [6 * hw] * 3 and st_d[0] == 6 * hw and plan._x_bound == 1 << 40
    plan.bind_own_input()
    assert pl_rgb[0] == base0 and st_rgb[0] == plan.x_in.shape[1] * hw and plan._x_bound == base0
