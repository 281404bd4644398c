# end-of-round collection ON THE GPU BOX (started by tools/gpu_collect.sh, which refuses a dirty tree and passes the commit in RD_HEAD):
# full GPU test suite; kernel traces + counter passes of the default bench command line (the split plan) and of its alternates; every
# bench line; per-layer tables; workgroup traces / ablations of the split kernels; stress; resource audit.  Output: gpurun_out/$RD_ROUND/
R=$GRAFT_REPO_ROOT; RD_ROUND=${RD_ROUND:-r05}; export RD_ROUND; export RD_HEAD=${RD_HEAD:-$(cat $R/.collect_head 2>/dev/null || echo unknown)}
O=$R/gpurun_out/$RD_ROUND; mkdir -p $O; cd $R
echo "collecting $RD_ROUND at $RD_HEAD"
timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
# the margins the parity tests print: worst / median errors per plan, fp64-anchored gradients, the three-step trajectory
{ echo "# collected at git $RD_HEAD: margins printed by pytest -m gpu -s (tests/test_gpu_margins.py, tests/test_gpu_configs.py)"; grep -aE "fp64-anchored|three steps|gradient norms|config4 grad|plain multistage" $O/pytest.txt | sed 's/^[.sF]*//'; } > $O/parity_margins.txt
{ echo "# collected at git $RD_HEAD: python tools/diag_fp64.py (b=2 97x161) and 2 129 193"; python tools/diag_fp64.py 2>&1 | grep -v amdgpu.ids; python tools/diag_fp64.py 2 129 193 2>&1 | grep -v amdgpu.ids | tail -4; } > $O/diag_fp64.txt
# ---- bench lines (JSON, one per file)
python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
python bench.py --operands fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_fp32_mfma.json
RD_SPLIT_PRE=0 python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 > $O/bench_c2_split_while_staging.json
python bench.py --config 3 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 4 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 5 2>/dev/null | tail -1 > $O/bench_c5.json
RD_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1 > $O/bench_c2_dp1_torchrun.json
# ---- per-layer tables and kernel-level studies
python tools/bench_ops.py > $O/bench_ops_per_layer.txt 2>/dev/null
python tools/bench_split_pre.py 2>&1 | grep -v amdgpu.ids > $O/bench_split_pre.txt
python tools/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids > $O/bench_wgrad_split.txt
# round 5: slot-map A/B in one job (RD_GCONV_SPLIT_NATURAL=1: the round-4 row-major slots), bf16-storage per-layer A/B, ablations of the
# persistent bf16-storage kernel, fill-rate / zero-fill microbenchmarks, host issue time
{ echo "# collected at git $RD_HEAD: slot map A/B, one box, one job"; python tools/bench_split_pre.py 2>&1 | grep TOTAL | sed 's/^/slot map    : /'; RD_GCONV_SPLIT_NATURAL=1 python tools/bench_split_pre.py 2>&1 | grep TOTAL | sed 's/^/row-major   : /';
  for i in 1 2; do python bench.py --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('slot map    : %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))";
  RD_GCONV_SPLIT_NATURAL=1 python bench.py --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('row-major   : %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"; done; } > $O/slot_map_ab.txt 2>&1
python tools/bench_bf16_storage_ops.py 2>&1 | grep -v amdgpu.ids > $O/bf16_storage_ops_new.txt
RD_GCONV_BF16P=0 python tools/bench_bf16_storage_ops.py 2>&1 | grep -v amdgpu.ids > $O/bf16_storage_ops_old.txt
RD_GCONV_BF16P=all python tools/ablate_bf16p.py 2>&1 | grep -v amdgpu.ids > $O/ablate_bf16p.txt
{ /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_rate tools/micro/fill_rate.hip 2>/dev/null && /tmp/fill_rate; } > $O/fill_rate.txt 2>&1
{ /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/buf_lds_oob tools/micro/buf_lds_oob.hip 2>/dev/null && /tmp/buf_lds_oob; } > $O/buf_lds_oob.txt 2>&1
# host issue time: plain launches vs hipGraph replay of the SPLIT plan, configs 2 and 4, and config 4 through the data-parallel code path
{ echo "# collected at git $RD_HEAD: bench.py host_issue_ms_per_step; plain launches (default) vs --graph (hipGraph replay), split plan; dp1 = RD_FORCE_DP=1 under torchrun, one rank"
  for cfg in 2 4; do for g in "" "--graph"; do
    python bench.py --config $cfg $g --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('config $cfg %-8s: %8.1f samples/s  step %.3f ms  host issue %.3f ms' % ('$g' or 'plain', d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step') or float('nan')))"
  done; done
  RD_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --config 4 --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('config 4 dp1     : %8.1f samples/s  step %.3f ms  host issue %.3f ms' % (d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step') or float('nan')))"
} > $O/host_time.txt 2>&1
python tools/bench_bn.py fp32 2>/dev/null | grep "ch " | cut -c1-330 > $O/bench_bn.txt
# round 5 (second half): the stems' kernels alone (split weight gradient, folded apply pass, 2 x 2 pooling gather), ablations of the split stem
# weight gradient, un-profiled stream probes of configs 2 / 3 / 4, the unaligned-LDS-read microbenchmark
{ echo "# collected at git $RD_HEAD: python tools/bench_stem.py (b = 16, 450 x 800; kernels alone)"; python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids
  for d in 1 2; do echo "# RD_STEM_WGRAD_SPLIT_DEBUG=$d (1: no MFMA walk, 2: no staging after the first tile)"; RD_STEM_WGRAD_SPLIT_DEBUG=$d python tools/bench_stem.py 2>&1 | grep "on the bf16"; done; } > $O/bench_stem.txt
{ echo "# collected at git $RD_HEAD: python tools/tail_probe.py <config> 12 (RD_TAIL_EVENTS=1: timing events recorded from the op list of an UN-PROFILED step)"
  for c in 2 3 4; do echo "## config $c"; python tools/tail_probe.py $c 12 2>&1 | grep -v "amdgpu.ids\|Info"; done; } > $O/tail_probe.txt
{ /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_unaligned tools/micro/lds_unaligned.hip 2>/dev/null && /tmp/lds_unaligned; } > $O/lds_unaligned.txt 2>&1
for r in 1 2; do RD_GCONV_SPLIT_TRACE=$r python tools/trace_gconv_split.py 2>&1 | grep -v amdgpu.ids; done > $O/trace_gconv_split.txt
RD_GCONV_SPLIT_TRACE=1 python tools/trace_gconv_split.py --pre 2>&1 | grep -v amdgpu.ids > $O/trace_gconv_sp2.txt
for r in 1 2; do RD_GCONV_SP2=0 RD_GCONV_SPLIT_TRACE=$r python tools/trace_gconv_split.py --pre 2>&1 | grep -v amdgpu.ids; done > $O/trace_gconv_split_pre8.txt
python tools/ablate_gconv_split.py --pre 2>&1 | grep -v amdgpu.ids > $O/ablate_gconv_sp2.txt
RD_GCONV_SP2=0 python tools/ablate_gconv_split.py --pre --all 2>&1 | grep -v amdgpu.ids > $O/ablate_gconv_split_pre8.txt
python tools/ablate_gconv_split.py --all 2>&1 | grep -v amdgpu.ids > $O/ablate_gconv_split.txt
# ---- rocprofv3: kernel traces (multi-stream = the timed command line; single-stream = per-kernel durations) and HBM counter passes
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
P="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace -d $O/kt -o tr -- $B > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d $O/kt1 -o tr -- $B > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d $O/kt1f -o tr -- $B --operands fp32 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d $O/kt116 -o tr -- $B --storage bf16 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $P > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $P > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetchf -o f -- $P --operands fp32 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/writef -o w -- $P --operands fp32 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch16 -o f -- $P --storage bf16 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write16 -o w -- $P --storage bf16 > /dev/null 2>&1
# MFMA-pipe utilisation / LDS conflicts of the split kernels (separate --pmc passes, --kernel-trace only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcs -o p -- python $R/tools/pmc_split.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcl -o p -- python $R/tools/pmc_split.py > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --steady > $O/kernel_stats.txt
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --steady --timeline > $O/timeline.txt
python tools/exclusive_time.py $(find $O/kt -name "*.db" | head -1) 12 > $O/exclusive_time.txt
{ echo "# collected at git $RD_HEAD: python tools/step_trace.py (one step of the multi-stream kernel trace, per queue; under the profiler the host falls behind -- see tail_probe.txt for the un-profiled stream ends)"; python tools/step_trace.py $(find $O/kt -name "*.db" | head -1); } > $O/step_trace.txt
python tools/rocpd_stats.py $(find $O/kt1 -name "*.db" | head -1) --steady > $O/kernel_stats_single_stream.txt
python tools/rocpd_stats.py $(find $O/kt1f -name "*.db" | head -1) --steady > $O/kernel_stats_fp32_mfma_single_stream.txt
python tools/rocpd_stats.py $(find $O/kt116 -name "*.db" | head -1) --steady > $O/kernel_stats_bf16_storage_single_stream.txt
python tools/pmc_traffic.py --steady $O/fetch $O/write > $O/pmc_traffic.json
python tools/pmc_traffic.py --steady $O/fetchf $O/writef > $O/pmc_traffic_fp32_mfma.json
python tools/pmc_traffic.py --steady $O/fetch16 $O/write16 > $O/pmc_traffic_bf16_storage.json
{ echo "# collected at git $RD_HEAD: rocprofv3 --pmc passes (--kernel-trace only) on tools/pmc_split.py (B=16 3x3 layers: split-while-staging, pre-split sp2, weight gradient)"; python tools/pmc_mfma.py $O/pmcs; python tools/pmc_mfma.py $O/pmcl; } > $O/pmc_split.txt 2>&1
rm -rf $O/kt $O/kt1 $O/kt1f $O/kt116 $O/fetch $O/write $O/fetchf $O/writef $O/fetch16 $O/write16 $O/pmcs $O/pmcl
bash tools/stress_all.sh > $O/stress_all.txt 2>&1; tail -3 $O/stress_all.txt
python tools/stress_split.py > $O/stress_split.txt 2>&1; tail -2 $O/stress_split.txt
python tools/audit_resources.py --spills-only > $O/audit_resources.txt 2>&1; tail -1 $O/audit_resources.txt
for f in c2 c2_fp32_mfma c3 c4 c5; do python - <<P
import json
d=json.load(open("gpurun_out/$RD_ROUND/bench_$f.json")); r=d.get("roofline",{})
print("$f", d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), (d.get("alt_fp32_mfma") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
P
done
head -14 $O/kernel_stats_single_stream.txt
