# end-of-round collection ON THE GPU BOX (started by tools/gpu_collect.sh, which refuses a dirty tree and passes the commit in RD_HEAD):
# the FULL GPU test suite (RD_SLOW=1: the cases the default run skips included); kernel traces + counter passes of the default bench command
# line (the split plan) and of its alternates; every bench line; per-layer tables; the Winograd gate; host issue time; stress; resource audit.
# Output: gpurun_out/$RD_ROUND/.   (The round-5 experiments -- slot-map A/B, bf16-storage kernel ablations, fill-rate / unaligned-LDS
# microbenchmarks, stem studies, workgroup traces of the split kernels -- stay on file as profiles/r05_*: those kernels did not change.)
R=$GRAFT_REPO_ROOT; RD_ROUND=${RD_ROUND:-r06}; export RD_ROUND; export RD_HEAD=${RD_HEAD:-$(cat $R/.collect_head 2>/dev/null || echo unknown)}
O=$R/gpurun_out/$RD_ROUND; mkdir -p $O; cd $R
echo "collecting $RD_ROUND at $RD_HEAD"
RD_SLOW=1 timeout 1800 python -m pytest tests -m gpu -q -s --durations=30 > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
{ echo "# collected at git $RD_HEAD: RD_SLOW=1 python -m pytest tests -m gpu -q (the FULL suite; the default -m gpu run skips the tests marked slow)"; grep -aE "passed|failed|error|^[0-9.]+s (call|setup)" $O/pytest.txt | tail -40; } > $O/pytest_full.txt
# the margins the parity tests print: worst / median errors per plan, fp64-anchored gradients, the decision-floor table over seeds, the three-step trajectory
{ echo "# collected at git $RD_HEAD: margins printed by RD_SLOW=1 pytest -m gpu -s (tests/test_gpu_margins.py, tests/test_gpu_configs.py, tests/test_gpu_wino.py)"; grep -aE "fp64-anchored|three steps|gradient norms|config4 grad|plain multistage|seed [0-9]+:|clean seeds|wino [0-9]" $O/pytest.txt | sed 's/^[.sF]*//'; } > $O/parity_margins.txt
# ---- bench lines (JSON, one per file)
python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
python bench.py --operands fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_fp32_mfma.json
python bench.py --mode eager --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_eager.json
RD_WINO=0 python bench.py --no-cpu-baseline --no-alt --no-clock 2>/dev/null | tail -1 > $O/bench_c2_no_winograd.json
python bench.py --config 3 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 4 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 5 2>/dev/null | tail -1 > $O/bench_c5.json
RD_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1 > $O/bench_c2_dp1_torchrun.json
# Winograd A/B in one job on one box (boxes of the pool differ by several percent: only same-job pairs compare)
{ echo "# collected at git $RD_HEAD: Winograd layers on (default planner rule) / off (RD_WINO=0) / everywhere supported (RD_WINO=all), same job, same box"
  for i in 1 2; do for w in default 0 all; do
    if [ $w = default ]; then unset RD_WINO; else export RD_WINO=$w; fi
    python bench.py --no-cpu-baseline --no-alt --no-roofline --no-clock 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('RD_WINO=%-8s: %8.1f samples/s  step %.3f ms' % ('$w', d['value'], d['ms_per_step']))"
  done; done; unset RD_WINO; } > $O/wino_ab.txt 2>&1
# ---- per-layer tables and kernel-level studies
python tools/bench_ops.py > $O/bench_ops_per_layer.txt 2>/dev/null
python tools/bench_split_pre.py 2>&1 | grep -v amdgpu.ids > $O/bench_split_pre.txt
python tools/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids > $O/bench_wgrad_split.txt
{ echo "# collected at git $RD_HEAD: python tools/bench_wino.py; python tools/ablate_wino.py (the shipped kernel; the history of the gate is profiles/r06_wino_gate.txt)"; python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids; python tools/ablate_wino.py 2>&1 | grep -v amdgpu.ids; } > $O/wino_kernel.txt
python tools/bench_bn.py fp32 2>/dev/null | grep "ch " | cut -c1-330 > $O/bench_bn.txt
# host issue time: plain launches vs hipGraph replay of the SPLIT plan, configs 2 and 4
{ echo "# collected at git $RD_HEAD: bench.py host_issue_ms_per_step; plain launches (default) vs --graph (hipGraph replay), split plan"
  for cfg in 2 4; do for g in "" "--graph"; do
    python bench.py --config $cfg $g --no-cpu-baseline --no-alt --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('config $cfg %-8s: %8.1f samples/s  step %.3f ms  host issue %.3f ms' % ('$g' or 'plain', d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step') or float('nan')))"
  done; done
} > $O/host_time.txt 2>&1
{ echo "# collected at git $RD_HEAD: python tools/tail_probe.py <config> 12 (RD_TAIL_EVENTS=1: timing events recorded from the op list of an UN-PROFILED step)"
  for c in 2 4; do echo "## config $c"; python tools/tail_probe.py $c 12 2>&1 | grep -v "amdgpu.ids\|Info"; done; } > $O/tail_probe.txt
# ---- rocprofv3: kernel traces (multi-stream = the timed command line; single-stream = per-kernel durations) and HBM counter passes
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-alt"
P="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-alt"
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
python tools/rocpd_stats.py $(find $O/kt1 -name "*.db" | head -1) --steady > $O/kernel_stats_single_stream.txt
python tools/rocpd_stats.py $(find $O/kt1f -name "*.db" | head -1) --steady > $O/kernel_stats_fp32_mfma_single_stream.txt
python tools/rocpd_stats.py $(find $O/kt116 -name "*.db" | head -1) --steady > $O/kernel_stats_bf16_storage_single_stream.txt
python tools/pmc_traffic.py --steady $O/fetch $O/write > $O/pmc_traffic.json
python tools/pmc_traffic.py --steady $O/fetchf $O/writef > $O/pmc_traffic_fp32_mfma.json
python tools/pmc_traffic.py --steady $O/fetch16 $O/write16 > $O/pmc_traffic_bf16_storage.json
{ echo "# collected at git $RD_HEAD: rocprofv3 --pmc passes (--kernel-trace only) on tools/pmc_split.py (B=16 3x3 layers: split-while-staging, pre-split sp2, weight gradient, Winograd)"; python tools/pmc_mfma.py $O/pmcs; python tools/pmc_mfma.py $O/pmcl; } > $O/pmc_split.txt 2>&1
rm -rf $O/kt $O/kt1 $O/kt1f $O/kt116 $O/fetch $O/write $O/fetchf $O/writef $O/fetch16 $O/write16 $O/pmcs $O/pmcl
bash tools/stress_all.sh > $O/stress_all.txt 2>&1; tail -3 $O/stress_all.txt
python tools/stress_split.py > $O/stress_split.txt 2>&1; tail -2 $O/stress_split.txt
python tools/audit_resources.py --spills-only > $O/audit_resources.txt 2>&1; tail -1 $O/audit_resources.txt
for f in c2 c2_fp32_mfma c2_eager c3 c4 c5; do python - <<P
import json
d=json.load(open("gpurun_out/$RD_ROUND/bench_$f.json")); r=d.get("roofline",{})
print("$f", d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), (r.get("shader_clock_mhz") or {}).get("mean"), (d.get("alt_fp32_mfma") or {}).get("value"), (d.get("alt_eager") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
P
done
head -14 $O/kernel_stats_single_stream.txt
