# Round-3 evidence for the split-bf16 fp32 convolution path (operands="split"): bench lines, rocprofv3 kernel stats + PMC traffic of
# the same command line, per-layer table, workgroup traces, MFMA rate microbenchmarks.  Run on the GPU box: bash tools/collect_split.sh
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/split; mkdir -p $O
cd $R
python bench.py --steps 40 --warmup 10 --operands split 2>/dev/null | tail -1 > $O/bench_c2_split.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_fp32_same_box.json
python bench.py --config 4 --steps 30 --warmup 8 --operands split --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_split.json
python tools/bench_split.py > $O/bench_split.txt 2>&1
python tools/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids > $O/bench_wgrad_split.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_c2_default_with_alt.json
RD_GCONV_SPLIT_ALL=1 python tools/bench_split.py > $O/bench_split_all_shapes.txt 2>&1
for r in 1 2; do RD_GCONV_SPLIT_TRACE=$r python tools/trace_gconv_split.py 2>&1 | grep -v amdgpu.ids; done > $O/trace_gconv_split.txt
tools/micro/mfma_bf16_peak > $O/mfma_bf16_peak.txt 2>&1
tools/micro/mfma_agpr > $O/mfma_agpr.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --operands split"
rocprofv3 --kernel-trace -d $O/kt -o tr -- $B > /dev/null 2>&1
P="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --operands split"
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $P > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $P > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --steady > $O/kernel_stats_split.txt
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --steady --timeline > $O/timeline_split.txt
python tools/pmc_traffic.py --steady $O/fetch $O/write > $O/pmc_traffic_split.json
rm -rf $O/kt $O/fetch $O/write
head -12 $O/kernel_stats_split.txt
python -c "
import json
for f in ('bench_c2_split','bench_c2_fp32_same_box','bench_c4_split'):
    d=json.load(open('gpurun_out/split/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))
"
