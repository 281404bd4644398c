# every bench line kept under profiles/ (run on the GPU box: bash tools/collect_bench.sh; results in gpurun_out/bench/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bench; mkdir -p $O; cd $R
MS="--arch resnet18_multistage_uncertainty_fixs --batch 8"
python bench.py 2>/dev/null | tail -1 > $O/bench.json
python bench.py --autotune --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fp32_autotune.json
python bench.py --operands bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16.json
python bench.py --storage bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16s.json
python bench.py $MS --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_fp32.json
python bench.py $MS --autotune --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_fp32_autotune.json
python bench.py $MS --storage bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_bf16s.json
python bench.py $MS --height 900 --width 1600 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_fp32_900.json
python bench.py $MS --height 900 --width 1600 --autotune --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_fp32_900_autotune.json
python bench.py $MS --height 900 --width 1600 --storage bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_bf16s_900.json
python tools/bench_ops.py > $O/bench_ops_per_layer.txt 2>/dev/null
python tools/bench_stem.py 2>/dev/null | grep stem > $O/bench_stem.txt
for m in fp32 bf16 bf16s; do python tools/host_time.py 16 450 800 $m 2>/dev/null | grep "host enqueue"; done > $O/host_time.txt
