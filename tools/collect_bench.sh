# every bench line kept under profiles/ (run on the GPU box: bash tools/collect_bench.sh; results in gpurun_out/bench/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bench; mkdir -p $O; cd $R
python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
python bench.py --config 3 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 4 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 5 2>/dev/null | tail -1 > $O/bench_c5.json
python bench.py --operands bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16_operands.json
python bench.py --config 4 --height 900 --width 1600 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_fp32_900.json
python tools/bench_ops.py > $O/bench_ops_per_layer.txt 2>/dev/null
python tools/bench_stem.py 2>/dev/null | grep stem > $O/bench_stem.txt
for f in c2 c3 c4 c5; do python - <<P
import json
d=json.load(open("gpurun_out/bench/bench_$f.json")); r=d.get("roofline",{})
print("$f", d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("step_frac_of_bound"), (d.get("cpu_baseline") or {}).get("value"))
P
done
