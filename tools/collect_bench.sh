# every bench line kept under profiles/ (run on the GPU box: bash tools/collect_bench.sh; results in gpurun_out/bench/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bench; mkdir -p $O; cd $R
python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json
python bench.py --config 3 2>/dev/null | tail -1 > $O/bench_c3.json
python bench.py --config 4 2>/dev/null | tail -1 > $O/bench_c4.json
python bench.py --config 5 2>/dev/null | tail -1 > $O/bench_c5.json
python bench.py --operands bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16_operands.json
python bench.py --config 4 --height 900 --width 1600 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ms_fp32_900.json
# the launcher path + the data-parallel code path (state broadcast, bucket events, rd_allreduce_bucket, 1/world SGD) on ONE rank
RD_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | tail -1 > $O/bench_c2_dp1_torchrun.json
python tools/bench_ops.py > $O/bench_ops_per_layer.txt 2>/dev/null
python tools/bench_bn.py fp32 2>/dev/null | grep "ch " | cut -c1-330 > $O/bench_bn.txt; python tools/bench_bn.py bf16 2>/dev/null | grep "ch " | cut -c1-330 >> $O/bench_bn.txt
python tools/bench_head.py 2>/dev/null | grep "head\|bilinear" > $O/bench_head.txt
python tools/bench_stem.py 2>/dev/null | grep stem > $O/bench_stem.txt
for f in c2 c3 c4 c5; do python - <<P
import json
d=json.load(open("gpurun_out/bench/bench_$f.json")); r=d.get("roofline",{})
print("$f", d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("step_frac_of_bound"), (d.get("cpu_baseline") or {}).get("value"))
P
done
