# round 3, call b: GPU tests (new parity tests + batched reduce + unspilled groups), planner sweep, bench + per-layer table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json
RD_WGRAD_REDUCE_BATCH=0 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-250 > $O/bench_nobatch.json; cat $O/bench_nobatch.json
python tools/bench_ops.py > $O/bench_ops_per_layer.txt 2>/dev/null; tail -3 $O/bench_ops_per_layer.txt
timeout 1500 python tools/sweep_plan_layers.py > $O/sweep_plan_layers.txt 2>/dev/null; grep "<--\|sum over" $O/sweep_plan_layers.txt
