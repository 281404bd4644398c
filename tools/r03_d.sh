# round 3, call d: offline plan table (fixed enumeration), its effect, reduce batching under bf16 storage
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
B="python bench.py --no-cpu-baseline --no-roofline --steps 30"
timeout 2400 python tools/make_tuned_table.py $O/tuned_plans.json > $O/make_tuned_table.txt 2>&1; tail -3 $O/make_tuned_table.txt
cp $O/tuned_plans.json radar_depth_amd/tuned_plans.json
for t in 0 1 0 1; do echo "table=$t $(RD_TUNED_TABLE=$t $B 2>/dev/null | tail -1 | cut -c88-190)"; done > $O/table_effect.txt; cat $O/table_effect.txt
for t in 0 1 0 1; do echo "config4 table=$t $(RD_TUNED_TABLE=$t $B --config 4 2>/dev/null | tail -1 | cut -c60-170)"; done >> $O/table_effect.txt; tail -4 $O/table_effect.txt
for n in 0 4 99 0 4 99; do echo "bf16s reduce_batch=$n $(RD_WGRAD_REDUCE_BATCH=$n $B --config 3 2>/dev/null | tail -1 | cut -c88-400 | grep -o '"value": [0-9.]*, .*"ms_per_step": [0-9.]*')"; done > $O/reduce_batch_bf16s.txt; cat $O/reduce_batch_bf16s.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q -k batched_slab 2>&1 | tail -2
