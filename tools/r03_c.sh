# round 3, call c: offline plan table, reduce batching granularity, bench with / without the table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
B="python bench.py --no-cpu-baseline --no-roofline --steps 30"
for n in 0 2 4 8 99; do echo "reduce_batch=$n $(RD_TUNED_TABLE=0 RD_WGRAD_REDUCE_BATCH=$n $B 2>/dev/null | tail -1 | cut -c88-190)"; done > $O/reduce_batch.txt; cat $O/reduce_batch.txt
timeout 1500 python tools/make_tuned_table.py $O/tuned_plans.json > $O/make_tuned_table.txt 2>&1; tail -3 $O/make_tuned_table.txt
cp $O/tuned_plans.json radar_depth_amd/tuned_plans.json
for t in 0 1 0 1; do echo "table=$t $(RD_TUNED_TABLE=$t $B 2>/dev/null | tail -1 | cut -c88-190)"; done > $O/table_effect.txt; cat $O/table_effect.txt
M="--config 4"
for t in 0 1; do echo "config4 table=$t $(RD_TUNED_TABLE=$t $B $M 2>/dev/null | tail -1 | cut -c60-170)"; done >> $O/table_effect.txt; tail -2 $O/table_effect.txt
