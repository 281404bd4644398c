R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout 2400 python tools/make_tuned_table.py $O/tuned_plans.json > $O/make_tuned_table.txt 2>&1; tail -2 $O/make_tuned_table.txt; grep -c ", 2) *KEEP" $O/make_tuned_table.txt
cp $O/tuned_plans.json radar_depth_amd/tuned_plans.json
B="python bench.py --no-cpu-baseline --no-roofline --steps 30"
for t in 0 1 0 1; do echo "table=$t $(RD_TUNED_TABLE=$t $B 2>/dev/null | tail -1 | cut -c88-190)"; done > $O/table_effect.txt; cat $O/table_effect.txt
for t in 0 1; do echo "config4 table=$t $(RD_TUNED_TABLE=$t $B --config 4 2>/dev/null | tail -1 | cut -c60-170)"; done >> $O/table_effect.txt; tail -2 $O/table_effect.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
