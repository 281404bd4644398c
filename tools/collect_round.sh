# end-of-round collection on the GPU box:  RD_ROUND=r03 bash tools/collect_round.sh
# full GPU test suite, kernel traces + counter passes (tools/collect_profiles.sh), every bench line (tools/collect_bench.sh), race hunt
R=$GRAFT_REPO_ROOT; RD_ROUND=${RD_ROUND:-r03}; export RD_ROUND; O=$R/gpurun_out/$RD_ROUND; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
bash tools/collect_profiles.sh > $O/collect_profiles.log 2>&1
bash tools/collect_bench.sh
for m in fp32 bf16s; do python tools/host_time.py 2 97 161 $m 2>/dev/null | grep "host issue"; python tools/host_time.py 16 450 800 $m 2>/dev/null | grep "host issue"; done > $O/host_time.txt
bash tools/stress_all.sh > $O/stress_all.txt 2>&1; cat $O/stress_all.txt
python tools/audit_resources.py --spills-only > $O/audit_resources.txt 2>&1; tail -1 $O/audit_resources.txt
