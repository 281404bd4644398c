# round 3, first GPU call: GPU tests, bench line, host issue time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json; cut -c1-600 $O/bench.json
for m in fp32 bf16s; do python tools/host_time.py 2 97 161 $m 2>/dev/null | grep "host issue"; python tools/host_time.py 16 450 800 $m 2>/dev/null | grep "host issue"; done > $O/host_time.txt; cat $O/host_time.txt
