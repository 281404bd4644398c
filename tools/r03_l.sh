R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --steps 30"
for t in 0 1 0 1; do echo "bf16s fuse_bn_bwd=$t $(RD_FUSE_BN_BWD=$t $B --config 3 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, .*"ms_per_step": [0-9.]*')"; done > $O/fuse_bf16s.txt; cat $O/fuse_bf16s.txt
for t in 0 1; do echo "config5 fuse_bn_bwd=$t $(RD_FUSE_BN_BWD=$t $B --config 5 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, .*"ms_per_step": [0-9.]*')"; done >> $O/fuse_bf16s.txt; tail -2 $O/fuse_bf16s.txt
