R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gconv.py -x -q -m gpu -k "bnbwd" 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_norm.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --steps 30"
for t in 0 1 0 1; do echo "fuse_bn_bwd=$t $(RD_FUSE_BN_BWD=$t $B 2>/dev/null | tail -1 | cut -c88-190)"; done > $O/fuse.txt; cat $O/fuse.txt
for t in 0 1; do echo "config4 fuse_bn_bwd=$t $(RD_FUSE_BN_BWD=$t $B --config 4 2>/dev/null | tail -1 | cut -c60-170)"; done >> $O/fuse.txt; tail -2 $O/fuse.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench.json
python - <<'P'
import json
r=json.load(open("gpurun_out/r03j/bench.json"))["roofline"]
print(r["eager_ms_by_family"]); print(list(r["eager_ms_by_kernel"].items())[:4])
P
