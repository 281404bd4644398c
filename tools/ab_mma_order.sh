# ON THE GPU BOX: A/B of the order of the six split MFMA terms (build-time switch RD_MMA_ORDER, csrc/common.h).  Output: gpurun_out/mmaorder/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mmaorder; mkdir -p $O; cd $R
run() {
  tag=$1
  python tools/bench_split_pre.py 2>&1 | grep -v amdgpu.ids > $O/bench_split_pre_$tag.txt; tail -1 $O/bench_split_pre_$tag.txt
  python tools/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids > $O/bench_wgrad_split_$tag.txt; tail -2 $O/bench_wgrad_split_$tag.txt
  for i in 1 2; do python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 > $O/bench_${tag}_$i.json; python -c "import json;d=json.load(open('$O/bench_${tag}_$i.json'));print('$tag', d['value'], d['ms_per_step'])"; done
}
run order0
RD_EXTRA_FLAGS="-DRD_MMA_ORDER=1" python -m radar_depth_amd.build --force 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_gconv_split.py tests/test_gpu_wgrad_split.py -m gpu -q -x > $O/pytest_order1.txt 2>&1; tail -3 $O/pytest_order1.txt
run order1
python -m radar_depth_amd.build --force 2>&1 | tail -1
run order0b
