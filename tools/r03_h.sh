R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gconv.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_conv.py 150 7 2>&1 | tail -3
python tools/fuzz_conv.py 80 8 --poison 2>&1 | tail -2
for a in "4 113 200 64 128 25 2" "4 57 100 128 256 25 2" "2 30 50 128 128 25 updgrad" "2 15 25 256 256 25 updgrad" "2 30 50 128 128 25 up"; do echo "== $a"; python tools/stress_plans.py $a 2>&1 | grep -c " ok$"; python tools/stress_plans.py $a 2>&1 | grep "FLAKY\|WRONG\|candidates"; done > $O/stress.txt 2>&1; cat $O/stress.txt
for n in s2_64 s2_128; do echo "=== $n"; RD_TUNED_TABLE=0 RD_GCONV_TRACE=1 python tools/trace_gconv.py $n 2>&1 | grep -E "^kernel|totals" ; done
RD_TUNED_TABLE=0 python tools/bench_ops.py 2>/dev/null | grep "s2\|up5x5\|TOTAL"
