R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
for dbg in 0 264 520 776 1288 2056; do for n in layer2 layer1 layer4 up256 s2_64 dec1c2; do echo "debug=$dbg $n $(RD_TUNED_TABLE=0 RD_GCONV_DEBUG=$dbg RD_GCONV_TRACE=1 python tools/trace_gconv.py $n 2>&1 | grep -E "^kernel [0-9.]+ us|totals" | tr '\n' ' ' | cut -c1-300)"; done; done > $O/dephase.txt 2>&1
cat $O/dephase.txt
