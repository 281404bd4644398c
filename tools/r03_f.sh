R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
for kb in 40 20 10; do for n in layer4 up256 up128 d2 d3 dec1c2 fusion; do echo "WSD_KB=$kb $n $(RD_TUNED_TABLE=0 RD_GCONV_WSD_KB=$kb RD_GCONV_TRACE=1 python tools/trace_gconv.py $n 2>&1 | grep -E "^kernel [0-9.]+ us|workgroups,|totals" | tr '\n' ' ' | cut -c1-420)"; done; done > $O/wsd.txt 2>&1
cat $O/wsd.txt
