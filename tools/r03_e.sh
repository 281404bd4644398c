R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
for n in layer2 layer4 up256 up128 up32 d2 dec1c2 dec3c2 s2_64 s2_128 fusion; do echo "=== $n"; RD_GCONV_TRACE=1 python tools/trace_gconv.py $n 2>&1 | grep -v "^  CU" ; done > $O/trace_gconv.txt 2>&1
cat $O/trace_gconv.txt | head -150
