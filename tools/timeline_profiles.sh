R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace -d $O/kt -o tr -- $B --no-autotune > /dev/null 2>&1

cd $R
for k in kt; do
  db=$(find $O/$k -name "*.db" | head -1)
  python tools/rocpd_stats.py $db --timeline > $O/${k}_timeline.txt
  python tools/rocpd_stats.py $db --gaps > $O/${k}_gaps.txt
  python tools/rocpd_stats.py $db > $O/${k}_stats.txt
done
rm -rf $O/kt
cat $O/kt_timeline.txt
