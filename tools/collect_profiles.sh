set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${RD_ROUND:-r03}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"   # (heuristic plans: the default)
rocprofv3 --kernel-trace -d $O/kt -o tr -- $B > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d $O/kt1 -o tr -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace -d $O/kt16 -o tr -- $B --storage bf16 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace -d $O/kt161 -o tr -- $B --storage bf16 > /dev/null 2>&1
P="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $P > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $P > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch16 -o f -- $P --storage bf16 > /dev/null 2>&1
RD_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write16 -o w -- $P --storage bf16 > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --steady > $O/kernel_stats.txt
python tools/rocpd_stats.py $(find $O/kt1 -name "*.db" | head -1) --steady > $O/kernel_stats_single_stream.txt
python tools/rocpd_stats.py $(find $O/kt16 -name "*.db" | head -1) --steady > $O/kernel_stats_bf16_storage.txt
python tools/rocpd_stats.py $(find $O/kt161 -name "*.db" | head -1) --steady > $O/kernel_stats_bf16_storage_single_stream.txt
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) --steady --timeline > $O/timeline.txt
python tools/rocpd_stats.py $(find $O/kt16 -name "*.db" | head -1) --steady --timeline > $O/timeline_bf16_storage.txt
python tools/pmc_traffic.py --steady $O/fetch $O/write > $O/pmc_traffic.json
python tools/pmc_traffic.py --steady $O/fetch16 $O/write16 > $O/pmc_traffic_bf16_storage.json
rm -rf $O/kt16 $O/kt161 $O/kt $O/kt1 $O/fetch $O/write $O/fetch16 $O/write16
head -20 $O/kernel_stats_single_stream.txt
python - <<'P'
import json, os
for f in ("pmc_traffic.json","pmc_traffic_bf16_storage.json"):
    d=json.load(open("gpurun_out/"+os.environ.get("RD_ROUND","r03")+"/"+f))["kernels"]
    print(f, len(d), list(d.items())[:2])
P
