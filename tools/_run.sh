cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gconv_split.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/gs_tests.log
cat gpurun_out/gs_tests.log
timeout 600 python bench.py --steps 30 --warmup 8 --operands split 2>gpurun_out/gs_bench_split.err | tail -1 > gpurun_out/gs_bench_split.json
timeout 600 python bench.py --steps 30 --warmup 8 --no-roofline 2>/dev/null | tail -1 > gpurun_out/gs_bench_fp32.json
python - <<'PY'
import json
for f in ("gpurun_out/gs_bench_split.json","gpurun_out/gs_bench_fp32.json"):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("eager_ms_by_family"))
        for k,v in list(d.get("roofline",{}).get("eager_ms_by_kernel",{}).items())[:14]: print("   ",k,v)
    except Exception as e: print(f, "ERR", e); print(open("gpurun_out/gs_bench_split.err").read()[-2000:])
PY
