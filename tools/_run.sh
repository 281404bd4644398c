cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_bf16_storage.py -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/w1_tests.log
cat gpurun_out/w1_tests.log
