#!/bin/bash
# One job, one box: the training step with the 1x1 layers on gemm1_split_kernel (default) vs on rd_gconv's fp32 MFMA kernel
# (RD_GEMM1_SPLIT=0), interleaved three times.  Output: gpurun_out/gemm1_ab.txt
mkdir -p gpurun_out
{
for i in 1 2 3; do
  for v in 1 0; do
    echo "RD_GEMM1_SPLIT=$v"
    RD_GEMM1_SPLIT=$v python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])"
  done
done
} > gpurun_out/gemm1_ab.txt 2>&1
cat gpurun_out/gemm1_ab.txt
