# race hunt over many geometries and modes (GPU box): bash tools/stress_all.sh > gpurun_out/stress_all.txt
L=resnet18_latefusion; M=resnet18_multistage_uncertainty_fixs
for cfg in "$L 16 450 800 20 fp32" "$L 1 450 800 25 fp32" "$L 3 225 401 25 fp32" "$M 8 450 800 20 fp32" "$M 2 900 1600 20 fp32" "$L 5 97 161 30 fp32" "$M 2 450 800 25 fp32" \
           "$L 16 450 800 20 bf16s" "$M 2 450 800 25 bf16s" "$L 3 225 401 25 bf16s" "$L 2 450 800 25 bf16" "$M 2 225 400 25 bf16"; do
  python tools/stress_desc.py $cfg 2>&1 | grep -E "FLAKY|distinct|Error|error|Traceback" | cut -c1-250
done
