# ON THE GPU BOX: A/B of the slot map (RD_GCONV_SPLIT_NATURAL=1: round-4 row-major slots at the same LDS pitch) -- parity tests, per-layer
# table, LDS-conflict counters, bench line.  Output: gpurun_out/slotmap/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/slotmap; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gconv_split.py tests/test_gpu_model.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/bench_split_pre.py 2>&1 | grep -v amdgpu.ids > $O/bench_split_pre_new.txt; tail -1 $O/bench_split_pre_new.txt
RD_GCONV_SPLIT_NATURAL=1 python tools/bench_split_pre.py 2>&1 | grep -v amdgpu.ids > $O/bench_split_pre_natural.txt; tail -1 $O/bench_split_pre_natural.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 > $O/bench_new_$i.json; python -c "import json;d=json.load(open('$O/bench_new_$i.json'));print('new', d['value'], d['ms_per_step'])"
RD_GCONV_SPLIT_NATURAL=1 python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 > $O/bench_natural_$i.json; python -c "import json;d=json.load(open('$O/bench_natural_$i.json'));print('natural', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcl -o p -- python $R/tools/pmc_split.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmcs -o p -- python $R/tools/pmc_split.py > /dev/null 2>&1
cd $R
{ python tools/pmc_mfma.py $O/pmcs; python tools/pmc_mfma.py $O/pmcl; } > $O/pmc_split.txt 2>&1
rm -rf $O/pmcs $O/pmcl
cat $O/pmc_split.txt | grep -v "^#"
