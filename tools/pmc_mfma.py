#!/usr/bin/env python3
"""MFMA-pipe utilisation and instruction mix per kernel from a rocprofv3 counter pass.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU \\
              SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o p -- python tools/pmc_conv.py
    python tools/pmc_mfma.py $OUT > profiles/r01_pmc_conv.txt
MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the
8 XCDs, the busy counter over the 1024 SIMDs of the device."""
import csv
import glob
import os
import re
import sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
assert files, "no counter csv under " + sys.argv[1]
disp = {}
for f in files:
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            key = (row["Dispatch_Id"], re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void rd::", "").replace("rd::", ""), row["Grid_Size"])
            disp.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
agg = {}
for (_, name, grid), c in disp.items():
    a = agg.setdefault((name, grid), [0, {}])
    a[0] += 1
    for k, v in c.items():
        a[1][k] = a[1].get(k, 0.0) + v
print("# rocprofv3 --pmc pass (--kernel-trace only) on tools/pmc_conv.py: B=16 forward conv + wgrad of layer1 3x3 64->64 @113x200,")
print("# layer2 3x3 128 @57x100, layer4 3x3 512 @15x25.  Averages per dispatch.  MFMA pipe utilisation =")
print("# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)   [GRBM_GUI_ACTIVE is summed over the 8 XCDs]")
for (name, grid), (n, c) in sorted(agg.items(), key=lambda kv: -kv[1][1].get("GRBM_GUI_ACTIVE", 0)):
    if "gconv" not in name and "wgrad" not in name and "wino" not in name:
        continue
    g = {k: v / n for k, v in c.items()}
    if "SQ_LDS_IDX_ACTIVE" in g:       # the LDS pass (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE): extra cycles / all LDS-array cycles
        print("%s grid=%s: LDS bank conflict cycles / LDS active cycles = %.4f" % (name, grid, g.get("SQ_LDS_BANK_CONFLICT", 0) / max(g["SQ_LDS_IDX_ACTIVE"], 1)))
        continue
    act = g.get("GRBM_GUI_ACTIVE", 0) / 8.0
    wc = max(g.get("SQ_WAVE_CYCLES", 0), 1.0)
    print("%s grid=%s (%d dispatches)" % (name, grid, n))
    print("   MFMA_pipe_util %.1f%%  GRBM_GUI_ACTIVE/8 %.3g cyc  SQ_INSTS_VALU %.3g  SQ_INSTS_LDS %.3g  SQ_INSTS_SALU %.3g  "
          "SQ_ACTIVE_INST_ANY %.1f%%  SQ_WAIT_INST_ANY %.1f%% (of SQ_WAVE_CYCLES)" % (
              100.0 * g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(act * 1024, 1), act, g.get("SQ_INSTS_VALU", 0), g.get("SQ_INSTS_LDS", 0),
              g.get("SQ_INSTS_SALU", 0), 100.0 * g.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100.0 * g.get("SQ_WAIT_INST_ANY", 0) / wc))
