#!/usr/bin/env python3
"""Per-kernel register / spill / scratch table from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(radar_depth_amd/build.py keeps them next to every object: csrc/build/<source>.resource.txt).

    python tools/audit_resources.py [--spills-only] [source.hip ...]
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "radar_depth_amd", "csrc", "build")
FIELDS = ["VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"^void rd::|^void |\(.*$", "", o) for o in out[:len(names)]]
    except OSError:
        return names


def parse(path):
    """-> [(kernel, {field: int})] in file order."""
    rows, cur = [], None
    for ln in open(path, errors="replace"):
        m = re.search(r"remark: (?:\S+: )?Function Name: (\S+)", ln)
        if m:
            cur = (m.group(1), {})
            rows.append(cur)
            continue
        if cur is None:
            continue
        for f in FIELDS:
            m = re.search(r"remark: (?:\S+: )?\s*" + re.escape(f) + r": (\d+)", ln)
            if m:
                cur[1][f] = int(m.group(1))
    names = demangle([r[0] for r in rows])
    return [(n, r[1]) for n, r in zip(names, rows)]


def main():
    spills_only = "--spills-only" in sys.argv
    srcs = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(f[:-len(".resource.txt")] for f in os.listdir(BUILD) if f.endswith(".resource.txt"))
    bad = 0
    for s in srcs:
        rows = parse(os.path.join(BUILD, s + ".resource.txt"))
        print("# %s: %d kernels" % (s, len(rows)))
        for name, r in rows:
            sp = r.get("VGPRs Spill", 0) + r.get("SGPRs Spill", 0)
            bad += r.get("VGPRs Spill", 0) > 0
            if spills_only and not sp:
                continue
            print("%-64s vgpr %3d agpr %3d  vgpr_spill %3d sgpr_spill %3d scratch %4d B/lane  occupancy %d  lds %6d" % (
                name[:64], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0),
                r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", 0)))
    print("# kernels with VGPR spills: %d" % bad)


if __name__ == "__main__":
    main()
