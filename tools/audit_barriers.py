"""ISA audit (no GPU needed): compile every kernel source to gfx950 assembly and check that each s_barrier is preceded, in the same
basic block, by an s_waitcnt that includes lgkmcnt(0) (csrc/common.h rd_sync / glds_wait put it there explicitly because hipcc's
own wait-count pass dropped it at a loop header: the LDS race fixed in round 2).      python tools/audit_barriers.py"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "radar_depth_amd", "csrc")
bad_total = 0
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + [os.path.join(CSRC, "api.cpp")]):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip", "-I" + CSRC,
               "-I" + os.path.join(ROOT, "include"), src, "-o", tmp.name]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(os.path.basename(src), "does not compile on its own:", r.stderr.strip().splitlines()[-1][:160])
            continue
        lines = open(tmp.name).read().split("\n")
    kern, n, bad = None, 0, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        t = l.strip()
        if t.startswith("s_barrier"):
            n += 1
            j, ok = i - 1, False
            while j > 0:
                u = lines[j].strip()
                j -= 1
                if u == "" or u.startswith(";"):
                    continue
                if u.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):
                    continue                               # cross-lane moves through the LDS crossbar: no LDS memory involved
                if u.endswith(":") or u.startswith("s_cbranch") or u.startswith("s_branch") or u.startswith("ds_") or u.startswith("s_barrier"):
                    break                                  # left the block / an LDS memory operation after the last wait
                if u.startswith("s_waitcnt") and "lgkmcnt(0)" in u:
                    ok = True
                    break
            if not ok:
                bad.append((kern, i + 1))
    bad_total += len(bad)
    print("%-18s %4d barriers, %d without an lgkmcnt(0) wait in front%s" % (os.path.basename(src), n, len(bad), (": " + ", ".join("%s:%d" % (k[:48], ln) for k, ln in bad[:4])) if bad else ""))
print("%d unguarded barriers" % bad_total)
sys.exit(1 if bad_total else 0)
