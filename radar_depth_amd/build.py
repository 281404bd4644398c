"""Build libradardepth_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m radar_depth_amd.build [--force]

Objects are cached under radar_depth_amd/csrc/build/ keyed by source mtime; the shared library is
written in-tree to radar_depth_amd/lib/ (git-ignored, but it travels to the GPU box with gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libradardepth_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("RD_EXTRA_FLAGS", "").split()      # (build-time experiments: RD_EXTRA_FLAGS="-DRD_MMA_ORDER=1" python -m radar_depth_amd.build --force)
# every object keeps hipcc's per-kernel register / spill / scratch report next to it (<source>.resource.txt): the ISA audit
# (tools/audit_resources.py, tests/test_abi_host.py) reads those instead of recompiling
RES_FLAG = "-Rpass-analysis=kernel-resource-usage"


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(src, obj, headers):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(p) > t for p in [src] + headers)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "radar_depth_hip.h"))
    jobs, objs = [], []
    for f in _sources():
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or _stale(src, obj, headers) or not os.path.exists(os.path.join(OBJ, f + ".resource.txt")):
            lang = ["-x", "hip"] if f.endswith(".cpp") else []
            jobs.append((f, [HIPCC] + FLAGS + [RES_FLAG] + lang + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        text = r.stdout + r.stderr
        if r.returncode == 0:
            with open(os.path.join(OBJ, name + ".resource.txt"), "w") as fh:
                fh.write("".join(ln + "\n" for ln in text.splitlines() if RES_FLAG in ln))
        # the resource remarks are not diagnostics: only real warnings / errors are shown
        shown, after_remark = [], False
        for ln in text.splitlines():
            if RES_FLAG in ln or "remarks generated" in ln:
                after_remark = True
                continue
            if after_remark and ln.lstrip()[:1].isdigit() is False and "|" in ln[:12]:
                continue                                   # the "      | ^" caret line under a remark
            if after_remark and ln.split("|")[0].strip().isdigit():
                continue                                   # the quoted source line under a remark
            after_remark = False
            shown.append(ln + "\n")
        return name, r.returncode, "".join(shown)

    failed = False
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for name, rc, out in ex.map(run, jobs):
            if verbose and (rc != 0 or out.strip()):
                print("[build] %s rc=%d\n%s" % (name, rc, out))
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc failed")
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print("[build] %s (%d objects rebuilt)" % (LIB, len(jobs)))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
