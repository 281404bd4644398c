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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


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
        if force or _stale(src, obj, headers):
            lang = ["-x", "hip"] if f.endswith(".cpp") else []
            jobs.append((f, [HIPCC] + FLAGS + lang + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r.returncode, r.stdout + r.stderr

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
