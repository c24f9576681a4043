"""Build libsemabs_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python semantic-abstraction_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libsemabs_hip.so")
TUNE_LIB = os.path.join(OUT_DIR, "libsemabs_hip_tune.so")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC]
# files whose integer outputs are bit-exact targets: no FMA contraction, IEEE division
PER_FILE = {"geometry.hip": ["-ffp-contract=off"], "tiles.hip": ["-ffp-contract=off"], "relio.hip": ["-ffp-contract=off"]}


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, tuning: bool = False) -> str:
    """tuning=True: the same sources with -DSEMABS_TUNING -> lib/libsemabs_hip_tune.so (ablation switches, alternative tile configurations,
    per-workgroup traces; used by tools/ only, selected with SEMABS_TUNE_LIB=1).  The production library carries none of it."""
    if tuning:
        return _build(force, verbose, "libsemabs_hip_tune.so", "_tune.o", ["-DSEMABS_TUNING"])
    return _build(force, verbose, "libsemabs_hip.so", ".o", [])


def _build(force, verbose, libname, osuffix, extra) -> str:
    LIB = os.path.join(OUT_DIR, libname)
    os.makedirs(OUT_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    cc = hipcc()
    jobs = []
    objs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OUT_DIR, f.replace(".hip", osuffix))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append((f, [cc, "-c", src, "-o", obj] + COMMON + extra + PER_FILE.get(f, [])))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r.returncode, r.stdout + r.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name, rc, log in ex.map(run, jobs):
                if verbose and log.strip():
                    print(log)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {name}:\n{log}")
                if verbose:
                    print(f"[semabs build] compiled {name}")
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[semabs build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
