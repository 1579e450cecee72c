"""Build libvmm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m videometamaterials_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvmm_hip.so")
OBJDIR = os.path.join(HERE, "_obj")
# -munsafe-fp-atomics: hardware fp32 atomic add (valid for the coarse-grained device memory all buffers live in) instead of a CAS loop
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-munsafe-fp-atomics"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    os.makedirs(OBJDIR, exist_ok=True)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose:
                    print(" ".join(os.path.basename(x) if x.endswith((".hip", ".o")) else x for x in cmd))
                if res.returncode != 0:
                    sys.stderr.write(res.stdout + res.stderr)
                    raise RuntimeError(f"hipcc failed on {cmd[-3]}")
    if jobs or not os.path.exists(OUT):
        res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT], capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
