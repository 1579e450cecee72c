"""Build libvmm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m videometamaterials_amd.build [--force]
    VMM_EXPERIMENTS=1 python -m videometamaterials_amd.build     # -> libvmm_hip_exp.so (select it with VMM_LIB_PATH)

The product library holds the kernels a plan can launch.  The experiments that were built, are parity-green and lost their A/B inside the captured
step -- the Winograd F(2x2, 3x3) convolution (conv3x3_wino.hip, include/vmm_experiments.h), the persistent wave-specialised 3 x 3 kernel and the
three-workgroups-per-CU instance (-DVMM_EXPERIMENTS=1 sections of conv3x3_bf16x3.hip) -- are compiled only into the second library.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
EXPERIMENTS = os.environ.get("VMM_EXPERIMENTS", "0") not in ("", "0")
OUT = os.path.join(HERE, "libvmm_hip_exp.so" if EXPERIMENTS else "libvmm_hip.so")
OBJDIR = os.path.join(HERE, "_obj_exp" if EXPERIMENTS else "_obj")
EXPERIMENT_SOURCES = {"conv3x3_wino.hip"}
# compiled a second time with -DVMM_SINGLE_PASS=1 (object *_sp.o): the `_bf16` entry points of the backward kernels (vmm_common.h, VMM_X3)
# (object suffix, VMM_SINGLE_PASS value): 1 = bf16-rounded operands (`_bf16` entry points), 2 = fp16-rounded operands (`_fp16`: the reference's autocast dtype)
SINGLE_PASS_MODES = (("_sp", 1), ("_h", 2))
# forward kernels with a single-pass template instance: compiled once more with -DVMM_SINGLE_PASS=2 (object *_h.o), exporting that instance on fp16 operands
# alone (`vmm_conv3x3_fp16`, `vmm_conv_s2_acc_fp16`, `vmm_temporal_block_fp16`, `vmm_linattn_block_fp16`)
FP16_FORWARD_SOURCES = {"conv3x3_bf16x3.hip", "temporal_block.hip", "linattn_block.hip"}
# EXPERIMENTS build: the sampler's fused attention blocks compiled once more with -DVMM_SPLIT_F16=1 (object *_f3.o): three passes on IEEE-half hi | lo operands,
# `_f16x3` entry points (include/vmm_experiments.h)
SPLIT_F16_SOURCES = {"temporal_block.hip", "linattn_block.hip"}
SINGLE_PASS_SOURCES = {"temporal_block_bwd.hip", "linattn_block_bwd.hip", "qkv_bwd.hip", "wgrad3x3_bf16x3.hip", "wgrad1x1_bf16x3.hip", "igemm_bf16x3.hip"}
# -munsafe-fp-atomics: hardware fp32 atomic add (valid for the coarse-grained device memory all buffers live in) instead of a CAS loop
# -fno-slp-vectorize: the SLP vectoriser packs adjacent scalar fp32 adds / multiplies into v_pk_*_f32, which issue at 40 % of their rate beside a busy
#   matrix pipe (55 % for the scalar forms; MI355X_MICROARCH.md "price of one filler beside MFMAs", LABNOTES 7.6).  Measured on one box, libraries
#   alternating (LABNOTES 10.2): captured sampling step 13.31 -> 13.08 ms, training step 33.97 -> 33.75 ms; it also removes the compiler-made packed
#   sequences of the kind LABNOTES 9.8 suspected behind the rotated-token-key corruption (the hand-written v_pk intrinsics stay).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
if EXPERIMENTS:
    FLAGS.append("-DVMM_EXPERIMENTS=1")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_changed(cmd) -> bool:
    """An object is also stale when the command that made it differs from the one about to run (a changed FLAGS list or per-object define must not leave
    a library linked from objects of two flag sets): the command line is kept beside the object as <object>.cmd."""
    stamp, want = cmd[-1] + ".cmd", " ".join(cmd)
    try:
        return open(stamp).read() != want
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [s for s in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if EXPERIMENTS or os.path.basename(s) not in EXPERIMENT_SOURCES]
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    os.makedirs(OBJDIR, exist_ok=True)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
        if force or _stale(o, [s] + hdrs) or _flags_changed(cmd):
            jobs.append(cmd)
        if EXPERIMENTS and os.path.basename(s) in SPLIT_F16_SOURCES:  # (experiments library only: measured, worth 0-4 % of a block, not used by any plan -- LABNOTES 11.2)
            o3 = o[:-2] + "_f3.o"
            objs.append(o3)
            cmd = [hipcc, *FLAGS, "-DVMM_SPLIT_F16=1", "-c", s, "-o", o3]
            if force or _stale(o3, [s] + hdrs) or _flags_changed(cmd):
                jobs.append(cmd)
        if os.path.basename(s) in SINGLE_PASS_SOURCES or os.path.basename(s) in FP16_FORWARD_SOURCES:
            for tag, mode in (SINGLE_PASS_MODES if os.path.basename(s) in SINGLE_PASS_SOURCES else SINGLE_PASS_MODES[1:]):
                o2 = o[:-2] + tag + ".o"
                objs.append(o2)
                cmd = [hipcc, *FLAGS, f"-DVMM_SINGLE_PASS={mode}", "-c", s, "-o", o2]
                if force or _stale(o2, [s] + hdrs) or _flags_changed(cmd):
                    jobs.append(cmd)
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose:
                    print(" ".join(os.path.basename(x) if x.endswith((".hip", ".o")) else x for x in cmd))
                if res.returncode != 0:
                    sys.stderr.write(res.stdout + res.stderr)
                    raise RuntimeError(f"hipcc failed on {cmd[-3]}")
                with open(cmd[-1] + ".cmd", "w") as f:
                    f.write(" ".join(cmd))
    if jobs or not os.path.exists(OUT):
        res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT], capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
