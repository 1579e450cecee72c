"""Measurement aid: a second library with extra -D flags on selected sources, for A/B runs of two builds on ONE box (box-to-box
variance of the captured step is ~3 %):   python tools/build_ab.py conv3x3_bf16x3 -DVMM_C3_XCD_ORDER=0   (`all` = every source)
writes videometamaterials_amd/libvmm_hip_ab.so (git-ignored; select it with VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so)."""
import glob
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videometamaterials_amd import build as b  # noqa: E402

if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("-")]
    flags = [a for a in sys.argv[1:] if a.startswith("-")]
    b.build(verbose=False)
    objs = []
    for s in sorted(glob.glob(os.path.join(b.CSRC, "*.hip"))):
        if not b.EXPERIMENTS and os.path.basename(s) in b.EXPERIMENT_SOURCES:
            continue
        stem = os.path.basename(s)[:-4]
        o = os.path.join(b.OBJDIR, stem + ".o")
        if stem in names or "all" in names:
            o = os.path.join("/tmp", stem + "_ab.o")
            subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, "-w", *flags, "-c", s, "-o", o], check=True)
        objs.append(o)
        base = os.path.basename(s)
        if b.EXPERIMENTS and base in b.SPLIT_F16_SOURCES:  # (*_f3.o: the fp16 hi | lo three-pass objects of the experiments build)
            o3 = os.path.join(b.OBJDIR, stem + "_f3.o")
            if stem in names or "all" in names:
                o3 = os.path.join("/tmp", stem + "_f3_ab.o")
                subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, "-w", *flags, "-DVMM_SPLIT_F16=1", "-c", s, "-o", o3], check=True)
            objs.append(o3)
        if base in b.SINGLE_PASS_SOURCES or base in b.FP16_FORWARD_SOURCES:  # the single-pass objects of build.py (*_sp.o: bf16 operands, *_h.o: fp16 operands)
            for tag, mode in (b.SINGLE_PASS_MODES if base in b.SINGLE_PASS_SOURCES else b.SINGLE_PASS_MODES[1:]):
                o2 = os.path.join(b.OBJDIR, stem + tag + ".o")
                if stem in names or "all" in names:
                    o2 = os.path.join("/tmp", stem + tag + "_ab.o")
                    subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, "-w", *flags, f"-DVMM_SINGLE_PASS={mode}", "-c", s, "-o", o2], check=True)
                objs.append(o2)
    out = os.path.join(os.path.dirname(b.OUT), "libvmm_hip_ab.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
    print(out)
