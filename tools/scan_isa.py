"""Compile every csrc/*.hip to gfx950 assembly and print, per kernel: VGPRs, scratch bytes, instruction count, MFMAs, v_cndmask, SGPR-spill lane
operations, s_nop.  (LABNOTES 8.8: a scratch access is a vector-memory operation on the same in-order counter the operand loads wait on; a private
array indexed at run time, or a pointer test that keeps values alive, shows up here before it shows up in a profile.)

    python tools/scan_isa.py            # kernels with scratch or more than 2000 instructions
    python tools/scan_isa.py --all [-D...]   # every kernel; extra flags go to hipcc
No GPU needed (hipcc cross-compiles)."""
import glob
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videometamaterials_amd import build as b  # noqa: E402


def kernels(path):
    name, body, v, out = None, [], None, []
    for line in open(path):
        m = re.match(r"\s*\.globl\s+(\S+)", line)
        if m and "kernel" in m.group(1):
            name, body, v = m.group(1), [], None
        if name and line.startswith("\t") and not line.startswith(("\t.", "\t;")):
            body.append(line.split()[0])
        m = re.match(r"; NumVgprs: (\d+)", line)
        if m and name:
            v = int(m.group(1))
        m = re.match(r"; ScratchSize: (\d+)", line)
        if m and name:
            c = Counter(body)
            out.append(dict(name=name, vgprs=v, scratch=int(m.group(1)), n=len(body), mfma=sum(x for k, x in c.items() if "mfma" in k),
                            cndmask=sum(x for k, x in c.items() if "cndmask" in k), lane=sum(x for k, x in c.items() if "readlane" in k or "writelane" in k),
                            nop=c["s_nop"]))
            name = None
    return out


def main():
    show_all = "--all" in sys.argv
    flags = [a for a in sys.argv[1:] if a.startswith("-") and a != "--all"]
    tmp = tempfile.mkdtemp(prefix="vmm_isa_")
    srcs = sorted(glob.glob(os.path.join(b.CSRC, "*.hip")))

    def comp(src):
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, "-w", *flags, "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        return src, out, r.returncode, r.stderr

    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(comp, srcs))
    print(f"{'file':18s} {'kernel':64s} {'vgprs':>5s} {'scr':>4s} {'instr':>6s} {'mfma':>5s} {'cnd':>5s} {'lane':>5s} {'nop':>4s}")
    for src, out, rc, err in results:
        if rc != 0:
            print(os.path.basename(src), "FAILED", err[-300:])
            continue
        for k in kernels(out):
            if show_all or k["scratch"] > 0 or k["n"] > 2000:
                nm = k["name"].replace("_ZN12_GLOBAL__N_1", "")[:64]
                print(f"{os.path.basename(src)[:-4]:18s} {nm:64s} {k['vgprs']:5d} {k['scratch']:4d} {k['n']:6d} {k['mfma']:5d} {k['cndmask']:5d} {k['lane']:5d} {k['nop']:4d}")


if __name__ == "__main__":
    main()
