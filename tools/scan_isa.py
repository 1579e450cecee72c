"""Compile every csrc/*.hip to gfx950 assembly and print, per kernel: VGPRs, scratch bytes, instruction count, MFMAs, v_cndmask, SGPR-spill lane
operations, s_nop.  (LABNOTES 8.8: a scratch access is a vector-memory operation on the same in-order counter the operand loads wait on; a private
array indexed at run time, or a pointer test that keeps values alive, shows up here before it shows up in a profile.)

    python tools/scan_isa.py            # kernels with scratch or more than 2000 instructions
    python tools/scan_isa.py --all [-D...]   # every kernel; extra flags go to hipcc
Column `wbl2!`: agent-scope release fences (buffer_wbl2 sc1) NOT followed by an `s_waitcnt vmcnt(0)` before the next barrier / store / atomic / end of
the kernel -- MI355X_MICROARCH.md "Compiler hazard (ROCm 7.2, gfx950)": the compiler may drop that wait, and a flag store can then overtake the
write-back.  Any non-zero count makes the script exit 1 (tests/test_host.py runs `unguarded_release_fences` on the 3 x 3 kernel's file).
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


def unguarded_release_fences(lines):
    """Count `buffer_wbl2` instructions of one kernel body (full instruction lines) that reach a barrier, a store, an atomic or the end of the kernel
    without an `s_waitcnt vmcnt(0)` in between."""
    bad = 0
    for i, ins in enumerate(lines):
        if not ins.startswith("buffer_wbl2"):
            continue
        ok = False
        for nxt in lines[i + 1:]:
            op = nxt.split()[0]
            if op == "s_waitcnt" and re.search(r"vmcnt\(0\)", nxt):
                ok = True
                break
            if op in ("s_barrier", "s_endpgm") or op.startswith(("global_store", "global_atomic", "buffer_store", "buffer_atomic", "flat_store", "flat_atomic")):
                break
        bad += 0 if ok else 1
    return bad


def kernels(path):
    name, body, full, v, out = None, [], [], None, []
    for line in open(path):
        m = re.match(r"\s*\.globl\s+(\S+)", line)
        if m and "kernel" in m.group(1):
            name, body, full, v = m.group(1), [], [], None
        if name and line.startswith("\t") and not line.startswith(("\t.", "\t;")):
            body.append(line.split()[0])
            full.append(line.strip())
        m = re.match(r"; NumVgprs: (\d+)", line)
        if m and name:
            v = int(m.group(1))
        m = re.match(r"; ScratchSize: (\d+)", line)
        if m and name:
            c = Counter(body)
            out.append(dict(name=name, vgprs=v, scratch=int(m.group(1)), n=len(body), mfma=sum(x for k, x in c.items() if "mfma" in k),
                            cndmask=sum(x for k, x in c.items() if "cndmask" in k), lane=sum(x for k, x in c.items() if "readlane" in k or "writelane" in k),
                            nop=c["s_nop"], wbl2=c["buffer_wbl2"], wbl2_bad=unguarded_release_fences(full)))
            name = None
    return out


def main():
    show_all = "--all" in sys.argv
    flags = [a for a in sys.argv[1:] if a.startswith("-") and a != "--all"]
    tmp = tempfile.mkdtemp(prefix="vmm_isa_")
    srcs = [s for s in sorted(glob.glob(os.path.join(b.CSRC, "*.hip"))) if b.EXPERIMENTS or os.path.basename(s) not in b.EXPERIMENT_SOURCES]

    def comp(src):
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, "-w", *flags, "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
        return src, out, r.returncode, r.stderr

    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(comp, srcs))
    print(f"{'file':18s} {'kernel':64s} {'vgprs':>5s} {'scr':>4s} {'instr':>6s} {'mfma':>5s} {'cnd':>5s} {'lane':>5s} {'nop':>4s} {'wbl2':>4s} {'wbl2!':>5s}")
    bad_total = 0
    for src, out, rc, err in results:
        if rc != 0:
            print(os.path.basename(src), "FAILED", err[-300:])
            continue
        for k in kernels(out):
            bad_total += k["wbl2_bad"]
            if show_all or k["scratch"] > 0 or k["n"] > 2000 or k["wbl2_bad"]:
                nm = k["name"].replace("_ZN12_GLOBAL__N_1", "")[:64]
                print(f"{os.path.basename(src)[:-4]:18s} {nm:64s} {k['vgprs']:5d} {k['scratch']:4d} {k['n']:6d} {k['mfma']:5d} {k['cndmask']:5d} {k['lane']:5d} {k['nop']:4d} {k['wbl2']:4d} {k['wbl2_bad']:5d}")
    if bad_total:
        print(f"{bad_total} agent-scope release fence(s) without an s_waitcnt vmcnt(0) before the hand-over")
        sys.exit(1)


if __name__ == "__main__":
    main()
