#!/usr/bin/env python3
"""ISA lint of a built libenarf_hip.so (or variant): every v_mfma in every kernel must accumulate in a hardware-safe form.

Rejected (see csrc/enarf_query.h, "split-precision MLP", and profiles/r02_mfma_chain_hazard.md):
  * vDst partially overlapping SrcC (same registers shifted);
  * a CHAINED MFMA - its SrcC is the vDst of an MFMA at most `WINDOW` instructions earlier - whose vDst differs from its
    SrcC. The compiler emits these with no wait state in between; on gfx950 the second one does not reliably see the
    first result. In-place chains (vDst == SrcC) are the form every GEMM uses and are accepted, as is an MFMA whose SrcC
    comes from anything but a recent MFMA.
Usage: python tools/check_mfma_chains.py [path/to/lib.so]    exit code 1 on a violation."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MFMA = re.compile(r"\s(v_mfma_\w+)\s+(\S+), (\S+), (\S+), (\S+?)(\s|$)")
NOP = re.compile(r"\ss_nop\s+(\d+)")


def passes(op):
    """matrix-pipe passes (4 clk each) of the MFMA forms this library uses; unknown forms count as 16"""
    if "16x16x4_f32" in op:
        return 8
    if "16x16x32" in op:
        return 8          # 4 on gfx950 (double rate); 8 keeps a margin
    return 16


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return tok[0], int(m.group(1)), int(m.group(2))
    m = re.match(r"[va](\d+)$", tok)
    if m:
        return tok[0], int(m.group(1)), int(m.group(1))
    return None          # inline constant / literal


def check(path):
    work = tempfile.mkdtemp(prefix="enarf_isa_")
    try:
        lib = os.path.join(work, os.path.basename(path))
        shutil.copy(path, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True)
        objs = sorted(f for f in os.listdir(work) if "amdgcn" in f)
        if not objs:
            raise SystemExit(f"{path}: no gfx950 code object found")
        problems, n_mfma, n_kernels = [], 0, 0
        for o in objs:
            txt = subprocess.run([OBJDUMP, "-d", os.path.join(work, o)], check=True, capture_output=True, text=True).stdout
            kernel, recent, clock = None, [], 0
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    kernel, recent, clock = m.group(1), [], 0
                    n_kernels += 1
                    continue
                if not kernel or ":" not in line:
                    continue
                m = MFMA.search(line)
                if not m:
                    n = NOP.search(line)
                    clock += (int(n.group(1)) + 1) if n else 1          # issue slots ("wait states")
                    continue
                n_mfma += 1
                op = m.group(1)
                d, c = regs(m.group(2)), regs(m.group(5))
                if c is not None and c[0] == d[0]:
                    overlap = not (d[2] < c[1] or c[2] < d[1])
                    if overlap and (d[1], d[2]) != (c[1], c[2]):
                        problems.append((kernel, line.strip(), "vDst partially overlaps SrcC"))
                    for (pd, pclock, ppasses) in recent:
                        gap = clock - pclock          # issue slots between the end of the producer's issue and this MFMA
                        if pd == c and d != c and gap < ppasses + 4:
                            problems.append((kernel, line.strip(), f"chained on an MFMA {gap} issue slots back ({ppasses} passes) but not in place"))
                            break
                clock += passes(op)                   # the matrix pipe is busy for the instruction's passes
                recent = [(pd, pc, pp) for (pd, pc, pp) in recent if clock - pc <= 64 and pd != d] + [(d, clock, passes(op))]
        return n_kernels, n_mfma, problems
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    target = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "enarf-gan_amd", "csrc", "libenarf_hip.so")
    k, n, bad = check(target)
    print(f"{target}: {n} MFMA instructions in {k} symbols, {len(bad)} unsafe")
    for kern, ins, why in bad[:40]:
        print(f"  {kern[:70]}: {ins}   <- {why}")
    sys.exit(1 if bad else 0)
