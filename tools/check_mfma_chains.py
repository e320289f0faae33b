#!/usr/bin/env python3
"""ISA lint of a built libenarf_hip.so (or variant): every v_mfma in every kernel must accumulate in a hardware-safe form.

Rejected (see csrc/enarf_query.h, "split-precision MLP", and profiles/r02_mfma_chain_hazard.md):
  * vDst partially overlapping SrcC (same registers shifted);
  * a CHAINED MFMA - its SrcC is the vDst of an MFMA at most `WINDOW` instructions earlier - whose vDst differs from its
    SrcC. The compiler emits these with no wait state in between; on gfx950 the second one does not reliably see the
    first result. In-place chains (vDst == SrcC) are the form every GEMM uses and are accepted, as is an MFMA whose SrcC
    comes from anything but a recent MFMA.
  * an MFMA result READ TOO EARLY: between an MFMA and the first later instruction that reads or writes any register of its
    vDst - other than an MFMA that takes the vDst whole as its SrcC, in place (the accumulate chain: 0 states) - there must be
    at least passes + 4 wait states (issue slots; `s_nop N` = N + 1): the rule the inline-asm blocks' trailing `s_nop`s are
    written to (cdna_hip_programming.md 5.7 item 2: "an MFMA's D -> any reader or writer except the next MFMA taking it
    whole as C ... 8-pass XDL: 12 states"; hipcc pads nothing inside an asm string). v_mfma_f32_16x16x4_f32 is 8 passes (32
    clk per instruction per SIMD), v_mfma_f32_16x16x32_{f16,bf16} 4 passes on gfx950 (16 clk, MI355X_MICROARCH.md cycle
    table): 12 and 8 states. Counted conservatively: an intervening MFMA counts as ONE state. The scan follows the
    fall-through path (a conditional branch is one state) and ends at an unconditional branch or the end of the program:
    what it guards are the hand-written blocks, whose pads sit inside the block, ahead of any branch; across blocks of
    compiler-generated code hipcc's own hazard recogniser is in charge.
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


def hw_passes(op):
    """passes of the instruction on gfx950 (for the result-read distance)"""
    if "16x16x4_f32" in op:
        return 8
    if "16x16x32" in op:
        return 4
    return 16


REGTOK = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))\b")
CTRL = re.compile(r"\s(s_branch|s_endpgm|s_setpc_b64|s_swappc_b64)\b")


def touched(line):
    """(file, lo, hi) register ranges an instruction names (operands only: the text before the encoding comment)"""
    body = line.split("//")[0]
    out = []
    for m in REGTOK.finditer(body):
        if m.group(2) is not None:
            out.append((m.group(1), int(m.group(2)), int(m.group(3))))
        else:
            out.append((m.group(1), int(m.group(4)), int(m.group(4))))
    return out


def check_read_distance(kernel, lines, problems):
    """every MFMA's vDst against the first later non-chain instruction that names one of its registers"""
    ins = []
    for line in lines:
        m = MFMA.search(line)
        n = NOP.search(line)
        ins.append((line, m, (int(n.group(1)) + 1) if n else 1, bool(CTRL.search(line))))
    for i, (line, m, _, _) in enumerate(ins):
        if not m:
            continue
        d = regs(m.group(2))
        need = hw_passes(m.group(1)) + 4
        states = 0
        for j in range(i + 1, len(ins)):
            l2, m2, st2, ctrl2 = ins[j]
            if m2:
                d2, c2 = regs(m2.group(2)), regs(m2.group(5))
                if c2 == d and d2 == d:
                    break                                  # in-place accumulate chain: the chain's last MFMA is checked instead
            if ctrl2:
                break                                      # leaves the straight line: not followed
            if any(f == d[0] and not (hi < d[1] or d[2] < lo) for f, lo, hi in touched(l2)):
                if states < need:
                    problems.append((kernel, line.strip(), f"vDst touched after {states} wait states (needs {need}) by: {l2.strip()[:60]}"))
                break
            states += st2
            if states >= need:
                break


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
            kernel, recent, clock, body = None, [], 0, []
            for line in txt.splitlines() + ["0 <end>:"]:
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    if kernel and body:
                        check_read_distance(kernel, body, problems)
                    kernel, recent, clock, body = m.group(1), [], 0, []
                    n_kernels += 1
                    continue
                if not kernel or ":" not in line:
                    continue
                body.append(line)
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
