#!/usr/bin/env python3
"""Audit of conv_dma.hip's hand-counted weight-fragment loads in the compiled ISA.

The kernel requests its weight fragments with inline-asm `global_load_dwordx4` (hidden from hipcc's s_waitcnt bookkeeping on
purpose) and waits for them with inline-asm `s_waitcnt vmcnt(N)`.  hipcc treats an asm load's destination as written when the
statement ends, so NOTHING may read, copy, spill or overwrite those registers until the counted wait that covers them has executed.
This script replays every basic block of every conv_dma_kernel instantiation in program order with an in-order model of the
vector-memory queue and fails on:
  * any instruction that touches a still-pending destination register,
  * pending registers at a basic-block boundary (a branch or label) - the kernel's contract is that a requested fragment never
    crosses control flow,
  * scratch (spill) traffic anywhere in the kernel.
usage: isa_audit_asm_loads.py <file.s>      (hipcc -save-temps output for gfx950)
"""
import re
import sys


def regs_of(tok):
    """VGPR indices named by an operand token like v12 or v[4:7]."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def audit(path):
    text = open(path).read().split("\n")
    kernels, cur, name = {}, None, None
    for ln in text:
        m = re.match(r"^(_ZN2pf15conv_dma_kernel\w+):", ln)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append(ln)
            if "s_endpgm" in ln:
                cur = None
    if not kernels:
        print("no conv_dma_kernel found"); return 1
    bad = 0
    for name, lines in kernels.items():
        queue = []            # outstanding vector-memory operations, oldest first: set of destination VGPRs (empty for LDS-DMA)
        in_asm = False
        n_loads = n_waits = 0
        for i, raw in enumerate(lines):
            ln = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
            if ln.startswith(";;#ASMSTART"): in_asm = True; continue
            if ln.startswith(";;#ASMEND"): in_asm = False; continue
            if not ln: continue
            if "scratch_" in ln:
                print(f"{name}: scratch access at +{i}: {ln}"); bad += 1
            pend = set().union(*queue) if queue else set()
            if re.match(r"^\.?LBB\w+:", ln) or ln.startswith("s_cbranch") or ln.startswith("s_branch") or "s_endpgm" in ln:
                if pend:
                    print(f"{name}: {len(pend)} pending asm-load registers cross control flow at +{i}: {ln}"); bad += 1
                    queue = []
                continue
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", ln)
            touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
            if in_asm and ln.startswith("global_load_lds"):
                queue.append(set()); continue
            if in_asm and ln.startswith("global_load_dwordx4"):
                dst = regs_of(toks[0])
                if (touched - dst) & pend or dst & pend:
                    print(f"{name}: asm load at +{i} touches pending registers: {ln}"); bad += 1
                queue.append(dst); n_loads += 1; continue
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ln)
            if in_asm and m:
                n = int(m.group(1)); n_waits += 1
                while len(queue) > n: queue.pop(0)
                continue
            if not in_asm and ln.startswith("s_waitcnt") and "vmcnt(0)" in ln:
                queue = []; continue
            if touched & pend:
                print(f"{name}: pending asm-load register touched at +{i}: {ln}"); bad += 1
        print(f"{name}: {n_loads} asm fragment loads, {n_waits} counted waits, {'OK' if not bad else 'FAILED'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(audit(sys.argv[1]))
