#!/usr/bin/env python3
"""Audit of conv_dma.hip's hand-counted weight-fragment loads in the compiled ISA.

The kernel requests its weight fragments with inline-asm `global_load_dwordx4` (hidden from hipcc's s_waitcnt bookkeeping on
purpose) and waits for them with inline-asm `s_waitcnt vmcnt(N)`.  hipcc treats an asm load's destination as written when the
statement ends, so NOTHING may read, copy, spill or overwrite those registers until the counted wait that covers them has executed.
This script walks the control-flow graph of every conv_dma_kernel instantiation (every label is re-walked for every distinct
set of pending registers that can reach it: fall-through, forward branches, loop back-edges) with an in-order model of the
vector-memory queue and fails on:
  * any instruction that touches a still-pending destination register,
  * pending registers at s_endpgm,
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


def parse(lines):
    """-> list of (kind, payload, text): kind in {label, branch, cbranch, end, asmload, asmdma, asmwait, wait0, other}"""
    out, in_asm = [], False
    for raw in lines:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"): in_asm = True; continue
        if st.startswith(";;#ASMEND"): in_asm = False; continue
        ln = raw.split(";")[0].strip()
        if not ln: continue
        m = re.match(r"^(\.?LBB\w+):", ln)
        if m: out.append(("label", m.group(1), ln)); continue
        m = re.match(r"^s_branch\s+(\.?LBB\w+)", ln)
        if m: out.append(("branch", m.group(1), ln)); continue
        m = re.match(r"^s_cbranch\w*\s+(\.?LBB\w+)", ln)
        if m: out.append(("cbranch", m.group(1), ln)); continue
        if "s_endpgm" in ln: out.append(("end", None, ln)); continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", ln)
        touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
        if in_asm and ln.startswith("global_load_lds"): out.append(("asmdma", None, ln)); continue
        if in_asm and ln.startswith("global_load_dwordx4"): out.append(("asmload", (regs_of(toks[0]), touched), ln)); continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ln)
        if in_asm and m: out.append(("asmwait", int(m.group(1)), ln)); continue
        if not in_asm and ln.startswith("s_waitcnt") and "vmcnt(0)" in ln: out.append(("wait0", None, ln)); continue
        out.append(("other", (touched, "scratch_" in ln), ln))
    return out


def audit_kernel(name, lines):
    prog = parse(lines)
    label_at = {p[1]: i for i, p in enumerate(prog) if p[0] == "label"}
    bad, seen = [], set()
    work = [(0, ())]          # (instruction index, queue = tuple of frozensets of pending destination registers, oldest first)
    n_states = 0
    while work:
        pc, queue = work.pop()
        queue = list(queue)
        while pc < len(prog):
            kind, pay, ln = prog[pc]
            pend = frozenset().union(*queue) if queue else frozenset()
            if kind == "label":
                key = (pc, pend)
                if key in seen: break
                seen.add(key); n_states += 1
            elif kind in ("branch", "cbranch"):
                if pay in label_at:
                    work.append((label_at[pay], tuple(queue)))
                if kind == "branch": break
            elif kind == "end":
                if pend: bad.append(f"pending asm-load registers at s_endpgm: {sorted(pend)}")
                break
            elif kind == "asmdma":
                queue.append(frozenset())
            elif kind == "asmload":
                dst, touched = pay
                if (touched - dst) & pend: bad.append(f"asm load reads pending registers: {ln}")
                # re-requesting a still-pending destination is fine: loads return in order, the newer data lands last
                queue = [q - dst for q in queue]
                queue.append(frozenset(dst))
            elif kind == "asmwait":
                while len(queue) > pay: queue.pop(0)
            elif kind == "wait0":
                queue = []
            else:
                touched, scratch = pay
                if scratch: bad.append(f"scratch access: {ln}")
                if touched & pend: bad.append(f"pending asm-load register touched (+{pc}): {ln}")
            pc += 1
        if len(bad) > 20: break
        if n_states > 20000:
            bad.append("state cap reached before the walk finished: coverage is partial")      # a correctness gate must not fail open
            break
    n_loads = sum(1 for p in prog if p[0] == "asmload"); n_waits = sum(1 for p in prog if p[0] == "asmwait")
    bad = sorted(set(bad))
    for b in bad[:12]: print(f"{name}: {b}")
    print(f"{name}: {n_loads} asm fragment loads, {n_waits} counted waits, {n_states} (label, pending-set) states walked, {'OK' if not bad else 'FAILED'}")
    return len(bad)


def audit(path):
    text = open(path).read().split("\n")
    kernels, cur = {}, None
    for ln in text:
        m = re.match(r"^(_ZN2pf15conv_dma_kernel\w+):", ln)
        if m:
            cur = []; kernels[m.group(1)] = cur; continue
        if cur is not None:
            # the body ends at the function-end marker, not at the first s_endpgm: code behind an early exit is audited as well
            if re.match(r"^\.Lfunc_end\d+:", ln) or ln.lstrip().startswith(".size"): cur = None
            else: cur.append(ln)
    if not kernels:
        print("no conv_dma_kernel found"); return 1
    return 1 if sum(audit_kernel(n, l) for n, l in kernels.items()) else 0


if __name__ == "__main__":
    sys.exit(audit(sys.argv[1]))
