"""Build gate for the kernels that refill LDS with inline-asm LDS-DMA (global_load_lds_dwordx4: conv_sp.hip).
hipcc's waitcnt pass does not see those loads, so the hand-over of a refilled slot rests on a hand-written `s_waitcnt vmcnt(N)` in
front of the workgroup barrier that publishes it.  This walks the compiled ISA of every kernel in layout order and fails when an
`s_barrier` is reached while an LDS-DMA piece has been issued since the last `s_waitcnt` that names vmcnt - e.g. after a compiler
upgrade moved the wait, or an edit dropped it.  (Layout order, not a control-flow walk: the refill code of these kernels is
straight-line between barriers; a wait on ANY path in between is accepted, which is why the counted value itself is reviewed by hand.)
Second check: the refill taps keep ONE M0 value live across up to four LDS-DMA pieces that are separate asm statements (offset:0 /
1024 / 2048 / 3072 move the global and the LDS address together), and nothing tells hipcc that M0 is live in between.  So: the most recent
write of M0 in front of a piece WITHOUT an instruction offset must be the kernel's own `s_mov_b32 m0, sN` + `s_nop 0` pair, and between that
write and a piece WITH an offset there must be the group's offset-free piece and no other M0 write - a compiler-inserted M0 write inside a
group, or a piece issued behind a foreign M0 value, fails the build.
    python tools/isa_audit_lds_dma.py <file.s>      exit status 0 = clean"""
import re
import sys


def audit(path):
    bad = []
    kernel, pending_at = None, None
    n_dma = n_bar = 0
    # M0 state: m0_own = the last M0 write was `s_mov_b32 m0, sN` directly followed by `s_nop 0`; m0_base = an offset-free piece has been
    # issued since that write; m0_line = where it was written
    m0_own, m0_base, m0_line, expect_nop = False, False, None, False
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel, pending_at = m.group(1), None
            m0_own, m0_base, m0_line, expect_nop = False, False, None, False
            continue
        if t.startswith(".Lfunc_end"):
            kernel, pending_at = None, None
            continue
        if kernel is None or not t or t.startswith(";") or t.startswith("."):
            continue
        code = t.split(";")[0]
        op = code.split()[0]
        if expect_nop:
            m0_own = op == "s_nop"
            expect_nop = False
        # any instruction whose first operand is m0 writes it (s_mov_b32 m0, ..., s_add_u32 m0, ..., v_readfirstlane_b32 m0, ...)
        ops = code[len(op):].split(",")
        if ops and ops[0].strip() == "m0" and not op.startswith("s_cmp") and not op.startswith("s_bitcmp"):
            m0_own, m0_base, m0_line = False, False, ln
            expect_nop = op == "s_mov_b32" and len(ops) > 1 and re.match(r"^s\d+$", ops[1].strip()) is not None
        if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in t):
            pending_at = ln if pending_at is None else pending_at
            n_dma += 1
            has_off = re.search(r"offset:\s*[1-9]", code) is not None
            if not m0_own:
                bad.append(f"{path}:{ln}: LDS-DMA piece in {kernel} behind an M0 write (line {m0_line}) that is not the kernel's own s_mov_b32 m0, sN + s_nop 0")
            elif has_off and not m0_base:
                bad.append(f"{path}:{ln}: LDS-DMA piece with an instruction offset in {kernel}, but no offset-free piece since M0 was written (line {m0_line}): M0 was rewritten inside the group")
            if not has_off:
                m0_base = True
        elif op == "s_waitcnt" and "vmcnt" in t:
            pending_at = None
        elif op == "s_barrier":
            n_bar += 1
            if pending_at is not None:
                bad.append(f"{path}:{ln}: s_barrier in {kernel} with an LDS-DMA piece (line {pending_at}) not covered by an s_waitcnt vmcnt")
                pending_at = None
    return bad, n_dma, n_bar


if __name__ == "__main__":
    errs, n_dma, n_bar = audit(sys.argv[1])
    if n_dma == 0:
        errs.append(f"{sys.argv[1]}: no LDS-DMA instruction found - wrong file?")
    for e in errs:
        print(e, file=sys.stderr)
    print(f"{sys.argv[1]}: {n_dma} LDS-DMA pieces, {n_bar} barriers, {len(errs)} findings")
    sys.exit(1 if errs else 0)
