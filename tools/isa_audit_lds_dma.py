"""Build gate for the kernels that refill LDS with inline-asm LDS-DMA (global_load_lds_dwordx4: conv_sp.hip).
hipcc's waitcnt pass does not see those loads, so the hand-over of a refilled slot rests on a hand-written `s_waitcnt vmcnt(N)` in
front of the workgroup barrier that publishes it.  This walks the compiled ISA of every kernel in layout order and fails when an
`s_barrier` is reached while an LDS-DMA piece has been issued since the last `s_waitcnt` that names vmcnt - e.g. after a compiler
upgrade moved the wait, or an edit dropped it.  (Layout order, not a control-flow walk: the refill code of these kernels is
straight-line between barriers; a wait on ANY path in between is accepted, which is why the counted value itself is reviewed by hand.)
    python tools/isa_audit_lds_dma.py <file.s>      exit status 0 = clean"""
import re
import sys


def audit(path):
    bad = []
    kernel, pending_at = None, None
    n_dma = n_bar = 0
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel, pending_at = m.group(1), None
            continue
        if t.startswith(".Lfunc_end"):
            kernel, pending_at = None, None
            continue
        if kernel is None or not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in t):
            pending_at = ln if pending_at is None else pending_at
            n_dma += 1
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
