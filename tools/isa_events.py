"""Compressed event sequence (MFMA / waitcnt / loads / LDS / barriers / branches) of one kernel in a hipcc -S listing.
    python tools/isa_events.py file.s <mangled-name-substring> [max_lines]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith('E'))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
def kind(l):
    l = l.strip()
    if l.startswith('v_mfma'): return 'MFMA'
    if l.startswith('s_waitcnt'): return l.split(';')[0].strip()
    if l.startswith('global_load') or l.startswith('buffer_load'): return 'GLOAD'
    if l.startswith('global_store') or l.startswith('global_atomic'): return 'GSTORE'
    if l.startswith('ds_read') or l.startswith('ds_load'): return 'DSR'
    if l.startswith('ds_write') or l.startswith('ds_store'): return 'DSW'
    if l.startswith('s_barrier'): return 'BARRIER'
    if l.startswith('v_cvt_f16'): return 'CVT'
    if l.startswith('v_exp') : return 'EXP'
    if l.startswith('s_cbranch') or l.startswith('s_branch'): return ' '.join(l.split()[:2])
    if re.match(r'\.LBB\d+_\d+:', l): return l
    return None
out = []; last = None; cnt = 0
for l in lines[start:end]:
    k = kind(l)
    if k is None: continue
    if k == last: cnt += 1
    else:
        if last: out.append(f"{last} x{cnt}" if cnt > 1 else last)
        last = k; cnt = 1
out.append(f"{last} x{cnt}")
print(end - start, "lines")
print('\n'.join(out[:int(sys.argv[3]) if len(sys.argv) > 3 else 10**9]))
