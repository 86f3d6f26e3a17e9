"""Static instruction histogram of one kernel in a hipcc -save-temps .s file.   python tools/isa_hist.py <file.s> <mangled-name regex> [n]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for i, l in enumerate(txt):
    m = re.match(r'^(\w+):', l)
    if m and pat.search(m.group(1)) and not l.startswith('.'):
        start = i; name = m.group(1); break
else:
    sys.exit("kernel not found")
print(name)
end = next(j for j in range(start, len(txt)) if 's_endpgm' in txt[j])
cnt = collections.Counter()
for l in txt[start:end]:
    l = l.split(';')[0].strip()
    if not l or l.endswith(':') or l.startswith('.'): continue
    cnt[l.split()[0]] += 1
print('static instructions', sum(cnt.values()))
for k, v in cnt.most_common(top): print(f'{k:32s} {v}')
for l in txt[end:end + 80]:
    if re.search(r'NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize', l): print(l.strip())
