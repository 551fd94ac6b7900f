#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing:  python tools/isa_stats.py file.s <name-substring> [--loop]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*:', l) and pat in l)
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
body = s[start:end]
c = Counter()
for l in body:
    l = l.strip()
    if not l or l[0] in ';.' or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
print(s[start][:100], len(body), 'lines')
for k, v in c.most_common(28):
    print('%6d %s' % (v, k))
for l in s[end:end + 60]:
    if any(k in l for k in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize', 'NumSgprs')):
        print(l.strip())
