#!/usr/bin/env python3
"""usage: isa_blocks.py file.s <mangled-substring> [min_instrs] -- instruction mix of the big basic blocks of one kernel"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r'^(_Z\S*' + re.escape(key) + r'\S*):', s, re.M)
a = m.start(); b = s.index('.Lfunc_end', a)
blocks = []; cur = ['entry', []]; blocks.append(cur)
for ln in s[a:b].split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):', ln)
    if mm: cur = [mm.group(1) + (' (loop)' if 'Loop' in ln else ''), []]; blocks.append(cur)
    elif ln.startswith('\t') and not ln.strip().startswith(('.', ';')): cur[1].append(ln.strip())
for nm, ins in blocks:
    if len(ins) >= mn:
        c = Counter(i.split()[0] for i in ins)
        print(nm, len(ins), dict(c.most_common(12)))
