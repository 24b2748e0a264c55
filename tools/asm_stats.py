#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -S listing: tools/asm_stats.py file.s [name-substring]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r'\s*\.type\s+(\S+),@function', l))]
for k, (i, name) in enumerate(starts):
    if want not in name:
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    c = collections.Counter()
    for l in lines[i:end]:
        l = l.strip()
        if not l or l[0] in '.;/' or l.endswith(':'):
            continue
        op = l.split()[0]
        c['all'] += 1
        for pre, key in (('v_', 'valu'), ('s_', 'salu'), ('ds_', 'lds'), ('buffer_', 'vmem'), ('global_', 'vmem'), ('scratch_', 'scratch')):
            if op.startswith(pre):
                c[key] += 1
        for key in ('v_rcp', 'v_cndmask', 'v_and_or', 'v_fma_f64', 'v_add_f64', 'v_mul_f64', 'v_cmp', 's_waitcnt', 's_cbranch', 'v_mfma', 'v_cvt'):
            if op.startswith(key):
                c[key] += 1
    print(name[-75:], dict(c))
