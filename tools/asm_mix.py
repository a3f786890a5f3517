#!/usr/bin/env python3
"""Instruction mix of device functions in a hipcc -S listing.  usage: tools/asm_mix.py file.s substring [substring ...]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
labels = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for idx, (i, name) in enumerate(labels):
    if not any(s in name for s in sys.argv[2:]):
        continue
    end = labels[idx + 1][0] if idx + 1 < len(labels) else len(lines)
    ins = []
    for l in lines[i:end]:
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        ins.append(t.split()[0])
        if t.startswith('s_endpgm') or t.startswith('s_setpc'):
            pass
    c = Counter(ins)
    g = Counter()
    for k, v in c.items():
        if k.startswith('v_pk'): g['v_pk'] += v
        elif k.startswith('v_'): g['valu'] += v
        elif k.startswith('s_waitcnt'): g['waitcnt'] += v
        elif k.startswith('s_barrier'): g['barrier'] += v
        elif k.startswith('s_'): g['salu'] += v
        elif k.startswith('ds_'): g['lds'] += v
        elif k.startswith('global_') or k.startswith('flat_'): g['vmem'] += v
        elif k.startswith('scratch_') or k.startswith('buffer_'): g['scratch'] += v
        else: g[k] += v
    print(name[:70], len(ins), dict(g))
    print("    top:", c.most_common(16))
