#!/usr/bin/env python3
"""Issue-cycle model of a device function: its ISA listing (LISTING=<file>, default /tmp/w3/k.s: `hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/w3/k.s
gr_lora_amd/csrc/lora_kernels.hip`) weighted with the measured
cycles per wave64 instruction (tools/ubench_valu.hip): packed fp32 4.3, plain fp32 fma / mul / add 2.6, DPP / select / compare /
min / max / integer 4.3, v_rcp and v_permlane*_swap 8.2, s_nop N = N + 1.   usage: tools/cycle_model.py <substring of the symbol>"""
import os,re,sys
from collections import Counter
lines=open(os.environ.get('LISTING','/tmp/w3/k.s')).read().split('\n')
labels=[(i,l.split(':')[0]) for i,l in enumerate(lines) if re.match(r'^_Z\w+:',l)]
for idx,(i,name) in enumerate(labels):
    if sys.argv[1] not in name: continue
    end=labels[idx+1][0] if idx+1<len(labels) else len(lines)
    cyc=Counter(); cnt=Counter()
    for l in lines[i:end]:
        t=l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;': continue
        op=t.split()[0]
        dpp = 'dpp' in t or 'quad_perm' in t or 'row_' in t or 'wave_' in t
        if op.startswith('v_pk'): k,c='pk',4.3
        elif op.startswith('v_rcp') or op.startswith('v_permlane') or op.startswith('v_sqrt') or op.startswith('v_rsq'): k,c=op[:10],8.2
        elif dpp: k,c=op+'(dpp)',4.3
        elif re.match(r'v_(fma|fmac|mul|add|sub|subrev|mac)_f32',op): k,c='f32 plain',2.6
        elif op.startswith('v_cndmask'): k,c='cndmask',4.3
        elif op.startswith('v_cmp'): k,c='cmp',4.3
        elif re.match(r'v_(max|min)',op): k,c='minmax',4.3
        elif op.startswith('v_mov'): k,c='mov',4.3
        elif op.startswith('v_'): k,c='other valu '+op,4.3
        elif op=='s_nop': k,c='s_nop',float(t.split()[1])+1
        else: continue
        cyc[k]+=c; cnt[k]+=1
    tot=sum(cyc.values())
    print(name[:60],'total valu-ish cycles',round(tot))
    for k,v in cyc.most_common(25): print('  %-28s n=%4d cyc=%6.0f %4.1f%%'%(k,cnt[k],v,100*v/tot))
