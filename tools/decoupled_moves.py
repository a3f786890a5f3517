#!/usr/bin/env python3
"""How much a demodulator taken out of the per-packet chain would have to redo: from the oracle's per-step traces of the bench workloads, the
`fine_sync` moves per packet, and - for a resolve that keeps every (symbol, cumulative offset) result it has - the streaming launches a packet needs and the
windows they demodulate relative to one pass over the symbols.  CPU only: python tools/decoupled_moves.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O
for sf, packets in ((7, 256), (8, 128), (9, 64), (10, 64)):
    cfg, iq, offs, lens, expect = bench.make_workload(sf, 4, packets, 32, 8, seed=(2 if sf==7 else 100*sf+4))
    tot_sym=0; work=0; rounds=[]; moves_tot=0; npk=0; maxoff=0
    for s in range(len(offs)):
        o = O.Oracle(sf=sf); o.enable_trace(True)
        o.run(iq[offs[s]:offs[s]+lens[s]])
        tr = o.trace()
        cur=[]; pk=[]
        for t in tr:
            if t[0] in (4,5): cur.append(t[4])
            else:
                if cur: pk.append(cur); cur=[]
        if cur: pk.append(cur)
        for fines in pk:
            npk+=1
            pay=fines   # header + payload symbols alike (the header could be streamed too once the position is known)
            n=len(pay); tot_sym+=n
            mv=sum(1 for f in pay if f!=0); moves_tot+=mv
            # cache-aware scheme: a launch computes, for one offset value d, every symbol from index `start` on; the resolve walks the symbols with the
            # cumulative offset and stops where the needed (symbol, offset) is not computed yet
            have={}  # offset -> first symbol index computed at that offset
            r=0; i=0; off=0
            while i<n:
                if off not in have or have[off]>i:
                    r+=1; have[off]=i; work+=n-i
                # walk
                while i<n and off in have and have[off]<=i:
                    off+= -pay[i]   # d_fine_sync moves the next window; sign irrelevant here
                    i+=1
                maxoff=max(maxoff,abs(off))
            rounds.append(r)
    print('sf',sf,'packets',npk,'symbols',tot_sym,'moves %.2f %%'%(100*moves_tot/tot_sym),'work factor %.3f'%(work/tot_sym),'launch rounds mean %.2f max %d'%(np.mean(rounds),max(rounds)),'hist',np.bincount(rounds)[:8],'max |offset|',maxoff,flush=True)
