#!/usr/bin/env python3
"""Static scan of the score kernels' assembly (hipcc -S --cuda-device-only of a csrc/mtm_mfma_*.hip unit) for register spills
whose STORE sits inside a vector-divergent region (s_and_saveexec ... s_or_b64 exec) that is closed again where the slot is
RELOADED.  For a lane-private value this is harmless (the lane that reloads is a lane that stored); for the accumulators of the
LDS transposition - written by lanes (q == stage) on behalf of OTHER lanes' pixels - it is how round 5's uint16 bug happened
(DESIGN 9).  Heuristic: it flags candidates to read, it does not prove a defect.   Usage: tools/spill_scan.py file.s [...]"""
import re, sys
# Spill stores executed inside a vector-divergent region (s_and_saveexec ... s_or_b64 exec) whose slot is reloaded at a point
# where that region is no longer open: the reload's lanes may include lanes that never stored.
def scan(path):
    out=[]; cur=None
    for ln,line in enumerate(open(path),1):
        m=re.match(r'^(_ZN3mtm15ncc_mfma_kernel\S+):', line)
        if m:
            cur={'name':m.group(1),'stack':[], 'rid':0, 'stores':{}, 'risky':[], 'nsp':0}
            continue
        if cur is None: continue
        if line.startswith('.Lfunc_end'):
            out.append(cur); cur=None; continue
        m=re.search(r's_(?:and|or|xor|andn2)_saveexec_b64 (s\[\d+:\d+\])', line)
        if m:
            cur['rid']+=1; cur['stack'].append((m.group(1), cur['rid'])); continue
        m=re.search(r's_or_b64 exec, exec, (s\[\d+:\d+\])', line)
        if m:
            reg=m.group(1)
            idx=[i for i,(r,_) in enumerate(cur['stack']) if r==reg]
            if idx: cur['stack']=cur['stack'][:idx[-1]]
            continue
        if 'Folded Spill' in line:
            off=re.search(r'offset:(\d+)', line); off=int(off.group(1)) if off else 0
            cur['nsp']+=1
            cur['stores'].setdefault(off,[]).append((ln, tuple(r for _,r in cur['stack'])))
        elif 'Folded Reload' in line:
            off=re.search(r'offset:(\d+)', line); off=int(off.group(1)) if off else 0
            here=set(r for _,r in cur['stack'])
            sts=cur['stores'].get(off,[])
            if sts:
                # the most recent store decides
                sl, sreg = sts[-1]
                if sreg and not set(sreg) <= here:
                    cur['risky'].append((off, sl, ln))
    return out
for f in sys.argv[1:]:
    for k in scan(f):
        name=re.sub(r'EEvNS.*','',k['name'].replace('_ZN3mtm15ncc_mfma_kernel',''))
        if k['nsp']:
            r=sorted(set((o) for o,_,_ in k['risky']))
            print("%-42s %3d spill stores; slots whose last store sat in a region closed at the reload: %s %s" % (name, k['nsp'], r[:10], [(a,b) for _,a,b in k['risky'][:2]]))
