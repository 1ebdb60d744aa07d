"""Print a kernel's memory events in program order from hipcc's assembly listing (measurement tool).

    hipcc -O3 --offload-arch=gfx950 -DMSM_REAL=double -DMSM_SUFFIX=f64 --cuda-device-only -S -c msm_amd/csrc/engine.hip -o /tmp/e64.s
    python tools/isa_trace.py /tmp/e64.s k_psi_rtILi0 k_poisson_bE 7k_kick3

L global/buffer load, S store, x scratch access, d LDS instruction, s scalar load, Wn s_waitcnt vmcnt(n), | s_barrier,
b branch; a count follows a repeated event.  What to look for: a W0 in front of an LDS write (a table load issued behind
tile loads), loads between the last barrier and the stores, runs of b (selects compiled to exec-mask branches).
"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
names={}
for i,l in enumerate(lines):
    m=re.match(r'^(_ZN3msm\S+):', l)
    if m: names[m.group(1)]=i
for key in sys.argv[2:]:
    cands=[n for n in names if key in n]
    if not cands: print(key,'?'); continue
    st=names[cands[0]]
    ev=[]; n=0
    for l in lines[st+1:]:
        t=l.strip()
        if t.startswith('.Lfunc_end'): break
        if not t or t.startswith(('.',';')) or t.endswith(':'): continue
        n+=1
        if re.match(r'(buffer_load|global_load)', t): ev.append('L')
        elif re.match(r'(buffer_store|global_store)', t): ev.append('S')
        elif 'scratch_' in t: ev.append('x')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t: ev.append('W'+re.search(r'vmcnt\((\d+)\)', t).group(1))
        elif t.startswith('s_barrier'): ev.append('|')
        elif t.startswith('s_cbranch') or t.startswith('s_branch'): ev.append('b')
        elif t.startswith('s_load'): ev.append('s')
        elif t.startswith('ds_'): ev.append('d')
        elif t.startswith('s_endpgm'): ev.append('END')
    out=[]
    for e in ev:
        if out and out[-1][0]==e and e in 'LSx|bsd': out[-1][1]+=1
        else: out.append([e,1])
    print(key, n, 'instr:', ' '.join((e if c==1 else f'{e}{c}') for e,c in out))
    print()
