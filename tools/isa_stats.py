#!/usr/bin/env python3
"""ISA statistics of the wave kernel: instruction mix, waits, spills (from hipcc -S output)."""
import sys
from collections import Counter
s = open(sys.argv[1]).read()
name = [l for l in s.split('\n') if ('blind_rotate_wave_kernel' in l or 'blind_rotate_pair_kernel' in l) and l.startswith('_ZN')][0].split(':')[0] + ':'
start = s.index(name)
end = s.index('.Lfunc_end', start)
lines = [l.strip() for l in s[start:end].split('\n') if l.strip() and not l.strip().startswith((';', '.', '//'))]
# loop body = between the loop header label and backward branch: approximate by whole function
c = Counter(l.split()[0] for l in lines if not l.endswith(':'))
print('total instr', sum(c.values()))
f64 = sum(v for k, v in c.items() if 'f64' in k)
print('f64 instr', f64, ' s_nop', c['s_nop'], ' accvgpr', c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32'],
      ' scratch', c['scratch_load_dword'] + c['scratch_store_dword'], ' ds', sum(v for k, v in c.items() if k.startswith('ds_')),
      ' global_load', sum(v for k, v in c.items() if k.startswith('global_load')))
w = Counter(l for l in lines if l.startswith('s_waitcnt'))
print('waits:', sum(w.values()), ' lgkmcnt(0):', w['s_waitcnt lgkmcnt(0)'], ' vmcnt(0):', w['s_waitcnt vmcnt(0)'])
if len(sys.argv) > 2:
    for k, v in c.most_common(40):
        print('  ', k, v)
