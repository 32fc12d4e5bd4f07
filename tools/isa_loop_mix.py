#!/usr/bin/env python3
"""Instruction mix of the CMUX loop of a blind-rotation kernel, hot path only (from `hipcc -S` output).

usage: isa_loop_mix.py file.s kernel_substring [-v]
The loop is the outermost `Loop Header: Depth=1` block that contains most instructions; blocks that contain
s_swappc (out-of-line cold calls) or s_sleep (the bounded spins) are left out."""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split('\n')
kern = sys.argv[2]
verbose = '-v' in sys.argv
# function extent
start = next(i for i, l in enumerate(src) if l.startswith('_Z') and kern in l and re.match(r'^_Z\S+:', l))
end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
body = src[start:end]
# split into basic blocks by labels
blocks = []
cur = ('entry', [], '')
for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):(.*)$', l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [], m.group(2))
        continue
    t = l.strip()
    if not t or t.startswith((';', '.', '//')):
        continue
    cur[1].append(t)
blocks.append(cur)
# the loop: the Depth=1 header with the most instructions in the blocks that name it as their (parent) loop
hdrs = [i for i, b in enumerate(blocks) if 'Loop Header: Depth=1' in b[2]]
best = None
for h in hdrs:
    tag = blocks[h][0][2:]  # "BB3_13"
    member = [i for i, b in enumerate(blocks) if i == h or re.search(r'(Header=|Parent Loop )' + re.escape(tag) + r'\b', b[2])]
    span = sum(len(blocks[i][1]) for i in member)
    if best is None or span > best[2]:
        best = (h, member, span)
h, member, _ = best
last = member[-1]
hot = Counter()
skipped = 0
for name, ins, comment in (blocks[i] for i in member):
    if any(x.startswith(('s_swappc', 's_sleep')) for x in ins):
        skipped += len(ins)
        continue
    for x in ins:
        hot[x.split()[0]] += 1
def cat(op):
    if op.startswith('v_') and 'f64' in op: return 'valu_f64'
    if op.startswith('v_permlane'): return 'valu_swap'
    if op.startswith('v_'): return 'valu_int'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    return 'salu'
cats = Counter()
for op, n in hot.items():
    cats[cat(op)] += n
valu = cats['valu_f64'] + cats['valu_swap'] + cats['valu_int']
print(f'loop blocks {blocks[h][0]}..{blocks[last][0]}  hot instructions {sum(hot.values())}  (cold skipped {skipped})')
print(f'VALU {valu}: f64 {cats["valu_f64"]}  int {cats["valu_int"]}  lane swaps {cats["valu_swap"]} | LDS {cats["lds"]}  vmem {cats["vmem"]}  SALU {cats["salu"]}  waits {cats["wait"]}  nops {cats["nop"]}')
if verbose:
    for op, n in hot.most_common(60):
        print(f'   {op:28s} {n}')
