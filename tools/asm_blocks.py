#!/usr/bin/env python3
"""asm_blocks.py <file.s> <kernel-name-substring>: basic blocks of a kernel in hipcc's -S output with their VALU /
SALU / VMEM / LDS instruction counts and branch targets (which loops cost what)."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l.split(':')[0] and ':' in l)
blocks, cur = [], {'name': 'entry', 'v': 0, 's': 0, 'm': 0, 'l': 0, 'br': [], 'line': start}
for i in range(start + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith('.Lfunc_end'): break
    if not l or l.startswith(';') or l.startswith('.') and not l.startswith('.LBB'): continue
    m = re.match(r'^(\.LBB\w+):', l)
    if m:
        blocks.append(cur); cur = {'name': m.group(1), 'v': 0, 's': 0, 'm': 0, 'l': 0, 'br': [], 'line': i + 1}; continue
    op = l.split()[0]
    if op.startswith('v_'): cur['v'] += 1
    elif op.startswith('s_'):
        cur['s'] += 1
        if 'branch' in op: cur['br'].append(op.replace('s_cbranch_', 'c_').replace('s_branch', 'jmp') + '>' + l.split()[-1])
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['m'] += 1
    elif op.startswith('ds_'): cur['l'] += 1
blocks.append(cur)
for b in blocks:
    print(f"{b['line']:6d} {b['name']:14s} V{b['v']:4d} S{b['s']:4d} M{b['m']:3d} L{b['l']:3d}  {' '.join(b['br'])}")
print('total V', sum(b['v'] for b in blocks), 'S', sum(b['s'] for b in blocks))
