#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc -save-temps .s file (whole function bodies; loops are counted once).
usage: tools/isa_mix.py file.s [name-filter]"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
name, body, out = None, [], []
for line in s:
    m = re.match(r'^(_Z\w+):', line)
    if m:
        name, body = m.group(1), []
    elif line.startswith('.Lfunc_end') and name:
        out.append((name, body)); name = None
    elif name is not None:
        body.append(line.strip())
for name, body in out:
    if flt not in name: continue
    cnt, ops = collections.Counter(), collections.Counter()
    for line in body:
        if not line or line[0] in '.;/' or line.endswith(':'): continue
        op = line.split()[0]
        k = ('mfma' if op.startswith('v_mfma') else 'accvgpr' if op.startswith('v_accvgpr') else 'valu' if op.startswith('v_') else
             'lds' if op.startswith('ds_') else 'scratch' if op.startswith('scratch_') else 'vmem' if op.startswith(('global_', 'flat_', 'buffer_')) else
             'waitcnt' if op.startswith('s_waitcnt') else 'barrier' if op.startswith('s_barrier') else 'salu' if op.startswith('s_') else 'other')
        cnt[k] += 1
        if k == 'valu': ops[op] += 1
    print(name[:70], dict(cnt))
    print('  ', ops.most_common(22))
