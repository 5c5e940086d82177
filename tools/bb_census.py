#!/usr/bin/env python3
"""Per-basic-block instruction census of a kernel in hipcc's assembly output (-S).
usage: bb_census.py file.s [kernel-name-prefix]    prints: label line valu (of which spill readlane/writelane) salu lds vmem branch-targets"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2] if len(sys.argv) > 2 else '_Z18fcm_step_mw_kernel'
on = False; blocks = []; cur = None
for ln, l in enumerate(src, 1):
    if not on:
        if l.startswith(pref) and ':' in l: on = True; cur = dict(name='entry', line=ln, valu=0, spill=0, salu=0, lds=0, vmem=0, tgt=[], loop=''); blocks.append(cur)
        continue
    s = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', s) or re.match(r'^; %bb\.(\d+):(.*)', s)
    if m:
        cur = dict(name=m.group(1), line=ln, valu=0, spill=0, salu=0, lds=0, vmem=0, tgt=[], loop=('D' + m.group(2).split('Depth=')[1][:1]) if 'Depth=' in m.group(2) else ''); blocks.append(cur); continue
    if not s or s.startswith(';') or s.startswith('.'): continue
    op = s.split()[0]
    if op.startswith('v_'):
        cur['valu'] += 1
        if re.match(r'v_readlane_b32 s\d+, v\d+, \d+$', s) or re.match(r'v_writelane_b32 v\d+, s\d+, \d+$', s): cur['spill'] += 1   # immediate lane index: SGPR spill slots (and a few constant-lane reads of the code's own)
    elif op.startswith('s_'):
        cur['salu'] += 1
        if op.startswith('s_cbranch') or op == 's_branch': cur['tgt'].append(s.split()[-1])
        if op == 's_endpgm': break
    elif op.startswith('ds_'): cur['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): cur['vmem'] += 1
for b in blocks:
    print(f"{b['name']:>12} L{b['line']:<6} {b['loop']:3} valu {b['valu']:4} (spill {b['spill']:3}) salu {b['salu']:4} lds {b['lds']:3} vmem {b['vmem']:3}  -> {' '.join(b['tgt'])}")
print('total valu', sum(b['valu'] for b in blocks), 'spill', sum(b['spill'] for b in blocks))
