#!/usr/bin/env python3
"""Static instruction census of a kernel by source line: compile with -gline-tables-only -S and give the .s and the kernel's
symbol prefix.  Prints, per source line, vector / scalar (without s_waitcnt, s_nop) / branch / LDS+memory instruction counts.
usage: tools/isa_by_line.py /tmp/m5g.s _Z18fcm_step_mw_kernel [top]"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().split('\n')
files = {}
start = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
end = [i for i, l in enumerate(lines) if i > start and 's_endpgm' in l][0]
cur = None
cnt = collections.defaultdict(lambda: [0, 0, 0, 0])
for i, l in enumerate(lines):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    if i < start or i > end:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z]\w+)', l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith('v_'): cnt[cur][0] += 1
    elif op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc')): cnt[cur][2] += 1
    elif op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop')): cnt[cur][1] += 1
    elif op.startswith(('ds_', 'buffer_', 'global_', 'flat_', 'scratch_')): cnt[cur][3] += 1
tot = [sum(v[j] for v in cnt.values()) for j in range(4)]
print('total: vector %d scalar %d branch %d lds/mem %d' % tuple(tot))
for k, v in sorted(cnt.items(), key=lambda kv: -(kv[1][0] + kv[1][1] + kv[1][2]))[:top]:
    print('%5d v %5d s %4d br %4d mem   %s:%s' % (v[0], v[1], v[2], v[3], k[0] if k else None, k[1] if k else None))
