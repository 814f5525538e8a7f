#!/usr/bin/env python3
"""tools/isa_blocks.py FILE.s KERNEL_SUBSTRING [MIN] -- static instruction counts of a kernel's
basic blocks (total / VALU / VMEM / LDS / SALU) from hipcc -S output, in program order."""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
m = re.search(r'^(\S*%s\S*):[^\n]*\n' % re.escape(pat), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
cur, cnt = 'entry', [0, 0, 0, 0, 0]
tot = [0, 0, 0, 0, 0]
def cls(l):
    o = l.split()[0]
    return (1, o.startswith('v_'), o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')), o.startswith('ds_'),
            o.startswith('s_'))
for l in lines:
    if l.endswith(':'):
        if cnt[0] >= mn: print('%-14s total %4d valu %4d vmem %3d lds %3d salu %4d' % (cur, *cnt))
        cur, cnt = l[:-1], [0, 0, 0, 0, 0]
    elif not l.startswith('.'):
        c = cls(l)
        for i in range(5):
            cnt[i] += c[i]; tot[i] += c[i]
print('%-14s total %4d valu %4d vmem %3d lds %3d salu %4d' % (cur, *cnt))
print('KERNEL total %d valu %d vmem %d lds %d salu %d; lane-spill ops %d' % (*tot, sum('readlane' in l or 'writelane' in l for l in lines)))
