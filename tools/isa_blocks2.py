#!/usr/bin/env python3
"""tools/isa_blocks2.py FILE.s KERNEL_SUBSTRING [MIN] -- the basic blocks of a kernel in hipcc -S output, in program
order: instructions / VALU / SALU / VMEM / LDS and the branches each block ends in (labels with trailing comments
are handled, which tools/isa_blocks.py does not)."""
import re
import sys
s = open(sys.argv[1]).read()
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r'^(\S*%s\S*):[^\n]*\n' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
blocks, cur = [], ['entry', []]
for l in body.split('\n'):
    l = l.strip()
    if not l or l.startswith(';'):
        continue
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1), []]
        continue
    if not l.startswith('.'):
        cur[1].append(l)
blocks.append(cur)
tot = [0] * 5
for name, ins in blocks:
    c = [len(ins), sum(i.startswith('v_') for i in ins), sum(i.startswith('s_') for i in ins),
         sum(i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) for i in ins), sum(i.startswith('ds_') for i in ins)]
    tot = [a + b for a, b in zip(tot, c)]
    if c[0] >= mn:
        print('%-10s n %4d valu %4d salu %3d vmem %2d lds %2d  %s' % (
            name, *c, ' | '.join(b.split(';')[0].strip() for b in ins if 'branch' in b)))
print('KERNEL     n %4d valu %4d salu %3d vmem %2d lds %2d' % tuple(tot))
