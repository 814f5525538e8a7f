#!/usr/bin/env python3
"""tools/isa_marks.py FILE.s KERNEL_SUBSTRING -- static instruction counts between the `; VRMARK x`
comments of a -DVR_ISA_MARKS build (program order; a region runs from its mark to the next one)."""
import re
import sys
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*%s\S*):[^\n]*\n' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
cur, cnt, order = 'entry', {}, []
for l in body.split('\n'):
    l = l.strip()
    if not l:
        continue
    mk = re.match(r';\s*VRMARK (\S+)', l)
    if mk:
        cur = mk.group(1)
        continue
    if l.startswith(';') or l.endswith(':') or l.startswith('.'):
        continue
    if cur not in cnt:
        cnt[cur] = [0, 0, 0, 0, 0]
        order.append(cur)
    o = l.split()[0]
    c = cnt[cur]
    c[0] += 1
    c[1] += o.startswith('v_')
    c[2] += o.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))
    c[3] += o.startswith('ds_')
    c[4] += o.startswith('s_')
for k in order:
    print('%-12s total %5d valu %5d vmem %3d lds %3d salu %4d' % (k, *cnt[k]))
