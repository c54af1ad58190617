#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: isa_blocks.py file.s <substring of the mangled kernel name> [min block size]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 60
i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
blocks, cur, label = [], [], 'entry'
for l in lines[i0 + 1:i1]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        blocks.append((label, cur)); cur = []; label = t
    elif t and not t.startswith(';') and not t.startswith('.'):
        cur.append(t)
blocks.append((label, cur))
for lab, b in blocks:
    if len(b) < minsz:
        continue
    c = Counter(x.split()[0] for x in b)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print('%s  n=%d  valu=%d salu=%d ds=%d vmem=%d' % (lab, len(b), grp('v_'), grp('s_'), grp('ds_'),
                                                   grp('global_') + grp('buffer_') + grp('flat_')))
    print('   ', ' '.join('%s:%d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])[:60]))
