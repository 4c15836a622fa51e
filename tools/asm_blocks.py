#!/usr/bin/env python3
"""Basic-block instruction mix of one kernel in a hipcc -S listing (static view: which blocks are
large and what they are made of).  usage: asm_blocks.py file.s mangled_name_substring [min_instrs]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 90
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
a = m.start()
b = s.index('.Lfunc_end', a)
blocks = []
cur = ('entry', [])
for l in s[a:b].split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        blocks.append(cur)
        cur = (mm.group(1), [])
    else:
        t = l.strip()
        if t and not t.startswith(('.', ';', '//')) and not t.endswith(':'):
            cur[1].append(t)
blocks.append(cur)
tot = 0
for name, ins in blocks:
    tot += len(ins)
    if len(ins) >= minsz:
        c = Counter(i.split()[0] for i in ins)
        v = sum(n for o, n in c.items() if o.startswith('v_'))
        print(name, len(ins), 'valu', v, c.most_common(10))
print('total', tot)
