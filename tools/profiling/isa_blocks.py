#!/usr/bin/env python3
"""Opcode histogram of the large basic blocks of one kernel:  isa_blocks.py file.s <name-substring> [min-instructions]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"^(\S*" + re.escape(sys.argv[2]) + r"\S*):", s, re.M)
i = m.start()
j = s.index("s_endpgm", i)
least = int(sys.argv[3]) if len(sys.argv) > 3 else 150
blocks, cur, name = [], [], "entry"
for l in s[i:j].splitlines()[1:]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb."):
        blocks.append((name, cur))
        cur, name = [], t
    elif t and not t.startswith(";") and not t.startswith("."):
        cur.append(t.split(";")[0].strip())
blocks.append((name, cur))
for name, ins in blocks:
    if len(ins) >= least:
        c = collections.Counter(x.split()[0] for x in ins)
        kinds = collections.Counter("valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else "vmem" for o in
                                    (x.split()[0] for x in ins))
        print(name, len(ins), dict(kinds))
        print("   ", sorted(c.items(), key=lambda x: -x[1]))
