#!/usr/bin/env python3
"""Static instruction mix of one kernel (device assembly), whole kernel and per basic block:

    hipcc <flags> --cuda-device-only -S file.hip -o /tmp/file.s ; python3 tools/profiling/isa_count.py /tmp/file.s <mangled-name-substring>
"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M)
if not m:
    raise SystemExit("kernel not found")
print(m.group(1))


def kind(op):
    return ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")


tot = collections.Counter()
blk, name = collections.Counter(), "entry"
for l in m.group(2).splitlines():
    l = l.strip()
    if not l or l.startswith((";", ".")) and not l.endswith(":"):
        continue
    if l.endswith(":") or re.match(r"^\.?LBB\S+:", l):
        if sum(blk.values()):
            print(f"  {name:14s}", dict(blk))
        blk, name = collections.Counter(), l.split(":")[0]
        continue
    op = l.split()[0]
    tot[kind(op)] += 1
    blk[kind(op)] += 1
if sum(blk.values()):
    print(f"  {name:14s}", dict(blk))
print("total", dict(tot))
