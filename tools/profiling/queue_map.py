#!/usr/bin/env python3
"""Which hardware queue each stream's kernels ran on (rocprofv3 --kernel-trace CSV of a pipelined bench run):
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --timed-only --steps 8 --warmup 4 ; python3 tools/profiling/queue_map.py DIR
Two busy streams on one queue run one after the other."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
qs = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    if "aej::" in r["Kernel_Name"]:
        qs[r["Queue_Id"]][r.get("Stream_Id", "?")] += 1
for q, c in sorted(qs.items(), key=lambda kv: int(kv[0])):
    print("queue", q, "streams", dict(c))
print(len(qs), "queues carried the library's kernels")
