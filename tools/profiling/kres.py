#!/usr/bin/env python3
"""Register / LDS / occupancy figures of every kernel of one source as the compiler reports them (no GPU needed):

    python3 tools/profiling/kres.py canny.hip [-DAEJ_X_... extra flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "adaptive_edge_aware_jpeg_amd", "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402

src = sys.argv[1]
cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:] + \
      ["-c", os.path.join(CSRC, src), "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for ln in out.splitlines():
    m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass", ln)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("aej::", "")[:64]}
    cur[k] = v
    if k.startswith("LDS Size"):
        g = lambda key: str(cur.get(key, "-"))
        print(f"{cur['name']:64s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} sgpr {g('SGPRs'):>4s} occ {g('Occupancy [waves/SIMD]'):>2s} "
              f"lds {v:>6s} scratch {g('ScratchSize [bytes/lane]')}")
