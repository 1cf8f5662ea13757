"""Build libaejpeg_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:

    python adaptive_edge_aware_jpeg_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "color.hip", "canny.hip", "quadtree.hip", "dct.hip", "decode.hip", "metrics.hip", "deflate.hip"]
HEADERS = ["aej_common.h", "aej_launch.h", "aej_devmath.h", "inv_constants.h", "pow_tables.h", "aej_mfma.h", "aej_bigblock.h", os.path.join("..", "..", "include", "aej.h")]
LIB = os.path.join(HERE, "..", "libaejpeg_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# per-source additions.  canny.hip: the SLP vectoriser pairs the bilateral filter's float adds / fmas into v_pk_* instructions,
# which issue at half rate on gfx950 (profiles/r02_valu_issue_ubench.txt) and cost register pairs: measured 20 % slower.
# color.hip: the same for the 3 x 3 colour transforms (pairs formed through v_mov shuffles), and the strip kernel has to stay within 56 VGPRs.
EXTRA_FLAGS = {"canny.hip": ["-fno-slp-vectorize", "-Wno-pass-failed"], "color.hip": ["-fno-slp-vectorize"],
               "metrics.hip": ["-fno-slp-vectorize"]}      # the SSIM kernel's fma chains: v_pk_fma_f32 issues at half rate and its pairs cost moves


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    lib = os.path.abspath(LIB)
    if jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
