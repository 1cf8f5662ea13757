#!/usr/bin/env python3
"""Side-by-side timing of experimental builds of ONE kernel file (developer tool, not product).

    python3 tools/profiling/variants.py build canny.hip name1="-DAEJ_X_FOO" name2="-DAEJ_X_BAR -fno-slp-vectorize" ...   (build box: hipcc cross-compiles)
    python3 tools/profiling/variants.py run [bench args]                                                              (GPU box)

`build` compiles the named source with the extra flags into build/variants/<name>/ and links it with the in-tree objects of
the other sources into build/variants/<name>/libaejpeg_hip.so (the in-tree library is never touched; build/ is git-ignored
but travels to the GPU box).  `run` executes bench.py once per variant (and once for the in-tree library, "base") in separate
processes on the same device and prints the stage times, interleaved twice to expose run-to-run noise.
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "adaptive_edge_aware_jpeg_amd", "csrc")
VDIR = os.path.join(ROOT, "build", "variants")
sys.path.insert(0, CSRC)


def build(src, variants):
    import build as B     # csrc/build.py: SOURCES, FLAGS
    B.build(verbose=False)
    hipcc = B._hipcc()

    def one(item):
        name, flags = item
        d = os.path.join(VDIR, name)
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, src.replace(".hip", ".o"))
        fl = flags.split()
        srcfile = os.path.join(CSRC, src)
        for f in list(fl):                      # --src=<file>: compile another version of the source (e.g. one taken from git history)
            if f.startswith("--src="):
                srcfile = f[6:]
                fl.remove(f)
        subprocess.check_call([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-Wno-pass-failed", "-I", CSRC] + fl + ["-c", srcfile, "-o", obj])
        objs = [obj if s == src else os.path.join(CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libaejpeg_hip.so")] + objs)
        open(os.path.join(d, "flags.txt"), "w").write(f"{src} {flags}\n")
        return name

    with ThreadPoolExecutor(max_workers=6) as ex:
        print("built:", list(ex.map(one, variants)))


def run(bench_args):
    names = ["base"] + sorted(os.listdir(VDIR)) if os.path.isdir(VDIR) else ["base"]
    for rep in range(2):
        for n in names:
            env = dict(os.environ)
            if n != "base":
                env["AEJ_LIBRARY"] = os.path.join(VDIR, n, "libaejpeg_hip.so")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "6", "--warmup", "2"] + bench_args,
                               env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print(f"{n:16s} FAILED: {r.stderr[-400:]}", flush=True)
                continue
            d = json.loads(line[-1])
            st = {k: round(v["ms"], 3) for k, v in d["stages"].items()}
            print(f"{n:16s} rep {rep}: {d['ms_per_step']:7.3f} ms/step  verified={d['verified']['ok'] if d.get('verified') else None}  {st}", flush=True)


def pmc(kernel, bench_args):
    """per variant: one `pmc.py valu` run, print the named kernel's LDS / VALU counters per launch"""
    names = ["base"] + sorted(os.listdir(VDIR)) if os.path.isdir(VDIR) else ["base"]
    keys = ["SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
    for n in names:
        env = dict(os.environ)
        if n != "base":
            env["AEJ_LIBRARY"] = os.path.join(VDIR, n, "libaejpeg_hip.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profiling", "pmc.py"), "valu"] + bench_args, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{n:16s} FAILED {r.stderr[-300:]}", flush=True)
            continue
        d = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_valu.json")))["kernels"]
        k = [v for name, v in d.items() if kernel in name][0]
        print(f"{n:16s}", {c.replace("SQ_", ""): f"{k.get(c + '_per_encode', 0):.3g}" for c in keys}, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "build":
        build(sys.argv[2], [tuple(a.split("=", 1)) for a in sys.argv[3:]])
    else:
        run(sys.argv[2:])
