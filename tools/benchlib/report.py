"""Roofline bookkeeping of bench.py: algorithmic bytes per stage, the committed rocprofv3 counter profiles, and the JSON objects built
from them.  Peaks are /opt/skills/guides/MI355X_MICROARCH.md's."""
import hashlib
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0          # HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 / 16x16x4_f32
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
# issue limit measured here (profiles/r02_valu_issue_ubench.txt): one plain 32-bit VALU instruction per SIMD every ~1.1 ns when >= 2
# waves share the SIMD (shifts / conversions / SDWA / 3-operand integer / packed ops take ~1.75 ns)
VALU_ISSUE_NS = 1.1

# algorithmic HBM bytes per INPUT pixel of each stage for 4:2:0-type spaces (1.5 plane-pixels per pixel); DESIGN.md section 4
ALGO_BYTES_PER_PX = {
    "color_planes": 12.0 + 1.5 * 4 + 1.5 * 1,   # f32 RGB in (3 B with --ingest u8); normalised f32 planes + u8 planes out
    "clahe_blur": 1.5 * (1 + 1),                 # u8 in, u8 out
    "sobel_nms": 1.5 * (1 + 1),                  # u8 in, u8 map out
    "hysteresis": 1.5 * (1 + 1),                 # map in, map out (one sweep is the algorithmic minimum)
    "quadtree": 1.5 * 1,                         # map in (leaf/state tables are < 0.1 B/px)
}
WHOLE_PATH_BYTES_PER_PX = 18.0                   # 12 B f32 RGB in + 4 B x 1.5 coefficients out
# The colour stage is scheduled as a BACKGROUND kernel (one workgroup per CU, sized to run in the gaps beside the
# other stages of the calls in flight); the chain a step waits for is everything else.  `roofline` is the longest kernel of THAT chain --
# a fixed rule, so that two kernels within 5 % of each other cannot swap the headline fraction from run to run (round 4's line did).
BACKGROUND_STAGES = ("color_planes",)
# kernel-name prefixes (as rocprofv3 prints them, tools/profiling/pmc.py short()) of every stage; a stage may be served by more than one
# kernel (hysteresis: pass 0 + bulk + drain; 64 x 64 DCT: one-wave or four-wave kernel by company)
KERNEL_OF_STAGE = {"color_planes": ("k_color_planes",), "clahe_blur": ("k_clahe_blur",), "sobel_nms": ("k_sobel_nms",), "hysteresis": ("k_hyst_",),
                   "quadtree": ("k_qt_",), "dct2": ("k_dct_small<2",), "dct4": ("k_dct4",), "dct8": ("k_dct8_shfl",), "dct16": ("k_dct16_mfma",),
                   "dct32": ("k_dct_mfma<32",), "dct64": ("k_dct_mfma<64", "k_dct64_wave"), "dct128": ("k_dct_mfma<128",),
                   "dct256": ("k_dct_big<256",), "dct512": ("k_dct_big<512",), "dct1024": ("k_dct_big<1024",)}
KERNEL_LABEL = {"quadtree": "k_qt_upper+count+scan+emit", "hysteresis": "k_hyst_pass0+k_hyst_bulk+k_hyst_drain"}
FIXED_STAGES = ("color_planes", "clahe_blur", "sobel_nms", "hysteresis", "quadtree")


def kernels_of_stage(stage, profiled_names):
    """Names in a PMC profile that belong to `stage`."""
    return [k for k in profiled_names if any(p in k for p in KERNEL_OF_STAGE[stage])]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def source_hash(root=None):
    """sha256 (first 16 hex digits) over the CODE of everything the benchmarked launches are built from: the kernel sources, their
    headers AND api.hip, whose launch shapes, chaining and sub-batching are what a pipelined profile measures -- comments and white space
    stripped, so that editing a comment does not invalidate a profile.  Left out: deflate / decode / metrics (kernels and entry points
    this benchmark never launches).  tools/profiling/pmc.py stores the same figure in the profile it writes, so staleness needs no git on
    the GPU box."""
    d = os.path.join(root or ROOT, "adaptive_edge_aware_jpeg_amd", "csrc")
    h = hashlib.sha256()
    not_benchmarked = ("deflate.hip", "decode.hip", "metrics.hip")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and f not in not_benchmarked:
            text = open(os.path.join(d, f), errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
            text = re.sub(r"//[^\n]*", " ", text)                        # line comments (no string literal of these sources holds "//")
            h.update(f.encode())
            h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def _load(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p)) if os.path.exists(p) else None
    except Exception:
        return None


def _newest(suffix):
    names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix) and f[:1] == "r" and f[1:3].isdigit())
    return names[-1] if names else suffix


class CounterProfiles:
    """The newest committed rocprofv3 PMC summaries (HBM bytes, VALU instruction counts).  They cannot be collected inside a normal run,
    so every figure taken from them says which profile it is from, for which sources, and whether those are the sources of this run."""

    def __init__(self, shape_key):
        self.traffic_file, self.valu_file = _newest("_hbm_traffic.json"), _newest("_pmc_valu.json")
        self.tj, self.vj = _load(self.traffic_file), _load(self.valu_file)
        self.shape_key = shape_key                  # (B, H, W, space, brange)
        self.head, self.src_hash = git_head(), source_hash()

    def same_shape(self, j):
        B, H, W, space, brange = self.shape_key
        return bool(j) and (j.get("batch"), j.get("height"), j.get("width")) == (B, H, W) and j.get("space", "YCbCr") == space and \
            tuple(j.get("blocks", (4, 64))) == tuple(brange)

    def stale(self, j):
        if j.get("src_hash"):
            return j["src_hash"] != self.src_hash
        return (j.get("head") != self.head) if (self.head and j.get("head")) else "unknown (profile predates source hashes and there is no git on this box)"

    def src_of(self, j, fname):
        return {"file": "profiles/" + fname, "profiled_commit": j.get("head"), "profiled_src_hash": j.get("src_hash"), "this_commit": self.head,
                "this_src_hash": self.src_hash, "stale": self.stale(j)}

    def gaps(self, brange):
        """every stage of this run must resolve to at least one kernel of the profile it is priced with (a renamed kernel would otherwise
        silently drop out of `traffic` / `valu`)"""
        out = {}
        for fname, j in ((self.traffic_file, self.tj), (self.valu_file, self.vj)):
            if j and j.get("kernels"):
                missing = [st for st in KERNEL_OF_STAGE if (st in FIXED_STAGES or (st.startswith("dct") and brange[0] <= int(st[3:]) <= brange[1]))
                           and not kernels_of_stage(st, j["kernels"])]
                if missing:
                    out[fname] = missing
        return out or None

    def traffic_of(self, stage):
        if not self.same_shape(self.tj):
            return None, None
        hit = [self.tj["kernels"][k]["hbm_bytes"] for k in kernels_of_stage(stage, self.tj["kernels"])]
        return (sum(hit), self.src_of(self.tj, self.traffic_file)) if hit else (None, None)

    def valu_insts_of(self, stage):
        if not self.same_shape(self.vj):
            return None
        hit = [self.vj["kernels"][k] for k in kernels_of_stage(stage, self.vj["kernels"])]
        # (wave-level VALU instructions of the stage's launches of one blocking call: one launch per kernel in the profiled, unsplit call)
        return sum(x.get("SQ_INSTS_VALU_per_encode", x["valu_insts_per_launch"]) for x in hit) if hit else None

    def whole_step(self):
        """-> (HBM bytes, wave VALU instructions) of one whole step: the sum over every kernel of the profile"""
        t = sum(k["hbm_bytes"] for k in self.tj["kernels"].values()) if self.same_shape(self.tj) else None
        v = (sum(k.get("SQ_INSTS_VALU_per_encode", 0.0) for k in self.vj["kernels"].values()) if self.same_shape(self.vj) else None)
        return t, v


def roofline_of(stage, algo, kernels_ms, local_px, prof):
    """-> (roofline object, valu object) of one stage: achieved = ALGORITHMIC bytes per launch / its HIP-event time"""
    ms = kernels_ms[stage]
    achieved = algo[stage] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, traffic_src = prof.traffic_of(stage)
    r = {"bound": "hbm", "kernel": KERNEL_LABEL.get(stage, KERNEL_OF_STAGE[stage][0]), "stage": stage, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
         "algorithmic_bytes_per_launch": algo[stage], "avg_launch_ms": round(ms, 4)}
    v = None
    insts = prof.valu_insts_of(stage)
    if insts:
        # frac = time the instructions need at the plain issue rate / measured time
        t_issue = insts / N_SIMD * VALU_ISSUE_NS * 1e-9
        v = {"kernel": r["kernel"], "wave_instructions_per_launch": insts, "instructions_per_plane_px": round(insts * 64 / (1.5 * local_px), 1),
             "frac_of_issue_peak": round(t_issue / (ms * 1e-3), 3), "issue_ns_per_simd_instruction": VALU_ISSUE_NS, "source": prof.src_of(prof.vj, prof.valu_file)}
    return r, v


def dct_by_block_size(leaf_hist, stage_ms):
    """DCT per block size: time, bytes moved per second (8 B per coefficient) and, for the MFMA sizes, the fraction of the
    157.3 TFLOP/s float32 MFMA peak (4 s^3 FLOP per leaf: two s x s x s products)"""
    out = {}
    for sz, n_leaves in leaf_hist.items():
        ms = stage_ms.get(f"dct{sz}", 0.0)
        if ms <= 0 or n_leaves == 0:
            continue
        gbps = 8.0 * sz * sz * n_leaves / (ms * 1e-3) / 1e9
        e = {"ms": round(ms, 4), "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 3)}
        if sz >= 16:
            tf = 4.0 * sz ** 3 * n_leaves / (ms * 1e-3) / 1e12
            e.update({"TFLOPs": round(tf, 1), "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 3)})
        out[str(sz)] = e
    return out
