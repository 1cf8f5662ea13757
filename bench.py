#!/usr/bin/env python3
"""bench.py -- encode-hot-path throughput on MI355X (BASELINE.json metric: megapixels/s, 4K batch).

One "step" = one pass of the whole encode hot path (colour convert -> chroma down-sample -> Canny chain ->
quadtree -> DCT -> quantise -> zigzag) over one device-resident batch of
64 synthetic 3840x2160 float32 RGB images per GPU (BASELINE config 4: 512 4K images over 8 GPUs = 64 per GPU;
weak scaling).  Inputs are in HBM before the timed region; outputs stay in HBM.

    python bench.py                      # 1 GPU, 64 x 4K, 10 steps
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

What the timed region does and does not contain:
* the K steps rotate over `--pipeline` (default 3; round 2 used 2 -- with the colour stage off the critical path a third call in flight
  fills what the other two leave idle: 64 x 4K 6.7 -> 6.45 ms, four: 6.85) contexts, each on its own stream with its own output buffers and workspace:
  step i is enqueued with aej_encode_batch_begin on context i % n after the step that used that context before has been ended
  (aej_encode_batch_end: waited for, device counters checked).  Two calls in flight let the HBM-bound
  stages of one (colour planes, DCT) run beside the issue-bound stages of the other (blur, Sobel / NMS, quadtree); the library
  keeps them one stage apart.  All K steps are complete inside the timed region (sync() ends every call in flight before the
  clock stops).  `pipeline.serial_ms_per_step` is the same K steps as blocking aej_encode_batch calls on one context.
* TWO different device-resident batches (different seeds) alternate across the steps, so nothing data-dependent can be remembered
  from one call to the next (round 4: nothing is -- the hysteresis completes on the device; `hysteresis` in the JSON line reports
  how many tiles went through its work queue).
* no profiling events: stage times come from separate, untimed steps afterwards.
* after the timed region the outputs of the LAST timed step are compared with the CPU oracle for the first and the last image
  of the batch (`"verified"`), so the number is tied to correct output.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (dominant kernel, HIP-event timed on the
launch stream inside the library), `valu` (the same kernel against the vector-issue limit, which is what actually bounds it)
and `cpu_baseline` (the C oracle, one image per host core, bounded sample, plus the reference-structured NumPy restatement on
one core).
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # same guide: v_mfma_f32_32x32x2_f32 / 16x16x4_f32
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
H4K, W4K = 2160, 3840

# algorithmic HBM bytes per INPUT pixel of each stage for 4:2:0-type spaces (1.5 plane-pixels per pixel); DESIGN.md section 4
ALGO_BYTES_PER_PX = {
    "color_planes": 12.0 + 1.5 * 4 + 1.5 * 1,   # f32 RGB in (3 B with --ingest u8); normalised f32 planes + u8 planes out
    "clahe_blur": 1.5 * (1 + 1),                 # u8 in, u8 out
    "sobel_nms": 1.5 * (1 + 1),                  # u8 in, u8 map out
    "hysteresis": 1.5 * (1 + 1),                 # map in, map out (one sweep is the algorithmic minimum)
    "quadtree": 1.5 * 1,                         # map in (leaf/state tables are < 0.1 B/px)
}
WHOLE_PATH_BYTES_PER_PX = 18.0                   # 12 B f32 RGB in + 4 B x 1.5 coefficients out
# kernel-name prefixes (as rocprofv3 prints them, tools/profiling/pmc.py short()) of every stage; a stage may be served by more than one
# kernel (hysteresis: pass 0 + the drain; 64 x 64 DCT: one-wave or four-wave kernel by company)
KERNEL_OF_STAGE = {"color_planes": ("k_color_planes",), "clahe_blur": ("k_clahe_blur",), "sobel_nms": ("k_sobel_nms",), "hysteresis": ("k_hyst_",),
                   "quadtree": ("k_qt_",), "dct2": ("k_dct_small<2",), "dct4": ("k_dct4",), "dct8": ("k_dct8_shfl",), "dct16": ("k_dct16_mfma",),
                   "dct32": ("k_dct_mfma<32",), "dct64": ("k_dct_mfma<64", "k_dct64_wave"), "dct128": ("k_dct_mfma<128",),
                   "dct256": ("k_dct_big<256",), "dct512": ("k_dct_big<512",), "dct1024": ("k_dct_big<1024",)}


def kernels_of_stage(stage, profiled_names):
    """Names in a PMC profile that belong to `stage`."""
    return [k for k in profiled_names if any(p in k for p in KERNEL_OF_STAGE[stage])]


def synth_batch(torch, B, H, W, seed, device):
    """'mixed' synthetic images of SURVEY.md 8d, generated on the GPU: smooth sinusoidal background, K = ceil(N/32768)
    opaque rectangles, N(0, 1.5^2) noise, rounded to uint8 levels, /255 -> float32 [B, H, W, 3].  Image i uses seed + i."""
    out = torch.empty((B, H, W, 3), dtype=torch.float32, device=device)
    yy = (torch.arange(H, device=device, dtype=torch.float32) / H)[:, None]
    xx = (torch.arange(W, device=device, dtype=torch.float32) / W)[None, :]
    K = -(-(H * W) // 32768)
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        img = out[b]
        for c in range(3):
            fx, fy = rng.integers(1, 4, size=2)
            phi, psi = rng.uniform(0, 2 * np.pi, size=2)
            img[:, :, c] = 127.5 + 80.0 * torch.sin(2 * math.pi * float(fx) * xx + float(phi)) * torch.cos(2 * math.pi * float(fy) * yy + float(psi))
        x0 = rng.integers(0, W, size=K); y0 = rng.integers(0, H, size=K)
        ww = rng.integers(16, 257, size=K); hh = rng.integers(16, 257, size=K)
        col = rng.integers(0, 256, size=(K, 3)).astype(np.float32)
        colt = torch.from_numpy(col).to(device)
        for k in range(K):
            img[y0[k]:y0[k] + hh[k], x0[k]:x0[k] + ww[k], :] = colt[k]
        g = torch.Generator(device=device)
        g.manual_seed(seed + b)
        img += torch.randn(img.shape, generator=g, device=device) * 1.5
        img.round_().clamp_(0, 255)
    # uint8 levels -> float32 exactly as image.py:80 does (`astype(np.float32) / 255.0`, a true IEEE division): torch divides
    # by a scalar through a reciprocal multiply, which is 1 ulp off for some levels, so the quotients come from a NumPy table
    lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(device)
    for b in range(B):
        out[b] = lut[out[b].to(torch.int64)]
    return out


def natural_batch(torch, B, H, W, seed, device):
    """Labelled variant (--data natural): the reference's own natural test images (tests/golden/natural/*.png = its test_images/,
    metrics_computation.py:307-324) mirror-tiled to H x W -- reflected copies side by side, so the seams add no artificial edges --
    each batch image from another source image / tile offset; uint8 levels -> float32 by the exact division of image.py:80."""
    from PIL import Image as PILImage
    d = os.path.join(ROOT, "tests", "golden", "natural")
    names = sorted(f for f in os.listdir(d) if f.endswith(".png"))
    srcs = [np.asarray(PILImage.open(os.path.join(d, f)).convert("RGB")) for f in names]
    lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(device)
    out = torch.empty((B, H, W, 3), dtype=torch.float32, device=device)
    for b in range(B):
        k = seed + b
        src = srcs[k % len(srcs)]
        period = np.concatenate([np.concatenate([src, src[:, ::-1]], 1), np.concatenate([src[::-1], src[::-1, ::-1]], 1)], 0)   # 2h x 2w, tiles seamlessly
        ph, pw = period.shape[:2]
        oy, ox = (k * 37) % ph, (k * 53) % pw
        ys = (np.arange(H) + oy) % ph
        xs = (np.arange(W) + ox) % pw
        img = torch.from_numpy(np.ascontiguousarray(period[ys][:, xs])).to(device)
        out[b] = lut[img.to(torch.int64)]
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("AEJ_BENCH_BATCH", "64")), help="images per GPU (weak scaling)")
    ap.add_argument("--total-images", type=int, default=0,
                    help="strong-scaling variant: this many images in total, cut into contiguous shards by image index (sharding.shard_bounds)")
    ap.add_argument("--height", type=int, default=H4K)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--space", default="YCbCr")
    ap.add_argument("--blocks", type=int, nargs=2, default=[4, 64])
    ap.add_argument("--quality", type=int, nargs=2, default=[40, 80])
    ap.add_argument("--ingest", choices=["f32", "u8"], default="f32",
                    help="f32 = the BASELINE metric's float32 RGB input; u8 = 8-bit ingest (aej_encode_batch_u8, 3 B/px in), reported as a variant")
    ap.add_argument("--data", choices=["synthetic", "natural"], default="synthetic",
                    help="synthetic = SURVEY 8d's 'mixed' generator (the headline); natural = the reference's own test images mirror-tiled to "
                         "the image size (a labelled variant: textures change the leaf mix and the hysteresis pass count)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing oracle comparison")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads for cpu_baseline; 0 = min(cores this process may use, 64)")
    ap.add_argument("--graph", type=int, choices=[0, 1, 2], default=0,
                    help="aej_set_graph_mode: 0 never replay a captured hipGraph (the library default), 1 automatic (calls of at most 8 Mpx), 2 whenever possible")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="aej_set_sub_batches: 0 automatic (the library default: 4 sub-batches on private streams for calls of at least 64 Mpx), 1 never, 2..8")
    ap.add_argument("--pipeline", type=int, choices=[1, 2, 3, 4], default=3,
                    help="contexts (each on its own stream, with its own output buffers and workspace) the timed steps rotate over: step i is "
                         "enqueued with aej_encode_batch_begin on context i %% n after the step that used it before has been ended; 1 = blocking calls")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="aej_set_option on every context (include/aej.h has the table), e.g. --option dct64_kernel=1; A / B runs only")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiler runs (tools/profiling/*.sh): only the W warm-up and K timed steps, so every kernel is launched a known "
                         "number of times; prints value / ms_per_step only")
    ap.add_argument("--rehearse-control-flow", action="store_true",
                    help="NO GPU work: run only the multi-rank control flow (rendezvous, per-rank seeds, barriers, reduction, rank-0 "
                         "JSON) with a sleep in place of the encode; used by the CPU gloo test, never a measurement")
    return ap.parse_args(argv)


def spawn_ranks_if_asked(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.  The parent never imports torch or
    touches the GPU; it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same
    args>` as a CHILD process (never exec), whose rank 0 prints the JSON line on the inherited stdout, and exits with the child's code.
    The reference fans out the same way, from its own harness (test/analysis/metrics_computation.py:253).  Under an external launcher
    (WORLD_SIZE set) nothing is spawned, but --gpus must agree with it."""
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world} of the launcher that started this process")
        return
    if args.gpus <= 1:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    raise SystemExit(subprocess.run(cmd).returncode)


def build_oracle_once(dist, local_rank):
    """The checker's shared object is git-ignored: on a fresh box it is compiled here, by ONE process per node, before anything is timed;
    the other ranks wait at a barrier and only load it (oracle.build() itself also renames into place atomically)."""
    from oracle import oracle as O
    if local_rank == 0:
        O.build()
    group_barrier(dist)
    O.build()          # (no-op when rank 0 of this node has just built it)
    return O


def rank_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(torch, backend, rank, local_rank, world):
    """One process per GPU; rendezvous on 127.0.0.1 (the container hostname may not resolve).  Returns torch.distributed or None.
    The RCCL communicator is NOT created here (no `device_id`): it comes up at the first collective, which main() places after every
    stream of this process exists -- see `hw_queue_default`."""
    if world <= 1 and not os.environ.get("AEJ_BENCH_FORCE_DIST"):      # (rehearsal switch: a ONE-rank process group, so that a one-GPU box can
        return None                                                     # run the barriers and the counter collectives through RCCL itself)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl" and os.environ.get("AEJ_BENCH_NCCL_EAGER"):    # (A / B only: profiles/r04_rccl_hw_queues.txt)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def group_barrier(dist):
    if dist is None:
        return
    if dist.get_backend() == "nccl":
        import torch
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def hw_queue_default(world):
    """HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and two streams on one queue run one after the other.  This
    process uses one stream per context plus the library's sub-batch streams: 16 with the null stream, so 16 queues give every stream its
    own -- until somebody else creates streams first.  A live RCCL communicator does (measured on one GPU, one rank, profiles/
    r04_rccl_hw_queues.txt: 6.85-6.96 ms per step against 5.96-5.98 with the same blocking-call time, back to 5.95-5.99 with 24 queues or
    with the communicator created after our streams), so a process that will hold a process group asks for 24 AND creates its
    communicator last."""
    return "24" if world > 1 or os.environ.get("AEJ_BENCH_FORCE_DIST") else "16"


def local_batch_and_seed(args, rank, world):
    """-> (images on this rank, seed of its first image of batch A, seed of batch B, scaling label).  Seeds are distinct per
    rank and per image: weak scaling gives rank r the images [r*B, (r+1)*B) of an endless seeded sequence; the strong-scaling
    variant cuts --total-images into contiguous shards (sharding.shard_bounds)."""
    from adaptive_edge_aware_jpeg_amd.sharding import shard_bounds
    if args.total_images > 0:
        lo, hi = shard_bounds(args.total_images, rank, world)
        return hi - lo, 20250718 + lo, 20250718 + 1_000_000 + lo, "strong"
    return args.batch, 20250718 + rank * args.batch, 20250718 + 1_000_000 + rank * args.batch, "weak"


def timed_loop(torch, dist, step, steps, sync):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides -> local seconds."""
    group_barrier(dist)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    timed_loop.own_seconds = time.perf_counter() - t0      # this rank's own K steps, before it waits for the others (per-rank report)
    group_barrier(dist)
    return time.perf_counter() - t0


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def source_hash(root=None):
    """sha256 (first 16 hex digits) over the CODE of the kernel sources (kernels + their launchers; not the host orchestration in api.hip) and
    headers the library is built from -- comments and white space stripped, so that editing a comment does not invalidate a profile: what a
    per-kernel PMC profile is valid for.
    tools/profiling/pmc.py stores the same figure in the profile it writes, so staleness needs no git on the GPU box."""
    import hashlib
    import re
    d = os.path.join(root or ROOT, "adaptive_edge_aware_jpeg_amd", "csrc")
    h = hashlib.sha256()
    # api.hip is host orchestration (no kernel, no launch shape); deflate / decode / metrics hold kernels this benchmark never launches
    not_profiled = ("api.hip", "deflate.hip", "decode.hip", "metrics.hip")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and f not in not_profiled:
            text = open(os.path.join(d, f), errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
            text = re.sub(r"//[^\n]*", " ", text)                        # line comments (no string literal of these sources holds "//")
            h.update(f.encode())
            h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def load_profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return None
    return None


def rehearse(args):
    """Control flow only (see --rehearse-control-flow)."""
    import torch
    rank, local_rank, world = rank_env()
    dist = init_distributed(torch, os.environ.get("AEJ_BENCH_BACKEND", "gloo"), rank, local_rank, world)
    from adaptive_edge_aware_jpeg_amd.sharding import aggregate_throughput, gather_rank_report
    B, seed_a, seed_b, scaling = local_batch_and_seed(args, rank, world)
    dt = timed_loop(torch, dist, lambda i: time.sleep(0.002 * (1 + rank)), args.steps, lambda: None)
    dt_local = timed_loop.own_seconds
    px, dt = aggregate_throughput(dist, B * args.height * args.width * args.steps, dt, None)
    fake_bad = os.environ.get("AEJ_REHEARSE_BAD_RANK")            # the CPU test makes one rank report a failed oracle check
    verdict = not (fake_bad is not None and int(fake_bad) == rank)
    if os.environ.get("AEJ_REHEARSE_UNVERIFIED_RANK") is not None and int(os.environ["AEJ_REHEARSE_UNVERIFIED_RANK"]) == rank:
        verdict = None                                            # ... or never reach its check
    ranks = gather_rank_report(dist, local_rank, dt_local / args.steps * 1e3, B, verdict, None)
    if rank == 0:
        emit(json.dumps({"metric": "REHEARSAL of bench.py's multi-rank control flow (no GPU work, not a measurement)", "value": None,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": scaling, "pixels_total": px,
                          "seconds_max": round(dt, 4), "rank0_images": B, "rank0_seeds": [seed_a, seed_b], "ranks": ranks}))
    if dist is not None:
        dist.destroy_process_group()
    if not ranks["all_verified"]:
        raise SystemExit(3)                                       # every rank exits non-zero, as in the real run


def keep_stdout_for_the_json_line():
    """Rank 0 prints ONE JSON line on stdout, and nothing else may: libraries write there too (RCCL prints a five-line version banner on
    stdout when its first communicator comes up).  File descriptor 1 is pointed at stderr for the rest of the process, and `print` is
    given a copy of the real stdout only through `emit()`."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def emit(line):
    _REAL_STDOUT.write(line + "\n")
    _REAL_STDOUT.flush()


_REAL_STDOUT = sys.stdout


def main():
    global _REAL_STDOUT
    args = parse_args()
    spawn_ranks_if_asked(args, sys.argv[1:])
    _REAL_STDOUT = keep_stdout_for_the_json_line()
    if args.rehearse_control_flow:
        return rehearse(args)

    # HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams on one queue run one after the other;
    # this process uses one stream per context plus the library's sub-batch streams, so give every stream a queue of its own
    # (must be set before the HIP runtime initialises)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", hw_queue_default(rank_env()[2]))
    import torch
    # imported BEFORE anything initialises HIP: the package then knows that the queue count in the environment is the one the runtime
    # will start with and tells the library (aej_set_hw_queues); imported later it would have to assume HIP's default of 4
    import adaptive_edge_aware_jpeg_amd as A
    assert A.hw_queues()[0] == int(os.environ["GPU_MAX_HW_QUEUES"]), A.hw_queues()      # the variable was in the environment before torch was imported
    rank, local_rank, world = rank_env()
    # rehearsal switches (not used by the driver): AEJ_BENCH_BACKEND=gloo + AEJ_BENCH_ONE_DEVICE=1 run the multi-rank control flow
    # with every rank on GPU 0 of a one-GPU box; RCCL needs one GPU per rank
    backend = os.environ.get("AEJ_BENCH_BACKEND", "nccl")
    if os.environ.get("AEJ_BENCH_ONE_DEVICE"):
        local_rank = 0
    # device_count() does not initialise the GPU; everything that does comes after the rendezvous below
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py needs an MI355X: no HIP device for local rank {local_rank}")
    torch.cuda.set_device(local_rank)
    dist = init_distributed(torch, backend, rank, local_rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")

    from adaptive_edge_aware_jpeg_amd.sharding import aggregate_throughput, gather_rank_report
    H, W = args.height, args.width
    B, seed_a, seed_b, scaling = local_batch_and_seed(args, rank, world)
    if B < 1:
        raise SystemExit("no images for this rank")
    space, qrange, brange = args.space, tuple(args.quality), tuple(args.blocks)
    dev = torch.device("cuda", local_rank)
    make_batch = synth_batch if args.data == "synthetic" else natural_batch
    batches_f32 = [make_batch(torch, B, H, W, seed_a, dev), make_batch(torch, B, H, W, seed_b, dev)]
    batches = batches_f32 if args.ingest == "f32" else [(x * 255.0).round().to(torch.uint8) for x in batches_f32]

    jpeg = A.Jpeg(A.JpegCompressionSettings(space, qrange, brange), device=local_rank)

    class Pipe:
        """one context on one stream with its own outputs (and, inside the context, its own workspace)"""
        def __init__(self, stream):
            self.stream = stream
            with torch.cuda.stream(stream):
                self.ctx = jpeg._bind()
                self.ctx.set_graph_mode(args.graph)
                self.ctx.set_sub_batches(args.sub_batches)
                for kv in args.option:
                    self.ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                self.plan = self.ctx.plan(B, H, W)
                self.out = (self.ctx.empty((B * self.plan.coeff_stride,), torch.int32), self.ctx.empty((B * self.plan.leaf_stride, 4), torch.int32),
                            self.ctx.empty((B * self.plan.state_stride,), torch.uint8), self.ctx.empty((B, 3, 4), torch.int64))
            self.pending = False
            self.input = 0

        def end(self):
            if self.pending:
                self.pending = False
                with torch.cuda.stream(self.stream):
                    jpeg.encode_end(self.ctx)

        def begin(self, which):
            self.end()
            with torch.cuda.stream(self.stream):
                jpeg.encode_begin(self.ctx, batches[which], self.plan, *self.out)
            self.pending, self.input = True, which

    # every context on a stream of its own, none on the legacy null stream: once other streams exist, launches on the null stream shift
    # the HIP-event stage attribution (colour planes +0.25 ms, blur -0.09 ms, profiles/r02_null_stream_stage_attribution.txt)
    pipes = [Pipe(torch.cuda.Stream(device=dev)) for _ in range(args.pipeline)]
    torch.cuda.synchronize()
    ctx, plan = pipes[0].ctx, pipes[0].plan
    coeffs, leaves, states, counts = pipes[0].out
    n_pipe = len(pipes)

    def input_of(i):
        # inputs alternate on EVERY context (a context that saw the same batch each time could never miss its hysteresis hint)
        return (i // n_pipe + i) & 1

    def step(i):                       # throughput loop: enqueue step i on context i % n; the step that used it before is ended first
        pipes[i % n_pipe].begin(input_of(i))

    def serial_step(i):                # one blocking call at a time on context 0
        with torch.cuda.stream(pipes[0].stream):
            jpeg.encode_into(ctx, batches[i & 1], plan, coeffs, leaves, states, counts)

    def sync():
        for p in pipes:
            p.end()
        torch.cuda.synchronize()

    def hyst_stats():
        tot = {}
        for p in pipes:
            for k, v in p.ctx.hysteresis_stats().items():
                tot[k] = (tot.get(k, 0) + v) if k != "queued" else max(tot.get(k, 0), v)
        return tot

    # ---- the measurement: W warm-up steps, then exactly K timed steps on alternating inputs, profiling off ----
    ctx.set_profiling(False)
    n_warm = max(args.warmup, 2 * n_pipe)               # every context has seen both inputs
    for i in range(n_warm):
        step(i)
    sync()
    # the first collective of the process -- after every context, stream and sub-batch stream exists (hw_queue_default) -- holds the
    # barrier behind which ONE process per node builds the checker: it exists before anything is timed and is not used until after
    if not (args.no_verify and args.no_cpu_baseline) and not args.timed_only:      # (the same decision on every rank)
        build_oracle_once(dist, int(os.environ.get("LOCAL_RANK", "0")))
    h0 = hyst_stats()
    dt_local = timed_loop(torch, dist, step, args.steps, sync)
    own_ms_per_step = timed_loop.own_seconds / args.steps * 1e3
    h1 = hyst_stats()
    px_total, dt = aggregate_throughput(dist, B * H * W * args.steps, dt_local, dev if backend == "nccl" else None)   # SUM of pixels, MAX of seconds
    value = px_total / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    last_pipe, last_batch = pipes[(args.steps - 1) % n_pipe], input_of(args.steps - 1)
    if args.timed_only:
        if rank == 0:
            emit(json.dumps({"metric": "megapixels/sec encode (Canny+quadtree+DCT+quant), 4K batch", "value": round(value, 1), "unit": "MP/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
                              "encode_calls": args.steps + n_warm, "timed_only": True}))
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- tie the number to correct output: first and last image of the LAST timed step's batch against the CPU oracle ----
    verified = None
    if not args.no_verify:                   # every rank checks its own outputs (rank 0's result goes into the line, all of them into `ranks`)
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
        enc = EncodedBatch(last_pipe.plan, *last_pipe.out)
        picks = sorted({0, B - 1})
        with ThreadPoolExecutor(max_workers=len(picks)) as ex:
            refs = list(ex.map(lambda b: O.encode_image(batches_f32[last_batch][b].cpu().numpy(), space, qrange, brange), picks))
        ok = True
        for b, ref in zip(picks, refs):
            for l in range(3):
                got = enc.layer(b, l)
                ok = ok and got["root_size"] == ref[l]["root_size"] and all(np.array_equal(got[k], ref[l][k]) for k in ("states", "leaves", "coeffs"))
        verified = {"ok": bool(ok), "images": picks, "of_batch": "A" if last_batch == 0 else "B",
                    "what": "quadtree states, leaf table and quantised zigzag coefficients of all 3 layers, bit-exact vs the CPU oracle"}

    # ---- first-contact evidence for N > 1: which ranks the collective saw, their own step times and oracle checks ----
    ranks = gather_rank_report(dist, local_rank, own_ms_per_step, B, None if verified is None else verified["ok"],
                               dev if backend == "nccl" else None, require_verified=not args.no_verify)
    ranks["collectives"] = (None if dist is None else
                            {"backend": dist.get_backend(), "what": "barriers around the timed region, all_reduce(SUM / MAX) of two float64 counters, all_gather of "
                                                                    "four float64 per rank" + ("; tensors on the GPU" if backend == "nccl" else "")})

    # ---- strictly serial figure: blocking calls on one context, nothing in flight between them ----
    serial_step(0); serial_step(1)
    dt_s = timed_loop(torch, dist, serial_step, args.steps, sync)
    _, dt_s = aggregate_throughput(dist, 0, dt_s, dev if backend == "nccl" else None)

    # ---- per-stage times: separate, untimed, profiled steps (HIP events on the launch stream inside the library) ----
    ctx.set_profiling(True)
    stage_acc, n_prof = {}, 4
    serial_step(0); serial_step(1)
    for i in range(n_prof):
        serial_step(i)
        for k, v in ctx.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    ctx.set_profiling(False)
    stage_ms = {k: v / n_prof for k, v in stage_acc.items()}

    # leaf-size histogram of this rank's last batch ("report the leaf-size histogram with every number")
    cnt = counts.cpu().numpy()
    lv = leaves.view(B, plan.leaf_stride, 4)
    leaf_hist = {}
    for l in range(3):
        n_l = torch.from_numpy(cnt[:, l, 1].copy()).to(dev)
        lo = int(plan.leaf_off[l])
        cap = int(cnt[:, l, 1].max())
        sz = lv[:, lo:lo + cap, 2]
        valid = torch.arange(cap, device=dev)[None, :] < n_l[:, None]
        s = brange[0]
        while s <= brange[1]:
            leaf_hist[s] = leaf_hist.get(s, 0) + int(((sz == s) & valid).sum().item())
            s *= 2
    leaf_area = sum(k * k * v for k, v in leaf_hist.items())

    # ---- kernels of the step, their algorithmic bytes, and the roofline of the dominant one ----
    local_px = B * H * W
    algo = {k: v * local_px for k, v in ALGO_BYTES_PER_PX.items()}
    if args.ingest == "u8":
        algo["color_planes"] -= 9.0 * local_px
    kernels = {k: stage_ms.get(k, 0.0) for k in ("color_planes", "clahe_blur", "sobel_nms", "hysteresis", "quadtree")}
    for sz, n_leaves in leaf_hist.items():
        ms = stage_ms.get(f"dct{sz}", 0.0)
        if ms > 0 and n_leaves > 0:
            kernels[f"dct{sz}"] = ms
            algo[f"dct{sz}"] = 8.0 * sz * sz * n_leaves          # f32 in + int32 out per coefficient
    head = git_head()
    # HBM bytes and VALU instruction counts come from rocprofv3 PMC passes of this same command (profiles/, tools/profiling/pmc.py);
    # they cannot be collected inside a normal run, so the line says which profile they are from and for which commit
    def newest(suffix):
        names = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix) and f[:1] == "r" and f[1:3].isdigit())
        return names[-1] if names else suffix
    traffic_file, valu_file = newest("_hbm_traffic.json"), newest("_pmc_valu.json")
    tj = load_profile_json(traffic_file)
    vj = load_profile_json(valu_file)
    src_hash = source_hash()
    # every stage of this run must resolve to at least one kernel of the profile it is priced with (a renamed kernel would otherwise
    # silently drop out of `traffic` / `valu`)
    profile_gaps = {}
    for fname, j in ((traffic_file, tj), (valu_file, vj)):
        if j and j.get("kernels"):
            missing = [st for st in KERNEL_OF_STAGE if (st in ("color_planes", "clahe_blur", "sobel_nms", "hysteresis", "quadtree") or
                                                        (st.startswith("dct") and brange[0] <= int(st[3:]) <= brange[1]))
                       and not kernels_of_stage(st, j["kernels"])]
            if missing:
                profile_gaps[fname] = missing

    def same_shape(j):
        return bool(j) and (j.get("batch"), j.get("height"), j.get("width")) == (B, H, W) and j.get("space", "YCbCr") == space and \
            tuple(j.get("blocks", (4, 64))) == brange

    def stale(j):
        # a profile is valid for the kernel sources it was taken with (source_hash, stored by tools/profiling/pmc.py); older
        # profiles only carry the git commit
        if j.get("src_hash"):
            return j["src_hash"] != src_hash
        return (j.get("head") != head) if (head and j.get("head")) else "unknown (profile predates source hashes and there is no git on this box)"

    def src_of(j, fname):
        return {"file": "profiles/" + fname, "profiled_commit": j.get("head"), "profiled_src_hash": j.get("src_hash"), "this_commit": head,
                "this_src_hash": src_hash, "stale": stale(j)}

    def roofline_of(stage):
        achieved = algo[stage] / (kernels[stage] * 1e-3) / 1e9 if kernels[stage] > 0 else 0.0
        traffic, traffic_src = None, None
        if same_shape(tj):
            hit = [tj["kernels"][k]["hbm_bytes"] for k in kernels_of_stage(stage, tj["kernels"])]
            if hit:
                traffic = sum(hit)
                traffic_src = src_of(tj, traffic_file)
        r = {"bound": "hbm", "kernel": {"quadtree": "k_qt_upper+count+scan+emit", "hysteresis": "k_hyst_pass0+k_hyst_bulk+k_hyst_drain"}.get(stage, KERNEL_OF_STAGE[stage][0]),
             "stage": stage, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
             "algorithmic_bytes_per_launch": algo[stage], "avg_launch_ms": round(kernels[stage], 4)}
        v = None
        if same_shape(vj):
            hit = [vj["kernels"][k] for k in kernels_of_stage(stage, vj["kernels"])]
            if hit:
                insts = sum(x["valu_insts_per_launch"] for x in hit)       # wave-level VALU instructions of one launch
                plane_px = 1.5 * local_px
                # issue limit measured here (profiles/r02_valu_issue_ubench.txt): one plain 32-bit VALU instruction per SIMD every ~1.1 ns
                # when >= 2 waves share the SIMD; shifts / conversions / SDWA / 3-operand integer / packed ops take ~1.75 ns.
                # frac = time the instructions need at the plain rate / measured time.
                t_issue = insts / N_SIMD * 1.1e-9
                v = {"kernel": r["kernel"], "wave_instructions_per_launch": insts, "instructions_per_plane_px": round(insts * 64 / plane_px, 1),
                     "frac_of_issue_peak": round(t_issue / (kernels[stage] * 1e-3), 3), "issue_ns_per_simd_instruction": 1.1,
                     "source": src_of(vj, valu_file)}
        return r, v

    ranked = sorted(kernels, key=kernels.get, reverse=True)
    dom = ranked[0]
    roofline, valu = roofline_of(dom)
    # the two longest kernels of this path are within a few per cent of each other (the HBM-bound colour stage and the issue-bound blur):
    # which one is "dominant" can change from run to run, so the runner-up is reported the same way
    runner_up = None
    if len(ranked) > 1:
        r2, v2 = roofline_of(ranked[1])
        runner_up = {"roofline": r2, "valu": v2}
    whole_bpp = WHOLE_PATH_BYTES_PER_PX - (9.0 if args.ingest == "u8" else 0.0)
    whole = whole_bpp * local_px / (ms_per_step * 1e-3) / 1e9
    per_stage = {k: {"ms": round(v, 4), "GBps": round(algo[k] / (v * 1e-3) / 1e9, 1) if v > 0 else None} for k, v in kernels.items()}

    out = {
        "metric": "megapixels/sec encode (Canny+quadtree+DCT+quant), 4K batch",
        "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32",
        "data": ("synthetic ('mixed' images generated on the GPU, SURVEY.md 8d recipe); two batches of different seeds alternate across steps"
                 if args.data == "synthetic" else
                 "natural: the reference's own test images (baboon, peppers, house, jelly_beans, LIVE bikes / buildings) mirror-tiled to the image "
                 "size, uint8 levels / 255; two batches alternate across steps -- a labelled variant, not the headline recipe"),
        "config": {"workload": f"{B} x {W}x{H} {'uint8' if args.ingest == 'u8' else 'float32'} RGB per GPU, {space}, blocks {brange[0]}-{brange[1]}, quality {qrange[0]}-{qrange[1]} "
                               + ("(BASELINE config 4: 512 4K images / 8 GPUs)" if (B, H, W, space, tuple(brange), args.data) == (64, H4K, W4K, "YCbCr", (4, 64), "synthetic") else "(not the headline workload)"),
                   "images_per_gpu": B, "height": H, "width": W, "color_space": space,
                   "block_size_range": list(brange), "quality_range": list(qrange), "steps_in_flight": n_pipe,
                   "options": args.option or None},
        "roofline": roofline,
        "valu": valu,
        "runner_up": runner_up,
        "verified": verified,
        "profile_gaps": profile_gaps or None,
        "ranks": ranks,
        "hysteresis": {"timed_calls": h1["calls"] - h0["calls"], "launches_per_part": "2 (up to 8192 tiles), 3 (up to 32768), 4 (larger parts: the bench's sub-batches)",
                       "tiles_through_the_work_queue_last_call": h1["queued"], "tiles": int(B * sum(-(-plan.layer_h[l] // 64) * -(-plan.layer_w[l] // 64) for l in range(3))),
                       "what": "a pass over every 64 x 64 tile that flags dirtied neighbours, for large parts two bulk launches over the flagged tiles, then a "
                               "device-side work queue drained to the fix-point by one small persistent launch: no pass count guessed by the host, nothing "
                               "read back, nothing to repair"},
        "pipeline": {"contexts": n_pipe, "what": "timed step i is enqueued (aej_encode_batch_begin) on context i % n, each context on its own stream with its own "
                                                  "output buffers and workspace, after the step that used that context before has been ended (aej_encode_batch_end: "
                                                  "waited for and verified); all K steps are complete inside the timed region",
                     "serial_ms_per_step": round(dt_s / args.steps * 1e3, 3),
                     "serial_note": "the same K steps as blocking aej_encode_batch calls on one context (nothing in flight between calls)"},
        "graph": dict(ctx.graph_stats(), mode=args.graph),
        "sub_batches": {"mode": args.sub_batches, "split_calls": sum(p.ctx.split_calls() for p in pipes),
                        "note": "timed steps run as sub-batches on private streams when split_calls > 0; the per-stage times below come from separate, "
                                "unsplit profiled steps (stages of different sub-batches overlap in the timed region, so they add up to more than ms_per_step)"},
        "whole_path": {"bytes_per_px": whole_bpp, "achieved_GBps": round(whole, 1), "frac_of_hbm_peak": round(whole / HBM_PEAK_GBS, 4)},
        "stages": per_stage,
        "stage_ms_source": f"{n_prof} separate profiled steps after the timed region (sum {sum(stage_ms.values()):.3f} ms)",
        "leaves_per_image": {"luma": int(cnt[:, 0, 1].mean()), "chroma": int(cnt[:, 1:, 1].mean())},
        "leaf_histogram": {"per_image": {str(k): round(v / B, 1) for k, v in leaf_hist.items()},
                           "area_share": {str(k): round(k * k * v / leaf_area, 4) for k, v in leaf_hist.items()}},
    }
    # DCT per block size: time, bytes moved per second (8 B per coefficient) and, for the MFMA sizes, the
    # fraction of the 157.3 TFLOP/s float32 MFMA peak (4 s^3 FLOP per leaf: two s x s x s products)
    dct_sizes = {}
    for sz, n_leaves in leaf_hist.items():
        ms = stage_ms.get(f"dct{sz}", 0.0)
        if ms <= 0 or n_leaves == 0:
            continue
        gbps = 8.0 * sz * sz * n_leaves / (ms * 1e-3) / 1e9
        e = {"ms": round(ms, 4), "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 3)}
        if sz >= 16:
            tf = 4.0 * sz ** 3 * n_leaves / (ms * 1e-3) / 1e12
            e.update({"TFLOPs": round(tf, 1), "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 3)})
        if sz == 64:
            e["kernel"] = ("k_dct_mfma<64> (four waves per leaf): what a sub-batched / pipelined call runs, the headline path included; a call that has the "
                           "device to itself runs k_dct64_wave (one wave per leaf): 0.80-0.81 ms = 0.66 of the MFMA peak, "
                           "profiles/r03_dct64_kernels_alone.txt, EXPERIMENTS.md 4c")
        dct_sizes[str(sz)] = e
    out["dct_by_block_size"] = dct_sizes
    # Canny chain a-3 .. a-8 (5 B per plane pixel = float32 plane in, uint8 edge map out -> 7.5 B per image pixel)
    canny_ms = sum(stage_ms.get(k, 0.0) for k in ("clahe_lut", "clahe_blur", "thresholds", "sobel_nms", "hysteresis"))
    if canny_ms > 0:
        gbs = 7.5 * local_px / (canny_ms * 1e-3) / 1e9
        out["canny_chain"] = {"ms": round(canny_ms, 4), "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}

    # ---- CPU baselines on this box's host cores (rank 0, bounded sample) ----
    # "port": the C oracle (a scalar port of the reference algorithm), one image per thread (ctypes releases the GIL inside the
    #   C call) -- the fan-out the reference's own sweep uses (one image per worker process, metrics_computation.py:253);
    # "reference_structured": SURVEY 8d's figure -- one Python thread, per-layer stages, per-node quadtree tests and per-leaf
    #   Python loops exactly as jpeg.py:393-404,471,499-502,581-585, native calls where the reference calls OpenCV / numba.
    if rank == 0 and not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        from oracle import reference_structured as RS
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        # "across all host cores" (BASELINE.md B2): one image per thread on min(cores this process may use, 64, B) threads -- 64 images are
        # resident, each encode holds a few hundred MB -- and the 16-thread figure of rounds 1-3 beside it (a container's CPU quota can be
        # smaller than its affinity mask: the better of the two is the baseline, with the thread count that produced it)
        wide = max(1, min(args.cpu_threads or min(avail, 64), B))
        imgs = batches_f32[0][:min(B, wide)].cpu().numpy()
        t0 = time.perf_counter()
        O.encode_image(imgs[0], space, qrange, brange)
        t1 = time.perf_counter() - t0

        def port_rate(threads):
            n = min(len(imgs), threads)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=threads) as ex:
                list(ex.map(lambda im: O.encode_image(im, space, qrange, brange), [imgs[i] for i in range(n)]))
            dt_ = time.perf_counter() - t0
            return {"threads": threads, "images": n, "seconds": round(dt_, 2), "MP/s": round(n * H * W / dt_ / 1e6, 2)}
        runs = [port_rate(wide)] + ([port_rate(16)] if wide > 16 else [])
        best = max(runs, key=lambda r: r["MP/s"])
        t0 = time.perf_counter()
        RS.encode_image(imgs[0], space, qrange, brange)
        t_rs = time.perf_counter() - t0
        t_fan, n_fan = RS.fan_out(imgs[:wide], space, qrange, brange, wide)      # (ii): one image per worker process
        out["cpu_baseline"] = {"value": best["MP/s"], "unit": "MP/s", "cores": best["threads"], "kind": "port",
                               "sample": f"{best['images']} of the {B} bench images ({W}x{H}), one image per thread on {best['threads']} threads, whole path a-1..a-15 in "
                                         f"the C oracle, {best['seconds']} s; single core: 1 image in {t1:.1f} s",
                               "runs": runs, "single_core_value": round(H * W / t1 / 1e6, 2), "host_cpus": os.cpu_count(), "usable_cpus": avail,
                               "reference_structured": {"value": round(H * W / t_rs / 1e6, 2), "unit": "MP/s", "cores": 1, "kind": "port",
                                                        "sample": f"1 bench image ({W}x{H}), {t_rs:.1f} s: the reference's control structure (one Python thread, "
                                                                  "per-node quadtree tests, per-leaf pad / DCT / quantise / zigzag loops) with the C oracle standing in for "
                                                                  "its OpenCV / numba calls -- not the reference binary stack (cv2 / numba absent)",
                                                        "fan_out": {"value": round(n_fan * H * W / t_fan / 1e6, 2), "unit": "MP/s", "cores": n_fan,
                                                                    "sample": f"{n_fan} bench images, one per worker process (fresh interpreters, as the reference's "
                                                                              f"sweep fans out, metrics_computation.py:253), {t_fan:.1f} s wall including interpreter start-up"}}}
    if rank == 0:
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    # a rank whose outputs differ from the oracle's makes the whole run fail (every rank has the same `ranks` dict)
    if not ranks["all_verified"]:
        sys.stderr.write(f"bench.py: oracle check failed on rank(s) {[i for i, v in enumerate(ranks['verified_ok_by_rank']) if v is False]}\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
