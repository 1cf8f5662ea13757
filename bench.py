#!/usr/bin/env python3
"""bench.py -- encode-hot-path throughput on MI355X (BASELINE.json metric: megapixels/s, 4K batch).

One "step" = one pass of the whole encode hot path (colour convert -> chroma down-sample -> Canny chain -> quadtree -> DCT ->
quantise -> zigzag) over one device-resident batch of 64 synthetic 3840x2160 float32 RGB
images per GPU (BASELINE config 4: 512 4K images over 8 GPUs = 64 per GPU; weak scaling).  Inputs are in HBM before the timed
region; outputs stay in HBM.

    python bench.py                      # 1 GPU, 64 x 4K, 10 steps
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The measurement (main(), section "the measurement"): W warm-up steps, then EXACTLY K steps inside timed_loop() -- barrier +
synchronize on both sides -- rotating over `--pipeline` contexts (tools/benchlib/workload.py: what a step is, how the two input batches
alternate).  Nothing is profiled, verified or read back inside the timed region; after it the outputs of the LAST timed step are
compared with the CPU oracle (`verified`), stage times come from separate profiled steps, and rank 0 prints ONE JSON line with
* `roofline`: the longest kernel of the chain a step waits for (the colour stage is a background kernel: `runner_up`), its
  ALGORITHMIC bytes per launch over its HIP-event time, against the 8 TB/s HBM peak; `traffic` from the committed PMC profile;
* `whole_path`: the same for the whole step, with the measured HBM traffic and the vector-issue share of the step;
* `other_configs`: BASELINE configs 2, 3, 5 and the natural-image / 8-bit-ingest variants of the headline, each timed the same way
  (fewer steps) and each with one image checked against the oracle -- the reference sweeps images x settings in one run the same
  way (test/analysis/metrics_computation.py:307-324);
* `cpu_baseline`: the C oracle on this box's host cores (tools/benchlib/cpu_baseline.py), a reported baseline, not the target.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from benchlib.data import natural_batch, synth_batch                      # noqa: E402,F401  (tests and tools import them from here)
from benchlib.report import (ALGO_BYTES_PER_PX, BACKGROUND_STAGES, HBM_PEAK_GBS, N_SIMD, VALU_ISSUE_NS,      # noqa: E402
                             WHOLE_PATH_BYTES_PER_PX, CounterProfiles, dct_by_block_size, roofline_of, source_hash)  # noqa: F401

H4K, W4K = 2160, 3840
METRIC = "megapixels/sec encode (Canny+quadtree+DCT+quant), 4K batch"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
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
                         "the image size (a labelled variant: textures change the leaf mix and the hysteresis work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing oracle comparison")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE configurations after the headline measurement")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads for cpu_baseline; 0 = min(cores this process may use, 64)")
    ap.add_argument("--graph", type=int, choices=[0, 1, 2], default=0,
                    help="aej_set_graph_mode: 0 never replay a captured hipGraph (the library default), 1 automatic (calls of at most 8 Mpx), 2 whenever possible")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="aej_set_sub_batches: 0 automatic (the library default: 4 sub-batches on private streams for calls of at least 64 Mpx), 1 never, 2..8. "
                         "Applies to the blocking calls; the pipelined steps take --pipelined-sub-batches")
    ap.add_argument("--pipelined-sub-batches", type=int, default=-1,
                    help="aej_set_sub_batches for the steps that rotate over the contexts: -1 = 2 for calls of at least 384 Mpx on four or more contexts "
                         "(with four calls in flight two parts per call fill the chip best: profiles/r05_sched_sweep.txt), otherwise --sub-batches")
    ap.add_argument("--pipeline", type=int, choices=[1, 2, 3, 4, 5, 6, 8], default=4,
                    help="contexts (each on its own stream, with its own output buffers and workspace) the timed steps rotate over; 1 = blocking calls")
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
    process uses one stream per context plus the library's sub-batch streams: 15 with the null stream (four contexts x (1 + 2), two more for the
    blocking-call shape; round 4: three x (1 + 4) + 1 = 16), so 16 queues give every stream its own (tools/profiling/queue_map.py shows the mapping)
    -- until somebody else creates streams first.  A live RCCL communicator does (measured on one GPU, one rank, profiles/
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


# BASELINE.json's other configurations on one GPU (the per-GPU share of the multi-GPU ones) and the two labelled variants of the headline
# workload.  `steps`: timed steps (enough for the timed region to be tens of milliseconds); `check`: the image compared with the oracle;
# `contexts`: calls in flight where that is not the headline's (three contexts x the library's automatic sub-batches for the smaller calls, four
# contexts x two sub-batches for the 64 x 4K ones: profiles/r05_sched_sweep.txt).
OTHER_CONFIGS = [
    {"name": "BASELINE configs[1]: single 1920x1080 image, adaptive 4-64 blocks, YCbCr", "batch": 1, "H": 1080, "W": 1920, "space": "YCbCr",
     "blocks": (4, 64), "data": "synthetic", "ingest": "f32", "steps": 200, "blocking_steps": 100, "check": 0, "contexts": 3},
    {"name": "BASELINE configs[2]: batch of 64 1080p images, adaptive 4-64 blocks", "batch": 64, "H": 1080, "W": 1920, "space": "YCbCr",
     "blocks": (4, 64), "data": "synthetic", "ingest": "f32", "steps": 8, "blocking_steps": 4, "check": 63, "contexts": 3},
    {"name": "BASELINE configs[4] per-GPU share: 8 x 8K (7680x4320), full 4-128 block range, OKLAB", "batch": 8, "H": 4320, "W": 7680, "space": "OKLAB",
     "blocks": (4, 128), "data": "synthetic", "ingest": "f32", "steps": 8, "blocking_steps": 4, "check": 7, "contexts": 3},
    {"name": "headline workload on natural images (the reference's test images mirror-tiled to 4K)", "batch": 64, "H": H4K, "W": W4K, "space": "YCbCr",
     "blocks": (4, 64), "data": "natural", "ingest": "f32", "steps": 8, "blocking_steps": 4, "check": 17},
    {"name": "headline workload with 8-bit ingest (uint8 RGB in, 3 B/px: how real inputs arrive, image.py:80)", "batch": 64, "H": H4K, "W": W4K,
     "space": "YCbCr", "blocks": (4, 64), "data": "synthetic", "ingest": "u8", "steps": 8, "blocking_steps": 4, "check": 31},
]


def pipelined_sub_batches(args, n_pipe, B, H, W):
    """aej_set_sub_batches of the pipelined steps (see --pipelined-sub-batches)"""
    if args.pipelined_sub_batches >= 0:
        return args.pipelined_sub_batches
    if args.sub_batches == 0 and n_pipe >= 4 and B * H * W >= 384_000_000:
        return 2
    return args.sub_batches


def run_other_configs(torch, A, dev, args, O, qrange, time_budget_s=150.0):
    """After the headline measurement, in the same process: every entry of OTHER_CONFIGS timed like the headline (warm-up, then exactly
    `steps` steps between two synchronisations, three or four contexts in flight) plus the same steps as blocking calls, one image of the last
    timed step checked against the CPU oracle.  Single GPU only (they are per-GPU shares; the multi-GPU line carries the headline)."""
    from benchlib.workload import Workload
    out, t_start = [], time.perf_counter()
    for cfg in OTHER_CONFIGS:
        if time.perf_counter() - t_start > time_budget_s:
            out.append({"workload": cfg["name"], "skipped": f"time budget of {time_budget_s:.0f} s for other_configs used up"})
            continue
        B, H, W = cfg["batch"], cfg["H"], cfg["W"]
        make = synth_batch if cfg["data"] == "synthetic" else natural_batch
        batches = [make(torch, B, H, W, 20250718, dev), make(torch, B, H, W, 20250718 + 1_000_000, dev)]
        n_pipe = cfg.get("contexts", args.pipeline)
        wl = Workload(torch, A, dev, batches, space=cfg["space"], qrange=qrange, brange=cfg["blocks"], ingest=cfg["ingest"], n_pipe=n_pipe,
                      graph=args.graph, sub_batches=args.sub_batches, pipelined_sub_batches=pipelined_sub_batches(args, n_pipe, B, H, W), options=args.option)
        wl.warm(6)
        dt = timed_loop(torch, None, wl.step, cfg["steps"], wl.sync)
        verified = None
        if O is not None:
            verified = wl.verify(O, cfg["steps"], [cfg["check"]])
        wl.serial_step(0); wl.serial_step(1)
        dt_s = timed_loop(torch, None, wl.serial_step, cfg["blocking_steps"], wl.sync)
        ms, ms_s = dt / cfg["steps"] * 1e3, dt_s / cfg["blocking_steps"] * 1e3
        out.append({"workload": cfg["name"], "images": B, "height": H, "width": W, "color_space": cfg["space"], "block_size_range": list(cfg["blocks"]),
                    "data": cfg["data"], "ingest": cfg["ingest"], "steps": cfg["steps"], "contexts": n_pipe, "pipelined_sub_batches": wl.pipelined_sub_batches,
                    "ms_per_step": round(ms, 4), "MP/s": round(B * H * W / ms / 1e3, 1),
                    "blocking_ms_per_call": round(ms_s, 4), "blocking_MP/s": round(B * H * W / ms_s / 1e3, 1), "verified": verified})
        wl.close()
        del wl, batches
        torch.cuda.empty_cache()
    return out


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
    from benchlib.workload import Workload
    H, W = args.height, args.width
    B, seed_a, seed_b, scaling = local_batch_and_seed(args, rank, world)
    if B < 1:
        raise SystemExit("no images for this rank")
    space, qrange, brange = args.space, tuple(args.quality), tuple(args.blocks)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else None
    make_batch = synth_batch if args.data == "synthetic" else natural_batch
    wl = Workload(torch, A, dev, [make_batch(torch, B, H, W, seed_a, dev), make_batch(torch, B, H, W, seed_b, dev)], space=space, qrange=qrange,
                  brange=brange, ingest=args.ingest, n_pipe=args.pipeline, graph=args.graph, sub_batches=args.sub_batches,
                  pipelined_sub_batches=pipelined_sub_batches(args, args.pipeline, B, H, W), options=args.option)
    n_pipe = wl.n_pipe

    # ---- the measurement: W warm-up steps, then exactly K timed steps on alternating inputs, profiling off ----
    wl.ctx.set_profiling(False)
    n_warm = wl.warm(args.warmup)
    # the first collective of the process -- after every context, stream and sub-batch stream exists (hw_queue_default) -- holds the
    # barrier behind which ONE process per node builds the checker: it exists before anything is timed and is not used until after
    O = None
    if not (args.no_verify and args.no_cpu_baseline) and not args.timed_only:      # (the same decision on every rank)
        O = build_oracle_once(dist, int(os.environ.get("LOCAL_RANK", "0")))
    h0 = wl.hyst_stats()
    dt_local = timed_loop(torch, dist, wl.step, args.steps, wl.sync)
    own_ms_per_step = timed_loop.own_seconds / args.steps * 1e3
    h1 = wl.hyst_stats()
    px_total, dt = aggregate_throughput(dist, wl.pixels_per_step * args.steps, dt_local, coll_dev)   # SUM of pixels, MAX of seconds
    value = px_total / dt / 1e6
    ms_per_step = dt / args.steps * 1e3
    if args.timed_only:
        if rank == 0:
            emit(json.dumps({"metric": METRIC, "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(ms_per_step, 3), "encode_calls": args.steps + n_warm, "timed_only": True}))
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- tie the number to correct output: first and last image of the LAST timed step's batch against the CPU oracle (every rank) ----
    verified = None if args.no_verify else wl.verify(O, args.steps, sorted({0, B - 1}))

    # ---- first-contact evidence for N > 1: which ranks the collective saw, their own step times and oracle checks ----
    ranks = gather_rank_report(dist, local_rank, own_ms_per_step, B, None if verified is None else verified["ok"], coll_dev,
                               require_verified=not args.no_verify)
    ranks["collectives"] = (None if dist is None else
                            {"backend": dist.get_backend(), "what": "barriers around the timed region, all_reduce(SUM / MAX) of two float64 counters, all_gather of "
                                                                    "four float64 per rank" + ("; tensors on the GPU" if backend == "nccl" else "")})

    # ---- strictly serial figure: blocking calls on one context, nothing in flight between them ----
    wl.serial_step(0); wl.serial_step(1)
    dt_s = timed_loop(torch, dist, wl.serial_step, args.steps, wl.sync)
    _, dt_s = aggregate_throughput(dist, 0, dt_s, coll_dev)

    stage_ms = wl.stage_ms(4)
    leaf_hist, cnt = wl.leaf_histogram()
    leaf_area = sum(k * k * v for k, v in leaf_hist.items())
    plan = wl.plan

    # ---- kernels of the step, their algorithmic bytes, and the roofline of the dominant one ----
    local_px = wl.pixels_per_step
    algo = {k: v * local_px for k, v in ALGO_BYTES_PER_PX.items()}
    if args.ingest == "u8":
        algo["color_planes"] -= 9.0 * local_px
    kernels = {k: stage_ms.get(k, 0.0) for k in ("color_planes", "clahe_blur", "sobel_nms", "hysteresis", "quadtree")}
    for sz, n_leaves in leaf_hist.items():
        ms = stage_ms.get(f"dct{sz}", 0.0)
        if ms > 0 and n_leaves > 0:
            kernels[f"dct{sz}"] = ms
            algo[f"dct{sz}"] = 8.0 * sz * sz * n_leaves          # f32 in + int32 out per coefficient
    prof = CounterProfiles((B, H, W, space, brange))
    # the dominant kernel: the longest launch of the chain a step waits for.  The colour stage runs in the background of the other calls
    # in flight and is reported as `runner_up`, whatever its own time (round 4: the two swapped places on a 5 % wobble)
    foreground = sorted((k for k in kernels if k not in BACKGROUND_STAGES), key=kernels.get, reverse=True)
    roofline, valu = roofline_of(foreground[0], algo, kernels, local_px, prof)
    roofline["rule"] = "longest kernel of the foreground chain (every stage but the background colour kernel), HIP-event time of blocking calls"
    r2, v2 = roofline_of(BACKGROUND_STAGES[0], algo, kernels, local_px, prof)
    runner_up = {"roofline": r2, "valu": v2, "why": "the background stage: its launch overlaps the other stages of the calls in flight"}
    whole_bpp = WHOLE_PATH_BYTES_PER_PX - (9.0 if args.ingest == "u8" else 0.0)
    whole = whole_bpp * local_px / (ms_per_step * 1e-3) / 1e9
    step_traffic, step_valu = prof.whole_step()
    whole_path = {"bytes_per_px": whole_bpp, "achieved_GBps": round(whole, 1), "frac_of_hbm_peak": round(whole / HBM_PEAK_GBS, 4),
                  "traffic": step_traffic,
                  "traffic_over_algorithmic": round(step_traffic / (whole_bpp * local_px), 3) if step_traffic else None,
                  "traffic_GBps": round(step_traffic / (ms_per_step * 1e-3) / 1e9, 1) if step_traffic else None,
                  "traffic_frac_of_peak": round(step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if step_traffic else None,
                  "traffic_source": prof.src_of(prof.tj, prof.traffic_file) if step_traffic else None,
                  "valu_wave_instructions": step_valu,
                  "valu_frac_of_step": round(step_valu / N_SIMD * VALU_ISSUE_NS * 1e-9 / (ms_per_step * 1e-3), 3) if step_valu else None,
                  "valu_source": prof.src_of(prof.vj, prof.valu_file) if step_valu else None,
                  "note": "traffic / valu: every kernel of one blocking call summed (rocprofv3 PMC passes of this command, tools/profiling/pmc.py), "
                          "divided by THIS run's step time; valu_frac_of_step = instructions x %.1f ns / %d SIMDs / step" % (VALU_ISSUE_NS, N_SIMD)}
    per_stage = {k: {"ms": round(v, 4), "GBps": round(algo[k] / (v * 1e-3) / 1e9, 1) if v > 0 else None} for k, v in kernels.items()}
    headline = (B, H, W, space, tuple(brange), args.data, args.ingest) == (64, H4K, W4K, "YCbCr", (4, 64), "synthetic", "f32")

    out = {
        "metric": METRIC, "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": ("synthetic ('mixed' images generated on the GPU, SURVEY.md 8d recipe); two batches of different seeds alternate across steps"
                 if args.data == "synthetic" else
                 "natural: the reference's own test images (baboon, peppers, house, jelly_beans, LIVE bikes / buildings) mirror-tiled to the image "
                 "size, uint8 levels / 255; two batches alternate across steps -- a labelled variant, not the headline recipe"),
        "config": {"workload": f"{B} x {W}x{H} {'uint8' if args.ingest == 'u8' else 'float32'} RGB per GPU, {space}, blocks {brange[0]}-{brange[1]}, "
                               f"quality {qrange[0]}-{qrange[1]} " + ("(BASELINE config 4: 512 4K images / 8 GPUs)" if headline else "(not the headline workload)"),
                   "images_per_gpu": B, "height": H, "width": W, "color_space": space, "block_size_range": list(brange),
                   "quality_range": list(qrange), "steps_in_flight": n_pipe, "options": args.option or None},
        "roofline": roofline, "valu": valu, "runner_up": runner_up, "whole_path": whole_path,
        "verified": verified, "profile_gaps": prof.gaps(brange), "ranks": ranks,
        "hysteresis": {"timed_calls": h1["calls"] - h0["calls"], "tiles_through_the_work_queue_last_call": h1["queued"],
                       "tiles": int(B * sum(-(-plan.layer_h[l] // 64) * -(-plan.layer_w[l] // 64) for l in range(3))),
                       "what": "completes on the device: a pass over every 64 x 64 tile, bulk launches over the flagged tiles, then a device-side "
                               "work queue drained to the fix-point by one small persistent launch; nothing read back"},
        "pipeline": {"contexts": n_pipe, "serial_ms_per_step": round(dt_s / args.steps * 1e3, 3),
                     "serial_note": "the same K steps as blocking aej_encode_batch calls on one context (nothing in flight between calls)"},
        "graph": dict(wl.ctx.graph_stats(), mode=args.graph),
        "sub_batches": {"mode": args.sub_batches, "pipelined_steps": wl.pipelined_sub_batches, "split_calls": sum(p.ctx.split_calls() for p in wl.pipes),
                        "note": "aej_set_sub_batches: mode (0 = the library's automatic choice, 4 here) for the blocking calls, pipelined_steps for the timed steps"},
        "stages": per_stage,
        "stage_ms_source": f"4 separate profiled blocking steps after the timed region (sum {sum(stage_ms.values()):.3f} ms); stages of different "
                           "sub-batches overlap in the timed region, so they add up to more than ms_per_step",
        "leaves_per_image": {"luma": int(cnt[:, 0, 1].mean()), "chroma": int(cnt[:, 1:, 1].mean())},
        "leaf_histogram": {"per_image": {str(k): round(v / B, 1) for k, v in leaf_hist.items()},
                           "area_share": {str(k): round(k * k * v / leaf_area, 4) for k, v in leaf_hist.items()}},
        "dct_by_block_size": dct_by_block_size(leaf_hist, stage_ms),
    }
    # Canny chain a-3 .. a-8 (5 B per plane pixel = float32 plane in, uint8 edge map out -> 7.5 B per image pixel)
    canny_ms = sum(stage_ms.get(k, 0.0) for k in ("clahe_lut", "clahe_blur", "thresholds", "sobel_nms", "hysteresis"))
    if canny_ms > 0:
        gbs = 7.5 * local_px / (canny_ms * 1e-3) / 1e9
        out["canny_chain"] = {"ms": round(canny_ms, 4), "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline on this box's host cores (rank 0, bounded sample), before the headline batches are released ----
    if rank == 0 and not args.no_cpu_baseline:
        from benchlib import cpu_baseline
        out["cpu_baseline"] = cpu_baseline.measure(wl.batches_f32[0], space, qrange, brange, args.cpu_threads)

    # ---- the other BASELINE configurations, same process, after the headline (single GPU, headline invocation only) ----
    # (A / B invocations -- --option, --no-verify, another shape -- do not pay for them)
    if rank == 0 and world == 1 and headline and not (args.no_other_configs or args.no_verify or args.option):
        wl.close()
        out["other_configs"] = run_other_configs(torch, A, dev, args, O, qrange)
    if rank == 0:
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    # a rank whose outputs differ from the oracle's makes the whole run fail (every rank has the same `ranks` dict)
    bad_other = [c["workload"] for c in out.get("other_configs", []) if c.get("verified") and not c["verified"]["ok"]]
    if not ranks["all_verified"] or bad_other:
        sys.stderr.write(f"bench.py: oracle check failed on rank(s) {[i for i, v in enumerate(ranks['verified_ok_by_rank']) if v is False]} {bad_other}\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
