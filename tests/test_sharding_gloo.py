"""CPU: the multi-GPU path (independent shards + counter reduction) with world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adaptive_edge_aware_jpeg_amd.sharding import aggregate_throughput, shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(7, rank, world)
    px, sec = aggregate_throughput(dist, (hi - lo) * 1000, 1.0 + rank)
    q.put((rank, lo, hi, px, sec))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregation_over_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]
    assert all(r[3] == 7000 and r[4] == 2.0 for r in res)      # SUM of pixels, MAX of seconds on every rank


def test_single_process_passthrough():
    assert aggregate_throughput(None, 10, 2.5) == (10, 2.5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("extra,scaling,images", [([], "weak", [4, 4]), (["--total-images", "7"], "strong", [4, 3])])
def test_bench_multi_rank_control_flow_rehearsal(extra, scaling, images):
    """bench.py's own N > 1 control flow -- env parsing, rendezvous on 127.0.0.1, per-rank shard and seeds, barrier-bracketed
    timing, SUM / MAX reduction, rank-0-only JSON line -- as two real processes over gloo, with a sleep in place of the GPU step
    (--rehearse-control-flow: no GPU work, no measurement).  The driver runs the same code path with backend nccl on 8 GPUs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AEJ_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--rehearse-control-flow", "--gpus", "2", "--steps", "3",
                                       "--warmup", "1", "--batch", "4", "--height", "100", "--width", "200"] + extra,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    json_lines = [[ln for ln in o[0].splitlines() if ln.startswith("{")] for o in outs]
    assert len(json_lines[0]) == 1 and json_lines[1] == []           # only rank 0 prints the result line
    # ... and NOTHING else reaches stdout: bench.py points file descriptor 1 at stderr (gloo chats there, RCCL prints a banner there) and
    # writes its line to a saved copy of the real stdout
    assert outs[0][0].strip() == json_lines[0][0] and outs[1][0].strip() == "", [o[0][-500:] for o in outs]
    line = json.loads(json_lines[0][0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == scaling
    assert line["pixels_total"] == sum(images) * 100 * 200 * 3       # SUM over ranks
    assert line["seconds_max"] >= 3 * 0.004 * 0.9                      # MAX over ranks: rank 1 sleeps twice as long
    assert line["rank0_images"] == images[0] and line["rank0_seeds"][0] == 20250718
    # first-contact evidence: the collective saw both ranks, each with its own step time
    rk = line["ranks"]
    assert rk["world_size"] == 2 and rk["n_ranks_seen"] == 2 and rk["local_ranks_seen"] == [0, 1]
    assert rk["ms_per_step_by_rank"][1] > rk["ms_per_step_by_rank"][0] > 0 and rk["all_verified"] and rk["images_per_step_by_rank"] == images


def test_bench_starts_its_own_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with NO rank variables in the environment starts its two ranks itself (a child
    `python -m torch.distributed.run`, never an exec), relays rank 0's JSON line and exits with the child's code; a --gpus that
    disagrees with an external launcher's WORLD_SIZE fails loudly."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env["AEJ_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-control-flow", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "3", "--height", "64", "--width", "96"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0], r.stdout[-2000:]      # the one JSON line and nothing else, also through the launcher
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["n_ranks_seen"] == 2 and line["ranks"]["local_ranks_seen"] == [0, 1]
    assert line["pixels_total"] == 2 * 3 * 64 * 96 * 2
    # the child's failure is the parent's exit code
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-control-flow", "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--batch", "1", "--height", "64", "--width", "64"], env=dict(env, AEJ_REHEARSE_BAD_RANK="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    # --gpus against a launcher's WORLD_SIZE
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-control-flow", "--gpus", "4"],
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def test_a_rank_that_never_verified_fails_the_run():
    """`all_verified` needs an explicit pass from EVERY rank; a rank that skipped or never reached its check (None) fails the
    run unless verification was switched off for the whole run."""
    from adaptive_edge_aware_jpeg_amd.sharding import gather_rank_report
    r = gather_rank_report(None, 0, 1.0, 0, None)
    assert not r["all_verified"] and r["unverified_ranks"] == [0]
    assert gather_rank_report(None, 0, 1.0, 0, None, require_verified=False)["all_verified"]
    assert not gather_rank_report(None, 0, 1.0, 0, False, require_verified=False)["all_verified"]


def test_bench_fails_on_every_rank_when_one_rank_fails_its_oracle_check():
    """A rank whose outputs differ from the oracle's must fail the whole run: rank 1 reports a failed check (rehearsal switch), the
    all_gather carries it to rank 0, the JSON line shows it and BOTH processes exit non-zero."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AEJ_BENCH_BACKEND="gloo",
                   AEJ_REHEARSE_BAD_RANK="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--rehearse-control-flow", "--gpus", "2", "--steps", "2",
                                       "--warmup", "1", "--batch", "2", "--height", "64", "--width", "64"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [3, 3], [o[1][-1500:] for o in outs]
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])
    assert line["ranks"]["verified_ok_by_rank"] == [True, False] and not line["ranks"]["all_verified"]


def test_rank_report_single_process():
    from adaptive_edge_aware_jpeg_amd.sharding import gather_rank_report
    r = gather_rank_report(None, 0, 7.25, 64, True)
    assert r["n_ranks_seen"] == 1 and r["ms_per_step_by_rank"] == [7.25] and r["all_verified"] and r["images_per_step_by_rank"] == [64]
    assert not gather_rank_report(None, 0, 7.25, 2, False)["all_verified"]
