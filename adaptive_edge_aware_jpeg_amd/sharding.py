"""Multi-GPU sharding of the encode path: images are independent units (no cross-image state anywhere in
Jpeg.compress, src/jpeg/jpeg.py:240-272), so a batch is cut into contiguous slices by image index, one process
per GPU, with NO data-path collective.  The only communication is the reduction of a few throughput counters
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
from typing import Tuple


def shard_bounds(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [start, stop) of the batch owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(n_images, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def aggregate_throughput(dist, pixels_local: int, seconds_local: float, device=None) -> Tuple[int, float]:
    """-> (total pixels over all ranks, max seconds over ranks).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized():
        return int(pixels_local), float(seconds_local)
    import torch
    px = torch.tensor([float(pixels_local)], dtype=torch.float64, device=device)
    sec = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    dist.all_reduce(px, op=dist.ReduceOp.SUM)
    dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    return int(round(px.item())), float(sec.item())


def gather_rank_report(dist, local_rank: int, ms_per_step: float, images: int, verified_ok, device=None, require_verified: bool = True) -> dict:
    """First-contact evidence for the multi-GPU run (all_gather of four numbers per rank, after the timed region): which local ranks
    the collective really saw, every rank's own ms per step, how many images it encodes per step (its shard) and whether its
    post-timing check against the CPU oracle passed (None = the rank did not verify).  `all_verified` needs ok == True on EVERY rank
    -- a rank that never reached its check counts as a failure -- unless the run was asked not to verify (`require_verified=False`:
    then only an explicit False fails).  Every rank gets the same dict."""
    ok = -1.0 if verified_ok is None else (1.0 if verified_ok else 0.0)
    if dist is None or not dist.is_initialized():
        rows = [[float(local_rank), float(ms_per_step), float(images), ok]]
        world = 1
    else:
        import torch
        world = dist.get_world_size()
        mine = torch.tensor([float(local_rank), float(ms_per_step), float(images), ok], dtype=torch.float64, device=device)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        rows = [g.cpu().tolist() for g in got]
    return {"world_size": world, "n_ranks_seen": len(rows), "local_ranks_seen": [int(r[0]) for r in rows],
            "ms_per_step_by_rank": [round(r[1], 3) for r in rows], "images_per_step_by_rank": [int(r[2]) for r in rows],
            "verified_ok_by_rank": [None if r[3] < 0 else bool(r[3]) for r in rows],
            "unverified_ranks": [i for i, r in enumerate(rows) if r[3] < 0],
            "all_verified": all((r[3] == 1.0) if require_verified else (r[3] != 0.0) for r in rows)}
