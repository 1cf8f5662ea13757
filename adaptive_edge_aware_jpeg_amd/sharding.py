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
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(pixels_local), float(seconds_local)
    import torch
    px = torch.tensor([float(pixels_local)], dtype=torch.float64, device=device)
    sec = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    dist.all_reduce(px, op=dist.ReduceOp.SUM)
    dist.all_reduce(sec, op=dist.ReduceOp.MAX)
    return int(round(px.item())), float(sec.item())
