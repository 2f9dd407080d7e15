"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  The hot path shards by tile/image with no data-path collective; the only exchange
is the all-reduce of the metric sums [sum_psnr, sum_ssim, n]."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n independent units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_metric_sums(sums):
    """sums: float64 tensor [sum_psnr, sum_ssim, count] on this rank's device -> global sums (in place)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


def mean_metrics(sums):
    s = sums.detach().cpu().tolist()
    n = max(s[2], 1.0)
    return {"psnr": s[0] / n, "ssim": s[1] / n, "n": int(s[2])}
