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


def partition_tiles(n_tiles, rank, world, scaling="strong"):
    """"strong" = ONE batch of n_tiles split B/g over the ranks in contiguous slices (16 tiles -> 16/8/4/2 per GPU
    at 1/2/4/8 GPUs); "weak" = every rank owns n_tiles of its own.  -> (tile ids of this rank, tiles of the whole job)."""
    if scaling == "weak":
        return list(range(n_tiles)), n_tiles * world
    if scaling != "strong":
        raise ValueError(f"scaling must be 'strong' or 'weak', not {scaling!r}")
    lo, hi = shard_range(n_tiles, rank, world)
    return list(range(lo, hi)), n_tiles


def _allreduce(t, op):
    """In-place all-reduce of a small tensor.  RCCL ("nccl") reduces device tensors directly; under gloo (CPU tests, and the
    N>1 rehearsal of bench.py on a one-GPU box) a device tensor is staged through the host."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return t
    if t.is_cuda and dist.get_backend() != "nccl":
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def allreduce_metric_sums(sums):
    """sums: float64 tensor [sum_psnr, sum_ssim, count] on this rank's device -> global sums (in place).  The path's only
    exchange step."""
    return _allreduce(sums, dist.ReduceOp.SUM)


def allreduce_max(t):
    """max over ranks, in place (bench.py: the slowest rank's wall time)."""
    return _allreduce(t, dist.ReduceOp.MAX)


def allreduce_mean_grads(grads, device=None):
    """Data-parallel training (ESRGANTrainer's `allreduce` hook, sr355.train.fit): {layer: (dk, db)} host arrays of this rank's
    batch shard -> the mean over ranks, same structure.  All layers travel as ONE flat fp32 bucket (a full ESRGAN generator is
    ~67 MB: one ring all-reduce at xGMI link rate, not hundreds of latency-bound small ones); under RCCL the bucket is reduced on
    `device`, under gloo on the host.  Equal shard sizes make the mean of the per-rank batch-mean gradients the full-batch gradient."""
    import numpy as np
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return grads
    names = sorted(grads)
    parts = [np.asarray(a, np.float32).ravel() for n in names for a in grads[n]]
    flat = torch.from_numpy(np.concatenate(parts))
    if dist.get_backend() == "nccl":
        flat = flat.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat = (flat / dist.get_world_size()).cpu().numpy()
    out, o = {}, 0
    for n in names:
        pair = []
        for a in grads[n]:
            a = np.asarray(a)
            pair.append(flat[o:o + a.size].reshape(a.shape).copy())
            o += a.size
        out[n] = tuple(pair)
    return out


def allreduce_mean_flat(flat):
    """Data-parallel training, the generator's gradients: ONE flat fp32 device bucket (67.7 MB at the default depth) -> its mean over the
    ranks.  Under RCCL the bucket is reduced where it lies -- it never touches the host (round 2 staged it through NumPy both ways); under
    gloo (CPU tests, the one-GPU rehearsal) a device tensor goes through the host."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return flat
    _allreduce(flat, dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    return flat


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def mean_metrics(sums):
    s = sums.detach().cpu().tolist()
    n = max(s[2], 1.0)
    return {"psnr": s[0] / n, "ssim": s[1] / n, "n": int(s[2])}
