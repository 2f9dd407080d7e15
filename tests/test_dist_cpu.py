"""The N>1 path on CPU: two gloo ranks shard the tiles and all-reduce the metric sums exactly as bench.py does
over RCCL (the collective is the path's only exchange step)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_units, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import ops as O
    from sr355 import dist as D
    import torch.distributed as dist
    r, w, _ = D.init_from_env(backend="gloo")
    lo, hi = D.shard_range(n_units, r, w)
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (n_units, 24, 24, 3)).astype(np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    sums = torch.zeros(3, dtype=torch.float64)
    if hi > lo:
        sums[0] = float(O.psnr(a[lo:hi], b[lo:hi], dtype=np.float64).sum())
        sums[1] = float(O.ssim(a[lo:hi], b[lo:hi], dtype=np.float64).sum())
        sums[2] = hi - lo
    D.allreduce_metric_sums(sums)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py's max-over-ranks of the elapsed time
    if rank == 0:
        q.put((D.mean_metrics(sums), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_metric_allreduce():
    from oracle import ops as O
    n = 5
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got, tmax = q.get()
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (n, 24, 24, 3)).astype(np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    assert got["n"] == n and tmax == 2.0
    assert abs(got["psnr"] - O.psnr(a, b, dtype=np.float64).mean()) < 1e-9
    assert abs(got["ssim"] - O.ssim(a, b, dtype=np.float64).mean()) < 1e-9
