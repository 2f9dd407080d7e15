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
    D.allreduce_max(t)                                # bench.py's max-over-ranks of the elapsed time
    mine, total = D.partition_tiles(16, r, w, "strong")
    cnt = torch.tensor([float(len(mine)), float(sum(mine))], dtype=torch.float64)
    D.allreduce_metric_sums(cnt)                      # every tile of the batch owned by exactly one rank
    assert total == 16 and cnt.tolist() == [16.0, float(sum(range(16)))]
    D.barrier()
    if rank == 0:
        q.put((D.mean_metrics(sums), float(t.item())))
    D.shutdown()


def test_partition_tiles():
    """16 tiles -> 16/8/4/2 per GPU at 1/2/4/8 GPUs; weak = 16 each."""
    from sr355 import dist as D
    for world in (1, 2, 4, 8):
        owned = [D.partition_tiles(16, r, world, "strong") for r in range(world)]
        assert all(t == 16 for _, t in owned) and all(len(m) == 16 // world for m, _ in owned)
        assert sorted(i for m, _ in owned for i in m) == list(range(16))
    assert D.partition_tiles(16, 3, 8, "weak") == (list(range(16)), 128)
    assert [len(D.partition_tiles(16, r, 3, "strong")[0]) for r in range(3)] == [6, 5, 5]
    import pytest
    with pytest.raises(ValueError):
        D.partition_tiles(16, 0, 2, "diagonal")


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
