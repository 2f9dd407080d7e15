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


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import models as M, train as OT
    from sr355 import dist as D
    from sr355.weights import init_weights
    r, w, _ = D.init_from_env(backend="gloo")
    wts = init_weights(M.srcnn_layers(), seed=11)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (4, 12, 12, 3)).astype(np.float32)
    t = rng.uniform(0, 1, (4, 12, 12, 3)).astype(np.float32)
    lo, hi = D.shard_range(4, r, w)
    opt = OT.AdamRef(wts, 1e-3)
    w64 = {n: (np.asarray(k, np.float64), np.asarray(b, np.float64)) for n, (k, b) in wts.items()}
    for _ in range(2):                                   # two steps: the ranks must stay bit-identical to each other
        _, _, g = OT.loss_and_grads(OT.srcnn_forward_t, w64, x[lo:hi], t[lo:hi])
        g = D.allreduce_mean_grads(g)
        w64 = opt.apply(w64, g)
    # the generator's route (ESRGANTrainer.allreduce_flat): one flat fp32 bucket averaged in place
    bucket = torch.full((1000,), float(rank + 1), dtype=torch.float32)
    bucket[::7] = float(10 * (rank + 1))
    out = D.allreduce_mean_flat(bucket)
    assert out is bucket and float(bucket[1]) == 1.5 and float(bucket[0]) == 15.0 and float(bucket.sum()) == 1.5 * 857 + 15.0 * 143
    flat = torch.from_numpy(np.concatenate([np.asarray(a, np.float64).ravel() for n in sorted(w64) for a in w64[n]]))
    other = flat.clone()
    D.allreduce_max(other)
    assert torch.equal(other, flat)                      # max over ranks == own value on every rank: replicas agree exactly
    if rank == 0:
        q.put(flat.numpy())
    D.shutdown()


def test_two_rank_data_parallel_gradients():
    """Training shards the batch over the ranks and averages the gradients in one flat bucket (sr355.dist.allreduce_mean_grads):
    two gloo ranks with half a batch each take the same Adam steps as one process with the whole batch."""
    from oracle import models as M, train as OT
    from sr355.weights import init_weights
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import time
    t0 = time.time()
    while q.empty() and time.time() - t0 < 180 and all(p.exitcode in (None, 0) for p in procs):
        time.sleep(0.2)
    assert not q.empty(), [p.exitcode for p in procs]
    got = q.get()                                        # read before join: the weights do not fit the pipe's buffer
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    wts = init_weights(M.srcnn_layers(), seed=11)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (4, 12, 12, 3)).astype(np.float32)
    t = rng.uniform(0, 1, (4, 12, 12, 3)).astype(np.float32)
    ref, _ = OT.train_steps(OT.srcnn_forward_t, wts, [(x, t), (x, t)], 1e-3)
    flat = np.concatenate([np.asarray(a, np.float64).ravel() for n in sorted(ref) for a in ref[n]])
    # the bucket travels in fp32: the averaged gradient carries fp32 rounding, Adam's first steps are ~lr * sign(g)
    assert np.abs(got - flat).max() <= 2e-5
    assert np.abs(got - np.concatenate([np.asarray(a, np.float64).ravel() for n in sorted(wts) for a in wts[n]])).max() > 5e-4   # it moved


def test_partitions_at_world_8():
    """the shards of BASELINE configs[2] (16 tiles), configs[4] (a stream of frames) and configs[3] (a batch) at the node's 8 ranks:
    contiguous, disjoint, complete, sizes within one of each other -- for counts that divide, that do not, and that are smaller than the world."""
    from sr355 import dist as D
    for n in (16, 3600, 14400, 30, 17, 8, 5, 1, 0):
        ranges = [D.shard_range(n, r, 8) for r in range(8)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n and sizes == sorted(sizes, reverse=True)
    owned = [D.partition_tiles(16, r, 8, "strong")[0] for r in range(8)]
    assert owned == [[2 * r, 2 * r + 1] for r in range(8)]
    # the cfg4 stream: stream_sr_classify takes frames[lo:hi] of shard_range(len(frames), rank, world) -- 30 frames at 8 ranks: 4,4,4,4,4,4,3,3
    assert [hi - lo for lo, hi in (D.shard_range(30, r, 8) for r in range(8))] == [4, 4, 4, 4, 4, 4, 3, 3]
    # the dense-block kernels' row partition at a rank's share (csrc/dense_fused.hip): a range of R stream rows costs ceil((R + 3) / 8) eight-row steps
    # (one warm-up row at either end, the second layer one row behind), so 882 patches x 49 rows on 256 workgroups = 169 rows each take 22 steps where
    # 21.5 would do -- the 4 % of profiles/r0*_tiles_sweep.jsonl at 2 tiles; 7056 patches: 1351 rows, 170 steps, 99.3 % of the slots
    steps = lambda rows: -(-(rows + 3) // 8)
    for patches, want in ((882, 22), (7056, 170), (1764, 43), (3528, 85)):
        rows = -(-patches * 49 // 256)
        assert steps(rows) == want, (patches, rows, steps(rows))
