#!/usr/bin/env python3
"""The PRODUCT trainer under data parallelism, as ranks of torch.distributed.run (SR355_ONE_DEVICE=1 + gloo on a one-GPU box, RCCL on a node):
ESRGAN.enable_data_parallel() installs both all-reduce hooks (the generator's flat device bucket, the discriminator's dict route), every rank
feeds _train_step its own shard of each batch, and after the steps the replicas must be BIT-identical (max over ranks == min over ranks of every
parameter, Adam moment and spectral-norm vector).  Rank 0 prints one JSON line with checksums; `--world-one` runs the same batches in one
process (the reference a 2-rank run is compared with).  tests/test_00_multirank_gpu.py drives it."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

STEPS, BATCH = 2, 4


def main():
    from sr355 import dist as D
    torch.cuda.set_device(0 if os.environ.get("SR355_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, _ = D.init_from_env(backend=os.environ.get("SR355_DIST_BACKEND", "gloo" if os.environ.get("SR355_ONE_DEVICE") else "nccl"))
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    m = ESRGAN(compute_dtype="f32")
    m.setup_model(scale_factor=2, growth_channels=8, num_rrdb_blocks=1)
    if world > 1:
        m.enable_data_parallel()
    rng = np.random.default_rng(9)
    losses = []
    for _ in range(STEPS):
        lr = rng.uniform(-1, 1, (BATCH, 12, 12, 3)).astype(np.float32)
        hr = rng.uniform(-1, 1, (BATCH, 24, 24, 3)).astype(np.float32)
        lo, hi = D.shard_range(BATCH, rank, world)
        losses.append(m._train_step(lr[lo:hi], hr[lo:hi]))
    tr = m._trainer
    state = [tr._gflat, tr.g_opt.m, tr.g_opt.v]
    state += [torch.from_numpy(np.concatenate([np.asarray(a, np.float32).ravel() for n in sorted(tr.dw) for a in tr.dw[n]])).to(tr._gflat.device)]
    state += [torch.from_numpy(np.concatenate([tr.u[n].ravel() for n in sorted(tr.u)])).to(tr._gflat.device)]
    identical = True
    for t in state:
        hi_, lo_ = t.clone(), (-t).clone()
        D.allreduce_max(hi_)
        D.allreduce_max(lo_)
        identical = identical and bool(torch.equal(hi_, -lo_))
    flat = torch.cat([t.reshape(-1).float() for t in state]).cpu().numpy()
    if rank == 0:
        print(json.dumps({"world": world, "replicas_bit_identical": identical, "sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
                          "g_weights_head": [float(x) for x in flat[:4]], "g_abs_sum": float(np.abs(tr._gflat.cpu().numpy()).sum()),
                          "d_abs_sum": float(np.abs(state[3].cpu().numpy()).sum()), "losses_rank0": [{k: float(v) for k, v in l.items()} for l in losses]}), flush=True)
        np.save(os.environ["DP_REHEARSAL_OUT"], flat) if os.environ.get("DP_REHEARSAL_OUT") else None
    if world > 1:
        D.shutdown()


if __name__ == "__main__":
    main()
