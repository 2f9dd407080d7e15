"""BASELINE configs[3] row: ESRGAN._train_step (ESRGAN_model.py:475-533) at the reference's full training configuration --
x4, NB=23, G=32, both SelfAttention layers, a batch of 16 LR patches 24x24 -> 96x96 per GPU, fp32 -- on sr355.gan_train.ESRGANTrainer.
Prints one JSON line: ms per step, patches/s, the finite-loss check, and (N>1) the time of the flat-bucket gradient all-reduce.

    python tools/bench_train.py [steps=3] [batch=16]
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_train.py     # data parallel: every rank its own batch, gradients averaged

Round 3: the generator's parameters, Adam moments and gradient bucket stay on the device (one fused optimiser kernel, RCCL reduces the
bucket in place); the fp32 convs tile small batches 8 x 16; the weight-gradient kernel requests eight pixel pairs ahead of its MFMAs.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context
from sr355 import dist as D
from sr355.bench_rows import cfg3_train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
local = 0 if os.environ.get("SR355_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))   # SR355_ONE_DEVICE: N>1 rehearsal on a 1-GPU box
torch.cuda.set_device(local)
rank, world, _ = D.init_from_env(backend=os.environ.get("SR355_DIST_BACKEND"))                 # "gloo" only for that rehearsal
ctx = Context.get(local)
allreduce_ms = []


def allreduce(grads):
    t0 = time.perf_counter()
    out = D.allreduce_mean_grads(grads, device=ctx.torch_device)
    allreduce_ms.append(1e3 * (time.perf_counter() - t0))
    return out


def allreduce_flat(flat):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = D.allreduce_mean_flat(flat)
    torch.cuda.synchronize()
    allreduce_ms.append(1e3 * (time.perf_counter() - t0))
    return out


D.barrier()
row = cfg3_train_step(ctx, steps, batch, allreduce if world > 1 else None, allreduce_flat if world > 1 else None, seed_offset=rank)
t = torch.tensor([row.pop("wall_s")], dtype=torch.float64, device="cuda")
D.allreduce_max(t)
if rank == 0:
    wall = float(t.item())
    row.update({"n_gpus": world, "ms_per_step": 1e3 * wall / steps, "patches_per_s": batch * world * steps / wall,
                # the warm-up step's calls (RCCL communicator set-up among them) are reported on their own, not folded into the per-step figure
                "allreduce_ms_per_step": (sum(allreduce_ms[len(allreduce_ms) // (steps + 1):]) / steps) if allreduce_ms else None,
                "allreduce_ms_warmup_step": sum(allreduce_ms[:len(allreduce_ms) // (steps + 1)]) if allreduce_ms else None})
    print(json.dumps(row))
D.shutdown()
