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

from sr355 import Context, Model
from sr355 import dist as D
from sr355.gan_train import ESRGANTrainer
from sr355.weights import condition_attention, init_weights

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
local = 0 if os.environ.get("SR355_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))   # SR355_ONE_DEVICE: N>1 rehearsal on a 1-GPU box
torch.cuda.set_device(local)
rank, world, _ = D.init_from_env(backend=os.environ.get("SR355_DIST_BACKEND"))                 # "gloo" only for that rehearsal
ctx = Context.get(local)
g = Model("esrgan_g", compute_dtype="f32", scale_factor=4, num_blocks=23, growth_channels=32, use_attention=True, ctx=ctx)
d = Model("esrgan_d", compute_dtype="f32", ctx=ctx)
v = Model("vgg19_features", compute_dtype="f32", ctx=ctx)
gw = condition_attention(init_weights(g.layer_shapes(), seed=3000))
# glorot-initialised RRDBs have gain ~1.2 per block: 23 of them in fp32 stay finite but the losses would be astronomically large;
# scale the residual branches' last convs so that the step's numbers are ordinary (timing does not depend on the values)
gw = {n: ((k * 0.1, b * 0.1) if n.endswith("_conv5") else (k, b)) for n, (k, b) in gw.items()}
dw = init_weights(d.layer_shapes(), seed=5000)
vw = init_weights(v.layer_shapes(), scheme="he_normal", seed=6000)
vw = {n: (k * 0.05 if n == "block1_conv1" else k, b) for n, (k, b) in vw.items()}
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


tr = ESRGANTrainer(ctx, gw, dw, vw, 4, 23, attention=True, allreduce=allreduce if world > 1 else None, allreduce_flat=allreduce_flat if world > 1 else None)
rng = np.random.default_rng(42 + 3 + rank)
lr = rng.uniform(-1, 1, (batch, 24, 24, 3)).astype(np.float32)
hr = rng.uniform(-1, 1, (batch, 96, 96, 3)).astype(np.float32)
out = tr.train_step(lr, hr)                                      # warm-up
torch.cuda.synchronize()
D.barrier()
allreduce_ms.clear()
t0 = time.perf_counter()
for _ in range(steps):
    out = tr.train_step(lr, hr)
torch.cuda.synchronize()
t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
D.allreduce_max(t)
if rank == 0:
    wall = float(t.item())
    n_g = sum(int(np.prod(k.shape)) + int(np.prod(b.shape)) for k, b in gw.values())
    n_d = sum(int(np.prod(k.shape)) + int(np.prod(b.shape)) for k, b in dw.values())
    print(json.dumps({"row": "cfg3 ESRGAN _train_step", "n_gpus": world, "batch_per_gpu": batch, "lr_patch": 24, "scale": 4, "num_rrdb": 23,
                      "growth_channels": 32, "dtype": "f32", "steps": steps, "ms_per_step": 1e3 * wall / steps,
                      "patches_per_s": batch * world * steps / wall, "generator_params": n_g, "discriminator_params": n_d,
                      "gradient_bucket_mb": 4e-6 * (n_g + n_d), "allreduce_ms_per_step": (sum(allreduce_ms) / steps) if allreduce_ms else None,
                      "losses_finite": bool(all(np.isfinite(float(x)) for x in out.values())), "losses": {k: float(x) for k, x in out.items()}}))
D.shutdown()
