"""Profiling probe: a short ESRGAN trunk (no attention) on one tile's worth of 48x48 patches, bf16."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context, Model
from sr355.weights import init_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 441
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = Context.get(0)
m = Model("esrgan_g", compute_dtype="bf16", scale_factor=4, num_blocks=2, growth_channels=32, use_attention=False, ctx=ctx)
m.set_weights(init_weights(m.layer_shapes(), seed=3000))
x = ctx.to_device(np.random.default_rng(0).uniform(-1, 1, (B, 48, 48, 3)).astype(np.float32))
for _ in range(reps):
    y = m.forward(x)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
