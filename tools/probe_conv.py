"""Profiling probe: one 3x3 bf16 conv (B patches 48x48) launched a few times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context

cin, cout = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 441
ctx = Context.get(0)
x = torch.randn(B, 48, 48, cin, device="cuda").to(torch.bfloat16)
w = (np.random.default_rng(1).standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
for _ in range(4):
    y = ctx.conv2d(x, w, None, act="relu")
torch.cuda.synchronize()
print("ok")
