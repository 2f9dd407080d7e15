"""Per-kernel HIP-event times of SRCNN (cfg1) on 4 x 1024x1024 images."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import torch
from sr355 import Context, Model
from sr355.weights import init_weights
ctx = Context.get(0)
for dt in ("f32", "bf16"):
    m = Model("srcnn", compute_dtype=dt, ctx=ctx); m.set_weights(init_weights(m.layer_shapes(), seed=1000))
    x = torch.rand(4, 1024, 1024, 3, device="cuda")
    m.forward(x); torch.cuda.synchronize()
    ctx.profile_begin()
    for _ in range(3):
        m.forward(x)
    for r in ctx.profile_end():
        print(dt, r["kernel"], round(r["total_ms"] / r["launches"], 3), "ms", round(r["flops"] / r["total_ms"] / 1e9, 1), "TFLOP/s (algorithmic)")
