"""Throughput of the other SURVEY.md section-8a rows on one MI355X (secondary numbers for DESIGN.md; bench.py carries the
headline metric).  Device-resident inputs, torch events around the calls, median of a few repetitions."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context, Model
from sr355.weights import init_weights

ctx = Context.get(0)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


out = {}
# a1 bicubic x4: 32 x 256x256 -> 1024x1024 fp32 (cfg1 pre-upscale)
x = torch.rand(32, 256, 256, 3, device="cuda")
ms = timeit(lambda: ctx.bicubic(x, 1024, 1024))
out["a1_bicubic_f32_32x256to1024"] = {"ms": ms, "GB/s": (x.numel() + 32 * 1024 * 1024 * 3) * 4 / ms / 1e6, "MPix/s": 32 * 1.048576 / ms * 1e3}
# a2 psnr / ssim on 16 x 2048x2048x3
a = torch.rand(16, 2048, 2048, 3, device="cuda"); b = (a + 0.01 * torch.randn_like(a)).clamp(0, 1)
ms = timeit(lambda: ctx.psnr(a, b)); out["a2_psnr_16x2048"] = {"ms": ms, "GB/s": 2 * a.numel() * 4 / ms / 1e6}
ms = timeit(lambda: ctx.ssim(a, b)); out["a2_ssim_16x2048"] = {"ms": ms, "GB/s": 2 * a.numel() * 4 / ms / 1e6}
del a, b
# a3 SRCNN fp32 (cfg1): 1024x1024 HR-resolution images, 4 per forward (the 96-channel fp32 intermediate is 1.6 GB per 4 images)
for dt in ("f32", "bf16"):
    m = Model("srcnn", compute_dtype=dt, ctx=ctx); m.set_weights(init_weights(m.layer_shapes(), seed=1000))
    xx = torch.rand(4, 1024, 1024, 3, device="cuda")
    ms = timeit(lambda: m.forward(xx), reps=3, warm=1)
    out[f"a3_srcnn_{dt}_4x1024x1024"] = {"ms": ms, "MPix/s": 4 * 1.048576 / ms * 1e3, "TFLOP/s": 4 * 1048576 * 57600 / ms / 1e9}
    del m
# a4 EDSR x4 (B=16, F=64) on 441 LR patches 48x48
for dt in ("f32", "bf16"):
    m = Model("edsr", compute_dtype=dt, scale_factor=4, num_blocks=16, num_filters=64, res_scaling=0.1, ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), scheme="he_normal", seed=2000))
    xx = torch.rand(441, 48, 48, 3, device="cuda")
    ms = timeit(lambda: m.forward(xx), reps=3, warm=1)
    out[f"a4_edsr_x4_{dt}_441x48x48"] = {"ms": ms, "out MPix/s": 441 * 192 * 192 / 1e6 / ms * 1e3, "TFLOP/s": 441 * 2304 * 2 * 1983168 / ms / 1e9}
    del m
# a8 VGG16 classifier on 96x96 patches
for dt in ("f32", "bf16"):
    m = Model("vgg16", compute_dtype=dt, num_classes=2, ctx=ctx); m.set_weights(init_weights(m.layer_shapes(), scheme="he_normal", seed=4000))
    xx = torch.rand(1024, 96, 96, 3, device="cuda")
    ms = timeit(lambda: m.forward(xx), reps=3, warm=1)
    out[f"a8_vgg16_{dt}_1024x96x96"] = {"ms": ms, "patches/s": 1024 / ms * 1e3, "TFLOP/s": 1024 * 2 * 2.819e9 / ms / 1e9}
    del m
print(json.dumps(out, indent=1))
