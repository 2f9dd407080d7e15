"""Secondary row of SURVEY.md 8d: the literal whole-tile forward G([B,512,512,3]) of the bench generator (ESRGAN x4, NB=23, G=32,
both SelfAttention layers at N = 262144 and 1048576 tokens) -- the mode the reference never leaves patch mode for, because it
materialises the N x N score matrix.  One MI355X, bf16, device-resident input."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context, Model
from sr355.synth import make_pairs
from sr355.weights import init_weights

ctx = Context.get(0)
out = {}
for att in (True, False):
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=4, num_blocks=23, growth_channels=32, use_attention=att, ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), seed=3000))
    lr, _ = make_pairs(2, 512, 512, 4, seed=44)
    x = ctx.to_device(lr * 2 - 1)
    ctx.profile_begin()
    y = m.forward(x); torch.cuda.synchronize()
    ctx.profile_end()
    ts = []
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = m.forward(x); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = min(ts)
    ctx.profile_begin(); m.forward(x); torch.cuda.synchronize(); prof = ctx.profile_end()
    prof.sort(key=lambda r: -r["total_ms"])
    out["attention" if att else "no_attention"] = {"tiles": 2, "ms": ms, "MPix/s": 2 * 2048 * 2048 / 1e6 / ms * 1e3,
                                                   "top": [(r["kernel"], round(r["total_ms"], 2), round(r["flops"] / r["total_ms"] / 1e9, 1)) for r in prof[:3]]}
    assert torch.isfinite(y.float()).all()
    del m
print(json.dumps(out, indent=1))
