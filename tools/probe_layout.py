"""Layout probe: the same 3x3 bf16 conv (1764 patches 48x48, Cout=32) on COMPACT NHWC inputs (pixel stride = Cin) for
several Cin, kernel time from the library's HIP-event profile.  Compared with the in-model per-launch times (inputs
embedded in 192-channel dense-block buffers) it shows what the 64-B-of-384-B access pattern costs."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1764
ctx = Context.get(0)
res = {}
CFGS = ((32, 32), (64, 32), (96, 32), (128, 32), (160, 32), (192, 64))
if len(sys.argv) > 2:
    CFGS = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]]
for cin, cout in CFGS:
    x = torch.randn(B, 48, 48, cin, device="cuda").to(torch.bfloat16)
    w = (np.random.default_rng(1).standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    for _ in range(2):
        ctx.conv2d(x, w, None, act="lrelu")
    torch.cuda.synchronize()
    ctx.profile_begin()
    for _ in range(6):
        ctx.conv2d(x, w, None, act="lrelu")
    torch.cuda.synchronize()
    recs = ctx.profile_end()
    r = [k for k in recs if k["kernel"].startswith("conv_rows")][0]
    ms = r["total_ms"] / r["launches"]
    res[f"{cin}->{cout}"] = {"us": round(ms * 1e3, 1), "TFLOP/s": round(r["flops"] / r["launches"] / ms / 1e9, 1),
                            "GB/s": round(r["bytes"] / r["launches"] / ms / 1e6, 1)}
    del x
print(json.dumps(res, indent=1))
