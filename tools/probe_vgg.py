"""Per-layer time of the VGG16 classifier's convs (B patches of P x P, bf16): which layers pay for tile quantisation (a 6 x 6 image in a
16 x 16 output tile uses 14 % of the MFMAs the tile issues).  python tools/probe_vgg.py [B] [P]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 96
ctx = Context.get(0)
CFG = [(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)]
rows, cin, hw, tot = [], 3, P, 0.0
for blk, (n, c) in enumerate(CFG):
    for k in range(n):
        x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
        w = (np.random.default_rng(1).standard_normal((3, 3, cin, c)) / np.sqrt(9 * cin)).astype(np.float32)
        for _ in range(2):
            ctx.conv2d(x, w, None, act="relu")
        torch.cuda.synchronize()
        ctx.profile_begin()
        for _ in range(5):
            ctx.conv2d(x, w, None, act="relu")
        torch.cuda.synchronize()
        r = [q for q in ctx.profile_end() if q["kernel"].startswith("conv")][0]
        ms = r["total_ms"] / r["launches"]
        tot += ms
        rows.append({"layer": f"block{blk + 1}_conv{k + 1}", "hw": hw, "cin": cin, "cout": c, "kernel": r["kernel"], "ms": round(ms, 3),
                     "TFLOP/s": round(r["flops"] / r["launches"] / ms / 1e9, 1)})
        cin = c
        del x
    hw //= 2
print(json.dumps({"B": B, "P": P, "conv_ms_total": round(tot, 2), "layers": rows}, indent=1))
