"""In-kernel timeline of the fused dense-block kernels (csrc/dense_fused.hip): the stamped diagnostic build writes s_memtime at four
points of every (chunk, kx) granule -- before the wait+barrier, after it, after the DMA issue, after the MFMAs -- for waves 0 and 5 of
the first 64 workgroups.  Prints, per granule position inside a step, the median cycles of each phase.

    python tools/probe_chain.py [patches] [mask]      # mask 1: conv4+conv5, 2: conv2+conv3
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

from sr355 import Context, Model
from sr355.weights import init_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1764
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = Context.get(0)
m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=1, growth_channels=32, use_attention=False, ctx=ctx)
m.set_weights(init_weights(m.layer_shapes(), seed=1))
x = torch.randn(B, 48, 48, 3, device="cuda").clamp(-1, 1).to(torch.bfloat16)
ctx.set_fused(mask, 0)
for _ in range(2):
    m.forward(x)
torch.cuda.synchronize()
buf = torch.zeros(64 * 4 * 64 * 4, dtype=torch.int64, device="cuda")
ctx.check(ctx.lib.sr_debug_set_chain_stamp_buffer(ctx.h, buf.data_ptr(), buf.numel() * 8))
m.forward(x)
torch.cuda.synchronize()
ctx.check(ctx.lib.sr_debug_set_chain_stamp_buffer(ctx.h, None, 0))
t = buf.cpu().numpy().reshape(64, 4, 64, 4).astype(np.int64)
ngr = 18 if mask == 1 else 12                    # granules per step of the LAST launch that wrote the buffer (dense3's pair)
valid = t[:, :, :, 0] > 0
print(f"B={B} mask={mask}: stamps of the last fused launch; granules per step = {ngr}")
print("gran   wait+barrier   issue   compute   | total   (median cycles over workgroups; compute waves 0, 5; loader waves 8, 11: phases = barrier wait, issue, counted wait)")
for g in range(min(2 * ngr, 63)):
    row = []
    for wv in (0, 1, 2, 3):
        ok = valid[:, wv, g] & valid[:, wv, g + 1]
        if not ok.any():
            row.append("   -")
            continue
        a = t[ok, wv, g]
        wait = np.median(a[:, 1] - a[:, 0]); iss = np.median(a[:, 2] - a[:, 1]); comp = np.median(a[:, 3] - a[:, 2])
        tot = np.median(t[ok, wv, g + 1, 0] - a[:, 0])
        skew = np.median(a[:, 1] - t[ok, 0, g, 1])           # barrier exit relative to wave 0's
        row.append(f"{wait:6.0f} {iss:5.0f} {comp:6.0f} |{tot:6.0f} {skew:+5.0f}")
    print(f"{g:3d} ({g % ngr:2d})  " + "   ".join(row))
