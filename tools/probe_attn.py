"""Profiling probe: SelfAttention core on B images of N=HxW tokens (bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np, torch
from sr355 import Context
from sr355.weights import init_weights
from oracle.models import self_attention_layers
B, H = int(sys.argv[1]), int(sys.argv[2])
ctx = Context.get(0)
w = init_weights(self_attention_layers("sa"), seed=1)
x = torch.randn(B, H, H, 64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    y = ctx.self_attention(x, *w["sa_f"], *w["sa_g"], *w["sa_h"], *w["sa_v"])
torch.cuda.synchronize()
print("ok")
