"""psnr / ssim with tf.image semantics (reference: SRModels/metrics.py:3-7), computed by the HIP
reduction kernels in libsr355 (sr_psnr / sr_ssim).  Inputs [B,H,W,C] (or [H,W,C]); NumPy in ->
NumPy float32 [B] out, device tensor in -> device tensor out."""
import numpy as np
import torch

from sr355 import Context


def _run(name, y_true, y_pred, max_val):
    ctx = Context.get()
    is_np = not isinstance(y_true, torch.Tensor)
    a = ctx.to_device(np.asarray(y_true, dtype=np.float32)) if is_np else y_true.to(ctx.torch_device, torch.float32).contiguous()
    b = ctx.to_device(np.asarray(y_pred, dtype=np.float32)) if not isinstance(y_pred, torch.Tensor) else y_pred.to(ctx.torch_device, torch.float32).contiguous()
    squeeze = a.dim() == 3
    if squeeze:
        a, b = a[None], b[None]
    out = getattr(ctx, name)(a, b, max_val)
    if squeeze:
        out = out[0]
    return out.cpu().numpy() if is_np else out


def psnr(y_true, y_pred):
    return _run("psnr", y_true, y_pred, 1.0)


def ssim(y_true, y_pred):
    return _run("ssim", y_true, y_pred, 1.0)
