"""Classical up-scalers.  Only bicubic sits on the accelerated path (BASELINE cfg0 and the SRCNN
pre-upscale); it runs the OpenCV-INTER_CUBIC-compatible HIP kernel (sr_bicubic).
Reference: classic_super_resolution_algorithms/classic_algorithms.py:11-13."""
import numpy as np
import torch

from sr355 import Context


def interpolate_bicubic(lr_img, target_shape):
    """cv2.resize(lr_img, target_shape, INTER_CUBIC): target_shape is (width, height) as in OpenCV.
    uint8 input -> uint8 (fixed-point path), float input -> float32, no clipping."""
    ctx = Context.get()
    out_w, out_h = int(target_shape[0]), int(target_shape[1])
    a = np.asarray(lr_img)
    gray = a.ndim == 2
    if gray:
        a = a[:, :, None]
    if a.dtype == np.uint8:
        x = ctx.to_device(a[None], torch.uint8)
    else:
        x = ctx.to_device(a[None].astype(np.float32, copy=False))
    y = ctx.bicubic(x, out_h, out_w)[0].cpu().numpy()
    return y[:, :, 0] if gray else y


def _out_of_scope(name):
    def fn(*_a, **_k):
        raise NotImplementedError(f"{name}: CPU image-processing baseline of the reference, outside the accelerated hot path (SURVEY.md section 8)")
    fn.__name__ = name
    return fn


interpolate_bilinear = _out_of_scope("interpolate_bilinear")
interpolate_area = _out_of_scope("interpolate_area")
interpolate_lanczos = _out_of_scope("interpolate_lanczos")
