"""The four cv2.resize up-scalers of the reference's classical baseline on MI355X (reference:
classic_super_resolution_algorithms/classic_algorithms.py:7-21).  Each is `cv2.resize(lr_img, target_shape, interpolation=...)`:
target_shape is (width, height) as in OpenCV; uint8 input -> uint8 (OpenCV's fixed-point path), float input -> float32, no
clipping.  Bicubic runs the tiled INTER_CUBIC kernel (BASELINE cfg0 and the SRCNN pre-upscale), the others the tap-table
kernel behind sr_resize.  The other four algorithms of that module (back-projection, non-local means, edge-guided, frequency
extrapolation) are CPU image-processing baselines outside the accelerated path."""
import numpy as np
import torch

from sr355 import Context


def _resize(lr_img, target_shape, interpolation):
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
    y = ctx.resize(x, out_h, out_w, interpolation)[0].cpu().numpy()
    return y[:, :, 0] if gray else y


def interpolate_bilinear(lr_img, target_shape):
    """Bilinear upscaling (classic_algorithms.py:7-9)."""
    return _resize(lr_img, target_shape, "INTER_LINEAR")


def interpolate_bicubic(lr_img, target_shape):
    """Bicubic upscaling (classic_algorithms.py:11-13)."""
    return _resize(lr_img, target_shape, "INTER_CUBIC")


def interpolate_area(lr_img, target_shape):
    """Area (resampling) upscaling (classic_algorithms.py:15-17)."""
    return _resize(lr_img, target_shape, "INTER_AREA")


def interpolate_lanczos(lr_img, target_shape):
    """Lanczos-4 upscaling (classic_algorithms.py:19-21)."""
    return _resize(lr_img, target_shape, "INTER_LANCZOS4")


def _out_of_scope(name):
    def fn(*_a, **_k):
        raise NotImplementedError(f"{name}: CPU image-processing baseline of the reference, outside the accelerated hot path (SURVEY.md section 8)")
    fn.__name__ = name
    return fn


back_projection = _out_of_scope("back_projection")
non_local_means = _out_of_scope("non_local_means")
edge_guided_interpolation = _out_of_scope("edge_guided_interpolation")
frequency_extrapolation = _out_of_scope("frequency_extrapolation")
