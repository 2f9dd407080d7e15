"""SRCNN on MI355X behind the reference's class surface (reference: deep_learning_models/SRCNN_model.py).

Graph (SRCNN_model.py:48-53): conv9x9x96 ReLU -> conv1x1x32 ReLU -> conv5x5x3 linear, SAME, on a
bicubic pre-upscaled image.  Forward, bicubic, patch plumbing and metrics are libsr355 kernels;
`fit` (MSE/Adam training) is a later row of SURVEY.md section 8f.
"""
import os

import numpy as np
import torch

from sr355 import pipeline as P
from sr355.wrappers import DeviceModelMixin, evaluate_sr, load_pretrained

INTER_CUBIC = 2   # cv2.INTER_CUBIC


class SRCNNModel(DeviceModelMixin):
    def __init__(self, compute_dtype="f32"):
        self.model = None
        self._trained = False
        self.compute_dtype = compute_dtype

    def _mark_trained(self, v):
        self._trained = v

    def setup_model(self, input_shape=None, learning_rate=1e-4, from_pretrained=False, pretrained_path=None):
        if from_pretrained:
            weights = load_pretrained(pretrained_path)
            self._make("srcnn", self.compute_dtype, channels=3)
            self.set_weights(weights)
            print(f"Loaded pretrained model from {pretrained_path}")
        else:
            if input_shape is None:
                raise ValueError("input_shape must be provided when not using a pretrained model.")
            self._make("srcnn", self.compute_dtype, channels=int(input_shape[-1]))
            self._random_init(seed=1000)
        self.learning_rate = learning_rate

    def fit(self, *args, **kwargs):
        raise NotImplementedError("SRCNN training is outside this round's hot path (SURVEY.md 8f row 4)")

    def evaluate(self, X_test, Y_test):
        if not self._trained:
            raise RuntimeError("Model has not been trained.")
        results = evaluate_sr(self.ctx, self.model.forward, X_test, Y_test)
        print(f"Loss: {results[0]:.4f}, PSNR: {results[1]:.2f} dB, SSIM: {results[2]:.4f}")
        return results

    def super_resolve_image(self, lr_img, hr_h, hr_w, patch_size=33, stride=14, interpolation=INTER_CUBIC):
        """Bicubic upscale to (hr_h, hr_w) (no clip, SRCNN_model.py:191), then patch-wise SRCNN with
        overlap averaging.  Returns (float32 RGB [hr_h,hr_w,3] in [0,1], inference_metrics)."""
        if not self._trained:
            raise RuntimeError("Model has not been trained.")
        if lr_img is None or not isinstance(lr_img, (np.ndarray, torch.Tensor)):
            raise ValueError("lr_img must be a numpy array (RGB).")
        if interpolation != INTER_CUBIC:
            raise NotImplementedError("only cv2.INTER_CUBIC is on the accelerated path")
        lr, is_np = P.as_device_image(self.ctx, lr_img)
        up = self.ctx.bicubic(lr[None], int(hr_h), int(hr_w))[0]
        sr, metrics = P.patchwise_sr(self.model, up, patch_size, stride, 1, chunk=256)
        return (sr.cpu().numpy() if is_np else sr), metrics

    def save(self, directory, timestamp):
        if not self._trained:
            raise RuntimeError("Cannot save an untrained model.")
        if not directory:
            raise ValueError("Directory path must be provided.")
        os.makedirs(directory, exist_ok=True)
        filepath = os.path.join(directory, f"SRCNN_{timestamp}.npz")
        self._save_npz(filepath)
        print(f"Model saved to {filepath}")
        return filepath
