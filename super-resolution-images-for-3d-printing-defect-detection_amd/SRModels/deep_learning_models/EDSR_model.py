"""EDSR on MI355X behind the reference's class surface (reference: deep_learning_models/EDSR_model.py).

Graph (EDSR_model.py:96-125): head conv, B x (conv-ReLU-conv * res_scaling + skip), body conv + global
skip, sub-pixel tail (depth_to_space in TF "DCR" order fused into the conv store), RGB conv, clip[0,1].
"""
import os

import numpy as np
import torch

from sr355 import pipeline as P
from sr355.wrappers import DeviceModelMixin, evaluate_sr, load_pretrained


class EDSR(DeviceModelMixin):
    _init_scheme = "he_normal"   # kernel_initializer="he_normal" (EDSR_model.py:61)

    def __init__(self, compute_dtype="f32"):
        self.model = None
        self.scale_factor = None
        self.trained = False
        self.compute_dtype = compute_dtype

    def _mark_trained(self, v):
        self.trained = v

    def setup_model(self, scale_factor=2, channels=3, num_res_blocks=16, num_filters=64, res_scaling=0.1, learning_rate=1e-4,
                    loss="mean_absolute_error", from_pretrained=False, pretrained_path=None):
        self.scale_factor = scale_factor
        weights = load_pretrained(pretrained_path) if from_pretrained else None
        if weights is not None:   # recover the depth from the checkpoint: 2B+3 convs + 1 (x2,x3) or 2 (x4) up-convs
            n_up = 2 if scale_factor == 4 else 1
            num_res_blocks = (len(weights) - 3 - n_up) // 2
            num_filters = int(weights["conv2d"][0].shape[-1])
        self._make("edsr", self.compute_dtype, scale_factor=scale_factor, channels=channels, num_blocks=num_res_blocks,
                   num_filters=num_filters, res_scaling=res_scaling)
        if weights is not None:
            self.set_weights(weights)
            print(f"Loaded pretrained model from {pretrained_path}")
        else:
            self._random_init(seed=2000)

    def fit(self, *args, **kwargs):
        raise NotImplementedError("EDSR training is outside this round's hot path (SURVEY.md 8f row 4)")

    def evaluate(self, X_test, Y_test):
        if not self.trained:
            raise RuntimeError("Model has not been trained.")
        results = evaluate_sr(self.ctx, self.model.forward, X_test, Y_test)
        print(f"Loss: {results[0]:.4f}, PSNR: {results[1]:.2f} dB, SSIM: {results[2]:.4f}")
        return results

    def super_resolve_image(self, lr_img, patch_size_lr=48, stride=24):
        if not self.trained:
            raise RuntimeError("Model has not been trained.")
        if self.scale_factor is None:
            raise ValueError("scale_factor is not set. Call setup_model first.")
        lr, is_np = P.as_device_image(self.ctx, lr_img)
        sr, metrics = P.patchwise_sr(self.model, lr, patch_size_lr, stride, self.scale_factor, chunk=256)
        return (sr.cpu().numpy() if is_np else sr), metrics

    def save(self, directory, timestamp):
        if not self.trained:
            raise RuntimeError("Cannot save an untrained model.")
        if not directory:
            raise ValueError("Directory path must be provided.")
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, f"EDSR_x{self.scale_factor}_{timestamp}.npz")
        self._save_npz(path)
        print(f"Model saved to {path}")
        return path
