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
        self.num_res_blocks, self.res_scaling, self.learning_rate = num_res_blocks, res_scaling, learning_rate
        if weights is not None:
            self.set_weights(weights)
            print(f"Loaded pretrained model from {pretrained_path}")
        else:
            self._random_init(seed=2000)

    def fit(self, X_train, Y_train, X_val, Y_val, batch_size=16, epochs=300, shuffle=True, verbose=True):
        """Train the model (EDSR_model.py:127-176): Adam(lr, eps 1e-8, clipnorm 1.0), loss = mean_squared_error whatever `loss` said
        at setup (reference quirk, :137), EarlyStopping(patience 5, restore_best_weights), ReduceLROnPlateau(0.5, patience 3, 1e-7).
        Returns (history, epoch-time record, epoch-memory record)."""
        if self.model is None:
            raise ValueError("Model is not built yet.")
        if self.compute_dtype not in ("f32", "float32"):
            raise ValueError("training runs in fp32 (the reference's precision); build the model with compute_dtype='f32'")
        from functools import partial
        from sr355 import train as T
        print("Training on GPU:", torch.cuda.get_device_name(self.ctx.torch_device))
        opt = T.Adam(self.weights, learning_rate=self.learning_rate, epsilon=1e-8, clipnorm=1.0)
        lg = partial(T.edsr_loss_and_grads, scale=self.scale_factor, num_res_blocks=self.num_res_blocks, res_scaling=self.res_scaling)
        weights, history, tcb, mcb = T.fit(self.ctx, self.weights, lg, self._predict_with, opt, X_train, Y_train, X_val, Y_val,
                                           batch_size=batch_size, epochs=epochs, es_patience=5, lr_patience=3, shuffle=shuffle, verbose=verbose)
        self.set_weights(weights)
        self.trained = True
        return history, tcb, mcb

    def _predict_with(self, ctx, weights, x):
        if weights is not getattr(self, "_loaded_for_predict", None):
            self.model.set_weights(weights)
            self._loaded_for_predict = weights
        return self.model.forward(x)

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

    def save(self, directory, timestamp, fmt="npz"):
        """EDSR_model.py:317-328; fmt="h5" writes Keras' weight layout (sr355.h5lite), "npz" this build's container."""
        if not self.trained:
            raise RuntimeError("Cannot save an untrained model.")
        if not directory:
            raise ValueError("Directory path must be provided.")
        path = self._save_weights(directory, f"EDSR_x{self.scale_factor}_{timestamp}", fmt)
        print(f"Model saved to {path}")
        return path
