"""SRCNN on MI355X behind the reference's class surface (reference: deep_learning_models/SRCNN_model.py).

Graph (SRCNN_model.py:48-53): conv9x9x96 ReLU -> conv1x1x32 ReLU -> conv5x5x3 linear, SAME, on a
bicubic pre-upscaled image.  Forward, bicubic, patch plumbing and metrics are libsr355 kernels; `fit` (Keras model.fit with
MSE / Adam / EarlyStopping / ReduceLROnPlateau, SRCNN_model.py:55-98) runs through sr355/train.py.
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

    def fit(self, X_train, Y_train, X_val, Y_val, batch_size=16, epochs=50, shuffle=True, verbose=True):
        """Trains the model (SRCNN_model.py:62-98): Adam(learning_rate), loss = mean_squared_error, metrics psnr / ssim,
        EarlyStopping(val_loss, patience 3, restore_best_weights), ReduceLROnPlateau(val_loss, factor 0.5, patience 2, min_lr 1e-7).
        Returns (history, epoch-time record, epoch-memory record) like the reference's (history, callbacks[2], callbacks[3])."""
        if self.model is None:
            raise ValueError("Model has not been set up.")
        if self.compute_dtype not in ("f32", "float32"):
            raise ValueError("training runs in fp32 (the reference's precision); build the model with compute_dtype='f32'")
        from sr355 import train as T
        print("Training on GPU:", torch.cuda.get_device_name(self.ctx.torch_device))
        opt = T.Adam(self.weights, learning_rate=self.learning_rate, epsilon=1e-7)
        weights, history, tcb, mcb = T.fit(self.ctx, self.weights, T.srcnn_loss_and_grads, self._predict_with, opt, X_train, Y_train, X_val, Y_val,
                                           batch_size=batch_size, epochs=epochs, es_patience=3, lr_patience=2, shuffle=shuffle, verbose=verbose)
        self.set_weights(weights)
        self._trained = True
        return history, tcb, mcb

    def _predict_with(self, ctx, weights, x):
        if weights is not getattr(self, "_loaded_for_predict", None):
            self.model.set_weights(weights)
            self._loaded_for_predict = weights
        return self.model.forward(x)

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
        lr, is_np = P.as_device_image(self.ctx, lr_img)
        up = self.ctx.resize(lr[None], int(hr_h), int(hr_w), interpolation)[0]       # any of OpenCV's four resize codes
        sr, metrics = P.patchwise_sr(self.model, up, patch_size, stride, 1, chunk=256)
        return (sr.cpu().numpy() if is_np else sr), metrics

    def save(self, directory, timestamp, fmt="npz"):
        """SRCNN_model.py:249-260 writes SRCNN_<timestamp>.h5; fmt="h5" writes that name in Keras' weight layout (sr355.h5lite),
        the default "npz" this build's own container.  Both load back through setup_model(from_pretrained=True)."""
        if not self._trained:
            raise RuntimeError("Cannot save an untrained model.")
        if not directory:
            raise ValueError("Directory path must be provided.")
        filepath = self._save_weights(directory, f"SRCNN_{timestamp}", fmt)
        print(f"Model saved to {filepath}")
        return filepath
