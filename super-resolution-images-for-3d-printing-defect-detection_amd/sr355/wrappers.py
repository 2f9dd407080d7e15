"""Common machinery of the reference-shaped model classes in SRModels/: weight files, the trained flag,
Keras-style evaluate() over batches."""
import os

import numpy as np
import torch

from .runtime import Context, Model
from .weights import init_weights, load_npz, save_npz


def load_pretrained(path):
    """Keras `load_model(path)` stand-in (SRCNN_model.py:35, ESRGAN_model.py:143-149, VGG16_model.py:42).  `.npz` is this build's own
    container; `.h5` is the reference's (`model.save`, SRCNN_model.py:249-260): read with h5py where it is installed, otherwise with the
    NumPy reader in sr355.h5lite (contiguous, unfiltered datasets in old-style groups: what Keras / h5py write by default)."""
    if path is None or not os.path.isfile(path):
        raise FileNotFoundError(f"Pretrained model file not found at {path}")
    if path.endswith((".h5", ".hdf5", ".keras.h5")):
        return _load_h5(path)
    return load_npz(path)


def _load_h5(path):
    try:
        import h5py
    except ImportError:
        from .h5lite import load_keras_weights
        return load_keras_weights(path)
    out = {}
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset) and name.split("/")[-1].split(":")[0] in ("kernel", "bias"):
                layer = name.split("/")[-2]
                slot = 0 if "kernel" in name.split("/")[-1] else 1
                out.setdefault(layer, [None, None])[slot] = np.asarray(obj)
        g.visititems(visit)
    return {n: (k, b if b is not None else np.zeros(k.shape[-1], np.float32)) for n, (k, b) in out.items()}


class DeviceModelMixin:
    """Holds a libsr355 Model plus host copies of its weights."""
    _init_scheme = "glorot_uniform"

    def _make(self, kind, compute_dtype="f32", **cfg):
        self.model = Model(kind, compute_dtype=compute_dtype, ctx=Context.get(), **cfg)
        self.ctx = self.model.ctx
        return self.model

    def set_weights(self, weights, trained=True):
        """Load {layer: (kernel, bias)} into the device model (set_weights / load_weights in Keras terms)."""
        self.weights = {n: (np.asarray(k, np.float32), np.asarray(b, np.float32)) for n, (k, b) in weights.items()}
        self.model.set_weights(self.weights)
        self._mark_trained(bool(trained))

    def _random_init(self, seed=1000):
        self.weights = init_weights(self.model.layer_shapes(), scheme=self._init_scheme, seed=seed)
        self.model.set_weights(self.weights)

    def _save_npz(self, path):
        save_npz(path, self.weights)

    def _save_weights(self, directory, stem, fmt="npz"):
        """`model.save(os.path.join(directory, stem + ".h5"))` in the reference (SRCNN_model.py:249-260).  fmt "npz" (default: this
        build's container) or "h5" (Keras' weight layout, written by sr355.h5lite: readable by this package and by Keras' load_weights)."""
        os.makedirs(directory, exist_ok=True)
        if fmt == "h5":
            from .h5lite import save_keras_weights
            path = os.path.join(directory, stem + ".h5")
            save_keras_weights(path, self.weights, model_name=stem)
        elif fmt == "npz":
            path = os.path.join(directory, stem + ".npz")
            save_npz(path, self.weights)
        else:
            raise ValueError(f"fmt must be 'npz' or 'h5', not {fmt!r}")
        return path


def evaluate_sr(ctx, predict, X, Y, batch_size=32):
    """keras Model.evaluate(X, Y) with loss=mse and metrics [psnr, ssim]: per-batch values averaged with
    sample-count weights -> [loss, psnr, ssim] (SRCNN_model.py:100-109, EDSR_model.py:178-187)."""
    n = len(X)
    tot = np.zeros(3, dtype=np.float64)
    for i in range(0, n, batch_size):
        x = ctx.to_device(np.asarray(X[i:i + batch_size], np.float32))
        y = ctx.to_device(np.asarray(Y[i:i + batch_size], np.float32))
        p = predict(x)
        k = x.shape[0]
        tot[0] += float(ctx.mse(y, p).item()) * k
        tot[1] += float(ctx.psnr(y, p).sum().item())
        tot[2] += float(ctx.ssim(y, p).sum().item())
    return list(tot / max(n, 1))
