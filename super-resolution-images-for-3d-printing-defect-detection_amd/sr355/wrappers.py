"""Common machinery of the reference-shaped model classes in SRModels/: weight files, the trained flag,
Keras-style evaluate() over batches."""
import os

import numpy as np
import torch

from .runtime import Context, Model
from .weights import init_weights, load_npz, save_npz


def load_pretrained(path):
    """Keras `load_model(path)` stand-in.  `.npz` (this build's container) always; `.h5` only where h5py exists."""
    if path is None or not os.path.isfile(path):
        raise FileNotFoundError(f"Pretrained model file not found at {path}")
    if path.endswith(".h5"):
        try:
            import h5py  # noqa: F401
        except ImportError as e:
            raise ImportError("reading Keras .h5 checkpoints needs h5py, which is not installed; convert to .npz") from e
        return _load_h5(path)
    return load_npz(path)


def _load_h5(path):
    import h5py
    out = {}
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset) and name.split("/")[-1].split(":")[0] in ("kernel", "bias"):
                layer = name.split("/")[-2]
                slot = 0 if "kernel" in name.split("/")[-1] else 1
                out.setdefault(layer, [None, None])[slot] = np.asarray(obj)
        g.visititems(visit)
    return {n: (k, b) for n, (k, b) in out.items()}


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
