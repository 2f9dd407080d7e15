"""Oracle for the training rows (CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Keras `model.fit(loss="mean_squared_error", optimizer=Adam)` for SRCNN / EDSR (SRCNN_model.py:55-98, EDSR_model.py:127-176)
restated with torch autograd in fp64: the graphs are rebuilt from torch primitives (not from oracle.ops, so that the gradients are
an independent derivation), the optimiser is the TF 2.10 optimizer_v2 Adam dense update.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _conv(x, k, b, act=None):
    """x NCHW fp64, k HWIO numpy/tensor -> SAME stride-1 conv (odd kernels: symmetric padding)."""
    kt = k.permute(3, 2, 0, 1)
    y = F.conv2d(x, kt, b, padding=k.shape[0] // 2)
    return F.relu(y) if act == "relu" else y


def _d2s(x, r):
    """tf.nn.depth_to_space, DCR order, on NCHW: channel (i*r + j)*C + c -> pixel (h*r + i, w*r + j), channel c."""
    B, Cr, H, W = x.shape
    C = Cr // (r * r)
    return x.reshape(B, r, r, C, H, W).permute(0, 3, 4, 1, 5, 2).reshape(B, C, H * r, W * r)


def _params(w):
    return {n: (torch.tensor(np.asarray(k, np.float64), requires_grad=True), torch.tensor(np.asarray(b, np.float64), requires_grad=True))
            for n, (k, b) in w.items()}


def srcnn_forward_t(p, x):
    x = _conv(x, *p["conv2d"], act="relu")
    x = _conv(x, *p["conv2d_1"], act="relu")
    return _conv(x, *p["conv2d_2"])


def edsr_forward_t(p, x, scale=2, num_res_blocks=16, res_scaling=0.1):
    names = iter(["conv2d"] + [f"conv2d_{i}" for i in range(1, 2 * num_res_blocks + 5)])
    x = _conv(x, *p[next(names)])
    head = x
    for _ in range(num_res_blocks):
        sc = x
        x = _conv(x, *p[next(names)], act="relu")
        x = _conv(x, *p[next(names)]) * res_scaling + sc
    x = _conv(x, *p[next(names)]) + head
    for r in ([scale] if scale in (2, 3) else [2, 2]):
        x = _d2s(_conv(x, *p[next(names)]), r)
    return torch.clamp(_conv(x, *p[next(names)]), 0.0, 1.0)


def loss_and_grads(forward, w, x, t, **kw):
    """x, t NHWC numpy.  -> (prediction NHWC, mse, {layer: (dk, db)}) in fp64."""
    p = _params(w)
    xt = torch.tensor(np.asarray(x, np.float64)).permute(0, 3, 1, 2)
    tt = torch.tensor(np.asarray(t, np.float64)).permute(0, 3, 1, 2)
    y = forward(p, xt, **kw)
    loss = torch.mean((y - tt) ** 2)
    loss.backward()
    g = {n: (k.grad.numpy(), b.grad.numpy()) for n, (k, b) in p.items()}
    return y.detach().permute(0, 2, 3, 1).numpy(), float(loss.item()), g


class AdamRef:
    """TF 2.10 optimizer_v2 Adam, dense, fp64: m, v moments; lr_t = lr sqrt(1-b2^t)/(1-b1^t); var -= lr_t m / (sqrt(v) + eps);
    clipnorm = per-variable tf.clip_by_norm before the moments."""

    def __init__(self, w, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm, self.t = lr, beta_1, beta_2, epsilon, clipnorm, 0
        self.m = {n: [np.zeros(k.shape), np.zeros(b.shape)] for n, (k, b) in w.items()}
        self.v = {n: [np.zeros(k.shape), np.zeros(b.shape)] for n, (k, b) in w.items()}

    def apply(self, w, g):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        out = {}
        for n, (k, b) in w.items():
            new = []
            for s, (var, gr) in enumerate(((k, g[n][0]), (b, g[n][1]))):
                gr = np.asarray(gr, np.float64)
                if self.clipnorm is not None:
                    nrm = np.sqrt(np.sum(gr ** 2))
                    if nrm > self.clipnorm:
                        gr = gr * (self.clipnorm / nrm)
                self.m[n][s] = self.b1 * self.m[n][s] + (1 - self.b1) * gr
                self.v[n][s] = self.b2 * self.v[n][s] + (1 - self.b2) * gr * gr
                new.append(np.asarray(var, np.float64) - lr_t * self.m[n][s] / (np.sqrt(self.v[n][s]) + self.eps))
            out[n] = tuple(new)
        return out


def train_steps(forward, w, batches, lr, epsilon=1e-7, clipnorm=None, **kw):
    """Apply one Adam step per (x, t) batch in order.  -> (weights fp64, [loss per step])."""
    opt = AdamRef(w, lr, epsilon=epsilon, clipnorm=clipnorm)
    w = {n: (np.asarray(k, np.float64), np.asarray(b, np.float64)) for n, (k, b) in w.items()}
    losses = []
    for x, t in batches:
        _, loss, g = loss_and_grads(forward, w, x, t, **kw)
        losses.append(loss)
        w = opt.apply(w, g)
    return w, losses
