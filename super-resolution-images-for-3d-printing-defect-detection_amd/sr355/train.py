"""Keras `model.fit` for the SRCNN and EDSR graphs on MI355X (reference: SRCNN_model.py:55-98, EDSR_model.py:127-176).

What the reference delegates to Keras is restated here as host orchestration over the C ABI: forward convs (the MFMA kernels, fp32 as
the reference trains), loss = mean_squared_error, the backward pass layer by layer -- input gradients are the SAME forward kernels on
180-degree-rotated, channel-swapped weights, weight / bias gradients the fp32-MFMA wgrad kernel, activations / clip / depth_to_space
their element-wise or permutation inverses (csrc/train_ops.hip) -- and Adam exactly as Keras applies it (optimizer_v2 dense update:
lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); var -= lr_t m / (sqrt(v) + eps); EDSR: per-variable clip-by-norm 1.0, eps 1e-8), plus the two
callbacks that change the result (EarlyStopping with restore_best_weights, ReduceLROnPlateau) and the two that only record
(epoch time, device memory).  Optimiser state and master weights live on the host (28 931 / 1.37 M parameters); every step re-packs the
updated kernels for the device through the single-op entry points.
"""
import time

import numpy as np
import torch

from . import _lib as L


def _rot(w):
    """Kernel of the input-gradient conv: rotate by 180 degrees, swap the channel axes (HWIO -> HW O I)."""
    return np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))


class DevWeights:
    """A parameter dict on the device for one optimiser step: one flat upload, per-array views; conv / dgrad run on them through
    sr_conv2d_dev (packed into MFMA fragment order by a device kernel, no host re-pack per call)."""

    def __init__(self, ctx, w):
        self.ctx = ctx
        arrs = [(n, s, np.asarray(a, np.float32)) for n, pair in w.items() for s, a in enumerate(pair)]
        flat = ctx.to_device(np.concatenate([a.ravel() for _, _, a in arrs]))
        self.t, o = {}, 0
        for n, s, a in arrs:
            self.t[(n, s)] = flat[o:o + a.size].view(tuple(a.shape))
            o += a.size

    def conv(self, x, name, **kw):
        k = self.t[(name, 0)]
        return self.ctx.conv2d_dev(x, k, self.t[(name, 1)], k.shape[3], **kw)

    def dgrad(self, dy, name):
        k = self.t[(name, 0)]
        return self.ctx.conv2d_dev(dy, k, None, k.shape[2], rot=True)


class Adam:
    """keras.optimizers.Adam (TF 2.10 optimizer_v2), dense update, fp32 state."""

    def __init__(self, weights, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = float(learning_rate), beta_1, beta_2, epsilon, clipnorm
        self.t = 0
        self.m = {n: [np.zeros_like(k), np.zeros_like(b)] for n, (k, b) in weights.items()}
        self.v = {n: [np.zeros_like(k), np.zeros_like(b)] for n, (k, b) in weights.items()}

    def apply(self, weights, grads):
        self.t += 1
        lr_t = np.float32(self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))
        out = {}
        for n, (k, b) in weights.items():
            if n not in grads:                                     # a variable the loss does not reach: Keras skips None gradients
                out[n] = (k, b)
                continue
            new = []
            for slot, (var, g) in enumerate(((k, grads[n][0]), (b, grads[n][1]))):
                g = np.asarray(g, np.float32)
                if self.clipnorm is not None:                      # tf.clip_by_norm per variable
                    nrm = float(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
                    if nrm > self.clipnorm:
                        g = g * np.float32(self.clipnorm / nrm)
                m = self.m[n][slot] = np.float32(self.b1) * self.m[n][slot] + np.float32(1.0 - self.b1) * g
                v = self.v[n][slot] = np.float32(self.b2) * self.v[n][slot] + np.float32(1.0 - self.b2) * g * g
                new.append((var - lr_t * m / (np.sqrt(v) + np.float32(self.eps))).astype(np.float32))
            out[n] = (new[0], new[1])
        return out


class DeviceAdam:
    """The same optimiser with its state on the device: parameters, both moments and the gradient are flat fp32 buckets and one fused kernel
    (sr_adam) updates them in place -- no parameter leaves the GPU between steps.  Bit for bit the update of `Adam.apply` on the same numbers
    (the kernel rounds every operation separately, in NumPy's order)."""

    def __init__(self, ctx, flat_weights, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        import torch
        self.ctx, self.lr, self.b1, self.b2, self.eps = ctx, float(learning_rate), beta_1, beta_2, epsilon
        self.t = 0
        self.m, self.v = torch.zeros_like(flat_weights), torch.zeros_like(flat_weights)

    def apply(self, flat_weights, flat_grads, grad_scale=1.0):
        self.t += 1
        lr_t = np.float32(self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))
        self.ctx.adam_step(flat_weights, flat_grads, self.m, self.v, lr_t, self.b1, self.b2, self.eps, grad_scale)


# ---------------------------------------------------------------------------------------------------------------- graphs
def srcnn_loss_and_grads(ctx, w, x, t):
    """SRCNN_model.py:48-53 + mean_squared_error.  x, t device fp32 [B,H,W,3].  -> (prediction, loss tensor [1], {layer: (dw, db)})."""
    dw = DevWeights(ctx, w)
    a1 = dw.conv(x, "conv2d", act="relu")
    a2 = dw.conv(a1, "conv2d_1", act="relu")
    y = dw.conv(a2, "conv2d_2")
    loss = ctx.mse(t, y)
    g = {}
    dy = ctx.eltwise(L.ELT_AXPBY, y, t, 2.0 / y.numel(), -2.0 / y.numel())
    g["conv2d_2"] = ctx.conv2d_wgrad(a2, dy, 5)
    d2 = ctx.eltwise(L.ELT_RELU_BWD, dw.dgrad(dy, "conv2d_2"), a2)
    g["conv2d_1"] = ctx.conv2d_wgrad(a1, d2, 1)
    d1 = ctx.eltwise(L.ELT_RELU_BWD, dw.dgrad(d2, "conv2d_1"), a1)
    g["conv2d"] = ctx.conv2d_wgrad(x, d1, 9)
    return y, loss, g


def edsr_loss_and_grads(ctx, w, x, t, scale=2, num_res_blocks=16, res_scaling=0.1):
    """EDSR_model.py:55-125 + mean_squared_error (the compile step ignores its loss argument, :137)."""
    names = ["conv2d"] + [f"conv2d_{i}" for i in range(1, 2 * num_res_blocks + 5)]
    it = iter(names)
    n_head = next(it)
    dw = DevWeights(ctx, w)
    h0 = dw.conv(x, n_head)
    cur, blocks = h0, []
    for _ in range(num_res_blocks):
        na, nb = next(it), next(it)
        tt = dw.conv(cur, na, act="relu")
        nxt = dw.conv(tt, nb, alpha=res_scaling, skip1=cur, beta1=1.0)
        blocks.append((na, nb, cur, tt))
        cur = nxt
    n_body = next(it)
    body = dw.conv(cur, n_body, skip1=h0, beta1=1.0)
    ups, up = [], body
    for r in ([scale] if scale in (2, 3) else [2, 2]):
        nu = next(it)
        v = dw.conv(up, nu, d2s=r)
        ups.append((nu, up, r))
        up = v
    n_out = next(it)
    pre = dw.conv(up, n_out)
    y = ctx.eltwise(L.ELT_CLIP01, pre)
    loss = ctx.mse(t, y)
    g = {}
    dy = ctx.eltwise(L.ELT_AXPBY, y, t, 2.0 / y.numel(), -2.0 / y.numel())
    dpre = ctx.eltwise(L.ELT_CLIP01_BWD, dy, pre)
    g[n_out] = ctx.conv2d_wgrad(up, dpre, 3)
    d = dw.dgrad(dpre, n_out)
    for nu, uin, r in reversed(ups):
        dv = ctx.space_to_depth(d, r)
        g[nu] = ctx.conv2d_wgrad(uin, dv, 3)
        d = dw.dgrad(dv, nu)
    d_h0 = d                                                  # global skip: body = conv(cur) + h0
    g[n_body] = ctx.conv2d_wgrad(cur, d, 3)
    d = dw.dgrad(d, n_body)
    for na, nb, cin, tt in reversed(blocks):
        du = ctx.eltwise(L.ELT_AXPBY, d, None, res_scaling, 0.0)
        g[nb] = ctx.conv2d_wgrad(tt, du, 3)
        dt = ctx.eltwise(L.ELT_RELU_BWD, dw.dgrad(du, nb), tt)
        g[na] = ctx.conv2d_wgrad(cin, dt, 3)
        d = ctx.eltwise(L.ELT_AXPBY, d, dw.dgrad(dt, na), 1.0, 1.0)
    d = ctx.eltwise(L.ELT_AXPBY, d, d_h0, 1.0, 1.0)
    g[n_head] = ctx.conv2d_wgrad(x, d, 3)
    return y, loss, g


# ---------------------------------------------------------------------------------------------------------------- fit
class History:
    def __init__(self):
        self.history = {}
        self.epoch = []


class EpochTimeRecord:
    """EpochTimeCallback's record (callbacks.py:21-42)."""
    def __init__(self):
        self.epoch_times_sec = []

    def mean_time_value(self):
        return float(np.mean(self.epoch_times_sec))


class EpochMemoryRecord:
    """EpochMemoryCallback's record (callbacks.py:44-): device memory held by libsr355, MB."""
    def __init__(self):
        self.gpu_mean_current_mb = []
        self.gpu_peak_mb = []

    def as_dict(self):
        return {"gpu_mean_current_mb": float(np.mean(self.gpu_mean_current_mb)), "gpu_peak_mb": float(np.max(self.gpu_peak_mb))}


class EpochTimeTracker(EpochTimeRecord):
    """Manual tracker of the custom ESRGAN loop (callbacks.py:104-121)."""
    def __init__(self):
        super().__init__()
        self._t0 = None

    def begin_epoch(self):
        self._t0 = time.perf_counter()

    def end_epoch(self):
        if self._t0 is not None:
            self.epoch_times_sec.append(time.perf_counter() - self._t0)
            self._t0 = None


class EpochMemoryTracker(EpochMemoryRecord):
    """callbacks.py:123-176 with libsr355's allocator counters in place of tf.config.experimental.get_memory_info."""
    def __init__(self, ctx):
        super().__init__()
        self.ctx, self._begin = ctx, None

    def begin_epoch(self):
        self._begin = self.ctx.mem_info()

    def end_epoch(self):
        end, begin = self.ctx.mem_info(), self._begin or self.ctx.mem_info()
        self.gpu_mean_current_mb.append((begin["current"] + end["current"]) / 2.0 / (1024.0 * 1024.0))
        self.gpu_peak_mb.append(max(begin["peak"], end["peak"]) / (1024.0 * 1024.0))
        self._begin = None


def fit(ctx, weights, loss_and_grads, predict, optimizer, X_train, Y_train, X_val, Y_val, batch_size=16, epochs=50, es_patience=3,
        lr_patience=2, lr_factor=0.5, min_lr=1e-7, shuffle=True, seed=42, verbose=True):
    """model.fit(X, Y, batch_size, epochs, validation_data, callbacks=[EarlyStopping(val_loss, patience, restore_best_weights=True),
    ReduceLROnPlateau(val_loss, factor, patience, min_lr), EpochTimeCallback, EpochMemoryCallback]).
    weights: {layer: (kernel, bias)} host fp32 (updated copy returned).  -> (weights, History, EpochTimeRecord, EpochMemoryRecord)."""
    X_train, Y_train = np.asarray(X_train, np.float32), np.asarray(Y_train, np.float32)
    X_val, Y_val = np.asarray(X_val, np.float32), np.asarray(Y_val, np.float32)
    hist, tcb, mcb = History(), EpochTimeRecord(), EpochMemoryRecord()
    for k in ("loss", "psnr", "ssim", "val_loss", "val_psnr", "val_ssim", "lr"):
        hist.history[k] = []
    rng = np.random.default_rng(seed)
    best, best_w, es_wait, lr_wait, lr_best = np.inf, None, 0, 0, np.inf
    n = len(X_train)
    for ep in range(epochs):
        mem0 = ctx.mem_info()
        t0 = time.perf_counter()
        order = rng.permutation(n) if shuffle else np.arange(n)
        tot = np.zeros(3, np.float64)
        for i in range(0, n, batch_size):
            idx = order[i:i + batch_size]
            x, t = ctx.to_device(X_train[idx]), ctx.to_device(Y_train[idx])
            y, loss, grads = loss_and_grads(ctx, weights, x, t)
            k = len(idx)
            tot += [float(loss.item()) * k, float(ctx.psnr(t, y).sum().item()), float(ctx.ssim(t, y).sum().item())]
            host_g = {name: (dw.cpu().numpy(), db.cpu().numpy()) for name, (dw, db) in grads.items()}
            weights = optimizer.apply(weights, host_g)
        vt = np.zeros(3, np.float64)
        for i in range(0, len(X_val), batch_size):
            x, t = ctx.to_device(X_val[i:i + batch_size]), ctx.to_device(Y_val[i:i + batch_size])
            y = predict(ctx, weights, x)
            vt += [float(ctx.mse(t, y).item()) * len(x), float(ctx.psnr(t, y).sum().item()), float(ctx.ssim(t, y).sum().item())]
        torch.cuda.synchronize(ctx.torch_device)
        tr, va = tot / max(n, 1), vt / max(len(X_val), 1)
        for key, val in zip(("loss", "psnr", "ssim", "val_loss", "val_psnr", "val_ssim"), list(tr) + list(va)):
            hist.history[key].append(float(val))
        hist.history["lr"].append(optimizer.lr)
        hist.epoch.append(ep)
        tcb.epoch_times_sec.append(time.perf_counter() - t0)
        mem1 = ctx.mem_info()
        mcb.gpu_mean_current_mb.append((mem0["current"] + mem1["current"]) / 2.0 / (1024.0 * 1024.0))
        mcb.gpu_peak_mb.append(max(mem0["peak"], mem1["peak"]) / (1024.0 * 1024.0))
        if verbose:
            print(f"Epoch {ep + 1}/{epochs} - loss: {tr[0]:.4f} - psnr: {tr[1]:.4f} - ssim: {tr[2]:.4f} - val_loss: {va[0]:.4f} - "
                  f"val_psnr: {va[1]:.4f} - val_ssim: {va[2]:.4f} - lr: {optimizer.lr:.2e}")
        # ReduceLROnPlateau(monitor="val_loss", mode min, min_delta 1e-4, cooldown 0)
        if va[0] < lr_best - 1e-4:
            lr_best, lr_wait = va[0], 0
        else:
            lr_wait += 1
            if lr_wait >= lr_patience and optimizer.lr > min_lr:
                optimizer.lr = max(optimizer.lr * lr_factor, min_lr)
                lr_wait = 0
                if verbose:
                    print(f"Epoch {ep + 1}: ReduceLROnPlateau reducing learning rate to {optimizer.lr}.")
        # EarlyStopping(monitor="val_loss", min_delta 0, restore_best_weights=True)
        if best_w is None:       # Keras snapshots the weights at the first epoch even when val_loss is NaN / never improves
            best_w = {k2: (a.copy(), b.copy()) for k2, (a, b) in weights.items()}
        if va[0] < best:
            best, es_wait = va[0], 0
            best_w = {k2: (a.copy(), b.copy()) for k2, (a, b) in weights.items()}
        else:
            es_wait += 1
            if es_wait >= es_patience:
                weights = best_w
                if verbose:
                    print(f"Epoch {ep + 1}: early stopping; restoring model weights from the end of the best epoch.")
                break
    return weights, hist, tcb, mcb


# ---------------------------------------------------------------------------------------------------------------- classifier head
def head_forward(g, w, training=False, dropout_rate=0.0, rng=None):
    """GAP features [N,512] -> Dropout -> Dense256 ReLU -> Dropout -> Dense softmax (VGG16_model.py:84-97), host fp32/fp64.
    -> (probabilities, cache for head_backward)."""
    k1, b1 = w["dense"]
    k2, b2 = w["predictions"]
    keep = 1.0 - dropout_rate
    m0 = m1 = None
    x0 = g
    if training and dropout_rate > 0:
        m0 = (rng.random(g.shape) < keep) / keep
        x0 = g * m0
    z1 = x0 @ k1 + b1
    a1 = np.maximum(z1, 0)
    x1 = a1
    if training and dropout_rate > 0:
        m1 = (rng.random(a1.shape) < keep) / keep
        x1 = a1 * m1
    z2 = x1 @ k2 + b2
    z2 = z2 - z2.max(axis=1, keepdims=True)
    e = np.exp(z2)
    p = e / e.sum(axis=1, keepdims=True)
    return p, (x0, z1, x1, m1)


def sparse_cce(p, y, eps=1e-7):
    """keras sparse_categorical_crossentropy on probabilities (clipped to [eps, 1-eps]) -> (mean loss, accuracy)."""
    pc = np.clip(p[np.arange(len(y)), y], eps, 1.0 - eps)
    return float(np.mean(-np.log(pc))), float(np.mean(np.argmax(p, axis=1) == y))


def head_backward(p, y, cache, w, l2_reg=0.0):
    """Gradients of mean sparse-CCE (+ l2_reg * sum(dense kernel^2), VGG16_model.py:90-92) w.r.t. the two Dense layers."""
    x0, z1, x1, m1 = cache
    n = len(y)
    dz2 = p.copy()
    dz2[np.arange(n), y] -= 1.0
    dz2 /= n
    k2 = w["predictions"][0]
    g = {"predictions": (x1.T @ dz2, dz2.sum(axis=0))}
    dx1 = dz2 @ k2.T
    da1 = dx1 * m1 if m1 is not None else dx1
    dz1 = da1 * (z1 > 0)
    g["dense"] = (x0.T @ dz1 + 2.0 * l2_reg * w["dense"][0], dz1.sum(axis=0))
    return g


def fit_head(features, w, X_train_batches, y_train, X_val, y_val, learning_rate=1e-3, batch_size=32, epochs=50, dropout_rate=0.2,
             l2_reg=0.0, seed=42, verbose=True):
    """model.fit of the frozen-base classifier (VGG16_model.py:111-157): Adam, sparse categorical cross-entropy, accuracy,
    EarlyStopping(val_loss, patience 3, restore_best_weights) and ReduceLROnPlateau(val_loss, 0.5, patience 2, min_lr 1e-7).
    features(images) -> [n,512] GAP features (the device runs the frozen conv base); X_train_batches(epoch) yields (images, labels)
    batches of one epoch (shuffled / augmented by the caller).  -> (head weights, History)."""
    head = {n: (np.asarray(w[n][0], np.float32), np.asarray(w[n][1], np.float32)) for n in ("dense", "predictions")}
    opt = Adam(head, learning_rate, epsilon=1e-7)
    rng = np.random.default_rng(seed)
    hist = History()
    for k in ("loss", "accuracy", "val_loss", "val_accuracy", "lr"):
        hist.history[k] = []
    g_val = features(X_val)
    best, best_w, es_wait, lr_wait, lr_best = np.inf, None, 0, 0, np.inf
    for ep in range(epochs):
        tot, n_seen = np.zeros(2, np.float64), 0
        for xb, yb in X_train_batches(ep):
            yb = np.asarray(yb, np.int64)
            p, cache = head_forward(features(xb).astype(np.float32), head, True, dropout_rate, rng)
            loss, acc = sparse_cce(p, yb)
            if l2_reg > 0:
                loss += l2_reg * float(np.sum(head["dense"][0].astype(np.float64) ** 2))
            tot += [loss * len(yb), acc * len(yb)]
            n_seen += len(yb)
            grads = head_backward(p, yb, cache, head, l2_reg)
            head = opt.apply(head, {n: (a.astype(np.float32), b.astype(np.float32)) for n, (a, b) in grads.items()})
        pv, _ = head_forward(g_val.astype(np.float32), head)
        vl, va = sparse_cce(pv, np.asarray(y_val, np.int64))
        if l2_reg > 0:
            vl += l2_reg * float(np.sum(head["dense"][0].astype(np.float64) ** 2))
        tr = tot / max(n_seen, 1)
        for key, val in zip(("loss", "accuracy", "val_loss", "val_accuracy", "lr"), (tr[0], tr[1], vl, va, opt.lr)):
            hist.history[key].append(float(val))
        hist.epoch.append(ep)
        if verbose:
            print(f"Epoch {ep + 1}/{epochs} - loss: {tr[0]:.4f} - accuracy: {tr[1]:.4f} - val_loss: {vl:.4f} - val_accuracy: {va:.4f} - lr: {opt.lr:.2e}")
        if vl < lr_best - 1e-4:
            lr_best, lr_wait = vl, 0
        else:
            lr_wait += 1
            if lr_wait >= 2 and opt.lr > 1e-7:
                opt.lr = max(opt.lr * 0.5, 1e-7)
                lr_wait = 0
                if verbose:
                    print(f"Epoch {ep + 1}: ReduceLROnPlateau reducing learning rate to {opt.lr}.")
        if best_w is None:       # as above: the first epoch's weights are the fallback of restore_best_weights
            best_w = {n: (a.copy(), b.copy()) for n, (a, b) in head.items()}
        if vl < best:
            best, es_wait = vl, 0
            best_w = {n: (a.copy(), b.copy()) for n, (a, b) in head.items()}
        else:
            es_wait += 1
            if es_wait >= 3:
                head = best_w
                if verbose:
                    print(f"Epoch {ep + 1}: early stopping; restoring model weights from the end of the best epoch.")
                break
    return head, hist
