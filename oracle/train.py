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
            if n not in g:
                out[n] = (k, b)
                continue
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


# =====================================================================================================================
# ESRGAN._train_step (ESRGAN_model.py:475-533) in torch autograd, fp64
# =====================================================================================================================
def _act(x, act):
    if act == "relu":
        return F.relu(x)
    if act == "lrelu":
        return F.leaky_relu(x, 0.2)
    if act == "tanh":
        return torch.tanh(x)
    return x


def _masked_act(y, act, mask):
    """The activation with its branch taken from `mask` (bool, y > 0 as ANOTHER evaluation of the same graph saw it) instead of from y's own sign:
    two correct forward passes that differ in their last bits disagree about the branch at the few elements whose pre-activation is within rounding
    error of zero, and each such element moves a gradient by a whole unit of slope; with the branches pinned the comparison is about arithmetic."""
    m = mask.to(y.dtype)
    return y * m if act == "relu" else y * (m + 0.2 * (1.0 - m))


def _conv_s(x, k, b, act=None, stride=1, mask=None):
    """Keras Conv2D SAME with TF's asymmetric padding for stride 2; x NCHW."""
    kt = k.permute(3, 2, 0, 1)
    kh = k.shape[0]
    pads = []
    for n in (x.shape[3], x.shape[2]):                    # F.pad order: W first
        out = -(-n // stride)
        tot = max((out - 1) * stride + kh - n, 0)
        pads += [tot // 2, tot - tot // 2]
    y = F.conv2d(F.pad(x, pads), kt, b, stride=stride)
    return _masked_act(y, act, mask) if (mask is not None and act in ("relu", "lrelu")) else _act(y, act)


def _sa_t(p, x, name):
    """SelfAttention.call (ESRGAN_model.py:48-70) on NCHW."""
    B, C, H, W = x.shape
    f = _conv_s(x, *p[name + "_f"]).reshape(B, -1, H * W)          # [B,8,N]
    g = _conv_s(x, *p[name + "_g"]).reshape(B, -1, H * W)
    h = _conv_s(x, *p[name + "_h"]).reshape(B, -1, H * W)
    s = torch.matmul(g.transpose(1, 2), f)                        # [B,N,N] = g f^T
    beta = torch.softmax(s, dim=-1)
    o = torch.matmul(beta, h.transpose(1, 2)).transpose(1, 2).reshape(B, -1, H, W)
    return x + _conv_s(o, *p[name + "_v"])


def generator_forward_t(p, x, scale, num_rrdb, attention=True, masks=None):
    """masks: {layer: bool NCHW tensor} -- activation branches pinned to another evaluation's (see _masked_act); layers not in it use their own."""
    mk = (lambda n: masks.get(n)) if masks else (lambda n: None)
    x = _conv_s(x, *p["initial_conv"])
    trunk = x
    for b in range(num_rrdb):
        r_in = x
        for d in (1, 2, 3):
            n = f"rrdb_{b}_dense{d}"
            feats = [x]
            for k in range(1, 5):
                feats.append(_conv_s(torch.cat(feats, dim=1), *p[f"{n}_conv{k}"], act="relu", mask=mk(f"{n}_conv{k}")))
            x = x + 0.2 * _conv_s(torch.cat(feats, dim=1), *p[f"{n}_conv5"])
        x = r_in + 0.2 * x
    x = trunk + _conv_s(x, *p["trunk_conv"])
    if attention:
        x = _sa_t(p, x, "self_attention_trunk")
    i, s = 0, scale
    while s > 1:
        x = _d2s(_conv_s(x, *p[f"upsample_{i}_conv"]), 2)
        x = _masked_act(x, "lrelu", mk(f"upsample_{i}_conv")) if mk(f"upsample_{i}_conv") is not None else F.leaky_relu(x, 0.2)
        if i == 0 and attention:
            x = _sa_t(p, x, "self_attention_upsample_0")
        s >>= 1
        i += 1
    x = _conv_s(x, *p["final_conv1"], act="relu", mask=mk("final_conv1"))
    return _conv_s(x, *p["final_conv2"], act="tanh")


DISC_NAMES = [f"disc_conv{i}" for i in range(1, 7)] + ["disc_dense1", "disc_output"]


def _sn_inplace(w, u):
    """tfa SpectralNormalization on every wrapper (training=True): kernel.assign(kernel / sigma), u.assign(u)."""
    from . import ops
    w, u = dict(w), dict(u)
    for n in DISC_NAMES:
        k, nu = ops.spectral_normalize(np.asarray(w[n][0], np.float64), u[n])
        w[n], u[n] = (k, w[n][1]), nu
    return w, u


def discriminator_forward_t(p, x, masks=None):
    for i, st in enumerate([1, 2, 1, 2, 1, 2]):
        x = _conv_s(x, *p[f"disc_conv{i + 1}"], act="lrelu", stride=st, mask=(masks or {}).get(f"disc_conv{i + 1}"))
    g = x.mean(dim=(2, 3))
    g = F.leaky_relu(g @ p["disc_dense1"][0] + p["disc_dense1"][1], 0.2)
    return torch.sigmoid(g @ p["disc_output"][0] + p["disc_output"][1])


def vgg19_features_t(p, x):
    """x NCHW in [-1,1] -> block5_conv4 features (caffe preprocessing inside)."""
    x = (x + 1.0) * 127.5
    x = x.flip(1) - torch.tensor([103.939, 116.779, 123.68], dtype=x.dtype).reshape(1, 3, 1, 1)
    for blk, n in ((1, 2), (2, 2), (3, 4), (4, 4), (5, 4)):
        for k in range(1, n + 1):
            x = _conv_s(x, *p[f"block{blk}_conv{k}"], act="relu")
            if (blk, k) == (5, 4):
                return x
        x = F.max_pool2d(x, 2)
    return x


def _bce(t, p, eps=1e-7):
    pc = torch.clamp(p, eps, 1.0 - eps)
    return torch.mean(-(t * torch.log(pc + eps) + (1.0 - t) * torch.log(1.0 - pc + eps)))


def _grads(p):
    return {n: (k.grad.numpy().copy(), b.grad.numpy().copy()) for n, (k, b) in p.items() if k.grad is not None}


def esrgan_train_step_ref(gw, dw, u, vw, lr_img, hr_img, scale, num_rrdb, attention=True, g_lr=1e-4, d_lr=1e-5, g_opt=None, d_opt=None,
                          dy_override=None, g_masks=None, fake_override=None, d_masks=None):
    """One _train_step.  gw / dw / vw: {layer: (kernel, bias)}; u: {layer: [1,Cout]}; images NHWC in [-1,1].
    -> dict(losses, g_grads, d_grads, gw, dw, u, g_opt, d_opt, dy, y) with the updated state (fp64).
    dy_override (NHWC) replaces d g_loss / d G(lr) in the generator's backward pass: the loss gradient has kinks (ReLU / max-pool /
    |.| in the loss networks) where two correct implementations that differ in the last bit of y may legitimately take different
    branches, so a test checks dy itself away from those kinks and the generator's backward on one common dy.
    g_masks ({layer: bool NHWC array}): the generator's ReLU / LeakyReLU branches in ITS backward pass pinned to the ones another evaluation took;
    fake_override (NHWC) / d_masks ({"real": {...}, "fake": {...}}): the same for the discriminator update -- its input G(lr) and its LeakyReLU branches."""
    tm = lambda d: None if d is None else {n: torch.tensor(np.asarray(a)).permute(0, 3, 1, 2) for n, a in d.items()}
    f64 = lambda w: {n: (np.asarray(k, np.float64), np.asarray(b, np.float64)) for n, (k, b) in w.items()}
    gw, dw, vw = f64(gw), f64(dw), f64(vw)
    g_opt = g_opt or AdamRef(gw, g_lr)
    d_opt = d_opt or AdamRef(dw, d_lr)
    x = torch.tensor(np.asarray(lr_img, np.float64)).permute(0, 3, 1, 2)
    hr = torch.tensor(np.asarray(hr_img, np.float64)).permute(0, 3, 1, 2)
    # ---- discriminator update: D(real) with K1 = SN(K0), D(fake) with K2 = SN(K1); gradients add; Adam acts on K2
    with torch.no_grad():
        fake = generator_forward_t(_params(gw), x, scale, num_rrdb, attention)
    if fake_override is not None:
        fake = torch.tensor(np.asarray(fake_override, np.float64)).permute(0, 3, 1, 2)
    dw, u = _sn_inplace(dw, u)
    p1 = _params(dw)
    d_real = discriminator_forward_t(p1, hr, tm((d_masks or {}).get("real")))
    l_real = _bce(torch.ones_like(d_real), d_real)
    l_real.backward()
    dw, u = _sn_inplace(dw, u)
    p2 = _params(dw)
    d_fake = discriminator_forward_t(p2, fake, tm((d_masks or {}).get("fake")))
    l_fake = _bce(torch.zeros_like(d_fake), d_fake)
    l_fake.backward()
    g1, g2 = _grads(p1), _grads(p2)
    d_grads = {n: (g1[n][0] + g2[n][0], g1[n][1] + g2[n][1]) for n in g1}
    dw = d_opt.apply(dw, d_grads)
    # ---- generator update (third renormalisation of D)
    pg = _params(gw)
    y = generator_forward_t(pg, x, scale, num_rrdb, attention, masks=tm(g_masks))
    y.retain_grad()
    dw, u = _sn_inplace(dw, u)
    pd = {n: (torch.tensor(k), torch.tensor(b)) for n, (k, b) in dw.items()}
    pv = {n: (torch.tensor(k), torch.tensor(b)) for n, (k, b) in vw.items()}
    d_f = discriminator_forward_t(pd, y)
    adv = _bce(torch.ones_like(d_f), d_f)
    perc = torch.mean((vgg19_features_t(pv, hr) - vgg19_features_t(pv, y)) ** 2)
    pix = torch.mean(torch.abs(hr - y))
    yn, hn = y.permute(0, 2, 3, 1), hr.permute(0, 2, 3, 1)                       # NHWC: fft2 over the innermost (W, C) axes
    spec = torch.mean(torch.abs(torch.abs(torch.fft.fft2(hn.to(torch.complex128))) - torch.abs(torch.fft.fft2(yn.to(torch.complex128)))))
    g_loss = adv + 1.0 * perc + 100.0 * pix + 1.0 * spec
    g_loss.backward(retain_graph=dy_override is not None)
    dy = y.grad.permute(0, 2, 3, 1).numpy().copy()
    if dy_override is not None:
        for k_, b_ in pg.values():
            k_.grad = None
            b_.grad = None
        y.backward(torch.tensor(np.asarray(dy_override, np.float64)).permute(0, 3, 1, 2))
    g_grads = _grads(pg)
    gw = g_opt.apply(gw, g_grads)
    val = lambda t_: float(t_.detach().item())
    losses = {"g_loss": val(g_loss), "d_loss": val(l_real + l_fake), "adversarial": val(adv), "perceptual": val(perc), "pixel": val(pix),
              "spectral": val(spec)}
    return dict(losses=losses, g_grads=g_grads, d_grads=d_grads, gw=gw, dw=dw, u=u, g_opt=g_opt, d_opt=d_opt,
                dy=dy, y=y.detach().permute(0, 2, 3, 1).numpy())


# =====================================================================================================================
# FineTunedVGG16.fit with the default frozen base (VGG16_model.py:84-97, :111-157): the two Dense layers, torch autograd fp64
# =====================================================================================================================
def vgg16_head_fit_ref(g_train_batches, g_val, y_val, w, lr=1e-3, l2_reg=0.0):
    """g_train_batches: [[(features [n,512], labels)] per epoch] (no dropout: deterministic).  -> (head weights fp64, history dict)."""
    head = {n: (np.asarray(w[n][0], np.float64), np.asarray(w[n][1], np.float64)) for n in ("dense", "predictions")}
    opt = AdamRef(head, lr)
    hist = {"loss": [], "accuracy": [], "val_loss": [], "val_accuracy": []}

    def forward(p, g):
        h = torch.relu(torch.tensor(np.asarray(g, np.float64)) @ p["dense"][0] + p["dense"][1])
        return torch.softmax(h @ p["predictions"][0] + p["predictions"][1], dim=1)

    def cce(prob, y):
        pc = torch.clamp(prob[torch.arange(len(y)), torch.tensor(np.asarray(y, np.int64))], 1e-7, 1.0 - 1e-7)
        return torch.mean(-torch.log(pc))

    for batches in g_train_batches:
        tot, n = np.zeros(2), 0
        for g, y in batches:
            p = _params(head)
            prob = forward(p, g)
            loss = cce(prob, y) + l2_reg * (p["dense"][0] ** 2).sum()
            loss.backward()
            tot += [float(loss.item()) * len(y), float((prob.argmax(dim=1).numpy() == np.asarray(y)).mean()) * len(y)]
            n += len(y)
            head = opt.apply(head, _grads(p))
        with torch.no_grad():
            p = {k: (torch.tensor(a), torch.tensor(b)) for k, (a, b) in head.items()}
            prob = forward(p, g_val)
            vl = float((cce(prob, y_val) + l2_reg * (p["dense"][0] ** 2).sum()).item())
            va = float((prob.argmax(dim=1).numpy() == np.asarray(y_val)).mean())
        hist["loss"].append(tot[0] / n); hist["accuracy"].append(tot[1] / n); hist["val_loss"].append(vl); hist["val_accuracy"].append(va)
    return head, hist
