"""ESRGAN._train_step on MI355X (reference: ESRGAN_model.py:475-533; networks :212-399, losses :401-473, optimisers :176-195).

One training step = discriminator update (BCE on D(real) vs 1 and D(G(lr)) vs 0) then generator update (BCE(1, D(fake)) + 1.0 x
perceptual + 100 x L1 + 1.0 x spectral), both Adam with a staircase learning-rate decay, D's SpectralNormalization wrappers
renormalising their kernels in place on each of the three training=True calls.

The forward and backward passes are host orchestration over the C ABI's single ops, fp32 as the reference trains:
  * convs: the MFMA forward kernels; input gradients are the same kernels on 180-degree-rotated, channel-swapped weights; kernel / bias
    gradients the fp32-MFMA wgrad kernel (csrc/train_ops.hip);
  * SelfAttention of the small training patches is materialised (sr_matmul + row softmax) so that its backward is six GEMMs and a
    softmax-backward kernel; pooling / stride-2 sampling / depth_to_space / activations / the (W,C)-FFT loss have their own adjoint kernels;
  * a minimal reverse-mode tape (below) keeps the graph bookkeeping out of the kernels; torch is used for memory only (cat / slice /
    flip / expand: data movement).
Tiny vectors -- the discriminator's [B,256] dense head, BCE on [B,1], spectral-norm power iteration, Adam -- run on the host in NumPy.
Parameters are uploaded once per step and packed into MFMA fragment order on the device per layer call (sr_conv2d_dev); the
optimisers and the tape bookkeeping stay on the host.  cfg3's throughput is recorded by tools/bench_train.py, not yet a bench line.
"""
import numpy as np
import torch

from . import _lib as L
from .train import Adam, DeviceAdam, _rot


# ----------------------------------------------------------------------------------------------------------------- tape
class Var:
    __slots__ = ("v", "g", "need")

    def __init__(self, v, need=True):
        self.v, self.g, self.need = v, None, need


class Tape:
    """Reverse-mode bookkeeping: ops push a closure; backward() runs them last-to-first.  Parameter gradients land in `grads`
    ({layer: [dk, db]}, summed over uses) when `wgrad` is on."""

    def __init__(self, ctx, weights, wgrad=True, devcache=None):
        self.ctx, self.w, self.wgrad = ctx, weights, wgrad
        self.ops, self.grads = [], {}
        self.masks = None                                       # diagnostics: a dict here collects {layer: activation > 0} of every ReLU / LeakyReLU conv
        self.dev = devcache if devcache is not None else {}     # id(host array) -> (host array, device tensor): one upload per array

    def _dev(self, a):
        hit = self.dev.get(id(a))
        if hit is None or hit[0] is not a:
            hit = (a, self.ctx.to_device(np.ascontiguousarray(a, np.float32)))
            self.dev[id(a)] = hit
        return hit[1]

    def _acc(self, var, g):
        if not var.need:
            return
        var.g = g if var.g is None else self.ctx.eltwise(L.ELT_AXPBY, var.g, g, 1.0, 1.0)

    def _pgrad(self, name, dw, db):
        if name in self.grads:
            self.grads[name][0] = self.ctx.eltwise(L.ELT_AXPBY, self.grads[name][0], dw, 1.0, 1.0)
            self.grads[name][1] = self.ctx.eltwise(L.ELT_AXPBY, self.grads[name][1], db, 1.0, 1.0)
        else:
            self.grads[name] = [dw, db]

    # ---- ops
    def conv(self, x, name, act="linear", d2s=1, kernel=None):
        """Keras Conv2D SAME stride 1 (+ activation, + depth_to_space for the upsample blocks)."""
        ctx = self.ctx
        k, b = kernel if kernel is not None else self.w[name]
        kd = self._dev(k)
        y = Var(ctx.conv2d_dev(x.v, kd, self._dev(b), k.shape[3], act=act, d2s=d2s))
        if self.masks is not None and act in ("relu", "lrelu"):
            self.masks[name] = y.v > 0

        def bwd():
            if y.g is None:
                return
            dz = y.g
            if act == "relu":
                dz = ctx.eltwise(L.ELT_RELU_BWD, dz, y.v)
            elif act == "lrelu":
                dz = ctx.eltwise(L.ELT_LRELU_BWD, dz, y.v)
            elif act == "tanh":
                dz = ctx.eltwise(L.ELT_TANH_BWD, dz, y.v)
            if d2s > 1:
                dz = ctx.space_to_depth(dz, d2s)                   # activation is element-wise: its mask commutes with the shuffle
            if self.wgrad:
                dw, db = ctx.conv2d_wgrad(x.v, dz, k.shape[0])
                self._pgrad(name, dw, db)
            if x.need:
                self._acc(x, ctx.conv2d_dev(dz, kd, None, k.shape[2], rot=True))
        self.ops.append(bwd)
        return y

    def cat(self, xs):
        if len(xs) == 1:
            return xs[0]
        y = Var(torch.cat([x.v for x in xs], dim=-1).contiguous())
        sizes = [x.v.shape[-1] for x in xs]

        def bwd():
            if y.g is None:
                return
            o = 0
            for x, c in zip(xs, sizes):
                self._acc(x, y.g[..., o:o + c].contiguous())
                o += c
        self.ops.append(bwd)
        return y

    def dense_block(self, x, name, growth):
        """ESRGAN._dense_block (ESRGAN_model.py:212-254) as ONE tape op on a virtual-concat buffer: F = [x | f1 | f2 | f3 | f4] lives in one
        [B,H,W,64+4g] tensor, conv k reads its channel prefix and writes slice k (sr_conv2d_dev_views); returns x + 0.2 * conv5(F).  Backward: the
        gradient of F is one buffer too -- conv5's input gradient fills it, every growth conv's accumulates into its prefix IN PLACE (skip = output
        range), ReLU masks and weight gradients read their slices where they lie.  Round 3 built F by four torch.cat copies and took the gradient apart
        by ~14 slice copies and as many accumulate launches per block."""
        ctx = self.ctx
        B, H, W, C0 = x.v.shape
        g = int(growth)
        Ct = C0 + 4 * g
        F = ctx.empty((B, H, W, Ct))
        ctx.eltwise_view(L.ELT_AXPBY, x.v, 0, None, 0, F, 0, C0, 1.0, 0.0)
        ws = [self.w[f"{name}_conv{k}"] for k in range(1, 6)]
        for k in range(1, 5):
            kk, bb = ws[k - 1]
            cin = C0 + (k - 1) * g
            ctx.conv2d_dev_view(F, 0, cin, self._dev(kk), self._dev(bb), g, F, cin, act="relu")
            if self.masks is not None:
                self.masks[f"{name}_conv{k}"] = F[..., cin:cin + g] > 0
        k5, b5 = ws[4]
        y = Var(ctx.empty((B, H, W, C0)))
        ctx.conv2d_dev_view(F, 0, Ct, self._dev(k5), self._dev(b5), C0, y.v, 0, alpha=0.2, skip_buf=F, skip_coff=0, beta1=1.0)      # x + 0.2 * conv5

        def bwd():
            if y.g is None:
                return
            G = ctx.empty((B, H, W, Ct))
            dz5 = ctx.eltwise(L.ELT_AXPBY, y.g, None, 0.2, 0.0)
            if self.wgrad:
                self._pgrad(f"{name}_conv5", *ctx.conv2d_wgrad_view(F, 0, Ct, dz5, 0, C0, 3))
            ctx.conv2d_dev_view(dz5, 0, C0, self._dev(k5), None, Ct, G, 0, rot=True)                        # fills every channel of G
            dz = ctx.empty((B, H, W, g))
            for k in range(4, 0, -1):
                kk, _ = ws[k - 1]
                cin = C0 + (k - 1) * g
                ctx.eltwise_view(L.ELT_RELU_BWD, G, cin, F, cin, dz, 0, g)                                 # slice k of G is complete: convs k+1..5 have added to it
                if self.wgrad:
                    self._pgrad(f"{name}_conv{k}", *ctx.conv2d_wgrad_view(F, 0, cin, dz, 0, g, 3))
                ctx.conv2d_dev_view(dz, 0, g, self._dev(kk), None, cin, G, 0, rot=True, skip_buf=G, skip_coff=0, beta1=1.0)   # G[..., :cin] += dgrad
            if x.need:
                dx = ctx.empty((B, H, W, C0))
                ctx.eltwise_view(L.ELT_AXPBY, G, 0, y.g, 0, dx, 0, C0, 1.0, 1.0)                           # + the skip path's share
                self._acc(x, dx)
        self.ops.append(bwd)
        return y

    def axpby(self, a, b, alpha, beta):
        y = Var(self.ctx.eltwise(L.ELT_AXPBY, a.v, b.v, alpha, beta))

        def bwd():
            if y.g is None:
                return
            self._acc(a, y.g if alpha == 1.0 else self.ctx.eltwise(L.ELT_AXPBY, y.g, None, alpha, 0.0))
            self._acc(b, y.g if beta == 1.0 else self.ctx.eltwise(L.ELT_AXPBY, y.g, None, beta, 0.0))
        self.ops.append(bwd)
        return y

    def attention(self, x, name):
        """SelfAttention.call (ESRGAN_model.py:48-70), materialised: s = g f^T, beta = softmax(s), o = beta h, y = x + v(o)."""
        ctx = self.ctx
        B, H, W, C = x.v.shape
        N = H * W
        f, g, h = self.conv(x, name + "_f"), self.conv(x, name + "_g"), self.conv(x, name + "_h")
        f3, g3, h3 = f.v.reshape(B, N, -1), g.v.reshape(B, N, -1), h.v.reshape(B, N, -1)
        beta = ctx.softmax_rows_(ctx.matmul(g3, f3, trans_b=True))            # [B,N,N]
        o = Var(ctx.matmul(beta, h3).reshape(B, H, W, -1))

        def bwd():
            if o.g is None:
                return
            do = o.g.reshape(B, N, -1)
            dbeta = ctx.matmul(do, h3, trans_b=True)                           # [B,N,N]
            self._acc(h, ctx.matmul(beta, do, trans_a=True).reshape(h.v.shape))
            ds = ctx.softmax_bwd(beta, dbeta)
            self._acc(g, ctx.matmul(ds, f3).reshape(g.v.shape))
            self._acc(f, ctx.matmul(ds, g3, trans_a=True).reshape(f.v.shape))
        self.ops.append(bwd)
        ov = self.conv(o, name + "_v")
        return self.axpby(x, ov, 1.0, 1.0)

    def pick2(self, x):
        y = Var(self.ctx.spatial_op(L.SP_PICK2, x.v))
        H, W = x.v.shape[1:3]

        def bwd():
            if y.g is not None:
                self._acc(x, self.ctx.zero_insert2(y.g, H, W))
        self.ops.append(bwd)
        return y

    def maxpool(self, x):
        y = Var(self.ctx.spatial_op(L.SP_MAXPOOL2, x.v))

        def bwd():
            if y.g is not None:
                self._acc(x, self.ctx.maxpool2_bwd(x.v, y.g))
        self.ops.append(bwd)
        return y

    def backward(self):
        for op in reversed(self.ops):
            op()
        self.ops = []


# ----------------------------------------------------------------------------------------------------------------- networks
# dense blocks on one virtual-concat buffer per block (Tape.dense_block) where the growth width is a whole number of the fp32 conv's 16-channel chunks
# (G = 32: yes; the notebook's G = 8: the cat path); SR355_TAPE_CAT=1 restores round 3's cat path for A/B runs
import os as _os
VIRTUAL_CONCAT = _os.environ.get("SR355_TAPE_CAT") is None


def generator_forward(t, x, scale, num_rrdb, attention=True):
    """ESRGAN_model.py:303-345 on the tape; x Var [B,h,w,3] in [-1,1]."""
    x = t.conv(x, "initial_conv")
    trunk = x
    for b in range(num_rrdb):
        r_in = x
        for d in (1, 2, 3):
            n = f"rrdb_{b}_dense{d}"
            growth = t.w[f"{n}_conv1"][0].shape[3]
            if VIRTUAL_CONCAT and growth % 16 == 0:
                x = t.dense_block(x, n, growth)
                continue
            feats = [x]
            for k in range(1, 5):
                feats.append(t.conv(t.cat(feats), f"{n}_conv{k}", act="relu"))
            x = t.axpby(x, t.conv(t.cat(feats), f"{n}_conv5"), 1.0, 0.2)
        x = t.axpby(r_in, x, 1.0, 0.2)
    x = t.axpby(trunk, t.conv(x, "trunk_conv"), 1.0, 1.0)
    if attention:
        x = t.attention(x, "self_attention_trunk")
    s, i = scale, 0
    while s > 1:
        x = t.conv(x, f"upsample_{i}_conv", act="lrelu", d2s=2)
        if i == 0 and attention:
            x = t.attention(x, "self_attention_upsample_0")
        s >>= 1
        i += 1
    x = t.conv(x, "final_conv1", act="relu")
    return t.conv(x, "final_conv2", act="tanh")


DISC_STRIDES = [1, 2, 1, 2, 1, 2]
DISC_LAYERS = [f"disc_conv{i}" for i in range(1, 7)] + ["disc_dense1", "disc_output"]


def spectral_normalize(kernel, u):
    """tfa SpectralNormalization.normalize_weights, one power iteration -> (kernel / sigma, new u)."""
    w = kernel.reshape(-1, kernel.shape[-1]).astype(np.float64)
    u = u.reshape(1, -1).astype(np.float64)
    l2n = lambda a: a / np.sqrt(max(float(np.sum(a * a)), 1e-12))
    v = l2n(u @ w.T)
    u = l2n(v @ w)
    sigma = float((v @ w @ u.T).item())
    return (kernel / np.float32(sigma)).astype(np.float32), u.astype(np.float32)


def discriminator_forward(t, x, training, u=None):
    """ESRGAN_model.py:347-377 on the tape.  training=True first renormalises every stored kernel in place (t.w and u are updated).
    Returns (probabilities np [B,1], backward(dp) -> None that seeds the tape with d loss / d probabilities)."""
    ctx = t.ctx
    if training:
        for n in DISC_LAYERS:
            k, nu = spectral_normalize(t.w[n][0], u[n])
            t.w[n], u[n] = (k, t.w[n][1]), nu
    h = x
    for i, st in enumerate(DISC_STRIDES):
        h = t.conv(h, f"disc_conv{i + 1}", act="lrelu")
        if st == 2:
            h = t.pick2(h)
    B, H, W, C = h.v.shape
    g = ctx.spatial_op(L.SP_GAP, h.v).cpu().numpy().astype(np.float64)          # [B,256]: the head runs on the host
    k1, b1 = (a.astype(np.float64) for a in t.w["disc_dense1"])
    k2, b2 = (a.astype(np.float64) for a in t.w["disc_output"])
    z1 = g @ k1 + b1
    a1 = np.where(z1 > 0, z1, 0.2 * z1)
    z2 = a1 @ k2 + b2
    p = 1.0 / (1.0 + np.exp(-z2))

    def seed(dp):
        """dp = d loss / d p  [B,1]."""
        dz2 = dp * p * (1.0 - p)
        if t.wgrad:
            t.grads["disc_output"] = [a1.T @ dz2, dz2.sum(axis=0)]
        da1 = dz2 @ k2.T
        dz1 = np.where(z1 > 0, da1, 0.2 * da1)
        if t.wgrad:
            t.grads["disc_dense1"] = [g.T @ dz1, dz1.sum(axis=0)]
        dg = (dz1 @ k1.T) / float(H * W)                                         # GAP backward: spread over the H x W map
        dgt = ctx.to_device(dg.astype(np.float32))
        h.g = dgt[:, None, None, :].expand(B, H, W, C).contiguous()
    return p, seed


VGG19_CFG = [(1, 2, 64), (2, 2, 128), (3, 4, 256), (4, 4, 512), (5, 4, 512)]


def vgg19_features(t, x):
    """ESRGAN_model.py:379-408 on the tape (frozen: run it on a Tape with wgrad=False); x Var in [-1,1]."""
    ctx = t.ctx
    pre = Var(ctx.spatial_op(L.SP_VGG_PREPROCESS, x.v))

    def bwd():
        if pre.g is not None and x.need:      # (x+1)*127.5 with RGB -> BGR: the adjoint flips the channels back and scales
            t._acc(x, ctx.eltwise(L.ELT_AXPBY, torch.flip(pre.g, dims=[-1]).contiguous(), None, 127.5, 0.0))
    t.ops.append(bwd)
    h = pre
    for blk, n, _ in VGG19_CFG:
        for k in range(1, n + 1):
            h = t.conv(h, f"block{blk}_conv{k}", act="relu")
            if (blk, k) == (5, 4):
                return h
        h = t.maxpool(h)
    return h


def bce_mean(target, p, eps=1e-7):
    """mean(keras.backend.binary_crossentropy) on probabilities and its gradient w.r.t. p (clip passes the gradient inside [eps, 1-eps])."""
    pc = np.clip(p, eps, 1.0 - eps)
    loss = float(np.mean(-(target * np.log(pc + eps) + (1.0 - target) * np.log(1.0 - pc + eps))))
    inside = ((p >= eps) & (p <= 1.0 - eps)).astype(np.float64)
    dp = -(target / (pc + eps) - (1.0 - target) / (1.0 - pc + eps)) * inside / p.size
    return loss, dp


def staircase_lr(lr0, step, decay_steps=10000, decay_rate=0.5):
    """ExponentialDecay(staircase=True) (ESRGAN_model.py:176-195)."""
    return lr0 * decay_rate ** (step // decay_steps)


# ----------------------------------------------------------------------------------------------------------------- the step
class ESRGANTrainer:
    """Holds generator / discriminator / VGG19 weights (host fp32), the SN vectors u, the two Adam states and the step counter."""

    def __init__(self, ctx, g_weights, d_weights, vgg_weights, scale, num_rrdb, attention=True, g_lr=1e-4, d_lr=1e-5, u_seed=0, allreduce=None,
                 allreduce_flat=None):
        self.ctx, self.scale, self.nb, self.att = ctx, scale, num_rrdb, attention
        # Generator (16.9 M parameters at the default depth): resident on the device as ONE flat fp32 bucket with its Adam moments beside it;
        # the per-layer tensors the tape multiplies with are views of it, the gradients are gathered into a bucket of the same order, and
        # the optimiser is one fused kernel (round 2: NumPy Adam on the host, 65 ms of a 258 ms step, plus 70 MB each way over PCIe).
        # `self.gw` stays available as host arrays: they are refreshed from the device when somebody reads them.
        self._gw = {n: (np.array(k, np.float32), np.array(b, np.float32)) for n, (k, b) in g_weights.items()}
        arrs = [a for pair in self._gw.values() for a in pair]
        self._gflat = ctx.to_device(np.concatenate([a.ravel() for a in arrs]))
        self._gdev, o = {}, 0
        for a in arrs:
            self._gdev[id(a)] = (a, self._gflat[o:o + a.size].view(tuple(a.shape)))
            o += a.size
        self._g_host_stale = False
        # d_weights / vgg_weights None: a generator-only trainer (pixel_step: sr355.recipes' L1 fit); train_step then refuses
        self.dw = None if d_weights is None else {n: (np.asarray(k, np.float32), np.asarray(b, np.float32)) for n, (k, b) in d_weights.items()}
        self.vw = vgg_weights
        rng = np.random.default_rng(u_seed)               # tfa initialises u ~ TruncatedNormal(stddev 0.02), shape [1, Cout]
        self.u = None if self.dw is None else {n: np.clip(rng.normal(0, 0.02, (1, self.dw[n][0].shape[-1])), -0.04, 0.04).astype(np.float32)
                                               for n in DISC_LAYERS}
        self.g_lr0, self.d_lr0 = g_lr, d_lr
        self.g_opt = DeviceAdam(ctx, self._gflat, g_lr, epsilon=1e-7)
        self.d_opt = None if self.dw is None else Adam(self.dw, d_lr, epsilon=1e-7)
        self.step = 0
        # data parallel: `allreduce` = callable(dict of host grads) -> averaged dict (the discriminator's, whose spectral normalisation lives on
        # the host); `allreduce_flat` = callable(flat device tensor) -> averaged tensor for the generator's bucket (RCCL reduces it where it
        # lies).  With only `allreduce` given the generator's bucket takes the dict route too (the gloo CPU tests).
        self.allreduce, self.allreduce_flat = allreduce, allreduce_flat
        self._last = None
        self._packs = {}

    def _prepack(self, with_vgg):
        """Pack every generator (and VGG19) conv's weights for its forward and its input-gradient use by ONE launch (sr_conv_prepack; ~700 of a step's ~800 per-use
        packs, 3.5 ms of 53).  The discriminator's kernels are renormalised inside the step and keep packing per use.  The list lives until `_unpack`, which every
        step calls before it returns: nobody else on this context may meet a pack older than the weights."""
        if with_vgg not in self._packs:
            uses = []
            devs = [(self._gw, self._gdev)] + ([(self.vw, self._vggc)] if with_vgg else [])
            for w, dev in devs:
                for k, b in w.values():
                    if getattr(k, "ndim", 0) == 4 and k.shape[0] == k.shape[1] and k.shape[0] in (1, 3) and id(k) in dev and id(b) in dev:
                        kd, bd = dev[id(k)][1], dev[id(b)][1]
                        uses += [(kd, bd, False), (kd, None, True)]
            self._packs[with_vgg] = self.ctx.pack_list(uses)
        self.ctx.conv_prepack(self._packs[with_vgg])

    def _unpack(self):
        self.ctx.conv_prepack(None)

    @property
    def gw(self):
        """{layer: (kernel, bias)} host copies of the generator's parameters (refreshed from the device bucket when it has moved on)."""
        if self._g_host_stale:
            flat, o = self._gflat.cpu().numpy(), 0
            for pair in self._gw.values():
                for a in pair:
                    np.copyto(a, flat[o:o + a.size].reshape(a.shape))
                    o += a.size
            self._g_host_stale = False
        return self._gw

    @gw.setter
    def gw(self, weights):
        self.load_generator_weights(weights)

    def load_generator_weights(self, weights, reset_optimizer=True):
        """Replace the generator's parameters (ESRGAN.set_weights / load after the trainer exists): copies into the device bucket in
        parameter order; by default Adam's moments and step count start over, as a freshly compiled Keras model's would."""
        if set(weights) != set(self._gw):
            raise ValueError("generator weights: layer names differ from the trainer's graph")
        parts = []
        for n, pair in self._gw.items():
            for a, new in zip(pair, weights[n]):
                new = np.asarray(new, np.float32)
                if new.shape != a.shape:
                    raise ValueError(f"{n}: shape {new.shape} != {a.shape}")
                np.copyto(a, new)
                parts.append(a.ravel())
        self._gflat.copy_(self.ctx.to_device(np.concatenate(parts)))
        self._g_host_stale = False
        if reset_optimizer:
            self.g_opt.m.zero_()
            self.g_opt.v.zero_()
            self.g_opt.t = 0

    def _bucket_to_dict(self, flat):
        """flat host array in parameter order -> {layer: (dk, db)} views"""
        out, o = {}, 0
        for n, pair in self._gw.items():
            pr = []
            for a in pair:
                pr.append(flat[o:o + a.size].reshape(a.shape))
                o += a.size
            out[n] = tuple(pr)
        return out

    def _gather_grads(self, grads):
        """{layer: [dk, db]} device tensors of the generator's tape -> one flat device bucket in parameter order (zeros where the loss does
        not reach a variable: its moments and value then stay put, as Keras' skipping of None gradients leaves them)."""
        parts = []
        for n, pair in self._gw.items():
            g = grads.get(n)
            for s_, a in enumerate(pair):
                t = None if g is None else g[s_]
                parts.append(torch.zeros(a.size, dtype=torch.float32, device=self._gflat.device) if t is None else
                             (t if isinstance(t, torch.Tensor) else self.ctx.to_device(np.asarray(t, np.float32))).reshape(-1))
        return torch.cat(parts)

    def generator_tape(self, wgrad=True):
        """A tape over the generator's device-resident parameters (no upload, no host copy)."""
        return Tape(self.ctx, self._gw, wgrad=wgrad, devcache=dict(self._gdev))

    @property
    def last_grads(self):
        """{"g": {layer: (dk, db)}, "d": {...}} host arrays of the last step (the generator's are downloaded on first use)."""
        if self._last is None:
            return None
        if "g" not in self._last:
            full = self._bucket_to_dict(self._last.pop("g_flat").cpu().numpy())
            self._last["g"] = {n: full[n] for n in full if n in self._last["g_names"]}     # only the variables the loss reaches, as Keras reports them
        return self._last

    def _host(self, grads):
        """{layer: [dk, db]} (device tensors, or host arrays for the discriminator's dense head) -> host fp32 arrays; the device ones
        cross PCIe as ONE flat buffer."""
        dev = [(n, s) for n, pair in grads.items() for s in (0, 1) if isinstance(pair[s], torch.Tensor)]
        out = {n: [None, None] for n in grads}
        if dev:
            flat = torch.cat([grads[n][s].reshape(-1) for n, s in dev]).cpu().numpy()
            o = 0
            for n, s in dev:
                t = grads[n][s]
                out[n][s] = flat[o:o + t.numel()].reshape(tuple(t.shape))
                o += t.numel()
        for n, pair in grads.items():
            for s in (0, 1):
                if out[n][s] is None:
                    out[n][s] = np.asarray(pair[s], np.float32)
        return {n: (a, b) for n, (a, b) in out.items()}

    def _upload(self, weights, cache):
        """All arrays of a parameter dict in one host-to-device copy; `cache` (Tape.devcache) then maps every array to its view."""
        arrs = [a for pair in weights.values() for a in pair]
        flat = self.ctx.to_device(np.concatenate([np.asarray(a, np.float32).ravel() for a in arrs]))
        o = 0
        for a in arrs:
            cache[id(a)] = (a, flat[o:o + a.size].view(tuple(a.shape)))
            o += a.size

    def pixel_step(self, lr_images, hr_images):
        """One generator update on the pixel loss alone (mean |hr - G(lr)|, ESRGAN_model.py:433-445; the generator half of _train_step,
        :506-531, without the adversarial / perceptual / spectral terms): sr355.recipes' fit.  -> the L1 value before the update."""
        ctx = self.ctx
        lr_t, hr_t = ctx.to_device(np.asarray(lr_images, np.float32)), ctx.to_device(np.asarray(hr_images, np.float32))
        tg = Tape(ctx, self._gw, devcache=dict(self._gdev))
        self._prepack(False)
        try:
            y = generator_forward(tg, Var(lr_t, need=False), self.scale, self.nb, self.att)
            pix = float(ctx.l1(hr_t, y.v).item())
            y.g = ctx.eltwise(L.ELT_SIGN_DIFF, y.v, hr_t, 1.0 / y.v.numel(), 0.0)
            tg.backward()
        finally:
            self._unpack()
        g_flat = self._gather_grads(tg.grads)
        if self.allreduce_flat is not None:
            g_flat = self.allreduce_flat(g_flat)
        self.g_opt.lr = staircase_lr(self.g_lr0, self.step)
        self.g_opt.apply(self._gflat, g_flat)
        self._g_host_stale = True
        self.step += 1
        return pix

    def train_step(self, lr_images, hr_images):
        """-> {'g_loss', 'd_loss', parts...}; weights, u, optimiser states advance in place (ESRGAN_model.py:475-533)."""
        if self.dw is None or self.vw is None:
            raise RuntimeError("ESRGANTrainer was built without discriminator / VGG19 weights: only pixel_step is available")
        if not hasattr(self, "_vggc"):
            self._vggc = {}                                # the frozen VGG19 stays on the device
            self._upload(self.vw, self._vggc)
        self._prepack(True)                                # the generator's weights change only in the step's last line, VGG19's never
        try:
            return self._train_step(lr_images, hr_images)
        finally:
            self._unpack()

    def _train_step(self, lr_images, hr_images):
        ctx = self.ctx
        lr_t, hr_t = ctx.to_device(np.asarray(lr_images, np.float32)), ctx.to_device(np.asarray(hr_images, np.float32))
        devc = {}                                          # device copies of this step's parameter arrays (one upload per array)
        if not hasattr(self, "_vggc"):
            self._vggc = {}                                # the frozen VGG19 stays on the device
            self._upload(self.vw, self._vggc)
        devc.update(self._gdev)                            # the generator's parameters are already there (views of the flat bucket)
        # The reference runs the generator twice per step, once under each tape (ESRGAN_model.py:490, :508); its weights do not change
        # in between (the discriminator is updated first), so both runs are the same tensor: one taped forward serves both.
        tg = Tape(ctx, self._gw, devcache=devc)
        collect = getattr(self, "collect_masks", False)           # tests: the activation branches of this step's forward passes (oracle/train.py _masked_act)
        self.last_masks = {"g": {}, "d_real": {}, "d_fake": {}} if collect else None
        if collect:
            tg.masks = self.last_masks["g"]
        y = generator_forward(tg, Var(lr_t, need=False), self.scale, self.nb, self.att)
        fake = y.v
        # ---- discriminator update
        td = Tape(ctx, self.dw, devcache=devc)
        if collect:
            td.masks = self.last_masks["d_real"]
        p_real, seed_real = discriminator_forward(td, Var(hr_t, need=False), True, self.u)       # renormalisation 1
        l_real, dp = bce_mean(np.ones_like(p_real), p_real)
        seed_real(dp)
        td.backward()
        g_real = self._host(td.grads)
        td2 = Tape(ctx, self.dw, devcache=devc)
        if collect:
            td2.masks = self.last_masks["d_fake"]
        p_fake, seed_fake = discriminator_forward(td2, Var(fake, need=False), True, self.u)      # renormalisation 2
        l_fake, dp = bce_mean(np.zeros_like(p_fake), p_fake)
        seed_fake(dp)
        td2.backward()
        g_fake = self._host(td2.grads)
        d_grads = {n: (g_real[n][0] + g_fake[n][0], g_real[n][1] + g_fake[n][1]) for n in g_real}
        if self.allreduce is not None:
            d_grads = self.allreduce(d_grads)
        self.d_opt.lr = staircase_lr(self.d_lr0, self.step)
        self.dw = self.d_opt.apply(self.dw, d_grads)
        # ---- generator update
        td3 = Tape(ctx, self.dw, wgrad=False, devcache=devc)
        yv = Var(y.v)
        p, seed = discriminator_forward(td3, yv, True, self.u)                                   # renormalisation 3
        self.dw = td3.w
        adv, dp = bce_mean(np.ones_like(p), p)
        seed(dp)
        td3.backward()
        tv = Tape(ctx, self.vw, wgrad=False, devcache=self._vggc)
        fr = vgg19_features(tv, Var(hr_t, need=False))
        tv.ops = []
        yv2 = Var(y.v)
        ff = vgg19_features(tv, yv2)
        perc = float(ctx.mse(fr.v, ff.v).item())
        ff.g = ctx.eltwise(L.ELT_AXPBY, ff.v, fr.v, 2.0 / ff.v.numel(), -2.0 / ff.v.numel())
        tv.backward()
        pix = float(ctx.l1(hr_t, y.v).item())
        spec = float(ctx.spectral_l1(y.v, hr_t).item())
        dy = ctx.eltwise(L.ELT_SIGN_DIFF, y.v, hr_t, 100.0 / y.v.numel(), 0.0)
        dy = ctx.eltwise(L.ELT_AXPBY, dy, ctx.spectral_l1_bwd(y.v, hr_t, 1.0), 1.0, 1.0)
        dy = ctx.eltwise(L.ELT_AXPBY, dy, yv.g, 1.0, 1.0)
        dy = ctx.eltwise(L.ELT_AXPBY, dy, yv2.g, 1.0, 1.0)
        y.g = dy
        self.last_dy = dy
        tg.backward()
        g_flat = self._gather_grads(tg.grads)
        if self.allreduce_flat is not None:
            g_flat = self.allreduce_flat(g_flat)
        elif self.allreduce is not None:                   # dict route (host): the 2-rank gloo tests
            avg = self.allreduce(self._bucket_to_dict(g_flat.cpu().numpy()))
            g_flat = ctx.to_device(np.concatenate([np.asarray(a, np.float32).ravel() for n in self._gw for a in avg[n]]))
        self.g_opt.lr = staircase_lr(self.g_lr0, self.step)
        self.g_opt.apply(self._gflat, g_flat)
        self._g_host_stale = True
        self.step += 1
        self._last = {"g_flat": g_flat, "g_names": set(tg.grads), "d": d_grads}
        self.last_fake = fake
        return {"g_loss": adv + 1.0 * perc + 100.0 * pix + 1.0 * spec, "d_loss": l_real + l_fake, "adversarial": adv, "perceptual": perc,
                "pixel": pix, "spectral": spec}
