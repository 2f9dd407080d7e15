"""A seeded recipe for generator weights whose output RESEMBLES the HR image (no trained checkpoint exists in the reference's snapshot).

Why: the north star's parity clause is |PSNR(gpu, HR) - PSNR(cpu reference, HR)| <= 0.01 dB.  On random-init weights the generator's
output is noise relative to HR (9.6 dB) and that figure cannot fail; the reference's own generators are trained (28.8-31.3 dB,
ESRGAN.ipynb:L3723-3725), where the same storage noise weighs a hundred times more.  This module builds, from seeds only, a weight set
for the reference graph (ESRGAN_model.py:303-345) in that regime:

  1. `near_identity_generator`: an analytic start.  The image rides through the network as a +/- channel pair (LeakyReLU and ReLU are
     invertible on a pair: lrelu(a) - lrelu(-a) = 1.2 a, relu(a) - relu(-a) = a), the two up-sampling convs hold the 3 x 3 taps of a
     half-pixel bilinear x2 interpolation per sub-pixel of depth_to_space (DCR order), final_conv2 undoes the pair in front of tanh.
     Every other weight is Keras' glorot-uniform draw, scaled so that the residual branches are perturbations of the stream and not
     its equal (conv5 of every dense block, trunk_conv -- which sees the 1.2^23 = 66 x gain of the RRDB stream --, the attention's
     output projection): ~17-24 dB against HR on the synthetic tiles, about what bilinear interpolation gives.
  2. `GeneratorPixelFit`: a short L1-only fit of ALL generator parameters on LR / HR crops of the synthetic tiles, fp32, on the
     training kernels of sr355.gan_train (the generator half of ESRGAN._train_step with the pixel loss alone, Adam as there), so that
     the branches carry real detail and no weight sits on an analytic lattice.

Both steps are deterministic (seeded draws, fixed-order reductions on the device): the recipe, not a blob, is what the repository holds.
"""
import numpy as np

from .weights import condition_attention, init_weights

# gains of the analytic image path from stage to stage: generic numbers on purpose -- with unit gains every stage maps bf16 lattice
# points onto (almost) lattice points and the storage rounding of the next stage acquires a systematic sign
_G0, _G1, _G2, _G3 = 0.8, 0.9, 1.1, 0.7
_BILINEAR = {0: np.array([0.25, 0.75, 0.0]), 1: np.array([0.0, 0.75, 0.25])}     # half-pixel bilinear x2: taps over (y-1, y, y+1) per sub-pixel


def near_identity_generator(layer_shapes, seed=7000, eps_out=0.02, g_conv5=0.3, g_trunk=0.005, g_v=0.1, gain_out=1.1):
    """-> {layer: (kernel HWIO, bias)} fp32 for an ESRGAN generator graph of any depth / growth / power-of-two scale."""
    w = condition_attention(init_weights(layer_shapes, seed=seed))
    out = {}
    for n, (k, b) in w.items():
        k, b = k.copy(), b.copy()
        if n == "initial_conv":
            k[..., :6], b[:6] = 0.0, 0.0
            for c in range(3):
                k[1, 1, c, c], k[1, 1, c, 3 + c] = _G0, -_G0
        elif n.endswith("_conv5"):
            k *= g_conv5
            b *= g_conv5
        elif n == "trunk_conv":
            k *= g_trunk
            b *= g_trunk
        elif n.endswith("_v"):
            k *= g_v
            b *= g_v
        elif n.startswith("upsample_") and n.endswith("_conv"):
            first = n == "upsample_0_conv"
            k *= eps_out
            b *= eps_out
            # the +/- pair in front: (v, -v) behind initial_conv / the trunk, (lrelu(v), lrelu(-v)) behind a LeakyReLU: difference 2 v resp. 1.2 v
            g = (_G1 / (2.0 * _G0)) if first else (_G2 / (1.2 * _G1))
            for i in (0, 1):
                for j in (0, 1):
                    taps, o = g * np.outer(_BILINEAR[i], _BILINEAR[j]), (i * 2 + j) * 64
                    for c in range(3):
                        k[..., o + c], k[..., o + 3 + c], b[o + c], b[o + 3 + c] = 0.0, 0.0, 0.0, 0.0
                        k[:, :, c, o + c], k[:, :, c, o + 3 + c] = taps, -taps
                        k[:, :, 3 + c, o + c], k[:, :, 3 + c, o + 3 + c] = -taps, taps
        elif n == "final_conv1":
            k *= eps_out
            b *= eps_out
            g = _G3 / (1.2 * _G2)
            for c in range(3):
                k[..., c], k[..., 3 + c], b[c], b[3 + c] = 0.0, 0.0, 0.0, 0.0
                k[1, 1, c, c], k[1, 1, 3 + c, c] = g, -g
                k[1, 1, c, 3 + c], k[1, 1, 3 + c, 3 + c] = -g, g
        elif n == "final_conv2":
            k *= eps_out
            b *= eps_out
            for c in range(3):
                k[1, 1, c, c] += gain_out / _G3
                k[1, 1, 3 + c, c] -= gain_out / _G3
        out[n] = (k.astype(np.float32), b.astype(np.float32))
    return out


def crop_batches(lr_tiles, hr_tiles, scale, patch_lr, batch, steps, seed):
    """Seeded random LR / HR crops of the tiles, in [-1, 1] as the generator takes them (ESRGAN_model.py:883-941): a generator of
    `steps` batches ([batch, p, p, 3], [batch, p s, p s, 3])."""
    rng = np.random.default_rng(seed)
    T, H, W, _ = lr_tiles.shape
    for _ in range(steps):
        xs, ys = [], []
        for _ in range(batch):
            t, y, x = int(rng.integers(T)), int(rng.integers(H - patch_lr + 1)), int(rng.integers(W - patch_lr + 1))
            xs.append(lr_tiles[t, y:y + patch_lr, x:x + patch_lr])
            ys.append(hr_tiles[t, y * scale:(y + patch_lr) * scale, x * scale:(x + patch_lr) * scale])
        yield np.stack(xs).astype(np.float32) * 2.0 - 1.0, np.stack(ys).astype(np.float32) * 2.0 - 1.0


class GeneratorPixelFit:
    """L1-only fit of the generator on the device: forward on the tape of sr355.gan_train, d mean|y - hr| / dy, backward, one Adam update
    of the flat parameter bucket (the generator half of ESRGAN._train_step, ESRGAN_model.py:506-531, with the pixel term alone)."""

    def __init__(self, ctx, weights, scale, num_rrdb, attention=True, learning_rate=1e-4):
        from .gan_train import ESRGANTrainer
        self.tr = ESRGANTrainer(ctx, weights, None, None, scale, num_rrdb, attention=attention, g_lr=learning_rate)

    def step(self, lr_batch, hr_batch):
        return self.tr.pixel_step(lr_batch, hr_batch)

    @property
    def weights(self):
        return {n: (k.copy(), b.copy()) for n, (k, b) in self.tr.gw.items()}


def trained_like_generator(ctx, layer_shapes, lr_tiles, hr_tiles, scale=4, num_rrdb=23, attention=True, steps=600, batch=16, patch_lr=24,
                           learning_rate=2e-4, seed=7000, log=None):
    """The whole recipe: analytic start + `steps` L1 steps on seeded crops of (lr_tiles, hr_tiles) in [0, 1].  -> fp32 weights."""
    w0 = near_identity_generator(layer_shapes, seed=seed)
    if steps <= 0:
        return w0
    fit = GeneratorPixelFit(ctx, w0, scale, num_rrdb, attention, learning_rate)
    for i, (x, y) in enumerate(crop_batches(lr_tiles, hr_tiles, scale, patch_lr, batch, steps, seed + 1)):
        l1 = fit.step(x, y)
        if log is not None and (i % 50 == 0 or i == steps - 1):
            log(i, l1)
    return fit.weights
