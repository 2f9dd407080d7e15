"""Forward halves of the ESRGAN training step (BASELINE configs[3]) against the oracle: discriminator with
stride-2 SAME convs (ESRGAN_model.py:347-377), VGG19 extractor with caffe preprocessing (:379-408), the four terms of the generator
loss (:410-473) and ESRGAN.evaluate's avg_g_loss (:782-856)."""
import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ops as O
from sr355 import Model
from sr355.weights import init_weights

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("hw", [(48, 48), (24, 40), (21, 35)])          # the notebook's 48 -> 24 -> 12 -> 6 maps; odd sizes pad 1 / 1
def test_discriminator_forward(ctx, hw):
    m = Model("esrgan_d", compute_dtype="f32", ctx=ctx)
    assert m.layer_shapes() == M.discriminator_layers()
    assert m.count_params() == 658305                                     # + 961 spectral-norm u vectors = 659 266 (ESRGAN.ipynb:L693-695)
    w = init_weights(m.layer_shapes(), seed=5000)
    m.set_weights(w)
    x = np.random.default_rng(3).uniform(-1, 1, (4, hw[0], hw[1], 3)).astype(np.float32)
    ref = M.discriminator_forward(x, w, dtype=np.float64)
    got = m.forward(ctx.to_device(x)).cpu().numpy()
    assert got.shape == ref.shape == (4, 1)
    assert np.max(np.abs(got - ref)) <= 1e-6, float(np.max(np.abs(got - ref)))
    # stage check: the last conv map (after three stride-2 convs) element by element
    _, taps = m.forward_with_taps(ctx.to_device(x), ["stride2_pick"])
    h = x.astype(np.float64)
    for i, st in enumerate(M.DISC_STRIDES):
        h = O.conv2d(h, *w[f"disc_conv{i + 1}"], stride=st, act="lrelu", dtype=np.float64)
    assert taps["stride2_pick"].shape == h.shape and rel_l2(taps["stride2_pick"].cpu().numpy(), h) <= 1e-5


def test_spectral_normalisation_training_semantics_in_the_oracle():
    """training=True renormalises every stored kernel in place before the layer runs: three calls per train step."""
    w = init_weights(M.discriminator_layers(), seed=5000)
    u = {n: np.random.default_rng(i).normal(0, 0.02, (1, k.shape[-1])).astype(np.float32) for i, (n, (k, _)) in enumerate(w.items())}
    x = np.random.default_rng(3).uniform(-1, 1, (2, 24, 24, 3)).astype(np.float32)
    p1, w1, u1 = M.discriminator_forward(x, w, u, training=True)
    assert p1.shape == (2, 1) and set(u1) == set(w)
    for n in w:
        k0, k1 = w[n][0].reshape(-1, w[n][0].shape[-1]), w1[n][0].reshape(-1, w1[n][0].shape[-1])
        sigma = np.linalg.norm(k0) / np.linalg.norm(k1)                  # kernel / sigma: one positive scalar per layer
        assert np.allclose(k0 / sigma, k1, rtol=1e-5, atol=1e-7) and np.array_equal(w[n][1], w1[n][1])
    assert np.allclose(M.discriminator_forward(x, w1), p1, atol=1e-6)     # inference on the updated kernels = what training returned


def test_vgg19_features_with_preprocessing(ctx):
    m = Model("vgg19_features", compute_dtype="f32", ctx=ctx)
    assert m.layer_shapes() == M.vgg19_extractor_layers()
    assert m.count_params() == 20024384                                   # ESRGAN.ipynb:L748
    w = init_weights(m.layer_shapes(), scheme="he_normal", seed=6000)
    m.set_weights(w)
    x = np.random.default_rng(4).uniform(-1, 1, (2, 48, 64, 3)).astype(np.float32)
    ref = M.vgg19_features(O.vgg19_preprocess(x), w, dtype=np.float64)
    got = m.forward(ctx.to_device(x)).cpu().numpy()
    assert got.shape == ref.shape == (2, 3, 4, 512)
    assert rel_l2(got, ref) <= 2e-5, rel_l2(got, ref)


@pytest.mark.parametrize("shape", [(3, 24, 48, 3), (2, 17, 96, 3), (1, 5, 37, 3)])
def test_pixel_and_spectral_loss(ctx, shape):
    rng = np.random.default_rng(8)
    a = rng.uniform(-1, 1, shape).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(shape), -1, 1).astype(np.float32)
    ad, bd = ctx.to_device(a), ctx.to_device(b)
    assert abs(float(ctx.l1(ad, bd).item()) - O.pixel_loss(a, b)) <= 1e-6
    ref = O.spectral_loss(a, b)                                           # FFT over (W, C): SURVEY.md A.9
    assert abs(float(ctx.spectral_l1(ad, bd).item()) - ref) <= 2e-5 * max(1.0, ref), (float(ctx.spectral_l1(ad, bd).item()), ref)
    assert float(ctx.spectral_l1(ad, ad).item()) == 0.0
    with pytest.raises(ValueError):
        ctx.spectral_l1(ctx.to_device(a[..., :2].copy()), ctx.to_device(b[..., :2].copy()))


def test_generator_loss_and_evaluate_avg_g_loss(ctx):
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    m = ESRGAN(compute_dtype="f32")
    m.setup_model(scale_factor=2, growth_channels=8, num_rrdb_blocks=1)
    m.set_weights(m.weights)
    rng = np.random.default_rng(11)
    lr = rng.uniform(-1, 1, (3, 24, 24, 3)).astype(np.float32)
    hr = rng.uniform(-1, 1, (3, 48, 48, 3)).astype(np.float32)
    fake = m.generate(lr)
    g, parts = m.generator_loss(hr, fake)
    rg, rparts = M.generator_loss(hr, fake, m.d_weights, m.vgg_weights, dtype=np.float64)
    for k in parts:
        assert abs(parts[k] - rparts[k]) <= 2e-5 * max(1.0, abs(rparts[k])), (k, parts[k], rparts[k])
    assert abs(g - rg) <= 1e-4 * max(1.0, abs(rg))
    ev = m.evaluate([(lr[:2], hr[:2]), (lr[2:], hr[2:])])
    r1 = M.generator_loss(hr[:2], fake[:2], m.d_weights, m.vgg_weights, dtype=np.float64)[0]
    r2 = M.generator_loss(hr[2:], fake[2:], m.d_weights, m.vgg_weights, dtype=np.float64)[0]
    assert abs(ev["avg_g_loss"] - 0.5 * (r1 + r2)) <= 1e-4 * max(1.0, abs(r1))      # mean of per-batch means (Appendix C.8)
    with pytest.raises(ValueError):                       # fit exists now (tests/test_train_gpu.py); without data it raises as the reference does
        m.fit()


def test_spectral_loss_kernels_follow_the_image_width_on_one_context(ctx):
    """the spectral-loss kernels size their LDS by the image width (56 / 80 bytes per pixel).  The dynamic-LDS attribute of
    a kernel used to be set once, at the first width above 48 KiB, and a later wider image failed with an opaque HIP error.  700 then 1500
    pixels on the same context, forward and backward, against the oracle; widths that cannot fit a CU's 160 KiB are refused by name."""
    rng = np.random.default_rng(8)
    for W in (700, 1500, 900):                                   # 1500 raises both attributes; 900 afterwards must still work
        a = rng.uniform(-1, 1, (1, 2, W, 3)).astype(np.float32)
        b = rng.uniform(-1, 1, (1, 2, W, 3)).astype(np.float32)
        ad, bd = ctx.to_device(a), ctx.to_device(b)
        ref = O.spectral_loss(a, b)
        got = float(ctx.spectral_l1(ad, bd).item())
        assert abs(got - ref) <= 5e-5 * max(1.0, ref), (W, got, ref)
        g = ctx.spectral_l1_bwd(ad, bd, 1.0)
        assert tuple(g.shape) == a.shape and bool(torch.isfinite(g).all())
        eps = 1e-3                                               # directional derivative against the oracle's finite difference
        d = rng.standard_normal(a.shape).astype(np.float32)
        fd = (O.spectral_loss(a + eps * d, b) - O.spectral_loss(a - eps * d, b)) / (2 * eps)
        dd = float((g.cpu().numpy().astype(np.float64) * d).sum())
        assert abs(dd - fd) <= 0.05 * abs(fd) + 1e-6, (W, dd, fd)
    z = ctx.to_device(np.zeros((1, 1, 3000, 3), np.float32))
    with pytest.raises(ValueError, match="exceeds"):
        ctx.spectral_l1(z, z)
    z2 = ctx.to_device(np.zeros((1, 1, 2100, 3), np.float32))
    with pytest.raises(ValueError, match="exceeds"):
        ctx.spectral_l1_bwd(z2, z2, 1.0)
