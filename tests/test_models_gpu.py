"""Model-level parity through sr_forward vs the CPU oracle graphs, seeded synthetic weights.

fp32 models: rel-L2 <= 1e-5 vs the fp32 oracle run in fp64.
bf16 models: fp32 oracle on bf16-rounded weights/inputs; activations are stored in bf16 between
layers, so the bound is loose (rel-L2 <= 3e-2) and a PSNR floor is asserted as well.
"""
import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ops as O
from sr355 import Model
from sr355.weights import bf16_rounded, init_weights, round_to_bf16

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def run(ctx, model, w, x, dtype):
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    model.set_weights(w)
    return model.forward(ctx.to_device(x, td)).float().cpu().numpy()


def prep(w, x, dtype):
    if dtype == "bf16":
        return bf16_rounded(w), round_to_bf16(x)
    return w, x


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_srcnn_forward(ctx, dtype):
    m = Model("srcnn", compute_dtype=dtype, ctx=ctx)
    assert m.layer_shapes() == M.srcnn_layers()
    assert m.count_params() == 28931                      # SRCNN.ipynb:L141
    w = init_weights(m.layer_shapes(), seed=1000)
    x = np.random.default_rng(1).uniform(0, 1, (3, 33, 33, 3)).astype(np.float32)
    w, x = prep(w, x, dtype)
    ref = M.srcnn_forward(x, w, dtype=np.float64)
    got = run(ctx, m, w, x, dtype)
    assert rel_l2(got, ref) <= (1e-5 if dtype == "f32" else 2e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("scale,nb", [(2, 16), (4, 3), (3, 2)])
def test_edsr_forward(ctx, scale, nb, dtype):
    m = Model("edsr", compute_dtype=dtype, scale_factor=scale, num_blocks=nb, num_filters=64, res_scaling=0.1, ctx=ctx)
    assert m.layer_shapes() == M.edsr_layers(scale, 3, nb, 64)
    if scale == 2 and nb == 16:
        assert m.count_params() == 1369859                # EDSR.ipynb:L392
    w = init_weights(m.layer_shapes(), scheme="he_normal", seed=2000)
    x = np.random.default_rng(2).uniform(0, 1, (2, 24, 24, 3)).astype(np.float32)
    w, x = prep(w, x, dtype)
    ref = M.edsr_forward(x, w, scale, nb, 0.1, dtype=np.float64)
    got = run(ctx, m, w, x, dtype)
    assert got.shape == ref.shape == (2, 24 * scale, 24 * scale, 3)
    assert rel_l2(got, ref) <= (1e-5 if dtype == "f32" else 3e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("scale,G,nb,att", [(2, 8, 4, True), (4, 32, 2, True), (4, 32, 2, False)])
def test_esrgan_generator_forward(ctx, scale, G, nb, att, dtype):
    m = Model("esrgan_g", compute_dtype=dtype, scale_factor=scale, num_blocks=nb, growth_channels=G, use_attention=att, ctx=ctx)
    if att:
        assert m.layer_shapes() == M.esrgan_g_layers(scale, G, nb)
    if (scale, G, nb, att) == (2, 8, 4, True):
        assert m.count_params() == 1162915                # ESRGAN.ipynb:L636
    w = init_weights(m.layer_shapes(), seed=3000)
    x = np.random.default_rng(3).uniform(-1, 1, (2, 24, 24, 3)).astype(np.float32)
    w, x = prep(w, x, dtype)
    ref = M.esrgan_g_forward(x, w, scale, nb, dtype=np.float64, attention=att)
    got = run(ctx, m, w, x, dtype)
    assert got.shape == ref.shape
    err = rel_l2(got, ref)
    assert err <= (2e-5 if dtype == "f32" else 3e-2), err
    if dtype == "bf16":   # outputs in [-1,1] -> [0,1] PSNR against the oracle
        assert O.psnr((got + 1) / 2, (ref + 1) / 2, dtype=np.float64).min() >= 38.0


@pytest.mark.parametrize("hw", [(19, 37), (5, 3), (48, 17)])
def test_esrgan_row_blocked_buffers_ragged_shapes(ctx, hw):
    """bf16 with G = 32 keeps the concat buffers row-blocked ([B][H][C/32][W][32], csrc/conv_common.h): image sizes that are not
    multiples of the 16x16 / 12x16 tiles, narrower than one tile, and a batch of 3 -- against the fp64 oracle."""
    H, W = hw
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=1, growth_channels=32, use_attention=False, ctx=ctx)
    w = init_weights(m.layer_shapes(), seed=3200)
    x = np.random.default_rng(5).uniform(-1, 1, (3, H, W, 3)).astype(np.float32)
    w, x = prep(w, x, "bf16")
    ref = M.esrgan_g_forward(x, w, 2, 1, dtype=np.float64, attention=False)
    got = run(ctx, m, w, x, "bf16")
    assert got.shape == ref.shape == (3, 2 * H, 2 * W, 3)
    assert rel_l2(got, ref) <= 3e-2
    assert np.array_equal(got, run(ctx, m, w, x, "bf16"))      # same buffers, second call: same image


def test_esrgan_predict_chunking_invariant(ctx):
    """keras predict(batch_size) chunking must not change results (no batch-coupled op)."""
    m = Model("esrgan_g", compute_dtype="f32", scale_factor=2, num_blocks=1, growth_channels=8, ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), seed=3100))
    x = np.random.default_rng(4).uniform(-1, 1, (5, 24, 24, 3)).astype(np.float32)
    a = m.predict(x, batch_size=16)
    b = m.predict(x, batch_size=2)
    assert np.array_equal(a, b)
    assert m.predict(x[:0], batch_size=16).shape == (0, 48, 48, 3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_vgg16_classifier_forward(ctx, dtype):
    m = Model("vgg16", compute_dtype=dtype, num_classes=2, ctx=ctx)
    assert m.layer_shapes() == M.vgg16_classifier_layers(2)
    assert m.count_params() == 14846530                   # VGG16.ipynb:L151-153
    w = init_weights(m.layer_shapes(), scheme="he_normal", seed=4000)
    x = np.random.default_rng(5).uniform(0, 1, (3, 96, 96, 3)).astype(np.float32)
    w, x = prep(w, x, dtype)
    ref = M.vgg16_classifier_forward(x, w, dtype=np.float64)
    got = run(ctx, m, w, x, dtype)
    assert got.shape == (3, 2)
    assert np.allclose(got.sum(axis=1), 1.0, atol=1e-2 if dtype == "bf16" else 1e-5)
    assert np.max(np.abs(got - ref)) <= (1e-5 if dtype == "f32" else 3e-2)


def test_forward_errors(ctx):
    m = Model("srcnn", compute_dtype="f32", ctx=ctx)
    x = ctx.to_device(np.zeros((1, 16, 16, 3), np.float32))
    with pytest.raises(RuntimeError):      # not finalised (weights never set)
        m.forward(x)
    with pytest.raises(KeyError):
        m.set_weights({})
    with pytest.raises(ValueError):
        Model("edsr", scale_factor=5, ctx=ctx)


def test_cfg1_full_size_srcnn_fp32(ctx):
    """BASELINE configs[1] at its full size: LR [32,256,256,3] fp32 -> bicubic x4 -> SRCNN -> [32,1024,1024,3] (33.55 MPix).  The oracle
    cannot run 1.9 TFLOP, so: the bicubic of two whole images against the oracle's cv2.resize restatement; the network on 76x76
    windows of the device's own up-scaled batch (receptive field radius 4 + 0 + 2 = 6: the window's 64x64 core is exact), at a
    corner, an edge and in the interior -- which also pins the SAME zero padding at full size; PSNR / SSIM of two output images; and
    the batch is independent image by image (image 7 alone gives the same bytes as image 7 of the batch)."""
    rng = np.random.default_rng(43)
    lr = rng.uniform(0, 1, (32, 256, 256, 3)).astype(np.float32)
    m = Model("srcnn", compute_dtype="f32", ctx=ctx)
    w = init_weights(m.layer_shapes(), seed=1000)
    m.set_weights(w)
    x = ctx.to_device(lr)
    up = ctx.bicubic(x, 1024, 1024)
    assert up.shape == (32, 1024, 1024, 3)
    for b in (0, 31):
        ref = O.cv_resize(lr[b], 1024, 1024, O.INTER_CUBIC)
        assert np.max(np.abs(up[b].cpu().numpy() - ref)) <= 2e-6
    y = m.forward(up)
    assert y.shape == (32, 1024, 1024, 3) and bool(torch.isfinite(y).all())
    uph = up.cpu().numpy()
    for b, y0, x0 in [(0, 0, 0), (5, 0, 500), (17, 960, 960), (31, 400, 0), (12, 300, 700)]:
        ys, xs = max(y0 - 6, 0), max(x0 - 6, 0)
        ye, xe = min(y0 + 70, 1024), min(x0 + 70, 1024)
        win = uph[b:b + 1, ys:ye, xs:xe]
        # the oracle pads the window with zeros at its borders: identical to the network only where the window border IS the image border
        ref = M.srcnn_forward(win, w, dtype=np.float64)[0]
        cy0, cx0 = y0 - ys, x0 - xs
        core_ref = ref[cy0:cy0 + 64, cx0:cx0 + 64]
        core_got = y[b, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
        assert core_ref.shape == (64, 64, 3) and rel_l2(core_got, core_ref) <= 1e-5, (b, y0, x0, rel_l2(core_got, core_ref))
    hr = ctx.to_device(np.clip(uph[[3, 30]] + 0.02 * rng.standard_normal((2, 1024, 1024, 3)), 0, 1).astype(np.float32))
    sel = y[[3, 30]].contiguous()
    p, s = ctx.psnr(hr, sel).cpu().numpy(), ctx.ssim(hr, sel).cpu().numpy()
    pr = O.psnr(hr.cpu().numpy(), sel.cpu().numpy(), dtype=np.float64)
    sr = O.ssim(hr.cpu().numpy(), sel.cpu().numpy(), dtype=np.float64)
    assert np.max(np.abs(p - pr)) <= 1e-3 and np.max(np.abs(s - sr)) <= 1e-4
    alone = m.forward(up[7:8].contiguous())
    assert torch.equal(alone[0], y[7])


# ---- SRCNN: conv2d_1 (1x1, 96 -> 32, ReLU) inside conv2d's epilogue (conv.hip conv_thin_kernel PW2; SRCNN_model.py:48-53) ----

@pytest.mark.parametrize("shape", [(2, 24, 16, 3), (1, 33, 33, 3), (3, 50, 21, 3), (1, 7, 5, 3)])     # one tile exactly; ragged both ways; several tiles; smaller than a tile
def test_srcnn_fused_1x1_matches_the_two_kernel_path_and_the_oracle(ctx, shape):
    m = Model("srcnn", compute_dtype="f32", ctx=ctx)
    w = init_weights(m.layer_shapes(), seed=1200 + shape[1])
    m.set_weights(w)
    x = np.random.default_rng(shape[2]).uniform(0, 1, shape).astype(np.float32)
    xd = ctx.to_device(x)
    try:
        ctx.set_fused(ctx.FUSED_ALL & ~256, 0)
        ctx.profile_begin()
        y0, t0 = m.forward_with_taps(xd, ["conv2d_1"])
        k0 = {r["kernel"] for r in ctx.profile_end()}
        ctx.set_fused(ctx.FUSED_ALL, 0)
        ctx.profile_begin()
        y1, t1 = m.forward_with_taps(xd, ["conv2d_1"])
        k1 = {r["kernel"] for r in ctx.profile_end()}
    finally:
        ctx.set_fused(ctx.FUSED_ALL, 0)
    assert not any("pw2" in k for k in k0) and any(k.startswith("conv_thin_pw2") for k in k1), (k0, k1)      # the fused kernel is the one that ran
    assert not any(k.startswith("conv_wide<f32,k1") for k in k1), k1                                         # ... and the 1x1 kernel did not
    assert torch.equal(y1, m.forward(xd))
    # a tap on the head needs its 96-channel output in memory: the pair then runs as two kernels
    y2, t2 = m.forward_with_taps(xd, ["conv2d"])
    assert torch.equal(y2, y0) and t2["conv2d"].shape == shape[:3] + (96,)
    a1 = O.conv2d(x, *w["conv2d"], act="relu", dtype=np.float64)
    a2 = O.conv2d(a1, *w["conv2d_1"], act="relu", dtype=np.float64)
    ref = M.srcnn_forward(x, w, dtype=np.float64)
    for got in (t0["conv2d_1"], t1["conv2d_1"]):
        assert rel_l2(got.cpu().numpy(), a2) <= 2e-6
    assert rel_l2(y1.cpu().numpy(), ref) <= 1e-5 and rel_l2(y0.cpu().numpy(), ref) <= 1e-5
    assert np.abs(t1["conv2d_1"].cpu().numpy() - t0["conv2d_1"].cpu().numpy()).max() <= 1e-5 * max(1.0, float(np.abs(a2).max()))


def test_srcnn_fused_1x1_exact_integers(ctx):
    """Index arithmetic, exactly: small integer weights and pixels make every partial sum an exactly representable integer whatever the order
    of the fp32 additions, so the fused kernel, the two-kernel path and the oracle must agree to the last bit -- every one of the 96 x 32
    channel pairs of the 1x1 carries its own weight (a mis-ordered k index cannot cancel), ReLU cuts at both layers, ragged tiles, a batch."""
    m = Model("srcnn", compute_dtype="f32", ctx=ctx)
    rng = np.random.default_rng(77)
    w = {"conv2d": (rng.integers(-2, 3, (9, 9, 3, 96)).astype(np.float32), rng.integers(-3, 4, 96).astype(np.float32)),
         "conv2d_1": (rng.integers(-3, 4, (1, 1, 96, 32)).astype(np.float32), rng.integers(-5, 6, 32).astype(np.float32)),
         "conv2d_2": (rng.integers(-1, 2, (5, 5, 32, 3)).astype(np.float32), np.zeros(3, np.float32))}
    m.set_weights(w)
    x = rng.integers(0, 3, (2, 29, 37, 3)).astype(np.float32)
    xd = ctx.to_device(x)
    try:
        ctx.set_fused(ctx.FUSED_ALL & ~256, 0)
        y0, t0 = m.forward_with_taps(xd, ["conv2d_1"])
        ctx.set_fused(ctx.FUSED_ALL, 0)
        y1, t1 = m.forward_with_taps(xd, ["conv2d_1"])
    finally:
        ctx.set_fused(ctx.FUSED_ALL, 0)
    a2 = O.conv2d(O.conv2d(x, *w["conv2d"], act="relu", dtype=np.float64), *w["conv2d_1"], act="relu", dtype=np.float64)
    assert a2.max() < 2 ** 24 and len(np.unique(a2)) > 1000 and (a2 == 0).mean() > 0.05          # exact in fp32, non-trivial, the second ReLU cuts
    assert np.array_equal(t1["conv2d_1"].cpu().numpy(), a2) and np.array_equal(t0["conv2d_1"].cpu().numpy(), a2)
    ref = M.srcnn_forward(x, w, dtype=np.float64)
    assert np.abs(ref).max() < 2 ** 24
    assert np.array_equal(y1.cpu().numpy(), ref) and np.array_equal(y0.cpu().numpy(), ref)
