"""Training rows (SURVEY.md 8f-4 and the conv backward kernels of 8f-1): the wgrad kernel and the dgrad-by-rotated-forward identity
against torch autograd, one SRCNN / EDSR step gradient by gradient, a few Adam steps weight by weight, and `fit` end to end
(Keras callbacks' semantics) -- oracle/train.py is the independent fp64 derivation."""
import os

import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ops as O
from oracle import train as OT
from sr355 import _lib as L
from sr355 import train as T
from sr355.weights import init_weights

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("case", [(2, 24, 24, 3, 96, 9), (2, 24, 24, 96, 32, 1), (3, 17, 13, 32, 3, 5), (2, 20, 12, 64, 64, 3), (1, 9, 11, 64, 256, 3),
                                  (1, 48, 48, 40, 70, 3)])
def test_conv_wgrad_and_dgrad(ctx, case):
    B, H, W, Cin, Cout, K = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    dy = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    xt = torch.tensor(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w.astype(np.float64), requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = OT._conv(xt, wt, bt)
    y.backward(torch.tensor(dy.astype(np.float64)).permute(0, 3, 1, 2))
    dw, db = ctx.conv2d_wgrad(ctx.to_device(x), ctx.to_device(dy), K)
    assert rel_l2(dw.cpu().numpy(), wt.grad.numpy()) <= 2e-6 and rel_l2(db.cpu().numpy(), bt.grad.numpy()) <= 2e-6
    if K == 9 and Cout > 4:
        return            # a 9x9 input gradient over 96 channels is never needed: SRCNN's 9x9 layer is the first one (its input is the data)
    dx = ctx.conv2d(ctx.to_device(dy), T._rot(w), None).cpu().numpy()
    assert rel_l2(dx, xt.grad.permute(0, 2, 3, 1).numpy()) <= 1e-5


def test_eltwise_and_space_to_depth(ctx):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((2, 6, 8, 12)).astype(np.float32)
    b = rng.standard_normal((2, 6, 8, 12)).astype(np.float32)
    ad, bd = ctx.to_device(a), ctx.to_device(b)
    assert np.allclose(ctx.eltwise(L.ELT_AXPBY, ad, bd, 0.5, -2.0).cpu().numpy(), 0.5 * a - 2.0 * b, atol=1e-6)
    assert np.array_equal(ctx.eltwise(L.ELT_RELU_BWD, ad, bd).cpu().numpy(), np.where(b > 0, a, 0))
    assert np.allclose(ctx.eltwise(L.ELT_LRELU_BWD, ad, bd).cpu().numpy(), np.where(b > 0, a, 0.2 * a))
    assert np.array_equal(ctx.eltwise(L.ELT_CLIP01_BWD, ad, bd).cpu().numpy(), np.where((b >= 0) & (b <= 1), a, 0))
    assert np.array_equal(ctx.eltwise(L.ELT_CLIP01, ad).cpu().numpy(), np.clip(a, 0, 1))
    assert np.allclose(ctx.eltwise(L.ELT_TANH_BWD, ad, bd).cpu().numpy(), a * (1 - b * b), atol=1e-6)
    for r in (2, 3):
        x = rng.standard_normal((2, 4, 5, 2 * r * r)).astype(np.float32)
        up = O.depth_to_space(x, r)
        assert np.array_equal(ctx.space_to_depth(ctx.to_device(up), r).cpu().numpy(), x)      # exact inverse of the DCR shuffle


def test_srcnn_step_gradients_and_adam(ctx):
    w = init_weights(M.srcnn_layers(), seed=1000)
    rng = np.random.default_rng(5)
    batches = [(rng.uniform(0, 1, (4, 24, 24, 3)).astype(np.float32), rng.uniform(0, 1, (4, 24, 24, 3)).astype(np.float32)) for _ in range(3)]
    x, t = batches[0]
    y, loss, g = T.srcnn_loss_and_grads(ctx, w, ctx.to_device(x), ctx.to_device(t))
    ry, rloss, rg = OT.loss_and_grads(OT.srcnn_forward_t, w, x, t)
    assert abs(float(loss.item()) - rloss) <= 1e-6 * max(1, rloss) and rel_l2(y.cpu().numpy(), ry) <= 1e-5
    for n in rg:
        assert rel_l2(g[n][0].cpu().numpy(), rg[n][0]) <= 2e-5, n
        assert rel_l2(g[n][1].cpu().numpy(), rg[n][1]) <= 2e-5, n
    opt, cur = T.Adam(w, learning_rate=1e-3), w
    for bx, bt in batches:
        _, _, gg = T.srcnn_loss_and_grads(ctx, cur, ctx.to_device(bx), ctx.to_device(bt))
        cur = opt.apply(cur, {n: (a.cpu().numpy(), b.cpu().numpy()) for n, (a, b) in gg.items()})
    ref, _ = OT.train_steps(OT.srcnn_forward_t, w, batches, 1e-3)
    for n in ref:   # three steps of lr 1e-3 move every weight by ~3e-3: compare the MOVEMENT
        assert rel_l2(cur[n][0] - w[n][0], ref[n][0] - w[n][0]) <= 2e-3, n


@pytest.mark.parametrize("scale", [2, 4, 3])
def test_edsr_step_gradients(ctx, scale):
    nb = 2
    w = init_weights(M.edsr_layers(scale, 3, nb, 64), scheme="he_normal", seed=2000)
    rng = np.random.default_rng(6)
    x = rng.uniform(0, 1, (2, 12, 10, 3)).astype(np.float32)
    t = rng.uniform(0, 1, (2, 12 * scale, 10 * scale, 3)).astype(np.float32)
    y, loss, g = T.edsr_loss_and_grads(ctx, w, ctx.to_device(x), ctx.to_device(t), scale=scale, num_res_blocks=nb)
    ry, rloss, rg = OT.loss_and_grads(OT.edsr_forward_t, w, x, t, scale=scale, num_res_blocks=nb)
    assert abs(float(loss.item()) - rloss) <= 1e-6 * max(1, rloss) and rel_l2(y.cpu().numpy(), ry) <= 1e-5
    assert set(g) == set(rg)
    for n in rg:
        assert rel_l2(g[n][0].cpu().numpy(), rg[n][0]) <= 5e-5, (n, rel_l2(g[n][0].cpu().numpy(), rg[n][0]))
        assert rel_l2(g[n][1].cpu().numpy(), rg[n][1]) <= 5e-5, n
    # clipnorm 1.0 + eps 1e-8 (EDSR_model.py:130-136): one step against the reference optimiser
    opt = T.Adam(w, learning_rate=1e-4, epsilon=1e-8, clipnorm=1.0)
    new = opt.apply(w, {n: (a.cpu().numpy(), b.cpu().numpy()) for n, (a, b) in g.items()})
    ref = OT.AdamRef(w, 1e-4, epsilon=1e-8, clipnorm=1.0).apply({n: (np.asarray(k, np.float64), np.asarray(b, np.float64)) for n, (k, b) in w.items()}, rg)
    for n in ref:
        assert rel_l2(new[n][0] - w[n][0], ref[n][0] - w[n][0]) <= 1e-3, n


def test_srcnn_fit_end_to_end(ctx):
    """SRCNNModel.fit: the loss goes down, the history has Keras' keys, ReduceLROnPlateau halves the rate on a plateau, EarlyStopping
    stops and restores the best epoch's weights, the trained model serves super_resolve_image."""
    from SRModels.deep_learning_models.SRCNN_model import SRCNNModel
    from sr355.synth import make_pairs
    lr, hr = make_pairs(12, 12, 12, 2, seed=31)
    up = np.stack([O.bicubic_resize(a, 24, 24) for a in lr]).astype(np.float32)
    m = SRCNNModel()
    m.setup_model(input_shape=(24, 24, 3), learning_rate=2e-3)
    with pytest.raises(RuntimeError):
        m.evaluate(up, hr)
    hist, tcb, mcb = m.fit(up[:8], hr[:8], up[8:], hr[8:], batch_size=4, epochs=6, shuffle=False, verbose=False)
    h = hist.history
    assert set(h) == {"loss", "psnr", "ssim", "val_loss", "val_psnr", "val_ssim", "lr"} and len(h["loss"]) == len(tcb.epoch_times_sec) == len(mcb.gpu_peak_mb) <= 6
    assert h["loss"][-1] < h["loss"][0] and tcb.mean_time_value() > 0 and mcb.gpu_peak_mb[0] > 0
    # the same steps from the oracle (no shuffle, no callback fired yet in epoch 0): loss of the first epoch = mean of its two batch losses
    w0 = init_weights(M.srcnn_layers(), seed=1000)
    _, losses = OT.train_steps(OT.srcnn_forward_t, w0, [(up[0:4], hr[0:4]), (up[4:8], hr[4:8])], 2e-3)
    assert abs(h["loss"][0] - np.mean(losses)) <= 1e-5 * max(1.0, np.mean(losses))
    res = m.evaluate(up[8:], hr[8:])
    assert abs(res[0] - min(h["val_loss"])) <= 1e-6 + 1e-4 * res[0]         # best (restored or last) weights are the ones loaded
    sr, _ = m.super_resolve_image(lr[0], 24, 24, patch_size=12, stride=6)
    assert sr.shape == (24, 24, 3)
    # plateau: an absurd learning rate makes val_loss stall -> the rate halves after 2 epochs, training stops after 3 without improvement
    m2 = SRCNNModel()
    m2.setup_model(input_shape=(24, 24, 3), learning_rate=0.5)
    hist2, _, _ = m2.fit(up[:8], hr[:8], up[8:], hr[8:], batch_size=4, epochs=12, shuffle=False, verbose=False)
    lrs = hist2.history["lr"]
    assert len(lrs) < 12 and min(lrs) < 0.5


def test_early_stopping_restores_the_first_epoch_when_val_loss_never_improves(ctx):
    """with a NaN val_loss from the first epoch on, Keras' EarlyStopping(restore_best_weights=True) stops after `patience` epochs
    and hands back the weights it snapshotted at the first epoch; the restatement used to crash on `set_weights(None)`."""
    rng = np.random.default_rng(3)
    w = init_weights(M.srcnn_layers(), seed=1)
    X, Y = rng.uniform(0, 1, (4, 12, 12, 3)).astype(np.float32), rng.uniform(0, 1, (4, 12, 12, 3)).astype(np.float32)
    nan_predict = lambda c, ww, x: torch.full_like(x, float("nan"))
    seen = []

    def lg(c, ww, x, t):
        seen.append({n: ww[n][0].copy() for n in ww})
        return T.srcnn_loss_and_grads(c, ww, x, t)
    out, hist, _, _ = T.fit(ctx, w, lg, nan_predict, T.Adam(w, 1e-3), X, Y, X[:2], Y[:2], batch_size=4, epochs=10, es_patience=3, shuffle=False, verbose=False)
    assert len(hist.history["val_loss"]) == 3 and all(np.isnan(v) for v in hist.history["val_loss"])      # wait reaches the patience of 3 at the third epoch
    assert all(np.array_equal(out[n][0], seen[1][n]) for n in out)          # the weights after epoch 1 (= what epoch 2's first batch started from)


def test_edsr_fit_wrapper(ctx):
    """EDSR.fit (EDSR_model.py:127-176): clipnorm / eps 1e-8 Adam through the reference-shaped class; the first epoch's loss equals
    the oracle's two steps; the trained model serves super_resolve_image."""
    from SRModels.deep_learning_models.EDSR_model import EDSR
    from sr355.synth import make_pairs
    lr, hr = make_pairs(10, 12, 12, 2, seed=33)
    m = EDSR()
    m.setup_model(scale_factor=2, num_res_blocks=2, learning_rate=1e-3)
    w0 = {n: (k.copy(), b.copy()) for n, (k, b) in m.weights.items()}
    with pytest.raises(RuntimeError):
        m.evaluate(lr, hr)
    hist, tcb, _ = m.fit(lr[:8], hr[:8], lr[8:], hr[8:], batch_size=4, epochs=3, shuffle=False, verbose=False)
    h = hist.history
    _, losses = OT.train_steps(OT.edsr_forward_t, w0, [(lr[0:4], hr[0:4]), (lr[4:8], hr[4:8])], 1e-3, epsilon=1e-8, clipnorm=1.0, scale=2, num_res_blocks=2)
    assert abs(h["loss"][0] - np.mean(losses)) <= 1e-5 * max(1.0, np.mean(losses)), (h["loss"][0], losses)
    assert h["loss"][-1] < h["loss"][0] and len(tcb.epoch_times_sec) == len(h["loss"])
    sr, _ = m.super_resolve_image(lr[0], patch_size_lr=12, stride=6)
    assert sr.shape == (24, 24, 3)


# ------------------------------------------------------------------------------------------------ ESRGAN._train_step (BASELINE configs[3])
def _gan_setup(scale, nb, G, seed=0):
    gw = init_weights(M.esrgan_g_layers(scale, G, nb), seed=3000)
    gw = {n: ((k * 0.25, b * 0.25) if (n.endswith("_f") or n.endswith("_g")) else (k, b)) for n, (k, b) in gw.items()}     # moderate attention logits
    dw = init_weights(M.discriminator_layers(), seed=5000)
    vw = init_weights(M.vgg19_extractor_layers(), scheme="he_normal", seed=6000)
    vw = {n: (k * 0.05 if n == "block1_conv1" else k, b) for n, (k, b) in vw.items()}          # inputs are +-128 after caffe preprocessing: keep the features O(1)
    return gw, dw, vw


def test_attention_forward_backward_materialised(ctx):
    from sr355 import gan_train as GT
    rng = np.random.default_rng(4)
    w = init_weights(M.self_attention_layers("sa"), seed=9)
    x = (0.5 * rng.standard_normal((2, 6, 5, 64))).astype(np.float32)
    dy = rng.standard_normal((2, 6, 5, 64)).astype(np.float32)
    t = GT.Tape(ctx, w)
    xv = GT.Var(ctx.to_device(x))
    y = t.attention(xv, "sa")
    assert rel_l2(y.v.cpu().numpy(), O.self_attention(x, *w["sa_f"], *w["sa_g"], *w["sa_h"], *w["sa_v"], dtype=np.float64)) <= 1e-5
    y.g = ctx.to_device(dy)
    t.backward()
    p = OT._params(w)
    xt = torch.tensor(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    OT._sa_t(p, xt, "sa").backward(torch.tensor(dy.astype(np.float64)).permute(0, 3, 1, 2))
    assert rel_l2(xv.g.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy()) <= 2e-5
    for n in w:
        assert rel_l2(t.grads[n][0].cpu().numpy(), p[n][0].grad.numpy()) <= 5e-5, n
        if n.endswith("_f"):       # a key bias shifts every score of a row by the same amount: softmax ignores it, the gradient is exactly zero
            assert np.abs(t.grads[n][1].cpu().numpy()).max() <= 1e-5
        else:
            assert rel_l2(t.grads[n][1].cpu().numpy(), p[n][1].grad.numpy()) <= 5e-5, n


def test_spectral_loss_gradient_and_pool_adjoints(ctx):
    rng = np.random.default_rng(6)
    a = rng.uniform(-1, 1, (2, 5, 24, 3)).astype(np.float32)
    b = rng.uniform(-1, 1, (2, 5, 24, 3)).astype(np.float32)
    at = torch.tensor(a.astype(np.float64), requires_grad=True)
    loss = torch.mean(torch.abs(torch.abs(torch.fft.fft2(at.to(torch.complex128))) - torch.abs(torch.fft.fft2(torch.tensor(b.astype(np.float64)).to(torch.complex128)))))
    loss.backward()
    got = ctx.spectral_l1_bwd(ctx.to_device(a), ctx.to_device(b), 1.0).cpu().numpy()
    assert rel_l2(got, at.grad.numpy()) <= 2e-4, rel_l2(got, at.grad.numpy())
    # maxpool backward and the stride-2 pick's adjoint against autograd
    x = rng.standard_normal((2, 8, 6, 5)).astype(np.float32)
    dy = rng.standard_normal((2, 4, 3, 5)).astype(np.float32)
    xt = torch.tensor(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    F = torch.nn.functional
    F.max_pool2d(xt, 2).backward(torch.tensor(dy.astype(np.float64)).permute(0, 3, 1, 2))
    assert np.allclose(ctx.maxpool2_bwd(ctx.to_device(x), ctx.to_device(dy)).cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-6)
    for hw in ((8, 6), (7, 5)):
        z = rng.standard_normal((2, hw[0], hw[1], 3)).astype(np.float32)
        pick = ctx.spatial_op(L.SP_PICK2, ctx.to_device(z))
        g = rng.standard_normal(tuple(pick.shape)).astype(np.float32)
        back = ctx.zero_insert2(ctx.to_device(g), hw[0], hw[1]).cpu().numpy()
        assert abs(float((back * z).sum()) - float((g * pick.cpu().numpy()).sum())) <= 1e-4        # <A^T g, z> == <g, A z>


@pytest.mark.parametrize("cfg", [(2, 1, 8, True), (4, 1, 32, True), (2, 2, 8, False)])
def test_esrgan_train_step(ctx, cfg):
    """One _train_step against the torch-autograd restatement: the six loss terms, every generator and discriminator gradient,
    the updated weights, and the discriminator kernels after their three in-place spectral renormalisations."""
    from sr355 import gan_train as GT
    scale, nb, G, att = cfg
    gw, dw, vw = _gan_setup(scale, nb, G)
    rng = np.random.default_rng(7)
    lr = rng.uniform(-1, 1, (2, 12, 12, 3)).astype(np.float32)
    hr = rng.uniform(-1, 1, (2, 12 * scale, 12 * scale, 3)).astype(np.float32)
    tr = GT.ESRGANTrainer(ctx, gw, dw, vw, scale, nb, attention=att, g_lr=1e-4, d_lr=1e-5, u_seed=3)
    u0 = {n: v.copy() for n, v in tr.u.items()}
    out = tr.train_step(lr, hr)
    ref = OT.esrgan_train_step_ref(gw, dw, u0, vw, lr, hr, scale, nb, attention=att)
    for k, v in ref["losses"].items():
        assert abs(out[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, out[k], v)
    for n, (rk, rb) in ref["d_grads"].items():
        assert rel_l2(tr.last_grads["d"][n][0], rk) <= 2e-4, ("d", n, rel_l2(tr.last_grads["d"][n][0], rk))
        assert rel_l2(tr.last_grads["d"][n][1], rb) <= 2e-4, ("d bias", n)
    # d g_loss / d G(lr): the loss networks have kinks (VGG's ReLUs and max-pools, |.| of the pixel term), and the two G(lr) differ in
    # their last bits, so a handful of pixels may sit on different branches; everywhere else the gradient agrees tightly.
    dy = tr.last_dy.cpu().numpy().astype(np.float64)
    off = np.abs(dy - ref["dy"]) > 1e-5 * np.abs(ref["dy"]).max()
    assert off.mean() <= 0.01, off.mean()
    assert rel_l2(np.where(off, 0, dy), np.where(off, 0, ref["dy"])) <= 1e-4
    # ... and the generator's backward pass is checked on one common dy (the device's), so that its tolerance stays tight.
    ref = OT.esrgan_train_step_ref(gw, dw, u0, vw, lr, hr, scale, nb, attention=att, dy_override=dy)
    assert set(tr.last_grads["g"]) == set(ref["g_grads"])
    errs = sorted(((rel_l2(tr.last_grads["g"][n][0], ref["g_grads"][n][0]), n) for n in ref["g_grads"]), reverse=True)
    assert errs[0][0] <= 2e-4, errs[:3]
    berrs = sorted(((float(np.abs(tr.last_grads["g"][n][1] - ref["g_grads"][n][1]).max() / max(np.abs(ref["g_grads"][n][1]).max(), 1e-3)), n)
                    for n in ref["g_grads"]), reverse=True)
    assert berrs[0][0] <= 2e-4, berrs[:3]
    for n in ref["dw"]:        # kernels after Adam and the third renormalisation, and the power-iteration vectors
        assert rel_l2(tr.dw[n][0], ref["dw"][n][0]) <= 1e-5 and rel_l2(tr.u[n], ref["u"][n]) <= 1e-5, n
    for n in ref["gw"]:
        assert rel_l2(tr.gw[n][0] - gw[n][0], ref["gw"][n][0] - gw[n][0]) <= 5e-3, n
    assert tr.step == 1 and GT.staircase_lr(1e-4, 9999) == 1e-4 and GT.staircase_lr(1e-4, 10000) == 5e-5


def test_device_adam_is_the_host_adam_bit_for_bit(ctx):
    """sr_adam (the fused optimiser kernel of the GAN trainer's flat generator bucket) against sr355.train.Adam (the NumPy restatement of
    keras.optimizers.Adam that SRCNN / EDSR fit and the discriminator use, itself checked against oracle/train.py's AdamRef): three steps on one
    bucket, with a gradient scale as the data-parallel path applies it."""
    rng = np.random.default_rng(12)
    w0 = {"a": (rng.standard_normal((3, 3, 8, 16)).astype(np.float32), rng.standard_normal(16).astype(np.float32))}
    host = T.Adam(w0, 1e-4, epsilon=1e-7)
    flat = lambda d: np.concatenate([d["a"][0].ravel(), d["a"][1].ravel()])
    wd = ctx.to_device(flat(w0))
    dev = T.DeviceAdam(ctx, wd, 1e-4, epsilon=1e-7)
    wh = w0
    for step in range(3):
        g = {"a": ((rng.standard_normal((3, 3, 8, 16)) * 10.0 ** rng.integers(-6, 2)).astype(np.float32), rng.standard_normal(16).astype(np.float32))}
        scale = 1.0 if step < 2 else 0.5
        wh = host.apply(wh, {"a": (g["a"][0] * np.float32(scale), g["a"][1] * np.float32(scale))})
        dev.apply(wd, ctx.to_device(flat(g)), grad_scale=scale)
        assert np.array_equal(wd.cpu().numpy(), flat(wh)), step
    assert np.array_equal(dev.m.cpu().numpy(), np.concatenate([host.m["a"][0].ravel(), host.m["a"][1].ravel()]))
    assert np.array_equal(dev.v.cpu().numpy(), np.concatenate([host.v["a"][0].ravel(), host.v["a"][1].ravel()]))
    with pytest.raises(ValueError):
        ctx.adam_step(wd, wd[:5].contiguous(), dev.m, dev.v, 1e-4)


def test_esrgan_train_step_at_the_baseline_configuration(ctx):
    """BASELINE configs[3] at its stated size: x4, NB = 23, G = 32, both SelfAttention layers, 16 LR patches
    24x24 -> 96x96 per GPU, fp32 (ESRGAN_model.py:475-533 with the defaults of :108-112).  Shapes and the parameter bucket
    (16 930 019 generator + 658 305 discriminator parameters, ESRGAN.ipynb:L636/L693), finite losses, two runs from the same state
    give the same numbers bit for bit; then, on 2 of the 16 patches (what the fp64 CPU oracle finishes in seconds at this depth), the
    six loss terms and sampled generator / discriminator gradients against oracle/train.py."""
    from sr355 import gan_train as GT
    scale, nb, G = 4, 23, 32
    gw, dw, vw = _gan_setup(scale, nb, G)
    # glorot-initialised RRDBs have gain ~1.2 per block: 23 of them would make the losses astronomically large; damp the residual branches
    gw = {n: ((k * 0.1, b * 0.1) if n.endswith("_conv5") else (k, b)) for n, (k, b) in gw.items()}
    count = lambda w: sum(int(np.prod(k.shape)) + int(np.prod(b.shape)) for k, b in w.values())
    assert count(gw) == 16930019 and count(dw) == 658305
    rng = np.random.default_rng(42 + 3)
    lr = rng.uniform(-1, 1, (16, 24, 24, 3)).astype(np.float32)
    hr = rng.uniform(-1, 1, (16, 96, 96, 3)).astype(np.float32)
    runs = []
    for _ in range(2):
        tr = GT.ESRGANTrainer(ctx, gw, dw, vw, scale, nb, attention=True, g_lr=1e-4, d_lr=1e-5, u_seed=3)
        out = tr.train_step(lr, hr)
        assert all(np.isfinite(float(v)) for v in out.values()), out
        assert set(tr.last_grads["g"]) == set(gw) and set(tr.last_grads["d"]) == set(dw)
        for n, (k, b) in gw.items():
            assert tr.last_grads["g"][n][0].shape == k.shape and tr.last_grads["g"][n][1].shape == b.shape, n
            assert tr.gw[n][0].shape == k.shape and np.isfinite(tr.gw[n][0]).all(), n
        bucket = sum(a.size for side in ("g", "d") for pair in tr.last_grads[side].values() for a in pair)
        assert bucket == 16930019 + 658305                       # what allreduce_mean_grads carries as one flat fp32 bucket (70.4 MB)
        assert tuple(tr.last_dy.shape) == (16, 96, 96, 3)
        runs.append((out, {n: tr.last_grads["g"][n][0].copy() for n in ("initial_conv", "rrdb_11_dense2_conv3", "final_conv2")},
                     tr.gw["rrdb_22_dense3_conv5"][0].copy(), tr.dw["disc_conv1"][0].copy()))
    assert runs[0][0] == runs[1][0]                               # deterministic reductions: the same losses ...
    assert all(np.array_equal(runs[0][1][n], runs[1][1][n]) for n in runs[0][1]) and np.array_equal(runs[0][2], runs[1][2]) and np.array_equal(runs[0][3], runs[1][3])   # ... gradients, weights
    # ---- the oracle at this depth, on two of the patches
    tr = GT.ESRGANTrainer(ctx, gw, dw, vw, scale, nb, attention=True, g_lr=1e-4, d_lr=1e-5, u_seed=3)
    tr.collect_masks = True
    u0 = {n: v.copy() for n, v in tr.u.items()}
    out = tr.train_step(lr[:2], hr[:2])
    dy = tr.last_dy.cpu().numpy().astype(np.float64)
    host = lambda d: {n: m.cpu().numpy() for n, m in d.items()}

    def dhost(d):
        # the device runs a stride-2 layer as the stride-1 conv + LeakyReLU sampled at every second position (offset 1 for even sizes, 0 for odd:
        # SURVEY.md A.1); the oracle convolves with stride 2 -- its mask is the sampled one
        out_ = {}
        for i, st in enumerate(GT.DISC_STRIDES):
            m = d[f"disc_conv{i + 1}"].cpu().numpy()
            if st == 2:
                m = m[:, (0 if m.shape[1] % 2 else 1)::2, (0 if m.shape[2] % 2 else 1)::2]
            out_[f"disc_conv{i + 1}"] = m
        return out_
    # The device's forward passes are fp32, the oracle's fp64: at the few elements whose pre-activation lies within the forward rounding error of zero
    # the two take different ReLU / LeakyReLU branches, and one flipped activation in 1.2 M moves a gradient by ~1e-3 (round 3 widened the bounds to 5e-3
    # on that argument).  Here the argument is tested instead: the oracle runs with the DEVICE's branches (oracle/train.py _masked_act) and, for the
    # discriminator update, with the device's G(lr) -- what is left is arithmetic, and the bound is 5e-4 again.
    ref = OT.esrgan_train_step_ref(gw, dw, u0, vw, lr[:2], hr[:2], scale, nb, attention=True, dy_override=dy, g_masks=host(tr.last_masks["g"]),
                                   fake_override=tr.last_fake.cpu().numpy(), d_masks={"real": dhost(tr.last_masks["d_real"]), "fake": dhost(tr.last_masks["d_fake"])})
    assert len(tr.last_masks["g"]) == 23 * 3 * 4 + 2 + 1 and len(tr.last_masks["d_real"]) == 6
    for k, v in ref["losses"].items():
        assert abs(out[k] - v) <= 5e-4 * max(1.0, abs(v)), (k, out[k], v)
    errs = {n: rel_l2(tr.last_grads["g"][n][0], ref["g_grads"][n][0])
            for n in ("final_conv2", "upsample_1_conv", "trunk_conv", "rrdb_22_dense3_conv5", "rrdb_11_dense2_conv3", "rrdb_0_dense1_conv1", "initial_conv")}
    print("\ngenerator gradient rel-L2 vs the fp64 oracle on the device's branches:", {n: f"{e:.1e}" for n, e in errs.items()})
    assert errs["final_conv2"] <= 2e-4, errs
    assert max(errs.values()) <= 5e-4, errs
    derrs = {n: rel_l2(tr.last_grads["d"][n][0], rk) for n, (rk, rb) in ref["d_grads"].items()}
    print("discriminator gradient rel-L2:", {n: f"{e:.1e}" for n, e in derrs.items()})
    assert max(derrs.values()) <= 5e-4, derrs
    # and the premise, measured: with its OWN branches the fp64 oracle differs from the device in a handful of activations
    ref_own = OT.esrgan_train_step_ref(gw, dw, u0, vw, lr[:2], hr[:2], scale, nb, attention=True, dy_override=dy)
    own = rel_l2(tr.last_grads["g"]["initial_conv"][0], ref_own["g_grads"]["initial_conv"][0])
    print(f"initial_conv gradient against the oracle on its own branches: {own:.1e}")
    assert own <= 5e-3


def test_esrgan_fit_wrapper(ctx, tmp_path):
    """ESRGAN.fit (ESRGAN_model.py:535-779): arrays in [0,1] -> shuffled batches in [-1,1] -> _train_step per batch; the returned
    record is the last epoch's; the first step of the first epoch is checked against the oracle's step on the same batch."""
    from PIL import Image
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    m = ESRGAN(compute_dtype="f32")
    m.setup_model(scale_factor=2, growth_channels=8, num_rrdb_blocks=1, use_attention=True)
    with pytest.raises(ValueError):
        m.fit()
    with pytest.raises(ValueError):
        m.fit(train_dataset=[(np.zeros((1, 8, 8, 3)), np.zeros((1, 16, 16, 3)))])            # steps_per_epoch is mandatory here
    _, dw, vw = _gan_setup(2, 1, 8)
    m.set_loss_network_weights(discriminator=dw, vgg19=vw)
    gw0 = {n: (k.copy(), b.copy()) for n, (k, b) in m.weights.items()}
    rng = np.random.default_rng(12)
    X, Y = rng.uniform(0, 1, (6, 12, 12, 3)).astype(np.float32), rng.uniform(0, 1, (6, 24, 24, 3)).astype(np.float32)
    Xv, Yv = rng.uniform(0, 1, (3, 12, 12, 3)).astype(np.float32), rng.uniform(0, 1, (3, 24, 24, 3)).astype(np.float32)
    save_dir = str(tmp_path / "grids")
    # one epoch first, so that the first step is observable ...
    losses, tt, mt = m.fit(X, Y, X_val=Xv, Y_val=Yv, epochs=1, batch_size=4, save_dir=save_dir, shuffle_seed=42)
    assert m.trained and len(losses["g_loss"]) == 2 == len(losses["d_loss"]) == len(losses["psnr"]) == len(losses["ssim"])
    order = np.random.default_rng(42).permutation(6)
    tr = m._trainer
    u0 = None                                            # the trainer drew its u with seed 0: rebuild it the same way
    from sr355 import gan_train as GT
    u0 = GT.ESRGANTrainer(ctx, gw0, dw, vw, 2, 1, u_seed=0).u
    ref = OT.esrgan_train_step_ref(gw0, dw, u0, vw, X[order[:4]] * 2 - 1, Y[order[:4]] * 2 - 1, 2, 1, attention=True)
    assert abs(losses["g_loss"][0] - ref["losses"]["g_loss"]) <= 2e-4 * abs(ref["losses"]["g_loss"])
    assert abs(losses["d_loss"][0] - ref["losses"]["d_loss"]) <= 2e-4
    assert losses["g_lr"] == [float(np.float32(1e-4))] * 2 and losses["d_lr"] == [float(np.float32(1e-5))] * 2
    for k in ("val_psnr", "val_ssim", "val_g_loss"):
        assert isinstance(losses[k], float) and np.isfinite(losses[k])
    # validation g_loss is the evaluate() formula on the validation batches with the trained networks
    ev = m.evaluate([(Xv * 2 - 1, Yv * 2 - 1)])
    assert abs(ev["avg_g_loss"] - losses["val_g_loss"]) <= 1e-3 * abs(losses["val_g_loss"])
    assert abs(ev["avg_psnr"] - losses["val_psnr"]) <= 1e-3
    grid = np.asarray(Image.open(f"{save_dir}/epoch_001_sr_grid.png"))
    assert grid.shape == (5 * 24, 5 * 24, 3) and grid[:24, :72].any() and not grid[24:].any()     # 3 validation previews, the rest black
    assert len(tt.epoch_times_sec) == 1 and tt.mean_time_value() > 0 and len(mt.gpu_peak_mb) == 1 and mt.as_dict()["gpu_peak_mb"] > 0
    # ... then two more through an external, already normalised dataset: the optimiser state carries on (step counter 2 -> 6)
    ds = [(X[:3] * 2 - 1, Y[:3] * 2 - 1), (X[3:] * 2 - 1, Y[3:] * 2 - 1)]
    losses2, tt2, _ = m.fit(train_dataset=ds, epochs=2, steps_per_epoch=2, normalize=False)
    assert tr is m._trainer and tr.step == 6 and len(tt2.epoch_times_sec) == 2
    assert losses2["val_g_loss"] == [] and len(losses2["g_loss"]) == 2
    moved = max(float(np.abs(m.weights[n][0] - gw0[n][0]).max()) for n in gw0)
    assert 1e-4 < moved < 1e-2                           # six Adam steps of 1e-4
    gpath = m.save(str(tmp_path), "t1")                  # generator and (trained) discriminator, the reference's two files
    dpath = gpath.replace("ESRGAN_generator", "ESRGAN_discriminator")
    assert os.path.exists(dpath)
    m3 = ESRGAN(compute_dtype="f32")
    m3.setup_model(scale_factor=2, from_trained=True, generator_pretrained_path=gpath, discriminator_pretrained_path=dpath, use_attention=True)
    m3._ensure_loss_networks()
    assert all(np.array_equal(m3.d_weights[n][0], tr.dw[n][0]) for n in tr.dw) and all(np.array_equal(m3.weights[n][0], tr.gw[n][0]) for n in tr.gw)
    sr = m.generate(X[:2] * 2 - 1)                       # the inference model carries the trained weights
    t = GT.Tape(ctx, tr.gw, wgrad=False)
    want = GT.generator_forward(t, GT.Var(ctx.to_device(X[:2] * 2 - 1), need=False), 2, 1, True).v.cpu().numpy()
    assert rel_l2(np.asarray(sr.cpu() if isinstance(sr, torch.Tensor) else sr), want) <= 1e-5


@pytest.mark.parametrize("case", [(3, 64, 32, "relu", 1), (3, 3, 64, "linear", 1), (3, 64, 3, "tanh", 1), (1, 64, 8, "linear", 1), (5, 32, 3, "linear", 1),
                                  (9, 3, 96, "relu", 1), (3, 64, 256, "lrelu", 2), (1, 96, 32, "relu", 1)])
def test_conv2d_dev_matches_host_packed_conv(ctx, case):
    """sr_conv2d_dev (device fp32 weights, packed into MFMA fragment order by a device kernel) against sr_conv2d (host packing) for
    every weight layout the fp32 path has -- wide, thin (<= 4 input channels), few-cout -- and the rot flag against the explicitly
    rotated, channel-swapped kernel: identical bytes, since both run the same conv kernel on the same packed image."""
    K, cin, cout, act, r = case
    rng = np.random.default_rng(K * 1000 + cin + cout)
    x = ctx.to_device(rng.standard_normal((2, 13, 11, cin)).astype(np.float32))
    k = (rng.standard_normal((K, K, cin, cout)) / np.sqrt(K * K * cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
    kd, bd = ctx.to_device(k), ctx.to_device(b)
    want = ctx.conv2d(x, k, b, act=act, d2s=r)
    got = ctx.conv2d_dev(x, kd, bd, cout, act=act, d2s=r)
    assert torch.equal(got, want)
    if K != 9:                                           # the input-gradient conv of that layer (9x9 wide is not built: SRCNN's first layer needs none)
        dy = ctx.to_device(rng.standard_normal((2, 13, 11, cout)).astype(np.float32))
        want = ctx.conv2d(dy, T._rot(k), None)
        got = ctx.conv2d_dev(dy, kd, None, cin, rot=True)
        assert got.shape == (2, 13, 11, cin) and torch.equal(got, want)
    with pytest.raises(ValueError):
        ctx.conv2d_dev(x, kd, bd, cout + 1)


def test_conv_prepack_is_the_per_use_pack(ctx):
    """sr_conv_prepack (round 4): the weights of a list of conv uses packed by one launch.  Every layout of the fp32 path, forward and input-gradient use, plain and
    channel-range entry points: the same bytes as a call that packs for itself.  The cache really is what the call reads (a weight changed behind the library's
    back is NOT seen until the list is packed again or forgotten -- the documented contract, which the trainer honours by packing at the start and forgetting at
    the end of every step)."""
    rng = np.random.default_rng(77)
    cases = [(3, 64, 32), (3, 3, 64), (3, 64, 3), (1, 64, 8), (5, 32, 3), (3, 96, 32), (3, 160, 32), (1, 32, 64), (3, 64, 256)]
    ws = []
    for K, cin, cout in cases:
        k = ctx.to_device((rng.standard_normal((K, K, cin, cout)) / np.sqrt(K * K * cin)).astype(np.float32))
        b = ctx.to_device(rng.uniform(-0.1, 0.1, cout).astype(np.float32))
        ws.append((k, b))

    def run():
        out = []
        for (K, cin, cout), (k, b) in zip(cases, ws):
            x = ctx.to_device(np.random.default_rng(cin).standard_normal((2, 13, 11, cin)).astype(np.float32))
            dy = ctx.to_device(np.random.default_rng(cout).standard_normal((2, 13, 11, cout)).astype(np.float32))
            out.append(ctx.conv2d_dev(x, k, b, cout, act="relu"))
            out.append(ctx.conv2d_dev(dy, k, None, cin, rot=True))
            if cin % 16 == 0 and cout % 32 == 0:
                buf = ctx.to_device(np.random.default_rng(cin + 1).standard_normal((2, 13, 11, cin + 32)).astype(np.float32))
                yb = torch.zeros((2, 13, 11, cout + 16), dtype=torch.float32, device=buf.device)
                ctx.conv2d_dev_view(buf, 16, cin, k, b, cout, yb, 16, act="lrelu")
                out.append(yb)
        return out

    base = run()
    uses = [u for k, b in ws for u in ((k, b, False), (k, None, True))]
    pl = ctx.pack_list(uses)
    try:
        ctx.conv_prepack(pl)
        for a, c in zip(base, run()):
            assert torch.equal(a, c)
        ctx.conv_prepack(pl)                                     # the same list again: the table on the device is reused
        for a, c in zip(base, run()):
            assert torch.equal(a, c)
        ws[0][0].mul_(2.0)                                       # behind the library's back
        stale = run()
        assert torch.equal(stale[0], base[0])                    # ... the pack made before the change is what the conv multiplies with
        ctx.conv_prepack(pl)
        fresh = run()
        assert not torch.equal(fresh[0], base[0])                # packed again: the new weights (checked against self-packing calls just below)
        ctx.conv_prepack(None)
        for a, c in zip(fresh, run()):                           # forgotten: every call packs for itself again, from the current weights
            assert torch.equal(a, c)
        ctx.conv_prepack(ctx.pack_list(uses[:3]))                # a shorter list: the other uses pack for themselves
        for a, c in zip(fresh, run()):
            assert torch.equal(a, c)
    finally:
        ctx.conv_prepack(None)
        ws[0][0].mul_(0.5)


def test_vgg16_fit_frozen_base(ctx, tmp_path):
    """FineTunedVGG16.fit (VGG16_model.py:111-157) with the default frozen base: conv base on the device per batch, the two Dense
    layers trained on the host.  Without augmentation and dropout the run is deterministic: history and final head against the
    torch-autograd oracle fed the oracle's own GAP features; with augmentation + dropout: runs, is reproducible from its seed,
    changes only the head, and evaluate() / the saved checkpoint agree with it."""
    from SRModels.defect_detection_models.VGG16_model import FineTunedVGG16
    rng = np.random.default_rng(21)
    X, y = rng.uniform(0, 1, (20, 32, 32, 3)).astype(np.float32), rng.integers(0, 2, 20)
    Xv, yv = rng.uniform(0, 1, (8, 32, 32, 3)).astype(np.float32), rng.integers(0, 2, 8)
    m = FineTunedVGG16()
    with pytest.raises(ValueError):
        m.fit(X, y, Xv, yv)
    m.setup_model(input_shape=(32, 32, 3), num_classes=2, dropout_rate=0.0, learning_rate=1e-3)
    w0 = {n: (k.copy(), b.copy()) for n, (k, b) in m.weights.items()}
    with pytest.raises(RuntimeError):
        m.evaluate(Xv, yv)
    hist = m.fit(X, y, Xv, yv, batch_size=8, epochs=3, use_augmentation=False, seed=5)
    assert m.trained and len(hist.history["loss"]) == 3
    # the oracle: same batches (same permutations), its own fp64 features
    feats = lambda a: M.vgg16_features(a, w0, dtype=np.float64).mean(axis=(1, 2))
    prng, batches = np.random.default_rng(5), []
    for _ in range(3):
        order = prng.permutation(20)
        batches.append([(feats(X[order[i:i + 8]]), y[order[i:i + 8]]) for i in range(0, 20, 8)])
    head, rh = OT.vgg16_head_fit_ref(batches, feats(Xv), yv, w0, lr=1e-3)
    for k in ("loss", "val_loss"):
        assert np.allclose(hist.history[k], rh[k], rtol=2e-4, atol=1e-6), (k, hist.history[k], rh[k])
    assert hist.history["accuracy"] == pytest.approx(rh["accuracy"]) and hist.history["val_accuracy"] == pytest.approx(rh["val_accuracy"])
    for n in ("dense", "predictions"):
        assert rel_l2(m.weights[n][0] - w0[n][0], head[n][0] - w0[n][0]) <= 5e-3, n       # three epochs of Adam steps ~ lr * sign(g)
    assert all(np.array_equal(m.weights[n][0], w0[n][0]) for n in w0 if n.startswith("block"))      # the base is frozen
    loss, acc = m.evaluate(Xv, yv)
    assert abs(loss - hist.history["val_loss"][-1]) <= 1e-4 and acc == pytest.approx(hist.history["val_accuracy"][-1])
    path = m.save(str(tmp_path), "t0")
    m2 = FineTunedVGG16()
    m2.setup_model(input_shape=(32, 32, 3), from_pretrained=True, pretrained_path=path)
    assert np.array_equal(np.asarray(m2.predict(Xv)), np.asarray(m.predict(Xv)))
    # augmentation + dropout: seeded, reproducible
    runs = []
    for _ in range(2):
        a = FineTunedVGG16()
        a.setup_model(input_shape=(32, 32, 3), num_classes=2, dropout_rate=0.2)
        h = a.fit(X, y, Xv, yv, epochs=2, use_augmentation=True, seed=9)
        runs.append((h.history["loss"], a.weights["dense"][0].copy()))
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.isfinite(runs[0][0]).all()
    aug = FineTunedVGG16._augment(X[:4], np.random.default_rng(1))
    assert aug.shape == X[:4].shape and aug.min() >= 0 and aug.max() <= 1 and not np.array_equal(aug, X[:4])
    # the notebook's call (VGG16.ipynb:L161-165): setup_model(train_last_n_layers=6, base_trainable=True).  In the reference the base
    # stays frozen all the same (VGG16_model.py:76-82; 131 842 trainable parameters, VGG16.ipynb:L152): same head-only training,
    # bit for bit the run without the flag
    b = FineTunedVGG16()
    b.setup_model(input_shape=(32, 32, 3), num_classes=2, train_last_n_layers=6, base_trainable=True, dropout_rate=0.0, learning_rate=1e-3)
    assert b.count_params() == 14846530 and b.count_params(trainable_only=True) == 131842 and b.trainable_layers() == ["dense", "predictions"]
    hb = b.fit(X, y, Xv, yv, batch_size=8, epochs=3, use_augmentation=False, seed=5)
    assert hb.history["loss"] == hist.history["loss"] and hb.history["val_accuracy"] == hist.history["val_accuracy"]
    assert all(np.array_equal(b.weights[n][0], w0[n][0]) for n in w0 if n.startswith("block"))
    assert all(np.array_equal(b.weights[n][0], m.weights[n][0]) for n in ("dense", "predictions"))


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", [(2, 64, 64, 16), (3, 70, 45, 8), (1, 129, 33, 37), (2, 576, 32, 576)])        # (batch, M, N, K): tile-exact, ragged, K not a multiple of 16, the trunk attention's beta h
def test_matmul_all_transposes(ctx, shape, ta, tb):
    """sr_matmul (the training attention's products: s = g f^T, o = beta h and their four backward products) on the fp32 matrix core, against fp64 NumPy."""
    B, M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((B, K, M) if ta else (B, M, K)).astype(np.float32)
    b = rng.standard_normal((B, N, K) if tb else (B, K, N)).astype(np.float32)
    got = ctx.matmul(ctx.to_device(a), ctx.to_device(b), trans_a=ta, trans_b=tb, alpha=0.5).cpu().numpy()
    A = a.transpose(0, 2, 1) if ta else a
    Bm = b.transpose(0, 2, 1) if tb else b
    ref = 0.5 * np.matmul(A.astype(np.float64), Bm.astype(np.float64))
    assert got.shape == ref.shape == (B, M, N)
    assert np.abs(got - ref).max() <= 2e-6 * np.sqrt(K) * max(1.0, np.abs(ref).max())
