"""Training rows (SURVEY.md 8f-4 and the conv backward kernels of 8f-1): the wgrad kernel and the dgrad-by-rotated-forward identity
against torch autograd, one SRCNN / EDSR step gradient by gradient, a few Adam steps weight by weight, and `fit` end to end
(Keras callbacks' semantics) -- oracle/train.py is the independent fp64 derivation."""
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
