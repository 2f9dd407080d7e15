"""HIP path (through the C ABI) against the committed golden vectors; fp32 kernels, tolerances as in the kernel tests."""
import os

import numpy as np
import pytest
import torch

from oracle import models as M
from sr355 import Model
from sr355 import pipeline as P
from sr355.weights import init_weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda n: np.load(os.path.join(G, n + ".npz"))


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def test_ops_golden(ctx):
    d = load("conv3x3_d2s")
    x = ctx.to_device(d["x"])
    assert rel_l2(ctx.conv2d(x, d["w"], d["b"], act="relu").cpu().numpy(), d["y"]) <= 1e-5
    assert rel_l2(ctx.conv2d(x, d["w"], d["b"], d2s=2).cpu().numpy(), d["y_d2s"]) <= 1e-5
    d = load("bicubic_64_256")
    assert np.max(np.abs(ctx.bicubic(ctx.to_device(d["x"][None]), 256, 256)[0].cpu().numpy() - d["y"])) <= 2e-6
    assert np.array_equal(ctx.bicubic(ctx.to_device(d["x_u8"][None], torch.uint8), 256, 256)[0].cpu().numpy(), d["y_u8"])
    d = load("metrics")
    for a, b, p, s in ((d["a"], d["b"], d["psnr"], d["ssim"]), (d["a2"], d["b2"], d["psnr2"], d["ssim2"])):
        da, db = ctx.to_device(a), ctx.to_device(b)
        assert np.allclose(ctx.psnr(da, db).cpu().numpy(), p, atol=2e-4) and np.allclose(ctx.ssim(da, db).cpu().numpy(), s, atol=5e-5)
    d = load("plumbing")
    assert np.array_equal(ctx.extract_patches(ctx.to_device(d["img"]), 24, 12).cpu().numpy(), d["patches"])
    rec = ctx.overlap_add(ctx.to_device(d["hr_patches"]), 50, 37, 24, 12, 2).cpu().numpy()
    assert np.max(np.abs(rec - d["recon"])) <= 1e-6
    d = load("self_attention")
    w = init_weights(M.self_attention_layers("sa"), seed=int(d["seed"]))
    y = ctx.self_attention(ctx.to_device(d["x"]), *w["sa_f"], *w["sa_g"], *w["sa_h"], *w["sa_v"]).cpu().numpy()
    assert rel_l2(y, d["y"]) <= 2e-5


def test_models_golden(ctx):
    def run(kind, w, x, **cfg):
        m = Model(kind, compute_dtype="f32", ctx=ctx, **cfg)
        m.set_weights(w)
        return m.forward(ctx.to_device(x)).cpu().numpy()
    d = load("srcnn")
    assert rel_l2(run("srcnn", init_weights(M.srcnn_layers(), seed=int(d["seed"])), d["x"]), d["y"]) <= 1e-5
    for s in (2, 4):
        d = load(f"edsr_x{s}")
        w = init_weights(M.edsr_layers(s, 3, 2, 64), scheme="he_normal", seed=int(d["seed"]))
        assert rel_l2(run("edsr", w, d["x"], scale_factor=s, num_blocks=2, num_filters=64, res_scaling=0.1), d["y"]) <= 1e-5
    for tag in ("nb_cfg", "x4"):
        d = load(f"esrgan_g_{tag}")
        s, g, nb = int(d["scale"]), int(d["growth"]), int(d["num_rrdb"])
        w = init_weights(M.esrgan_g_layers(s, g, nb), seed=int(d["seed"]))
        assert rel_l2(run("esrgan_g", w, d["x"], scale_factor=s, num_blocks=nb, growth_channels=g), d["y"]) <= 2e-5
    d = load("vgg16")
    w = init_weights(M.vgg16_classifier_layers(2), scheme="he_normal", seed=int(d["seed"]))
    assert np.max(np.abs(run("vgg16", w, d["x"], num_classes=2) - d["probs"])) <= 1e-5
    assert P.majority_vote(d["vote_probs"]) == (int(d["vote"][0]), float(d["vote"][1]))
