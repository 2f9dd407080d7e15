"""The oracle reproduces the committed golden vectors (tests/golden/make_golden.py) -- guards the checker itself."""
import os

import numpy as np

from oracle import models as M
from oracle import ops as O
from sr355.weights import init_weights

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda n: np.load(os.path.join(G, n + ".npz"))


def test_ops_golden():
    d = load("conv3x3_d2s")
    assert np.allclose(O.conv2d(d["x"], d["w"], d["b"], act="relu"), d["y"], atol=1e-5)
    assert np.allclose(O.depth_to_space(O.conv2d(d["x"], d["w"], d["b"]), 2), d["y_d2s"], atol=1e-5)
    d = load("bicubic_64_256")
    assert np.allclose(O.bicubic_resize(d["x"], 256, 256), d["y"], atol=1e-6)
    assert np.array_equal(O.bicubic_resize_u8(d["x_u8"], 256, 256), d["y_u8"])
    d = load("metrics")
    assert np.allclose(O.psnr(d["a"], d["b"]), d["psnr"], atol=1e-4) and np.allclose(O.ssim(d["a"], d["b"]), d["ssim"], atol=1e-5)
    assert np.allclose(O.psnr(d["a2"], d["b2"]), d["psnr2"], atol=1e-4) and np.allclose(O.ssim(d["a2"], d["b2"]), d["ssim2"], atol=1e-5)
    d = load("plumbing")
    padded = O.add_padding(d["img"], 24, 12)
    patches, pos = O.extract_patches(padded, 24, 12)
    assert np.array_equal(patches, d["patches"])
    assert np.allclose(O.overlap_add(d["hr_patches"], pos, padded.shape, d["img"].shape[:2], 24, 2), d["recon"], atol=1e-7)


def test_models_golden():
    d = load("srcnn")
    assert np.allclose(M.srcnn_forward(d["x"], init_weights(M.srcnn_layers(), seed=int(d["seed"]))), d["y"], atol=1e-4)
    for s in (2, 4):
        d = load(f"edsr_x{s}")
        w = init_weights(M.edsr_layers(s, 3, 2, 64), scheme="he_normal", seed=int(d["seed"]))
        assert np.allclose(M.edsr_forward(d["x"], w, s, 2, 0.1), d["y"], atol=1e-4)
    for tag in ("nb_cfg", "x4"):
        d = load(f"esrgan_g_{tag}")
        s, g, nb = int(d["scale"]), int(d["growth"]), int(d["num_rrdb"])
        w = init_weights(M.esrgan_g_layers(s, g, nb), seed=int(d["seed"]))
        assert np.allclose(M.esrgan_g_forward(d["x"], w, s, nb), d["y"], atol=1e-4)
    d = load("vgg16")
    assert tuple(O.majority_vote(d["vote_probs"])) == (int(d["vote"][0]), float(d["vote"][1]))
