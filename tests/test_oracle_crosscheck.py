"""Independent second derivations of every oracle primitive (the oracle is self-written, so each piece is checked
against something that shares no code with it: naive fp64 loops, torch's own bicubic, scipy filters, closed forms)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models as M
from oracle import ops as O


@pytest.mark.parametrize("k,stride", [(1, 1), (3, 1), (5, 1), (9, 1), (3, 2)])
def test_conv2d_vs_naive_loops(k, stride):
    rng = np.random.default_rng(k)
    x = rng.standard_normal((2, 9, 8, 5))
    w = rng.standard_normal((k, k, 5, 7))
    b = rng.standard_normal(7)
    a = O.conv2d(x, w, b, stride=stride, dtype=np.float64)
    n = O.conv2d_naive(x, w, b, stride=stride)
    assert a.shape == n.shape and np.allclose(a, n, atol=1e-12)


def test_same_padding_stride2_is_bottom_right():
    # TF SAME, k=3, stride 2, even input: pad 0 before / 1 after (differs from torch padding=1), SURVEY.md A.1
    assert O.same_pads(48, 3, 2) == (0, 1) and O.same_pads(48, 3, 1) == (1, 1) and O.same_pads(7, 3, 2) == (1, 1)


def test_depth_to_space_is_dcr_not_pixel_shuffle():
    x = np.arange(1 * 2 * 2 * 8, dtype=np.float32).reshape(1, 2, 2, 8)
    y = O.depth_to_space(x, 2)
    for h in range(2):
        for w in range(2):
            for i in range(2):
                for j in range(2):
                    for c in range(2):
                        assert y[0, 2 * h + i, 2 * w + j, c] == x[0, h, w, (i * 2 + j) * 2 + c]
    crd = F.pixel_shuffle(torch.from_numpy(x).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).numpy()
    assert not np.array_equal(y, crd)        # torch's CRD order permutes channels


@pytest.mark.parametrize("shape,out", [((64, 64, 3), (256, 256)), ((23, 31, 3), (46, 62)), ((10, 10, 1), (37, 23))])
def test_bicubic_vs_torch(shape, out):
    """torch bicubic(align_corners=False) uses the same a=-0.75 / half-pixel / clamp convention as OpenCV."""
    x = np.random.default_rng(0).uniform(0, 1, shape).astype(np.float32)
    ours = O.bicubic_resize(x, *out)
    t = F.interpolate(torch.from_numpy(x).permute(2, 0, 1)[None], size=out, mode="bicubic", align_corners=False)
    assert np.max(np.abs(ours - t[0].permute(1, 2, 0).numpy())) < 5e-6


def test_bicubic_weights_and_u8():
    w = O.cubic_coeffs(np.float32(0.0))
    assert np.allclose(w, [0, 1, 0, 0], atol=1e-7)
    w = O.cubic_coeffs(np.linspace(0, 1, 11, dtype=np.float32))
    assert np.allclose(w.sum(-1), 1.0, atol=1e-6)
    x = np.random.default_rng(1).integers(0, 256, (16, 16, 3), dtype=np.uint8)
    u = O.bicubic_resize_u8(x, 64, 64)
    f = O.bicubic_resize(x.astype(np.float32), 64, 64)
    assert u.dtype == np.uint8 and np.max(np.abs(u.astype(np.float32) - np.clip(np.rint(f), 0, 255))) <= 1   # +-1 LSB
    assert np.array_equal(O.bicubic_resize_u8(x, 16, 16), x)   # identity resize


def test_psnr_closed_form():
    a = np.zeros((2, 8, 8, 3), np.float32)
    b = np.full((2, 8, 8, 3), 0.1, np.float32)
    assert np.allclose(O.psnr(a, b), 20.0, atol=1e-4)            # mse = 0.01 -> 20 dB
    assert np.isinf(O.psnr(a, a)).all()


def test_ssim_vs_scipy_derivation():
    rng = np.random.default_rng(2)
    a = rng.uniform(0, 1, (3, 24, 30, 3))
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1)
    assert np.allclose(O.ssim(a, b, dtype=np.float64), O.ssim_naive(a, b), atol=1e-10)
    assert np.allclose(O.ssim(a, a, dtype=np.float64), 1.0)
    k = O.gauss_kernel_1d()
    g2 = np.exp(-(np.add.outer((np.arange(11) - 5.0) ** 2, (np.arange(11) - 5.0) ** 2)) / 4.5)
    assert np.allclose(np.outer(k, k), g2 / g2.sum(), atol=1e-15)   # 121-entry softmax == outer product
    with pytest.raises(ValueError):
        O.ssim(a[:, :10], b[:, :10])


def test_add_padding_and_overlap_add():
    assert O.pad_amount(239, 24, 12) == 12 and O.pad_amount(512, 48, 24) == 24 and O.pad_amount(100, 33, 14) == 19
    img = np.arange(5 * 4 * 1, dtype=np.float32).reshape(5, 4, 1)
    p = O.add_padding(img, 4, 2)                       # pad 3 rows (5%2=1 -> (4-1)%2=1 -> max(1,2)=2?) check explicitly below
    assert p.shape[0] == 5 + O.pad_amount(5, 4, 2) and p.shape[1] == 4 + O.pad_amount(4, 4, 2)
    assert np.array_equal(p[5, :4, 0], img[3, :, 0])   # reflect without repeating the edge row
    rng = np.random.default_rng(3)
    im = rng.uniform(0, 1, (37, 29, 3)).astype(np.float32)
    padded = O.add_padding(im, 12, 6)
    patches, pos = O.extract_patches(padded, 12, 6)
    rec = O.overlap_add(patches, pos, padded.shape, im.shape[:2], 12, 1)
    assert np.allclose(rec, im, atol=1e-6)             # identity model -> identity reconstruction


def test_self_attention_explicit():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 3, 4, 64))
    ws = [rng.standard_normal(s) * 0.2 for s in [(1, 1, 64, 8), (8,), (1, 1, 64, 8), (8,), (1, 1, 64, 32), (32,), (1, 1, 32, 64), (64,)]]
    y = O.self_attention(x, *ws, dtype=np.float64)
    X = x.reshape(12, 64)
    f, g, h = X @ ws[0][0, 0] + ws[1], X @ ws[2][0, 0] + ws[3], X @ ws[4][0, 0] + ws[5]
    s = g @ f.T
    beta = np.exp(s - s.max(1, keepdims=True)); beta /= beta.sum(1, keepdims=True)
    ref = X + (beta @ h) @ ws[6][0, 0] + ws[7]
    assert np.allclose(y.reshape(12, 64), ref, atol=1e-10)
    rows = O.attention_rows_streaming(g[:5], f, h)
    assert np.allclose(rows, (beta @ h)[:5], atol=1e-10)


def test_majority_vote_tie_break():
    probs = np.array([[0.9, 0.1], [0.4, 0.6], [0.45, 0.55], [0.8, 0.2]])      # 2 votes each -> higher mean prob wins
    assert O.majority_vote(probs) == (0, pytest.approx(probs[:, 0].mean()))
    assert O.majority_vote(np.array([[0.2, 0.8], [0.3, 0.7], [0.9, 0.1]]))[0] == 1


def test_edsr_and_esrgan_graph_shapes():
    from sr355.weights import init_weights
    w = init_weights(M.edsr_layers(4, 3, 1, 64), scheme="he_normal")
    y = M.edsr_forward(np.zeros((1, 6, 5, 3), np.float32), w, 4, 1)
    assert y.shape == (1, 24, 20, 3) and y.min() >= 0 and y.max() <= 1
    w = init_weights(M.esrgan_g_layers(4, 8, 1))
    y = M.esrgan_g_forward(np.zeros((1, 6, 5, 3), np.float32), w, 4, 1)
    assert y.shape == (1, 24, 20, 3) and np.abs(y).max() <= 1


def test_u8_area_shrink_and_nearest_against_naive_loops():
    """Round 4: uint8 INTER_AREA shrinking and INTER_NEAREST (cv2.resize; classic_algorithms.py:15-17, loading_methods.py:146-147) against an
    independent per-pixel derivation: exact rational cell means for whole-number factors, float32 tap loops otherwise."""
    from fractions import Fraction
    rng = np.random.default_rng(12)
    a = rng.integers(0, 256, (12, 18, 3), dtype=np.uint8)
    # whole-number factors: 3 x 3 cells -> round-half-even of sum * float32(1/9); 2 x 2 cells -> (sum + 2) >> 2
    got = O.cv_resize_u8(a, 4, 6, O.INTER_AREA)
    for y in range(4):
        for x in range(6):
            for c in range(3):
                s = int(a[3 * y:3 * y + 3, 3 * x:3 * x + 3, c].astype(np.int64).sum())
                assert abs(int(got[y, x, c]) - float(Fraction(s, 9))) <= 0.5 + 1e-4
    got = O.cv_resize_u8(a, 6, 9, O.INTER_AREA)
    s = a.astype(np.int64).reshape(6, 2, 9, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(got, (s + 2) // 4)
    assert np.array_equal(O.cv_resize_u8(a, 6, 9, O.INTER_LINEAR), got)            # bilinear halving is the box mean
    two = a[:, :, :2].copy()                                                        # two channels: the float product, ties to even
    got2 = O.cv_resize_u8(two, 6, 9, O.INTER_AREA)
    s2 = two.astype(np.int64).reshape(6, 2, 9, 2, 2).sum(axis=(1, 3))
    assert np.array_equal(got2, np.rint(s2 * 0.25).astype(np.uint8)) and np.any(got2 != (s2 + 2) // 4)
    # no whole-number factor: within half a grey level of the exact area integral, and monotone under a constant image
    got = O.cv_resize_u8(a, 5, 7, O.INTER_AREA)
    f = O.cv_resize(a.astype(np.float32), 5, 7, O.INTER_AREA)
    assert np.array_equal(got, np.clip(np.rint(f), 0, 255).astype(np.uint8))
    assert np.array_equal(O.cv_resize_u8(np.full((12, 18, 3), 200, np.uint8), 5, 7, O.INTER_AREA), np.full((5, 7, 3), 200, np.uint8))
    # nearest: floor(d * src / dst), clamped
    n = O.cv_resize_u8(a, 24, 27, O.INTER_NEAREST)
    for y in (0, 5, 23):
        for x in (0, 13, 26):
            assert np.array_equal(n[y, x], a[min(int(np.floor(y * (1 / (24 / 12)))), 11), min(int(np.floor(x * (1 / (27 / 18)))), 17)])
    assert np.array_equal(O.cv_resize(a.astype(np.float32), 5, 7, O.INTER_NEAREST), O.cv_resize_u8(a, 5, 7, O.INTER_NEAREST).astype(np.float32))
