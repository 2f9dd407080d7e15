"""Kernel-level parity: HIP ops called through the C ABI vs the CPU oracle on seeded inputs.

Tolerances: fp32 kernels (exact-fp32 MFMA fma chains) rel-L2 <= 1e-5 against the fp32 oracle
(SURVEY.md 8d); bf16 kernels are compared with the fp32 oracle evaluated on bf16-rounded
weights/inputs, rel-L2 <= 1e-2 per op (one bf16 rounding of the output, fp32 accumulation).
"""
import numpy as np
import pytest
import torch

from oracle import ops as O
from sr355.weights import round_to_bf16

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _dev(ctx, a, dtype):
    return ctx.to_device(np.asarray(a, np.float32), dtype)


CONV_CASES = [
    # B, H,  W,  Cin, Cout, K, act
    (2, 24, 16, 64, 32, 3, "relu"),      # exact tile
    (1, 48, 48, 96, 32, 3, "relu"),      # dense-block conv2
    (2, 33, 33, 3, 96, 9, "relu"),       # SRCNN conv1 (thin, 9x9), ragged tile
    (2, 33, 29, 96, 32, 1, "relu"),      # SRCNN conv2 (1x1)
    (1, 33, 33, 32, 3, 5, "linear"),     # SRCNN conv3 (5x5, Cout=3)
    (1, 20, 37, 3, 64, 3, "linear"),     # RGB head (thin 3x3)
    (1, 17, 9, 192, 64, 3, "linear"),    # dense-block conv5
    (1, 12, 12, 64, 256, 3, "lrelu"),    # upsample conv
    (1, 7, 5, 72, 8, 3, "relu"),         # G=8 growth conv (Cin not a multiple of the chunk)
    (1, 16, 16, 64, 3, 3, "tanh"),       # final conv
    (1, 6, 6, 512, 512, 3, "relu"),      # VGG block5
    (1, 9, 11, 64, 48, 1, "linear"),     # attention projection
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_oracle(ctx, case, dtype):
    B, H, W, Cin, Cout, K, act = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        x, w = round_to_bf16(x), round_to_bf16(w)
    ref = O.conv2d(x, w, b, act=act, dtype=np.float64)
    got = ctx.conv2d(_dev(ctx, x, td), w, b, act=act).float().cpu().numpy()
    assert got.shape == ref.shape
    err = rel_l2(got, ref)
    assert err <= (1e-5 if dtype == "f32" else 1e-2), err


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_conv2d_epilogue_skips_clip(ctx, dtype):
    rng = np.random.default_rng(5)
    B, H, W, Cin, Cout = 2, 26, 18, 64, 64
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    s1 = rng.uniform(0, 1, (B, H, W, Cout)).astype(np.float32)
    s2 = rng.uniform(0, 1, (B, H, W, Cout)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / 24).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        x, w, s1, s2 = (round_to_bf16(a) for a in (x, w, s1, s2))
    ref = np.clip(0.04 * O.conv2d(x, w, b, dtype=np.float64) + s1 + 0.2 * s2, 0, 1)
    got = ctx.conv2d(_dev(ctx, x, td), w, b, alpha=0.04, skip1=_dev(ctx, s1, td), beta1=1.0, skip2=_dev(ctx, s2, td), beta2=0.2,
                     clip01=True).float().cpu().numpy()
    assert rel_l2(got, ref) <= (1e-5 if dtype == "f32" else 1e-2)


@pytest.mark.parametrize("r,cout", [(2, 256), (3, 72), (2, 12)])
def test_conv2d_depth_to_space_dcr(ctx, r, cout):
    """TF depth_to_space is DCR: out[b,h*r+i,w*r+j,c] = in[b,h,w,(i*r+j)*C+c] (not torch pixel_shuffle)."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 10, 13, 64)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, cout)) / 24).astype(np.float32)
    ref = O.depth_to_space(O.conv2d(x, w, None, dtype=np.float64), r)
    got = ctx.conv2d(_dev(ctx, x, torch.float32), w, None, d2s=r).cpu().numpy()
    assert got.shape == ref.shape
    assert rel_l2(got, ref) <= 1e-5
