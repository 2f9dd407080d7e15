"""csrc/conv_stream.hip: the persistent 64-input-channel 3x3 kernel (one cout tile's weights resident in LDS, 24 x 16 tiles streamed through three
halo buffers by loader waves) against the tile kernel it replaces -- every output pixel is the same sum in the same order, so the two must agree bit for
bit -- and against the CPU oracle: plain / ReLU / LeakyReLU epilogues, one and two skips, depth_to_space, 64 / 128 / 256 couts, image sizes whose
tiles are exact, several images per workgroup and more workgroups than tiles."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from sr355.weights import round_to_bf16

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused_ctx(ctx):
    yield ctx
    ctx.set_fused(ctx.FUSED_ALL, 0)


CASES = [
    # B, H, W, Cout, act, skips, d2s
    (3, 24, 16, 64, "linear", 0, 1),       # one tile per image
    (2, 48, 48, 64, "relu", 0, 1),         # 2 x 3 tiles
    (5, 48, 48, 64, "linear", 1, 1),       # EDSR body: alpha * conv + skip
    (2, 48, 32, 64, "linear", 2, 1),       # two skips
    (2, 48, 48, 256, "lrelu", 0, 2),       # up-sampling conv: four cout tiles, depth_to_space
    (1, 96, 96, 128, "relu", 0, 1),        # two cout tiles
    (300, 24, 16, 64, "relu", 0, 1),       # more tiles than workgroups: several tiles per workgroup
]


@pytest.mark.parametrize("case", CASES)
def test_conv_stream_is_the_tile_kernel_bit_for_bit(fused_ctx, case):
    ctx = fused_ctx
    B, H, W, Cout, act, nskip, r = case
    rng = np.random.default_rng(B * 1000 + H + Cout)
    x = round_to_bf16(rng.uniform(-1, 1, (B, H, W, 64)).astype(np.float32))
    w = round_to_bf16((rng.standard_normal((3, 3, 64, Cout)) / np.sqrt(9 * 64)).astype(np.float32))
    b = rng.uniform(-0.05, 0.05, Cout).astype(np.float32)
    s1 = round_to_bf16(rng.uniform(-1, 1, (B, H, W, Cout)).astype(np.float32)) if nskip >= 1 else None
    s2 = round_to_bf16(rng.uniform(-1, 1, (B, H, W, Cout)).astype(np.float32)) if nskip >= 2 else None
    xd = ctx.to_device(x, torch.bfloat16)
    kw = dict(act=act, alpha=0.5 if nskip else 1.0, d2s=r)
    if s1 is not None:
        kw.update(skip1=ctx.to_device(s1, torch.bfloat16), beta1=1.0)
    if s2 is not None:
        kw.update(skip2=ctx.to_device(s2, torch.bfloat16), beta2=0.25)

    def run(mask):
        ctx.set_fused(mask, 0)
        ctx.profile_begin()
        y = ctx.conv2d(xd, w, b, **kw)
        torch.cuda.synchronize()
        return y, {k["kernel"] for k in ctx.profile_end()}

    y0, k0 = run(127)
    y1, k1 = run(255)
    assert any(k.startswith("conv_rows") for k in k0), k0
    # (the persistent kernel carries the vector epilogues with at most one skip; two skips stay on the tile kernel)
    assert any(k.startswith("conv_stream" if nskip < 2 else "conv_rows") for k in k1), k1
    assert torch.equal(y0, y1), float((y0.float() - y1.float()).abs().max())
    ref = O.conv2d(x.astype(np.float64), w, b, act=act, dtype=np.float64) * kw["alpha"]
    if s1 is not None:
        ref = ref + s1
    if s2 is not None:
        ref = ref + 0.25 * s2
    if r > 1:
        ref = O.depth_to_space(ref, r)
    got = y1.float().cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2.0 ** -7 * max(1.0, np.abs(ref).max()), float(np.abs(got - ref).max())
