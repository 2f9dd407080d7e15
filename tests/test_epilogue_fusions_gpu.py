"""Work computed in the epilogue of the kernel that holds its operands (csrc/conv_rows.hip), each against the path that runs the layers as separate
kernels (sr_debug_set_fused without the bit) and against the CPU oracle:
  * the generator's last two convs as one kernel (rows_fuse2 + rgbtail_finish_kernel): final_conv2 (64 -> image channels, tanh; ESRGAN_model.py:341)
    as a 1x1 conv to 9 x C "tap channels" inside final_conv1's epilogue followed by a shifted sum over each tile's halo'd region -- tile-exact, ragged
    and multi-tile images, one and three image channels, fp32 and bf16 caller tensors, and an exact-integer case in which both device paths and the
    oracle must agree to the last bit;
  * SelfAttention's f / g / h projections inside the conv that produces the layer's input (rows_epilogue_proj);
  * the VGG16 classifier's block 5 on batches packed into one tall image (CellGrid) and every block's MaxPooling2D inside the conv in front of it
    (rows_pool2): bit for bit the plain path."""
import numpy as np
import pytest
import torch

from oracle import models as M
from sr355 import Model
from sr355.weights import bf16_rounded, init_weights, round_to_bf16

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused_ctx(ctx):
    yield ctx
    ctx.set_fused(ctx.FUSED_ALL, 0)


def kernels_of(ctx, fn):
    ctx.profile_begin()
    y = fn()
    torch.cuda.synchronize()
    return y, {r["kernel"] for r in ctx.profile_end()}


# (B, LR height, LR width, scale, channels): the output tile of the 64-cout kernel is 12 x 16
CASES = [
    (1, 6, 8, 2, 3),        # 12 x 16: exactly one tile -- only the tile's own sums
    (2, 7, 9, 2, 3),        # 14 x 18: 2 x 2 tiles, the second row / column of tiles two pixels deep (ragged: masked source pixels)
    (3, 12, 16, 2, 3),      # 24 x 32: 2 x 2 full tiles, every interior border crossed by the 3x3 window
    (2, 24, 24, 4, 3),      # 96 x 96: 8 x 6 tiles, two up-sampling stages (the bench's layer shapes at half size)
    (1, 5, 13, 2, 1),       # one image channel: 9 tap channels
    (2, 3, 3, 4, 3),        # 12 x 12: one ragged tile column
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("io", ["f32", "bf16"])
def test_rgb_tail_matches_two_kernel_path_and_oracle(fused_ctx, case, io):
    ctx = fused_ctx
    B, H, W, s, C = case
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=s, channels=C, num_blocks=1, growth_channels=32, use_attention=False, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), seed=5100 + H))
    m.set_weights(w)
    x = round_to_bf16(np.random.default_rng(17 * H + W).uniform(-1, 1, (B, H, W, C)).astype(np.float32))
    xd = ctx.to_device(x, torch.float32 if io == "f32" else torch.bfloat16)
    ctx.set_fused(3, 0)
    y0, k0 = kernels_of(ctx, lambda: m.forward(xd))
    ctx.set_fused(7, 0)
    y1, k1 = kernels_of(ctx, lambda: m.forward(xd))
    assert not any("rgbtail" in k for k in k0)
    assert any(k.startswith("conv_rows_rgbtail") for k in k1) and "rgbtail_finish" in k1, k1      # the fused path is the one that ran
    assert torch.equal(y1, m.forward(xd))                                                         # deterministic: fixed order of the partial sums
    a, b = y0.float().cpu().numpy(), y1.float().cpu().numpy()
    assert a.shape == (B, H * s, W * s, C)
    # Round 4: the fused kernel feeds final_conv1's activation -- which it never stores -- into the 1x1 product as a bf16 hi + lo pair; the
    # two-kernel path stores and re-reads ONE bf16 value.  Each path is held to the oracle restating ITS roundings (fused_tail), to the
    # accumulation-order residue the other one shows (tanh output in [-1, 1]); between them lies one bf16 rounding of a 64-channel activation.
    ref0 = M.esrgan_g_forward(x, w, s, 1, dtype=np.float64, attention=False, bf16_storage=True, bf16_output=io == "bf16", fused_tail=False)
    ref1 = M.esrgan_g_forward(x, w, s, 1, dtype=np.float64, attention=False, bf16_storage=True, bf16_output=io == "bf16", fused_tail=True)
    e0, e1 = np.abs(a - ref0).max(), np.abs(b - ref1).max()
    assert e1 <= max(2.0 * e0, 1e-5) + (2.0 ** -8 if io == "bf16" else 0.0), (float(e0), float(e1))
    full = M.esrgan_g_forward(x, w, s, 1, dtype=np.float64, attention=False, bf16_storage=True, bf16_output=False, fused_tail=False)
    keep = M.esrgan_g_forward(x, w, s, 1, dtype=np.float64, attention=False, bf16_storage=True, bf16_output=False, fused_tail=True)
    assert np.abs(a - b).max() <= 2.0 * np.abs(full - keep).max() + 1e-5 + (2.0 ** -7 if io == "bf16" else 0.0)     # no further apart than that one rounding
    # a tap on final_conv1 needs that conv's output in memory: the pair then runs as two kernels and the tap holds the activation
    parts = {}
    M.esrgan_g_forward(x, w, s, 1, dtype=np.float64, attention=False, bf16_storage=True, parts=parts, fused_tail=False)
    y2, taps = m.forward_with_taps(xd, ["final_conv1"])
    assert torch.equal(y2, y0)
    t = taps["final_conv1"].cpu().numpy()
    assert np.abs(t - parts["final_conv1"]).max() <= 2.0 ** -7 * max(1.0, np.abs(parts["final_conv1"]).max())


def test_rgb_tail_exact_integers(fused_ctx):
    """Index arithmetic, exactly: with small non-negative integers everywhere (so LeakyReLU is the identity and every bf16 rounding is
    exact) the pre-activation of final_conv2 is an exactly representable integer whatever the order of the additions, so the fused
    kernel, the two-kernel path and the oracle must produce the SAME tanh values -- over ragged tiles, image borders and a batch."""
    ctx = fused_ctx
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, channels=3, num_blocks=0, growth_channels=32, use_attention=False, ctx=ctx)
    rng = np.random.default_rng(11)
    w = {}
    for name, shape in m.layer_shapes():
        k = np.zeros(shape, np.float32)
        b = np.zeros(shape[-1], np.float32)
        if name == "initial_conv":
            k[1, 1] = rng.integers(0, 2, size=shape[2:])
        elif name in ("trunk_conv", "upsample_0_conv"):
            nz = rng.random(shape) < 0.004
            k[nz] = 1.0
        elif name == "final_conv1":
            nz = rng.random(shape) < 0.01
            k[nz] = rng.integers(-1, 2, size=int(nz.sum()))
            b = rng.integers(0, 2, size=shape[-1]).astype(np.float32)
        elif name == "final_conv2":
            nz = rng.random(shape) < 0.05
            k[nz] = rng.integers(-1, 2, size=int(nz.sum())) * 2.0 ** -7        # a power of two: still exact, and tanh stays unsaturated (|pre-activation| < 1.4)
        w[name] = (k, b)
    m.set_weights(w)
    x = rng.integers(0, 3, size=(3, 13, 21, 3)).astype(np.float32)               # 26 x 42 output: 3 x 3 tiles, ragged both ways
    xd = ctx.to_device(x, torch.float32)
    parts = {}
    ref = M.esrgan_g_forward(x, w, 2, 0, dtype=np.float64, attention=False, bf16_storage=True, bf16_output=False, parts=parts)
    h = parts["final_conv1"]
    assert np.array_equal(h, np.round(h)) and 0 < h.max() < 256, float(h.max())   # the premise: exact in bf16, and not all zeros
    ctx.set_fused(3, 0)
    y0 = m.forward(xd).cpu().numpy()
    ctx.set_fused(7, 0)
    y1, k1 = kernels_of(ctx, lambda: m.forward(xd))
    assert any(k.startswith("conv_rows_rgbtail") for k in k1)
    y1 = y1.cpu().numpy()
    assert len(np.unique(y1)) > 100                                                 # a non-trivial image
    assert np.array_equal(y0, y1), (float(np.abs(y0 - y1).max()), np.argwhere(y0 != y1)[:5])
    assert np.abs(y1 - ref).max() <= 1e-6, float(np.abs(y1 - ref).max())         # tanhf against the fp64 tanh of the same integer


def test_rgb_tail_falls_back_when_the_workspace_is_too_small(fused_ctx):
    """The partial sums live in the buffer final_conv1's output would have taken; an image so small that one tile's 3 KiB of sums exceed
    its 64-channel pixels (plus the workspaces' 4 KiB of slack) runs as two kernels: two 2 x 2 images = 6048 bytes of sums against 5120."""
    ctx = fused_ctx
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, channels=3, num_blocks=1, growth_channels=32, use_attention=False, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), seed=9))
    m.set_weights(w)
    x = round_to_bf16(np.random.default_rng(2).uniform(-1, 1, (2, 1, 1, 3)).astype(np.float32))
    xd = ctx.to_device(x, torch.float32)
    y, ks = kernels_of(ctx, lambda: m.forward(xd))
    assert not any("rgbtail" in k for k in ks)
    ref = M.esrgan_g_forward(x, w, 2, 1, dtype=np.float64, attention=False, bf16_storage=True, bf16_output=False, fused_tail=False)
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-2


# ---- SelfAttention's f / g / h projections in the epilogue of the conv that produces the layer's input (rows_epilogue_proj) ----------

PROJ_CASES = [
    (2, 5, 7, 2),          # ragged tiles at both resolutions; the trunk conv carries its skip, the up-sampling conv depth_to_space
    (1, 12, 16, 2),        # exactly one tile at the trunk resolution, 2 x 2 tiles after the shuffle
    (2, 13, 20, 4),        # two up-sampling stages: attention after the first only
]


@pytest.mark.parametrize("case", PROJ_CASES)
def test_attention_projections_in_the_producing_conv(fused_ctx, case):
    from sr355.weights import condition_attention
    ctx = fused_ctx
    B, H, W, s = case
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=s, num_blocks=1, growth_channels=32, use_attention=True, ctx=ctx)
    w = bf16_rounded(condition_attention(init_weights(m.layer_shapes(), seed=6200 + H)))
    m.set_weights(w)
    x = round_to_bf16(np.random.default_rng(H * W).uniform(-1, 1, (B, H, W, 3)).astype(np.float32))
    xd = ctx.to_device(x, torch.bfloat16)
    taps = ["self_attention_trunk_f", "self_attention_upsample_0_f", "trunk_conv", "upsample_0_conv"]
    ctx.set_fused(7, 0)
    (y0, t0), k0 = kernels_of(ctx, lambda: m.forward_with_taps(xd, taps))
    ctx.set_fused(15, 0)
    (y1, t1), k1 = kernels_of(ctx, lambda: m.forward_with_taps(xd, taps))
    assert not any(k.startswith("conv_rows_proj") for k in k0) and any(k.startswith("conv_pw<bf16,k1,kg1,nt3>") for k in k0), k0
    assert any(k.startswith("conv_rows_proj") for k in k1) and not any(k.startswith("conv_pw<bf16,k1,kg1,nt3>") for k in k1), k1
    assert torch.equal(y1, m.forward(xd))
    assert torch.equal(t0["trunk_conv"], t1["trunk_conv"])                   # the conv's own output: same arithmetic, same stores
    # (upsample_0_conv sits behind the trunk attention, whose projections may differ by a flipped bf16 rounding: close, not identical)
    a, b = t0["upsample_0_conv"].cpu().numpy(), t1["upsample_0_conv"].cpu().numpy()
    assert np.abs(a - b).max() <= 2.0 ** -6 * max(1.0, np.abs(a).max()), float(np.abs(a - b).max())
    # the trunk attention's 48 channels f | g | h come from bit-identical inputs: same products, another fp32 summation order -- a flipped bf16
    # rounding here and there, no more
    a, b = t0["self_attention_trunk_f"].cpu().numpy(), t1["self_attention_trunk_f"].cpu().numpy()
    assert a.shape[-1] == 48 and a.shape == b.shape
    assert np.all(np.abs(a - b) <= 2.0 ** -7 * np.maximum(np.abs(a), 2.0 ** -10)), float(np.abs(a - b).max())
    assert np.mean(a != b) < 0.02, float(np.mean(a != b))
    # the second attention's inputs already differ by such flips (they sit behind the first): close
    a, b = t0["self_attention_upsample_0_f"].cpu().numpy(), t1["self_attention_upsample_0_f"].cpu().numpy()
    assert a.shape[-1] == 48 and a.shape == b.shape
    assert np.abs(a - b).max() <= 2.0 ** -5 * max(1.0, np.abs(a).max()), float(np.abs(a - b).max())
    ref = M.esrgan_g_forward(x, w, s, 1, dtype=np.float64, attention=True, bf16_storage=True)
    e0 = np.abs(y0.float().cpu().numpy() - ref).max()
    e1 = np.abs(y1.float().cpu().numpy() - ref).max()
    assert e1 <= max(2.0 * e0, 2.0 ** -6), (float(e0), float(e1))


# ---- the VGG16 classifier's block 5 on a batch packed into one tall image (CellGrid, csrc/common.h; VGG16_model.py:57-97) --------------

@pytest.mark.parametrize("patch", [96, 48, 128, 64])      # block-5 images 6 x 6 (two per tile row), 3 x 3 (four), 8 x 8 (one), 4 x 4 (three)
def test_vgg16_block5_packed_batches(fused_ctx, patch):
    """Every output pixel is the same sum in the same order whatever tile it falls into, so the packed path must give the plain path's
    probabilities bit for bit: odd batches (a half-empty cell row), a batch after a larger one (stale cells beside live ones), a tapped
    forward in between (plain layout over the packed buffers, then packed again: the separators are cleared)."""
    ctx = fused_ctx
    m = Model("vgg16", compute_dtype="bf16", num_classes=2, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), scheme="he_normal", seed=4100))
    m.set_weights(w)
    rng = np.random.default_rng(patch)
    xs = [round_to_bf16(rng.uniform(0, 1, (n, patch, patch, 3)).astype(np.float32)) for n in (5, 2, 7, 1)]
    ctx.set_fused(15, 0)
    want = [m.forward(ctx.to_device(x, torch.bfloat16)).clone() for x in xs]
    ctx.set_fused(31, 0)
    got = [m.forward(ctx.to_device(x, torch.bfloat16)).clone() for x in xs]
    for a, b in zip(want, got):
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    # ... and with every block's MaxPooling2D computed in the epilogue of the conv in front of it (max of bf16 values is exact), alone and together with the packing
    for mask in (64, 127):
        ctx.set_fused(mask, 0)
        (y_p, ks) = kernels_of(ctx, lambda: m.forward(ctx.to_device(xs[0], torch.bfloat16)))
        assert any(k.startswith("conv_rows_pool") for k in ks), ks
        assert torch.equal(y_p, want[0]), (mask, float((y_p.float() - want[0].float()).abs().max()))
        assert torch.equal(m.forward(ctx.to_device(xs[2], torch.bfloat16)), want[2]), mask
    ctx.set_fused(31, 0)
    y_t, taps = m.forward_with_taps(ctx.to_device(xs[0], torch.bfloat16), ["block5_conv3"])       # plain layout for this call
    assert torch.equal(y_t, want[0]) and taps["block5_conv3"].shape == (5, patch // 16, patch // 16, 512)
    assert torch.equal(m.forward(ctx.to_device(xs[2], torch.bfloat16)), want[2])                  # packed again
    ref = M.vgg16_classifier_forward(xs[0], w, dtype=np.float64)
    assert np.max(np.abs(got[0].float().cpu().numpy() - ref)) <= 3e-2
