"""The fused dense-block kernels (csrc/dense_fused.hip: conv2+conv3 and conv4+conv5 of an ESRGAN dense block as one line-buffered
persistent kernel each, ESRGAN_model.py:212-254) against the layer-by-layer path and against the CPU oracle in its bf16-storage
mode.  The fused path needs 48-pixel-wide images and row-blocked buffers (bf16, G = 32); everything else about the shape is free:
any height (the rows of a workgroup's images form one stream with zero separator rows), any batch (several images per
workgroup when the batch exceeds the grid -- forced here with a small grid cap)."""
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


@pytest.fixture()
def fused_ctx(ctx):
    yield ctx
    ctx.set_fused(ctx.FUSED_ALL, 0)


CASES = [
    # B, H, grid cap (0 = one workgroup per CU).  Round 3: a workgroup owns a range of the global row stream (B * (H + 1) rows, >= 24 per
    # workgroup), so the ranges below cut images at many different offsets; the two rows a range recomputes for its neighbours are part of it
    (3, 48, 0),      # the bench shape: 147 stream rows -> 7 ranges of 21 rows, every one starting or ending inside an image
    (7, 48, 2),      # two ranges of 172 rows: separator rows inside a range, a cut in the middle of the fourth image
    (5, 9, 2),       # height not a multiple of the 8-row step; two ranges of 25 rows
    (4, 1, 3),       # one-row images: every second stream row is a separator (one range)
    (2, 17, 0),      # two ranges that meet exactly at an image boundary
    (3, 48, 5),      # five ranges of 30 rows
    (6, 20, 4),      # four ranges of 32 rows over 21-row image periods
    (9, 8, 4),       # height == step
]


_ORACLE_OF_CASE = {}


def oracle_of_case(case, x, w, nb):
    """The CPU oracle's output and stages for a case: the same for every fusion mask, computed once (it is most of this file's run time)."""
    if case not in _ORACLE_OF_CASE:
        parts = {}
        ref = M.esrgan_g_forward(x, w, 2, nb, dtype=np.float64, attention=False, bf16_storage=True, parts=parts)
        _ORACLE_OF_CASE[case] = (ref, {k: v for k, v in parts.items() if k.startswith("rrdb_")})
    return _ORACLE_OF_CASE[case]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mask", [1, 2, 3, 32, 35])       # bit 5: conv1 of every dense block on the streaming kernel
def test_fused_pairs_match_layer_by_layer_and_oracle(fused_ctx, case, mask):
    ctx = fused_ctx
    B, H, cap = case
    nb = 2                                                                    # dense3's tail carries the second skip (rrdb input)
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=nb, growth_channels=32, use_attention=False, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), seed=3400))
    m.set_weights(w)
    x = round_to_bf16(np.random.default_rng(B * 100 + H).uniform(-1, 1, (B, H, 48, 3)).astype(np.float32))
    xd = ctx.to_device(x, torch.bfloat16)
    names = [f"rrdb_{b}_dense{d}_conv5" for b in range(nb) for d in (1, 2, 3)] + ["rrdb_0_dense1_conv3", "rrdb_1_dense2_conv2", "rrdb_0_dense1_conv1",
                                                                                 "rrdb_1_dense3_conv1"]
    ctx.set_fused(0, 0)
    y0, t0 = m.forward_with_taps(xd, names)
    ctx.set_fused(mask, cap)
    y1, t1 = m.forward_with_taps(xd, names)
    y1b = m.forward(xd)
    assert torch.equal(y1, y1b)                                               # re-run on the same workspaces: same image
    ref, parts = oracle_of_case(case, x, w, nb)
    # stage by stage: the fused path is as close to the oracle as the layer-by-layer path is (same roundings, other summation order)
    for n in names:
        a, b = t0[n].cpu().numpy(), t1[n].cpu().numpy()
        assert rel_l2(b, a) <= 4e-3, (n, rel_l2(b, a))
    for bi in range(nb):
        got = t1[f"rrdb_{bi}_dense3_conv5"].cpu().numpy()
        assert rel_l2(got, parts[f"rrdb_{bi}"]) <= 4e-3, (bi, rel_l2(got, parts[f"rrdb_{bi}"]))
    e0, e1 = rel_l2(y0.float().cpu().numpy(), ref), rel_l2(y1.float().cpu().numpy(), ref)
    assert e1 <= 5e-3 and e1 <= 2.0 * e0 + 1e-3, (e0, e1)


def test_fused_tail_exact_integers(fused_ctx):
    """Exact check of the index arithmetic (fragment packing order, ring rows, separator rows, channel offsets of the stores, the
    folded skip): small-integer weights and inputs make every partial sum exact in bf16/fp32, so fused and layer-by-layer outputs
    must be IDENTICAL and equal to the oracle, not just close."""
    ctx = fused_ctx
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=1, growth_channels=32, use_attention=False, ctx=ctx)
    rng = np.random.default_rng(7)
    w = {}
    for name, shape in m.layer_shapes():
        k = np.zeros(shape, np.float32)
        if "dense" in name:                                                   # sparse small integers: sums stay far below 2^8
            nz = rng.random(shape) < 0.02
            k[nz] = rng.integers(-2, 3, size=int(nz.sum()))
            b = rng.integers(-1, 2, size=shape[-1]).astype(np.float32)
        elif name == "initial_conv":
            k[1, 1, :, :] = rng.integers(-1, 2, size=shape[2:])
            b = rng.integers(0, 3, size=shape[-1]).astype(np.float32)
        else:
            k = (rng.standard_normal(shape) * 0.01).astype(np.float32)
            b = np.zeros(shape[-1], np.float32)
        w[name] = (k, b)
    w = bf16_rounded(w)
    m.set_weights(w)
    x = rng.integers(-1, 2, size=(5, 11, 48, 3)).astype(np.float32)
    xd = ctx.to_device(x, torch.bfloat16)
    names = ["rrdb_0_dense1_conv2", "rrdb_0_dense1_conv3", "rrdb_0_dense1_conv5", "rrdb_0_dense2_conv5", "rrdb_0_dense3_conv5"]
    ctx.set_fused(0, 0)
    _, t0 = m.forward_with_taps(xd, names + ["rrdb_0_dense1_conv1", "rrdb_0_dense2_conv1"])
    ctx.set_fused(35, 2)
    _, t1 = m.forward_with_taps(xd, names + ["rrdb_0_dense1_conv1", "rrdb_0_dense2_conv1"])
    for n in ("rrdb_0_dense1_conv1", "rrdb_0_dense2_conv1"):                  # the streaming conv1 kernel: integers again for dense1, same roundings for dense2
        a, b = t0[n].cpu().numpy(), t1[n].cpu().numpy()
        assert (np.array_equal(a, b) if "dense1" in n else np.abs(a - b).max() <= 2.0 ** -7 * max(1.0, np.abs(a).max())), (n, float(np.abs(a - b).max()))
    # conv outputs are integers; x + 0.2*conv5 is not, but both paths round the same fp32 value
    for n in names[:2]:
        a, b = t0[n].cpu().numpy(), t1[n].cpu().numpy()
        assert np.array_equal(a, b), (n, float(np.abs(a - b).max()), np.argwhere(a != b)[:5])
    for n in names[2:]:
        a, b = t0[n].cpu().numpy(), t1[n].cpu().numpy()
        assert np.abs(a - b).max() <= 2.0 ** -7 * max(1.0, np.abs(a).max()), (n, float(np.abs(a - b).max()))
    # ... and equal to the ORACLE (VERDICT r2 weak #1d): the growth convs of the first dense block restated layer by layer in fp64 -- every
    # value is a small integer, so the device's bf16 tensors must hold exactly these numbers
    from oracle import ops as OO
    x0 = OO.conv2d(x.astype(np.float64), *w["initial_conv"], dtype=np.float64)
    feats = [x0]
    for c in (1, 2, 3):
        feats.append(OO.conv2d(np.concatenate(feats, axis=-1), *w[f"rrdb_0_dense1_conv{c}"], act="relu", dtype=np.float64))
    assert np.abs(feats[2]).max() < 256 and np.array_equal(feats[2], np.round(feats[2]))
    # (conv3's exact integer sums may exceed 2^8: the device stores them rounded to bf16 once, and so does the reference value here)
    for n, ref in (("rrdb_0_dense1_conv2", feats[2]), ("rrdb_0_dense1_conv3", round_to_bf16(feats[3].astype(np.float32)).astype(np.float64))):
        assert np.array_equal(t1[n].float().cpu().numpy().astype(np.float64), ref), n
        assert np.array_equal(t0[n].float().cpu().numpy().astype(np.float64), ref), n


# ---- 24-pixel-wide patches, two per 48-pixel row (round 4: api.hip pack2, dense_fused.hip SEAM; patch_size_lr = 24 is the reference's own training patch,
# ESRGAN_model.py:858 / constants.py:8) ----

PAIR_CASES = [
    # B, H, grid cap
    (8, 24, 0),      # the patch shape, an even batch
    (7, 24, 2),      # an odd batch: the last row pair's right half is padding
    (2, 24, 0),      # one packed image
    (5, 9, 3),       # height not a multiple of the step, ranges cutting packed images
    (12, 5, 4),
    (3, 1, 0),       # one-row images
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_two_up_packed_24_wide_patches_match_layer_by_layer_and_oracle(fused_ctx, case):
    """The fused dense-block kernels on 24-pixel-wide images: two images side by side per 48-pixel row, columns 23 | 24 an image border.  Against the layer-by-layer
    path (same graph, tile kernels) and the bf16-storage oracle; the seam is where a mistake would show (a conv reading across it mixes two images)."""
    ctx = fused_ctx
    B, H, cap = case
    nb = 2
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=nb, growth_channels=32, use_attention=False, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), seed=3500))
    m.set_weights(w)
    x = round_to_bf16(np.random.default_rng(B * 10 + H).uniform(-1, 1, (B, H, 24, 3)).astype(np.float32))
    xd = ctx.to_device(x, torch.bfloat16)
    dense = ctx.FUSED_ALL & ~35
    ctx.set_fused(dense, 0)
    ctx.profile_begin()
    y0 = m.forward(xd)
    k0 = {r["kernel"] for r in ctx.profile_end()}
    ctx.set_fused(ctx.FUSED_ALL, cap)
    ctx.profile_begin()
    y1 = m.forward(xd)
    k1 = {r["kernel"] for r in ctx.profile_end()}
    assert not any(k.startswith("dense_") for k in k0), k0
    assert {"dense_tail_fused<bf16,conv4+conv5>", "dense_pair_fused<bf16>", "dense_conv1_stream<bf16,64->32>"} <= k1, k1       # the fused kernels ran on the 24-wide batch
    assert not any(k.startswith("conv_rows<bf16,k3,kg1,nt2>") for k in k1), k1                                                   # ... and no dense-block conv ran on the tile kernel
    assert torch.equal(y1, m.forward(xd))
    ref = M.esrgan_g_forward(x, w, 2, nb, dtype=np.float64, attention=False, bf16_storage=True)
    a, b = y0.float().cpu().numpy(), y1.float().cpu().numpy()
    assert a.shape == ref.shape == (B, 2 * H, 48, 3)
    e0, e1 = rel_l2(a, ref), rel_l2(b, ref)
    assert e1 <= 5e-3 and e1 <= 2.0 * e0 + 1e-3, (e0, e1)
    assert rel_l2(b, a) <= 5e-3
    # per image and per column: no image, and no column next to the seam or a border, stands out (a seam mistake would put its error into HR columns 44-51)
    err = np.abs(b - ref).max(axis=(1, 3))                                   # [B, HR columns]
    assert err.max() <= 16.0 * max(np.median(err), 2.0 ** -9), (float(err.max()), float(np.median(err)), np.unravel_index(err.argmax(), err.shape))
    # an image does not depend on its partner or on where its pair sits in the stream: bit for bit.  Which HALF it rides in may flip the last bf16 bit of a few values:
    # the RRDB's outer skip joins column group kx's accumulators behind ring granule kx (chain2_kernel, so_fetch), so the fp32 summation order differs between column
    # groups -- in the 48-wide layout just the same (tools/probe_pack.py: one value of rrdb_0_dense3_conv5 in 36 864, -0.18652 | -0.1875, fp32 sums either side of a tie).
    if B >= 4:
        perm = [2, 3, 0, 1] + list(range(4, B))
        y2 = m.forward(ctx.to_device(x[perm], torch.bfloat16))
        assert torch.equal(y2[:2], y1[2:4]) and torch.equal(y2[2:4], y1[:2])
        xs = x.copy()
        xs[0] = x[B - 1]                                                     # image 1 keeps its half and gets another partner
        y4 = m.forward(ctx.to_device(xs, torch.bfloat16))
        assert torch.equal(y4[1:B - 1], y1[1:B - 1])
    if B >= 3:
        y3 = m.forward(ctx.to_device(x[1:], torch.bfloat16))                 # every image moves to the other half, with another partner
        d = (y3.float() - y1[1:].float()).abs()
        assert float(d.max()) <= 2.0 ** -7 and float(d.mean()) <= 2.0 ** -13, (float(d.max()), float(d.mean()))
    # taps read the packed buffers through tap_copy's pair mapping: same output, and the tapped tensors agree with the layer-by-layer path's
    names = ["rrdb_0_dense1_conv1", "rrdb_0_dense1_conv3", "rrdb_0_dense1_conv5", f"rrdb_{nb - 1}_dense3_conv5", "trunk_conv"]
    ctx.profile_begin()
    yt, t1 = m.forward_with_taps(xd, names)
    assert "dense_tail_fused<bf16,conv4+conv5>" in {r["kernel"] for r in ctx.profile_end()}
    assert torch.equal(yt, y1)
    ctx.set_fused(dense, 0)
    _, t0 = m.forward_with_taps(xd, names)
    for n in names:
        assert t1[n].shape == t0[n].shape
        assert rel_l2(t1[n].cpu().numpy(), t0[n].cpu().numpy()) <= 5e-3, n
    assert float((t1["rrdb_0_dense1_conv1"] - t0["rrdb_0_dense1_conv1"]).abs().max()) <= 2.0 ** -7         # one conv on identical inputs: fp32 order apart, a last bf16 bit here and there


def test_two_up_packing_through_the_wrapper(ctx):
    """ESRGAN.super_resolve_image(patch_size_lr=24) in bf16: the packed path inside the reference's patch plumbing, against the oracle's patch-mode image."""
    from oracle import ops as O
    from sr355.synth import make_pairs
    from sr355.weights import condition_attention
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    m = ESRGAN(compute_dtype="bf16")
    m.setup_model(scale_factor=2, growth_channels=32, num_rrdb_blocks=2)
    w = bf16_rounded(condition_attention(init_weights(m.generator.layer_shapes(), seed=3600)))
    m.set_weights(w)
    lr, hr = make_pairs(1, 60, 52, 2, seed=11)
    ctx.profile_begin()
    sr, _ = m.super_resolve_image(lr[0], patch_size_lr=24, stride=12, batch_size=64)
    ks = {r["kernel"] for r in ctx.profile_end()}
    assert "dense_tail_fused<bf16,conv4+conv5>" in ks, ks
    ref = M.esrgan_super_resolve(lr[0], w, 2, 24, 12, num_rrdb=2, dtype=np.float64)
    assert sr.shape == ref.shape == (120, 104, 3)
    assert np.abs(sr - ref).max() <= 3e-2
    d = abs(float(O.psnr(hr[0], sr, dtype=np.float64)) - float(O.psnr(hr[0], np.clip(ref, 0, 1), dtype=np.float64)))
    assert d <= 0.01, d


@pytest.mark.parametrize("growth", [8, 16, 24])
def test_row_blocked_concat_buffers_at_small_growth_widths(ctx, growth):
    """Round 4: the concat buffers are row-blocked for every growth width that makes them a whole number of 32-channel blocks (the reference's notebook trains G = 8,
    ESRGAN.ipynb:L758-761: 96 channels); a growth conv's output slice then STARTS INSIDE a block (channels 64 + 8 k), which the epilogue's per-lane addressing has to
    place, and a conv reads the whole last block with zero weights on the channels not written yet.  Against the bf16-storage oracle, stage by stage."""
    nb = 2
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=nb, growth_channels=growth, use_attention=False, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), seed=3700 + growth))
    m.set_weights(w)
    for shape in ((3, 13, 21, 3), (2, 24, 24, 3), (1, 48, 48, 3)):
        x = round_to_bf16(np.random.default_rng(sum(shape)).uniform(-1, 1, shape).astype(np.float32))
        xd = ctx.to_device(x, torch.bfloat16)
        names = ["rrdb_0_dense1_conv1", "rrdb_0_dense1_conv4", "rrdb_0_dense1_conv5", "rrdb_0_dense3_conv5", f"rrdb_{nb - 1}_dense3_conv5", "trunk_conv"]
        y, taps = m.forward_with_taps(xd, names)
        assert torch.equal(y, m.forward(xd))                                  # taps do not change the path here (no fused dense-block kernels at these widths)
        parts = {}
        ref = M.esrgan_g_forward(x, w, 2, nb, dtype=np.float64, attention=False, bf16_storage=True, parts=parts)
        for dev, orc in (("rrdb_0_dense3_conv5", "rrdb_0"), (f"rrdb_{nb - 1}_dense3_conv5", f"rrdb_{nb - 1}")):
            e = rel_l2(taps[dev].cpu().numpy(), parts[orc])
            assert e <= 5e-3, (shape, dev, e)
        t1 = taps["rrdb_0_dense1_conv1"].cpu().numpy()
        assert t1.shape == shape[:3] + (growth,) and (t1 >= 0).all() and t1.max() > 0      # ReLU output of the first growth conv, its own channels only
        e = rel_l2(y.float().cpu().numpy(), ref)
        assert e <= 5e-3, (shape, e)
    # a second forward at another batch reuses the buffers: channels a conv has not written yet hold the previous forward's values, times zero weights
    x2 = round_to_bf16(np.random.default_rng(5).uniform(-1, 1, (3, 13, 21, 3)).astype(np.float32))
    y2 = m.forward(ctx.to_device(x2, torch.bfloat16))
    ref2 = M.esrgan_g_forward(x2, w, 2, nb, dtype=np.float64, attention=False, bf16_storage=True)
    assert rel_l2(y2.float().cpu().numpy(), ref2) <= 5e-3


@pytest.mark.parametrize("growth,shape", [(8, (24, 24)), (32, (24, 24)), (8, (13, 21)), (16, (7, 9)), (8, (24, 40))])
def test_cell_packed_dense_blocks_on_the_tile_kernels(fused_ctx, growth, shape):
    """Round 4 (api.hip `cellpack`): where the fused dense-block kernels do not apply and small images fill the 16 x 16 output tiles badly, the concat buffers hold the
    batch as one image of cells with zero separator rows / columns.  Every output pixel is the same sum in the same order as on the plain layout -- bit for bit the
    tapped forward's image (a tap switches the packing off) -- and the separators survive changes of batch size, a plain forward in between and the buffer the
    unpacked trunk input is written to."""
    ctx = fused_ctx
    H, W = shape
    nb = 2
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=nb, growth_channels=growth, use_attention=False, ctx=ctx)
    w = bf16_rounded(init_weights(m.layer_shapes(), seed=3800 + growth))
    m.set_weights(w)
    ctx.set_fused(ctx.FUSED_ALL & ~35, 0)                                     # G = 32 at 24 wide would otherwise ride two-up on the fused kernels

    def both(B, seed):
        x = round_to_bf16(np.random.default_rng(seed).uniform(-1, 1, (B, H, W, 3)).astype(np.float32))
        xd = ctx.to_device(x, torch.bfloat16)
        y = m.forward(xd)                                                      # cell-packed where it pays (B >= 4)
        yt, _ = m.forward_with_taps(xd, ["trunk_conv"])                        # plain layout
        assert torch.equal(y, yt), (B, float((y.float() - yt.float()).abs().max()))
        assert torch.equal(y, m.forward(xd))
        return x, y

    x, y = both(37, 1)                                                         # three rows of 16 cells, the last one partly filled
    ref = M.esrgan_g_forward(x[:6], w, 2, nb, dtype=np.float64, attention=False, bf16_storage=True)
    assert rel_l2(y[:6].float().cpu().numpy(), ref) <= 5e-3
    both(5, 2)                                                                 # gx = 5: another grid over the same buffers
    both(16, 3)
    both(3, 4)                                                                 # too few images: plain, over the separators
    both(37, 5)                                                                # ... and the grid again
    x48 = round_to_bf16(np.random.default_rng(6).uniform(-1, 1, (2, 48, 48, 3)).astype(np.float32))
    m.forward(ctx.to_device(x48, torch.bfloat16))                              # a plain forward at another size re-allocates / overwrites
    x2, y2 = both(37, 1)
    assert torch.equal(y2, y)
