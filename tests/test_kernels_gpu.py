"""Kernel-level parity: HIP ops called through the C ABI vs the CPU oracle on seeded inputs.

Tolerances: fp32 kernels (exact-fp32 MFMA fma chains) rel-L2 <= 1e-5 against the fp64 oracle
(SURVEY.md 8d).  bf16 kernels: operands are bf16-exact, the oracle's fp64 result is rounded to
bf16 where the device stores bf16 (oracle.ops.round_bf16), so all that may differ is an
accumulation-order flip across a rounding boundary: every element within one bf16 ulp
(2^-7 relative, plus an absolute floor for cancelled sums) and rel-L2 <= 1e-3.
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


def assert_bf16_close(got, ref, scale=1.0, rel=1e-3):
    """got: device bf16 result (as fp32); ref: fp64 oracle BEFORE the storage rounding."""
    refq = O.round_bf16(np.asarray(ref, np.float64))
    d = np.abs(np.asarray(got, np.float64) - refq)
    bound = 2.0 ** -7 * np.abs(refq) + 3e-5 * scale
    bad = d > bound
    assert not bad.any(), (int(bad.sum()), float(d.max()), float((d / bound).max()))
    err = rel_l2(got, refq)
    assert err <= rel, err


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
    (1, 9, 11, 64, 48, 1, "linear"),     # attention projection (streaming 1x1 kernel: 3 cout blocks x 2 chunks)
    (1, 13, 21, 32, 64, 1, "linear"),    # attention output conv (4 cout blocks x 1 chunk), ragged 16-pixel segments
    (2, 7, 40, 128, 16, 1, "lrelu"),     # 1x1, 1 cout block x 4 chunks
    (1, 5, 5, 160, 64, 1, "relu"),       # 1x1 too wide for the register-resident kernel: first-generation path
    (2, 48, 48, 32, 32, 3, "lrelu"),     # single 32-channel chunk on conv_rows (EDSR num_filters=32; the shape of round 1's probe fault)
    (3, 21, 35, 32, 64, 3, "relu"),      # single chunk, 64 couts per workgroup (36 weight DMA pieces), ragged tiles
    (1, 16, 16, 32, 16, 3, "linear"),    # single chunk, one 16-cout block (register-prefetched weights)
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
    if dtype == "f32":
        err = rel_l2(got, ref)
        assert err <= 1e-5, err
    else:
        assert_bf16_close(got, ref)


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
    if dtype == "f32":
        assert rel_l2(got, ref) <= 1e-5
    else:
        assert_bf16_close(got, ref)


def test_conv1x1_with_skips(ctx):
    """SelfAttention tail: y = conv1x1(o) + x (ESRGAN_model.py:66-69) on the streaming 1x1 kernel, one and two skips."""
    rng = np.random.default_rng(17)
    B, H, W = 2, 11, 37
    o = round_to_bf16(rng.standard_normal((B, H, W, 32)).astype(np.float32))
    s1 = round_to_bf16(rng.standard_normal((B, H, W, 64)).astype(np.float32))
    s2 = round_to_bf16(rng.standard_normal((B, H, W, 64)).astype(np.float32))
    w = round_to_bf16((rng.standard_normal((1, 1, 32, 64)) / 6).astype(np.float32))
    b = rng.uniform(-0.1, 0.1, 64).astype(np.float32)
    conv = O.conv2d(o, w, b, dtype=np.float64)
    od, d1, d2 = (_dev(ctx, a, torch.bfloat16) for a in (o, s1, s2))
    got1 = ctx.conv2d(od, w, b, skip1=d1, beta1=1.0).float().cpu().numpy()
    assert_bf16_close(got1, conv + s1)
    got2 = ctx.conv2d(od, w, b, alpha=0.5, skip1=d1, beta1=1.0, skip2=d2, beta2=0.2).float().cpu().numpy()
    assert_bf16_close(got2, 0.5 * conv + s1 + 0.2 * s2)


@pytest.mark.parametrize("which", [1, 2])
@pytest.mark.parametrize("hw", [(26, 18), (48, 48), (7, 5)])
def test_conv2d_skip_is_own_input(ctx, hw, which):
    """Dense-block tail: y = alpha*conv(x) + beta*x (+ another skip).  When a skip tensor IS the conv's first 64 input channels the
    bf16 3x3 kernel takes it from its LDS tile instead of reading it again (conv_rows.hip "skip from LDS"); same numbers either way."""
    rng = np.random.default_rng(11 + which)
    H, W = hw
    x = round_to_bf16(rng.standard_normal((3, H, W, 64)).astype(np.float32))
    other = round_to_bf16(rng.uniform(-1, 1, (3, H, W, 64)).astype(np.float32))
    w = round_to_bf16((rng.standard_normal((3, 3, 64, 64)) / 24).astype(np.float32))
    b = rng.uniform(-0.1, 0.1, 64).astype(np.float32)
    xd, od = _dev(ctx, x, torch.bfloat16), _dev(ctx, other, torch.bfloat16)
    conv = O.conv2d(x, w, b, dtype=np.float64)
    if which == 1:      # x is skip 1, alone
        ref = 0.2 * conv + x
        got = ctx.conv2d(xd, w, b, alpha=0.2, skip1=xd, beta1=1.0)
    else:               # RRDB tail: another tensor is skip 1, x is skip 2
        ref = 0.04 * conv + other + 0.2 * x
        got = ctx.conv2d(xd, w, b, alpha=0.04, skip1=od, beta1=1.0, skip2=xd, beta2=0.2)
    assert_bf16_close(got.float().cpu().numpy(), ref)
    # and against the same op with the skip passed as a separate copy (HBM path): equal up to one bf16 ulp of fp32 re-association
    xc = xd.clone()
    got2 = ctx.conv2d(xd, w, b, alpha=0.2, skip1=xc, beta1=1.0) if which == 1 else \
        ctx.conv2d(xd, w, b, alpha=0.04, skip1=od, beta1=1.0, skip2=xc, beta2=0.2)
    d = (got.float() - got2.float()).abs().max().item()
    assert d <= 2 ** -6 * max(1.0, float(np.abs(ref).max())), d


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


# ------------------------------------------------------------------------------------------------ attention
# bf16 attention vs oracle.ops.self_attention_bf16_storage (f/g/h, the probabilities and the output rounded to bf16 where the kernel
# rounds them).  What remains: the kernel takes probabilities relative to a possibly stale running max (another rounding draw of the
# same 2^-9 size) and accumulates in fp32.
BF16_ATTN_TOL = 4e-3


def _sa_weights(rng, C=64):
    mk = lambda ci, co: (rng.standard_normal((1, 1, ci, co)) / np.sqrt(ci)).astype(np.float32)
    bias = lambda co: rng.uniform(-0.1, 0.1, co).astype(np.float32)
    return [mk(C, C // 8), bias(C // 8), mk(C, C // 8), bias(C // 8), mk(C, C // 2), bias(C // 2), mk(C // 2, C), bias(C)]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("hw", [(24, 24), (48, 48), (13, 9), (1, 1), (11, 12)])   # N = 576, 2304, ragged 117, 1, 132
def test_self_attention_matches_oracle(ctx, hw, dtype):
    H, W = hw
    rng = np.random.default_rng(H * 100 + W)
    x = rng.standard_normal((2, H, W, 64)).astype(np.float32)
    ws = _sa_weights(rng)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        x = round_to_bf16(x)
        ws = [round_to_bf16(a) if a.ndim == 4 else a for a in ws]
    ref = (O.self_attention if dtype == "f32" else O.self_attention_bf16_storage)(x, *ws, dtype=np.float64)
    got = ctx.self_attention(_dev(ctx, x, td), *ws).float().cpu().numpy()
    err = rel_l2(got, ref)
    assert err <= (2e-5 if dtype == "f32" else BF16_ATTN_TOL), err


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("spike", [(15, 3), (5, 9), (31, 30)])
def test_self_attention_forced_max_jump(ctx, spike, dtype):
    """Online-softmax rescale: spike one key so the running max jumps by far more than the fast tiles tolerate, late in the key
    sweep (bf16: inside a fast key group, in its first tile, and in the ragged tail group), N = 32 * 31 + 8 keys."""
    rng = np.random.default_rng(11)
    H, W = 32, 31
    x = (0.1 * rng.standard_normal((1, H, W, 64))).astype(np.float32)
    if spike[0] < H and spike[1] < W:
        x[0, spike[0], spike[1], :] = 6.0          # a token with a huge projection
    ws = _sa_weights(rng)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        x = round_to_bf16(x)
        ws = [round_to_bf16(a) if a.ndim == 4 else a for a in ws]
    ref = (O.self_attention if dtype == "f32" else O.self_attention_bf16_storage)(x, *ws, dtype=np.float64)
    got = ctx.self_attention(_dev(ctx, x, td), *ws).float().cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) <= (2e-5 if dtype == "f32" else BF16_ATTN_TOL), rel_l2(got, ref)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("sign", [-1.0, 1.0])
def test_self_attention_large_uniform_scores(ctx, sign, dtype):
    """Scores far from zero: q.k = -/+400 for every pair plus small noise.  With the negative sign the first key tile's row maximum
    is far below zero (the running max starts there, exp2 of its negation overflows); with the positive sign the exponent range is
    used from the other end.  The softmax is shift-invariant, so the result must still match the fp64 oracle."""
    rng = np.random.default_rng(23)
    H, W = 12, 20                                   # N = 240: seven full key tiles + a ragged one
    x = (0.05 * rng.standard_normal((2, H, W, 64))).astype(np.float32)
    x[..., 0] = 20.0
    ws = _sa_weights(rng)
    ws[0][0, 0, 0, :] = 0.0; ws[0][0, 0, 0, 0] = 1.0          # f (keys):    dim 0 <- +channel 0
    ws[2][0, 0, 0, :] = 0.0; ws[2][0, 0, 0, 0] = sign * 1.0   # g (queries): dim 0 <- +/- channel 0
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    if dtype == "bf16":
        x = round_to_bf16(x)
        ws = [round_to_bf16(a) if a.ndim == 4 else a for a in ws]
    ref = (O.self_attention if dtype == "f32" else O.self_attention_bf16_storage)(x, *ws, dtype=np.float64)
    got = ctx.self_attention(_dev(ctx, x, td), *ws).float().cpu().numpy()
    assert np.isfinite(got).all()
    # bf16: k and q are rounded to bf16 after the projection, so scores of magnitude 400 carry errors of order 1 -- the bf16-storage
    # restatement of the oracle rounds them at the same place, which is what lets this bound be the ordinary one
    assert rel_l2(got, ref) <= (2e-5 if dtype == "f32" else BF16_ATTN_TOL), rel_l2(got, ref)


# ------------------------------------------------------------------------------------------------ image ops
@pytest.mark.parametrize("shape,out", [((1, 64, 64, 3), (256, 256)), ((2, 23, 31, 3), (46, 62)), ((1, 239, 239, 3), (478, 478)),
                                       ((1, 10, 10, 1), (37, 23)),
                                       ((2, 40, 40, 3), (30, 30)),      # mild down-scale: still the LDS-window kernel
                                       ((1, 64, 64, 3), (16, 16)),      # 4x down-scale: window too large, per-pixel kernel
                                       ((1, 3, 5, 3), (130, 70))])      # tiny source, every tap clamped somewhere
def test_bicubic_f32(ctx, shape, out):
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, shape).astype(np.float32)
    ref = O.bicubic_resize(x, out[0], out[1])
    got = ctx.bicubic(_dev(ctx, x, torch.float32), out[0], out[1]).cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 2e-6


def test_bicubic_u8_bit_exact(ctx):
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    ref = O.bicubic_resize_u8(x, 256, 256)
    got = ctx.bicubic(ctx.to_device(x[None], torch.uint8), 256, 256).cpu().numpy()[0]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("shape", [(4, 24, 24, 3), (2, 256, 256, 3), (1, 11, 40, 3), (3, 37, 53, 1)])
def test_psnr_ssim(ctx, shape):
    rng = np.random.default_rng(9)
    a = rng.uniform(0, 1, shape).astype(np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(shape), 0, 1).astype(np.float32)
    da, db = _dev(ctx, a, torch.float32), _dev(ctx, b, torch.float32)
    p = ctx.psnr(da, db).cpu().numpy()
    s = ctx.ssim(da, db).cpu().numpy()
    assert np.allclose(p, O.psnr(a, b, dtype=np.float64), atol=2e-4)          # dB
    assert np.allclose(s, O.ssim(a, b, dtype=np.float64), atol=5e-5)
    m = ctx.mse(da, db).cpu().numpy()[0]
    assert np.isclose(m, np.mean((a.astype(np.float64) - b) ** 2), rtol=1e-5)


def test_psnr_identical_is_inf_and_ssim_one(ctx):
    a = _dev(ctx, np.random.default_rng(1).uniform(0, 1, (2, 24, 24, 3)), torch.float32)
    assert torch.isinf(ctx.psnr(a, a)).all()
    assert np.allclose(ctx.ssim(a, a).cpu().numpy(), 1.0, atol=1e-6)


def test_ssim_too_small_raises(ctx):
    a = _dev(ctx, np.zeros((1, 10, 24, 3)), torch.float32)
    with pytest.raises(ValueError):
        ctx.ssim(a, a)


@pytest.mark.parametrize("hw,p,s,scale", [((239, 239), 24, 12, 2), ((512, 512), 48, 24, 4), ((100, 77), 33, 14, 1), ((48, 48), 48, 24, 2)])
def test_extract_and_overlap_add(ctx, hw, p, s, scale):
    rng = np.random.default_rng(12)
    img = rng.uniform(0, 1, (*hw, 3)).astype(np.float32)
    padded = O.add_padding(img, p, s)
    ref_patches, pos = O.extract_patches(padded, p, s)
    got = ctx.extract_patches(_dev(ctx, img, torch.float32), p, s, mul=2.0, add=-1.0)
    assert got.shape == ref_patches.shape
    assert np.array_equal(got.cpu().numpy(), ref_patches * 2.0 - 1.0)
    # fake "HR" patches: arbitrary values, reconstruct with the reference's scatter-add semantics
    if scale * hw[0] * scale * hw[1] > 3e6:
        hr = np.repeat(np.repeat(ref_patches, scale, axis=1), scale, axis=2)
    else:
        hr = rng.uniform(-0.2, 1.2, (len(pos), p * scale, p * scale, 3)).astype(np.float32)
    ref = O.overlap_add(hr, pos, padded.shape, hw, p, scale)
    out = ctx.overlap_add(_dev(ctx, hr, torch.float32), hw[0], hw[1], p, s, scale).cpu().numpy()
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) <= 1e-6


# ------------------------------------------------------------------------------------------------ cv2.resize family (sr_resize)
RESIZE_CASES = [
    # shape, (out_h, out_w)
    ((1, 24, 20, 3), (48, 40)),       # x2 up: the loader's LR -> HR
    ((2, 23, 31, 3), (46, 93)),       # x2 / x3 up, ragged
    ((1, 17, 9, 1), (40, 37)),        # non-integer factors, single channel
    ((1, 30, 42, 3), (30, 42)),       # same size
]


@pytest.mark.parametrize("interp", ["INTER_LINEAR", "INTER_AREA", "INTER_LANCZOS4", "INTER_CUBIC"])
@pytest.mark.parametrize("shape,out", RESIZE_CASES)
def test_resize_f32_matches_oracle(ctx, shape, out, interp):
    x = np.random.default_rng(sum(shape) + out[0]).uniform(0, 1, shape).astype(np.float32)
    code = {"INTER_LINEAR": O.INTER_LINEAR, "INTER_AREA": O.INTER_AREA, "INTER_LANCZOS4": O.INTER_LANCZOS4, "INTER_CUBIC": O.INTER_CUBIC}[interp]
    ref = O.cv_resize(x, out[0], out[1], code)
    got = ctx.resize(ctx.to_device(x), out[0], out[1], interp).cpu().numpy()
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-6, float(np.max(np.abs(got - ref)))
    assert np.array_equal(got, ctx.resize(ctx.to_device(x), out[0], out[1], code).cpu().numpy())      # OpenCV integer code == name


@pytest.mark.parametrize("shape,out", [((1, 48, 40, 3), (24, 20)), ((2, 37, 50, 3), (11, 17)), ((1, 64, 64, 3), (5, 9))])
def test_resize_area_shrinking_f32(ctx, shape, out):
    """INTER_AREA, both axes shrinking: weighted box average (computeResizeAreaTab); integer factors = plain block means."""
    x = np.random.default_rng(3).uniform(0, 1, shape).astype(np.float32)
    ref = O.cv_resize(x, out[0], out[1], O.INTER_AREA)
    got = ctx.resize(ctx.to_device(x), out[0], out[1], "INTER_AREA").cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 2e-6
    if shape[1] % out[0] == 0 and shape[2] % out[1] == 0:
        fy, fx = shape[1] // out[0], shape[2] // out[1]
        assert np.max(np.abs(got - x.reshape(shape[0], out[0], fy, out[1], fx, shape[3]).mean(axis=(2, 4)))) <= 1e-6


@pytest.mark.parametrize("interp", ["INTER_LINEAR", "INTER_AREA", "INTER_LANCZOS4", "INTER_CUBIC"])
def test_resize_u8_bit_exact(ctx, interp):
    rng = np.random.default_rng(5)
    for shape, out in (((1, 24, 20, 3), (48, 40)), ((1, 19, 23, 3), (50, 31)), ((1, 8, 8, 1), (8, 9))):
        x = rng.integers(0, 256, shape, dtype=np.uint8)
        code = ctx.INTERPOLATIONS[interp]
        ref = O.cv_resize_u8(x[0], out[0], out[1], code)
        got = ctx.resize(ctx.to_device(x, torch.uint8), out[0], out[1], interp).cpu().numpy()[0]
        assert np.array_equal(got, ref), (shape, out, int(np.abs(got.astype(int) - ref.astype(int)).max()))
    for bad in ("INTER_NEAREST_EXACT", 6, 7, -1, "INTER_LINEAR_EXACT", True):     # not restated / not a cv2.resize code (cv2.error in the reference)
        with pytest.raises(ValueError):
            ctx.resize(ctx.to_device(x, torch.uint8), 16, 16, bad)


@pytest.mark.parametrize("shape,out", [((1, 48, 40, 3), (24, 20)),      # 2 x 2 cells, 3 channels: (sum + 2) >> 2
                                       ((2, 48, 40, 2), (24, 20)),      # 2 x 2 cells, 2 channels: the rounded float product
                                       ((1, 36, 45, 3), (12, 9)),       # 3 x 5 cells
                                       ((1, 60, 32, 1), (15, 32)),      # 4 x 1 cells (one axis unchanged)
                                       ((2, 37, 50, 3), (11, 17)),      # no whole-number factor: float taps
                                       ((1, 64, 64, 4), (5, 9)),
                                       ((1, 30, 21, 3), (10, 20))])     # one axis whole-number, one not: float taps
def test_resize_area_shrinking_u8_bit_exact(ctx, shape, out):
    """uint8 INTER_AREA shrinking (classic_algorithms.py:15-17 on the notebook's uint8 images): resizeAreaFast_ for whole-number factors,
    resizeArea_<uchar, float> otherwise -- to the last bit of the CPU restatement, values at both ends of the range included."""
    rng = np.random.default_rng(shape[1] * 7 + out[0])
    x = rng.integers(0, 256, shape, dtype=np.uint8)
    x[:, : shape[1] // 3] = 255
    x[:, -(shape[1] // 4):, : shape[2] // 2] = 0
    got = ctx.resize(ctx.to_device(x, torch.uint8), out[0], out[1], "INTER_AREA").cpu().numpy()
    for b in range(shape[0]):
        ref = O.cv_resize_u8(x[b], out[0], out[1], O.INTER_AREA)
        assert np.array_equal(got[b], ref), (shape, out, int(np.abs(got[b].astype(int) - ref.astype(int)).max()))
    if shape[1] == 2 * out[0] and shape[2] == 2 * out[1]:                 # cv2.resize turns bilinear halving into this very box mean
        lin = ctx.resize(ctx.to_device(x, torch.uint8), out[0], out[1], "INTER_LINEAR").cpu().numpy()
        assert np.array_equal(lin, got)
        assert np.array_equal(lin[0], O.cv_resize_u8(x[0], out[0], out[1], O.INTER_LINEAR))


@pytest.mark.parametrize("shape,out", [((2, 24, 20, 3), (48, 40)), ((1, 17, 9, 1), (40, 37)), ((1, 37, 50, 3), (11, 17)), ((1, 8, 8, 3), (8, 8))])
def test_resize_nearest(ctx, shape, out):
    """INTER_NEAREST = 0: a code interpolation_map.pkl may hold as an integer (loading_methods.py:146-147 hands it to cv2.resize as it is)."""
    rng = np.random.default_rng(out[1])
    xf = rng.uniform(0, 1, shape).astype(np.float32)
    xu = rng.integers(0, 256, shape, dtype=np.uint8)
    for name in ("INTER_NEAREST", 0):
        assert np.array_equal(ctx.resize(ctx.to_device(xf), out[0], out[1], name).cpu().numpy(), O.cv_resize(xf, out[0], out[1], O.INTER_NEAREST))
        got = ctx.resize(ctx.to_device(xu, torch.uint8), out[0], out[1], name).cpu().numpy()
        assert np.array_equal(got, np.stack([O.cv_resize_u8(xu[b], out[0], out[1], O.INTER_NEAREST) for b in range(shape[0])]))
    assert np.array_equal(ctx.resize(ctx.to_device(xf), out[0], out[1], "INTER_LINEAR_EXACT").cpu().numpy(),
                          ctx.resize(ctx.to_device(xf), out[0], out[1], "INTER_LINEAR").cpu().numpy())      # float images: OpenCV falls back to INTER_LINEAR
