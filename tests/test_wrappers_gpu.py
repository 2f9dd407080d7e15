"""The reference-shaped classes (SRModels/*) end to end on the GPU: super_resolve_image / evaluate / classify /
metrics against the oracle pipelines, plus the reference's guard exceptions."""
import os

import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ops as O
from sr355.synth import make_pairs
from sr355.weights import bf16_rounded, init_weights

pytestmark = pytest.mark.gpu


def test_metrics_module(ctx):
    from SRModels.metrics import psnr, ssim
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (2, 24, 24, 3)).astype(np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    assert np.allclose(psnr(a, b), O.psnr(a, b, dtype=np.float64), atol=2e-4)
    assert np.allclose(ssim(a, b), O.ssim(a, b, dtype=np.float64), atol=5e-5)
    assert psnr(a[0], b[0]).shape == ()


def test_bicubic_cfg0_plumbing(ctx):
    """BASELINE configs[0]: bicubic 64x64 -> 256x256 + PSNR/SSIM against the HR tile."""
    from SRModels.classic_super_resolution_algorithms.classic_algorithms import interpolate_bicubic
    from SRModels.metrics import psnr, ssim
    lr, hr = make_pairs(1, 64, 64, 4, seed=42)
    up = interpolate_bicubic(lr[0], (256, 256))
    ref = O.bicubic_resize(lr[0], 256, 256)
    assert np.max(np.abs(up - ref)) <= 2e-6
    assert abs(float(psnr(hr[0], up)) - float(O.psnr(hr[0], ref, dtype=np.float64))) <= 1e-3
    assert abs(float(ssim(hr[0], up)) - float(O.ssim(hr[0], ref, dtype=np.float64))) <= 1e-4
    u8 = (lr[0] * 255).astype(np.uint8)
    assert np.array_equal(interpolate_bicubic(u8, (256, 256)), O.bicubic_resize_u8(u8, 256, 256))


def test_srcnn_wrapper(ctx, tmp_path):
    from SRModels.deep_learning_models.SRCNN_model import SRCNNModel
    m = SRCNNModel()
    with pytest.raises(ValueError):
        m.setup_model()
    with pytest.raises(FileNotFoundError):
        m.setup_model(from_pretrained=True, pretrained_path=str(tmp_path / "none.npz"))
    m.setup_model(input_shape=(24, 24, 3))
    lr, hr = make_pairs(1, 30, 26, 2, seed=3)
    with pytest.raises(RuntimeError):
        m.super_resolve_image(lr[0], 60, 52)
    m.set_weights(m.weights)
    with pytest.raises(ValueError):
        m.super_resolve_image([[1, 2]], 60, 52)
    sr, met = m.super_resolve_image(lr[0], 60, 52)
    ref = M.srcnn_super_resolve(lr[0], m.weights, 60, 52, dtype=np.float64)
    assert sr.shape == (60, 52, 3) and np.max(np.abs(sr - ref)) <= 1e-5
    assert set(met) == {"time_sec", "gpu_mean_current_mb", "gpu_peak_mb"} and met["gpu_peak_mb"] > 0
    # evaluate == keras [mse, psnr, ssim] sample-weighted means
    X = np.random.default_rng(1).uniform(0, 1, (5, 24, 24, 3)).astype(np.float32)
    Y = np.clip(X + 0.02, 0, 1)
    res = m.evaluate(X, Y)
    P_ = M.srcnn_forward(X, m.weights, dtype=np.float64)
    assert np.isclose(res[0], np.mean((P_ - Y) ** 2), rtol=1e-4)
    assert np.isclose(res[1], O.psnr(Y, P_, dtype=np.float64).mean(), atol=1e-3)
    assert np.isclose(res[2], O.ssim(Y, P_, dtype=np.float64).mean(), atol=1e-4)
    path = m.save(str(tmp_path), "t0")
    m2 = SRCNNModel()
    m2.setup_model(from_pretrained=True, pretrained_path=path)
    sr2, _ = m2.super_resolve_image(lr[0], 60, 52)
    assert np.array_equal(sr, sr2)


def test_edsr_wrapper(ctx):
    from SRModels.deep_learning_models.EDSR_model import EDSR
    m = EDSR()
    with pytest.raises(ValueError):
        m.setup_model(scale_factor=5)
    m.setup_model(scale_factor=4, num_res_blocks=2)
    m.set_weights(m.weights)
    lr, _ = make_pairs(1, 50, 37, 4, seed=4)
    sr, _ = m.super_resolve_image(lr[0], patch_size_lr=24, stride=12)
    ref = M.edsr_super_resolve(lr[0], m.weights, 4, 24, 12, num_res_blocks=2, dtype=np.float64)
    assert sr.shape == (200, 148, 3) and np.max(np.abs(sr - ref)) <= 1e-5


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("bf16", 2e-2)])
def test_esrgan_wrapper(ctx, dtype, tol):
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    m = ESRGAN(compute_dtype=dtype)
    m.setup_model(scale_factor=2, growth_channels=8, num_rrdb_blocks=4)      # the notebook's configuration
    lr, hr = make_pairs(1, 40, 52, 2, seed=5)
    with pytest.raises(RuntimeError):
        m.super_resolve_image(lr[0])
    w = bf16_rounded(m.weights) if dtype == "bf16" else m.weights
    m.set_weights(w)
    sr, met = m.super_resolve_image(lr[0], patch_size_lr=24, stride=12, batch_size=16)
    ref = M.esrgan_super_resolve(lr[0], w, 2, 24, 12, num_rrdb=4, dtype=np.float64)
    assert sr.shape == (80, 104, 3) and sr.min() >= 0 and sr.max() <= 1
    assert np.max(np.abs(sr - ref)) <= tol
    # north-star parity metric: |PSNR(gpu,HR) - PSNR(oracle,HR)| <= 0.01 dB
    d = abs(float(O.psnr(hr[0], sr, dtype=np.float64)) - float(O.psnr(hr[0], ref, dtype=np.float64)))
    assert d <= 0.01, d
    # device-tensor in -> device-tensor out, same numbers
    sr_t, _ = m.super_resolve_image(ctx.to_device(lr[0]), patch_size_lr=24, stride=12)
    assert isinstance(sr_t, torch.Tensor) and np.array_equal(sr_t.cpu().numpy(), sr)
    # several images per generator call: identical results, image by image
    many, _ = m.super_resolve_images([lr[0], lr[0][::-1].copy()], patch_size_lr=24, stride=12)
    assert np.array_equal(many[0], sr)
    assert np.array_equal(many[1], m.super_resolve_image(lr[0][::-1].copy(), patch_size_lr=24, stride=12)[0])
    ev = m.evaluate([(lr[:1, :24, :24] * 2 - 1, hr[:1, :48, :48] * 2 - 1)])
    assert set(ev) == {"avg_psnr", "avg_ssim", "avg_g_loss"}


def test_vgg16_wrapper(ctx):
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN  # noqa: F401  (import check of the sibling module)
    from SRModels.defect_detection_models.VGG16_model import FineTunedVGG16
    m = FineTunedVGG16()
    m.setup_model(input_shape=(96, 96, 3), num_classes=2)
    img = make_pairs(1, 60, 50, 4, seed=6)[1][0]           # 240 x 200 "SR output"
    with pytest.raises(ValueError):
        m.classify_defects_method(img[:, :, 0])
    cls, conf = m.classify_defects_method(img, patch_size=96, stride=48)
    rcls, rconf = M.classify_defects(img, m.weights, 96, 48, dtype=np.float64)
    assert cls == rcls and abs(conf - rconf) <= 1e-4


def test_sr_then_classify_pipeline(ctx):
    """BASELINE configs[4] in miniature: x4 ESRGAN super-resolution, then the VGG16 patch vote on the SR image, device-resident."""
    from sr355.pipeline import sr_then_classify
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from SRModels.defect_detection_models.VGG16_model import FineTunedVGG16
    g = ESRGAN(compute_dtype="f32")
    g.setup_model(scale_factor=4, growth_channels=8, num_rrdb_blocks=1)
    g.set_weights(g.weights)
    c = FineTunedVGG16()
    c.setup_model(input_shape=(96, 96, 3), num_classes=2)
    lr, _ = make_pairs(1, 40, 36, 4, seed=8)
    sr, met, cls, conf = sr_then_classify(g, c, lr[0], sr_kwargs=dict(patch_size_lr=24, stride=12))
    ref_sr = M.esrgan_super_resolve(lr[0], g.weights, 4, 24, 12, num_rrdb=1, dtype=np.float64)
    assert tuple(sr.shape) == (160, 144, 3) and np.max(np.abs(sr.cpu().numpy() - ref_sr)) <= 1e-5
    rcls, rconf = M.classify_defects(ref_sr.astype(np.float32), c.weights, 96, 48, dtype=np.float64)
    assert cls == rcls and abs(conf - rconf) <= 1e-4 and "time_sec" in met


@pytest.mark.parametrize("side", [128, 512])
def test_whole_image_attention_sampled_rows(ctx, side):
    """SelfAttention on a whole feature map -- 128x128 (N = 16384) and the 512x512 LR tile of BASELINE configs[2] (N = 262144: the
    score matrix would be 550 GB in fp64 and is never materialised on either side): the HIP streaming-softmax kernel
    against the fp64 streaming oracle on sampled query rows."""
    from sr355.weights import init_weights
    rng = np.random.default_rng(21)
    H = W = side
    x = (0.5 * rng.standard_normal((1, H, W, 64))).astype(np.float32)
    w = init_weights(M.self_attention_layers("sa"), seed=5)
    y = ctx.self_attention(ctx.to_device(x), *w["sa_f"], *w["sa_g"], *w["sa_h"], *w["sa_v"]).cpu().numpy()
    X = x.reshape(-1, 64).astype(np.float64)
    f = X @ w["sa_f"][0][0, 0] + w["sa_f"][1]
    g = X @ w["sa_g"][0][0, 0] + w["sa_g"][1]
    h = X @ w["sa_h"][0][0, 0] + w["sa_h"][1]
    rows = rng.choice(H * W, size=64, replace=False)
    o = O.attention_rows_streaming(g[rows], f, h)
    ref = X[rows] + o @ w["sa_v"][0][0, 0] + w["sa_v"][1]
    got = y.reshape(-1, 64)[rows]
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 2e-5


def test_full_depth_generator_bf16_tracks_fp32(ctx):
    """The bench generator at full depth (ESRGAN x4, NB=23, G=32, both SelfAttention layers) in reference patch mode: the bf16 path
    against the library's own fp32 path (both pinned to the oracle at full depth by tests/test_full_depth_gpu.py), on the
    attention-conditioned synthetic weights (sr355.weights.condition_attention); a second call on
    the same tile must give the same image (nothing may leak from one forward into the next through the workspaces)."""
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from sr355.weights import condition_attention, init_weights
    lr4, hr4 = make_pairs(1, 120, 144, 4, seed=44)                # 4 x 5 = 20 overlapping patches of 48 (stride 24); the CPU oracle below is what the test's time is
    lr, hr = ctx.to_device(lr4), ctx.to_device(hr4)
    out, w = {}, None
    for dt in ("f32", "bf16"):
        m = ESRGAN(compute_dtype=dt)
        m.setup_model(scale_factor=4, growth_channels=32, num_rrdb_blocks=23, use_attention=True)
        if w is None:
            w = condition_attention(init_weights(m.generator.layer_shapes(), seed=3000))
        m.set_weights(w)
        sr = m.super_resolve_image(lr[0], patch_size_lr=48, stride=24, batch_size=64)[0]
        sr2 = m.super_resolve_image(lr[0], patch_size_lr=48, stride=24, batch_size=64)[0]
        assert torch.isfinite(sr).all()
        assert torch.equal(sr, sr2)
        out[dt] = (sr, float(ctx.psnr(hr, sr[None])[0]))
        del m
    d = float(ctx.psnr(out["f32"][0][None], out["bf16"][0][None])[0])
    assert d >= 40.0, d                                            # bf16 storage noise floor at this depth: ~48-49 dB per patch, overlap-averaged here
    # The north star's figure, against the ORACLE (VERDICT r2 weak #1c): the CPU restatement's own patch-mode super-resolution of the same
    # image (fp32 reference graph, same weights) -- |PSNR(gpu, HR) - PSNR(oracle, HR)| <= 0.01 dB for the bf16 path, and the fp32 path on it
    ref = M.esrgan_super_resolve(lr4[0], w, 4, 48, 24, num_rrdb=23)
    p_ref = float(O.psnr(hr4, ref[None].astype(np.float32), dtype=np.float64)[0])
    assert abs(out["bf16"][1] - p_ref) <= 0.01, (out["bf16"][1], p_ref)
    assert abs(out["f32"][1] - p_ref) <= 1e-3, (out["f32"][1], p_ref)
    assert float(np.abs(out["f32"][0].cpu().numpy() - ref).max()) <= 1e-4


def test_classic_resizers_and_srcnn_mode_loader_with_interpolation_map(ctx, tmp_path):
    """classic_algorithms.py:7-21 through the reference-shaped wrappers, and load_dataset_as_patches(mode="srcnn") on a dataset whose
    interpolation_map.pkl names a different OpenCV interpolation per file -- constant names and integer codes, as
    loading_methods.py:131-148 accepts them -- against the oracle's restatement of the same loop."""
    import os
    import pickle
    from PIL import Image
    from SRModels import loading_methods as LM
    from SRModels.classic_super_resolution_algorithms import classic_algorithms as CA
    rng = np.random.default_rng(12)
    lr = rng.uniform(0, 1, (20, 26, 3)).astype(np.float32)
    for fn, code in ((CA.interpolate_bilinear, O.INTER_LINEAR), (CA.interpolate_area, O.INTER_AREA), (CA.interpolate_lanczos, O.INTER_LANCZOS4),
                     (CA.interpolate_bicubic, O.INTER_CUBIC)):
        up = fn(lr, (52, 40))                                     # (width, height), as cv2.resize takes it
        assert up.shape == (40, 52, 3) and up.dtype == np.float32
        assert np.max(np.abs(up - O.cv_resize(lr, 40, 52, code))) <= 2e-6
        u8 = (lr * 255).astype(np.uint8)
        assert np.array_equal(fn(u8, (52, 40)), O.cv_resize_u8(u8, 40, 52, code))
        assert fn(lr[:, :, 0], (52, 40)).shape == (40, 52)        # grayscale in, grayscale out
    with pytest.raises(NotImplementedError):
        CA.back_projection(lr, lr)
    # ---- the loader
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "hr"))
    os.makedirs(os.path.join(root, "lr"))
    names = ["a.png", "b.png", "c.png", "d.png", "e.png", "f.png"]
    methods = {"a.png": "INTER_LINEAR", "b.png": "INTER_AREA", "c.png": 4, "d.png": "INTER_SOMETHING_ELSE", "f.png": 0}     # e.png: not in the map -> cubic; 0 = cv2.INTER_NEAREST, as an integer straight to the resize
    imgs = {}
    for n in names:
        h = rng.integers(0, 256, (40, 36, 3), dtype=np.uint8)
        l = h.reshape(20, 2, 18, 2, 3).mean(axis=(1, 3)).astype(np.uint8)
        Image.fromarray(h).save(os.path.join(root, "hr", n))
        Image.fromarray(l).save(os.path.join(root, "lr", n))
        imgs[n] = (h.astype(np.float32) / 255.0, l.astype(np.float32) / 255.0)
    mp = os.path.join(root, "interpolation_map.pkl")
    with open(mp, "wb") as f:
        pickle.dump(methods, f)
    X, Y, hh, ww = LM.load_dataset_as_patches(os.path.join(root, "hr"), os.path.join(root, "lr"), mode="srcnn", patch_size=12, stride=6,
                                              interpolation_map_path=mp)
    codes = {"a.png": O.INTER_LINEAR, "b.png": O.INTER_AREA, "c.png": O.INTER_LANCZOS4, "d.png": O.INTER_CUBIC, "e.png": O.INTER_CUBIC, "f.png": O.INTER_NEAREST}
    Xr, Yr = [], []
    for n in names:
        h, l = imgs[n]
        up = np.clip(O.cv_resize(l, 40, 36, codes[n]), 0.0, 1.0)
        hp, lp = O.add_padding(h, 12, 6), O.add_padding(up, 12, 6)
        for i, j in O.patch_positions(hp.shape[0], hp.shape[1], 12, 6):
            Xr.append(lp[i:i + 12, j:j + 12])
            Yr.append(hp[i:i + 12, j:j + 12])
    assert (hh, ww) == (40, 36) and X.shape == np.array(Xr).shape and np.array_equal(Y, np.array(Yr))
    assert np.max(np.abs(X - np.array(Xr))) <= 2e-6
    # without a map every file takes INTER_CUBIC (the reference raises NameError there: SURVEY.md Appendix C.1)
    X2, _, _, _ = LM.load_dataset_as_patches(os.path.join(root, "hr"), os.path.join(root, "lr"), mode="srcnn", patch_size=12, stride=6)
    up = np.clip(O.cv_resize(imgs["a.png"][1], 40, 36, O.INTER_CUBIC), 0.0, 1.0)
    assert np.max(np.abs(X2[0] - O.add_padding(up, 12, 6)[:12, :12])) <= 2e-6


def test_streaming_sr_then_classify_frames(ctx):
    """BASELINE configs[4] on small frames: a stream of uint8 / float frames through ESRGAN x4 and the VGG16 vote, frame by frame
    against the oracle; the rank shards partition the stream."""
    from sr355.pipeline import stream_sr_classify
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from SRModels.defect_detection_models.VGG16_model import FineTunedVGG16
    g = ESRGAN(compute_dtype="f32")
    g.setup_model(scale_factor=4, growth_channels=8, num_rrdb_blocks=1)
    g.set_weights(g.weights)
    c = FineTunedVGG16()
    c.setup_model(input_shape=(96, 96, 3), num_classes=2)
    lr, _ = make_pairs(5, 40, 36, 4, seed=18)
    frames = [lr[0], (lr[1] * 255).astype(np.uint8), lr[2], ctx.to_device(lr[3]), lr[4]]          # float, uint8, float, device tensor, float
    kw = dict(patch_size_lr=24, stride=12, batch_size=64)
    res, stats = stream_sr_classify(g, c, frames, sr_kwargs=kw, keep_sr=True)
    assert [r["frame"] for r in res] == [0, 1, 2, 3, 4] and stats["frames"] == 5 and stats["frames_per_s"] > 0
    for r in res:
        f = frames[r["frame"]]
        f = f.cpu().numpy() if isinstance(f, torch.Tensor) else np.asarray(f)
        f01 = f.astype(np.float32) / 255.0 if f.dtype == np.uint8 else f
        ref_sr = M.esrgan_super_resolve(f01, g.weights, 4, 24, 12, num_rrdb=1, dtype=np.float64)
        assert np.max(np.abs(r["sr"].cpu().numpy() - ref_sr)) <= 1e-5
        rcls, rconf = M.classify_defects(ref_sr.astype(np.float32), c.weights, 96, 48, dtype=np.float64)
        assert r["class"] == rcls and abs(r["confidence"] - rconf) <= 1e-4
    parts = [stream_sr_classify(g, c, frames, sr_kwargs=kw, rank=r, world=2)[0] for r in range(2)]
    assert [x["frame"] for x in parts[0]] == [0, 1, 2] and [x["frame"] for x in parts[1]] == [3, 4]
    assert [(x["class"], x["confidence"]) for p_ in parts for x in p_] == [(r["class"], r["confidence"]) for r in res]


def test_streaming_pipeline_full_size_frame_properties(ctx):
    """One 1080x1920 frame through the bench generator (x4, NB=23, G=32, bf16) and the bf16 classifier: 45 x 80 = 3600 LR patches,
    a 4320 x 7680 SR frame, 90 x 160 = 14400 classifier patches.  Size-independent properties only (the
    oracle needs minutes per frame): shape, range, finiteness, the patch counts, and the same (class, confidence) on a second pass."""
    from sr355.pipeline import patch_grid, stream_sr_classify
    from sr355.synth import hr_tile
    from sr355.weights import condition_attention, init_weights
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from SRModels.defect_detection_models.VGG16_model import FineTunedVGG16
    g = ESRGAN(compute_dtype="bf16")
    g.setup_model(scale_factor=4, growth_channels=32, num_rrdb_blocks=23)
    g.set_weights(condition_attention(init_weights(g.generator.layer_shapes(), seed=3000)))
    c = FineTunedVGG16(compute_dtype="bf16")
    c.setup_model(input_shape=(96, 96, 3), num_classes=2)
    c.set_weights(c.weights)
    frame = (hr_tile(np.random.default_rng(4), 1080, 1920) * 255).astype(np.uint8)
    assert patch_grid(1080, 1920, 48, 24) == (45, 80) and patch_grid(4320, 7680, 96, 48) == (90, 160)
    kw = dict(patch_size_lr=48, stride=24, batch_size=3600)
    res, stats = stream_sr_classify(g, c, [frame, frame], sr_kwargs=kw, batch_size=1024, keep_sr=True)
    sr = res[0]["sr"]
    assert tuple(sr.shape) == (4320, 7680, 3) and bool(torch.isfinite(sr).all()) and float(sr.min()) >= 0.0 and float(sr.max()) <= 1.0
    assert (res[0]["class"], res[0]["confidence"]) == (res[1]["class"], res[1]["confidence"]) and torch.equal(res[0]["sr"], res[1]["sr"])
    assert res[0]["class"] in (0, 1) and 0.0 < res[0]["confidence"] <= 1.0
    print(f"\n1080p frame: {stats['frames_per_s']:.2f} frames/s, {stats['sr_output_mpix_per_s']:.1f} SR MPix/s (first frame includes workspace allocation)")
    g.generator.release_workspace()
    c.model.release_workspace()


def test_keras_h5_checkpoints_load_and_save(ctx, tmp_path):
    """`setup_model(from_pretrained=True, pretrained_path=...h5)` (SRCNN_model.py:23-43) on the committed Keras-layout file
    written by the genuine HDF5 library (tests/golden/srcnn_keras_layout.h5), and `save(..., fmt="h5")` -> the reference's file name, which
    loads back to the same predictions (SRCNN_model.py:249-260).  ESRGAN's generator / discriminator pair the same way (:981-995, :143-149)."""
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from SRModels.deep_learning_models.SRCNN_model import SRCNNModel
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "srcnn_keras_layout.h5")
    w = init_weights(M.srcnn_layers(), seed=1000)
    x = np.random.default_rng(5).uniform(0, 1, (2, 20, 24, 3)).astype(np.float32)
    m = SRCNNModel()
    m.setup_model(from_pretrained=True, pretrained_path=golden)
    got = np.asarray(m.model.predict(x))
    assert np.max(np.abs(got - M.srcnn_forward(x, w, dtype=np.float64))) <= 1e-5
    path = m.save(str(tmp_path), "20260101", fmt="h5")
    assert os.path.basename(path) == "SRCNN_20260101.h5"
    m2 = SRCNNModel()
    m2.setup_model(from_pretrained=True, pretrained_path=path)
    assert np.array_equal(np.asarray(m2.model.predict(x)), got)
    with pytest.raises(ValueError):
        m.save(str(tmp_path), "x", fmt="pkl")
    g = ESRGAN(compute_dtype="f32")
    g.setup_model(scale_factor=2, growth_channels=8, num_rrdb_blocks=1)
    g.trained = True
    gp = g.save(str(tmp_path), "t1", fmt="h5")
    assert os.path.basename(gp) == "ESRGAN_generator_x2_t1.h5"
    g2 = ESRGAN(compute_dtype="f32")
    g2.setup_model(scale_factor=2, from_trained=True, generator_pretrained_path=gp)
    assert g2.num_rrdb_blocks == 1 and all(np.array_equal(g2.weights[n][0], g.weights[n][0]) for n in g.weights)


def test_cfg2_full_size_batch_properties(ctx):
    """BASELINE configs[2] at its full size -- ESRGAN x4, NB = 23, G = 32, both SelfAttention layers, bf16, sixteen 512 x 512 LR tiles in
    reference patch mode = 7056 patches 48 x 48 through one set of launches -- checked through properties that do not need the CPU oracle at
    this size (the oracle pins the same graph on 36 patches in test_full_depth_generator_bf16_tracks_fp32 and stage by stage in
    tests/test_full_depth_gpu.py): every tile's image from the 16-tile call is BIT FOR BIT the image of that tile processed on its own (441
    patches: other launch shapes, other workgroup ranges of the fused kernels' row stream, other sub-batches everywhere), replicated tiles give
    replicated images, shapes / range / finiteness, and the second call reproduces the first."""
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from sr355.weights import condition_attention, init_weights
    m = ESRGAN(compute_dtype="bf16")
    m.setup_model(scale_factor=4, growth_channels=32, num_rrdb_blocks=23, use_attention=True)
    m.set_weights(bf16_rounded(condition_attention(init_weights(m.generator.layer_shapes(), seed=3000))))
    lr4, hr4 = make_pairs(4, 512, 512, 4, seed=44)
    lr = ctx.to_device(np.stack([lr4[t % 4] for t in range(16)]))
    srs, _ = m.super_resolve_images([lr[t] for t in range(16)], patch_size_lr=48, stride=24, batch_size=441 * 16, timed=False)
    assert len(srs) == 16
    for sr in srs:
        assert tuple(sr.shape) == (2048, 2048, 3) and bool(torch.isfinite(sr).all()) and float(sr.min()) >= 0.0 and float(sr.max()) <= 1.0
    for t in range(4, 16):
        assert torch.equal(srs[t], srs[t % 4]), t                       # the batch holds each synthetic tile four times
    for t in (0, 1, 2, 3):
        alone = m.super_resolve_image(lr[t], patch_size_lr=48, stride=24, batch_size=441)[0]
        assert torch.equal(alone, srs[t]), (t, float((alone - srs[t]).abs().max()))
    again, _ = m.super_resolve_images([lr[t] for t in range(16)], patch_size_lr=48, stride=24, batch_size=441 * 16, timed=False)
    assert all(torch.equal(a, b) for a, b in zip(again, srs))
    p = [float(ctx.psnr(ctx.to_device(hr4[t:t + 1]), srs[t][None])[0]) for t in range(4)]
    assert all(np.isfinite(p)), p
    m.generator.release_workspace()
