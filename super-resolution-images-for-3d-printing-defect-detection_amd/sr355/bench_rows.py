"""The BASELINE rows beyond configs[2], as functions: bench.py appends them to its JSON line (N = 1, outside the timed region of the headline)
and tools/bench_train.py / tools/bench_e2e.py print them on their own.  Synthetic data and seeded weights, as everywhere in the benches."""
import time

import numpy as np
import torch


def cfg3_train_step(ctx, steps=3, batch=16, allreduce=None, allreduce_flat=None, seed_offset=0):
    """BASELINE configs[3]: ESRGAN._train_step (ESRGAN_model.py:475-533) at the reference's defaults -- x4, NB = 23, G = 32, both SelfAttention
    layers, `batch` LR patches 24 x 24 -> 96 x 96 per GPU, fp32 -- on sr355.gan_train.ESRGANTrainer.  -> dict (ms per step, losses, ...)."""
    from .gan_train import ESRGANTrainer
    from .runtime import Model
    from .weights import condition_attention, init_weights
    g = Model("esrgan_g", compute_dtype="f32", scale_factor=4, num_blocks=23, growth_channels=32, use_attention=True, ctx=ctx)
    d = Model("esrgan_d", compute_dtype="f32", ctx=ctx)
    v = Model("vgg19_features", compute_dtype="f32", ctx=ctx)
    gw = condition_attention(init_weights(g.layer_shapes(), seed=3000))
    # glorot-initialised RRDBs have gain ~1.2 per block: 23 of them in fp32 stay finite but the losses would be astronomically large;
    # scale the residual branches' last convs so that the step's numbers are ordinary (timing does not depend on the values)
    gw = {n: ((k * 0.1, b * 0.1) if n.endswith("_conv5") else (k, b)) for n, (k, b) in gw.items()}
    dw = init_weights(d.layer_shapes(), seed=5000)
    vw = init_weights(v.layer_shapes(), scheme="he_normal", seed=6000)
    vw = {n: (k * 0.05 if n == "block1_conv1" else k, b) for n, (k, b) in vw.items()}
    del g, d, v
    tr = ESRGANTrainer(ctx, gw, dw, vw, 4, 23, attention=True, allreduce=allreduce, allreduce_flat=allreduce_flat)
    rng = np.random.default_rng(42 + 3 + seed_offset)
    lr = rng.uniform(-1, 1, (batch, 24, 24, 3)).astype(np.float32)
    hr = rng.uniform(-1, 1, (batch, 96, 96, 3)).astype(np.float32)
    out = tr.train_step(lr, hr)                                      # warm-up
    torch.cuda.synchronize(ctx.torch_device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.train_step(lr, hr)
    torch.cuda.synchronize(ctx.torch_device)
    wall = time.perf_counter() - t0
    count = lambda w: sum(int(np.prod(k.shape)) + int(np.prod(b.shape)) for k, b in w.values())
    return {"row": "cfg3 ESRGAN _train_step", "batch_per_gpu": batch, "lr_patch": 24, "scale": 4, "num_rrdb": 23, "growth_channels": 32, "dtype": "f32",
            "steps": steps, "ms_per_step": 1e3 * wall / steps, "patches_per_s_per_gpu": batch * steps / wall, "generator_params": count(gw),
            "discriminator_params": count(dw), "gradient_bucket_mb": 4e-6 * (count(gw) + count(dw)),
            "losses_finite": bool(all(np.isfinite(float(x)) for x in out.values())), "losses": {k: float(x) for k, x in out.items()}, "wall_s": wall}


def cfg4_streaming(ctx, n_frames=4, rank=0, world=1, generator=None):
    """BASELINE configs[4]: streaming 1080p frames -> ESRGAN x4 (NB = 23, G = 32, both SelfAttention layers, bf16, reference patch mode: 3600
    patches per frame) -> VGG16 defect vote on 14 400 patches 96 x 96 of the SR frame, device resident (sr355.pipeline.stream_sr_classify).
    `generator`: an ESRGAN wrapper to reuse (bench.py hands over its own); else one is built."""
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    from SRModels.defect_detection_models.VGG16_model import FineTunedVGG16
    from .pipeline import stream_sr_classify
    from .synth import hr_tile
    from .weights import bf16_rounded, condition_attention, init_weights
    g = generator
    if g is None:
        g = ESRGAN(compute_dtype="bf16")
        g.setup_model(scale_factor=4, growth_channels=32, num_rrdb_blocks=23)
        g.set_weights(bf16_rounded(condition_attention(init_weights(g.generator.layer_shapes(), seed=3000))))
    c = FineTunedVGG16(compute_dtype="bf16")
    c.setup_model(input_shape=(96, 96, 3), num_classes=2)
    c.set_weights(c.weights)
    rng = np.random.default_rng(42 + 4)
    base = [(hr_tile(rng, 1080, 1920) * 255).astype(np.uint8) for _ in range(2)]
    frames = [base[i % 2] for i in range(n_frames * world)]
    kw = dict(patch_size_lr=48, stride=24, batch_size=3600)
    stream_sr_classify(g, c, frames[:1], sr_kwargs=kw, batch_size=2048)                 # warm-up: workspaces, first-touch
    res, stats = stream_sr_classify(g, c, frames, sr_kwargs=kw, batch_size=2048, rank=rank, world=world)
    wall = stats["wall_s"]
    return {"row": "cfg4 streaming SR -> classifier", "frames": len(frames), "frame": "1080x1920 uint8 RGB (LR input)", "frames_per_s": len(frames) / wall,
            "sr_output_mpix_per_s": len(frames) * 4320 * 7680 / 1e6 / wall, "ms_per_frame": 1e3 * wall / len(frames) * world, "patches_sr_per_frame": 3600,
            "patches_classifier_per_frame": 14400, "host_ms_per_frame_sr": stats["host_ms_per_frame_sr_enqueue_plus_wait"],
            "host_ms_per_frame_classify": stats["host_ms_per_frame_classify"], "votes": [(r["class"], round(r["confidence"], 4)) for r in res[:4]], "wall_s": wall}


# ------------------------------------------------------------------------------------------------------------------------------------
# Round 4: every BASELINE config and the section-8a rows as driver-visible numbers (bench.py appends them to its line's `rows`, outside
# the headline's timed region).  Each row: device-resident synthetic inputs, seeded weights, median of a few repetitions between torch
# events on the launching stream, and -- where a roof applies -- its own fraction of it.
PEAK_F32_TFLOPS, PEAK_BF16_TFLOPS, PEAK_HBM_GBPS = 157.0, 2500.0, 8000.0


def _median_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def cfg0_bicubic_metrics(ctx):
    """BASELINE configs[0]: bicubic x4 on one 64 x 64 -> 256 x 256 tile + metrics.py PSNR / SSIM (classic_algorithms.py:11-13, metrics.py:3-7), uint8 and
    float32; the same three calls on the cfg1-sized batch beside it (one tile is a launch-latency measurement)."""
    from .synth import make_pairs
    lr, hr = make_pairs(1, 64, 64, 4, seed=42)
    x, h = ctx.to_device(lr), ctx.to_device(hr)
    xu = ctx.to_device((lr * 255).round().astype(np.uint8), torch.uint8)
    up = ctx.bicubic(x, 256, 256)
    row = {"row": "cfg0 bicubic x4 64x64 -> 256x256 + PSNR / SSIM", "bicubic_f32_us": 1e3 * _median_ms(lambda: ctx.bicubic(x, 256, 256)),
           "bicubic_u8_us": 1e3 * _median_ms(lambda: ctx.resize(xu, 256, 256, "INTER_CUBIC")), "psnr_us": 1e3 * _median_ms(lambda: ctx.psnr(h, up)),
           "ssim_us": 1e3 * _median_ms(lambda: ctx.ssim(h, up)), "psnr_db": float(ctx.psnr(h, up.clamp(0, 1))[0]), "ssim": float(ctx.ssim(h, up.clamp(0, 1))[0])}
    xb = torch.rand(32, 256, 256, 3, device=ctx.torch_device)
    ms = _median_ms(lambda: ctx.bicubic(xb, 1024, 1024))
    row["batch_32x256_to_1024_ms"] = ms
    row["batch_gbps"] = (xb.numel() + 32 * 1024 * 1024 * 3) * 4 / ms / 1e6
    row["batch_frac_of_hbm_peak"] = row["batch_gbps"] / PEAK_HBM_GBPS
    a = torch.rand(16, 2048, 2048, 3, device=ctx.torch_device)
    b = (a + 0.01 * torch.randn_like(a)).clamp(0, 1)
    for name, fn in (("psnr", ctx.psnr), ("ssim", ctx.ssim)):
        ms = _median_ms(lambda: fn(a, b), reps=3, warm=1)
        row[f"{name}_16x2048_ms"] = ms
        row[f"{name}_16x2048_frac_of_hbm_peak"] = 2 * a.numel() * 4 / ms / 1e6 / PEAK_HBM_GBPS
    return row


def cfg1_srcnn(ctx, batch=32):
    """BASELINE configs[1]: SRCNN 3-layer x4 inference, LR [32, 256, 256, 3] fp32 -> bicubic x4 (SRCNN_model.py:191) -> conv 9x9 / 1x1 / 5x5
    (:48-53) -> [32, 1024, 1024, 3], the whole batch in one forward.  1.933 TFLOP per batch against the 157 TFLOP/s fp32 MFMA peak."""
    from .runtime import Model
    from .weights import init_weights
    m = Model("srcnn", compute_dtype="f32", ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), seed=1000))
    lr = torch.rand(batch, 256, 256, 3, device=ctx.torch_device)
    out = ctx.empty((batch, 1024, 1024, 3))

    def step():
        m.forward(ctx.bicubic(lr, 1024, 1024), out=out)
    ms = _median_ms(step, reps=3, warm=1)
    ctx.profile_begin()
    step()
    torch.cuda.synchronize(ctx.torch_device)
    prof = ctx.profile_end()
    m.release_workspace()
    tflop = batch * 1048576 * 57600 / 1e12
    return {"row": "cfg1 SRCNN x4 fp32", "lr_batch": [batch, 256, 256, 3], "ms_per_batch": ms, "mpix_per_s": batch * 1.048576 / ms * 1e3, "tflop_per_batch": tflop,
            "tflops": tflop / ms * 1e3, "frac_of_f32_mfma_peak": tflop / ms * 1e3 / PEAK_F32_TFLOPS, "dtype": "f32",
            "kernels_ms": {r["kernel"]: round(r["total_ms"], 3) for r in prof}}


def edsr_x4(ctx, patches=441):
    """Row a4: EDSR x4 (16 residual blocks, 64 filters, EDSR_model.py:55-125) on one tile's 441 LR patches 48 x 48, bf16."""
    from .runtime import Model
    from .weights import init_weights
    m = Model("edsr", compute_dtype="bf16", scale_factor=4, num_blocks=16, num_filters=64, res_scaling=0.1, ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), scheme="he_normal", seed=2000))
    x = torch.rand(patches, 48, 48, 3, device=ctx.torch_device)
    ms = _median_ms(lambda: m.forward(x), reps=3, warm=1)
    m.release_workspace()
    tflop = patches * 2304 * 2 * 1983168 / 1e12
    return {"row": "a4 EDSR x4 bf16", "lr_patches": [patches, 48, 48, 3], "ms": ms, "out_mpix_per_s": patches * 192 * 192 / 1e6 / ms * 1e3, "tflops": tflop / ms * 1e3,
            "frac_of_bf16_mfma_peak": tflop / ms * 1e3 / PEAK_BF16_TFLOPS}


def vgg16_patches(ctx, patches=1024):
    """Row a8: the VGG16 defect classifier (VGG16_model.py:57-97) on 1024 patches 96 x 96, bf16: 2.819 GMAC per patch."""
    from .runtime import Model
    from .weights import init_weights
    m = Model("vgg16", compute_dtype="bf16", num_classes=2, ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), scheme="he_normal", seed=4000))
    x = torch.rand(patches, 96, 96, 3, device=ctx.torch_device)
    ms = _median_ms(lambda: m.forward(x), reps=3, warm=1)
    m.release_workspace()
    tflop = patches * 2 * 2.819e9 / 1e12
    return {"row": "a8 VGG16 classifier bf16", "patches": [patches, 96, 96, 3], "ms": ms, "patches_per_s": patches / ms * 1e3, "tflops": tflop / ms * 1e3,
            "frac_of_bf16_mfma_peak": tflop / ms * 1e3 / PEAK_BF16_TFLOPS}


def whole_tile(ctx, weights, tiles=2):
    """SURVEY.md 8(d)'s secondary row: the literal whole-tile forward G([2, 512, 512, 3]) of the bench generator, bf16 -- SelfAttention over
    N = 262 144 and 1 048 576 tokens, which the reference never leaves patch mode for (it materialises N x N).  Conv roofline from 9.412 TFLOP per tile."""
    from .runtime import Model
    from .synth import make_pairs
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=4, num_blocks=23, growth_channels=32, use_attention=True, ctx=ctx)
    m.set_weights(weights)
    lr, _ = make_pairs(tiles, 512, 512, 4, seed=44)
    x = ctx.to_device(lr * 2 - 1)
    m.forward(x)
    ms = _median_ms(lambda: m.forward(x), reps=2, warm=0)
    ctx.profile_begin()
    m.forward(x)
    torch.cuda.synchronize(ctx.torch_device)
    prof = ctx.profile_end()
    m.release_workspace()
    att = sum(r["total_ms"] for r in prof if r["kernel"].startswith("attn"))
    return {"row": "whole-tile forward G([2,512,512,3]) bf16 (non-reference mode)", "tiles": tiles, "ms": ms, "ms_per_tile": ms / tiles, "mpix_per_s": tiles * 4.194304 / ms * 1e3,
            "attention_ms": att, "conv_tflops_without_attention_time": tiles * 9.412 / max(ms - att, 1e-6) * 1e3,
            "conv_frac_of_bf16_mfma_peak_whole_forward": tiles * 9.412 / ms * 1e3 / PEAK_BF16_TFLOPS}


def esrgan_conv_mac_per_lr_pixel(scale, nb, g):
    """conv multiply-accumulates per LR pixel of the generator graph (ESRGAN_model.py:303-345; SURVEY.md Appendix B: 17 952 448 at x4 / NB 23 / G 32),
    the 1x1 convs of the two SelfAttention layers included, their score / value products not."""
    dense = sum(9 * (64 + k * g) * g for k in range(4)) + 9 * (64 + 4 * g) * 64
    mac = 9 * 3 * 64 + nb * 3 * dense + 9 * 64 * 64 + (64 * 48 + 32 * 64) * (1 + 4)
    px, s = 1, scale
    while s > 1:
        mac += px * 9 * 64 * 256
        px *= 4
        s >>= 1
    return mac + px * (9 * 64 * 64 + 9 * 64 * 3)


def generator_shapes(ctx, cases=((2, 4, 8, 24, 7056), (4, 23, 32, 24, 7056), (4, 23, 32, 96, 441), (4, 23, 8, 48, 1764))):
    """The generator at patch sizes / growth widths other than the bench's: the reference's OWN configuration
    (x2, 4 RRDBs, 8 growth channels, 24 x 24 LR patches: ESRGAN.ipynb:L758-761, constants.py:6-10) and the default graph at patch_size_lr 24 and 96
    (ESRGAN_model.py:858 takes it as an argument).  The fused dense-block kernels need G = 32 and 48-pixel-wide images; elsewhere the tile
    kernels run -- this row prices that.  (scale, NB, G, LR patch, patches) -> ms, output MPix/s, conv TFLOP/s and the fraction of the bf16 peak."""
    from .runtime import Model
    from .weights import condition_attention, init_weights
    out = []
    for scale, nb, g, p, n in cases:
        m = Model("esrgan_g", compute_dtype="bf16", scale_factor=scale, num_blocks=nb, growth_channels=g, use_attention=True, ctx=ctx)
        m.set_weights(condition_attention(init_weights(m.layer_shapes(), seed=3000)))
        x = (torch.rand(n, p, p, 3, device=ctx.torch_device) * 2 - 1).to(torch.bfloat16)
        ms = _median_ms(lambda: m.forward(x), reps=3, warm=1)
        ctx.profile_begin()
        m.forward(x)
        torch.cuda.synchronize(ctx.torch_device)
        prof = ctx.profile_end()
        m.release_workspace()
        del m
        att = sum(r["total_ms"] for r in prof if r["kernel"].startswith("attn"))
        tflop = 2.0 * esrgan_conv_mac_per_lr_pixel(scale, nb, g) * n * p * p / 1e12
        fused = sorted({r["kernel"] for r in prof if r["kernel"].startswith(("dense_", "conv_stream"))})
        out.append({"scale": scale, "num_rrdb": nb, "growth_channels": g, "patch_size_lr": p, "patches": n, "ms": ms, "out_mpix_per_s": n * (p * scale) ** 2 / 1e6 / ms * 1e3,
                    "attention_ms": att, "conv_tflops": tflop / ms * 1e3, "conv_frac_of_bf16_mfma_peak": tflop / ms * 1e3 / PEAK_BF16_TFLOPS,
                    "fused_dense_block_kernels": fused})
    return {"row": "generator at other patch sizes / growth widths (bf16)", "cases": out}


def attention_wide_logits(ctx, patches=1764):
    """attn<bf16> on data that leaves its rescale-free fast path: the bench conditions the query / key projections by 2^-8
    (sr355.weights.condition_attention) so that the group-level bound is always taken; the reference has no 1 / sqrt(d) scale
    (ESRGAN_model.py:61-65) and trained logits may be wide.  Same generator, same patches, raw glorot projections beside the conditioned ones."""
    from .runtime import Model
    from .weights import bf16_rounded, condition_attention, init_weights
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=4, num_blocks=23, growth_channels=32, use_attention=True, ctx=ctx)
    w0 = init_weights(m.layer_shapes(), seed=3000)
    x = (torch.rand(patches, 48, 48, 3, device=ctx.torch_device) * 2 - 1).to(torch.bfloat16)
    row = {"row": "attn<bf16>: conditioned vs raw glorot logits", "patches": patches}
    for name, w in (("conditioned_2^-8", condition_attention(w0)), ("raw_glorot", w0)):
        m.set_weights(bf16_rounded(w))
        m.forward(x)
        ctx.profile_begin()
        for _ in range(2):
            m.forward(x)
        torch.cuda.synchronize(ctx.torch_device)
        prof = ctx.profile_end()
        att = [r for r in prof if r["kernel"].startswith("attn")]
        row[name] = {"attn_ms_per_forward": sum(r["total_ms"] for r in att) / 2, "forward_ms": sum(r["total_ms"] for r in prof) / 2}
    m.release_workspace()
    row["raw_over_conditioned"] = row["raw_glorot"]["attn_ms_per_forward"] / row["conditioned_2^-8"]["attn_ms_per_forward"]
    row["scaled_to_the_bench_step_ms"] = {k: row[k]["attn_ms_per_forward"] * 7056 / patches for k in ("conditioned_2^-8", "raw_glorot")}
    return row
