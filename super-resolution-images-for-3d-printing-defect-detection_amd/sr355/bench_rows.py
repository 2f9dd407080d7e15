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
