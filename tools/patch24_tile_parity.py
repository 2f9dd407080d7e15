#!/usr/bin/env python3
"""Whole 512 x 512 bench tiles at the reference's TRAINING patch size -- super_resolve_image(patch_size_lr=24, stride=12): 1681 overlapping patches per tile,
ESRGAN_model.py:858-979 -- on the trained-like weights of sr355.recipes (60 L1 steps: 28-32 dB against HR): the bf16 path, whose dense blocks then run on the
fused kernels with two 24-pixel-wide patches per 48-pixel row (csrc/api.hip pack2, round 4), against the fp32 device path (pinned to the CPU oracle's fp32 graph by
tests/test_trained_like_gpu.py) and against the bf16 path with the packing off (layer-by-layer tile kernels).  The north star's figure, |PSNR(a, HR) - PSNR(b, HR)|,
per tile.  Prints one JSON object; kept in profiles/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import numpy as np
import torch

import bench as B


def main():
    from oracle import ops as OO
    from sr355 import Context
    from sr355.recipes import GeneratorPixelFit, crop_batches, near_identity_generator
    from sr355.synth import make_pairs
    from sr355.weights import bf16_rounded
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    ctx = Context.get(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else B.TRAINED_LIKE_LEVELS[0]
    lr4, hr4 = make_pairs(4, B.LR, B.LR, B.SCALE, seed=44)
    models = {}
    for dt in ("bf16", "f32"):
        m = ESRGAN(compute_dtype=dt)
        m.setup_model(scale_factor=B.SCALE, growth_channels=B.G, num_rrdb_blocks=B.NB)
        models[dt] = m
    fit = GeneratorPixelFit(ctx, near_identity_generator(models["f32"].generator.layer_shapes()), B.SCALE, B.NB, True, 2e-4)
    for x, y in crop_batches(lr4, hr4, B.SCALE, 24, 16, steps, 7001):
        fit.step(x, y)
    w = bf16_rounded(fit.weights)
    for m in models.values():
        m.set_weights(w)
    out = {"fit_steps": steps, "patch_size_lr": 24, "stride": 12, "tiles": []}
    for t in range(len(lr4)):
        hr = hr4[t].astype(np.float64)
        row = {"tile": t}
        for name, dt, mask in (("bf16_two_up_fused", "bf16", ctx.FUSED_ALL), ("bf16_layer_by_layer", "bf16", ctx.FUSED_ALL & ~35), ("f32", "f32", ctx.FUSED_ALL)):
            ctx.set_fused(mask, 0)
            ctx.profile_begin()
            t0 = time.perf_counter()
            sr, _ = models[dt].super_resolve_image(lr4[t], patch_size_lr=24, stride=12, batch_size=1681)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            ks = {r["kernel"] for r in ctx.profile_end()}
            row[name] = {"psnr_vs_hr_db": float(OO.psnr(hr, sr.astype(np.float64), dtype=np.float64)), "seconds": dt_s,
                         "fused_dense_kernels": sorted(k for k in ks if k.startswith("dense_"))}
        ctx.set_fused(ctx.FUSED_ALL, 0)
        row["abs_delta_two_up_vs_f32_db"] = abs(row["bf16_two_up_fused"]["psnr_vs_hr_db"] - row["f32"]["psnr_vs_hr_db"])
        row["abs_delta_layer_by_layer_vs_f32_db"] = abs(row["bf16_layer_by_layer"]["psnr_vs_hr_db"] - row["f32"]["psnr_vs_hr_db"])
        out["tiles"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    out["max_abs_delta_two_up_vs_f32_db"] = max(r["abs_delta_two_up_vs_f32_db"] for r in out["tiles"])
    out["north_star_bar_db"] = 0.01
    print(json.dumps(out))


if __name__ == "__main__":
    main()
