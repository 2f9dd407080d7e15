#!/usr/bin/env python3
"""Builder's probe for sr355.recipes: run the recipe on the GPU, print PSNR vs HR of the fp32 and the bf16 generator on the bench tiles
(patch mode) along the fit, optionally save the weights (bf16-rounded, uint16) for CPU-side oracle experiments.

    python tools/fit_trained_like.py --steps 600 --lr 2e-4 --patch 24 --batch 16 --out gpurun_out/tl.npz
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--patch", type=int, default=24)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--eval-every", type=int, default=200)
    ap.add_argument("--nb", type=int, default=23)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from sr355 import Context
    from sr355.recipes import GeneratorPixelFit, crop_batches, near_identity_generator
    from sr355.synth import make_pairs
    from sr355.weights import bf16_rounded
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    ctx = Context.get(0)
    lr4, hr4 = make_pairs(4, 512, 512, 4, seed=44)
    models = {}
    for dt in ("f32", "bf16"):
        m = ESRGAN(compute_dtype=dt)
        m.setup_model(scale_factor=4, growth_channels=32, num_rrdb_blocks=args.nb)
        models[dt] = m
    w = near_identity_generator(models["f32"].generator.layer_shapes())
    hr_d = ctx.to_device(hr4)

    def evaluate(w, tag):
        row = {"tag": tag}
        wb = bf16_rounded(w)
        for dt, m in models.items():
            m.set_weights(wb)
            ps = []
            for t in range(4):
                sr, _ = m.super_resolve_image(ctx.to_device(lr4[t]), patch_size_lr=48, stride=24, batch_size=441)
                ps.append(float(ctx.psnr(hr_d[t:t + 1], sr[None])[0]))
            row[dt] = [round(p, 4) for p in ps]
        row["delta"] = [round(abs(a - b), 4) for a, b in zip(row["f32"], row["bf16"])]
        print(json.dumps(row), flush=True)

    evaluate(w, "init")
    fit = GeneratorPixelFit(ctx, w, 4, args.nb, True, args.lr)
    t0 = time.perf_counter()
    for i, (x, y) in enumerate(crop_batches(lr4, hr4, 4, args.patch, args.batch, args.steps, 7001)):
        l1 = fit.step(x, y)
        if i % 50 == 0:
            torch.cuda.synchronize()
            print(f"step {i} l1 {l1:.5f}  {1e3 * (time.perf_counter() - t0) / (i + 1):.1f} ms/step", flush=True)
        if (i + 1) % args.eval_every == 0:
            evaluate(fit.weights, f"step {i + 1}")
    if args.out:
        from sr355.weights import round_to_bf16
        wts = fit.weights
        flat = {}
        for n, (k, b) in wts.items():
            flat[n + "/kernel"] = (round_to_bf16(k).view(np.uint32) >> 16).astype(np.uint16)
            flat[n + "/bias"] = b
        np.savez(args.out, **flat)


if __name__ == "__main__":
    main()
