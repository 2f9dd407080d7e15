#!/usr/bin/env python3
"""One whole 512 x 512 bench tile in reference patch mode (ESRGAN_model.py:858-979: reflect pad, 441 patches 48 x 48 stride 24, overlap average) through
the CPU ORACLE's fp32 graph, against the bf16 and fp32 device paths, on the trained-like weights of sr355.recipes at both fit levels: the north star's
|PSNR(gpu, HR) - PSNR(cpu reference, HR)| without the fp32 device path standing in for the CPU (tests/test_trained_like_gpu.py pins that stand-in on 16
patches; this script is the long form, ~3 min of CPU per level on the GPU box's 16 threads).  Prints one JSON object; kept in profiles/."""
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
    from oracle import models as OM
    from oracle import ops as OO
    from sr355 import Context
    from sr355.recipes import GeneratorPixelFit, crop_batches, near_identity_generator
    from sr355.synth import make_pairs
    from sr355.weights import bf16_rounded
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ctx = Context.get(0)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    lr4, hr4 = make_pairs(4, B.LR, B.LR, B.SCALE, seed=44)
    models = {}
    for dt in ("bf16", "f32"):
        m = ESRGAN(compute_dtype=dt)
        m.setup_model(scale_factor=B.SCALE, growth_channels=B.G, num_rrdb_blocks=B.NB)
        models[dt] = m
    fit = GeneratorPixelFit(ctx, near_identity_generator(models["f32"].generator.layer_shapes()), B.SCALE, B.NB, True, 2e-4)
    batches = crop_batches(lr4, hr4, B.SCALE, 24, 16, max(B.TRAINED_LIKE_LEVELS), 7001)
    out, done = {"tile": tile, "levels": {}}, 0
    hr = hr4[tile].astype(np.float64)
    for lv in sorted(B.TRAINED_LIKE_LEVELS):
        for _ in range(lv - done):
            fit.step(*next(batches))
        done = lv
        w = bf16_rounded(fit.weights)
        t0 = time.perf_counter()
        ref = OM.esrgan_super_resolve(lr4[tile], w, B.SCALE, B.PATCH, B.STRIDE, num_rrdb=B.NB, dtype=np.float32, chunk=2)     # the CPU reference: fp32 graph, patch mode
        cpu_s = time.perf_counter() - t0
        p_ref = float(OO.psnr(hr, np.clip(ref, 0, 1).astype(np.float64), dtype=np.float64))
        row = {"psnr_cpu_fp32_vs_hr_db": p_ref, "cpu_seconds": cpu_s}
        for dt, m in models.items():
            m.set_weights(w)
            sr, _ = m.super_resolve_image(lr4[tile], patch_size_lr=B.PATCH, stride=B.STRIDE, batch_size=441)
            p = float(OO.psnr(hr, sr.astype(np.float64), dtype=np.float64))
            row[dt] = {"psnr_vs_hr_db": p, "abs_delta_vs_cpu_db": abs(p - p_ref),
                       "psnr_vs_cpu_image_db": float(OO.psnr(np.clip(ref, 0, 1).astype(np.float64), sr.astype(np.float64), dtype=np.float64))}
        out["levels"][f"steps_{lv}"] = row
        print(f"steps_{lv}: {json.dumps(row)}", file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
