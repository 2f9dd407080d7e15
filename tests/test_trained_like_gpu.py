"""bf16 parity where it can fail: on a generator whose output RESEMBLES the HR image.

Random-init weights give PSNR(SR, HR) = 9.6 dB: against an MSE of 0.11 no storage noise moves the figure |PSNR(gpu, HR) - PSNR(cpu, HR)|.
The reference's generators are trained (28.8-31.3 dB, ESRGAN.ipynb:L3723-3725).  sr355.recipes builds, from seeds only, a weight set for
the bench graph (x4, NB 23, G 32, both SelfAttention layers) that reaches 34-36 dB on the bench's synthetic tiles; here the bf16 device path
is held to the north star's 0.01 dB against the CPU oracle's fp32 graph on the bench's 16 parity patches and on a whole 512 x 512 tile in
reference patch mode (ESRGAN_model.py:858-979)."""
import numpy as np
import pytest
import torch

import bench as B
from oracle import models as OM
from oracle import ops as OO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitted(ctx):
    from sr355.synth import make_pairs
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    lr4, hr4 = make_pairs(4, B.LR, B.LR, B.SCALE, seed=44)
    m = ESRGAN(compute_dtype="bf16")
    m.setup_model(scale_factor=B.SCALE, growth_channels=B.G, num_rrdb_blocks=B.NB)
    out, w = B.trained_like_parity(ctx, m, lr4, hr4)
    yield out, w, m, lr4, hr4
    m.generator.release_workspace()


LEVELS = [f"steps_{n}" for n in B.TRAINED_LIKE_LEVELS]


def test_recipe_reaches_the_trained_regime(fitted):
    for lv in LEVELS:
        out = fitted[0][lv]
        lo, hi = out["psnr_fp32_reference_graph_vs_hr_db"]
        print(f"\n{lv}: PSNR(fp32 oracle, HR) on the 16 parity patches: {lo:.2f} .. {hi:.2f} dB; whole tiles: "
              + ", ".join(f"{t['psnr_f32_vs_hr_db']:.2f}" for t in out["whole_tiles_patch_mode"]))
        assert lo >= 25.0, lo                                               # the verdict's bar; the reference's own models: 28.8-31.3 dB
        assert min(t["psnr_f32_vs_hr_db"] for t in out["whole_tiles_patch_mode"]) >= 25.0


def test_bf16_parity_on_the_16_bench_patches(fitted):
    """North star: |PSNR(gpu, HR) - PSNR(cpu reference, HR)| <= 0.01 dB, per patch, in the regime the reference's models work in (first
    level).  What bounds it is the bf16 path's noise floor against the fp32 graph, ~68 dB on these weights (five serial bf16 operand
    roundings of the image-carrying tensors behind the trunk): at 30 dB that is 1.6e-4 of the image's own MSE; at
    the stress level (33-37 dB) 1e-3, where a 6 % correlation between the two errors is already worth 0.01 dB -- the bound there is the
    floor itself and twice the bar."""
    for lv in LEVELS:
        out = fitted[0][lv]
        print(f"\n{lv}: |dPSNR vs HR| bf16 device vs fp32 oracle: {out['abs_psnr_delta_vs_hr_db']:.5f} dB (mean {out['mean_psnr_delta_vs_hr_db']:+.5f}); "
              f"PSNR(gpu, fp32 oracle) {out['psnr_gpu_vs_fp32_reference_graph_db']:.2f} dB; PSNR(gpu, bf16-storage oracle) {out['psnr_gpu_vs_oracle_db']:.2f} dB")
        assert out["psnr_gpu_f32_vs_fp32_reference_graph_db"] >= 100.0      # the fp32 device path IS the oracle's graph (pins the whole-tile figures)
        assert out["psnr_gpu_vs_fp32_reference_graph_db"] >= 66.0           # the floor
        assert out["psnr_gpu_vs_oracle_db"] >= 70.0                          # like for like: what is left is accumulation order
        assert out["abs_psnr_delta_vs_hr_db_bf16_storage_oracle"] <= 0.005
    assert fitted[0][LEVELS[0]]["abs_psnr_delta_vs_hr_db"] <= 0.01
    assert fitted[0][LEVELS[1]]["abs_psnr_delta_vs_hr_db"] <= 0.02


def test_bf16_parity_on_whole_tiles_in_patch_mode(fitted):
    """Every bench tile whole (441 overlapping patches, averaged: ESRGAN_model.py:903-921): <= 0.01 dB in the reference's regime; at the stress level the
    bound is the noise floor's again (36.8 dB against a 68 dB floor is a ratio of 7.6e-4: 0.003 dB uncorrelated, 0.013 measured on one tile of one build --
    the fit's weights, and with them this figure, move with every change of the fp32 training kernels' summation order)."""
    for lv, bar in zip(LEVELS, (0.01, 0.02)):
        out = fitted[0][lv]
        for t in out["whole_tiles_patch_mode"]:
            print(f"\n{lv} tile {t['tile']}: PSNR vs HR bf16 {t['psnr_bf16_vs_hr_db']:.4f} dB, fp32 {t['psnr_f32_vs_hr_db']:.4f} dB, |delta| {t['abs_delta_db']:.5f}")
        assert out["whole_tile_abs_psnr_delta_vs_hr_db"] <= bar


def test_recipe_is_deterministic(ctx):
    """Seeds only: two runs of the fit's first steps give the same weights bit for bit (fixed-order reductions on the device)."""
    from sr355.recipes import trained_like_generator
    from sr355.synth import make_pairs
    lr4, hr4 = make_pairs(2, 64, 64, 4, seed=3)
    layers = OM.esrgan_g_layers(4, 32, 2)
    a = trained_like_generator(ctx, layers, lr4, hr4, 4, 2, steps=5)
    b = trained_like_generator(ctx, layers, lr4, hr4, 4, 2, steps=5)
    assert all(np.array_equal(a[n][0], b[n][0]) and np.array_equal(a[n][1], b[n][1]) for n in a)
    w0 = trained_like_generator(ctx, layers, lr4, hr4, 4, 2, steps=0)
    assert any(not np.array_equal(a[n][0], w0[n][0]) for n in a)


def test_set_weights_reaches_a_live_trainer(ctx):
    """ESRGAN.set_weights after the trainer exists must move the trainer's device-resident parameter bucket too."""
    from sr355.gan_train import ESRGANTrainer
    from sr355.weights import init_weights
    layers = OM.esrgan_g_layers(2, 8, 1)
    w0, w1 = init_weights(layers, seed=1), init_weights(layers, seed=2)
    tr = ESRGANTrainer(ctx, w0, None, None, 2, 1, attention=True)
    x = np.random.default_rng(0).uniform(-1, 1, (2, 12, 12, 3)).astype(np.float32)
    y = np.random.default_rng(1).uniform(-1, 1, (2, 24, 24, 3)).astype(np.float32)
    tr.pixel_step(x, y)
    tr.gw = w1                                                               # the round-2 spelling: a plain attribute then, a setter now
    assert tr.g_opt.t == 0 and float(tr.g_opt.m.abs().max()) == 0.0
    flat = np.concatenate([a.ravel() for pair in w1.values() for a in pair])
    assert np.array_equal(tr._gflat.cpu().numpy(), flat)
    l1 = tr.pixel_step(x, y)
    tr2 = ESRGANTrainer(ctx, w1, None, None, 2, 1, attention=True)
    assert l1 == tr2.pixel_step(x, y)
    assert all(np.array_equal(tr.gw[n][0], tr2.gw[n][0]) for n in w1)
    with pytest.raises(RuntimeError):
        tr.train_step(x, y)
