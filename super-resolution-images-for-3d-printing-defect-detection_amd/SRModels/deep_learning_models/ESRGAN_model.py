"""ESRGAN generator on MI355X behind the reference's class surface
(reference: deep_learning_models/ESRGAN_model.py).

Generator (ESRGAN_model.py:303-345): conv3->64, NB x RRDB (3 dense blocks of 5 convs, residual scale 0.2),
trunk conv + skip, SelfAttention, log2(scale) x (conv64->256, depth_to_space, LeakyReLU 0.2,
SelfAttention after the first), conv64 ReLU, conv->3 tanh.
Discriminator (:347-377) and VGG19 perceptual extractor (:379-408) exist as inference graphs: together with the loss kernels
(pixel L1, FFT-(W,C) spectral L1, MSE of VGG19 features; BCE on the [B,1] discriminator output) they give the generator loss that
`evaluate` reports as avg_g_loss (:782-856).  The GAN training loop itself (_train_step :475-533: backward passes, spectral-norm
updates, Adam) is the next row and is not built.
"""
import os

import numpy as np
import torch

from sr355 import pipeline as P
from sr355.wrappers import DeviceModelMixin, load_pretrained


class ESRGAN(DeviceModelMixin):
    def __init__(self, compute_dtype="bf16"):
        self.generator = None
        self.discriminator = None
        self.vgg_model = None
        self.g_optimizer = None
        self.d_optimizer = None
        self.trained = False
        self.compute_dtype = compute_dtype

    def _mark_trained(self, v):
        self.trained = v

    def setup_model(self, scale_factor=2, growth_channels=32, num_rrdb_blocks=23, input_shape=(None, None, 3),
                    output_shape=(None, None, 3), from_trained=False, generator_pretrained_path=None,
                    discriminator_pretrained_path=None, use_attention=True):
        self.scale_factor = scale_factor
        weights = None
        if from_trained:
            if generator_pretrained_path is None or not os.path.exists(generator_pretrained_path):
                raise FileNotFoundError(f"Generator pretrained path does not exist: {generator_pretrained_path}")
            weights = load_pretrained(generator_pretrained_path)
            growth_channels = int(weights["rrdb_0_dense1_conv1"][0].shape[-1]) if "rrdb_0_dense1_conv1" in weights else growth_channels
            num_rrdb_blocks = len({n.split("_")[1] for n in weights if n.startswith("rrdb_")})
        self.generator = self._make("esrgan_g", self.compute_dtype, scale_factor=scale_factor, channels=int(input_shape[-1]),
                                    num_blocks=num_rrdb_blocks, growth_channels=growth_channels, use_attention=use_attention)
        if weights is not None:
            self.set_weights(weights)
            print(f"- Generator loaded from: {generator_pretrained_path}")
        else:
            self._random_init(seed=3000)
        self._d_path = discriminator_pretrained_path if from_trained else None
        self.d_weights = self.vgg_weights = None

    # ------------------------------------------------------------------ discriminator / VGG19 (inference graphs, built on first use)
    def _ensure_loss_networks(self):
        from sr355.runtime import Model
        from sr355.weights import init_weights
        if self.discriminator is None:
            self.discriminator = Model("esrgan_d", compute_dtype="f32", ctx=self.ctx)
            if self.d_weights is None:
                if self._d_path is not None:
                    if not os.path.exists(self._d_path):
                        raise FileNotFoundError(f"Discriminator pretrained path does not exist: {self._d_path}")
                    self.d_weights = load_pretrained(self._d_path)
                else:
                    self.d_weights = init_weights(self.discriminator.layer_shapes(), seed=5000)
            self.discriminator.set_weights(self.d_weights)
        if self.vgg_model is None:
            self.vgg_model = Model("vgg19_features", compute_dtype="f32", ctx=self.ctx)
            if self.vgg_weights is None:      # keras downloads ImageNet weights here (:388-392); offline: a seeded he-normal stand-in
                self.vgg_weights = init_weights(self.vgg_model.layer_shapes(), scheme="he_normal", seed=6000)
            self.vgg_model.set_weights(self.vgg_weights)

    def set_loss_network_weights(self, discriminator=None, vgg19=None):
        """{layer: (kernel, bias)} for the discriminator (disc_conv1..6, disc_dense1, disc_output: the stored, already spectrally
        normalised kernels) and / or the VGG19 extractor (block1_conv1 .. block5_conv4)."""
        if discriminator is not None:
            self.d_weights = {n: (np.asarray(k, np.float32), np.asarray(b, np.float32)) for n, (k, b) in discriminator.items()}
            self.discriminator = None
        if vgg19 is not None:
            self.vgg_weights = {n: (np.asarray(k, np.float32), np.asarray(b, np.float32)) for n, (k, b) in vgg19.items()}
            self.vgg_model = None

    def generator_loss(self, hr_real, hr_fake):
        """g_loss of _train_step / evaluate for given generator output (ESRGAN_model.py:511-523, :812-826), on the device:
        BCE(1, D(fake)) + 1.0 * MSE(VGG19(real), VGG19(fake)) + 100.0 * mean|real - fake| + 1.0 * spectral.
        hr_real, hr_fake: [B,H,W,3] in [-1,1] (NumPy or device tensors).  -> (g_loss float, parts dict)."""
        self._ensure_loss_networks()
        real = self.ctx.to_device(hr_real, torch.float32) if not isinstance(hr_real, torch.Tensor) else hr_real.to(torch.float32).contiguous()
        fake = self.ctx.to_device(hr_fake, torch.float32) if not isinstance(hr_fake, torch.Tensor) else hr_fake.to(torch.float32).contiguous()
        d_fake = self.discriminator.forward(fake).cpu().numpy().astype(np.float64)
        eps = 1e-7                                            # keras.backend.binary_crossentropy on probabilities
        p = np.clip(d_fake, eps, 1.0 - eps)
        adv = float(np.mean(-np.log(p + eps)))                # target = ones
        perc = float(self.ctx.mse(self.vgg_model.forward(real), self.vgg_model.forward(fake)).item())
        pix = float(self.ctx.l1(real, fake).item())
        spec = float(self.ctx.spectral_l1(real, fake).item())
        parts = {"adversarial": adv, "perceptual": perc, "pixel": pix, "spectral": spec}
        return adv + 1.0 * perc + 100.0 * pix + 1.0 * spec, parts

    def fit(self, *args, **kwargs):
        raise NotImplementedError("ESRGAN adversarial training (_train_step: backward passes, spectral-norm updates, Adam) is the next row "
                                  "(SURVEY.md 8f-1); its forward halves -- discriminator, VGG19 extractor, the four loss terms -- are "
                                  "built: see generator_loss()")

    def generate(self, lr_batch):
        """generator(lr, training=False) on a [-1,1] batch (ESRGAN_model.py:810); NumPy or device tensor."""
        if self.generator is None:
            raise RuntimeError("Generator is not initialized.")
        return self.generator.predict(lr_batch, batch_size=64)

    def evaluate(self, test_dataset):
        """ESRGAN.evaluate (ESRGAN_model.py:782-856): iterable of ([-1,1] LR batch, [-1,1] HR batch); per batch the generator
        loss (same formula as training, discriminator and VGG19 at inference) and the means of PSNR / SSIM; the three results are
        means of the per-batch values (the reference's weighting, last partial batch included)."""
        if not self.trained:
            raise RuntimeError("Model has not been trained.")
        tp = ts = tg = 0.0
        nb = 0
        for lr_batch, hr_batch in test_dataset:
            gen = self.ctx.to_device(self.generate(np.asarray(lr_batch, np.float32)))
            real = self.ctx.to_device(np.asarray(hr_batch, np.float32))
            tg += self.generator_loss(real, gen)[0]
            g01, r01 = (gen + 1.0) / 2.0, (real + 1.0) / 2.0
            tp += float(self.ctx.psnr(r01, g01).mean().item())
            ts += float(self.ctx.ssim(r01, g01).mean().item())
            nb += 1
        metrics = {"avg_psnr": tp / nb, "avg_ssim": ts / nb, "avg_g_loss": tg / nb}
        print("Evaluation Results:")
        print(f"  Average PSNR: {metrics['avg_psnr']:.4f}")
        print(f"  Average SSIM: {metrics['avg_ssim']:.4f}")
        print(f"  Average G Loss: {metrics['avg_g_loss']:.4f}")
        return metrics

    def super_resolve_image(self, lr_img, patch_size_lr=48, stride=24, batch_size=16):
        """Reflect-pad, cut LR patches, [0,1]->[-1,1], generator, (out+1)/2, overlap-average, crop, clip
        (ESRGAN_model.py:858-979).  `batch_size` is Keras' predict chunk (:941) and is honoured as given -- `time_sec` times
        exactly the chunking the caller asked for.  Chunking never changes results; for throughput pass a large batch_size
        (bench.py hands over all patches of its tiles at once)."""
        if not self.trained:
            raise RuntimeError("Model has not been trained or loaded.")
        if self.generator is None:
            raise RuntimeError("Generator is not initialized.")
        if not hasattr(self, "scale_factor") or self.scale_factor is None:
            raise ValueError("scale_factor is not set. Ensure setup_model was called.")
        lr, is_np = P.as_device_image(self.ctx, lr_img)
        sr, metrics = P.patchwise_sr(self.generator, lr, patch_size_lr, stride, self.scale_factor, chunk=max(int(batch_size), 1),
                                     in_mul=2.0, in_add=-1.0, out_mul=0.5, out_add=0.5)
        return (sr.cpu().numpy() if is_np else sr), metrics

    def super_resolve_images(self, lr_imgs, patch_size_lr=48, stride=24, batch_size=16, timed=True):
        """super_resolve_image over several equally sized LR images in one go (not in the reference; same results image
        by image, their patches just share the generator launches).  Returns ([sr_img...], inference_metrics)."""
        if not self.trained:
            raise RuntimeError("Model has not been trained or loaded.")
        conv = [P.as_device_image(self.ctx, im) for im in lr_imgs]
        srs, metrics = P.patchwise_sr_many(self.generator, [c[0] for c in conv], patch_size_lr, stride, self.scale_factor,
                                           chunk=max(int(batch_size), 1), in_mul=2.0, in_add=-1.0, out_mul=0.5, out_add=0.5, timed=timed)
        return [sr.cpu().numpy() if c[1] else sr for sr, c in zip(srs, conv)], metrics

    def save(self, directory, timestamp):
        if not self.trained:
            raise RuntimeError("Cannot save an untrained model.")
        os.makedirs(directory, exist_ok=True)
        generator_path = os.path.join(directory, f"ESRGAN_generator_x{self.scale_factor}_{timestamp}.npz")
        self._save_npz(generator_path)
        print(f"Generator model saved to {generator_path}")
        return generator_path
