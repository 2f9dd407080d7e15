"""ESRGAN generator on MI355X behind the reference's class surface
(reference: deep_learning_models/ESRGAN_model.py).

Generator (ESRGAN_model.py:303-345): conv3->64, NB x RRDB (3 dense blocks of 5 convs, residual scale 0.2),
trunk conv + skip, SelfAttention, log2(scale) x (conv64->256, depth_to_space, LeakyReLU 0.2,
SelfAttention after the first), conv64 ReLU, conv->3 tanh.
Discriminator (:347-377) and VGG19 perceptual extractor (:379-408) exist as inference graphs: together with the loss kernels
(pixel L1, FFT-(W,C) spectral L1, MSE of VGG19 features; BCE on the [B,1] discriminator output) they give the generator loss that
`evaluate` reports as avg_g_loss (:782-856).  The GAN training loop (_train_step :475-533, fit :535-779) runs on sr355.gan_train's
ESRGANTrainer in fp32: backward passes through the C ABI's ops, in-place spectral-norm updates, two Adams with staircase decay.
"""
import os

import numpy as np
import torch

from sr355 import _lib as L
from sr355 import pipeline as P
from sr355.wrappers import DeviceModelMixin, load_pretrained


class ESRGAN(DeviceModelMixin):
    def __init__(self, compute_dtype="f32"):
        """compute_dtype: "f32" (default: what the reference computes in, like SRCNNModel / EDSR / FineTunedVGG16 here; fp32 MFMA,
        >= 100 dB against the fp64 oracle) or "bf16" (opt-in: bf16 storage, fp32 accumulation -- BASELINE configs[2]'s dtype and what
        bench.py passes; ~49 dB against the fp32 graph on the bench patches, |dPSNR vs HR| well under 0.01 dB: INTEGRATION.md)."""
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
        self.num_rrdb_blocks, self.use_attention, self._trainer = int(num_rrdb_blocks), bool(use_attention), None
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
        tr = getattr(self, "_trainer", None)
        if tr is not None:                                 # a live trainer follows (its Adam moments and u vectors stay)
            if discriminator is not None and tr.dw is not discriminator:
                tr.dw = dict(self.d_weights)
            if vgg19 is not None:
                tr.vw = self.vgg_weights

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

    def _ensure_trainer(self):
        """The optimisers live from setup_model on in the reference (ESRGAN_model.py:176-195): one trainer (weights, Adam moments,
        spectral-norm vectors, step counter) per model, kept across fit() calls."""
        from sr355.gan_train import ESRGANTrainer
        if getattr(self, "_trainer", None) is None:
            self._ensure_loss_networks()
            self._trainer = ESRGANTrainer(self.ctx, self.weights, self.d_weights, self.vgg_weights, self.scale_factor,
                                          self.num_rrdb_blocks, attention=self.use_attention, g_lr=1e-4, d_lr=1e-5,
                                          allreduce=getattr(self, "grad_allreduce", None), allreduce_flat=getattr(self, "grad_allreduce_flat", None))
            self.g_optimizer, self.d_optimizer = self._trainer.g_opt, self._trainer.d_opt
        return self._trainer

    def enable_data_parallel(self):
        """One process per GPU (torch.distributed initialised, RCCL): every rank feeds fit() its own shard of each batch and the
        gradients of both networks are averaged as one flat bucket per optimiser step (sr355.dist.allreduce_mean_grads).  Seeded
        initial weights and spectral-norm vectors are identical on every rank, so the replicas stay in step.  Call before fit()."""
        from sr355 import dist as D
        self.grad_allreduce = lambda grads: D.allreduce_mean_grads(grads, device=self.ctx.torch_device)
        self.grad_allreduce_flat = D.allreduce_mean_flat          # the generator's bucket is reduced where it lies (RCCL on the device)
        if getattr(self, "_trainer", None) is not None:
            self._trainer.allreduce, self._trainer.allreduce_flat = self.grad_allreduce, self.grad_allreduce_flat

    def set_weights(self, weights, trained=True):
        """As every wrapper's; a live trainer follows: its device-resident parameter bucket takes the new values and Adam starts over
        (round 3 left the trainer on the old parameters without an error)."""
        super().set_weights(weights, trained)
        tr = getattr(self, "_trainer", None)
        if tr is not None and weights is not tr._gw:
            tr.load_generator_weights(self.weights)

    def _sync_from_trainer(self):
        tr = self._trainer
        self.set_weights(tr.gw)
        self.set_loss_network_weights(discriminator=tr.dw)

    def _train_step(self, lr_batch, hr_batch):
        """ESRGAN_model.py:475-533 on [-1,1] batches -> {'g_loss', 'd_loss', ...} (floats)."""
        return self._ensure_trainer().train_step(lr_batch, hr_batch)

    def fit(self, X_train=None, Y_train=None, train_dataset=None, X_val=None, Y_val=None, val_dataset=None, epochs=100, batch_size=16,
            steps_per_epoch=None, val_steps=None, normalize=True, save_dir=None, shuffle_seed=42):
        """ESRGAN.fit (ESRGAN_model.py:535-779).  (X_train, Y_train) arrays in [0,1] -- reshuffled every epoch, batched, last batch
        partial -- or `train_dataset`, a re-iterable of (lr, hr) batches that is cycled (steps_per_epoch then mandatory).  Per step:
        _train_step, then PSNR / SSIM of generator(lr, training=False) against hr in [0,1], and the two decayed learning rates.
        Validation (arrays or an iterable of batches) reports the generator loss with the discriminator / VGG19 at inference.
        With save_dir a 5x5 grid of generator outputs is written per epoch.  Returns (epoch_losses of the LAST epoch -- per-step lists,
        val_* scalars -- as the reference does, EpochTimeTracker, EpochMemoryTracker)."""
        from sr355.gan_train import Tape, Var, generator_forward, staircase_lr
        from sr355.train import EpochMemoryTracker, EpochTimeTracker
        if train_dataset is None and (X_train is None or Y_train is None):
            raise ValueError("Debe aportar (X_train,Y_train) o un train_dataset")
        if self.generator is None:
            raise RuntimeError("Generator is not initialized.")
        if train_dataset is not None and steps_per_epoch is None:
            raise ValueError("Debe indicar steps_per_epoch cuando aporta un dataset externo")
        print("Training on GPU:", [f"MI355X:{self.ctx.device}"])
        tr = self._ensure_trainer()
        rng = np.random.default_rng(shuffle_seed)
        norm = (lambda a: np.asarray(a, np.float32) * 2.0 - 1.0) if normalize else (lambda a: np.asarray(a, np.float32))
        if train_dataset is None:
            X_train, Y_train = np.asarray(X_train, np.float32), np.asarray(Y_train, np.float32)
            if steps_per_epoch is None:
                steps_per_epoch = int(np.ceil(len(X_train) / batch_size))

            def batches():                                  # from_tensor_slices().shuffle(len).batch(bs).repeat()
                while True:
                    order = rng.permutation(len(X_train))
                    for i in range(0, len(order), batch_size):
                        yield X_train[order[i:i + batch_size]], Y_train[order[i:i + batch_size]]
        else:
            def batches():                                  # dataset.repeat()
                while True:
                    n = 0
                    for pair in train_dataset:
                        n += 1
                        yield pair
                    if n == 0:
                        raise RuntimeError("train_dataset produced no batches")
        stream = batches()
        if val_dataset is not None:
            val_batches = lambda: iter(val_dataset)
        elif X_val is not None and Y_val is not None:
            X_val, Y_val = np.asarray(X_val, np.float32), np.asarray(Y_val, np.float32)
            val_batches = lambda: ((X_val[i:i + batch_size], Y_val[i:i + batch_size]) for i in range(0, len(X_val), batch_size))
            if val_steps is None:
                val_steps = int(np.ceil(len(X_val) / batch_size))
        else:
            val_batches = None
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
        preview = None                                      # (lr batch, already normalised?) fixed across epochs (:616-646)

        def generate_f32(lr_pm1):
            t = tr.generator_tape(wgrad=False)          # multiplies with the trainer's device-resident parameters
            out = generator_forward(t, Var(self.ctx.to_device(np.asarray(lr_pm1, np.float32)), need=False), tr.scale, tr.nb, tr.att).v
            t.ops = []
            return out

        def to01(t):                                       # (x + 1) / 2
            return self.ctx.eltwise(L.ELT_AXPBY, t, torch.ones_like(t), 0.5, 0.5)

        def save_grid(epoch_idx):
            nonlocal preview
            if save_dir is None:
                return
            if preview is None:
                if X_val is not None and len(X_val) > 0:
                    preview = (np.asarray(X_val[:25], np.float32), False)
                elif X_train is not None and len(X_train) > 0:
                    preview = (np.asarray(X_train[:25], np.float32), False)
                else:
                    src = val_batches() if val_batches is not None else iter(train_dataset)
                    first = next(src, None)
                    if first is None:
                        raise RuntimeError("No se pudo obtener un batch de previsualización para guardar imágenes.")
                    preview = (norm(first[0])[:25], True)
            lr_p, is_norm = preview
            sr = (generate_f32(lr_p if is_norm else lr_p * 2.0 - 1.0).cpu().numpy() + 1.0) / 2.0
            n, (h, w, ch) = min(25, sr.shape[0]), sr.shape[1:]
            grid = np.zeros((5 * h, 5 * w, ch), np.uint8)
            for idx in range(n):
                r, c = divmod(idx, 5)
                grid[r * h:(r + 1) * h, c * w:(c + 1) * w] = (np.clip(sr[idx], 0.0, 1.0) * 255.0).round().astype(np.uint8)
            from PIL import Image
            Image.fromarray(grid if ch != 1 else grid[..., 0]).save(os.path.join(save_dir, f"epoch_{epoch_idx:03d}_sr_grid.png"))

        time_tracker, memory_tracker = EpochTimeTracker(), EpochMemoryTracker(self.ctx)
        epoch_losses = {}
        for epoch in range(epochs):
            print(f"Epoch {epoch + 1}/{epochs}")
            time_tracker.begin_epoch()
            memory_tracker.begin_epoch()
            epoch_losses = {k: [] for k in ("g_loss", "val_g_loss", "d_loss", "psnr", "val_psnr", "ssim", "val_ssim", "g_lr", "d_lr")}
            for step in range(steps_per_epoch):
                lr_b, hr_b = next(stream)
                lr_b, hr_b = norm(lr_b), norm(hr_b)
                losses = tr.train_step(lr_b, hr_b)
                epoch_losses["g_loss"].append(float(losses["g_loss"]))
                epoch_losses["d_loss"].append(float(losses["d_loss"]))
                gen01 = to01(generate_f32(lr_b))
                real01 = self.ctx.to_device((hr_b + 1.0) / 2.0)
                epoch_losses["psnr"].append(float(self.ctx.psnr(real01, gen01).mean().item()))
                epoch_losses["ssim"].append(float(self.ctx.ssim(real01, gen01).mean().item()))
                epoch_losses["g_lr"].append(float(np.float32(staircase_lr(tr.g_lr0, tr.step))))
                epoch_losses["d_lr"].append(float(np.float32(staircase_lr(tr.d_lr0, tr.step))))
                if (step + 1) % 10 == 0 or (step + 1) == steps_per_epoch:
                    print(f"  Step {step + 1}/{steps_per_epoch} G_loss={epoch_losses['g_loss'][-1]:.4f} "
                          f"D_loss={epoch_losses['d_loss'][-1]:.4f} PSNR={epoch_losses['psnr'][-1]:.2f} SSIM={epoch_losses['ssim'][-1]:.4f}")
            print(f"- Epoch Summary - G_loss: {np.mean(epoch_losses['g_loss']):.4f}, D_loss: {np.mean(epoch_losses['d_loss']):.4f}, "
                  f"PSNR: {np.mean(epoch_losses['psnr']):.2f}, SSIM: {np.mean(epoch_losses['ssim']):.4f}")
            if val_batches is not None:
                self.set_loss_network_weights(discriminator=tr.dw)
                vp, vs, vg = [], [], []
                for i, (lr_v, hr_v) in enumerate(val_batches()):
                    if val_steps is not None and i >= val_steps:
                        break
                    lr_v, hr_v = norm(lr_v), norm(hr_v)
                    fake, real = generate_f32(lr_v), self.ctx.to_device(hr_v)
                    vg.append(float(self.generator_loss(real, fake)[0]))
                    g01, r01 = to01(fake), to01(real)
                    vp.append(float(self.ctx.psnr(r01, g01).mean().item()))
                    vs.append(float(self.ctx.ssim(r01, g01).mean().item()))
                nanmean = lambda v: float(np.mean(v)) if len(v) > 0 else float("nan")
                epoch_losses["val_psnr"], epoch_losses["val_ssim"], epoch_losses["val_g_loss"] = nanmean(vp), nanmean(vs), nanmean(vg)
                print(f"  Validation -> PSNR: {epoch_losses['val_psnr']:.2f}, SSIM: {epoch_losses['val_ssim']:.4f}, "
                      f"G_loss: {epoch_losses['val_g_loss']:.4f}")
            save_grid(epoch + 1)
            self.trained = True
            memory_tracker.end_epoch()
            time_tracker.end_epoch()
        self._sync_from_trainer()
        return epoch_losses, time_tracker, memory_tracker

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

    def save(self, directory, timestamp, fmt="npz"):
        """ESRGAN_model.py:981-995 writes ESRGAN_generator_..h5 / ESRGAN_discriminator_..h5; fmt="h5" writes those names in Keras' weight
        layout (sr355.h5lite), the default "npz" this build's own container."""
        if not self.trained:
            raise RuntimeError("Cannot save an untrained model.")
        generator_path = self._save_weights(directory, f"ESRGAN_generator_x{self.scale_factor}_{timestamp}", fmt)
        print(f"Generator model saved to {generator_path}")
        if self.d_weights is not None:        # the stored (spectrally normalised) discriminator kernels, as model.save keeps them (:990-993)
            g, self.weights = self.weights, self.d_weights
            try:
                discriminator_path = self._save_weights(directory, f"ESRGAN_discriminator_x{self.scale_factor}_{timestamp}", fmt)
            finally:
                self.weights = g
            print(f"Discriminator model saved to {discriminator_path}")
        return generator_path
