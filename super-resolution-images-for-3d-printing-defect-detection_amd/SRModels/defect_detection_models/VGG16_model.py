"""Frozen-VGG16 defect classifier on MI355X behind the reference's class surface
(reference: defect_detection_models/VGG16_model.py).

Graph (VGG16_model.py:69-97): VGG16 conv base (13 x conv3x3+ReLU, 5 x maxpool) -> GAP -> Dense256 ReLU ->
Dense(num_classes) softmax; inputs are [0,1] floats with no ImageNet preprocessing; Dropout is identity at
inference.  ImageNet weights cannot be fetched offline: weights come from a checkpoint or a seeded init.
"""
import os

import numpy as np
import torch

from sr355 import pipeline as P
from sr355.wrappers import DeviceModelMixin, load_pretrained


class FineTunedVGG16(DeviceModelMixin):
    _init_scheme = "he_normal"

    def __init__(self, compute_dtype="f32"):
        self.model = None
        self.trained = False
        self.compute_dtype = compute_dtype
        self.input_shape = None

    def _mark_trained(self, v):
        self.trained = v

    def setup_model(self, input_shape=(128, 128, 3), num_classes=2, train_last_n_layers=4, base_trainable=False, dropout_rate=0.2,
                    l2_reg=0.0, learning_rate=1e-3, loss="sparse_categorical_crossentropy", from_pretrained=False,
                    pretrained_path=None):
        assert input_shape[-1] == 3, "Input must have 3 channels (RGB)."
        self.input_shape = tuple(input_shape)
        weights = load_pretrained(pretrained_path) if from_pretrained else None
        if weights is not None:
            num_classes = int(weights["predictions"][0].shape[-1])
        # base_trainable / train_last_n_layers (VGG16_model.py:75-82): the reference sets `base.trainable = False` on the nested VGG16
        # model and then `layer.trainable = True` on its last n sub-layers.  In Keras a frozen parent model keeps its sub-layers' weights
        # out of trainable_weights whatever their own flag says, and the base is called with training=False: the base STAYS FROZEN, and
        # the notebook that passes (train_last_n_layers=6, base_trainable=True) reports 131 842 trainable parameters = the two Dense
        # layers (VGG16.ipynb:L152,L161-165).  So both arguments are accepted and recorded, and training is the
        # same head-only training in either case -- the reference's behaviour, not its apparent intent.
        self.base_trainable, self.train_last_n_layers = bool(base_trainable), int(train_last_n_layers)
        self.dropout_rate, self.l2_reg, self.learning_rate = float(dropout_rate), float(l2_reg), float(learning_rate)
        if loss != "sparse_categorical_crossentropy":
            raise ValueError("only sparse_categorical_crossentropy (the reference's default, VGG16_model.py:29) is built")
        self._make("vgg16", self.compute_dtype, num_classes=num_classes)
        if weights is not None:
            self.set_weights(weights)
            print(f"Loaded pretrained model from {pretrained_path}")
        else:
            self._random_init(seed=4000)

    def trainable_layers(self):
        """Layers whose parameters `fit` updates: the two Dense layers, with or without base_trainable (see setup_model)."""
        return ["dense", "predictions"]

    def count_params(self, trainable_only=False):
        """Keras `model.count_params()` / the "Trainable params" line of model.summary(): 14 846 530 / 131 842 at num_classes=2
        (VGG16.ipynb:L151-153)."""
        names = self.trainable_layers() if trainable_only else list(self.weights)
        return int(sum(np.asarray(k).size + np.asarray(b).size for k, b in (self.weights[n] for n in names)))

    # ------------------------------------------------------------------ training of the head on the frozen base (VGG16_model.py:111-166)
    def _gap_features(self, images, batch_size=256):
        """[n,H,W,3] in [0,1] -> [n,512]: the frozen conv base on the device (tap after block5_conv3), max-pool and GAP kernels."""
        from sr355 import _lib as L
        out = []
        for i in range(0, len(images), batch_size):
            x = self.ctx.to_device(np.asarray(images[i:i + batch_size], np.float32),
                                   torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32)
            _, taps = self.model.forward_with_taps(x, ["block5_conv3"])
            f = self.ctx.spatial_op(L.SP_MAXPOOL2, taps["block5_conv3"])
            out.append(self.ctx.spatial_op(L.SP_GAP, f).cpu().numpy().reshape(len(x), -1))
        return np.concatenate(out) if out else np.zeros((0, 512), np.float32)

    @staticmethod
    def _augment(x, rng, rotation_range=20, width_shift_range=0.2, height_shift_range=0.2, horizontal_flip=True):
        """One draw of keras ImageDataGenerator(rotation_range=20, width_shift_range=0.2, height_shift_range=0.2,
        horizontal_flip=True) per image (VGG16_model.py:129-134): random rotation about the centre, random shifts as fractions of
        the size, fill_mode 'nearest', bilinear interpolation, random flip.  A CPU step in the reference as well (NumPy / SciPy)."""
        from scipy import ndimage
        out = np.empty_like(x)
        h, w = x.shape[1:3]
        for i, img in enumerate(x):
            th = np.deg2rad(rng.uniform(-rotation_range, rotation_range))
            ty, tx = rng.uniform(-height_shift_range, height_shift_range) * h, rng.uniform(-width_shift_range, width_shift_range) * w
            c, s_ = np.cos(th), np.sin(th)
            rot = np.array([[c, -s_], [s_, c]])
            centre = np.array([(h - 1) / 2.0, (w - 1) / 2.0])
            offset = centre - rot @ centre + np.array([ty, tx])          # output pixel o samples input rot @ o + offset
            for ch in range(img.shape[2]):
                out[i, :, :, ch] = ndimage.affine_transform(img[:, :, ch], rot, offset=offset, order=1, mode="nearest")
            if horizontal_flip and rng.random() < 0.5:
                out[i] = out[i, :, ::-1]
        return out

    def fit(self, X_train, y_train, X_val, y_val, batch_size=32, epochs=50, use_augmentation=True, seed=42):
        """FineTunedVGG16.fit (VGG16_model.py:111-157): the conv base runs on the device per batch, the two Dense layers train on the
        host (Adam, sparse CCE, Dropout, EarlyStopping / ReduceLROnPlateau).  The base is frozen with and without base_trainable,
        as in the reference (setup_model).  With augmentation the batches are 32 images, as the reference's
        datagen.flow(..., batch_size=32) hard-codes."""
        from sr355.train import fit_head
        if self.model is None:
            raise ValueError("Model is not built yet.")
        X_train, y_train = np.asarray(X_train, np.float32), np.asarray(y_train, np.int64).reshape(-1)
        rng = np.random.default_rng(seed)
        bs = 32 if use_augmentation else int(batch_size)

        def batches(epoch):
            order = rng.permutation(len(X_train))
            for i in range(0, len(order), bs):
                idx = order[i:i + bs]
                xb = X_train[idx]
                yield (self._augment(xb, rng) if use_augmentation else xb), y_train[idx]

        head, history = fit_head(self._gap_features, self.weights, batches, y_train, np.asarray(X_val, np.float32), np.asarray(y_val, np.int64).reshape(-1),
                                 learning_rate=self.learning_rate, batch_size=bs, epochs=epochs, dropout_rate=self.dropout_rate, l2_reg=self.l2_reg, seed=seed)
        w = dict(self.weights)
        w.update(head)
        self.set_weights(w)
        self.trained = True
        return history

    def evaluate(self, X_test, y_test):
        """model.evaluate -> [loss, accuracy] (VGG16_model.py:159-166)."""
        from sr355.train import sparse_cce
        if not self.trained:
            raise RuntimeError("Model has not been trained.")
        p = np.asarray(self.predict(np.asarray(X_test, np.float32)), np.float64)
        loss, acc = sparse_cce(p, np.asarray(y_test, np.int64).reshape(-1))
        if self.l2_reg > 0:
            loss += self.l2_reg * float(np.sum(np.asarray(self.weights["dense"][0], np.float64) ** 2))
        print(f"Loss: {loss:.4f}, Accuracy: {acc:.4f}")
        return [loss, acc]

    def predict(self, patches, batch_size=32):
        return self.model.predict(patches, batch_size=batch_size)

    def classify_defects_method(self, image, patch_size=None, stride=None, batch_size=32):
        """Patch the image, classify every patch, majority vote (VGG16_model.py:168-270).
        Returns (predicted_class: int, confidence: float)."""
        if self.model is None:
            raise ValueError("Model is not built yet.")
        if image is None:
            raise ValueError("image must be provided")
        shape = tuple(image.shape)
        if len(shape) != 3 or shape[2] != 3:
            raise ValueError("image must be HxWx3 RGB array")
        if patch_size is None:
            if self.input_shape is None or self.input_shape[0] is None:
                raise ValueError("Model input size is dynamic; please set patch_size.")
            patch_size = int(self.input_shape[0])
        if stride is None:
            stride = max(1, patch_size // 2)
        img, _ = P.as_device_image(self.ctx, image)
        patches = self.ctx.extract_patches(img, patch_size, stride)
        probs = self.model.predict(patches, batch_size=max(int(batch_size), 128))
        return P.majority_vote(probs.float().cpu().numpy())

    def save(self, directory, timestamp, fmt="npz"):
        """VGG16_model.py:272-283; fmt="h5" writes Keras' weight layout (sr355.h5lite), "npz" this build's container."""
        if not self.trained:
            raise RuntimeError("Cannot save an untrained model.")
        path = self._save_weights(directory, f"VGG16_{timestamp}", fmt)
        print(f"Model saved to {path}")
        return path
