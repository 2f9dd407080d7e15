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
        self._make("vgg16", self.compute_dtype, num_classes=num_classes)
        if weights is not None:
            self.set_weights(weights)
            print(f"Loaded pretrained model from {pretrained_path}")
        else:
            self._random_init(seed=4000)

    def fit(self, *args, **kwargs):
        raise NotImplementedError("classifier training is outside this round's hot path")

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

    def save(self, directory, timestamp):
        if not self.trained:
            raise RuntimeError("Cannot save an untrained model.")
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, f"VGG16_{timestamp}.npz")
        self._save_npz(path)
        print(f"Model saved to {path}")
        return path
