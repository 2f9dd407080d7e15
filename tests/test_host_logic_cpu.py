"""Host-side logic that needs no GPU: padding/window arithmetic, loaders (call-site contract of
SRModels/loading_methods.py), weight containers, synthetic data, vote, sharding."""
import os
import pickle

import numpy as np
import pytest
from PIL import Image

from oracle import ops as O
from sr355 import pipeline as P
from sr355.dist import shard_range
from sr355.synth import make_pairs
from sr355.weights import init_weights, load_npz, round_to_bf16, save_npz
from SRModels import constants, loading_methods as LM


def test_constants_match_reference():
    assert (constants.SRCNN_PATCH_SIZE, constants.SRCNN_STRIDE) == (24, 12)
    assert (constants.EDSR_PATCH_SIZE, constants.EDSR_STRIDE, constants.EDSR_SCALE_FACTOR) == (24, 12, 2)
    assert (constants.VGG_PATCH_SIZE, constants.VGG_STRIDE, constants.RANDOM_SEED) == (96, 48, 42)


@pytest.mark.parametrize("n,p,s", [(239, 24, 12), (478, 24, 12), (512, 48, 24), (100, 33, 14), (96, 96, 48), (48, 48, 48), (7, 4, 2)])
def test_pad_amount_and_grid_agree_with_oracle(n, p, s):
    assert P.pad_amount(n, p, s) == O.pad_amount(n, p, s)
    ny, nx = P.patch_grid(n, n + 3, p, s)
    assert ny * nx == len(O.patch_positions(n + O.pad_amount(n, p, s), n + 3 + O.pad_amount(n + 3, p, s), p, s))


def test_add_padding_mirror():
    img = np.random.default_rng(0).uniform(0, 1, (30, 22, 3)).astype(np.float32)
    assert np.array_equal(LM.add_padding(img, 12, 6), O.add_padding(img, 12, 6))


def _write_dataset(root, n=3, hr=(48, 40), scale=2):
    rng = np.random.default_rng(1)
    os.makedirs(os.path.join(root, "hr", "cls_a"))
    os.makedirs(os.path.join(root, "lr", "cls_a"))
    labels = {}
    for i in range(n):
        h = rng.integers(0, 256, (*hr, 3), dtype=np.uint8)
        l = h.reshape(hr[0] // scale, scale, hr[1] // scale, scale, 3).mean(axis=(1, 3)).astype(np.uint8)
        name = f"img_{i:02d}.png"
        Image.fromarray(h).save(os.path.join(root, "hr", "cls_a", name))
        Image.fromarray(l).save(os.path.join(root, "lr", "cls_a", name))
        labels[name] = i % 2
    cmap = os.path.join(root, "class_labels_map.pkl")
    with open(cmap, "wb") as f:
        pickle.dump(labels, f)
    return os.path.join(root, "hr"), os.path.join(root, "lr"), cmap


def test_scale_mode_loader(tmp_path):
    hr, lr, _ = _write_dataset(str(tmp_path))
    X, Y = LM.load_dataset_as_patches(hr, lr, mode="scale", patch_size=8, stride=4, scale_factor=2)
    ny, nx = P.patch_grid(24, 20, 8, 4)
    assert X.shape == (3 * ny * nx, 8, 8, 3) and Y.shape == (3 * ny * nx, 16, 16, 3)
    assert X.dtype == np.float32 and 0 <= X.min() and X.max() <= 1
    with pytest.raises(ValueError):
        LM.load_dataset_as_patches(hr, lr, mode="nope")
    with pytest.raises(ValueError):
        LM.load_dataset_as_patches(hr, str(tmp_path / "missing"), mode="scale")
    with pytest.raises(ValueError):
        LM.load_dataset_as_patches(hr, lr, mode="scale", patch_size=0)


def test_defects_and_predictions_loaders(tmp_path):
    hr, lr, cmap = _write_dataset(str(tmp_path))
    X, y = LM.load_defects_dataset_as_patches(hr, patch_size=16, stride=8, class_map_path=cmap)
    per = len(O.patch_positions(48, 40, 16, 8))          # the loader walks the UNPADDED size
    assert X.shape == (3 * per, 16, 16, 3) and y.tolist() == [0] * per + [1] * per + [0] * per and y.dtype == np.int64
    XL, XH, yy = LM.load_predictions_dataset(lr, hr, cmap)
    assert XL.shape == (3, 24, 20, 3) and XH.shape == (3, 48, 40, 3) and yy.tolist() == [0, 1, 0]
    with pytest.raises(FileNotFoundError):
        LM.load_predictions_dataset(lr, hr, str(tmp_path / "nope.pkl"))
    with open(cmap, "wb") as f:
        pickle.dump({"other.png": 0}, f)
    with pytest.raises(KeyError):
        LM.load_defects_dataset_as_patches(hr, patch_size=16, stride=8, class_map_path=cmap)


def test_weights_roundtrip_and_bf16(tmp_path):
    w = init_weights([("conv2d", (3, 3, 3, 8)), ("dense", (8, 2))], seed=5)
    assert np.array_equal(w["conv2d"][0], init_weights([("conv2d", (3, 3, 3, 8))], seed=5)["conv2d"][0])   # seeded
    p = str(tmp_path / "w.npz")
    save_npz(p, w)
    r = load_npz(p)
    assert all(np.array_equal(r[n][0], w[n][0]) and np.array_equal(r[n][1], w[n][1]) for n in w)
    import torch
    a = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    assert np.array_equal(round_to_bf16(a), torch.from_numpy(a).to(torch.bfloat16).float().numpy())


def test_synth_pairs_deterministic():
    lr, hr = make_pairs(2, 16, 12, 4, seed=9)
    lr2, _ = make_pairs(2, 16, 12, 4, seed=9)
    assert lr.shape == (2, 16, 12, 3) and hr.shape == (2, 64, 48, 3) and np.array_equal(lr, lr2)
    assert np.allclose(lr, hr.reshape(2, 16, 4, 12, 4, 3).mean(axis=(2, 4)), atol=1e-6)


def test_vote_and_shards():
    probs = np.array([[0.9, 0.1], [0.4, 0.6], [0.45, 0.55], [0.8, 0.2]])
    assert P.majority_vote(probs) == O.majority_vote(probs)
    for n, w in [(16, 8), (441, 8), (7, 3), (2, 4)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_tensor_checks_in_front_of_the_abi():
    """sr355.runtime._check_tensor: device, dtype and contiguity are checked on the host before a raw pointer crosses the ABI."""
    from types import SimpleNamespace
    import torch
    from sr355.runtime import _check_tensor
    fake = SimpleNamespace(torch_device=torch.device("cpu"))
    t = torch.zeros(2, 4, 4, 3)
    assert _check_tensor(fake, t, "x") is t
    with pytest.raises(ValueError):
        _check_tensor(fake, t.to(torch.bfloat16), "x")
    with pytest.raises(ValueError):
        _check_tensor(fake, t[:, ::2], "x")
    with pytest.raises(ValueError):
        _check_tensor(fake, t.numpy(), "x")
    with pytest.raises(ValueError):
        _check_tensor(SimpleNamespace(torch_device=torch.device("meta")), t, "x")


def test_classifier_head_gradients_against_autograd():
    """sr355.train.head_forward / head_backward / sparse_cce (the host half of FineTunedVGG16.fit) against torch autograd in fp64,
    L2 regulariser of the Dense256 kernel included; dropout masks scale by 1 / keep and are repeated in the backward pass."""
    import torch
    from sr355 import train as T
    rng = np.random.default_rng(0)
    g, y = rng.standard_normal((7, 512)), rng.integers(0, 3, 7)
    w = {"dense": (rng.standard_normal((512, 256)) * 0.05, rng.standard_normal(256) * 0.1),
         "predictions": (rng.standard_normal((256, 3)) * 0.1, rng.standard_normal(3) * 0.1)}
    p, cache = T.head_forward(g, w)
    loss, acc = T.sparse_cce(p, y)
    gr = T.head_backward(p, y, cache, w, l2_reg=0.01)
    tw = {n: (torch.tensor(a, requires_grad=True), torch.tensor(b, requires_grad=True)) for n, (a, b) in w.items()}
    z = torch.relu(torch.tensor(g) @ tw["dense"][0] + tw["dense"][1]) @ tw["predictions"][0] + tw["predictions"][1]
    ref = torch.nn.functional.cross_entropy(z, torch.tensor(y)) + 0.01 * (tw["dense"][0] ** 2).sum()
    ref.backward()
    assert abs(loss + 0.01 * np.sum(w["dense"][0] ** 2) - ref.item()) <= 1e-12 and 0 <= acc <= 1
    for n in gr:
        for s in (0, 1):
            assert np.abs(gr[n][s] - tw[n][s].grad.numpy()).max() <= 1e-12
    pd, cd = T.head_forward(g, w, training=True, dropout_rate=0.5, rng=np.random.default_rng(3))
    assert set(np.unique(np.round(cd[0][g != 0] / g[g != 0], 6))) <= {0.0, 2.0} and np.allclose(pd.sum(axis=1), 1.0)
