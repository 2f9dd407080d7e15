"""Weight containers for the sr355 models.

The reference stores Keras HDF5 files (`*.h5`, SRCNN_model.py:249-260 ...) but none exists in its
snapshot and h5py is not installed here, so the container is a flat `.npz`:
`<layer>/kernel` (conv HWIO, dense [in,out]) and `<layer>/bias`, Keras layer names as keys
(SURVEY.md Appendix D).  Random initialisation follows Keras: glorot-uniform kernels by default,
he-normal for EDSR (EDSR_model.py:61), seeded per layer so CPU oracle and GPU see the same arrays.
"""
import numpy as np


def init_weights(layer_shapes, scheme="glorot_uniform", seed=1000, bias_range=0.05):
    """layer_shapes: [(name, kernel_shape)] -> {name: (kernel f32, bias f32)}; seed + layer index per layer."""
    out = {}
    for i, (name, shape) in enumerate(layer_shapes):
        rng = np.random.default_rng(seed + i)
        shape = tuple(int(s) for s in shape)
        receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = receptive * shape[-2], receptive * shape[-1]
        if scheme == "he_normal":
            k = rng.normal(0.0, np.sqrt(2.0 / fan_in), size=shape)
        elif scheme == "glorot_uniform":
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            k = rng.uniform(-lim, lim, size=shape)
        else:
            raise ValueError(scheme)
        b = rng.uniform(-bias_range, bias_range, size=(shape[-1],))
        out[name] = (k.astype(np.float32), b.astype(np.float32))
    return out


def condition_attention(weights, shift=4):
    """Scale the SelfAttention query/key projections (`*_f`, `*_g`: kernel and bias) by 2**-shift -- exact in bf16.

    Why the synthetic weights need it: the reference's SelfAttention has no 1/sqrt(d) scale (ESRGAN_model.py:61-65) and its
    RRDB graph has a gain of 1.2 per block even with zero convs (rrdb_in + 0.2*(x + ...), ESRGAN_model.py:249-281), i.e.
    1.2**23 = 66x at the default depth.  With untrained glorot weights the softmax logits then span about +-1700: a hard
    argmax, where the output image depends on the last bit of ANY arithmetic (the CPU oracle evaluated in fp32 and in fp64
    agree to only 11 dB PSNR).  With the logits brought to O(1) -- where a trained network keeps them -- the graph is
    well conditioned (oracle fp32 vs fp64: > 100 dB; bf16 storage noise floor: ~49 dB) and a parity number means
    something.  shift=4 scales the logits by 2**-8."""
    g = np.float32(2.0 ** -int(shift))
    return {n: ((k * g, b * g) if (n.endswith("_f") or n.endswith("_g")) else (k, b)) for n, (k, b) in weights.items()}


def round_to_bf16(a):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (what the device does when it packs weights)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def bf16_rounded(weights):
    return {n: (round_to_bf16(k), b.copy()) for n, (k, b) in weights.items()}


def save_npz(path, weights):
    flat = {}
    for n, (k, b) in weights.items():
        flat[f"{n}/kernel"] = k
        flat[f"{n}/bias"] = b
    np.savez(path, **flat)


def load_npz(path):
    out = {}
    with np.load(path) as z:
        for key in z.files:
            n, which = key.rsplit("/", 1)
            out.setdefault(n, [None, None])[0 if which == "kernel" else 1] = z[key]
    return {n: (k, b) for n, (k, b) in out.items()}
