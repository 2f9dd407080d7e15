"""Oracle primitive ops (CPU, torch/NumPy).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

All image tensors are NHWC, conv kernels HWIO, Dense kernels [in, out] (the reference's
Keras/TensorFlow conventions).  Functions accept and return NumPy
arrays; `dtype` selects the arithmetic precision (np.float32 or np.float64).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=dtype)))


# --------------------------------------------------------------------------------------
# bf16 storage model (not a reference semantic: BASELINE configs[2] asks for bf16 on the device, and
# the device keeps activations in bf16 BETWEEN layers while every layer accumulates in fp32)
# --------------------------------------------------------------------------------------
def round_bf16(x):
    """Round to the nearest bf16 (ties to even) and return it in x's dtype: what a bf16 store followed by a load does.
    The graphs in oracle/models.py take this as `store=` and apply it exactly where the device writes bf16 (after each
    fused conv epilogue, the attention projections, the probabilities and the attention output)."""
    x = np.asarray(x)
    out_dtype = x.dtype if x.dtype in (np.float32, np.float64) else np.float32
    a = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return a.to(torch.bfloat16).to(torch.float32).numpy().astype(out_dtype, copy=False)


def round_bf16_hilo(x):
    """x as the sum of two bf16 values, hi = bf16(x) and lo = bf16(x - hi): how the device's fused RGB tail feeds final_conv1's activation --
    never stored -- into final_conv2's product (csrc/conv_rows.hip rows_fuse2).  ~16 mantissa bits."""
    x = np.asarray(x)
    hi = round_bf16(x)
    return hi + round_bf16(x - hi)


# --------------------------------------------------------------------------------------
# Activations (Keras semantics)
# --------------------------------------------------------------------------------------
def activation(x, act):
    """act in {None/'linear','relu','lrelu','tanh','sigmoid'}; LeakyReLU alpha=0.2
    (reference: ESRGAN_model.py:299)."""
    if act in (None, "linear", "none"):
        return x
    if act == "relu":
        return np.maximum(x, 0)
    if act == "lrelu":
        return np.where(x > 0, x, x * np.asarray(0.2, dtype=x.dtype))
    if act == "tanh":
        return np.tanh(x)
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-x))
    raise ValueError(act)


# --------------------------------------------------------------------------------------
# Conv2D, Keras padding="same" / "valid" 
# --------------------------------------------------------------------------------------
def same_pads(in_size, k, stride):
    """TF SAME padding: out=ceil(in/stride); total=max((out-1)*stride+k-in,0);
    before=total//2, after=total-before (extra pixel bottom/right)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d(x, w, b=None, stride=1, padding="same", act=None, dtype=np.float32):
    """y = act(x (*) w + b): cross-correlation, x NHWC, w HWIO.
    Reference call sites: SRCNN_model.py:50-52, EDSR_model.py:61-121,
    ESRGAN_model.py:230-341 (Keras Conv2D)."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    kh, kw = w.shape[:2]
    xt = _t(x, dtype).permute(0, 3, 1, 2)
    wt = _t(w, dtype).permute(3, 2, 0, 1)
    if padding == "same":
        pt, pb = same_pads(x.shape[1], kh, stride)
        pl, pr = same_pads(x.shape[2], kw, stride)
        xt = F.pad(xt, (pl, pr, pt, pb))
    elif padding != "valid":
        raise ValueError(padding)
    bt = None if b is None else _t(b, dtype)
    y = F.conv2d(xt, wt, bt, stride=stride).permute(0, 2, 3, 1).contiguous().numpy()
    return activation(y, act)


def conv2d_naive(x, w, b=None, stride=1, padding="same"):
    """Independent second derivation: plain fp64 NumPy loops over taps (small cases only)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    B, H, W, C = x.shape
    kh, kw, _, O = w.shape
    if padding == "same":
        pt, pb = same_pads(H, kh, stride)
        pl, pr = same_pads(W, kw, stride)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    Ho = (x.shape[1] - kh) // stride + 1
    Wo = (x.shape[2] - kw) // stride + 1
    y = np.zeros((B, Ho, Wo, O))
    for i in range(kh):
        for j in range(kw):
            xs = x[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            y += np.einsum("bhwc,co->bhwo", xs, w[i, j])
    if b is not None:
        y += np.asarray(b, dtype=np.float64)
    return y


def dense(x, w, b=None, act=None, dtype=np.float32):
    """Keras Dense: y = x @ W + b, W [in,out] (VGG16_model.py:90-96)."""
    y = np.asarray(x, dtype=dtype) @ np.asarray(w, dtype=dtype)
    if b is not None:
        y = y + np.asarray(b, dtype=dtype)
    if act == "softmax":
        y = y - y.max(axis=-1, keepdims=True)
        e = np.exp(y)
        return e / e.sum(axis=-1, keepdims=True)
    return activation(y, act)


# --------------------------------------------------------------------------------------
# depth_to_space, TF "DCR" channel order
# --------------------------------------------------------------------------------------
def depth_to_space(x, r):
    """out[b, h*r+i, w*r+j, c] = in[b, h, w, (i*r + j)*C + c]
    (tf.nn.depth_to_space NHWC; EDSR_model.py:81-90, ESRGAN_model.py:298)."""
    B, H, W, Cr = x.shape
    C = Cr // (r * r)
    y = x.reshape(B, H, W, r, r, C)          # [b,h,w,i,j,c]
    y = y.transpose(0, 1, 3, 2, 4, 5)        # [b,h,i,w,j,c]
    return np.ascontiguousarray(y.reshape(B, H * r, W * r, C))


def maxpool2x2(x):
    """Keras MaxPooling2D(2,2) VALID (floor)."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    x = x[:, :Ho * 2, :Wo * 2, :].reshape(B, Ho, 2, Wo, 2, C)
    return x.max(axis=(2, 4))


# --------------------------------------------------------------------------------------
# SelfAttention (ESRGAN_model.py:48-70): no 1/sqrt(d) scale, no gamma gate
# --------------------------------------------------------------------------------------
def self_attention(x, wf, bf, wg, bg, wh, bh, wv, bv, dtype=np.float32, return_parts=False):
    """f,g = 1x1 conv -> C/8; h = 1x1 -> C/2; s = g . f^T  [B,N,N]; beta = softmax(s,-1);
    o = beta . h; o = 1x1 conv -> C; y = x + o."""
    x = np.asarray(x, dtype=dtype)
    B, H, W, C = x.shape
    f = conv2d(x, wf, bf, dtype=dtype).reshape(B, H * W, -1)
    g = conv2d(x, wg, bg, dtype=dtype).reshape(B, H * W, -1)
    h = conv2d(x, wh, bh, dtype=dtype).reshape(B, H * W, -1)
    s = np.matmul(g, f.transpose(0, 2, 1))                   # [B,N,N]: g rows (queries), f cols (keys)
    s -= s.max(axis=-1, keepdims=True)
    np.exp(s, out=s)
    s /= s.sum(axis=-1, keepdims=True)                       # beta
    o = np.matmul(s, h).reshape(B, H, W, -1).astype(dtype)
    ov = conv2d(o, wv, bv, dtype=dtype)
    y = x + ov
    if return_parts:
        return y, dict(f=f, g=g, h=h, o=o)
    return y


def self_attention_bf16_storage(x, wf, bf, wg, bg, wh, bh, wv, bv, dtype=np.float32, return_parts=False):
    """SelfAttention (ESRGAN_model.py:48-70) as the bf16 device path stores it -- same function as self_attention() up to
    rounding, restated with a rounding at every point where csrc/attention.hip and the projections keep bf16:
    the key projection f is packed pre-multiplied by log2(e) (so the softmax runs as 2^(s - max)) and rounded to bf16 again,
    f/g/h are stored in bf16, the probabilities are rounded to bf16 before they are summed (denominator) and multiplied with
    h (numerator), the normalised output is stored in bf16, and x + v(o) is stored in bf16."""
    q = round_bf16
    x = np.asarray(x, dtype=dtype)
    B, H, W, C = x.shape
    log2e = np.float32(1.4426950408889634)
    wf2 = q(np.asarray(wf, np.float32) * log2e)
    bf2 = None if bf is None else np.asarray(bf, np.float32) * log2e
    f = q(conv2d(x, wf2, bf2, dtype=dtype)).reshape(B, H * W, -1)
    g = q(conv2d(x, wg, bg, dtype=dtype)).reshape(B, H * W, -1)
    h = q(conv2d(x, wh, bh, dtype=dtype)).reshape(B, H * W, -1)
    s = np.matmul(g, f.transpose(0, 2, 1))
    s -= s.max(axis=-1, keepdims=True)
    p = q(np.exp2(s))
    o = q(np.matmul(p, h) / p.sum(axis=-1, keepdims=True)).reshape(B, H, W, -1).astype(dtype)
    y = q(x + conv2d(o, wv, bv, dtype=dtype))
    if return_parts:
        return y, dict(f=f, g=g, h=h, o=o)
    return y


def attention_rows_streaming(q_rows, k, v):
    """fp64 streaming-softmax oracle for sampled query rows (whole-tile attention cannot be
    materialised: SURVEY.md 8d).  q_rows [R,d], k [N,d], v [N,dv] -> [R,dv]."""
    q_rows = np.asarray(q_rows, np.float64)
    k = np.asarray(k, np.float64)
    v = np.asarray(v, np.float64)
    out = np.zeros((q_rows.shape[0], v.shape[1]))
    m = np.full(q_rows.shape[0], -np.inf)
    l = np.zeros(q_rows.shape[0])
    step = 65536
    for s0 in range(0, k.shape[0], step):
        s = q_rows @ k[s0:s0 + step].T
        mn = np.maximum(m, s.max(axis=1))
        a = np.exp(m - mn)
        p = np.exp(s - mn[:, None])
        l = l * a + p.sum(axis=1)
        out = out * a[:, None] + p @ v[s0:s0 + step]
        m = mn
    return out / l[:, None]


# --------------------------------------------------------------------------------------
# Bicubic resize, OpenCV INTER_CUBIC semantics
# reference call sites: classic_algorithms.py:11-13, SRCNN_model.py:191, loading_methods.py:147
# --------------------------------------------------------------------------------------
_CUBIC_A = -0.75


def cubic_coeffs(fx, dtype=np.float32):
    """OpenCV interpolateCubic: three Keys(a=-0.75) polynomials, fourth = 1 - sum."""
    A = dtype(_CUBIC_A)
    x = np.asarray(fx, dtype=dtype)
    one = dtype(1)
    c0 = ((A * (x + one) - dtype(5) * A) * (x + one) + dtype(8) * A) * (x + one) - dtype(4) * A
    c1 = ((A + dtype(2)) * x - (A + dtype(3))) * x * x + one
    c2 = ((A + dtype(2)) * (one - x) - (A + dtype(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(dtype)


def _cubic_axis(n_src, n_dst, dtype):
    """Per destination index: 4 clamped source indices and 4 weights (half-pixel centres,
    replicate border)."""
    scale = 1.0 / (float(n_dst) / float(n_src))                 # OpenCV: scale = 1/inv_scale, double
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)            # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    frac = (f - s.astype(np.float32)).astype(np.float32)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    return idx, cubic_coeffs(frac.astype(dtype), dtype)


def bicubic_resize(img, out_h, out_w, dtype=np.float32):
    """Float path of cv2.resize(img, (out_w, out_h), interpolation=INTER_CUBIC): horizontal
    pass then vertical pass, no clipping.  img [H,W,C] or [B,H,W,C] float."""
    img = np.asarray(img, dtype=dtype)
    squeeze = img.ndim == 3
    if squeeze:
        img = img[None]
    B, H, W, C = img.shape
    ix, wx = _cubic_axis(W, out_w, dtype)
    iy, wy = _cubic_axis(H, out_h, dtype)
    tmp = np.zeros((B, H, out_w, C), dtype=dtype)
    for k in range(4):
        tmp += img[:, :, ix[:, k], :] * wx[None, None, :, k, None]
    out = np.zeros((B, out_h, out_w, C), dtype=dtype)
    for k in range(4):
        out += tmp[:, iy[:, k], :, :] * wy[None, :, k, None, None]
    return out[0] if squeeze else out


def bicubic_resize_u8(img, out_h, out_w):
    """uint8 path of cv2.resize INTER_CUBIC: 11-bit fixed-point coefficients
    (INTER_RESIZE_COEF_BITS=11), int32 accumulation, rounding shift by 22, saturate
    (super_resolucion_clasica.ipynb cell 7 feeds uint8)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    H, W, C = img.shape
    ix, wx = _cubic_axis(W, out_w, np.float32)
    iy, wy = _cubic_axis(H, out_h, np.float32)
    iwx = np.clip(np.rint(wx * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    iwy = np.clip(np.rint(wy * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    src = img.astype(np.int64)
    tmp = np.zeros((H, out_w, C), dtype=np.int64)
    for k in range(4):
        tmp += src[:, ix[:, k], :] * iwx[None, :, k, None]
    out = np.zeros((out_h, out_w, C), dtype=np.int64)
    for k in range(4):
        out += tmp[iy[:, k], :, :] * iwy[:, k, None, None]
    out = (out + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# cv2.resize with INTER_LINEAR / INTER_AREA / INTER_LANCZOS4 (classic_algorithms.py:7-21; the per-file codes of
# interpolation_map.pkl in loading_methods.py:131-148).  OpenCV (imgproc/resize.cpp) is not vendored by the reference and not
# installed here: its published algorithm is restated -- per axis a table of (clamped source index, weight) taps, a horizontal
# pass into float rows, then a vertical pass -- with the coefficient formulas of resizeGeneric / interpolateLanczos4 /
# computeResizeAreaTab.
# --------------------------------------------------------------------------------------
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4


def lanczos4_coeffs(x):
    """OpenCV interpolateLanczos4: 8 taps at floor-3 .. floor+4 for a fractional offset x (float32 results)."""
    s45 = 0.70710678118654752440084436210485
    cs = [(1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45)]
    x = float(np.float32(x))
    if x < np.finfo(np.float32).eps:
        c = np.zeros(8, np.float32)
        c[3] = 1.0
        return c
    y0 = -(x + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    c = np.zeros(8, np.float32)
    total = np.float32(0)
    for i in range(8):
        y = -(x + 3 - i) * math.pi * 0.25
        c[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        total = np.float32(total + c[i])
    return (c * np.float32(np.float32(1.0) / total)).astype(np.float32)


def resize_axis_taps(n_src, n_dst, interpolation, area_up=False):
    """-> (idx [n_dst, T] clamped source indices, w [n_dst, T] float32 weights) for one axis."""
    scale = 1.0 / (float(n_dst) / float(n_src))                   # OpenCV: inv_scale = dsize/ssize; scale = 1/inv_scale
    inv_scale = float(n_dst) / float(n_src)
    if interpolation == INTER_AREA and not area_up:               # true area (shrinking): computeResizeAreaTab
        rows = []
        for d in range(n_dst):
            f1 = d * scale
            f2 = f1 + scale
            cell = min(scale, n_src - f1)
            s1, s2 = math.ceil(f1), math.floor(f2)
            s2 = min(s2, n_src - 1)
            s1 = min(s1, s2)
            taps = []
            if s1 - f1 > 1e-3:
                taps.append((s1 - 1, np.float32((s1 - f1) / cell)))
            for sx in range(s1, s2):
                taps.append((sx, np.float32(1.0 / cell)))
            if f2 - s2 > 1e-3:
                taps.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
            rows.append(taps)
        T = max(len(r) for r in rows)
        idx = np.zeros((n_dst, T), np.int64)
        w = np.zeros((n_dst, T), np.float32)
        for d, taps in enumerate(rows):
            for k, (i, a) in enumerate(taps):
                idx[d, k], w[d, k] = i, a
        return idx, w
    d = np.arange(n_dst, dtype=np.float64)
    if interpolation in (INTER_LINEAR, INTER_AREA):
        if interpolation == INTER_AREA:                           # INTER_AREA asked to enlarge: linear taps, "area" coordinates
            s = np.floor(d * scale).astype(np.int64)
            f = ((d + 1) - (s + 1) * inv_scale).astype(np.float32)
            f = np.where(f <= 0, np.float32(0), f - np.floor(f)).astype(np.float32)
        else:
            fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
            s = np.floor(fx).astype(np.int64)
            f = (fx - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f = np.where(lo, np.float32(0), f)
        s = np.where(lo, 0, s)
        hi = s >= n_src - 1
        f = np.where(hi, np.float32(0), f)
        s = np.where(hi, n_src - 1, s)
        idx = np.clip(np.stack([s, s + 1], axis=1), 0, n_src - 1)
        w = np.stack([np.float32(1) - f, f], axis=1).astype(np.float32)
        return idx, w
    if interpolation == INTER_LANCZOS4:
        fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(fx).astype(np.int64)
        f = (fx - s.astype(np.float32)).astype(np.float32)
        s, f = np.where(f >= 1, s + 1, s), np.where(f >= 1, np.float32(0), f)      # (s, 1.0) == (s + 1, 0.0)
        idx = np.clip(s[:, None] + np.arange(-3, 5)[None, :], 0, n_src - 1)
        w = np.stack([lanczos4_coeffs(v) for v in f])
        return idx, w
    if interpolation == INTER_CUBIC:
        return _cubic_axis(n_src, n_dst, np.float32)
    raise ValueError(f"interpolation code {interpolation}")


def nearest_indices(n_src, n_dst):
    """OpenCV resizeNN: source index of destination index d = min(cvFloor(d * (1 / (n_dst / n_src))), n_src - 1), in double."""
    ifx = 1.0 / (float(n_dst) / float(n_src))
    return np.minimum(np.floor(np.arange(n_dst, dtype=np.float64) * ifx).astype(np.int64), n_src - 1)


def cv_resize_nearest(img, out_h, out_w):
    """cv2.resize(..., interpolation=INTER_NEAREST), any dtype, [H,W,C] or [B,H,W,C]: a pure gather."""
    img = np.asarray(img)
    H, W = img.shape[-3], img.shape[-2]
    iy, ix = nearest_indices(H, out_h), nearest_indices(W, out_w)
    return img[..., iy, :, :][..., :, ix, :]


def cv_resize(img, out_h, out_w, interpolation):
    """cv2.resize(img, (out_w, out_h), interpolation=...) for float32 images [H,W,C] / [B,H,W,C]: horizontal pass into float
    rows, then the vertical pass (taps in ascending source order), no clipping."""
    img = np.asarray(img, np.float32)
    if interpolation == INTER_NEAREST:
        return cv_resize_nearest(img, out_h, out_w)
    squeeze = img.ndim == 3
    if squeeze:
        img = img[None]
    B, H, W, C = img.shape
    area_up = interpolation == INTER_AREA and not (out_w <= W and out_h <= H)
    ix, wx = resize_axis_taps(W, out_w, interpolation, area_up)
    iy, wy = resize_axis_taps(H, out_h, interpolation, area_up)
    tmp = np.zeros((B, H, out_w, C), np.float32)
    for k in range(ix.shape[1]):
        tmp += img[:, :, ix[:, k], :] * wx[None, None, :, k, None]
    out = np.zeros((B, out_h, out_w, C), np.float32)
    for k in range(iy.shape[1]):
        out += tmp[:, iy[:, k], :, :] * wy[None, :, k, None, None]
    return out[0] if squeeze else out


def cv_resize_u8(img, out_h, out_w, interpolation):
    """uint8 path of cv2.resize for INTER_LINEAR / INTER_AREA (enlarging) / INTER_LANCZOS4 / INTER_CUBIC: 11-bit fixed-point taps
    (saturate_cast<short>(w * 2048)), integer horizontal pass; vertical pass = OpenCV's VResizeLinear<uchar> special form for the
    2-tap kernels, the generic 22-bit rounding shift otherwise."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    H, W, C = img.shape
    if interpolation == INTER_NEAREST:
        return cv_resize_nearest(img, out_h, out_w)
    if interpolation == INTER_LINEAR and W == 2 * out_w and H == 2 * out_h:
        interpolation = INTER_AREA                                 # OpenCV's resize(): bilinear halving IS the 2 x 2 box mean
    area_up = interpolation == INTER_AREA and not (out_w <= W and out_h <= H)
    if interpolation == INTER_AREA and not area_up:
        return cv_resize_area_shrink_u8(img, out_h, out_w)
    ix, wx = resize_axis_taps(W, out_w, interpolation, area_up)
    iy, wy = resize_axis_taps(H, out_h, interpolation, area_up)
    iwx = np.clip(np.rint(wx * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    iwy = np.clip(np.rint(wy * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    src = img.astype(np.int64)
    tmp = np.zeros((H, out_w, C), np.int64)
    for k in range(ix.shape[1]):
        tmp += src[:, ix[:, k], :] * iwx[None, :, k, None]
    if ix.shape[1] == 2:
        s0, s1 = tmp[iy[:, 0]], tmp[iy[:, 1]]
        out = (((iwy[:, 0, None, None] * (s0 >> 4)) >> 16) + ((iwy[:, 1, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    else:
        out = np.zeros((out_h, out_w, C), np.int64)
        for k in range(iy.shape[1]):
            out += tmp[iy[:, k]] * iwy[:, k, None, None]
        out = (out + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def cv_resize_area_shrink_u8(img, out_h, out_w):
    """uint8 INTER_AREA, shrinking both ways (classic_algorithms.py:15-17 feeds uint8 images; OpenCV imgproc/resize.cpp).
    Whole-number factors on both axes (resizeAreaFast_): the integer sum over the iy x ix cell, then (sum + 2) >> 2 for the 2 x 2
    cell of 1-, 3- and 4-channel images (ResizeAreaFastVec) and saturate_cast<uchar>(sum * (1.f / area)) -- a float product,
    rounded half to even -- otherwise.  Any other factor (resizeArea_<uchar, float>): the float path's taps and order of
    operations on the uint8 values (per source row: sum of S * alpha in ascending source order; then sum of beta * row), the
    float sum rounded half to even and saturated."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    if W % out_w == 0 and H % out_h == 0:
        fy, fx = H // out_h, W // out_w
        sums = img.astype(np.int64).reshape(out_h, fy, out_w, fx, C).sum(axis=(1, 3))
        if fy == 2 and fx == 2 and C in (1, 3, 4):
            return ((sums + 2) >> 2).astype(np.uint8)
        v = np.rint(sums.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(fy * fx)))
        return np.clip(v, 0, 255).astype(np.uint8)
    v = np.rint(cv_resize(img.astype(np.float32), out_h, out_w, INTER_AREA))
    return np.clip(v, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# PSNR / SSIM, tf.image semantics (reference metrics.py:3-7)
# --------------------------------------------------------------------------------------
def psnr(a, b, max_val=1.0, dtype=np.float32):
    """Per-image: 20*log10(max) - 10*log10(mean((a-b)^2 over H,W,C)); mse=0 -> +inf."""
    a = np.asarray(a, dtype=dtype)
    b = np.asarray(b, dtype=dtype)
    mse = np.mean((a - b) ** 2, axis=(-3, -2, -1), dtype=dtype)
    with np.errstate(divide="ignore"):
        return (dtype(20.0) * np.log10(dtype(max_val)) - dtype(10.0) * np.log10(mse)).astype(dtype)


def gauss_kernel_1d(size=11, sigma=1.5, dtype=np.float64):
    """1-D factor of tf.image's _fspecial_gauss: softmax over -(x^2)/(2 sigma^2), x=-5..5.
    The 2-D 121-entry softmax is exactly the outer product of this vector with itself."""
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(x ** 2) / (2.0 * sigma * sigma))
    return (g / g.sum()).astype(dtype)


def ssim(a, b, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, dtype=np.float32):
    """tf.image.ssim: 11x11 gaussian depthwise VALID filtering, per channel
    lum*cs averaged over space, then over channels.  a,b [B,H,W,C] (or [H,W,C])."""
    a = np.asarray(a, dtype=dtype)
    b = np.asarray(b, dtype=dtype)
    squeeze = a.ndim == 3
    if squeeze:
        a, b = a[None], b[None]
    if a.shape[1] < filter_size or a.shape[2] < filter_size:
        raise ValueError("ssim needs H,W >= filter_size")
    g1 = gauss_kernel_1d(filter_size, filter_sigma, np.float64)
    g2 = np.outer(g1, g1)
    C = a.shape[-1]
    wt = _t(np.broadcast_to(g2[None, None], (C, 1, filter_size, filter_size)).copy(), dtype)

    def red(x):
        return F.conv2d(_t(x, dtype).permute(0, 3, 1, 2), wt, groups=C).permute(0, 2, 3, 1).numpy()

    c1 = dtype((k1 * max_val) ** 2)
    c2 = dtype((k2 * max_val) ** 2)
    m0, m1 = red(a), red(b)
    num0 = m0 * m1 * dtype(2.0)
    den0 = m0 * m0 + m1 * m1
    lum = (num0 + c1) / (den0 + c1)
    num1 = red(a * b) * dtype(2.0)
    den1 = red(a * a + b * b)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    val = np.mean(lum * cs, axis=(1, 2), dtype=dtype)        # [B,C]
    out = np.mean(val, axis=-1, dtype=dtype).astype(dtype)
    return out[0] if squeeze else out


def ssim_naive(a, b, max_val=1.0):
    """Independent second derivation of tf.image.ssim in fp64 via scipy separable filters."""
    from scipy.ndimage import correlate1d
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    g = gauss_kernel_1d(11, 1.5, np.float64)

    def red(x):  # x [H,W,C], VALID
        y = correlate1d(x, g, axis=0, mode="constant")
        y = correlate1d(y, g, axis=1, mode="constant")
        return y[5:-5, 5:-5, :]

    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    res = []
    for i in range(a.shape[0]):
        m0, m1 = red(a[i]), red(b[i])
        lum = (2 * m0 * m1 + c1) / (m0 ** 2 + m1 ** 2 + c1)
        cs = (2 * red(a[i] * b[i]) - 2 * m0 * m1 + c2) / (red(a[i] ** 2 + b[i] ** 2) - m0 ** 2 - m1 ** 2 + c2)
        res.append((lum * cs).mean(axis=(0, 1)).mean())
    return np.asarray(res)


# --------------------------------------------------------------------------------------
# ESRGAN losses (ESRGAN_model.py:401-473) and their library semantics
# --------------------------------------------------------------------------------------
def l2_normalize(x, eps=1e-12):
    """tf.math.l2_normalize over all elements: x / sqrt(max(sum(x^2), eps))."""
    x = np.asarray(x, np.float64)
    return x / np.sqrt(max(float(np.sum(x * x)), eps))


def spectral_normalize(kernel, u, power_iterations=1):
    """tfa.layers.SpectralNormalization.normalize_weights: w = reshape(kernel, [-1, Cout]); v = l2n(u w^T);
    u = l2n(v w); sigma = v w u^T; returns (kernel / sigma, new u).  The reference applies it IN PLACE to the stored kernel on every
    training=True call; with training=False the stored kernel is used as it is."""
    k = np.asarray(kernel, np.float64)
    w = k.reshape(-1, k.shape[-1])
    u = np.asarray(u, np.float64).reshape(1, -1)
    for _ in range(power_iterations):
        v = l2_normalize(u @ w.T)
        u = l2_normalize(v @ w)
    sigma = float((v @ w @ u.T).item())
    return (k / sigma).astype(np.asarray(kernel).dtype), u.astype(np.float32)


def binary_crossentropy_mean(y_true, p, eps=1e-7):
    """mean(keras.backend.binary_crossentropy(y_true, p)) on probabilities (ESRGAN_model.py:447-459): p clipped to
    [eps, 1-eps], -(t log(p + eps) + (1 - t) log(1 - p + eps))."""
    p = np.clip(np.asarray(p, np.float64), eps, 1.0 - eps)
    t = np.asarray(y_true, np.float64)
    return float(np.mean(-(t * np.log(p + eps) + (1.0 - t) * np.log(1.0 - p + eps))))


def pixel_loss(a, b):
    """ESRGAN_model.py:433-445: mean |hr_real - hr_fake|."""
    return float(np.mean(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def spectral_loss(a, b):
    """ESRGAN_model.py:461-473: tf.signal.fft2d transforms the INNERMOST two axes -- for NHWC batches that is (W, C), not (H, W)
     -- then mean | |F(a)| - |F(b)| |."""
    fa = np.abs(np.fft.fft2(np.asarray(a, np.float64), axes=(-2, -1)))
    fb = np.abs(np.fft.fft2(np.asarray(b, np.float64), axes=(-2, -1)))
    return float(np.mean(np.abs(fa - fb)))


def vgg19_preprocess(x):
    """ESRGAN_model.py:401-408 + keras.applications.vgg19.preprocess_input (caffe mode): [-1,1] -> [0,255],
    RGB -> BGR, minus the ImageNet channel means [103.939, 116.779, 123.68] (B, G, R), no scaling."""
    x = (np.asarray(x, np.float64) + 1.0) * 127.5
    return x[..., ::-1] - np.array([103.939, 116.779, 123.68])


# --------------------------------------------------------------------------------------
# Patch plumbing (loading_methods.py:6-26; SRCNN_model.py:127-188; EDSR_model.py:201-256;
# ESRGAN_model.py:883-921; VGG16_model.py:216-239)
# --------------------------------------------------------------------------------------
def pad_amount(n, patch, stride):
    """pad = (p - n%s)%s if n%s else 0; pad = max(pad, p - s)   (loading_methods.py:12-17)."""
    pad = (patch - (n % stride)) % stride if n % stride != 0 else 0
    return max(pad, patch - stride)


def add_padding(image, patch, stride):
    """Reflect padding (np.pad mode='reflect': mirror without repeating the edge sample),
    bottom and right only (loading_methods.py:20-24)."""
    h, w = image.shape[:2]
    return np.pad(image, ((0, pad_amount(h, patch, stride)), (0, pad_amount(w, patch, stride)), (0, 0)),
                  mode="reflect")


def patch_positions(h, w, patch, stride):
    return [(i, j) for i in range(0, h - patch + 1, stride) for j in range(0, w - patch + 1, stride)]


def extract_patches(image, patch, stride):
    pos = patch_positions(image.shape[0], image.shape[1], patch, stride)
    if not pos:
        return np.empty((0, patch, patch, image.shape[2]), dtype=np.float32), pos
    return np.asarray([image[i:i + patch, j:j + patch, :] for i, j in pos], dtype=np.float32), pos


def overlap_add(patches, positions, padded_hw, out_hw, patch, scale=1):
    """Scatter-add patches + count, divide (0 where count==0), crop, clip[0,1]
    (SRCNN_model.py:164-188, EDSR_model.py:225-256)."""
    hp, wp = padded_hw[0] * scale, padded_hw[1] * scale
    ps = patch * scale
    rec = np.zeros((hp, wp, 3), dtype=np.float32)
    cnt = np.zeros((hp, wp, 3), dtype=np.float32)
    for p, (i, j) in zip(patches, positions):
        rec[i * scale:i * scale + ps, j * scale:j * scale + ps, :] += p
        cnt[i * scale:i * scale + ps, j * scale:j * scale + ps, :] += 1.0
    rec = np.divide(rec, cnt, out=np.zeros_like(rec), where=cnt != 0)
    return np.clip(rec[:out_hw[0] * scale, :out_hw[1] * scale, :], 0.0, 1.0)


def majority_vote(probs):
    """VGG16_model.py:252-268: argmax per patch, bincount, tie -> highest mean prob among the
    tied classes, confidence = mean prob of the winner."""
    probs = np.asarray(probs)
    if probs.ndim != 2:
        probs = probs.reshape((probs.shape[0], -1))
    nc = int(probs.shape[1])
    votes = np.bincount(np.argmax(probs, axis=1), minlength=nc)
    top = np.where(votes == votes.max())[0]
    if len(top) == 1:
        win = int(top[0])
    else:
        win = int(top[np.argmax(probs.mean(axis=0)[top])])
    return win, float(probs[:, win].mean())
