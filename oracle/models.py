"""Oracle model graphs (CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates the Keras graphs the reference builds; weights are dicts
{keras_layer_name: (kernel, bias)} with conv kernels HWIO and Dense kernels [in,out]
(SURVEY.md Appendix D).  Layer names follow the reference where it names them
(ESRGAN_model.py:230-341) and Keras auto-naming order otherwise.
"""
import numpy as np

from . import ops

VGG16_CFG = [(1, 2, 64), (2, 2, 128), (3, 3, 256), (4, 3, 512), (5, 3, 512)]
VGG19_CFG = [(1, 2, 64), (2, 2, 128), (3, 4, 256), (4, 4, 512), (5, 4, 512)]


# ---------------------------------------------------------------- layer tables (name, shape)
def srcnn_layers(channels=3):
    """SRCNN_model.py:48-53: 9x9x96 relu, 1x1x32 relu, 5x5x3 linear."""
    return [("conv2d", (9, 9, channels, 96)), ("conv2d_1", (1, 1, 96, 32)), ("conv2d_2", (5, 5, 32, channels))]


def edsr_layers(scale=2, channels=3, num_res_blocks=16, num_filters=64):
    """EDSR_model.py:96-125 in construction order (Keras auto names conv2d, conv2d_1, ...)."""
    shapes = [(3, 3, channels, num_filters)]
    for _ in range(num_res_blocks):
        shapes += [(3, 3, num_filters, num_filters)] * 2
    shapes.append((3, 3, num_filters, num_filters))
    if scale == 2:
        shapes.append((3, 3, num_filters, num_filters * 4))
    elif scale == 3:
        shapes.append((3, 3, num_filters, num_filters * 9))
    elif scale == 4:
        shapes += [(3, 3, num_filters, num_filters * 4)] * 2
    else:
        raise ValueError(f"Scale factor {scale} not supported. Use 2, 3, or 4.")
    shapes.append((3, 3, num_filters, channels))
    return [("conv2d" if i == 0 else f"conv2d_{i}", s) for i, s in enumerate(shapes)]


def self_attention_layers(name, channels=64):
    """ESRGAN_model.py:41-44."""
    return [(f"{name}_f", (1, 1, channels, channels // 8)), (f"{name}_g", (1, 1, channels, channels // 8)),
            (f"{name}_h", (1, 1, channels, channels // 2)), (f"{name}_v", (1, 1, channels // 2, channels))]


def esrgan_g_layers(scale=2, growth=32, num_rrdb=23, channels=3):
    """ESRGAN_model.py:303-345."""
    L = [("initial_conv", (3, 3, channels, 64))]
    for b in range(num_rrdb):
        for d in (1, 2, 3):
            for k in range(1, 5):
                L.append((f"rrdb_{b}_dense{d}_conv{k}", (3, 3, 64 + (k - 1) * growth, growth)))
            L.append((f"rrdb_{b}_dense{d}_conv5", (3, 3, 64 + 4 * growth, 64)))
    L.append(("trunk_conv", (3, 3, 64, 64)))
    L += self_attention_layers("self_attention_trunk")
    for i in range(int(np.log2(scale))):
        L.append((f"upsample_{i}_conv", (3, 3, 64, 256)))
        if i == 0:
            L += self_attention_layers("self_attention_upsample_0")
    L += [("final_conv1", (3, 3, 64, 64)), ("final_conv2", (3, 3, 64, channels))]
    return L


def vgg_base_layers(cfg, channels=3):
    L, cin = [], channels
    for blk, n, f in cfg:
        for k in range(1, n + 1):
            L.append((f"block{blk}_conv{k}", (3, 3, cin, f)))
            cin = f
    return L


def vgg16_classifier_layers(num_classes=2):
    """VGG16_model.py:69-97: VGG16 base (no top) -> GAP -> Dense256 relu -> Dense(num_classes) softmax."""
    return vgg_base_layers(VGG16_CFG) + [("dense", (512, 256)), ("predictions", (256, num_classes))]


def discriminator_layers():
    """ESRGAN_model.py:347-377 (for the parameter-count pin only; training is a later row)."""
    L = [("disc_conv1", (3, 3, 3, 64))]
    cin = 64
    for i, f in enumerate([64, 64, 128, 128, 256]):
        L.append((f"disc_conv{i + 2}", (3, 3, cin, f)))
        cin = f
    return L + [("disc_dense1", (256, 256)), ("disc_output", (256, 1))]


# ---------------------------------------------------------------- Keras model.summary() rows (name, type, output shape, params)
# Restated from the graph-building code; pinned row by row against the summaries the reference's notebooks printed
# (tests/golden/notebook_summaries.json, tests/test_oracle_pins.py).  Shapes are [None, H, W, C] with the batch as None.
def _conv_params(kh, kw, cin, cout):
    return kh * kw * cin * cout + cout


def keras_summary_generator(scale=2, growth=32, num_rrdb=23, h=24, w=24, channels=3):
    """ESRGAN_model.py:212-345 (layer names :230-341)."""
    R = [["lr_input", "InputLayer", [None, h, w, channels], 0],
         ["initial_conv", "Conv2D", [None, h, w, 64], _conv_params(3, 3, channels, 64)]]
    for b in range(num_rrdb):
        for d in (1, 2, 3):
            n = f"rrdb_{b}_dense{d}"
            for k in range(1, 5):
                R.append([f"{n}_conv{k}", "Conv2D", [None, h, w, growth], _conv_params(3, 3, 64 + (k - 1) * growth, growth)])
                R.append([f"{n}_concat{k}", "Concatenate", [None, h, w, 64 + k * growth], 0])
            R.append([f"{n}_conv5", "Conv2D", [None, h, w, 64], _conv_params(3, 3, 64 + 4 * growth, 64)])
            R += [[f"{n}_scale", "Lambda", [None, h, w, 64], 0], [f"{n}_add", "Add", [None, h, w, 64], 0]]
        R += [[f"rrdb_{b}_scale", "Lambda", [None, h, w, 64], 0], [f"rrdb_{b}_add", "Add", [None, h, w, 64], 0]]
    R += [["trunk_conv", "Conv2D", [None, h, w, 64], _conv_params(3, 3, 64, 64)], ["trunk_add", "Add", [None, h, w, 64], 0]]
    sa = count_params(self_attention_layers("sa"))
    R.append(["self_attention_trunk", "SelfAttention", [None, h, w, 64], sa])
    for i in range(int(np.log2(scale))):
        R.append([f"upsample_{i}_conv", "Conv2D", [None, h, w, 256], _conv_params(3, 3, 64, 256)])
        h, w = 2 * h, 2 * w
        R += [[f"upsample_{i}_pixelshuffle", "Lambda", [None, h, w, 64], 0], [f"upsample_{i}_leaky", "LeakyReLU", [None, h, w, 64], 0]]
        if i == 0:
            R.append(["self_attention_upsample_0", "SelfAttention", [None, h, w, 64], sa])
    R += [["final_conv1", "Conv2D", [None, h, w, 64], _conv_params(3, 3, 64, 64)],
          ["final_conv2", "Conv2D", [None, h, w, channels], _conv_params(3, 3, 64, channels)]]
    return R


def keras_summary_discriminator(h=48, w=48):
    """ESRGAN_model.py:347-377: six SpectralNormalization(Conv2D 3x3 SAME) + LeakyReLU, strides 1,2,1,2,1,2; GAP; two
    spectrally normalised Dense layers.  A wrapper's count = kernel + bias + its u vector [1, Cout]."""
    R = [["hr_input", "InputLayer", [None, h, w, 3], 0]]
    cin = 3
    for i, (f, st) in enumerate([(64, 1), (64, 2), (64, 1), (128, 2), (128, 1), (256, 2)]):
        h, w = -(-h // st), -(-w // st)                      # SAME: ceil(in / stride)
        R.append(["spectral_normalization" + (f"_{i}" if i else ""), "SpectralNormalization", [None, h, w, f], _conv_params(3, 3, cin, f) + f])
        R.append([f"disc_leaky{i + 1}", "LeakyReLU", [None, h, w, f], 0])
        cin = f
    R.append(["disc_gap", "GlobalAveragePooling2D", [None, 256], 0])
    R.append(["spectral_normalization_6", "SpectralNormalization", [None, 256], 256 * 256 + 256 + 256])
    R.append(["disc_leaky_dense1", "LeakyReLU", [None, 256], 0])
    R.append(["spectral_normalization_7", "SpectralNormalization", [None, 1], 256 + 1 + 1])
    return R


def keras_summary_vgg(cfg, h, w, input_name, last=None):
    """keras.applications VGG16/VGG19 without top: blockN_convK 3x3 SAME, blockN_pool 2x2 VALID (floor); `last` = name of the
    last layer kept (ESRGAN_model.py:395 cuts VGG19 at block5_conv4)."""
    R = [[input_name, "InputLayer", [None, h, w, 3], 0]]
    cin = 3
    for blk, n, f in cfg:
        for k in range(1, n + 1):
            R.append([f"block{blk}_conv{k}", "Conv2D", [None, h, w, f], _conv_params(3, 3, cin, f)])
            cin = f
            if last == R[-1][0]:
                return R
        h, w = h // 2, w // 2
        R.append([f"block{blk}_pool", "MaxPooling2D", [None, h, w, f], 0])
    return R


def keras_summary_srcnn(h=24, w=24, channels=3):
    """SRCNN_model.py:48-53."""
    return [[n, "Conv2D", [None, h, w, s[-1]], _conv_params(*s)] for n, s in srcnn_layers(channels)]


def keras_summary_edsr(scale=2, num_res_blocks=16, num_filters=64, channels=3):
    """EDSR_model.py:55-125 with input_shape (None, None, 3): Keras auto-names in construction order."""
    F = num_filters
    sh = lambda c: [None, None, None, c]
    R = [["input", "InputLayer", sh(channels), 0], ["conv2d", "Conv2D", sh(F), _conv_params(3, 3, channels, F)]]
    ci, ai, li, di = 1, 0, 0, 0
    suffix = lambda base, i: base if i == 0 else f"{base}_{i}"
    for _ in range(num_res_blocks):
        R.append([f"conv2d_{ci}", "Conv2D", sh(F), _conv_params(3, 3, F, F)]); ci += 1
        R.append([suffix("activation", ai), "Activation", sh(F), 0]); ai += 1
        R.append([f"conv2d_{ci}", "Conv2D", sh(F), _conv_params(3, 3, F, F)]); ci += 1
        R.append([suffix("lambda", li), "Lambda", sh(F), 0]); li += 1
        R.append([suffix("add", di), "Add", sh(F), 0]); di += 1
    R.append([f"conv2d_{ci}", "Conv2D", sh(F), _conv_params(3, 3, F, F)]); ci += 1
    R.append([suffix("add", di), "Add", sh(F), 0]); di += 1
    for r in ([scale] if scale in (2, 3) else [2, 2]):
        R.append([f"conv2d_{ci}", "Conv2D", sh(F * r * r), _conv_params(3, 3, F, F * r * r)]); ci += 1
        R.append([suffix("lambda", li), "Lambda", sh(F), 0]); li += 1
    R.append([f"conv2d_{ci}", "Conv2D", sh(channels), _conv_params(3, 3, F, channels)])
    R.append(["clip_0_1", "Lambda", sh(channels), 0])
    return R


def keras_summary_vgg16_classifier(h=96, w=96, num_classes=2):
    """VGG16_model.py:69-97: the nested functional `vgg16` is one row."""
    base = keras_summary_vgg(VGG16_CFG, h, w, "input")
    return [["input_2", "InputLayer", [None, h, w, 3], 0],
            ["vgg16", "Functional", base[-1][2], sum(r[3] for r in base)],
            ["gap", "GlobalAveragePooling2D", [None, 512], 0], ["dropout", "Dropout", [None, 512], 0],
            ["dense", "Dense", [None, 256], 512 * 256 + 256], ["dropout_1", "Dropout", [None, 256], 0],
            ["predictions", "Dense", [None, num_classes], 256 * num_classes + num_classes]]


def count_params(layers):
    return int(sum(int(np.prod(s)) + s[-1] for _, s in layers))


# ---------------------------------------------------------------- forwards
def srcnn_forward(x, w, dtype=np.float32):
    x = ops.conv2d(x, *w["conv2d"], act="relu", dtype=dtype)
    x = ops.conv2d(x, *w["conv2d_1"], act="relu", dtype=dtype)
    return ops.conv2d(x, *w["conv2d_2"], dtype=dtype)


def edsr_forward(x, w, scale=2, num_res_blocks=16, res_scaling=0.1, dtype=np.float32):
    names = ["conv2d"] + [f"conv2d_{i}" for i in range(1, 2 * num_res_blocks + 5)]
    it = iter(names)
    x = ops.conv2d(x, *w[next(it)], dtype=dtype)
    head = x
    for _ in range(num_res_blocks):
        sc = x
        x = ops.conv2d(x, *w[next(it)], act="relu", dtype=dtype)
        x = ops.conv2d(x, *w[next(it)], dtype=dtype)
        if res_scaling != 1.0:
            x = x * dtype(res_scaling)
        x = x + sc
    x = ops.conv2d(x, *w[next(it)], dtype=dtype) + head
    if scale in (2, 3):
        x = ops.depth_to_space(ops.conv2d(x, *w[next(it)], dtype=dtype), scale)
    elif scale == 4:
        x = ops.depth_to_space(ops.conv2d(x, *w[next(it)], dtype=dtype), 2)
        x = ops.depth_to_space(ops.conv2d(x, *w[next(it)], dtype=dtype), 2)
    else:
        raise ValueError(scale)
    x = ops.conv2d(x, *w[next(it)], dtype=dtype)
    return np.clip(x, 0.0, 1.0)


def _id(x):
    return x


def _dense_block(x, w, name, dtype, q=_id):
    """ESRGAN_model.py:212-254.  q = storage rounding of the four growth convs (identity in the reference's fp32); the
    block's own output x + 0.2*conv5 is returned UNROUNDED: the caller rounds it where the device stores it."""
    feats = [x]
    for k in range(1, 5):
        feats.append(q(ops.conv2d(np.concatenate(feats, axis=-1), *w[f"{name}_conv{k}"], act="relu", dtype=dtype)))
    x5 = ops.conv2d(np.concatenate(feats, axis=-1), *w[f"{name}_conv5"], dtype=dtype)
    return x + x5 * dtype(0.2)


def _sa(x, w, name, dtype, parts=None, bf16_storage=False):
    fn = ops.self_attention_bf16_storage if bf16_storage else ops.self_attention
    r = fn(x, *w[f"{name}_f"], *w[f"{name}_g"], *w[f"{name}_h"], *w[f"{name}_v"], dtype=dtype,
                           return_parts=parts is not None)
    if parts is not None:
        parts[name + "/parts"] = r[1]
        return r[0]
    return r


def esrgan_g_forward(x, w, scale=2, num_rrdb=23, dtype=np.float32, attention=True, parts=None, bf16_storage=False,
                     bf16_output=True, fused_tail=True):
    """ESRGAN_model.py:303-345.  x in [-1,1]; output tanh in [-1,1].

    bf16_storage=True is the same graph with every tensor the bf16 device path keeps in HBM rounded to bf16 where the
    device rounds it (csrc/api.hip build_esrgan): one rounding per fused conv epilogue -- so x + 0.2*conv5 once, and
    rrdb_in + 0.2*(x + 0.2*conv5) once for the third dense block of an RRDB, whose own output is never stored -- plus the
    attention roundings of ops.self_attention_bf16_storage.  Arithmetic inside a layer stays in `dtype`.  It is the
    like-for-like reference for BASELINE configs[2] (bf16): what is left between it and the device is accumulation
    order.  fused_tail (bf16_storage only): final_conv1's activation is not a stored tensor on the device's default path -- final_conv2 is
    computed in final_conv1's epilogue from a bf16 hi + lo pair of it (ops.round_bf16_hilo); False = the layer-by-layer path, which
    stores it as one bf16 value.  `parts`, when a dict, also receives the stage outputs 'initial_conv', 'rrdb_<b>', 'trunk_add',
    'upsample_<i>', 'final_conv1' (the trace the parity tests compare stage by stage)."""
    q = ops.round_bf16 if bf16_storage else _id
    if bf16_storage:
        x = q(np.asarray(x, dtype=dtype))
    x = q(ops.conv2d(x, *w["initial_conv"], dtype=dtype))
    trunk = x
    if parts is not None:
        parts["initial_conv"] = x
    for b in range(num_rrdb):
        r_in = x
        for d in (1, 2, 3):
            x = _dense_block(x, w, f"rrdb_{b}_dense{d}", dtype, q)
            if d < 3:
                x = q(x)
        x = q(r_in + x * dtype(0.2))
        if parts is not None:
            parts[f"rrdb_{b}"] = x
    x = q(trunk + ops.conv2d(x, *w["trunk_conv"], dtype=dtype))
    if parts is not None:
        parts["trunk_add"] = x
    if attention:
        x = _sa(x, w, "self_attention_trunk", dtype, parts, bf16_storage)
        if parts is not None:
            parts["self_attention_trunk"] = x
    for i in range(int(np.log2(scale))):
        x = q(ops.activation(ops.conv2d(x, *w[f"upsample_{i}_conv"], dtype=dtype), "lrelu"))
        x = ops.depth_to_space(x, 2)
        if parts is not None:
            parts[f"upsample_{i}"] = x
        if i == 0 and attention:
            x = _sa(x, w, "self_attention_upsample_0", dtype, parts, bf16_storage)
            if parts is not None:
                parts["self_attention_upsample_0"] = x
    x = ops.conv2d(x, *w["final_conv1"], act="relu", dtype=dtype)
    if bf16_storage:
        x = ops.round_bf16_hilo(x) if fused_tail else q(x)
    if parts is not None:
        parts["final_conv1"] = x
    y = ops.conv2d(x, *w["final_conv2"], act="tanh", dtype=dtype)
    return q(y) if (bf16_storage and bf16_output) else y


DISC_STRIDES = [1, 2, 1, 2, 1, 2]


def discriminator_forward(x, w, u=None, training=False, dtype=np.float32):
    """ESRGAN_model.py:347-377: six SpectralNormalization(Conv2D 3x3 SAME, strides 1,2,1,2,1,2) + LeakyReLU(0.2), GAP,
    SN(Dense 256) + LeakyReLU, SN(Dense 1, sigmoid).  x in [-1,1], returns probabilities [B,1].
    training=False (evaluate, :810-812): the stored kernels are used as they are.  training=True needs `u` ({layer: [1,Cout]}):
    every wrapper renormalises its kernel in place first -- returns (probabilities, new weights, new u)."""
    names = [f"disc_conv{i}" for i in range(1, 7)] + ["disc_dense1", "disc_output"]
    if training:
        w, u = dict(w), dict(u)
        for n in names:
            k, nu = ops.spectral_normalize(w[n][0], u[n])
            w[n], u[n] = (k, w[n][1]), nu
    h = np.asarray(x, dtype=dtype)
    for i, st in enumerate(DISC_STRIDES):
        h = ops.conv2d(h, *w[f"disc_conv{i + 1}"], stride=st, act="lrelu", dtype=dtype)
    g = h.mean(axis=(1, 2), dtype=dtype)
    g = ops.dense(g, *w["disc_dense1"], act="lrelu", dtype=dtype)
    p = ops.dense(g, *w["disc_output"], act="sigmoid", dtype=dtype)
    return (p, w, u) if training else p


def vgg19_extractor_layers():
    """ESRGAN_model.py:379-399: keras VGG19 (no top) cut at block5_conv4."""
    L = vgg_base_layers(VGG19_CFG)
    return L[:[n for n, _ in L].index("block5_conv4") + 1]


def vgg19_features(x_pre, w, dtype=np.float32):
    """x_pre: the preprocessed image (ops.vgg19_preprocess).  Output: block5_conv4 after its ReLU, [B, H/16, W/16, 512]."""
    x = np.asarray(x_pre, dtype=dtype)
    for blk, n, _ in VGG19_CFG:
        for k in range(1, n + 1):
            x = ops.conv2d(x, *w[f"block{blk}_conv{k}"], act="relu", dtype=dtype)
            if (blk, k) == (5, 4):
                return x
        x = ops.maxpool2x2(x)
    return x


def generator_loss(hr_real, hr_fake, wd, wv, dtype=np.float32):
    """The generator loss of ESRGAN._train_step / evaluate (ESRGAN_model.py:511-523, :812-826) for given generator output:
    BCE(1, D(fake)) + 1.0 * perceptual + 100.0 * L1 + 1.0 * spectral; D and VGG19 run with training=False.
    -> (g_loss, {'adversarial', 'perceptual', 'pixel', 'spectral'})."""
    d_fake = discriminator_forward(hr_fake, wd, training=False, dtype=dtype)
    adv = ops.binary_crossentropy_mean(np.ones_like(d_fake), d_fake)
    fr = vgg19_features(ops.vgg19_preprocess(hr_real), wv, dtype=dtype)
    ff = vgg19_features(ops.vgg19_preprocess(hr_fake), wv, dtype=dtype)
    perc = float(np.mean((np.asarray(fr, np.float64) - np.asarray(ff, np.float64)) ** 2))
    pix = ops.pixel_loss(hr_real, hr_fake)
    spec = ops.spectral_loss(hr_real, hr_fake)
    parts = {"adversarial": adv, "perceptual": perc, "pixel": pix, "spectral": spec}
    return adv + 1.0 * perc + 100.0 * pix + 1.0 * spec, parts


def vgg16_features(x, w, cfg=VGG16_CFG, dtype=np.float32):
    for blk, n, _ in cfg:
        for k in range(1, n + 1):
            x = ops.conv2d(x, *w[f"block{blk}_conv{k}"], act="relu", dtype=dtype)
        x = ops.maxpool2x2(x)
    return x


def vgg16_classifier_forward(x, w, dtype=np.float32):
    """VGG16_model.py:84-97; inputs are [0,1] floats with NO ImageNet preprocessing; Dropout is
    identity at inference."""
    f = vgg16_features(x, w, dtype=dtype)
    g = f.mean(axis=(1, 2), dtype=dtype)
    h = ops.dense(g, *w["dense"], act="relu", dtype=dtype)
    return ops.dense(h, *w["predictions"], act="softmax", dtype=dtype)


# ---------------------------------------------------------------- pipelines (super_resolve_image etc.)
def srcnn_super_resolve(lr_img, w, hr_h, hr_w, patch_size=33, stride=14, dtype=np.float32):
    """SRCNN_model.py:111-247 (bicubic pre-upscale, no clip, patches at HR resolution)."""
    up = ops.bicubic_resize(lr_img, hr_h, hr_w)
    padded = ops.add_padding(up, patch_size, stride)
    patches, pos = ops.extract_patches(padded, patch_size, stride)
    preds = srcnn_forward(patches, w, dtype=dtype)
    return ops.overlap_add(preds, pos, padded.shape, up.shape[:2], patch_size, 1)


def edsr_super_resolve(lr_img, w, scale, patch_size_lr=48, stride=24, num_res_blocks=16, res_scaling=0.1,
                       dtype=np.float32):
    """EDSR_model.py:189-315."""
    padded = ops.add_padding(np.asarray(lr_img, np.float32), patch_size_lr, stride)
    patches, pos = ops.extract_patches(padded, patch_size_lr, stride)
    preds = edsr_forward(patches, w, scale, num_res_blocks, res_scaling, dtype=dtype)
    return ops.overlap_add(preds, pos, padded.shape, lr_img.shape[:2], patch_size_lr, scale)


def esrgan_super_resolve(lr_img, w, scale, patch_size_lr=48, stride=24, num_rrdb=23, dtype=np.float32,
                         chunk=16, attention=True):
    """ESRGAN_model.py:858-979 ([0,1] -> [-1,1] in, (out+1)/2 back)."""
    padded = ops.add_padding(np.asarray(lr_img, np.float32), patch_size_lr, stride)
    patches, pos = ops.extract_patches(padded, patch_size_lr, stride)
    x = patches * 2.0 - 1.0
    outs = [esrgan_g_forward(x[i:i + chunk], w, scale, num_rrdb, dtype=dtype, attention=attention)
            for i in range(0, len(x), chunk)]
    hr = (np.concatenate(outs, axis=0).astype(np.float32) + 1.0) / 2.0
    return ops.overlap_add(hr, pos, padded.shape, lr_img.shape[:2], patch_size_lr, scale)


def classify_defects(image, w, patch_size=96, stride=None, dtype=np.float32):
    """VGG16_model.py:168-270."""
    img = np.asarray(image)
    if stride is None:
        stride = max(1, patch_size // 2)
    ph, pw = ops.pad_amount(img.shape[0], patch_size, stride), ops.pad_amount(img.shape[1], patch_size, stride)
    padded = img if (ph == 0 and pw == 0) else ops.add_padding(img, patch_size, stride)
    patches, _ = ops.extract_patches(padded, patch_size, stride)
    return ops.majority_vote(vgg16_classifier_forward(patches, w, dtype=dtype))
