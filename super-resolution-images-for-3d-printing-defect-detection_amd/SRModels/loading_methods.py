"""Dataset -> patch loaders with the reference's call-site contract (reference: SRModels/loading_methods.py).

Disk + host code (PNG decode through PIL instead of cv2.imread; both give RGB after the reference's
BGR->RGB swap).  The only arithmetic with device relevance is the bicubic pre-upscale of `srcnn` mode,
which runs the OpenCV-compatible HIP kernel.  Reference quirks that shape the returned arrays are kept
(SURVEY.md Appendix C): scale mode pads HR with the LR stride; the defects loader pads but walks the
unpadded size; srcnn mode without an interpolation map raises NameError in the reference -- here it
falls back to INTER_CUBIC, the map's documented default.
"""
import os
import pickle

import numpy as np
from PIL import Image

_EXTS = (".jpg", ".jpeg", ".png", ".bmp", ".tiff")
# interpolation_map.pkl holds, per file name, an OpenCV constant name or its integer code (loading_methods.py:131-148); unknown
# names fall back to INTER_CUBIC as in the reference
_INTERP_CODES = {"INTER_LINEAR": 1, "INTER_CUBIC": 2, "INTER_AREA": 3, "INTER_LANCZOS4": 4}


def add_padding(image, patch_size, stride):
    """Reflect-pad bottom/right so sliding windows cover the image (loading_methods.py:6-26)."""
    h, w = image.shape[:2]

    def amount(n):
        pad = (patch_size - (n % stride)) % stride if n % stride != 0 else 0
        return max(pad, patch_size - stride)

    return np.pad(image, ((0, amount(h)), (0, amount(w)), (0, 0)), mode="reflect")


def get_all_image_paths(root):
    found = [os.path.join(d, f) for d, _, files in os.walk(root) for f in files if f.lower().endswith(_EXTS)]
    return sorted(found)


def _read_rgb01(path):
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0


def _windows(h, w, patch, stride):
    return [(i, j) for i in range(0, h - patch + 1, stride) for j in range(0, w - patch + 1, stride)]


def _check_dirs(*roots):
    for r in roots:
        if not os.path.exists(r):
            raise ValueError("Both HR and LR root directories must exist.")
        if not os.path.isdir(r):
            raise ValueError("Both HR and LR root paths must be directories.")


def _resize_up(lr_img, hr_h, hr_w, code=2):
    from sr355 import Context
    ctx = Context.get()
    return ctx.resize(ctx.to_device(lr_img[None]), hr_h, hr_w, code)[0].cpu().numpy()


def load_dataset_as_patches(hr_root, lr_root, mode="srcnn", patch_size=33, stride=14, scale_factor=2, interpolation_map_path=None):
    """Aligned LR/HR patch pairs.  'srcnn': LR bicubic-upscaled to HR size, equal-size patches, returns
    (X, Y, hr_h, hr_w); 'scale': LR patches of patch_size, HR patches of patch_size*scale, returns (X, Y)."""
    if mode not in ("srcnn", "scale"):
        raise ValueError("mode must be 'srcnn' or 'scale'")
    _check_dirs(hr_root, lr_root)
    if not isinstance(patch_size, int) or patch_size <= 0:
        raise ValueError("patch_size must be positive int.")
    if not isinstance(stride, int) or stride <= 0:
        raise ValueError("stride must be positive int.")
    if mode == "scale" and (not isinstance(scale_factor, int) or scale_factor <= 0):
        raise ValueError("scale_factor must be positive int.")
    hr_paths, lr_paths = get_all_image_paths(hr_root), get_all_image_paths(lr_root)
    if not hr_paths or not lr_paths:
        raise ValueError("No images found in provided directories.")
    hr_by, lr_by = {os.path.basename(p): p for p in hr_paths}, {os.path.basename(p): p for p in lr_paths}
    interp_map = None
    if mode == "srcnn" and interpolation_map_path is not None:
        with open(interpolation_map_path, "rb") as f:
            interp_map = pickle.load(f)
    X, Y = [], []
    hr_h = hr_w = None
    for fname in sorted(set(hr_by) & set(lr_by)):
        hr_img, lr_img = _read_rgb01(hr_by[fname]), _read_rgb01(lr_by[fname])
        hr_h, hr_w = hr_img.shape[:2]
        if mode == "srcnn":
            method = interp_map.get(fname, 2) if interp_map is not None else 2
            code = _INTERP_CODES.get(method, 2) if isinstance(method, str) else int(method) if isinstance(method, (int, np.integer)) else 2
            lr_up = np.clip(_resize_up(lr_img, hr_h, hr_w, code), 0.0, 1.0)
            hr_p, lr_p = add_padding(hr_img, patch_size, stride), add_padding(lr_up, patch_size, stride)
            for i, j in _windows(hr_p.shape[0], hr_p.shape[1], patch_size, stride):   # padded extents
                X.append(lr_p[i:i + patch_size, j:j + patch_size])
                Y.append(hr_p[i:i + patch_size, j:j + patch_size])
        else:
            ps_hr = patch_size * scale_factor
            hr_p = add_padding(hr_img, ps_hr, stride)          # LR stride on purpose (reference quirk)
            lr_p = add_padding(lr_img, patch_size, stride)
            for i, j in _windows(lr_p.shape[0], lr_p.shape[1], patch_size, stride):
                a = lr_p[i:i + patch_size, j:j + patch_size]
                b = hr_p[i * scale_factor:i * scale_factor + ps_hr, j * scale_factor:j * scale_factor + ps_hr]
                if a.shape[:2] == (patch_size, patch_size) and b.shape[:2] == (ps_hr, ps_hr):
                    X.append(a)
                    Y.append(b)
    X, Y = np.array(X), np.array(Y)
    return (X, Y, hr_h, hr_w) if mode == "srcnn" else (X, Y)


def _load_class_map(class_map_path):
    if not class_map_path or not isinstance(class_map_path, str):
        raise ValueError("class_map_path must be a non-empty string.")
    if not os.path.exists(class_map_path):
        raise FileNotFoundError(f"Class labels map not found: {class_map_path}")
    with open(class_map_path, "rb") as f:
        m = pickle.load(f)
    if not isinstance(m, dict):
        raise ValueError("class_labels_map pickle must contain a dict of {basename: class_id}.")
    return m


def load_defects_dataset_as_patches(hr_root, patch_size=33, stride=14, class_map_path=None):
    """HR patches + the class id of their image (loading_methods.py:194-285)."""
    if not os.path.exists(hr_root):
        raise ValueError("HR root directory must exist.")
    if not os.path.isdir(hr_root):
        raise ValueError("HR root path must be a directory.")
    if not isinstance(patch_size, int) or patch_size <= 0:
        raise ValueError("patch_size must be positive int.")
    if not isinstance(stride, int) or stride <= 0:
        raise ValueError("stride must be positive int.")
    labels = _load_class_map(class_map_path)
    paths = get_all_image_paths(hr_root)
    if not paths:
        raise ValueError("No images found under HR root directory.")
    X, y = [], []
    for path in sorted(paths, key=os.path.basename):
        img = _read_rgb01(path)
        base = os.path.basename(path)
        if base not in labels:
            raise KeyError(f"Missing class id for image basename in class_labels_map: {base}")
        padded = add_padding(img, patch_size, stride)
        for i, j in _windows(img.shape[0], img.shape[1], patch_size, stride):    # unpadded extents (reference quirk)
            X.append(padded[i:i + patch_size, j:j + patch_size])
            y.append(int(labels[base]))
    return np.array(X, dtype=np.float32), np.array(y, dtype=np.int64)


def load_predictions_dataset(lr_root, hr_root, class_map_path):
    """Whole LR/HR image pairs + class ids (loading_methods.py:288-386) -> (X_LR, X_HR, y)."""
    for r, nm in ((lr_root, "lr_root"), (hr_root, "hr_root")):
        if not r or not isinstance(r, str) or not os.path.exists(r):
            raise ValueError(f"{nm} must be an existing directory path.")
        if not os.path.isdir(r):
            raise ValueError(f"{nm} must be a directory.")
    labels = _load_class_map(class_map_path)
    lr_paths, hr_paths = get_all_image_paths(lr_root), get_all_image_paths(hr_root)
    if not lr_paths:
        raise ValueError("No images found under LR root directory.")
    if not hr_paths:
        raise ValueError("No images found under HR root directory.")
    lr_by, hr_by = {os.path.basename(p): p for p in lr_paths}, {os.path.basename(p): p for p in hr_paths}
    common = sorted(set(lr_by) & set(hr_by))
    if not common:
        raise ValueError("No matching basenames found between LR and HR roots.")
    for base in common:
        if base not in labels:
            raise KeyError(f"Missing class id for basename in class_labels_map: {base}")
    X_LR = np.array([_read_rgb01(lr_by[b]) for b in common], dtype=np.float32)
    X_HR = np.array([_read_rgb01(hr_by[b]) for b in common], dtype=np.float32)
    return X_LR, X_HR, np.array([int(labels[b]) for b in common], dtype=np.int64)
