"""ctypes binding of libsr355.so (C ABI: include/sr355.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is missing
the import of anything that needs it raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsr355.so")

SR_OK = 0
SR_ERR_INVALID, SR_ERR_HIP, SR_ERR_OOM, SR_ERR_STATE, SR_ERR_NAME, SR_ERR_CAPACITY = -1, -2, -3, -4, -5, -6
DTYPE_F32, DTYPE_BF16, DTYPE_U8 = 0, 1, 2
MODEL_SRCNN, MODEL_EDSR, MODEL_ESRGAN_G, MODEL_VGG16, MODEL_ESRGAN_D, MODEL_VGG19_FEATURES = 0, 1, 2, 3, 4, 5
ACT_LINEAR, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
WEIGHT_KERNEL, WEIGHT_BIAS = 0, 1
SP_MAXPOOL2, SP_GAP, SP_PICK2, SP_VGG_PREPROCESS = 0, 1, 2, 3
ELT_AXPBY, ELT_RELU_BWD, ELT_LRELU_BWD, ELT_CLIP01_BWD, ELT_MUL, ELT_TANH_BWD, ELT_CLIP01, ELT_SIGN_DIFF = 0, 1, 2, 3, 4, 5, 6, 7


class ModelCfg(C.Structure):
    _fields_ = [("compute_dtype", C.c_int32), ("scale_factor", C.c_int32), ("channels", C.c_int32),
                ("num_blocks", C.c_int32), ("num_filters", C.c_int32), ("growth_channels", C.c_int32),
                ("res_scaling", C.c_float), ("num_classes", C.c_int32), ("use_attention", C.c_int32)]


class View(C.Structure):
    """sr_view: a channel range of an NHWC fp32 buffer (include/sr355.h)."""
    _fields_ = [("p", C.c_void_p), ("cs", C.c_int64), ("coff", C.c_int32)]


class PackDesc(C.Structure):
    """sr_pack_desc: one conv use whose weights sr_conv_prepack packs (include/sr355.h)."""
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("K", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("rot", C.c_int32)]


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_vw = C.POINTER(View)
_fp = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/sr355.h declares
SIGNATURES = {
    "sr_init": (_i, [_i, C.POINTER(_vp)]),
    "sr_destroy": (None, [_vp]),
    "sr_last_error": (C.c_char_p, [_vp]),
    "sr_mem_info": (_i, [_vp, _i64p, _i64p]),
    "sr_last_forward_ms": (_i, [_vp, _fp]),
    "sr_debug_set_stamp_buffer": (_i, [_vp, _vp, _i64]),
    "sr_debug_stamp_bytes_needed": (_i64, [_i, _i64]),
    "sr_measure_clock": (_i, [_vp, _fp, _vp]),
    "sr_debug_set_chain_stamp_buffer": (_i, [_vp, _vp, _i64]),
    "sr_debug_set_fused": (_i, [_vp, _i, _i]),
    "sr_debug_set_alloc_cap": (_i, [_vp, _i64]),
    "sr_profile_begin": (_i, [_vp]),
    "sr_profile_end": (_i, [_vp, C.c_char_p, _i64]),
    "sr_model_create": (_i, [_vp, _i, C.POINTER(ModelCfg), C.POINTER(_vp)]),
    "sr_model_destroy": (None, [_vp]),
    "sr_model_release_workspace": (_i, [_vp]),
    "sr_model_num_ops": (_i, [_vp]),
    "sr_model_op_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "sr_model_set_tap": (_i, [_vp, _i, _vp, _i64]),
    "sr_model_num_params": (_i, [_vp]),
    "sr_model_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i), _i64p, C.POINTER(_i)]),
    "sr_model_set_weight": (_i, [_vp, C.c_char_p, _i, _fp, _i64p, _i]),
    "sr_model_finalize": (_i, [_vp]),
    "sr_model_output_shape": (_i, [_vp, _i, _i, _i, _i, _i64p]),
    "sr_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "sr_conv2d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, _f, _vp, _f, _vp, _f, _i, _i, _vp, _vp]),
    "sr_self_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp]),
    "sr_bicubic": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sr_resize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sr_psnr": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "sr_ssim": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "sr_mse": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "sr_conv2d_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _f, _vp, _f, _i, _i, _vp, _vp]),
    "sr_conv2d_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sr_conv2d_dev_views": (_i, [_vp, _vw, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _vw, _f, _vw, _vp]),
    "sr_conv_prepack": (_i, [_vp, C.POINTER(PackDesc), _i, _vp]),
    "sr_conv2d_wgrad_views": (_i, [_vp, _vw, _vw, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sr_eltwise_views": (_i, [_vp, _i, _vw, _vw, _f, _f, _vw, _i64, _i, _vp]),
    "sr_eltwise": (_i, [_vp, _i, _vp, _vp, _f, _f, _vp, _i64, _vp]),
    "sr_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "sr_space_to_depth": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sr_spatial_op": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sr_matmul": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "sr_softmax_rows": (_i, [_vp, _vp, _i64, _i, _vp]),
    "sr_softmax_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "sr_maxpool2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sr_zero_insert2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sr_spectral_l1_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "sr_l1": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "sr_spectral_l1": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sr_extract_patches": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp, _i64, C.POINTER(_i), _vp]),
    "sr_overlap_add": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp]),
}

_lib = None


def source_fingerprint():
    """sha256 over the kernel sources (csrc/*.hip, *.h, in name order): what a measurement taken on one build of the library is tagged with,
    so that a later build cannot quote it as its own (bench.py refuses a profiles/pmc_traffic.json collected on other sources)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(_HERE), "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load():
    """Load libsr355.so (once) and declare the prototypes.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"libsr355.so not found at {LIB_PATH}: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C <package>/csrc`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
