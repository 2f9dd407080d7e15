"""Host runtime over the C ABI: one Context per GPU, torch-ROCm tensors as device-memory containers.

Nothing here computes: every numeric operation is a call into libsr355.so on the tensor's device
and the current torch stream.  Errors coming back over the ABI are raised as the exception types
the reference raises at the same call sites (ValueError / RuntimeError / KeyError / MemoryError).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class Sr355Error(RuntimeError):
    pass


_EXC = {L.SR_ERR_INVALID: ValueError, L.SR_ERR_HIP: Sr355Error, L.SR_ERR_OOM: MemoryError,
        L.SR_ERR_STATE: RuntimeError, L.SR_ERR_NAME: KeyError, L.SR_ERR_CAPACITY: ValueError}

_TORCH2DT = {torch.float32: L.DTYPE_F32, torch.bfloat16: L.DTYPE_BF16, torch.uint8: L.DTYPE_U8}
_DT2TORCH = {v: k for k, v in _TORCH2DT.items()}


def dtype_code(dt):
    if isinstance(dt, str):
        dt = {"f32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
              "u8": torch.uint8}[dt]
    return _TORCH2DT[dt]


def _fptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _np32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _check_tensor(ctx, t, name, dtypes=(torch.float32,)):
    """The kernels read raw device pointers: a tensor handed over the ABI must live on the context's GPU, be dense and have
    a dtype the entry point understands.  Anything else raises ValueError here instead of reading out of bounds there."""
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch tensor on {ctx.torch_device}")
    if t.device != ctx.torch_device:
        raise ValueError(f"{name}: tensor is on {t.device}, the context drives {ctx.torch_device}")
    if t.dtype not in dtypes:
        raise ValueError(f"{name}: dtype {t.dtype} not supported here (expected one of {[str(d) for d in dtypes]})")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


class Context:
    """sr_ctx wrapper; `Context.get(i)` returns the process-wide context of GPU i."""
    _instances = {}

    @classmethod
    def get(cls, device=None):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("sr355 needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
            device = torch.cuda.current_device()
        if isinstance(device, torch.device):
            device = device.index if device.index is not None else torch.cuda.current_device()
        if device not in cls._instances:
            cls._instances[device] = cls(device)
        return cls._instances[device]

    def __init__(self, device):
        self.lib = L.load()
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.sr_init(self.device, C.byref(h))
        if rc != L.SR_OK:
            raise Sr355Error(f"sr_init(device={device}) failed with {rc} (no usable GPU?)")
        self.h = h
        self.torch_device = torch.device("cuda", self.device)

    # ------------------------------------------------------------------ helpers
    def check(self, rc):
        if rc != L.SR_OK:
            msg = self.lib.sr_last_error(self.h).decode("utf-8", "replace")
            raise _EXC.get(rc, Sr355Error)(msg)

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.torch_device).cuda_stream)

    def to_device(self, a, dtype=None):
        """NumPy array or tensor -> contiguous tensor on this GPU (the reference hands NumPy arrays to predict)."""
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.to(self.torch_device, non_blocking=False).contiguous()

    def empty(self, shape, dtype=torch.float32):
        return torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=self.torch_device)

    def mem_info(self):
        cur, peak = C.c_int64(), C.c_int64()
        self.check(self.lib.sr_mem_info(self.h, C.byref(cur), C.byref(peak)))
        return {"current": cur.value, "peak": peak.value}

    def last_forward_ms(self):
        ms = C.c_float()
        self.check(self.lib.sr_last_forward_ms(self.h, C.byref(ms)))
        return ms.value

    def measure_clock_mhz(self):
        """Shader clock held under a dense bf16 MFMA load (in-kernel s_memtime / s_memrealtime), MHz."""
        mhz = C.c_float()
        self.check(self.lib.sr_measure_clock(self.h, C.byref(mhz), self.stream()))
        return mhz.value

    FUSED_ALL = 511

    def set_fused(self, mask=511, max_workgroups=0):
        """Which dense-block conv pairs run as one fused kernel (bit 0: conv4+conv5, bit 1: conv2+conv3) and whether the generator's RGB conv rides in
        final_conv1's epilogue (bit 2) and SelfAttention's f / g / h projections in the epilogue of the conv before it (bit 3); bit 4: batches of small images (VGG16 block 5) packed into one tall image with zero separators; bit 5: conv1 of a dense block on the streaming line-buffer kernel; bit 6: a 2x2 max-pool inside the epilogue of the conv in front of it; bit 7: 3x3 convs from 64 input channels on the persistent kernel with resident weights; 0 = layer by layer."""
        self.check(self.lib.sr_debug_set_fused(self.h, int(mask), int(max_workgroups)))

    def set_alloc_cap(self, nbytes):
        """Test hook: allocations through this context fail once it would hold more than nbytes (0 = no cap)."""
        self.check(self.lib.sr_debug_set_alloc_cap(self.h, int(nbytes)))

    def profile_begin(self):
        self.check(self.lib.sr_profile_begin(self.h))

    def profile_end(self):
        """-> [{'kernel','launches','total_ms','flops','bytes'}] per kernel template instance (HIP-event timed)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        self.check(self.lib.sr_profile_end(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    # ------------------------------------------------------------------ single ops
    def conv2d(self, x, w, b=None, act="linear", alpha=1.0, skip1=None, beta1=0.0, skip2=None, beta2=0.0,
               clip01=False, d2s=1):
        """Keras Conv2D(padding='same') + fused epilogue; x NHWC tensor (f32/bf16), w HWIO NumPy."""
        w = _np32(w)
        b = _np32(b)
        kh, kw, cin, cout = w.shape
        _check_tensor(self, x, "conv2d input", (torch.float32, torch.bfloat16))
        for nm, sk in (("skip1", skip1), ("skip2", skip2)):
            if sk is not None:
                _check_tensor(self, sk, f"conv2d {nm}", (x.dtype,))
        B, H, W, Cx = x.shape
        if Cx != cin:
            raise ValueError("conv2d: input channels do not match the kernel")
        r = max(1, int(d2s))
        y = self.empty((B, H * r, W * r, cout // (r * r)), x.dtype)
        actc = {"linear": L.ACT_LINEAR, None: L.ACT_LINEAR, "relu": L.ACT_RELU, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[act]
        self.check(self.lib.sr_conv2d(self.h, x.data_ptr(), dtype_code(x.dtype), B, H, W, cin, _fptr(w), _fptr(b), kh, kw, cout,
                                      actc, float(alpha), None if skip1 is None else skip1.data_ptr(), float(beta1),
                                      None if skip2 is None else skip2.data_ptr(), float(beta2), int(bool(clip01)), r,
                                      y.data_ptr(), self.stream()))
        return y

    def conv2d_dev(self, x, w_dev, b_dev, cout, rot=False, act="linear", d2s=1, alpha=1.0, skip1=None, beta1=0.0):
        """conv2d with DEVICE fp32 weights: w_dev [K,K,Cin,Cout], or with rot=True the forward kernel [K,K,Cout,Cin] of the layer whose
        input gradient is wanted.  Packs on the device, asynchronous (sr_conv2d_dev)."""
        _check_tensor(self, x, "conv2d_dev input")
        _check_tensor(self, w_dev, "conv2d_dev kernel")
        if b_dev is not None:
            _check_tensor(self, b_dev, "conv2d_dev bias")
        if skip1 is not None:
            _check_tensor(self, skip1, "conv2d_dev skip1")
        B, H, W, Cx = x.shape
        k = w_dev.shape[0]
        want = (k, k, cout, Cx) if rot else (k, k, Cx, cout)
        if tuple(w_dev.shape) != want:
            raise ValueError(f"conv2d_dev: kernel shape {tuple(w_dev.shape)} does not match {want}")
        r = max(1, int(d2s))
        y = self.empty((B, H * r, W * r, cout // (r * r)), torch.float32)
        actc = {"linear": L.ACT_LINEAR, None: L.ACT_LINEAR, "relu": L.ACT_RELU, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[act]
        self.check(self.lib.sr_conv2d_dev(self.h, x.data_ptr(), B, H, W, Cx, w_dev.data_ptr(), None if b_dev is None else b_dev.data_ptr(), k,
                                          int(cout), int(bool(rot)), actc, float(alpha), None if skip1 is None else skip1.data_ptr(), float(beta1),
                                          None, 0.0, 0, r, y.data_ptr(), self.stream()))
        return y

    def self_attention(self, x, wf, bf, wg, bg, wh, bh, wv, bv):
        _check_tensor(self, x, "self_attention input", (torch.float32, torch.bfloat16))
        B, H, W, Cx = x.shape
        arrs = [_np32(a) for a in (wf, bf, wg, bg, wh, bh, wv, bv)]
        y = torch.empty_like(x)
        self.check(self.lib.sr_self_attention(self.h, x.data_ptr(), dtype_code(x.dtype), B, H, W, Cx, *[_fptr(a) for a in arrs],
                                              y.data_ptr(), self.stream()))
        return y

    def bicubic(self, x, out_h, out_w):
        _check_tensor(self, x, "bicubic input", (torch.float32, torch.uint8))
        B, H, W, Cx = x.shape
        y = self.empty((B, out_h, out_w, Cx), x.dtype)
        self.check(self.lib.sr_bicubic(self.h, x.data_ptr(), dtype_code(x.dtype), B, H, W, Cx, int(out_h), int(out_w), y.data_ptr(),
                                       self.stream()))
        return y

    INTERPOLATIONS = {"INTER_NEAREST": 0, "INTER_LINEAR": 1, "INTER_CUBIC": 2, "INTER_AREA": 3, "INTER_LANCZOS4": 4, "INTER_LINEAR_EXACT": 5}

    def resize(self, x, out_h, out_w, interpolation="INTER_CUBIC"):
        """cv2.resize(x, (out_w, out_h), interpolation): x [B,H,W,C] f32 or u8 tensor; interpolation = OpenCV name or code
        (INTER_NEAREST 0, INTER_LINEAR 1, INTER_CUBIC 2, INTER_AREA 3, INTER_LANCZOS4 4; INTER_LINEAR_EXACT 5 on float images, where
        OpenCV itself falls back to INTER_LINEAR).  Codes cv2.resize rejects, and the two it accepts that are not restated here
        (INTER_NEAREST_EXACT 6; INTER_LINEAR_EXACT on uint8), raise ValueError -- cv2.error where the reference runs."""
        code = self.INTERPOLATIONS.get(interpolation, interpolation)
        _check_tensor(self, x, "resize input", (torch.float32, torch.uint8))
        if code == 5 and x.dtype == torch.float32:
            code = 1
        if isinstance(code, bool) or not isinstance(code, (int, np.integer)) or int(code) not in (0, 1, 2, 3, 4):
            raise ValueError(f"unsupported interpolation {interpolation!r}")
        B, H, W, Cx = x.shape
        y = self.empty((B, out_h, out_w, Cx), x.dtype)
        self.check(self.lib.sr_resize(self.h, x.data_ptr(), dtype_code(x.dtype), B, H, W, Cx, int(out_h), int(out_w), int(code), y.data_ptr(),
                                      self.stream()))
        return y

    def _metric(self, fn, a, b, max_val):
        _check_tensor(self, a, "metric input a")
        _check_tensor(self, b, "metric input b")
        if a.shape != b.shape or a.dim() != 4:
            raise ValueError("metric inputs must be two [B,H,W,C] tensors of the same shape")
        B, H, W, Cx = a.shape
        out = self.empty((B,), torch.float32)
        if B:
            self.check(fn(self.h, a.data_ptr(), b.data_ptr(), B, H, W, Cx, float(max_val), out.data_ptr(), self.stream()))
        return out

    def psnr(self, a, b, max_val=1.0):
        return self._metric(self.lib.sr_psnr, a, b, max_val)

    def ssim(self, a, b, max_val=1.0):
        return self._metric(self.lib.sr_ssim, a, b, max_val)

    def mse(self, a, b):
        _check_tensor(self, a, "mse input a")
        _check_tensor(self, b, "mse input b")
        if a.shape != b.shape:
            raise ValueError("mse inputs must have the same shape")
        out = self.empty((1,), torch.float32)
        self.check(self.lib.sr_mse(self.h, a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), self.stream()))
        return out

    # ------------------------------------------------------------------ channel-range views (the training tape's dense blocks: sr_*_views)
    @staticmethod
    def _view(t, coff, c):
        """(tensor [B,H,W,Cbuf] fp32 contiguous, first channel, channels) -> sr_view."""
        if not (isinstance(t, torch.Tensor) and t.dim() == 4 and t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda):
            raise ValueError("a view needs a contiguous fp32 NHWC device tensor")
        if coff < 0 or c <= 0 or coff + c > t.shape[3]:
            raise ValueError(f"channel range [{coff}, {coff + c}) outside the buffer's {t.shape[3]} channels")
        return L.View(t.data_ptr(), int(t.shape[3]), int(coff))

    def conv2d_dev_view(self, xbuf, x_coff, cin, w_dev, b_dev, cout, ybuf, y_coff, rot=False, act="linear", alpha=1.0, skip_buf=None, skip_coff=0, beta1=0.0):
        """sr_conv2d_dev on channel ranges: reads xbuf[..., x_coff : x_coff + cin], writes ybuf[..., y_coff : y_coff + cout] (other channels untouched);
        skip_buf / skip_coff: out = alpha * act(conv) + beta1 * skip -- skip may be the output range itself (in-place accumulation)."""
        B, H, W, _ = xbuf.shape
        if tuple(ybuf.shape[:3]) != (B, H, W):
            raise ValueError("conv2d_dev_view: input and output buffers must share [B,H,W]")
        k = w_dev.shape[0]
        want = (k, k, cout, cin) if rot else (k, k, cin, cout)
        if tuple(w_dev.shape) != want:
            raise ValueError(f"conv2d_dev_view: kernel shape {tuple(w_dev.shape)} does not match {want}")
        xv, yv = self._view(xbuf, x_coff, cin), self._view(ybuf, y_coff, cout)
        sv = None if skip_buf is None else self._view(skip_buf, skip_coff, cout)
        actc = {"linear": L.ACT_LINEAR, None: L.ACT_LINEAR, "relu": L.ACT_RELU, "lrelu": L.ACT_LRELU, "tanh": L.ACT_TANH}[act]
        self.check(self.lib.sr_conv2d_dev_views(self.h, C.byref(xv), B, H, W, int(cin), w_dev.data_ptr(), None if b_dev is None else b_dev.data_ptr(), k, int(cout),
                                                int(bool(rot)), actc, float(alpha), None if sv is None else C.byref(sv), float(beta1), C.byref(yv), self.stream()))

    def pack_list(self, uses):
        """uses: [(w_dev [K,K,I,O] fp32 device tensor, b_dev or None, rot)] -> an object for conv_prepack.  rot=False is the layer's forward use (the call's Cin = I,
        Cout = O, with its bias), rot=True its input-gradient use (Cin = O, Cout = I, no bias) -- exactly the arguments conv2d_dev / conv2d_dev_view pass for them."""
        arr = (L.PackDesc * max(len(uses), 1))()
        keep = []
        for d, (w, b, rot) in zip(arr, uses):
            _check_tensor(self, w, "pack_list kernel")
            if b is not None:
                _check_tensor(self, b, "pack_list bias")
            k, _, ci, co = w.shape
            d.w, d.bias, d.K = w.data_ptr(), (None if b is None else b.data_ptr()), k
            d.Cin, d.Cout, d.rot = (co, ci, 1) if rot else (ci, co, 0)
            keep.append((w, b))
        return (arr, len(uses), keep)

    def conv_prepack(self, packed_list):
        """sr_conv_prepack: pack every listed use by one launch from the weights' CURRENT contents; later conv2d_dev / conv2d_dev_view calls with the same tensors skip
        their own pack.  Call again after every change of a listed weight; conv_prepack(None) forgets the list."""
        if packed_list is None:
            self.check(self.lib.sr_conv_prepack(self.h, None, 0, self.stream()))
            return
        arr, n, _ = packed_list
        self.check(self.lib.sr_conv_prepack(self.h, arr, n, self.stream()))

    def conv2d_wgrad_view(self, xbuf, x_coff, cin, dybuf, dy_coff, cout, k):
        """sr_conv2d_wgrad on channel ranges -> (dw HWIO [k,k,cin,cout], db [cout]) device tensors."""
        B, H, W, _ = xbuf.shape
        xv, dv = self._view(xbuf, x_coff, cin), self._view(dybuf, dy_coff, cout)
        dw = self.empty((k, k, cin, cout), torch.float32)
        db = self.empty((cout,), torch.float32)
        self.check(self.lib.sr_conv2d_wgrad_views(self.h, C.byref(xv), C.byref(dv), B, H, W, int(cin), int(cout), int(k), dw.data_ptr(), db.data_ptr(), self.stream()))
        return dw, db

    def eltwise_view(self, op, abuf, a_coff, bbuf, b_coff, obuf, o_coff, c, alpha=1.0, beta=0.0):
        """sr_eltwise over c channels of every pixel: out range = op(a range, b range) (bbuf None for one-operand ops)."""
        B, H, W, _ = abuf.shape
        av, ov = self._view(abuf, a_coff, c), self._view(obuf, o_coff, c)
        bv = None if bbuf is None else self._view(bbuf, b_coff, c)
        self.check(self.lib.sr_eltwise_views(self.h, int(op), C.byref(av), None if bv is None else C.byref(bv), float(alpha), float(beta), C.byref(ov), B * H * W, int(c),
                                             self.stream()))

    # ------------------------------------------------------------------ backward-pass pieces (sr355/train.py)
    def conv2d_wgrad(self, x, dy, k):
        """Kernel and bias gradient of Conv2D(k x k, SAME, stride 1): x [B,H,W,Cin], dy [B,H,W,Cout] fp32 -> (dw HWIO, db) device tensors."""
        _check_tensor(self, x, "wgrad x")
        _check_tensor(self, dy, "wgrad dy")
        B, H, W, cin = x.shape
        if tuple(dy.shape[:3]) != (B, H, W):
            raise ValueError("wgrad: x and dy must share [B,H,W]")
        cout = dy.shape[3]
        dw = self.empty((k, k, cin, cout), torch.float32)
        db = self.empty((cout,), torch.float32)
        self.check(self.lib.sr_conv2d_wgrad(self.h, x.data_ptr(), dy.data_ptr(), B, H, W, cin, cout, int(k), dw.data_ptr(), db.data_ptr(), self.stream()))
        return dw, db

    def eltwise(self, op, a, b=None, alpha=1.0, beta=0.0):
        """Element-wise halves of the chain rule (L.ELT_*): fp32 tensors of one shape -> new tensor."""
        _check_tensor(self, a, "eltwise a")
        if b is not None:
            _check_tensor(self, b, "eltwise b")
            if b.shape != a.shape:
                raise ValueError("eltwise operands must have the same shape")
        out = torch.empty_like(a)
        self.check(self.lib.sr_eltwise(self.h, int(op), a.data_ptr(), None if b is None else b.data_ptr(), float(alpha), float(beta), out.data_ptr(),
                                       a.numel(), self.stream()))
        return out

    def adam_step(self, w, g, m, v, lr_t, beta_1=0.9, beta_2=0.999, epsilon=1e-7, grad_scale=1.0):
        """In-place Keras-Adam update of the flat fp32 device bucket w with gradient g and moments m, v (sr_adam)."""
        import numpy as np
        for t, what in ((w, "adam w"), (g, "adam g"), (m, "adam m"), (v, "adam v")):
            _check_tensor(self, t, what)
            if t.dtype != torch.float32 or t.numel() != w.numel():
                raise ValueError("adam_step: four fp32 tensors of one size")
        f = lambda x: float(np.float32(x))
        self.check(self.lib.sr_adam(self.h, w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), f(lr_t), f(beta_1), f(1.0 - beta_1),
                                    f(beta_2), f(1.0 - beta_2), f(epsilon), f(grad_scale), self.stream()))

    def space_to_depth(self, x, r):
        """Inverse of tf.nn.depth_to_space (DCR): [B,H*r,W*r,C] -> [B,H,W,r*r*C]."""
        _check_tensor(self, x, "space_to_depth input")
        B, Hr, Wr, Cx = x.shape
        if Hr % r or Wr % r:
            raise ValueError("space_to_depth: spatial size not divisible by the block")
        y = self.empty((B, Hr // r, Wr // r, r * r * Cx), torch.float32)
        self.check(self.lib.sr_space_to_depth(self.h, x.data_ptr(), B, Hr // r, Wr // r, Cx, int(r), y.data_ptr(), self.stream()))
        return y

    def spatial_op(self, op, x):
        """L.SP_MAXPOOL2 / SP_GAP / SP_PICK2 / SP_VGG_PREPROCESS on an fp32 [B,H,W,C] tensor."""
        _check_tensor(self, x, "spatial op input")
        B, H, W, Cx = x.shape
        shape = {L.SP_MAXPOOL2: (B, H // 2, W // 2, Cx), L.SP_GAP: (B, Cx), L.SP_PICK2: (B, (H + 1) // 2, (W + 1) // 2, Cx),
                 L.SP_VGG_PREPROCESS: (B, H, W, Cx)}[op]
        y = self.empty(shape, torch.float32)
        self.check(self.lib.sr_spatial_op(self.h, int(op), x.data_ptr(), B, H, W, Cx, y.data_ptr(), self.stream()))
        return y

    def matmul(self, a, b, trans_a=False, trans_b=False, alpha=1.0):
        """Batched fp32 product alpha * op(a) @ op(b): a [batch, M, K] (or [batch, K, M] when trans_a), b likewise."""
        _check_tensor(self, a, "matmul a")
        _check_tensor(self, b, "matmul b")
        if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0]:
            raise ValueError("matmul operands must be [batch, rows, cols] with equal batch")
        M, K = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
        K2, N = (b.shape[2], b.shape[1]) if trans_b else (b.shape[1], b.shape[2])
        if K != K2:
            raise ValueError("matmul inner dimensions differ")
        c = self.empty((a.shape[0], M, N), torch.float32)
        self.check(self.lib.sr_matmul(self.h, a.data_ptr(), b.data_ptr(), c.data_ptr(), a.shape[0], M, N, K, int(trans_a), int(trans_b), float(alpha), self.stream()))
        return c

    def softmax_rows_(self, s):
        """softmax over the last axis, in place."""
        _check_tensor(self, s, "softmax input")
        self.check(self.lib.sr_softmax_rows(self.h, s.data_ptr(), s.numel() // s.shape[-1], s.shape[-1], self.stream()))
        return s

    def softmax_bwd(self, p, dp):
        _check_tensor(self, p, "softmax_bwd p")
        _check_tensor(self, dp, "softmax_bwd dp")
        ds = torch.empty_like(p)
        self.check(self.lib.sr_softmax_bwd(self.h, p.data_ptr(), dp.data_ptr(), ds.data_ptr(), p.numel() // p.shape[-1], p.shape[-1], self.stream()))
        return ds

    def maxpool2_bwd(self, x, dy):
        _check_tensor(self, x, "maxpool_bwd x")
        _check_tensor(self, dy, "maxpool_bwd dy")
        B, H, W, Cx = x.shape
        dx = torch.empty_like(x)
        self.check(self.lib.sr_maxpool2_bwd(self.h, x.data_ptr(), dy.data_ptr(), B, H, W, Cx, dx.data_ptr(), self.stream()))
        return dx

    def zero_insert2(self, dy, H, W):
        """Adjoint of SP_PICK2: dy [B,ceil(H/2),ceil(W/2),C] -> [B,H,W,C]."""
        _check_tensor(self, dy, "zero_insert dy")
        B, _, _, Cx = dy.shape
        out = self.empty((B, H, W, Cx), torch.float32)
        self.check(self.lib.sr_zero_insert2(self.h, dy.data_ptr(), B, int(H), int(W), Cx, out.data_ptr(), self.stream()))
        return out

    def spectral_l1_bwd(self, a, b, scale=1.0):
        _check_tensor(self, a, "spectral_bwd a")
        _check_tensor(self, b, "spectral_bwd b")
        B, H, W, Cx = a.shape
        da = torch.empty_like(a)
        self.check(self.lib.sr_spectral_l1_bwd(self.h, a.data_ptr(), b.data_ptr(), B, H, W, Cx, float(scale), da.data_ptr(), self.stream()))
        return da

    def l1(self, a, b):
        """mean |a - b| (ESRGAN _pixel_loss) -> [1] tensor."""
        _check_tensor(self, a, "l1 input a")
        _check_tensor(self, b, "l1 input b")
        if a.shape != b.shape:
            raise ValueError("l1 inputs must have the same shape")
        out = self.empty((1,), torch.float32)
        self.check(self.lib.sr_l1(self.h, a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), self.stream()))
        return out

    def spectral_l1(self, a, b):
        """mean | |fft2(a)| - |fft2(b)| | over the (W, C) axes of [B,H,W,3] tensors (ESRGAN _spectral_loss) -> [1] tensor."""
        _check_tensor(self, a, "spectral input a")
        _check_tensor(self, b, "spectral input b")
        if a.shape != b.shape or a.dim() != 4:
            raise ValueError("spectral loss inputs must be two [B,H,W,3] tensors of the same shape")
        B, H, W, Cx = a.shape
        out = self.empty((1,), torch.float32)
        self.check(self.lib.sr_spectral_l1(self.h, a.data_ptr(), b.data_ptr(), B, H, W, Cx, out.data_ptr(), self.stream()))
        return out

    def num_patches(self, H, W, C_, patch, stride):
        n = C.c_int()
        self.check(self.lib.sr_extract_patches(self.h, None, H, W, C_, patch, stride, 1.0, 0.0, L.DTYPE_F32, None, 0, C.byref(n), None))
        return n.value

    def extract_patches(self, img, patch, stride, mul=1.0, add=0.0, out_dtype=torch.float32):
        _check_tensor(self, img, "extract_patches image")
        H, W, Cx = img.shape
        n = self.num_patches(H, W, Cx, patch, stride)
        out = self.empty((n, patch, patch, Cx), out_dtype)
        cnt = C.c_int()
        self.check(self.lib.sr_extract_patches(self.h, img.data_ptr(), H, W, Cx, patch, stride, float(mul), float(add),
                                               dtype_code(out_dtype), out.data_ptr(), out.numel(), C.byref(cnt), self.stream()))
        return out

    def overlap_add(self, patches, H, W, patch, stride, scale=1, mul=1.0, add=0.0):
        _check_tensor(self, patches, "overlap_add patches", (torch.float32, torch.bfloat16))
        Cx = patches.shape[-1]
        out = self.empty((H * scale, W * scale, Cx), torch.float32)
        self.check(self.lib.sr_overlap_add(self.h, patches.data_ptr(), dtype_code(patches.dtype), H, W, Cx, patch, stride, scale,
                                           float(mul), float(add), out.data_ptr(), self.stream()))
        return out


class Model:
    """sr_model wrapper: build from the reference's setup_model() hyper-parameters, load Keras-named
    weights, run forward on device tensors."""
    KINDS = {"srcnn": L.MODEL_SRCNN, "edsr": L.MODEL_EDSR, "esrgan_g": L.MODEL_ESRGAN_G, "vgg16": L.MODEL_VGG16,
             "esrgan_d": L.MODEL_ESRGAN_D, "vgg19_features": L.MODEL_VGG19_FEATURES}

    def __init__(self, kind, compute_dtype="f32", scale_factor=1, channels=3, num_blocks=0, num_filters=64,
                 growth_channels=32, res_scaling=0.1, num_classes=2, use_attention=True, ctx=None):
        self.ctx = ctx or Context.get()
        self.kind = kind
        self.compute_dtype = _DT2TORCH[dtype_code(compute_dtype)]
        cfg = L.ModelCfg(dtype_code(compute_dtype), int(scale_factor), int(channels), int(num_blocks), int(num_filters),
                         int(growth_channels), float(res_scaling), int(num_classes), int(bool(use_attention)))
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.sr_model_create(self.ctx.h, self.KINDS[kind], C.byref(cfg), C.byref(h)))
        self.h = h
        self.finalized = False

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.sr_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def param_specs(self):
        """[(keras_layer_name, 'kernel'|'bias', shape)] in graph order."""
        lib = self.ctx.lib
        out = []
        for i in range(lib.sr_model_num_params(self.h)):
            name, which, nd = C.c_char_p(), C.c_int(), C.c_int()
            shape = (C.c_int64 * 4)()
            self.ctx.check(lib.sr_model_param_info(self.h, i, C.byref(name), C.byref(which), shape, C.byref(nd)))
            out.append((name.value.decode(), "bias" if which.value else "kernel", tuple(shape[:nd.value])))
        return out

    def layer_shapes(self):
        """[(layer_name, kernel_shape)] -- one entry per Keras layer."""
        return [(n, s) for n, w, s in self.param_specs() if w == "kernel"]

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, _, s in self.param_specs()))

    def set_weights(self, weights):
        """weights: {layer_name: (kernel, bias)} (conv HWIO, dense [in,out]).  Missing layers -> KeyError."""
        lib = self.ctx.lib
        for name, kshape in self.layer_shapes():
            if name not in weights:
                raise KeyError(f"weights for layer '{name}' missing")
            k, b = weights[name]
            for which, arr in ((L.WEIGHT_KERNEL, k), (L.WEIGHT_BIAS, b)):
                a = _np32(arr)
                shp = (C.c_int64 * a.ndim)(*a.shape)
                self.ctx.check(lib.sr_model_set_weight(self.h, name.encode(), which, _fptr(a), shp, a.ndim))
        self.ctx.check(lib.sr_model_finalize(self.h))
        self.finalized = True

    def output_shape(self, B, H, W, Cx):
        s = (C.c_int64 * 4)()
        self.ctx.check(self.ctx.lib.sr_model_output_shape(self.h, B, H, W, Cx, s))
        return tuple(s) if self.kind not in ("vgg16", "esrgan_d") else (s[0], s[1])

    def forward(self, x, out=None):
        """x [B,H,W,C] device tensor (f32, or bf16 for a bf16 model) -> output tensor of the same dtype."""
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError("expected a [B,H,W,C] tensor")
        x = x.contiguous()
        _check_tensor(self.ctx, x, "forward input", (torch.float32, torch.bfloat16))
        B, H, W, Cx = x.shape
        oshape = self.output_shape(B, H, W, Cx)
        if out is not None:
            _check_tensor(self.ctx, out, "forward out=", (x.dtype,))
            if tuple(out.shape) != tuple(oshape):
                raise ValueError(f"forward out= has shape {tuple(out.shape)}, the model produces {tuple(oshape)}")
        y = out if out is not None else self.ctx.empty(oshape, x.dtype)
        self.ctx.check(self.ctx.lib.sr_forward(self.h, x.data_ptr(), dtype_code(x.dtype), B, H, W, Cx, y.data_ptr(), y.numel(),
                                               self.ctx.stream()))
        return y

    # ------------------------------------------------------------------ workspaces / diagnostics
    def release_workspace(self):
        """Free the activation workspaces (grow-only otherwise); weights stay loaded."""
        self.ctx.check(self.ctx.lib.sr_model_release_workspace(self.h))

    def ops(self):
        """[(name, channels, mul, shift, ceil_halvings)] per graph op: Keras layer name of a conv, else the op kind; the op's output is
        [B, h, w, channels] with h = (H*mul)>>shift, then ceil-halved ceil_halvings times (channels 0: no activation output)."""
        lib, out = self.ctx.lib, []
        for i in range(lib.sr_model_num_ops(self.h)):
            name, ch, mul, sh, cs = C.c_char_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
            self.ctx.check(lib.sr_model_op_info(self.h, i, C.byref(name), C.byref(ch), C.byref(mul), C.byref(sh), C.byref(cs)))
            out.append((name.value.decode(), ch.value, mul.value, sh.value, cs.value))
        return out

    @staticmethod
    def _op_hw(op, H, W):
        h, w = (H * op[2]) >> op[3], (W * op[2]) >> op[3]
        for _ in range(op[4]):
            h, w = (h + 1) // 2, (w + 1) // 2
        return h, w

    def forward_with_taps(self, x, names):
        """Diagnostic forward: -> (y, {name: fp32 NHWC tensor}) with the output of the LAST op called `name` for every name
        (stage-by-stage parity traces; the taps are removed again before returning)."""
        ops = self.ops()
        B, H, W, _ = x.shape
        taps, idx = {}, {}
        for n in names:
            hits = [i for i, o in enumerate(ops) if o[0] == n and o[1] > 0]
            if not hits:
                raise KeyError(f"no op named '{n}' with an activation output")
            idx[n] = hits[-1]
        try:
            for n, i in idx.items():
                th, tw = self._op_hw(ops[i], H, W)
                t = self.ctx.empty((B, th, tw, ops[i][1]), torch.float32)
                self.ctx.check(self.ctx.lib.sr_model_set_tap(self.h, i, t.data_ptr(), t.numel()))
                taps[n] = t
            y = self.forward(x)
        finally:
            for i in idx.values():
                self.ctx.lib.sr_model_set_tap(self.h, i, None, 0)
        return y, taps

    def predict(self, x, batch_size=32):
        """keras Model.predict(x, batch_size): forward in chunks, outputs concatenated.  Accepts NumPy
        (returns NumPy, as Keras does) or a device tensor (returns a device tensor)."""
        is_np = not isinstance(x, torch.Tensor)
        xt = self.ctx.to_device(x, torch.float32) if is_np else x
        n = xt.shape[0]
        oshape = self.output_shape(n, *xt.shape[1:])
        y = self.ctx.empty(oshape, xt.dtype)
        for i in range(0, n, max(1, int(batch_size))):
            self.forward(xt[i:i + batch_size], out=y[i:i + batch_size])
        if is_np:
            return y.float().cpu().numpy()
        return y
