"""Runtime behaviour of the C ABI that is not arithmetic: the out-of-memory path of sr_forward (workspaces released, a smaller
retry starts clean) and the host-side tensor checks in front of the raw-pointer entry points."""
import numpy as np
import pytest
import torch

from sr355 import Model
from sr355.weights import init_weights

pytestmark = pytest.mark.gpu


def test_forward_oom_releases_workspaces_and_a_smaller_retry_is_clean(ctx):
    """a failed workspace growth must not leave the half-grown oversized buffers behind (bench.py halves its tiles per
    call after MemoryError and retries).  An artificial byte cap stands in for a full card."""
    m = Model("esrgan_g", compute_dtype="bf16", scale_factor=2, num_blocks=1, growth_channels=32, use_attention=False, ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), seed=1))
    base = ctx.mem_info()["current"]                      # weights only
    small = ctx.to_device(np.zeros((2, 24, 24, 3), np.float32), torch.bfloat16)
    big = ctx.to_device(np.zeros((64, 48, 48, 3), np.float32), torch.bfloat16)
    y0 = m.forward(small).clone()
    torch.cuda.synchronize()
    small_bytes = ctx.mem_info()["current"]
    assert small_bytes > base
    try:
        ctx.set_alloc_cap(small_bytes + (4 << 20))        # room for a few MB more: the 64-patch workspaces (~150 MB) cannot fit
        with pytest.raises(MemoryError) as ei:
            m.forward(big)
        assert "released" in str(ei.value)
        assert ctx.mem_info()["current"] == base          # nothing of the failed growth (or the earlier small set) is left
        y1 = m.forward(small)                             # the retry: fresh small workspaces, same numbers
        torch.cuda.synchronize()
        assert ctx.mem_info()["current"] == small_bytes
        assert torch.equal(y0, y1)
    finally:
        ctx.set_alloc_cap(0)
    m.forward(big)                                        # and without the cap the large batch runs
    m.release_workspace()
    assert ctx.mem_info()["current"] == base


def test_raw_pointer_entry_points_reject_foreign_tensors(ctx):
    a = ctx.to_device(np.random.default_rng(0).uniform(0, 1, (1, 16, 16, 3)).astype(np.float32))
    with pytest.raises(ValueError):
        ctx.psnr(a, a.to(torch.bfloat16))                 # wrong dtype would be read as float
    with pytest.raises(ValueError):
        ctx.ssim(a.cpu(), a.cpu())                        # host memory
    with pytest.raises(ValueError):
        ctx.psnr(a[:, ::2], a[:, ::2])                    # strided view
    with pytest.raises(ValueError):
        ctx.mse(a, a[:, :8])
    m = Model("srcnn", compute_dtype="f32", ctx=ctx)
    m.set_weights(init_weights(m.layer_shapes(), seed=2))
    with pytest.raises(ValueError):
        m.forward(a, out=torch.empty(1, 16, 16, 3, dtype=torch.bfloat16, device=a.device))
    with pytest.raises(ValueError):
        m.forward(a, out=torch.empty(1, 8, 8, 3, dtype=torch.float32, device=a.device))
    assert m.forward(a).shape == (1, 16, 16, 3)
    mhz = ctx.measure_clock_mhz()
    assert 800.0 < mhz < 2600.0, mhz


def test_stamp_buffers_are_bounded(ctx):
    """VERDICT r2 #8 / DESIGN 3.4: round 2's one fault was a probe handing the 3x3 conv's stamped build a buffer sized for another
    kernel.  The setters now take the capacity: an undersized fused-kernel buffer is refused at once, a conv launch whose grid would
    write beyond its buffer is refused instead of run, and a buffer that fits works."""
    L = ctx.lib
    small = torch.zeros(1024, dtype=torch.int64, device="cuda")                       # 8 KiB
    assert L.sr_debug_set_chain_stamp_buffer(ctx.h, small.data_ptr(), small.numel() * 8) == -1
    assert b"smaller" in L.sr_last_error(ctx.h)
    assert L.sr_debug_set_stamp_buffer(ctx.h, small.data_ptr(), 64) == -1            # not even one workgroup's 16 stamps
    x = torch.randn(8, 48, 48, 32, device="cuda").to(torch.bfloat16)
    w = (np.random.default_rng(1).standard_normal((3, 3, 32, 32)) / 17.0).astype(np.float32)
    ref = ctx.conv2d(x, w, None, act="relu").clone()
    nwg = 8 * 3 * 3                                                                   # 16x16 tiles of 8 images 48x48, one 32-cout tile
    try:
        assert L.sr_debug_set_stamp_buffer(ctx.h, small.data_ptr(), small.numel() * 8) == 0   # room for 64 workgroups only
        with pytest.raises(Exception) as ei:
            ctx.conv2d(x, w, None, act="relu")
        assert "stamp buffer is too small" in str(ei.value)
        big = torch.zeros(int(L.sr_debug_stamp_bytes_needed(0, nwg)) // 8, dtype=torch.int64, device="cuda")
        assert L.sr_debug_set_stamp_buffer(ctx.h, big.data_ptr(), big.numel() * 8) == 0
        got = ctx.conv2d(x, w, None, act="relu")
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        assert int((big.view(-1, 16)[:, 0] != 0).sum()) == nwg
    finally:
        assert L.sr_debug_set_stamp_buffer(ctx.h, None, 0) == 0
