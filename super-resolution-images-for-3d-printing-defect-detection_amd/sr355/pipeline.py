"""Patch-wise inference plumbing shared by the three SR wrappers and the classifier.

Device restatement of the helper trio every reference `super_resolve_image` defines locally
(add_padding / extract_patches / reconstruct_from_patches: SRCNN_model.py:127-188,
EDSR_model.py:201-256, ESRGAN_model.py:883-921).  Patches are cut, pushed through the model and
overlap-averaged without leaving the GPU; only the reference's NumPy-in / NumPy-out contract at
the very edge copies over PCIe.
"""
import time

import numpy as np
import torch

from .runtime import Context


def pad_amount(n, patch, stride):
    """loading_methods.py:12-17."""
    pad = (patch - (n % stride)) % stride if n % stride != 0 else 0
    return max(pad, patch - stride)


def patch_grid(h, w, patch, stride):
    """number of window positions (rows, cols) over the reflect-padded image."""
    hp, wp = h + pad_amount(h, patch, stride), w + pad_amount(w, patch, stride)
    return len(range(0, hp - patch + 1, stride)), len(range(0, wp - patch + 1, stride))


def as_device_image(ctx, img):
    """Reference inputs: np.ndarray RGB, uint8 [0,255] or float; tensors pass through."""
    if isinstance(img, torch.Tensor):
        return img.to(ctx.torch_device, torch.float32).contiguous(), False
    a = np.asarray(img)
    return ctx.to_device(a.astype(np.float32, copy=False)), True


def inference_metrics(ctx, elapsed, mem_begin, mem_end):
    """Same keys and arithmetic as the reference's metrics dict (SRCNN_model.py:215-242); memory is
    what libsr355 holds on the device (weights + workspaces)."""
    mb = lambda x: float(x) / (1024.0 * 1024.0)
    return {
        "time_sec": float(elapsed),
        "gpu_mean_current_mb": mb((mem_begin["current"] + mem_end["current"]) / 2.0),
        "gpu_peak_mb": mb(max(mem_begin["peak"], mem_end["peak"])),
    }


def patchwise_sr(model, lr, patch, stride, scale, chunk, in_mul=1.0, in_add=0.0, out_mul=1.0, out_add=0.0):
    """lr [H,W,3] fp32 device tensor in [0,1] -> (sr [H*scale,W*scale,3] fp32 device tensor, metrics)."""
    ctx = model.ctx
    H, W, _ = lr.shape
    patches = ctx.extract_patches(lr, patch, stride, mul=in_mul, add=in_add)
    mem0 = ctx.mem_info()
    torch.cuda.synchronize(ctx.torch_device)
    t0 = time.perf_counter()
    hr = model.predict(patches, batch_size=chunk)
    torch.cuda.synchronize(ctx.torch_device)
    elapsed = time.perf_counter() - t0
    sr = ctx.overlap_add(hr, H, W, patch, stride, scale, mul=out_mul, add=out_add)
    return sr, inference_metrics(ctx, elapsed, mem0, ctx.mem_info())


def patchwise_sr_many(model, lrs, patch, stride, scale, chunk, in_mul=1.0, in_add=0.0, out_mul=1.0, out_add=0.0, timed=True):
    """Same as patchwise_sr for several equally sized images at once: their patches share the predict() calls (Keras
    predict is chunk-invariant, so results are identical); bigger launches waste less of the last wave of workgroups."""
    ctx = model.ctx
    H, W, _ = lrs[0].shape
    groups = [ctx.extract_patches(lr, patch, stride, mul=in_mul, add=in_add) for lr in lrs]
    n = groups[0].shape[0]
    mem0 = ctx.mem_info()
    if timed:                                   # the reference brackets predict() with a host clock; needs the device idle
        torch.cuda.synchronize(ctx.torch_device)
    t0 = time.perf_counter()
    hr = model.predict(torch.cat(groups, dim=0), batch_size=chunk)
    if timed:
        torch.cuda.synchronize(ctx.torch_device)
    elapsed = time.perf_counter() - t0          # timed=False: enqueue time only, the work stays asynchronous on the stream
    srs = [ctx.overlap_add(hr[i * n:(i + 1) * n], H, W, patch, stride, scale, mul=out_mul, add=out_add) for i in range(len(lrs))]
    return srs, inference_metrics(ctx, elapsed, mem0, ctx.mem_info())


def majority_vote(probs):
    """VGG16_model.py:252-268 on host (a handful of scalars)."""
    probs = np.asarray(probs)
    if probs.ndim != 2:
        probs = probs.reshape((probs.shape[0], -1))
    votes = np.bincount(np.argmax(probs, axis=1), minlength=int(probs.shape[1]))
    tied = np.where(votes == votes.max())[0]
    win = int(tied[0]) if len(tied) == 1 else int(tied[np.argmax(probs.mean(axis=0)[tied])])
    return win, float(probs[:, win].mean())


def sr_then_classify(sr_model, classifier, lr_img, sr_kwargs=None, patch_size=96, stride=48, batch_size=32):
    """The (missing) defect_detection_pipeline notebook's inner loop, reconstructed from the helpers written for it
    (SURVEY.md 3.5): LR image -> `super_resolve_image` -> `classify_defects_method` on the SR output.  The SR image
    stays on the device between the two models.  Returns (sr_img device tensor, inference_metrics, class, confidence)."""
    ctx = sr_model.ctx
    lr, _ = as_device_image(ctx, lr_img)
    sr, metrics = sr_model.super_resolve_image(lr, **(sr_kwargs or {}))
    cls, conf = classifier.classify_defects_method(sr, patch_size=patch_size, stride=stride, batch_size=batch_size)
    return sr, metrics, cls, conf


def stream_sr_classify(sr_model, classifier, frames, sr_kwargs=None, patch_size=96, stride=48, batch_size=512, rank=0, world=1,
                       keep_sr=False):
    """BASELINE configs[4]: a stream of LR frames -> x4 super-resolution -> VGG16 patch vote per frame, everything between the host
    frame and the (class, confidence) pair resident on the GPU.

    Reconstructed from the helpers the reference wrote for its (missing) defect_detection_pipeline notebook: per frame
    `super_resolve_image` (ESRGAN_model.py:858-979) then `classify_defects_method` on the SR image (VGG16_model.py:168-270); the
    ingest contract is `load_predictions_dataset` (loading_methods.py:288-386): arrays of whole frames, uint8 or float RGB.

    * frames are independent units: rank r of `world` takes the contiguous slice dist.shard_range gives it -- no data-path
      collective;
    * double-buffered ingest: frame i+1 crosses PCIe (pinned staging, a second HIP stream) while frame i is in the generator;
    * the SR frame never leaves HBM: the classifier's 96x96 patches are cut from it on the device.

    frames: sequence of [H,W,3] arrays (uint8 [0,255] or float [0,1]) or device tensors.  Returns
    ([{'frame', 'class', 'confidence', 'sr' (only if keep_sr)}...] for this rank's frames, stats dict)."""
    from .dist import shard_range
    ctx = sr_model.ctx
    dev = ctx.torch_device
    lo, hi = shard_range(len(frames), rank, world)
    mine = list(range(lo, hi))
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    staged = [None, None]                                   # (device tensor, ready event) per buffer

    def ingest(slot, idx):
        f = frames[idx]
        if isinstance(f, torch.Tensor):
            t = f.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
            staged[slot] = (t, ev)
            return
        a = np.ascontiguousarray(np.asarray(f))
        host = torch.from_numpy(a).pin_memory()
        with torch.cuda.stream(copy_stream):
            t = host.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        staged[slot] = (t, ev, host)                        # keep the pinned buffer alive until the copy has been consumed

    results = []
    t_sr = t_cls = 0.0
    out_pix = 0
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if mine:
        ingest(0, mine[0])
    for k, idx in enumerate(mine):
        cur = staged[k & 1]
        if k + 1 < len(mine):
            ingest((k + 1) & 1, mine[k + 1])                # next frame's copy overlaps this frame's compute
        main.wait_event(cur[1])
        x = cur[0]
        x.record_stream(main)                               # allocated on the copy stream, consumed on this one: the allocator must not hand
                                                            # the block to the next ingest while `main` still reads it
        lr = (x.to(torch.float32) / 255.0) if x.dtype == torch.uint8 else x.to(torch.float32)
        a = time.perf_counter()
        sr, _ = sr_model.super_resolve_image(lr.contiguous(), **(sr_kwargs or {}))
        b = time.perf_counter()
        cls, conf = classifier.classify_defects_method(sr, patch_size=patch_size, stride=stride, batch_size=batch_size)
        c = time.perf_counter()                             # the vote needs the probabilities on the host: the frame is complete here
        t_sr += b - a
        t_cls += c - b
        out_pix += int(sr.shape[0]) * int(sr.shape[1])
        rec = {"frame": idx, "class": int(cls), "confidence": float(conf)}
        if keep_sr:
            rec["sr"] = sr
        results.append(rec)
        staged[k & 1] = None
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    n = max(len(mine), 1)
    stats = {"frames": len(mine), "wall_s": wall, "frames_per_s": len(mine) / wall if wall > 0 else 0.0,
             "sr_output_mpix_per_s": out_pix / 1e6 / wall if wall > 0 else 0.0,
             "host_ms_per_frame_sr_enqueue_plus_wait": 1e3 * t_sr / n, "host_ms_per_frame_classify": 1e3 * t_cls / n}
    return results, stats
