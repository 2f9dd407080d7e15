#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: 4x-SR output MPix/s on 512x512 LR tiles (configs[2]).

One step = one pass of the hot path over one batch: 16 LR tiles [512,512,3] (resident in HBM) ->
ESRGAN.super_resolve_image (reference patch mode: reflect pad, 441 LR patches 48x48 stride 24 per tile,
RRDB generator x4 NB=23 G=32 with both SelfAttention layers, bf16 storage / fp32 accumulate, overlap
average, crop, clip) -> PSNR/SSIM against the HR tiles -> (N>1) all-reduce of the metric sums over RCCL (sr355/dist.py).
N>1 default = SURVEY.md 8(e)'s partition: the 16-tile batch is split B/g over the ranks ("scaling": "strong": 16/8/4/2 tiles per
GPU at 1/2/4/8); --scaling weak gives every rank its own 16 tiles.  value = SR output megapixels of all ranks / wall time
(max over ranks).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...          (no torchrun environment: bench.py starts the N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

TILES_PER_GPU, LR, SCALE, NB, G, PATCH, STRIDE = 16, 512, 4, 23, 32, 48, 24
PEAK_HBM_GBPS = 8000.0        # HBM3E, MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


CONV_TFLOP_PER_TILE = 36.48  # conv FLOP one 512x512 LR tile needs in reference patch mode (441 patches 48x48)
# LR patches of tile 0 pushed through both the GPU and the oracle for the parity object: a 4 x 4 grid over the tile's 21 x 21 patch
# positions (corners, edges, interior), all inside the un-padded tile so that each has a true HR counterpart
PARITY_GRID = (0, 6, 13, 19)
PARITY_IDX = [r * 21 + c for r in PARITY_GRID for c in PARITY_GRID]


def tile_patches(lr_tile):
    from oracle import ops as OO
    padded = OO.add_padding(lr_tile, PATCH, STRIDE)
    patches, _ = OO.extract_patches(padded, PATCH, STRIDE)
    return patches * 2.0 - 1.0


def hr_patches(hr_tile, idx):
    """HR counterparts [n, 192, 192, 3] in [0, 1] of the LR patches `idx` of a tile (positions as extract_patches enumerates them)."""
    out = []
    for i in idx:
        r, c = divmod(i, 21)
        y, x = r * STRIDE * SCALE, c * STRIDE * SCALE
        out.append(hr_tile[y:y + PATCH * SCALE, x:x + PATCH * SCALE])
    return np.stack(out)


def host_cpu():
    """(logical CPUs of the node, CPUs this process may run on, model name) -- the north star asks for the core count beside the CPU number."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    return os.cpu_count(), usable, model


def cpu_baseline(weights, lr_tile, budget_s=20.0, keep=None):
    """The oracle (CPU restatement, torch-CPU fp32) on a bounded sample of the same patches: the PARITY_IDX patches first (their
    outputs go to `keep`, a dict, for the parity object), then further patches of the tile until the time budget is used."""
    from oracle import models as OM
    n_cpu, usable, model = host_cpu()
    threads = min(usable or 1, 16)     # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)
    x = tile_patches(lr_tile)
    order = PARITY_IDX + [i for i in range(len(x)) if i not in set(PARITY_IDX)]
    outs, n = [], 0
    t0 = time.perf_counter()
    while n < len(order) and (n < len(PARITY_IDX) or time.perf_counter() - t0 < budget_s):
        outs.append(OM.esrgan_g_forward(x[order[n:n + 2]], weights, SCALE, NB))
        n += 2
    dt = time.perf_counter() - t0
    n = min(n, len(order))
    if keep is not None:
        keep["fp32_reference_graph"] = np.concatenate(outs)[:len(PARITY_IDX)]
    per_tile = dt / n * len(x)
    return {"value": (LR * SCALE) ** 2 / 1e6 / per_tile, "unit": "MPix/s", "cores": threads, "cpu_count": n_cpu, "cpu_model": model,
            "kind": "port",
            "sample": f"{n} of {TILES_PER_GPU * len(x)} LR patches 48x48 (ESRGAN x4 NB=23 G=32 with attention, fp32 torch-CPU oracle, {threads} threads "
                      f"on a {n_cpu}-CPU host), {dt:.1f} s, extrapolated to a 441-patch tile"}


def parity_object(ctx, model, weights, lr_tile, hr_tile, fp32_ref, io="bf16", like_for_like=None):
    """GPU (the bench's bf16 generator) vs the oracle on the PARITY_IDX LR patches of tile 0, outside the timed region.
    Like-for-like = the oracle in its bf16-storage mode (rounds to bf16 where the device stores bf16); the plain fp32 reference
    graph is reported beside it.  abs_psnr_delta_vs_hr_db is the north star's figure |PSNR(gpu, HR) - PSNR(oracle, HR)| (<= 0.01 dB),
    per patch, worst case, against the fp32 reference graph (and against the bf16-storage oracle beside it).
    io: dtype of the caller's tensors -- "f32" is what super_resolve_image hands the generator (fp32 patches in, fp32 image out: the
    product path of the timed step), "bf16" a caller that keeps its tensors in bf16 (the generator's output is then rounded once more).
    fp32_ref: the fp32 reference graph's outputs for these patches if already computed (cpu_baseline), else None = computed here.
    like_for_like: how many of the patches the bf16-storage oracle is run on (None = all; it is the second CPU forward per patch)."""
    from oracle import models as OM
    from oracle import ops as OO
    from sr355.weights import round_to_bf16
    x = round_to_bf16(tile_patches(lr_tile)[PARITY_IDX].astype(np.float32))
    got = model.generator.forward(ctx.to_device(x, torch.bfloat16 if io == "bf16" else torch.float32)).float().cpu().numpy()
    nl = len(x) if like_for_like is None else max(2, min(len(x), int(like_for_like)))
    ref = np.concatenate([OM.esrgan_g_forward(x[i:i + 2], weights, SCALE, NB, bf16_storage=True, bf16_output=io == "bf16") for i in range(0, nl, 2)])
    if fp32_ref is None:
        fp32_ref = np.concatenate([OM.esrgan_g_forward(x[i:i + 2], weights, SCALE, NB) for i in range(0, len(x), 2)])
    hr = hr_patches(hr_tile, PARITY_IDX).astype(np.float64)
    to01 = lambda a: np.clip((a.astype(np.float64) + 1) / 2, 0.0, 1.0)
    p01 = lambda a, b: OO.psnr(to01(a), to01(b), dtype=np.float64)
    vs_hr = lambda a: OO.psnr(hr, to01(a), dtype=np.float64)
    f32r = fp32_ref[:len(x)]
    return {"psnr_gpu_vs_oracle_db": float(p01(got[:nl], ref).min()), "max_abs": float(np.abs(got[:nl] - ref).max()),
            "rel_l2": float(np.linalg.norm(got[:nl] - ref) / np.linalg.norm(ref)), "n_patches": int(len(x)), "n_patches_bf16_storage_oracle": int(nl), "caller_tensors": io,
            "patches": "4 x 4 grid over tile 0's 21 x 21 patch positions (rows/cols 0, 6, 13, 19)",
            "abs_psnr_delta_vs_hr_db_bf16_storage_oracle": float(np.abs(vs_hr(got)[:nl] - OO.psnr(hr[:nl], to01(ref), dtype=np.float64)).max()),
            "oracle": "CPU restatement, fp32 arithmetic, bf16 storage where the device stores bf16 (oracle.models bf16_storage=True)",
            "psnr_gpu_vs_fp32_reference_graph_db": float(p01(got, f32r).min()),
            "psnr_fp32_reference_graph_vs_hr_db": [float(vs_hr(f32r).min()), float(vs_hr(f32r).max())],
            "abs_psnr_delta_vs_hr_db": float(np.abs(vs_hr(got) - vs_hr(f32r)).max()),
            "mean_psnr_delta_vs_hr_db": float((vs_hr(got) - vs_hr(f32r)).mean()), "north_star_bar_db": 0.01}


TRAINED_LIKE_LEVELS = (60, 300)


def trained_like_parity(ctx, model_bf16, lr4, hr4, levels=TRAINED_LIKE_LEVELS, log=None):
    """The same parity object on weights in the regime the reference's generators work in (sr355.recipes: analytic start + L1 steps on
    crops of the bench's own tiles, seeded), at two points of ONE fit: after levels[0] steps (28-32 dB against HR: where the reference's
    trained models are, ESRGAN.ipynb:L3723-3725) and after levels[1] (33-37 dB: a stress level); random-init weights give 9.6 dB.  Per
    level: the 16 parity patches against the CPU oracle's fp32 graph (caller tensors fp32, as super_resolve_image hands them over), and
    every bench tile whole, in reference patch mode: PSNR(bf16 generator, HR) against PSNR(fp32 generator, HR) -- the fp32 device path
    being the one the 16 patches pin to the CPU oracle (psnr_gpu_f32_vs_fp32_reference_graph_db).  Leaves the model on the last weights.
    -> ({"steps_<n>": {...}, "weights": ..., "fit_seconds": ...}, last weights)."""
    from oracle import models as OM
    from oracle import ops as OO
    from sr355.recipes import GeneratorPixelFit, crop_batches, near_identity_generator
    from sr355.weights import bf16_rounded, round_to_bf16
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN
    fit = GeneratorPixelFit(ctx, near_identity_generator(model_bf16.generator.layer_shapes()), SCALE, NB, True, 2e-4)
    batches = crop_batches(lr4, hr4, SCALE, 24, 16, max(levels), 7001)
    m32 = ESRGAN(compute_dtype="f32")
    m32.setup_model(scale_factor=SCALE, growth_channels=G, num_rrdb_blocks=NB)
    x = round_to_bf16(tile_patches(lr4[0])[PARITY_IDX].astype(np.float32))
    to01 = lambda a: np.clip((a.astype(np.float64) + 1) / 2, 0.0, 1.0)
    out, done, fit_s, w = {}, 0, 0.0, None
    for lv in sorted(levels):
        t0 = time.perf_counter()
        for _ in range(lv - done):
            l1 = fit.step(*next(batches))
        done = lv
        fit_s += time.perf_counter() - t0
        if log is not None:
            log(lv, l1)
        w = bf16_rounded(fit.weights)
        model_bf16.set_weights(w)
        m32.set_weights(w)
        f32r = np.concatenate([OM.esrgan_g_forward(x[i:i + 2], w, SCALE, NB) for i in range(0, len(x), 2)])
        o = parity_object(ctx, model_bf16, w, lr4[0], hr4[0], f32r, io="f32", like_for_like=4)
        g32 = m32.generator.forward(ctx.to_device(x, torch.float32)).cpu().numpy()
        o["psnr_gpu_f32_vs_fp32_reference_graph_db"] = float(OO.psnr(to01(g32), to01(f32r), dtype=np.float64).min())
        tiles = []
        for t in range(len(lr4)):
            hr_d = ctx.to_device(hr4[t:t + 1])
            ps = []
            for m_ in (model_bf16, m32):
                sr, _ = m_.super_resolve_image(ctx.to_device(lr4[t]), patch_size_lr=PATCH, stride=STRIDE, batch_size=441)
                ps.append(float(ctx.psnr(hr_d, sr[None])[0].double()))
            tiles.append({"tile": t, "psnr_bf16_vs_hr_db": ps[0], "psnr_f32_vs_hr_db": ps[1], "abs_delta_db": abs(ps[0] - ps[1])})
        o["whole_tiles_patch_mode"] = tiles
        o["whole_tile_abs_psnr_delta_vs_hr_db"] = max(t["abs_delta_db"] for t in tiles)
        out[f"steps_{lv}"] = o
    m32.generator.release_workspace()
    del m32, fit
    out["weights"] = ("sr355.recipes: analytic near-identity start, then L1-only Adam steps (lr 2e-4, batch 16, 24 x 24 LR crops of the bench's 4 tiles), "
                      "fp32 on the device, seeded; rounded to bf16 for both the device and the oracle")
    out["fit_seconds"] = fit_s
    return out, w


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n, argv, port):
    """The command that runs this file as n ranks of one node (one rank per GPU over RCCL): what the driver's own torchrun line does."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children and relay rank 0's JSON line.  The parent must not
    hold a GPU context (a process that has initialised the GPU must never be replaced or forked into ranks on this pool): it touches
    no torch.cuda call before this point, and refuses if something already did."""
    if torch.cuda.is_initialized():
        raise RuntimeError("bench.py --gpus N: the launching process already holds a GPU context; start the ranks from a process that has not touched the GPU")
    have = torch.cuda.device_count()          # counting devices does not initialise the GPU on this image
    if n > have and not os.environ.get("SR355_ONE_DEVICE"):
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (SR355_ONE_DEVICE=1 rehearses the N-rank plumbing on one GPU over gloo)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # free_port() closes its socket before torchrun binds the number, so another process can take it in between: when the children die
    # without rank 0 having printed its line (a failed rendezvous prints nothing), try once more on a fresh port
    rc = 1
    for attempt in range(2):
        proc = subprocess.Popen(launch_command(n, argv, free_port()), env=env, stdout=subprocess.PIPE, text=True)
        printed = False
        for ln in proc.stdout:                # rank 0's line (and nothing else) goes to stdout; the ranks' stderr passes straight through
            sys.stdout.write(ln)
            sys.stdout.flush()
            printed = printed or bool(ln.strip())
        rc = proc.wait()
        if rc == 0 or printed:
            break
        print(f"[bench] the {n}-rank launch exited with {rc} before printing a line" + ("; retrying once on another port" if attempt == 0 else ""),
              file=sys.stderr, flush=True)
    return rc


def pmc_traffic(kernel, path=None, fingerprint=None):
    """Per-launch HBM bytes of `kernel` from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/collect_profiles_r02.sh),
    or None: the file is quoted only when it was collected on a library built from the kernel sources this run uses (its `_source_sha256` against
    sr355._lib.source_fingerprint()) -- a kernel change without a re-collection must not leave a stale figure in a driver-written record.
    -> (bytes or None, note)."""
    path = path or os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.isfile(path):
        return None, "no profiles/pmc_traffic.json"
    rec = json.load(open(path))
    if fingerprint is None:
        from sr355._lib import source_fingerprint
        fingerprint = source_fingerprint()
    if rec.get("_source_sha256") != fingerprint:
        return None, ("profiles/pmc_traffic.json was collected on other kernel sources (sha256 " + str(rec.get("_source_sha256"))[:12] + " != " + fingerprint[:12] +
                      "): not quoted; re-collect with tools/collect_profiles_r02.sh")
    if kernel not in rec:
        return None, "profiles/pmc_traffic.json holds no record for this kernel"
    return rec[kernel], "profiles/pmc_traffic.json (separate --pmc passes, 2 x FETCH_SIZE + WRITE_SIZE; kernel sources sha256 " + fingerprint[:12] + ")"


def roofline_object(dom, traffic, instrumented_ms_per_step, clock_mhz=None):
    """The `roofline` object of the bench line for the dominant kernel's profile record `dom` = {kernel, launches, total_ms, flops,
    bytes} (sums over its launches; flops / bytes are ALGORITHMIC).  SURVEY.md 8(d) and the north star define this path's roof as
    CONV ARITHMETIC: achieved = algorithmic conv FLOP / launch time against the dense bf16 MFMA peak (MI355X_MICROARCH.md:
    ~2.5 PFLOP/s).  The HBM view of the same launches (algorithmic bytes / time against 8 TB/s), the arithmetic intensity and which
    of the two roofs is the tighter one for this kernel are kept beside it; `peak_at_measured_clock` re-prices the MFMA peak at the
    shader clock measured under MFMA load in this run (256 CUs x 4096 FLOP/clk)."""
    secs = dom["total_ms"] * 1e-3
    tflops = dom["flops"] / secs / 1e12
    gbps = dom["bytes"] / secs / 1e9
    ai = dom["flops"] / dom["bytes"]
    ridge = PEAK_BF16_TFLOPS * 1e3 / PEAK_HBM_GBPS
    common = {"kernel": dom["kernel"], "traffic": traffic, "avg_launch_ms": dom["total_ms"] / dom["launches"], "launches": dom["launches"],
              "flop_per_launch": dom["flops"] / dom["launches"], "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
              "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
              "mfma_tflops": tflops, "mfma_frac": tflops / PEAK_BF16_TFLOPS, "hbm_gbps": gbps, "hbm_frac": gbps / PEAK_HBM_GBPS,
              "instrumented_ms_per_step": instrumented_ms_per_step}
    common["tighter_roof"] = "hbm" if ai < ridge else "mfma"
    if clock_mhz:
        pk = 256 * 4096 * clock_mhz * 1e6 / 1e12
        common.update({"clock_mhz_under_mfma_load": clock_mhz, "peak_at_measured_clock": pk, "frac_at_measured_clock": tflops / pk})
    return {"bound": "mfma", "achieved": tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_BF16_TFLOPS, **common}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=0, help="patches per sr_forward call (Keras predict chunking: result-invariant); 0 = all patches of a call")
    ap.add_argument("--tiles-per-call", type=int, default=16,
                    help="LR tiles whose patches share the generator launches (16 = the whole per-GPU batch: 7056 patches, ~115 GB of workspace)")
    ap.add_argument("--streams", type=int, default=1, help="independent HIP streams (one generator instance each) the tiles are dealt to")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event per-kernel pass (no roofline object)")
    ap.add_argument("--no-attention", action="store_true", help="non-reference graph, kernel tuning only")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="N>1: strong (default) = the 16-tile batch split over the ranks (SURVEY.md 8e); weak = 16 tiles per rank")
    ap.add_argument("--tiles", type=int, default=TILES_PER_GPU, help="tiles this rank processes at N=1 (profiles/: 2/4/8/16 = what a rank sees at N=8/4/2/1)")
    ap.add_argument("--raw-glorot", action="store_true",
                    help="round 1's weights: glorot without conditioning the attention logits (sr355.weights.condition_attention)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--zero-data", action="store_true",
                    help="DIAGNOSTIC, not a result: all-zero weights and tiles -- the same launches with no bit toggling in the operands (MI355X_MICROARCH.md DVFS item 1): what the "
                         "kernels' times become when the chip does not give clock back to data-dependent power; the line says so in `data`")
    ap.add_argument("--no-zero-step", action="store_true",
                    help="skip the in-run DVFS diagnostic (one extra pair of steps on all-zero operands): for the rocprofv3 --stats collection, whose per-kernel averages "
                         "then cover real-data launches only and are directly comparable with roofline.avg_launch_ms")
    ap.add_argument("--no-rows", action="store_true", help="skip the other BASELINE rows (cfg3 training step, cfg4 streaming) that the N = 1 line carries beside the headline")
    ap.add_argument("--fused", type=int, default=511,
                    help="dense-block conv pairs run as one fused kernel: bit 0 conv4+conv5, bit 1 conv2+conv3, bit 2 final_conv2 inside final_conv1, bit 3 attention projections inside the producing conv (0 = layer by layer, for A/B runs)")
    args = ap.parse_args()
    if args.chunk <= 0:
        args.chunk = 441 * max(1, args.tiles_per_call)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={env_world}: the two must agree")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from sr355 import dist as D
    if os.environ.get("SR355_ONE_DEVICE"):                     # rehearsal of the N>1 plumbing on a 1-GPU box: every rank drives cuda:0
        torch.cuda.set_device(0)
    else:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    # RCCL ("nccl") on a real node; gloo only for the one-device rehearsal (RCCL refuses two ranks on one GPU)
    rank, world, local = D.init_from_env(backend=os.environ.get("SR355_DIST_BACKEND", "gloo" if os.environ.get("SR355_ONE_DEVICE") else "nccl"))
    group_world = torch.distributed.get_world_size() if world > 1 else 1      # what the process group itself reports
    if os.environ.get("SR355_ONE_DEVICE"):
        local = 0
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    my_tiles, global_tiles = D.partition_tiles(args.tiles, rank, world, scaling)
    n_mine = len(my_tiles)

    from sr355 import Context
    from sr355.synth import make_pairs
    from sr355.weights import bf16_rounded, condition_attention, init_weights
    from SRModels.deep_learning_models.ESRGAN_model import ESRGAN

    ctx = Context.get(local)
    ctx.set_fused(args.fused, 0)
    models = []
    for _ in range(max(1, args.streams)):
        m_ = ESRGAN(compute_dtype="bf16")
        m_.setup_model(scale_factor=SCALE, growth_channels=G, num_rrdb_blocks=NB, use_attention=not args.no_attention)
        if not models:
            weights = init_weights(m_.generator.layer_shapes(), seed=3000)
            if not args.raw_glorot:
                weights = condition_attention(weights)
            weights = bf16_rounded(weights)      # what the device holds anyway; the oracle then sees bit-identical parameters
            if args.zero_data:
                weights = {n: (np.zeros_like(k), np.zeros_like(b)) for n, (k, b) in weights.items()}
        m_.set_weights(weights)
        models.append(m_)
    model = models[0]
    streams = [torch.cuda.Stream(device=ctx.torch_device) for _ in models] if len(models) > 1 else [None]

    # synthetic 3D-print tiles: 4 distinct tiles per global batch of 16, tile t of the batch = distinct tile t % 4;
    # a rank holds only its own shard of the batch in HBM
    seed = 42 + 2 + (1000 * rank if scaling == "weak" else 0)
    lr4, hr4 = make_pairs(4, LR, LR, SCALE, seed=seed)
    if args.zero_data:
        lr4, hr4 = np.full_like(lr4, 0.5), np.full_like(hr4, 0.5)      # 0.5 -> 0.0 after the generator's [0, 1] -> [-1, 1] rescale
    lr = ctx.to_device(np.stack([lr4[t % 4] for t in my_tiles])) if n_mine else None
    hr = ctx.to_device(np.stack([hr4[t % 4] for t in my_tiles])) if n_mine else None
    sums = torch.zeros(3, dtype=torch.float64, device=ctx.torch_device)

    def step():
        sums.zero_()
        g = max(1, args.tiles_per_call)
        groups = [list(range(t0, min(t0 + g, n_mine))) for t0 in range(0, n_mine, g)]
        done = []
        cur = torch.cuda.current_stream(ctx.torch_device)
        for gi, ts in enumerate(groups):
            mdl, st = models[gi % len(models)], streams[gi % len(models)]
            if st is not None:
                st.wait_stream(cur)
            with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
                if len(ts) == 1:
                    srs = [mdl.super_resolve_image(lr[ts[0]], patch_size_lr=PATCH, stride=STRIDE, batch_size=args.chunk)[0]]
                else:
                    srs, _ = mdl.super_resolve_images([lr[t] for t in ts], patch_size_lr=PATCH, stride=STRIDE, batch_size=args.chunk,
                                                      timed=False)
            done.append((ts, srs, st))
        for ts, srs, st in done:
            if st is not None:
                cur.wait_stream(st)
            for t, sr in zip(ts, srs):
                sums[0] += ctx.psnr(hr[t:t + 1], sr[None])[0].double()
                sums[1] += ctx.ssim(hr[t:t + 1], sr[None])[0].double()
                sums[2] += 1.0
        D.allreduce_metric_sums(sums)      # RCCL over xGMI: the path's only exchange step
        return sums

    def fence():
        D.barrier()
        torch.cuda.synchronize()

    # One un-timed sizing pass: the generator's workspaces are allocated on first use (~115 GB at 16 tiles per call).  Should that
    # not fit (a GPU that is not empty), fall back to fewer tiles per call -- same results, ~2 % less throughput -- rather than die.
    while True:
        try:
            step()
            break
        except (MemoryError, torch.OutOfMemoryError) as e:
            if args.tiles_per_call <= 1:
                raise
            args.tiles_per_call = max(1, args.tiles_per_call // 2)
            args.chunk = 441 * args.tiles_per_call
            print(f"[bench] workspace did not fit ({e}); retrying with {args.tiles_per_call} tiles per call", file=sys.stderr, flush=True)
            torch.cuda.empty_cache()
    for _ in range(args.warmup):
        step()
    fence()
    # per-step device time beside the contract's wall clock: one event between steps on the launching stream (no host sync inside the
    # timed region); SURVEY.md 8(d) asks for the median, the driver's contract for K steps / wall time -- `value` is the latter
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    # the same K steps again with a HIP-event pair around every hot-kernel launch (on the launching stream):
    # per-kernel durations for the roofline object.  `value` comes from the un-instrumented pass above.
    prof, elapsed_prof = [], None
    if not args.no_profile:
        ctx.profile_begin()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed_prof = time.perf_counter() - t1
        prof = ctx.profile_end()
    # DVFS diagnostic inside the run (MI355X_MICROARCH.md "DVFS give-back" item 1; N = 1 only, outside the timed region): the same launches once more on
    # all-zero weights and tiles -- what the dominant kernel's instruction stream sustains when no operand bit toggles and the chip keeps its clock
    zero_prof = None
    res = out.cpu().numpy()                      # the metric sums of the measured steps (the diagnostic below overwrites the device tensor)
    if not args.no_profile and world == 1 and not args.zero_data and not args.no_zero_step and n_mine:
        zw = {n: (np.zeros_like(k), np.zeros_like(b)) for n, (k, b) in weights.items()}
        lr_keep = lr
        for m_ in models:
            m_.set_weights(zw)
        lr = torch.full_like(lr_keep, 0.5)
        step()
        ctx.profile_begin()
        step()
        fence()
        zero_prof = ctx.profile_end()
        lr = lr_keep
        for m_ in models:
            m_.set_weights(weights)
    et = torch.tensor([elapsed], dtype=torch.float64, device=ctx.torch_device)
    D.allreduce_max(et)
    elapsed = float(et.item())

    if rank == 0:
        mpix = global_tiles * (LR * SCALE) ** 2 / 1e6
        clock_mhz = None if args.no_profile else ctx.measure_clock_mhz()
        roof = None
        if prof:
            prof.sort(key=lambda r: -r["total_ms"])
            dom = prof[0]
            traffic, traffic_note = pmc_traffic(dom["kernel"])
            roof = roofline_object(dom, traffic, elapsed_prof / args.steps * 1e3, clock_mhz)
            roof["traffic_source"] = traffic_note
            if zero_prof:
                z = next((r for r in zero_prof if r["kernel"] == dom["kernel"]), None)
                if z:
                    zt = z["flops"] / (z["total_ms"] * 1e-3) / 1e12
                    roof["same_kernel_on_all_zero_operands"] = {
                        "tflops": zt, "frac": zt / PEAK_BF16_TFLOPS, "avg_launch_ms": z["total_ms"] / z["launches"], "ratio_to_real_data": zt / roof["achieved"],
                        "step_ms": sum(r["total_ms"] for r in zero_prof),
                        "note": "one extra step on zero weights and tiles, same launches: the instruction stream's rate at the clock the chip holds when no operand bit toggles; "
                                "the difference to `achieved` is clock the chip gives back under data-dependent power, not issue slots (DESIGN.md 3.8)"}
        line = {
            "metric": "4x-SR MPix/s on 512x512 LR batch", "value": mpix * args.steps / elapsed, "unit": "MPix/s",
            "n_gpus": group_world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median_device": median_ms, "value_at_median_step": mpix / (median_ms * 1e-3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "bf16",
            "data": ("DIAGNOSTIC --zero-data: all-zero weights and tiles (clock / power experiment, not a throughput result); " if args.zero_data else "") +
                    "synthetic (seeded 3D-print tiles; seeded glorot-uniform weights" + ("" if args.raw_glorot else
                    ", attention logits conditioned by 2^-8: sr355.weights.condition_attention") + ")",
            "config": {"workload": f"BASELINE configs[2]: ESRGAN-RRDB x4 (NB=23,G=32,2xSelfAttention) on a batch of {global_tiles} LR tiles 512x512, "
                                   "reference patch mode p=48 s=24 (441 patches/tile)" + (" [NO-ATTENTION tuning variant]" if args.no_attention else ""),
                       "tiles_this_rank": n_mine, "global_batch": global_tiles, "patches_per_forward": min(args.chunk, 441 * max(n_mine, 1)),
                       "tiles_per_call": args.tiles_per_call, "fused_dense_pairs_mask": args.fused,
                       "fused_mask_bits": "1 conv4+conv5, 2 conv2+conv3, 4 final_conv2 inside final_conv1, 8 attention projections inside the producing conv, 16 packed small-image batches (classifier), 32 conv1 of a dense block on the streaming kernel, 64 max-pool inside the conv in front of it (classifier), 128 64-input-channel 3x3 convs on the persistent kernel, 256 SRCNN's 1x1 conv inside the 9x9 head's epilogue",
                       "distinct_tiles": 4, "tile_of_batch_index": "batch tile t is synthetic tile t % 4 (4 distinct 512x512 tiles, each 4 times; nothing is cached between tiles)",
                       "world_size_env": world, "world_size_process_group": group_world, "dist_backend": (torch.distributed.get_backend() if world > 1 else None),
                       "parallelism": f"dp{group_world} (tile shards, metric all-reduce only)"},
            "quality": {"mean_psnr_vs_hr_db": res[0] / res[2], "mean_ssim_vs_hr": res[1] / res[2], "note": "random-init weights"},
            "whole_step": {"conv_tflop": CONV_TFLOP_PER_TILE * global_tiles, "conv_tflops_all_ranks": CONV_TFLOP_PER_TILE * global_tiles / (elapsed / args.steps),
                           "frac_of_bf16_mfma_peak": CONV_TFLOP_PER_TILE * global_tiles / (elapsed / args.steps) / (PEAK_BF16_TFLOPS * world),
                           "note": "SURVEY.md 8(d): conv FLOP the patch-mode batch needs / wall time of the whole step (attention, plumbing and metrics included in the time)"},
            "roofline": roof,
            "kernels": [{"kernel": r["kernel"], "launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                         "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2),
                         "gbps": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)} for r in prof[:8]],
        }
        kept = {}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights, lr4[0], keep=kept)
        if world == 1 and not args.no_parity and not args.no_attention and not args.zero_data:
            line["parity"] = parity_object(ctx, model, weights, lr4[0], hr4[0], kept.get("fp32_reference_graph"))
            line["parity"]["weights"] = "the timed step's: seeded glorot-uniform (PSNR vs HR ~9.6 dB: the generator's output is noise relative to HR)"
            # the same figures where they can fail: a generator whose output resembles HR
            try:
                line["parity_trained_like"], _ = trained_like_parity(ctx, model, lr4, hr4)
            except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
                line["parity_trained_like"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            model.set_weights(weights)
        if world == 1 and not args.no_rows and not args.no_attention and not args.zero_data:
            # BASELINE configs[3] and configs[4] beside the headline, outside its timed region (~25 s): so that the driver's record carries them.
            # A failure here must not cost the headline line.
            from sr355 import bench_rows as BR
            rows = {}
            for m_ in models:
                m_.generator.release_workspace()         # ~115 GB at 16 tiles per call: the rows bring their own
            jobs = (("cfg4_streaming_1080p", lambda: BR.cfg4_streaming(ctx, 3, generator=model)), ("cfg3_train_step", lambda: BR.cfg3_train_step(ctx, 3, 16)),
                    ("cfg0_bicubic_psnr_ssim", lambda: BR.cfg0_bicubic_metrics(ctx)), ("cfg1_srcnn_fp32", lambda: BR.cfg1_srcnn(ctx, 32)),
                    ("a4_edsr_x4", lambda: BR.edsr_x4(ctx)), ("a8_vgg16_classifier", lambda: BR.vgg16_patches(ctx)),
                    ("whole_tile_forward", lambda: BR.whole_tile(ctx, weights, 2)), ("generator_other_shapes", lambda: BR.generator_shapes(ctx)),
                    ("attention_wide_logits", lambda: BR.attention_wide_logits(ctx)))
            for name, fn in jobs:
                t_row = time.perf_counter()
                try:
                    r = fn()
                    r.pop("wall_s", None)
                    rows[name] = r
                except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
                    rows[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                rows[name]["row_wall_s"] = round(time.perf_counter() - t_row, 2)
                if name == "cfg4_streaming_1080p":
                    for m_ in models:
                        m_.generator.release_workspace()     # the trainer wants room
            line["rows"] = rows
        print(json.dumps(line), flush=True)
    if world > 1:
        D.shutdown()


if __name__ == "__main__":
    main()
