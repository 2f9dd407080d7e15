"""The N > 1 paths with their PRODUCT code on a one-GPU box: ranks as children of torch.distributed.run, every rank on cuda:0,
gloo between them (RCCL refuses two ranks on one device) -- the plumbing, partition and hooks are the ones a node runs over RCCL; only the transport differs.

This file sorts first on purpose: the ranks must be started from a process that has made no GPU call (a process holding a GPU context must not be
replaced or forked into ranks on this pool -- bench.launch_ranks refuses as well), and the session's Context fixture has not been created yet."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import bench as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.update(SR355_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _need_clean_parent():
    if torch.cuda.device_count() < 1:              # counting devices does not initialise the GPU on this image; torch.cuda.is_available() would
        pytest.skip("no GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process already holds a GPU context: the rank launch must come from a GPU-free parent (run this file first / on its own)")


def test_product_trainer_two_ranks_stay_bit_identical_and_match_one_rank(tmp_path):
    """ESRGAN.enable_data_parallel() + _train_step on two ranks, half a batch each: replicas bit-identical after two steps (generator bucket, Adam moments,
    discriminator kernels, spectral-norm vectors), and equal to the one-process run on the whole batch up to the fp32 rounding of the averaged bucket."""
    _need_clean_parent()
    script = os.path.join(ROOT, "tools", "dp_rehearsal.py")
    outs = {}
    for world in (1, 2):
        env = _env()
        env["DP_REHEARSAL_OUT"] = str(tmp_path / f"w{world}.npy")
        cmd = [sys.executable, script] if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                                                           "--master-port", str(B.free_port()), script]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        outs[world] = (_last_json(r.stdout), np.load(tmp_path / f"w{world}.npy"))
    two, one = outs[2], outs[1]
    assert two[0]["world"] == 2 and two[0]["replicas_bit_identical"] is True
    assert all(np.isfinite(v) for l in two[0]["losses_rank0"] for v in l.values())
    a, b = two[1], one[1]
    assert a.shape == b.shape and np.isfinite(a).all()
    rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print(f"\n2-rank vs 1-rank trainer state after {2} steps: rel-L2 {rel:.2e} over {a.size} values")
    assert rel <= 2e-3, rel                    # Adam's first steps are ~lr * sign(g): a rounding-level change of a tiny gradient can flip one
    assert float(np.abs(a - b).max()) <= 2.1e-4          # ... by at most two learning rates (1e-4) per element
    assert two[0]["sha256"] != one[0]["sha256"] or rel == 0.0


def test_bench_two_ranks_strong_scaling_equals_the_one_rank_metric_sums():
    """`bench.py --gpus 2 --scaling strong` (self-launched ranks, tile partition of sr355.dist, metric all-reduce) against `--gpus 1`: the same 4 tiles, the
    same PSNR / SSIM means -- every tile is computed by exactly one rank and the sums are exact in fp64."""
    _need_clean_parent()
    common = ["--steps", "1", "--warmup", "0", "--tiles", "4", "--tiles-per-call", "2", "--no-cpu-baseline", "--no-parity", "--no-rows", "--no-profile"]
    lines = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + (["--scaling", "strong"] if n > 1 else []) + common,
                           env=_env(), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        lines[n] = _last_json(r.stdout)
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["global_batch"] == 4 and two["config"]["tiles_this_rank"] == 2
    assert two["config"]["dist_backend"] == "gloo" and two["config"]["parallelism"].startswith("dp2")
    assert one["n_gpus"] == 1 and one["config"]["global_batch"] == 4
    for k in ("mean_psnr_vs_hr_db", "mean_ssim_vs_hr"):
        assert abs(one["quality"][k] - two["quality"][k]) <= 1e-9 * max(1.0, abs(one["quality"][k])), (k, one["quality"][k], two["quality"][k])
    assert two["value"] > 0 and two["higher_is_better"] is True
