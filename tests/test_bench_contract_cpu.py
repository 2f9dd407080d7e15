"""bench.py's roofline object (the contract of its JSON line) on made-up profile records: which roof is chosen, and that achieved /
peak / frac / unit are consistent with it.  No GPU: only the pure function is imported."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # defines functions only; main() runs under __main__
    return mod


def test_headline_roof_is_conv_arithmetic_with_the_hbm_view_beside_it(bench):
    # 552 launches, 224 FLOP per byte (mean dense-block conv with 32 output channels): below the ridge, so HBM is the tighter roof --
    # the headline stays SURVEY.md 8(d)'s conv-arithmetic fraction, the HBM view and the verdict on the tighter roof ride along
    dom = {"kernel": "conv_rows<bf16,k3,kg1,nt2>", "launches": 552, "total_ms": 600.0, "flops": 552 * 1.0488e12, "bytes": 552 * 4.682e9}
    r = bench.roofline_object(dom, 4.778e9, 740.0, clock_mhz=2000.0)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.PEAK_BF16_TFLOPS
    assert r["achieved"] == pytest.approx(552 * 1.0488e12 / 0.6 / 1e12)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["tighter_roof"] == "hbm" and r["arithmetic_intensity_flop_per_byte"] < r["ridge_flop_per_byte"] == pytest.approx(312.5)
    assert r["hbm_gbps"] == pytest.approx(552 * 4.682e9 / 0.6 / 1e9) and r["hbm_frac"] == pytest.approx(r["hbm_gbps"] / bench.PEAK_HBM_GBPS)
    assert r["traffic"] == 4.778e9 and r["avg_launch_ms"] == pytest.approx(600.0 / 552)
    assert r["peak_at_measured_clock"] == pytest.approx(256 * 4096 * 2.0e9 / 1e12)
    assert r["frac_at_measured_clock"] == pytest.approx(r["achieved"] / r["peak_at_measured_clock"])


def test_high_intensity_kernel_is_mfma_bound(bench):
    dom = {"kernel": "conv_rows<bf16,k3,kg1,nt4>", "launches": 146, "total_ms": 560.0, "flops": 146 * 4.0e12, "bytes": 146 * 1.0e10}
    r = bench.roofline_object(dom, None, 740.0)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.PEAK_BF16_TFLOPS
    assert r["achieved"] == pytest.approx(r["mfma_tflops"]) and r["frac"] == pytest.approx(r["mfma_frac"])
    assert r["traffic"] is None and r["tighter_roof"] == "mfma" and "peak_at_measured_clock" not in r


# ---- `python bench.py --gpus N` outside torchrun starts its own ranks: the parent touches no GPU

def test_launch_command_is_the_drivers_torchrun_line(bench):
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], 12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3"]


def test_launcher_refuses_when_the_parent_holds_a_gpu_context(bench, monkeypatch):
    monkeypatch.setattr(bench.torch.cuda, "is_initialized", lambda: True)
    started = []
    monkeypatch.setattr(bench.subprocess, "Popen", lambda *a, **k: started.append(a))
    with pytest.raises(RuntimeError, match="GPU context"):
        bench.launch_ranks(2, ["--gpus", "2"])
    assert not started


def test_launcher_refuses_more_ranks_than_gpus_unless_rehearsing(bench, monkeypatch):
    monkeypatch.setattr(bench.torch.cuda, "is_initialized", lambda: False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("SR355_ONE_DEVICE", raising=False)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.launch_ranks(2, ["--gpus", "2"])


def test_main_self_launches_for_gpus_gt_1_without_torchrun_env_and_relays_the_exit_code(bench, monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    seen = {}

    def fake_launch(n, argv):
        seen["n"], seen["argv"] = n, list(argv)
        return 7
    monkeypatch.setattr(bench, "launch_ranks", fake_launch)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("the parent must not touch the GPU")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and seen == {"n": 2, "argv": ["--gpus", "2", "--steps", "1"]}


def test_main_rejects_a_world_size_that_disagrees_with_gpus(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit, match="must agree"):
        bench.main()


def test_launcher_relays_rank0_stdout_and_exit_code(bench, monkeypatch, capsys):
    """End to end on the CPU: the launcher's Popen/relay loop with a stand-in child command."""
    monkeypatch.setattr(bench.torch.cuda, "is_initialized", lambda: False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port: [bench.sys.executable, "-c", "import sys; print('{\"n_gpus\": 2}'); sys.exit(3)"])
    rc = bench.launch_ranks(2, ["--gpus", "2"])
    assert rc == 3 and capsys.readouterr().out.strip() == '{"n_gpus": 2}'


def test_parity_patches_cover_the_tile_and_have_hr_counterparts(bench):
    import numpy as np
    assert len(bench.PARITY_IDX) == 16 and len(set(bench.PARITY_IDX)) == 16 and max(bench.PARITY_IDX) < 441
    hr = np.arange(2048 * 2048 * 3, dtype=np.float32).reshape(2048, 2048, 3)
    hp = bench.hr_patches(hr, bench.PARITY_IDX)
    assert hp.shape == (16, 192, 192, 3)
    r, c = divmod(bench.PARITY_IDX[5], 21)
    assert hp[5, 0, 0, 0] == hr[r * 96, c * 96, 0] and r * 96 + 192 <= 2048 and c * 96 + 192 <= 2048


def test_launcher_retries_once_when_the_children_die_before_printing(bench, monkeypatch, capsys):
    """the port from free_port() can be taken between closing the probe socket and torchrun binding it; a launch that dies
    without rank 0's line is repeated once on a fresh port, one that printed is not."""
    monkeypatch.setattr(bench.torch.cuda, "is_initialized", lambda: False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    calls = []

    def cmd(n, argv, port):
        calls.append(port)
        if len(calls) == 1:
            return [bench.sys.executable, "-c", "import sys; sys.exit(1)"]
        return [bench.sys.executable, "-c", "print('{\"n_gpus\": 2}')"]
    monkeypatch.setattr(bench, "launch_command", cmd)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    assert len(calls) == 2 and '"n_gpus": 2' in capsys.readouterr().out


def test_steps_below_one_is_refused(bench, monkeypatch):
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--steps", "0"])
    with pytest.raises(SystemExit):
        bench.main()


def test_pmc_traffic_is_quoted_only_for_the_sources_it_was_collected_on(bench, tmp_path):
    """roofline.traffic comes from a committed file; a kernel change without a re-collection must not leave a stale figure."""
    from sr355._lib import source_fingerprint
    fp = source_fingerprint()
    assert len(fp) == 64 and fp == source_fingerprint()
    p = tmp_path / "pmc_traffic.json"
    p.write_text(bench.json.dumps({"dense_tail_fused<bf16,conv4+conv5>": 8.3e9, "_source_sha256": fp}))
    v, note = bench.pmc_traffic("dense_tail_fused<bf16,conv4+conv5>", str(p))
    assert v == 8.3e9 and fp[:12] in note
    v, note = bench.pmc_traffic("some_other_kernel", str(p))
    assert v is None and "no record" in note
    p.write_text(bench.json.dumps({"dense_tail_fused<bf16,conv4+conv5>": 8.3e9, "_source_sha256": "0" * 64}))
    v, note = bench.pmc_traffic("dense_tail_fused<bf16,conv4+conv5>", str(p))
    assert v is None and "other kernel sources" in note
    p.write_text(bench.json.dumps({"dense_tail_fused<bf16,conv4+conv5>": 8.3e9}))          # round 3's file: no fingerprint at all
    assert bench.pmc_traffic("dense_tail_fused<bf16,conv4+conv5>", str(p))[0] is None
    assert bench.pmc_traffic("x", str(tmp_path / "missing.json")) == (None, "no profiles/pmc_traffic.json")
