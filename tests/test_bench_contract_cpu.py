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


def test_dense_block_conv_is_hbm_bound(bench):
    # 552 launches, 224 FLOP per byte (mean dense-block conv with 32 output channels)
    dom = {"kernel": "conv_rows<bf16,k3,kg1,nt2>", "launches": 552, "total_ms": 600.0, "flops": 552 * 1.0488e12, "bytes": 552 * 4.682e9}
    r = bench.roofline_object(dom, 4.778e9, 740.0)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.PEAK_HBM_GBPS
    assert r["achieved"] == pytest.approx(552 * 4.682e9 / 0.6 / 1e9)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["arithmetic_intensity_flop_per_byte"] < r["ridge_flop_per_byte"] == pytest.approx(312.5)
    assert r["traffic"] == 4.778e9 and r["avg_launch_ms"] == pytest.approx(600.0 / 552)
    assert r["mfma_tflops"] == pytest.approx(552 * 1.0488e12 / 0.6 / 1e12) and r["mfma_frac"] == pytest.approx(r["mfma_tflops"] / 2500.0)


def test_high_intensity_kernel_is_mfma_bound(bench):
    dom = {"kernel": "conv_rows<bf16,k3,kg1,nt4>", "launches": 146, "total_ms": 560.0, "flops": 146 * 4.0e12, "bytes": 146 * 1.0e10}
    r = bench.roofline_object(dom, None, 740.0)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.PEAK_BF16_TFLOPS
    assert r["achieved"] == pytest.approx(r["mfma_tflops"]) and r["frac"] == pytest.approx(r["mfma_frac"])
    assert r["traffic"] is None
