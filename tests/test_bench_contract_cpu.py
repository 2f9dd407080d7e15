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
