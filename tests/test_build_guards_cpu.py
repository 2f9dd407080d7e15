"""Build-time guards that need no GPU: the assembly-level proof that the inline-asm prefetch loads of the 3x3 kernels are never
touched before their explicit wait and that those kernels do not spill (tools/check_prefetch_hazards.py), and a
self-test of the checker on a deliberately broken instruction stream."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
needs_hipcc = pytest.mark.skipif(not (os.path.isfile(HIPCC) or shutil.which("hipcc")), reason="no hipcc on this box (build() keeps the check hard)")


def _checker():
    spec = importlib.util.spec_from_file_location("check_prefetch_hazards", os.path.join(ROOT, "tools", "check_prefetch_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@needs_hipcc
def test_conv_rows_assembly_has_no_prefetch_hazard_and_no_spill():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_hazards.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 problems" in r.stdout


@needs_hipcc
def test_fused_dense_kernels_use_no_scratch():
    """csrc/dense_fused.hip: the loader waves' counted vmcnt waits and the 168-register budget of three waves per SIMD both need a
    spill-free build of every chain2_kernel instance."""
    src = os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd", "csrc", "dense_fused.hip")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_hazards.py"), "--source", src, "--match", "chain2_kernel",
                        "--scratch-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "15 kernels checked" in r.stdout and " 0 problems" in r.stdout          # 9 + the six two-up (SEAM) instantiations of round 4
    # the streaming conv1 kernel (same roles, same counted waits)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_hazards.py"), "--source", src, "--match", "conv1_stream_kernel",
                        "--scratch-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "2 kernels checked" in r.stdout and " 0 problems" in r.stdout
    # the persistent 64-input-channel kernel of conv_stream.hip (counted waits in its loaders as well)
    src2 = os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd", "csrc", "conv_stream.hip")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_hazards.py"), "--source", src2, "--match", "conv64_stream_kernel",
                        "--scratch-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "1 kernels checked" in r.stdout and " 0 problems" in r.stdout


def test_checker_flags_a_register_touched_under_an_outstanding_load():
    C = _checker()
    bad = """
_Zkernel_conv3_rows_kernel_bad:
	global_load_dwordx4 v[4:7], v[0:1], off
	global_load_dwordx4 v[8:11], v[0:1], off
	s_waitcnt vmcnt(1)
	v_mov_b32_e32 v12, v4
	v_mov_b32_e32 v13, v9
	s_waitcnt vmcnt(0)
	v_mov_b32_e32 v14, v9
	s_endpgm
.Lfunc_end0:
"""
    k = C.parse_kernels(bad)
    probs = C.check_kernel("bad", k["_Zkernel_conv3_rows_kernel_bad"])
    assert len(probs) == 1 and "v13, v9" in probs[0]          # v4 landed (vmcnt(1)), v9 had not; after vmcnt(0) it is fine
    loop = """
_Zkernel_conv3_rows_kernel_loop:
.LBB0_1:
	s_waitcnt vmcnt(0)
	ds_write_b128 v20, v[4:7]
	global_load_dwordx4 v[4:7], v[0:1], off
	s_cbranch_scc1 .LBB0_1
	v_mov_b32_e32 v12, v4
	s_endpgm
.Lfunc_end0:
"""
    k = C.parse_kernels(loop)
    probs = C.check_kernel("loop", k["_Zkernel_conv3_rows_kernel_loop"])
    assert len(probs) == 1 and "v12, v4" in probs[0]          # the loop body is clean (wait first); the exit path is not


def test_checker_allows_overlapping_destinations_only_among_in_order_loads():
    """a load into the destination of an in-flight load is harmless only while both are global_ / buffer_ loads (one in-order
    return queue, the one vmcnt counts); with a flat_ load on either side the overlap is reported."""
    C = _checker()

    def probs(body):
        src = "\n_Zkernel_conv3_rows_kernel_t:\n" + body + "\ts_waitcnt vmcnt(0)\n\ts_endpgm\n.Lfunc_end0:\n"
        return C.check_kernel("t", C.parse_kernels(src)["_Zkernel_conv3_rows_kernel_t"])

    assert probs("\tglobal_load_dwordx4 v[4:7], v[0:1], off\n\tglobal_load_dwordx4 v[4:7], v[2:3], off\n") == []
    assert probs("\tglobal_load_dwordx4 v[4:7], v[0:1], off\n\tbuffer_load_dwordx4 v[4:7], v2, s[0:3], 0 offen\n") == []
    assert len(probs("\tflat_load_dwordx4 v[4:7], v[0:1]\n\tglobal_load_dwordx4 v[4:7], v[2:3], off\n")) == 1
    assert len(probs("\tglobal_load_dwordx4 v[4:7], v[0:1], off\n\tflat_load_dwordx4 v[4:7], v[2:3]\n")) == 1
