"""The C-ABI shared library: loads on a CPU-only box and exports every symbol include/sr355.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sr355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from sr355 import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_everything():
    from sr355 import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_init_without_gpu_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sr355 import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.sr_init(0, ctypes.byref(h)) != 0 and not h.value
    from sr355 import Context
    with pytest.raises(RuntimeError):
        Context.get()


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from sr355 import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libsr355.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_stamp_buffer_sizes_are_exported_and_the_setters_take_a_capacity():
    """the diagnostic stamp setters carry the buffer's capacity; a launch / a buffer that would not fit is refused
    (tests/test_runtime_gpu.py runs the refusal on a context).  The pure size function answers without a GPU."""
    from sr355 import _lib
    lib = _lib.load()
    assert lib.sr_debug_stamp_bytes_needed(0, 63504) == 63504 * 16 * 8          # round 2's fault: 63 504 workgroups x 16 stamps
    assert lib.sr_debug_stamp_bytes_needed(1, 0) == 64 * 4 * 64 * 4 * 8 == 524288
    assert lib.sr_debug_stamp_bytes_needed(0, -1) == -1 and lib.sr_debug_stamp_bytes_needed(7, 1) == -1
    assert len(_lib.SIGNATURES["sr_debug_set_stamp_buffer"][1]) == 3 and len(_lib.SIGNATURES["sr_debug_set_chain_stamp_buffer"][1]) == 3
    # no context: invalid, and nothing is dereferenced
    assert lib.sr_debug_set_stamp_buffer(None, None, 0) == _lib.SR_ERR_INVALID
    assert lib.sr_debug_set_chain_stamp_buffer(None, None, 0) == _lib.SR_ERR_INVALID


def test_fused_mask_default_is_the_same_everywhere():
    """sr_debug_set_fused's default mask (every fused / persistent path on) is stated in the header, compiled into the context, used by the Python
    host's set_fused() and by bench.py --fused: the four must agree, or an A/B run compares something else than it says."""
    import inspect
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "super-resolution-images-for-3d-printing-defect-detection_amd")
    header = open(os.path.join(root, "include", "sr355.h")).read()
    m = re.findall(r"default (\d+)\.\n \* mask 0 = layer by layer", header)
    assert m, "include/sr355.h no longer states the default mask of sr_debug_set_fused"
    default = int(m[0])
    common = open(os.path.join(pkg, "csrc", "common.h")).read()
    assert re.search(r"int chain_mask = %d;" % default, common)
    from sr355 import runtime
    assert inspect.signature(runtime.Context.set_fused).parameters["mask"].default == default
    bench = open(os.path.join(root, "bench.py")).read()
    assert re.search(r'add_argument\("--fused", type=int, default=%d,' % default, bench)
    bits = re.search(r'"fused_mask_bits": "([^"]*)"', bench).group(1)
    named = [int(t) for t in re.findall(r"(?:^|, )(\d+) ", bits)]
    assert named == [1 << i for i in range(len(named))] and sum(named) == default, (named, default)


def test_struct_layouts_of_the_binding_are_the_headers(tmp_path):
    """sr_view and sr_pack_desc cross the ABI by pointer: the ctypes mirrors in sr355/_lib.py must have the C compiler's sizes and field offsets
    (a C program that includes the header prints them)."""
    import subprocess
    from sr355 import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sr355.h"\n'
                   'int main(void) {\n'
                   '  printf("sr_view %zu %zu %zu %zu\\n", sizeof(sr_view), offsetof(sr_view, p), offsetof(sr_view, cs), offsetof(sr_view, coff));\n'
                   '  printf("sr_pack_desc %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sr_pack_desc), offsetof(sr_pack_desc, w), offsetof(sr_pack_desc, bias),\n'
                   '         offsetof(sr_pack_desc, K), offsetof(sr_pack_desc, Cin), offsetof(sr_pack_desc, Cout), offsetof(sr_pack_desc, rot));\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict((l.split()[0], [int(v) for v in l.split()[1:]]) for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    V, P = _lib.View, _lib.PackDesc
    assert out["sr_view"] == [ctypes.sizeof(V), V.p.offset, V.cs.offset, V.coff.offset]
    assert out["sr_pack_desc"] == [ctypes.sizeof(P), P.w.offset, P.bias.offset, P.K.offset, P.Cin.offset, P.Cout.offset, P.rot.offset]
