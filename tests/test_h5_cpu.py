"""the reference's on-disk weight format is Keras `.h5` (`model.save`, SRCNN_model.py:249-260, ESRGAN_model.py:981-995;
`load_model`, ESRGAN_model.py:143-149).  h5py is absent from this image, so the product reads such files with sr355.h5lite.  Pins:
  * tests/golden/srcnn_keras_layout.h5 was written by the genuine HDF5 C library (tests/golden/make_h5_fixture.py) in Keras' layout:
    h5lite must return exactly the seeded weights that went in;
  * where libhdf5 is present (this image has it), the two are checked against each other live, both ways, including a group large enough
    for the library to build a multi-level B-tree (an ESRGAN generator has ~700 layer groups);
  * h5lite's own writer / reader round trip, and its refusals (chunked datasets, libver='latest' files) are explicit."""
import os
import struct

import numpy as np
import pytest

from oracle import models as M
from sr355 import h5lite
from sr355.weights import init_weights
from sr355.wrappers import load_pretrained

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "srcnn_keras_layout.h5")


def _libhdf5():
    try:
        from h5_ctypes import H5
        return H5()
    except (ImportError, OSError, ValueError):
        pytest.skip("no libhdf5 in this image")


def test_reader_parses_the_genuine_library_fixture():
    w = init_weights(M.srcnn_layers(), seed=1000)
    got = load_pretrained(FIXTURE)                      # the path the model classes take for from_pretrained=True
    assert set(got) == set(w) == {"conv2d", "conv2d_1", "conv2d_2"}
    for n, (k, b) in w.items():
        assert got[n][0].dtype == np.float32 and np.array_equal(got[n][0], k) and np.array_equal(got[n][1], b), n
    data, attrs = h5lite.read_h5(FIXTURE)
    assert data["/optimizer_weights/Adam/iter:0"].shape == () and int(data["/optimizer_weights/Adam/iter:0"]) == 1234
    assert list(attrs["/model_weights"]["layer_names"]) == [b"conv2d", b"conv2d_1", b"conv2d_2"]
    assert list(attrs["/model_weights/conv2d_1"]["weight_names"]) == [b"conv2d_1/kernel:0", b"conv2d_1/bias:0"]
    assert bytes(attrs["/"]["backend"]) == b"tensorflow"


def test_writer_reader_round_trip_and_keras_layout(tmp_path):
    rng = np.random.default_rng(0)
    w = {f"rrdb_{i}_dense{d}_conv{c}": (rng.standard_normal((3, 3, 4, 2)).astype(np.float32), rng.standard_normal(2).astype(np.float32))
         for i in range(23) for d in (1, 2, 3) for c in (1, 2, 3, 4, 5)}                      # 345 layer groups: several symbol-table nodes
    w["initial_conv"] = (rng.standard_normal((3, 3, 3, 64)).astype(np.float32), np.zeros(64, np.float32))
    path = str(tmp_path / "g.h5")
    h5lite.save_keras_weights(path, w)
    got = load_pretrained(path)
    assert set(got) == set(w) and all(np.array_equal(got[n][0], w[n][0]) and np.array_equal(got[n][1], w[n][1]) for n in w)
    data, attrs = h5lite.read_h5(path)
    assert "/model_weights/initial_conv/initial_conv/kernel:0" in data
    assert len(attrs["/model_weights"]["layer_names"]) == len(w)
    # a save_weights-style file (layer groups at the root) loads the same way
    h5lite.write_h5(str(tmp_path / "flat.h5"), {"/conv2d/conv2d/kernel:0": w["initial_conv"][0], "/conv2d/conv2d/bias:0": w["initial_conv"][1]}, leaf_k=4)
    flat = load_pretrained(str(tmp_path / "flat.h5"))
    assert np.array_equal(flat["conv2d"][0], w["initial_conv"][0])
    # other dtypes and ranks survive
    h5lite.write_h5(str(tmp_path / "t.h5"), {"/a": np.arange(5, dtype=np.int32), "/g/b": np.float64(2.5), "/g/c": np.zeros((0, 3), np.float32)})
    d, _ = h5lite.read_h5(str(tmp_path / "t.h5"))
    assert d["/a"].dtype == np.int32 and d["/g/b"].shape == () and float(d["/g/b"]) == 2.5 and d["/g/c"].shape == (0, 3)


def test_genuine_library_and_h5lite_agree_both_ways(tmp_path):
    h5 = _libhdf5()
    rng = np.random.default_rng(1)
    ds = {}
    for i in range(700):                                 # more entries than one B-tree node of the library's default fan-out holds
        ds[f"/model_weights/layer_{i:03d}/layer_{i:03d}/kernel:0"] = rng.standard_normal((1, 1, 2, 3)).astype(np.float32)
        ds[f"/model_weights/layer_{i:03d}/layer_{i:03d}/bias:0"] = rng.standard_normal(3).astype(np.float32)
    real = str(tmp_path / "real.h5")
    h5.write(real, ds, {"/model_weights": {"layer_names": [f"layer_{i:03d}".encode() for i in range(700)], "backend": b"tensorflow"}})
    got = h5lite.load_keras_weights(real)
    assert len(got) == 700
    for i in (0, 1, 63, 64, 350, 699):
        n = f"layer_{i:03d}"
        assert np.array_equal(got[n][0], ds[f"/model_weights/{n}/{n}/kernel:0"]) and np.array_equal(got[n][1], ds[f"/model_weights/{n}/{n}/bias:0"])
    _, attrs = h5lite.read_h5(real)
    assert len(attrs["/model_weights"]["layer_names"]) == 700 and attrs["/model_weights"]["layer_names"][699] == b"layer_699"
    # ... and the library reads what h5lite writes
    w = {n: (ds[f"/model_weights/{n}/{n}/kernel:0"], ds[f"/model_weights/{n}/{n}/bias:0"]) for n in (f"layer_{i:03d}" for i in range(700))}
    mine = str(tmp_path / "mine.h5")
    h5lite.save_keras_weights(mine, w)
    for i in (0, 127, 128, 500, 699):
        n = f"layer_{i:03d}"
        assert np.array_equal(h5.read_f32(mine, f"/model_weights/{n}/{n}/kernel:0"), w[n][0])
        assert np.array_equal(h5.read_f32(mine, f"/model_weights/{n}/{n}/bias:0"), w[n][1])
    names = h5.read_str_attr(mine, "/model_weights", "layer_names", 700, 10)
    assert names[0] == b"layer_000" and names[699] == b"layer_699"
    assert h5.read_str_attr(mine, "/model_weights/layer_005", "weight_names", 2, 19)[0] == b"layer_005/kernel:0"


def test_refusals_name_the_feature(tmp_path):
    p = str(tmp_path / "x.h5")
    h5lite.write_h5(p, {"/a": np.zeros(3, np.float32)})
    raw = bytearray(open(p, "rb").read())
    bad = bytes(raw)
    with open(p, "wb") as f:                              # superblock version 2 (libver='latest')
        f.write(bad[:8] + b"\x02" + bad[9:])
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        h5lite.read_h5(p)
    with open(p, "wb") as f:
        f.write(b"not hdf5" * 100)
    with pytest.raises(ValueError, match="not an HDF5 file"):
        h5lite.read_h5(p)
    # a chunked layout message (class 2) in place of the contiguous one
    i = bytes(raw).index(struct.pack("<HHB3x", 8, 24, 0) + bytes([3, 1]))
    raw[i + 9] = 2
    with open(p, "wb") as f:
        f.write(bytes(raw))
    with pytest.raises(NotImplementedError, match="chunked"):
        h5lite.read_h5(p)
    with pytest.raises(FileNotFoundError):
        load_pretrained(str(tmp_path / "missing.h5"))
