"""Test infrastructure: the genuine HDF5 C library (libhdf5.so, where the image has one) through ctypes, to pin sr355.h5lite against it in
both directions -- it writes the committed fixture tests/golden/srcnn_keras_layout.h5 (make_h5_fixture.py) that the NumPy reader must
parse, and it reads back what the NumPy writer emits.  h5py itself is not installed here; this is the library h5py wraps."""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np


def find_lib():
    cands = [p for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*")
             for p in sorted(glob.glob(pat)) if "_hl" not in p and "fortran" not in p and "cpp" not in p]
    name = ctypes.util.find_library("hdf5")
    if name:
        cands.append(name)
    for c in cands:
        try:
            return C.CDLL(c)
        except OSError:
            continue
    return None


class H5:
    def __init__(self):
        self.l = find_lib()
        if self.l is None:
            raise ImportError("no libhdf5 in this image")
        L = self.l
        hid = C.c_int64
        L.H5open.restype = C.c_int
        L.H5open()
        for name, res, args in (("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]), ("H5Fclose", C.c_int, [hid]),
                                ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]), ("H5Gclose", C.c_int, [hid]),
                                ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), ("H5Screate", hid, [C.c_int]), ("H5Sclose", C.c_int, [hid]),
                                ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ("H5Dopen2", hid, [hid, C.c_char_p, hid]),
                                ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                                ("H5Dclose", C.c_int, [hid]), ("H5Dget_space", hid, [hid]), ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
                                ("H5Sget_simple_extent_dims", C.c_int, [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                                ("H5Tcopy", hid, [hid]), ("H5Tset_size", C.c_int, [hid, C.c_size_t]), ("H5Tclose", C.c_int, [hid]),
                                ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]), ("H5Aclose", C.c_int, [hid]),
                                ("H5Aopen", hid, [hid, C.c_char_p, hid]), ("H5Aread", C.c_int, [hid, hid, C.c_void_p]), ("H5Aget_space", hid, [hid]),
                                ("H5Gopen2", hid, [hid, C.c_char_p, hid])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        g = lambda n: hid.in_dll(L, n).value
        self.F32, self.I64, self.C_S1 = g("H5T_NATIVE_FLOAT_g"), g("H5T_NATIVE_LLONG_g"), g("H5T_C_S1_g")

    def _ok(self, v, what):
        if v < 0:
            raise RuntimeError(f"libhdf5: {what} failed")
        return v

    def _space(self, shape):
        if len(shape) == 0:
            return self._ok(self.l.H5Screate(0), "H5Screate")          # H5S_SCALAR
        dims = (C.c_uint64 * len(shape))(*shape)
        return self._ok(self.l.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")

    def _str_attr(self, obj, name, values):
        """fixed-length byte strings (a numpy 'S' array), as h5py stores Keras' layer_names / weight_names / backend"""
        a = np.asarray(values, dtype="S")
        t = self._ok(self.l.H5Tcopy(self.C_S1), "H5Tcopy")
        self._ok(self.l.H5Tset_size(t, max(a.dtype.itemsize, 1)), "H5Tset_size")
        s = self._space(a.shape)
        at = self._ok(self.l.H5Acreate2(obj, name.encode(), t, s, 0, 0), "H5Acreate2")
        buf = np.asarray(a, order="C")
        self._ok(self.l.H5Awrite(at, t, buf.ctypes.data_as(C.c_void_p)), "H5Awrite")
        self.l.H5Aclose(at); self.l.H5Sclose(s); self.l.H5Tclose(t)

    def write(self, path, datasets, str_attrs=None):
        """datasets: {'/a/b/name': float32 or int64 ndarray}; str_attrs: {'/a' or '/': {name: bytes or list of bytes}}.  Library defaults
        throughout (what h5py passes on): earliest file format, contiguous layout."""
        str_attrs = str_attrs or {}
        f = self._ok(self.l.H5Fcreate(path.encode(), 2, 0, 0), "H5Fcreate")          # H5F_ACC_TRUNC
        groups = {"": f}
        try:
            def group(p):
                if p not in groups:
                    parent, _, _name = p.rpartition("/")
                    group(parent)
                    groups[p] = self._ok(self.l.H5Gcreate2(f, p.encode(), 0, 0, 0), "H5Gcreate2 " + p)
                return groups[p]
            for k in sorted(datasets):
                v = np.asarray(datasets[k], order="C")
                group(k.rpartition("/")[0])
                t = self.F32 if v.dtype == np.float32 else self.I64
                s = self._space(v.shape)
                d = self._ok(self.l.H5Dcreate2(f, k.encode(), t, s, 0, 0, 0), "H5Dcreate2 " + k)
                self._ok(self.l.H5Dwrite(d, t, 0, 0, 0, v.ctypes.data_as(C.c_void_p)), "H5Dwrite")
                self.l.H5Dclose(d); self.l.H5Sclose(s)
            for gpath, attrs in str_attrs.items():
                obj = group(gpath.rstrip("/"))
                for name, val in attrs.items():
                    self._str_attr(obj, name, val)
        finally:
            for p, g in groups.items():
                if p:
                    self.l.H5Gclose(g)
            self.l.H5Fclose(f)

    def read_f32(self, path, name):
        f = self._ok(self.l.H5Fopen(path.encode(), 0, 0), "H5Fopen")
        try:
            d = self._ok(self.l.H5Dopen2(f, name.encode(), 0), "H5Dopen2 " + name)
            s = self.l.H5Dget_space(d)
            n = self.l.H5Sget_simple_extent_ndims(s)
            dims = (C.c_uint64 * max(n, 1))()
            self.l.H5Sget_simple_extent_dims(s, dims, None)
            out = np.empty(tuple(dims[i] for i in range(n)), np.float32)
            self._ok(self.l.H5Dread(d, self.F32, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
            self.l.H5Sclose(s); self.l.H5Dclose(d)
            return out
        finally:
            self.l.H5Fclose(f)

    def read_str_attr(self, path, obj, name, count, size):
        f = self._ok(self.l.H5Fopen(path.encode(), 0, 0), "H5Fopen")
        try:
            g = self._ok(self.l.H5Gopen2(f, obj.encode(), 0), "H5Gopen2")
            a = self._ok(self.l.H5Aopen(g, name.encode(), 0), "H5Aopen")
            t = self.l.H5Tcopy(self.C_S1)
            self.l.H5Tset_size(t, size)
            out = np.zeros(count, dtype=f"S{size}")
            self._ok(self.l.H5Aread(a, t, out.ctypes.data_as(C.c_void_p)), "H5Aread")
            self.l.H5Tclose(t); self.l.H5Aclose(a); self.l.H5Gclose(g)
            return out
        finally:
            self.l.H5Fclose(f)
