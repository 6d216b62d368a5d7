"""Test infrastructure: write HDF5 files through the REAL libhdf5 (ctypes), so that hdf5_lite meets files its author did
not lay out byte by byte.  Only present in build containers that ship libhdf5 (here:
/opt/conda/lib/libhdf5.so.103, HDF5 1.10.6); the tests skip without it, and nothing here travels into the product."""
import ctypes as C
import glob
import os

import numpy as np

_LIB = None
H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL, H5T_VARIABLE = 2, 0, 0, C.c_size_t(-1).value
H5F_LIBVER_EARLIEST, H5F_LIBVER_LATEST = 0, 2          # 1.10: EARLIEST 0, V18 1, V110 = LATEST 2


def find():
    for pat in (os.environ.get("XB_LIBHDF5", ""), "/opt/conda/lib/libhdf5.so.10*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so.10*",
                "/usr/lib/x86_64-linux-gnu/libhdf5.so.10*"):
        for path in sorted(glob.glob(pat)) if pat else []:
            if os.path.isfile(path):
                return path
    return None


def lib():
    global _LIB
    if _LIB is None:
        path = find()
        if path is None:
            raise ImportError("no libhdf5 in this container")
        L = C.CDLL(path)
        hid = C.c_int64
        for name, res, args in [
                ("H5open", C.c_int, []), ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fclose", C.c_int, [hid]),
                ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]), ("H5Gclose", C.c_int, [hid]),
                ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]), ("H5Screate", hid, [C.c_int]),
                ("H5Sclose", C.c_int, [hid]), ("H5Pcreate", hid, [hid]), ("H5Pclose", C.c_int, [hid]),
                ("H5Pset_chunk", C.c_int, [hid, C.c_int, C.POINTER(C.c_uint64)]), ("H5Pset_shuffle", C.c_int, [hid]),
                ("H5Pset_deflate", C.c_int, [hid, C.c_uint]), ("H5Pset_fletcher32", C.c_int, [hid]),
                ("H5Pset_libver_bounds", C.c_int, [hid, C.c_int, C.c_int]),
                ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dclose", C.c_int, [hid]),
                ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]),
                ("H5Aclose", C.c_int, [hid]), ("H5Tcopy", hid, [hid]), ("H5Tset_size", C.c_int, [hid, C.c_size_t]),
                ("H5Tclose", C.c_int, [hid]), ("H5get_libversion", C.c_int, [C.POINTER(C.c_uint)] * 3)]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.H5open()
        _LIB = L
    return _LIB


def _g(name):
    return C.c_int64.in_dll(lib(), name).value


def _native(dtype):
    return _g({"int8": "H5T_NATIVE_SCHAR_g", "uint8": "H5T_NATIVE_UCHAR_g", "int16": "H5T_NATIVE_SHORT_g",
               "uint16": "H5T_NATIVE_USHORT_g", "int32": "H5T_NATIVE_INT_g", "uint32": "H5T_NATIVE_UINT_g",
               "int64": "H5T_NATIVE_LONG_g", "uint64": "H5T_NATIVE_ULONG_g", "float32": "H5T_NATIVE_FLOAT_g",
               "float64": "H5T_NATIVE_DOUBLE_g"}[np.dtype(dtype).name])


class Writer:
    """A small h5py look-alike over libhdf5: w = Writer(path[, latest=True]); g = w.group(parent, name); w.attr(obj, name,
    value) (numpy scalars / arrays, str -> fixed-length, ('vlen', str) -> variable-length string);
    w.dataset(parent, name, array, chunks=None, shuffle=False, deflate=0, fletcher32=False); w.close()."""

    def __init__(self, path, latest=False):
        L = lib()
        fapl = L.H5Pcreate(_g("H5P_CLS_FILE_ACCESS_ID_g"))
        if latest:
            L.H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST)
        self.file = L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl)
        L.H5Pclose(fapl)
        if self.file < 0:
            raise OSError("H5Fcreate failed for %s" % path)
        self.open = []

    def group(self, parent, name):
        g = lib().H5Gcreate2(parent, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        assert g >= 0
        self.open.append(("g", g))
        return g

    def attr(self, obj, name, value):
        L = lib()
        if isinstance(value, tuple) and value[0] == "vlen":
            t = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(t, H5T_VARIABLE)
            sp = L.H5Screate(0)                                           # scalar
            a = L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT)
            ptr = C.c_char_p(value[1].encode())
            assert L.H5Awrite(a, t, C.byref(ptr)) >= 0
        elif isinstance(value, str):
            raw = value.encode()
            t = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(t, max(len(raw), 1))
            sp = L.H5Screate(0)
            a = L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT)
            buf = C.create_string_buffer(raw, max(len(raw), 1))
            assert L.H5Awrite(a, t, buf) >= 0
        else:
            v = np.require(np.asarray(value), requirements="C")          # (ascontiguousarray would turn a scalar into shape (1,))
            t = L.H5Tcopy(_native(v.dtype))
            if v.ndim == 0:
                sp = L.H5Screate(0)
            else:
                dims = (C.c_uint64 * v.ndim)(*v.shape)
                sp = L.H5Screate_simple(v.ndim, dims, None)
            a = L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT)
            assert L.H5Awrite(a, t, v.ctypes.data_as(C.c_void_p)) >= 0
        L.H5Aclose(a)
        L.H5Sclose(sp)
        L.H5Tclose(t)

    def dataset(self, parent, name, array, chunks=None, shuffle=False, deflate=0, fletcher32=False):
        L = lib()
        v = np.ascontiguousarray(array)
        dims = (C.c_uint64 * v.ndim)(*v.shape)
        sp = L.H5Screate_simple(v.ndim, dims, None)
        dcpl = L.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
        if chunks is not None:
            ch = (C.c_uint64 * v.ndim)(*([chunks] if np.isscalar(chunks) else chunks))
            assert L.H5Pset_chunk(dcpl, v.ndim, ch) >= 0
            if shuffle:
                assert L.H5Pset_shuffle(dcpl) >= 0
            if deflate:
                assert L.H5Pset_deflate(dcpl, deflate) >= 0
            if fletcher32:
                assert L.H5Pset_fletcher32(dcpl) >= 0
        t = _native(v.dtype)
        d = L.H5Dcreate2(parent, name.encode(), t, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT)
        assert d >= 0
        assert L.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.ctypes.data_as(C.c_void_p)) >= 0
        L.H5Pclose(dcpl)
        L.H5Sclose(sp)
        self.open.append(("d", d))
        return d

    def close(self):
        L = lib()
        for kind, h in reversed(self.open):
            (L.H5Gclose if kind == "g" else L.H5Dclose)(h)
        self.open = []
        L.H5Fclose(self.file)


def write_multi_fast5(path, reads, vlen_strings=True, chunk=4096, latest=False, fillers=0):
    """The ont_fast5_api multi-read layout written by libhdf5: /read_<id>/{Raw{Signal, attrs}, channel_id{attrs},
    tracking_id{attrs}}, signal chunked + shuffle + deflate (VBZ needs ONT's plugin, which no image has).  `fillers` adds
    that many extra (empty) groups at the top level so that the root group's B-tree has several levels."""
    w = Writer(path, latest=latest)
    s = (lambda v: ("vlen", v)) if vlen_strings else (lambda v: v)
    w.attr(w.file, "file_type", s("multi-read"))
    w.attr(w.file, "file_version", s("2.2"))
    for raw, a in reads:
        raw = np.asarray(raw, dtype=np.int16)
        rd = w.group(w.file, "read_" + a["read_id"])
        w.attr(rd, "run_id", s(a.get("run_id", "")))
        rg = w.group(rd, "Raw")
        w.dataset(rg, "Signal", raw, chunks=min(chunk, max(len(raw), 1)), shuffle=True, deflate=4)
        w.attr(rg, "read_id", s(a["read_id"]))
        w.attr(rg, "start_mux", np.uint8(a.get("start_mux", 1)))
        w.attr(rg, "read_number", np.int32(a.get("read_number", 0)))
        w.attr(rg, "start_time", np.uint64(a.get("start_time", 0)))
        w.attr(rg, "duration", np.uint32(a.get("duration", len(raw))))
        w.attr(rg, "median_before", np.float64(200.0))
        ch = w.group(rd, "channel_id")
        w.attr(ch, "channel_number", s(str(a.get("channel_number", "1"))))
        for k in ("digitisation", "offset", "range", "sampling_rate"):
            w.attr(ch, k, np.float64(a[k]))
        tr = w.group(rd, "tracking_id")
        w.attr(tr, "run_id", s(a.get("run_id", "")))
        w.attr(tr, "sample_id", s(a.get("sample_id", "sample")))
        w.attr(tr, "exp_start_time", s(a.get("exp_start_time", "1970-01-01T00:00:00Z")))
        w.attr(tr, "flow_cell_id", s(a.get("flow_cell_id", "FAK00000")))
        w.attr(tr, "device_id", s(a.get("device_id", "MN00000")))
    for i in range(fillers):
        w.group(w.file, "zz_filler_%03d" % i)
    w.close()
