"""
A minimal read-only HDF5 parser -- just enough of the file format to open multi-read fast5 files
(ub-bonito/bonito/fast5.py:22-100 reads them through ont_fast5_api / h5py / libhdf5, none of which exist in this image).

Implemented from the HDF5 File Format Specification (version 3.0), the parts fast5 files use:
  * superblock versions 0-3;
  * object headers version 1 and 2 (with continuation blocks);
  * "old style" groups: symbol-table message -> B-tree v1 (node type 0) + local heap + symbol-table nodes, and
    "new style" groups with compact link storage (link messages); dense link storage (fractal heaps) is not supported;
  * attributes (message versions 1-3) of fixed-point, floating-point, fixed-length string and variable-length string
    (global heap) types, scalar or simple dataspaces;
  * datasets with compact, contiguous or chunked (layout version 3: B-tree v1, node type 1) storage, 1-D or N-D,
    fixed-point / floating-point elements;
  * filter pipeline (message versions 1 and 2): deflate (1), shuffle (2), fletcher32 (3) and ONT's VBZ (32020).

VBZ (github.com/nanoporetech/vbz_compression, the HDF5 filter registered as 32020; cd_values = [version, integer size,
delta + zig-zag flag, zstd level]): a chunk is a 4-byte little-endian uncompressed byte count followed by, when the zstd
level is non-zero, one zstd frame; inside it the samples are StreamVByte-coded, and the layout depends on the VERSION:
  * version 0 (what ont_fast5_api writes: VBZ = (32020, (0, 2, 1, 1)), and MinKNOW): samples of 1, 2 or 4 bytes are widened
    to 32 bits, replaced by the zig-zag code of their difference to the previous sample (first against 0) when the flag is
    set, and coded with CLASSIC StreamVByte: ceil(n/4) key bytes, two bits per value (bytes - 1, first value in the low
    bits), then the little-endian data bytes.
  * version 1: 1-byte samples are stored raw; 2-byte samples use "svb16": ceil(n/8) key bytes, one bit per value (least
    significant bit first: 0 = one data byte, 1 = two), differences and zig-zag in 16-bit arithmetic; 4-byte samples use
    the 0/1/2/4-byte StreamVByte variant (key code 0 = no data byte, 3 = four).
Restated from the published sources from memory -- no file written by ONT's own encoder exists in this image, so the
version-0 path is pinned by a hand-computed vector from the StreamVByte format description (tests/test_host.py), not by
a MinKNOW file.  zstd itself comes from the system's libzstd through ctypes.
"""
import ctypes
import ctypes.util
import mmap
import os
import struct
import zlib

import numpy as np

__all__ = ["File", "Group", "Dataset", "Hdf5Error", "vbz_decode", "zstd_decompress"]

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class Hdf5Error(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------------
# zstd (libzstd via ctypes) and VBZ
# ---------------------------------------------------------------------------------------------------------------
_zstd = None


def _libzstd():
    global _zstd
    if _zstd is None:
        try:                                                # the soname first: find_library runs ldconfig / gcc (0.1-1 s)
            lib = ctypes.CDLL("libzstd.so.1")
        except OSError:
            lib = ctypes.CDLL(ctypes.util.find_library("zstd") or "libzstd.so")
        lib.ZSTD_decompress.restype = ctypes.c_size_t
        lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_compress.restype = ctypes.c_size_t
        lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lib.ZSTD_compressBound.restype = ctypes.c_size_t
        lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_isError.restype = ctypes.c_uint
        lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
        lib.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        lib.ZSTD_getFrameContentSize.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        _zstd = lib
    return _zstd


def preload():
    """Resolve the lazily loaded pieces NOW (a reader pool calls this before it forks, so that no worker pays for them)."""
    try:
        _libzstd()
    except OSError:
        pass                                                # no libzstd: only VBZ chunks need it, and they will say so


def zstd_decompress(data, max_out):
    """One zstd frame -> its bytes (a uint8 array view of the output buffer: no copy on the way out)."""
    lib = _libzstd()
    src = bytes(data)
    known = lib.ZSTD_getFrameContentSize(src, len(src))
    cap = int(known) if known < (1 << 62) else int(max_out)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = lib.ZSTD_decompress(out.ctypes.data, cap, src, len(src))
    if lib.ZSTD_isError(n):
        raise Hdf5Error("zstd: corrupt frame")
    return out[:n]


def zstd_compress(data, level=1):
    lib = _libzstd()
    src = bytes(data)
    cap = lib.ZSTD_compressBound(len(src))
    out = ctypes.create_string_buffer(cap)
    n = lib.ZSTD_compress(out, cap, src, len(src), level)
    if lib.ZSTD_isError(n):
        raise Hdf5Error("zstd: compression failed")
    return out.raw[:n]


def _svb16_decode(buf, count):
    """count 16-bit values: key bits (1 per value, LSB first) then 1- or 2-byte little-endian data."""
    nkey = (count + 7) // 8
    if len(buf) < nkey:
        raise Hdf5Error("vbz: truncated svb16 keys")
    two = np.unpackbits(np.frombuffer(buf, dtype=np.uint8, count=nkey), bitorder="little")[:count].view(np.bool_)
    data = np.frombuffer(buf, dtype=np.uint8, offset=nkey)
    # the stream is value after value, low byte first: exactly the row-major order of the (count, 2) byte matrix
    take = np.ones((count, 2), dtype=np.bool_)
    take[:, 1] = two
    total = count + int(np.count_nonzero(two))
    if total > data.size:
        raise Hdf5Error("vbz: truncated svb16 stream")
    out = np.zeros((count, 2), dtype=np.uint8)
    out[take] = data[:total]
    return out.view("<u2").reshape(count)


def _svb32_decode(buf, count, lengths=(1, 2, 3, 4)):
    """StreamVByte: 2 key bits per value, four values per key byte (first value in the low bits), then the data bytes.
    `lengths` maps a key code to its byte count: (1, 2, 3, 4) classic, (0, 1, 2, 4) the 0124 variant."""
    nkey = (count + 3) // 4
    if len(buf) < nkey:
        raise Hdf5Error("vbz: truncated StreamVByte keys")
    kb = np.frombuffer(buf, dtype=np.uint8, count=nkey)
    codes = np.empty((nkey, 4), dtype=np.uint8)
    for i, sh in enumerate((0, 2, 4, 6)):
        codes[:, i] = (kb >> sh) & 3
    nbytes = np.asarray(lengths, dtype=np.uint8)[codes.reshape(-1)[:count]]
    data = np.frombuffer(buf, dtype=np.uint8, offset=nkey)
    # the stream is value after value, low byte first: exactly the row-major order of the (count, 4) byte matrix, so
    # ONE boolean-mask assignment places every data byte
    take = np.arange(4, dtype=np.uint8)[None, :] < nbytes[:, None]
    total = int(nbytes.sum(dtype=np.int64))
    if total > data.size:
        raise Hdf5Error("vbz: truncated StreamVByte data")
    out = np.zeros((count, 4), dtype=np.uint8)
    out[take] = data[:total]
    return out.view("<u4").reshape(count)


def _unzigzag_cumsum(v, dtype):
    """zig-zag codes of differences -> samples, in `dtype`'s modular arithmetic"""
    v = v.astype(dtype, copy=False)
    one = dtype(1)
    return np.cumsum((v >> one) ^ (dtype(0) - (v & one)), dtype=dtype)


def vbz_decode(chunk, cd_values):
    """One VBZ-filtered chunk -> the raw little-endian samples (bytes, or an array of the sample type: a buffer either way)."""
    version, int_size, zigzag, level = (list(cd_values) + [0, 0, 0, 0])[:4]
    if version not in (0, 1):
        raise Hdf5Error("vbz: unsupported version %d" % version)
    if len(chunk) < 4:
        raise Hdf5Error("vbz: short chunk")
    (size,) = struct.unpack_from("<I", chunk, 0)
    body = chunk[4:]
    if level:
        body = zstd_decompress(body, 5 * size // max(int_size, 1) + size + 64)
    if int_size == 0 or (int_size == 1 and version == 1):
        if len(body) < size:
            raise Hdf5Error("vbz: short chunk")
        return bytes(body[:size])
    if int_size not in (1, 2, 4):
        raise Hdf5Error("vbz: unsupported integer size %d" % int_size)
    count = size // int_size
    out_dtype = {1: "u1", 2: "<u2", 4: "<u4"}[int_size]
    if version == 0:
        # widened to 32 bits, classic StreamVByte; differences and zig-zag in 32-bit arithmetic, then cast back
        v = _svb32_decode(body, count)
        if zigzag:
            v = _unzigzag_cumsum(v, np.uint32)
        return v.astype(out_dtype)
    if int_size == 2:
        v = _svb16_decode(body, count)
        if zigzag:
            v = _unzigzag_cumsum(v, np.uint16)
        return v.astype(out_dtype)
    v = _svb32_decode(body, count, lengths=(0, 1, 2, 4))
    if zigzag:
        v = _unzigzag_cumsum(v, np.uint32)
    return v.astype(out_dtype)


def _svb16_encode(values):
    v = np.asarray(values, dtype=np.uint16)
    big = v > 255
    keys = np.packbits(big.astype(np.uint8), bitorder="little").tobytes()
    out = np.zeros(v.size * 2, dtype=np.uint8)
    starts = np.concatenate(([0], np.cumsum(1 + big.astype(np.int64))[:-1])) if v.size else np.zeros(0, np.int64)
    out[starts] = (v & 0xFF).astype(np.uint8)
    out[starts[big] + 1] = (v[big] >> 8).astype(np.uint8)
    total = int(v.size + big.sum())
    return keys + out[:total].tobytes()


def _svb32_encode(values):
    """classic StreamVByte of 32-bit values (the shortest of 1..4 bytes each)"""
    v = np.asarray(values, dtype=np.uint32)
    nbytes = 1 + (v > 0xFF).astype(np.int64) + (v > 0xFFFF) + (v > 0xFFFFFF)
    codes = (nbytes - 1).astype(np.uint8)
    pad = (-v.size) % 4
    c4 = np.concatenate((codes, np.zeros(pad, np.uint8))).reshape(-1, 4)
    keys = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8).tobytes()
    starts = np.concatenate(([0], np.cumsum(nbytes)[:-1])) if v.size else np.zeros(0, np.int64)
    out = np.zeros(int(nbytes.sum()), dtype=np.uint8)
    for b in range(4):
        take = nbytes > b
        out[starts[take] + b] = ((v[take] >> (8 * b)) & 0xFF).astype(np.uint8)
    return keys + out.tobytes()


def vbz_encode_int16(samples, level=1, version=0):
    """The writer side for 16-bit samples with delta + zig-zag (used by the tests to build synthetic fast5 files).
    version 0 = what ont_fast5_api / MinKNOW write (32-bit classic StreamVByte), version 1 = svb16."""
    x = np.asarray(samples, dtype="<i2")
    if version == 0:
        w = x.astype(np.int64)
        d = np.diff(np.concatenate(([0], w)))                                   # fits 32 bits: |d| < 2^17
        z = ((d << 1) ^ (d >> 63)).astype(np.uint32)
        body = _svb32_encode(z)
    elif version == 1:
        u = x.view(np.uint16)
        d = np.diff(np.concatenate(([np.uint16(0)], u)).astype(np.uint16)).astype(np.uint16)      # wraps mod 2^16
        s16 = d.view(np.int16).astype(np.int32)
        z = ((s16 << 1) ^ (s16 >> 15)).astype(np.uint16)
        body = _svb16_encode(z)
    else:
        raise Hdf5Error("vbz: unsupported version %d" % version)
    if level:
        body = zstd_compress(body, level)
    return struct.pack("<I", x.size * 2) + body


# ---------------------------------------------------------------------------------------------------------------
# low-level reader
# ---------------------------------------------------------------------------------------------------------------
_HHB = struct.Struct("<HHB").unpack_from
_ATTR = struct.Struct("<BxHHH").unpack_from
_DTYPE = struct.Struct("<B3sI").unpack_from


class _Buf:
    def __init__(self, data, offset_size=8, length_size=8):
        self.d, self.O, self.L = data, offset_size, length_size

    def u(self, pos, n):
        return int.from_bytes(self.d[pos:pos + n], "little")

    def off(self, pos):
        return self.u(pos, self.O)

    def len_(self, pos):
        return self.u(pos, self.L)


class _Datatype:
    """Decoded datatype message (class, element size, numpy dtype or string / vlen description)."""

    def __init__(self, buf, pos):
        b0, bits, self.size = _DTYPE(buf.d, pos)
        self.cls, self.version = b0 & 0x0F, b0 >> 4
        bits = int.from_bytes(bits, "little")
        self.np = None
        self.vlen_string = False
        self.base = None
        props = pos + 8
        if self.cls == 0:                                   # fixed point
            order = ">" if bits & 1 else "<"
            self.np = np.dtype("%s%s%d" % (order, "i" if bits & 8 else "u", self.size))
            self.end = props + 4
        elif self.cls == 1:                                 # floating point
            order = ">" if bits & 1 else "<"
            self.np = np.dtype("%sf%d" % (order, self.size))
            self.end = props + 12
        elif self.cls == 3:                                 # fixed-length string
            self.pad = bits & 0x0F
            self.end = props
        elif self.cls == 9:                                 # variable length
            self.vlen_string = (bits & 0x0F) == 1
            self.base = _Datatype(buf, props)
            self.end = self.base.end
        elif self.cls == 8:                                 # enumeration (e.g. booleans written by h5py)
            self.base = _Datatype(buf, props)
            self.np = self.base.np
            self.end = self.base.end                        # member names/values follow; not needed for values
        else:
            raise Hdf5Error("datatype class %d not supported" % self.cls)


def _dataspace(buf, pos):
    version = buf.u(pos, 1)
    rank = buf.u(pos + 1, 1)
    if version == 1:
        p = pos + 8
    elif version == 2:
        if buf.u(pos + 3, 1) == 2:                          # null dataspace
            return None
        p = pos + 4
    else:
        raise Hdf5Error("dataspace version %d" % version)
    return tuple(buf.len_(p + i * buf.L) for i in range(rank))


class File:
    """
    with File(path) as f:  f["read_<id>/Raw/Signal"][:] ;  f["read_<id>/Raw"].attrs["start_time"] ;  f.keys()
    The whole file is memory-mapped (a plain `mmap`: slices are `bytes`, integers come from `int.from_bytes`, names from
    `find(b"\0")` -- no numpy scalar indexing on the metadata path); nothing is written.  A group's links are enumerated once
    and kept (`Group._links`), and the root group lives as long as the File: looking a read up in an open multi-read file is
    one dict access (`object_at` skips even that when the caller kept the address from an earlier enumeration).
    """

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise Hdf5Error("hdf5_lite is read-only")
        self.path = str(path)
        with open(self.path, "rb") as fh:
            if os.fstat(fh.fileno()).st_size < 8:
                raise Hdf5Error("%s is not an HDF5 file" % self.path)
            self._mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        data = self._mm
        base = 0
        while data[base:base + 8] != SIGNATURE:              # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(data):
                self._mm.close()
                raise Hdf5Error("%s is not an HDF5 file" % self.path)
        ver = int(data[base + 8])
        if ver in (0, 1):
            O, L = int(data[base + 13]), int(data[base + 14])
            self.buf = _Buf(data, O, L)
            p = base + 24 + (4 if ver == 1 else 0)
            self.base_address = self.buf.off(p)
            p += 4 * O                                       # base, free-space, end-of-file, driver-info addresses
            root_header = self.buf.off(p + O)                # symbol table entry: name offset, header address, ...
        elif ver in (2, 3):
            O, L = int(data[base + 9]), int(data[base + 10])
            self.buf = _Buf(data, O, L)
            self.base_address = self.buf.off(base + 12)
            root_header = self.buf.off(base + 12 + 3 * O)
        else:
            raise Hdf5Error("superblock version %d not supported" % ver)
        self._gheap = {}
        self._attr_memo = {}
        self.root = Group(self, root_header + self.base_address, "/")

    # --- group protocol on the root --------------------------------------------------------------------------
    def __getitem__(self, name):
        return self.root[name]

    def __contains__(self, name):
        return name in self.root

    def keys(self):
        return self.root.keys()

    @property
    def attrs(self):
        return self.root.attrs

    def object_at(self, addr, name=""):
        """The group or dataset whose object header sits at absolute address `addr` (as enumerated by `Group.links()`)."""
        return _typed(_Object(self, addr, name))

    def close(self):
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:                              # a caller still holds a view: let the collector unmap it
                pass
        self._mm = None
        self.buf = None
        self.root = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- global heap (variable-length data) --------------------------------------------------------------------
    def _global_heap_object(self, addr, index):
        buf = self.buf
        col = self._gheap.get(addr)
        if col is None:
            p = addr + self.base_address
            if bytes(buf.d[p:p + 4]) != b"GCOL":
                raise Hdf5Error("bad global heap collection")
            size = buf.len_(p + 8)
            col, q, end = {}, p + 8 + buf.L, p + size
            while q + 8 + buf.L <= end:
                idx = buf.u(q, 2)
                osz = buf.len_(q + 8)
                if idx == 0:
                    break
                col[idx] = (q + 8 + buf.L, osz)
                q += 8 + buf.L + ((osz + 7) & ~7)
            self._gheap[addr] = col
        start, size = col[index]
        return bytes(buf.d[start:start + size])


class _Object:
    """An object header, parsed into its messages: [(type, flags, position, size)]."""

    def __init__(self, f, addr, name):
        self.file, self.addr, self.name = f, addr, name
        self.messages = []
        buf = f.buf
        if bytes(buf.d[addr:addr + 4]) == b"OHDR":
            self._parse_v2(addr)
        else:
            self._parse_v1(addr)

    def _parse_v1(self, addr):
        buf = self.file.buf
        if buf.u(addr, 1) != 1:
            raise Hdf5Error("object header version %d at %d" % (buf.u(addr, 1), addr))
        remaining = buf.u(addr + 2, 2)
        blocks = [(addr + 16, buf.u(addr + 8, 4))]
        while blocks and remaining > 0:
            p, size = blocks.pop(0)
            end = p + size
            d = buf.d
            while p + 8 <= end and remaining > 0:
                mtype, msize, mflags = _HHB(d, p)
                body = p + 8
                remaining -= 1
                if mtype == 0x0010:                         # continuation
                    blocks.append((buf.off(body) + self.file.base_address, buf.len_(body + buf.O)))
                elif mtype != 0:
                    self.messages.append((mtype, mflags, body, msize))
                p = body + msize

    def _parse_v2(self, addr):
        buf = self.file.buf
        flags = buf.u(addr + 5, 1)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        chunk0 = buf.u(p, nsz)
        p += nsz
        blocks = [(p, chunk0)]
        track = bool(flags & 0x04)
        while blocks:
            p, size = blocks.pop(0)
            end = p + size
            while p + 4 <= end:
                mtype, msize, mflags = buf.u(p, 1), buf.u(p + 1, 2), buf.u(p + 3, 1)
                body = p + 4 + (2 if track else 0)
                if body + msize > end:
                    break
                if mtype == 0x10:
                    caddr = buf.off(body) + self.file.base_address
                    clen = buf.len_(body + buf.O)
                    blocks.append((caddr + 4, clen - 8))    # skip "OCHK", drop the checksum
                elif mtype != 0:
                    self.messages.append((mtype, mflags, body, msize))
                p = body + msize

    def find(self, mtype):
        return [(pos, size) for t, _, pos, size in self.messages if t == mtype]

    # --- attributes ---------------------------------------------------------------------------------------------
    @property
    def attrs(self):
        if getattr(self, "_attrs", None) is None:
            self._attrs = {}
            if self.find(0x0015) and not self.find(0x000C):
                raise Hdf5Error("%s: densely stored attributes are not supported" % self.name)
            for pos, _ in self.find(0x000C):
                k, v = self._attribute(pos)
                self._attrs[k] = v
        return self._attrs

    def _attribute(self, pos):
        buf = self.file.buf
        ver, nsz, tsz, ssz = _ATTR(buf.d, pos)
        p = pos + 8 + (1 if ver == 3 else 0)
        if ver == 1:
            nsz8, tsz8, ssz8 = (nsz + 7) & ~7, (tsz + 7) & ~7, (ssz + 7) & ~7
        else:
            nsz8, tsz8, ssz8 = nsz, tsz, ssz
        # name, datatype and dataspace of an attribute are the same bytes in every read group of a fast5 file (only the value
        # behind them differs): decode each distinct description once per file
        head = buf.d[pos:p + nsz8 + tsz8 + ssz8]
        memo = self.file._attr_memo
        hit = memo.get(head)
        if hit is None:
            name = buf.d[p:p + nsz].split(b"\0", 1)[0].decode("utf-8", "replace")
            dt = _Datatype(_Buf(head, buf.O, buf.L), p - pos + nsz8)
            shape = _dataspace(_Buf(head, buf.O, buf.L), p - pos + nsz8 + tsz8)
            count = int(np.prod(shape)) if shape else (1 if shape == () else 0)
            hit = memo[head] = (name, dt, shape, count)
            if len(memo) > 4096:
                memo.clear()
        name, dt, shape, count = hit
        return name, self.file_value(dt, p + nsz8 + tsz8 + ssz8, count, shape)

    def file_value(self, dt, p, count, shape):
        buf = self.file.buf
        if dt.np is not None:
            a = np.frombuffer(buf.d[p:p + count * dt.size], dtype=dt.np, count=count)
            if shape == ():
                return a[0].item()
            return a.astype(a.dtype.newbyteorder("=")).reshape(shape)
        if dt.cls == 3:
            vals = [bytes(buf.d[p + i * dt.size:p + (i + 1) * dt.size]).split(b"\0")[0].rstrip(b" ") if dt.pad == 2 else
                    bytes(buf.d[p + i * dt.size:p + (i + 1) * dt.size]).split(b"\0")[0] for i in range(count)]
            vals = [v.decode("utf-8", "replace") for v in vals]
            return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        if dt.cls == 9:
            vals = []
            step = 4 + buf.O + 4
            for i in range(count):
                q = p + i * step
                n, addr, idx = buf.u(q, 4), buf.off(q + 4), buf.u(q + 4 + buf.O, 4)
                raw = self.file._global_heap_object(addr, idx) if n and addr not in (0, UNDEF & ((1 << (8 * buf.O)) - 1)) else b""
                if dt.vlen_string:
                    vals.append(raw[:n].decode("utf-8", "replace"))
                else:
                    vals.append(np.frombuffer(raw, dtype=dt.base.np, count=n))
            return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        raise Hdf5Error("unsupported attribute type")


class Group(_Object):
    def _links(self):
        if getattr(self, "_children", None) is not None:
            return self._children
        buf, f = self.file.buf, self.file
        children = {}
        sym = self.find(0x0011)
        if sym:                                             # old style: B-tree v1 + local heap
            pos, _ = sym[0]
            btree, heap = buf.off(pos) + f.base_address, buf.off(pos + buf.O) + f.base_address
            if bytes(buf.d[heap:heap + 4]) != b"HEAP":
                raise Hdf5Error("bad local heap")
            heap_data = buf.off(heap + 8 + 2 * buf.L) + f.base_address

            d = buf.d

            def name_at(off):
                q = heap_data + off
                return d[q:d.find(b"\0", q)].decode("utf-8", "replace")

            def walk(node):
                sig = bytes(buf.d[node:node + 4])
                if sig == b"TREE":
                    level, used = buf.u(node + 5, 1), buf.u(node + 6, 2)
                    p = node + 8 + 2 * buf.O
                    for i in range(used):
                        child = buf.off(p + buf.L + i * (buf.L + buf.O)) + f.base_address
                        walk(child)
                    _ = level
                elif sig == b"SNOD":
                    n = buf.u(node + 6, 2)
                    p = node + 8
                    for i in range(n):
                        e = p + i * (2 * buf.O + 24)
                        children[name_at(buf.off(e))] = buf.off(e + buf.O) + f.base_address
                else:
                    raise Hdf5Error("bad group B-tree node")
            walk(btree)
        else:                                               # new style, compact: link messages
            info = self.find(0x0002)
            for pos, _ in self.find(0x0006):
                ver, flags = buf.u(pos, 1), buf.u(pos + 1, 1)
                p = pos + 2
                ltype = 0
                if flags & 0x08:
                    ltype = buf.u(p, 1)
                    p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                lsz = 1 << (flags & 3)
                nlen = buf.u(p, lsz)
                p += lsz
                name = bytes(buf.d[p:p + nlen]).decode("utf-8", "replace")
                p += nlen
                if ltype == 0:
                    children[name] = buf.off(p) + f.base_address
                _ = ver
            if not children and info:
                ipos, _ = info[0]
                iflags = buf.u(ipos + 1, 1)
                q = ipos + 2 + (8 if iflags & 1 else 0)
                if buf.off(q) != UNDEF & ((1 << (8 * buf.O)) - 1):
                    raise Hdf5Error("%s: densely stored links (fractal heap) are not supported" % self.name)
        self._children = children
        return children

    def keys(self):
        return list(self._links().keys())

    def links(self):
        """{child name: absolute object-header address} (File.object_at opens one without another lookup)"""
        return self._links()

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        node = self
        for part in [p for p in name.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(name)
            links = node._links()
            if part not in links:
                raise KeyError(name)
            addr = links[part]
            child_name = (node.name.rstrip("/") + "/" + part)
            node = _typed(_Object(self.file, addr, child_name))
        return node


def _typed(obj):
    obj.__class__ = Dataset if (obj.find(0x0008) or obj.find(0x0001) and obj.find(0x0003)) else Group
    return obj


class Dataset(_Object):
    def _meta(self):
        if getattr(self, "_dt", None) is None:
            buf = self.file.buf
            self._dt = _Datatype(buf, self.find(0x0003)[0][0])
            self._shape = _dataspace(buf, self.find(0x0001)[0][0]) or ()
            self._filters = []
            for pos, _ in self.find(0x000B):
                ver, n = buf.u(pos, 1), buf.u(pos + 1, 1)
                p = pos + (8 if ver == 1 else 2)
                for _i in range(n):
                    fid = buf.u(p, 2)
                    if ver == 1 or fid >= 256:
                        nlen = buf.u(p + 2, 2)
                        p += 2
                    else:
                        nlen = 0
                    ncd = buf.u(p + 4, 2)
                    p += 6
                    p += (nlen + 7) & ~7 if ver == 1 else nlen
                    cd = [buf.u(p + 4 * i, 4) for i in range(ncd)]
                    p += 4 * ncd
                    if ver == 1 and ncd % 2:
                        p += 4
                    self._filters.append((fid, cd))
        return self._dt, self._shape, self._filters

    @property
    def shape(self):
        return self._meta()[1]

    @property
    def dtype(self):
        return self._meta()[0].np

    def _unfilter(self, raw, mask, nbytes):
        dt, _, filters = self._meta()
        data = raw
        for i, (fid, cd) in reversed(list(enumerate(filters))):
            if mask & (1 << i):
                continue
            if fid == 1:
                data = zlib.decompress(data)
            elif fid == 2:
                a = np.frombuffer(data, dtype=np.uint8)
                es = cd[0] if cd else dt.size
                n = a.size // es
                data = a[:n * es].reshape(es, n).T.tobytes() + a[n * es:].tobytes()
            elif fid == 3:
                data = data[:-4]
            elif fid == 32020:
                data = vbz_decode(data, cd)
            else:
                raise Hdf5Error("filter %d not supported" % fid)
        return data

    def __getitem__(self, key):
        dt, shape, _ = self._meta()
        if dt.np is None:
            raise Hdf5Error("%s: only numeric datasets are supported" % self.name)
        buf, f = self.file.buf, self.file
        pos, _ = self.find(0x0008)[0]
        ver = buf.u(pos, 1)
        count = int(np.prod(shape)) if shape else 1
        cls = buf.u(pos + 1, 1)
        # version 4 (files written with the 1.10 format) keeps version 3's compact and contiguous forms and replaces the
        # chunk B-tree by new chunk indexes, which are not implemented (no fast5 writer produces them)
        if ver != 3 and not (ver == 4 and cls in (0, 1)):
            raise Hdf5Error("data layout version %d%s not supported" % (ver, " (chunked)" if ver == 4 else ""))
        if cls == 0:
            size = buf.u(pos + 2, 2)
            out = np.frombuffer(bytes(buf.d[pos + 4:pos + 4 + size]), dtype=dt.np, count=count)
        elif cls == 1:
            addr = buf.off(pos + 2)
            if addr == UNDEF & ((1 << (8 * buf.O)) - 1):
                out = np.zeros(count, dtype=dt.np)
            else:
                a = addr + f.base_address
                out = np.frombuffer(bytes(buf.d[a:a + count * dt.size]), dtype=dt.np, count=count)
        elif cls == 2:
            ndim = buf.u(pos + 2, 1)                          # dataset rank + 1 (element size last)
            btree = buf.off(pos + 3)
            cdims = [buf.u(pos + 3 + buf.O + 4 * i, 4) for i in range(ndim)]
            chunk_shape = tuple(cdims[:-1])
            out = np.zeros(shape, dtype=dt.np)
            if btree != UNDEF & ((1 << (8 * buf.O)) - 1):
                nbytes = int(np.prod(chunk_shape)) * dt.size

                def walk(node):
                    if bytes(buf.d[node:node + 4]) != b"TREE" or buf.u(node + 4, 1) != 1:
                        raise Hdf5Error("bad chunk B-tree node")
                    level, used = buf.u(node + 5, 1), buf.u(node + 6, 2)
                    ksz = 8 + 8 * ndim
                    p = node + 8 + 2 * buf.O
                    for i in range(used):
                        k = p + i * (ksz + buf.O)
                        csize, mask = buf.u(k, 4), buf.u(k + 4, 4)
                        offs = [buf.u(k + 8 + 8 * d, 8) for d in range(ndim - 1)]
                        child = buf.off(k + ksz) + f.base_address
                        if level > 0:
                            walk(child)
                            continue
                        raw = self._unfilter(bytes(buf.d[child:child + csize]), mask, nbytes)
                        block = np.frombuffer(raw, dtype=dt.np, count=int(np.prod(chunk_shape))).reshape(chunk_shape)
                        sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk_shape, shape))
                        out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
                walk(btree + f.base_address)
            out = out.reshape(-1)
        else:
            raise Hdf5Error("layout class %d" % cls)
        out = out.astype(out.dtype.newbyteorder("=")).reshape(shape)
        return out[key]
