"""
Test helper: writes small HDF5 files in the classic on-disk layout (superblock version 0, version-1 object headers,
symbol-table groups with a B-tree v1 + local heap, chunked datasets indexed by a B-tree v1, filter pipeline messages) --
the layout h5py / libhdf5 produce with default settings, which is what multi-read fast5 files use.  It exists so that the
reader in xna_basecaller_amd/hdf5_lite.py can be exercised without libhdf5 (absent from this image); it follows the
HDF5 File Format Specification field by field, but only libhdf5 itself could certify the files.
"""
import struct

import numpy as np

from xna_basecaller_amd.hdf5_lite import vbz_encode_int16

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


class H5Writer:
    def __init__(self, fanout=0):
        self.buf = bytearray(96)                       # room for the superblock (version 0: 96 bytes with 8-byte sizes)
        self.gheap = []                                # objects of the single global heap collection
        self.fanout = fanout                           # > 0: at most `fanout` entries per B-tree / symbol-table node,
                                                       # i.e. multi-level trees like libhdf5 writes for large objects

    # ---- raw allocation --------------------------------------------------------------------------------------
    def alloc(self, data, align=8):
        self.buf += b"\0" * (-len(self.buf) % align)
        addr = len(self.buf)
        self.buf += data
        return addr

    # ---- datatype / dataspace encodings --------------------------------------------------------------------------
    @staticmethod
    def dtype_msg(dt):
        dt = np.dtype(dt)
        if dt.kind in "iu":
            bits = (8 if dt.kind == "i" else 0)
            return struct.pack("<B3BI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
        if dt.kind == "f":
            if dt.itemsize == 4:
                return struct.pack("<B3BI", 0x10 | 1, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            return struct.pack("<B3BI", 0x10 | 1, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        raise TypeError(dt)

    @staticmethod
    def string_dtype_msg(n):
        return struct.pack("<B3BI", 0x10 | 3, 0x00, 0, 0, n)                 # null-terminated ASCII

    @staticmethod
    def vlen_string_dtype_msg():
        base = struct.pack("<B3BI", 0x10 | 3, 0x00, 0, 0, 1)
        return struct.pack("<B3BI", 0x10 | 9, 0x01, 0, 0, 16) + base          # type 1 = string, padding 0, ASCII

    @staticmethod
    def dataspace_msg(shape):
        if shape is None:                                                     # scalar
            return struct.pack("<BBB5x", 1, 0, 0)
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)

    # ---- messages -------------------------------------------------------------------------------------------------
    def attr_msg(self, name, value):
        nm = name.encode() + b"\0"
        if isinstance(value, str):
            if value.startswith("vlen:"):                                     # stored as a variable-length string
                text = value[5:].encode()
                self.gheap.append(text)
                dt = self.vlen_string_dtype_msg()
                data = struct.pack("<IQI", len(text), 0, len(self.gheap))     # collection address patched at the end
                self._vlen_patches.append(None)
            else:
                raw = value.encode() + b"\0"
                dt = self.string_dtype_msg(len(raw))
                data = raw
            ds = self.dataspace_msg(None)
        else:
            a = np.asarray(value)
            dt = self.dtype_msg(a.dtype)
            ds = self.dataspace_msg(None if a.ndim == 0 else a.shape)
            data = a.astype(a.dtype.newbyteorder("<")).tobytes()
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data
        return (0x000C, body)

    def object_header(self, messages):
        blob = b""
        for mtype, body in messages:
            body = _pad8(body)
            blob += struct.pack("<HHB3x", mtype, len(body), 0) + body
        hdr = struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(blob))
        return self.alloc(hdr + blob)

    # ---- objects --------------------------------------------------------------------------------------------------
    def dataset(self, array, attrs=None, chunks=None, vbz=False):
        a = np.ascontiguousarray(array)
        msgs = [(0x0001, self.dataspace_msg(a.shape)), (0x0003, self.dtype_msg(a.dtype)),
                (0x0005, struct.pack("<BBBB", 2, 2, 2, 0))]
        if chunks is None:
            addr = self.alloc(a.astype(a.dtype.newbyteorder("<")).tobytes())
            msgs.append((0x0008, struct.pack("<BBQQ", 3, 1, addr, a.nbytes)))
        else:
            assert a.ndim == 1
            entries = []
            for o in range(0, a.shape[0], chunks):
                block = np.zeros(chunks, dtype=a.dtype)
                part = a[o:o + chunks]
                block[:part.size] = part
                raw = vbz_encode_int16(block, level=1) if vbz else block.tobytes()
                entries.append((o, len(raw), self.alloc(raw)))
            # B-tree v1 nodes of type 1; keys: chunk size, filter mask, offsets (rank + 1 values); children of a leaf are
            # the chunks, children of an internal node are nodes one level down (key = first chunk below)
            end_key = struct.pack("<IIQQ", 0, 0, a.shape[0] + ((-a.shape[0]) % chunks), 0)
            level, fan = 0, (self.fanout or len(entries) or 1)
            while True:
                nodes = []
                for g0 in range(0, max(len(entries), 1), fan):
                    grp = entries[g0:g0 + fan]
                    node = b"TREE" + struct.pack("<BBHQQ", 1, level, len(grp), UNDEF, UNDEF)
                    for o, size, addr in grp:
                        node += struct.pack("<IIQQ", size, 0, o, 0) + struct.pack("<Q", addr)
                    nodes.append((grp[0][0] if grp else 0, 0, self.alloc(node + end_key)))
                if len(nodes) == 1:
                    bt = nodes[0][2]
                    break
                entries, level = nodes, level + 1
            msgs.append((0x0008, struct.pack("<BBBQII", 3, 2, 2, bt, chunks, a.dtype.itemsize)))
            if vbz:
                name = _pad8(b"vbz\0")
                msgs.append((0x000B, struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 32020, len(name), 1, 4) + name +
                             struct.pack("<IIII", 0, a.dtype.itemsize, 1, 1)))
        for k, v in (attrs or {}).items():
            msgs.append(self.attr_msg(k, v))
        return self.object_header(msgs)

    def group(self, children, attrs=None):
        """children: {name: object header address}"""
        names = sorted(children)
        heap = bytearray(b"\0" * 8)                                   # offset 0: the empty string
        offs = {}
        for n in names:
            offs[n] = len(heap)
            heap += _pad8(n.encode() + b"\0")
        heap_data = self.alloc(bytes(heap))
        heap_hdr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), UNDEF, heap_data))
        # symbol-table nodes of at most `fan` entries under B-tree v1 nodes of type 0 (keys = heap offsets of names)
        fan = self.fanout or max(len(names), 1)
        kids = []                                                        # (last name offset, address)
        for g0 in range(0, max(len(names), 1), fan):
            grp = names[g0:g0 + fan]
            snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(grp))
            for n in grp:
                snod += struct.pack("<QQII16x", offs[n], children[n], 0, 0)
            kids.append((offs[grp[-1]] if grp else 0, self.alloc(snod)))
        level = 0
        while True:
            nodes = []
            for g0 in range(0, len(kids), fan):
                grp = kids[g0:g0 + fan]
                tree = b"TREE" + struct.pack("<BBHQQ", 0, level, len(grp), UNDEF, UNDEF) + struct.pack("<Q", 0)
                for last, addr in grp:
                    tree += struct.pack("<QQ", addr, last)
                nodes.append((grp[-1][0], self.alloc(tree)))
            if len(nodes) == 1:
                bt = nodes[0][1]
                break
            kids, level = nodes, level + 1
        msgs = [(0x0011, struct.pack("<QQ", bt, heap_hdr))]
        for k, v in (attrs or {}).items():
            msgs.append(self.attr_msg(k, v))
        return self.object_header(msgs), bt, heap_hdr

    _vlen_patches = []

    def finish(self, path, root):
        """root = (header address, btree address, heap address) of the root group"""
        gaddr = 0
        if self.gheap:
            body = b""
            for i, obj in enumerate(self.gheap, 1):
                body += struct.pack("<HH4xQ", i, 1, len(obj)) + _pad8(obj)
            size = 16 + len(body) + 16
            size += -size % 8
            free = size - 16 - len(body)
            body += struct.pack("<HH4xQ", 0, 0, free) + b"\0" * (free - 16)
            gaddr = self.alloc(b"GCOL" + struct.pack("<B3xQ", 1, size) + body)
            # patch the collection address into every vlen reference (length, address = 0 placeholder, index)
            raw = bytes(self.buf)
            for i, obj in enumerate(self.gheap, 1):
                needle = struct.pack("<IQI", len(obj), 0, i)
                at = raw.find(needle)
                assert at >= 0
                self.buf[at + 4:at + 12] = struct.pack("<Q", gaddr)
                raw = bytes(self.buf)
        hdr, bt, heap = root
        eof = len(self.buf)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 1024, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", bt, heap)
        assert len(sb) == 96
        self.buf[0:96] = sb
        with open(path, "wb") as fh:
            fh.write(bytes(self.buf))


def write_multi_fast5(path, reads, vbz=True, chunk=4096, vlen_strings=False, fanout=0):
    """
    reads: [(raw int16 array, attrs)] with the attrs of reads.write_bundle (read_id, range, digitisation, offset,
    sampling_rate, run_id, channel_number, start_mux, read_number, start_time, duration, exp_start_time, ...).
    Layout (ont_fast5_api multi-read files): /read_<id>/Raw{Signal + attrs}, /read_<id>/channel_id{attrs},
    /read_<id>/tracking_id{attrs}; the root carries file_type / file_version.
    """
    w = H5Writer(fanout=fanout)
    w._vlen_patches = []
    top = {}
    s = (lambda v: "vlen:" + v) if vlen_strings else (lambda v: v)
    for raw, a in reads:
        raw = np.asarray(raw, dtype=np.int16)
        sig = w.dataset(raw, chunks=chunk, vbz=vbz)
        raw_grp, _, _ = w.group({"Signal": sig}, attrs={
            "read_id": s(a["read_id"]), "start_mux": np.uint8(a.get("start_mux", 1)),
            "read_number": np.int32(a.get("read_number", 0)), "start_time": np.uint64(a.get("start_time", 0)),
            "duration": np.uint32(a.get("duration", len(raw))), "median_before": np.float64(200.0)})
        chan, _, _ = w.group({}, attrs={
            "channel_number": s(str(a.get("channel_number", "1"))), "digitisation": np.float64(a["digitisation"]),
            "offset": np.float64(a["offset"]), "range": np.float64(a["range"]),
            "sampling_rate": np.float64(a["sampling_rate"])})
        track, _, _ = w.group({}, attrs={
            "run_id": s(a.get("run_id", "")), "sample_id": s(a.get("sample_id", "sample")),
            "exp_start_time": s(a.get("exp_start_time", "1970-01-01T00:00:00Z")),
            "flow_cell_id": s(a.get("flow_cell_id", "FAK00000")), "device_id": s(a.get("device_id", "MN00000"))})
        rd, _, _ = w.group({"Raw": raw_grp, "channel_id": chan, "tracking_id": track},
                           attrs={"run_id": s(a.get("run_id", ""))})
        top["read_" + a["read_id"]] = rd
    root = w.group(top, attrs={"file_type": "multi-read", "file_version": "2.2"})
    w.finish(path, root)
