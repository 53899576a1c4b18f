"""Minimal reader for the MATLAB v7.3 (.mat = HDF5) prior files of the reference
(`gauss_priors.mat`, `UAV2_ob_priors_train.mat`, `AVS1K_ob_priors_train.mat`), which the reference
loads with `hdf5storage.loadmat(path)["PriorMaps"]` (utils_data.py:459, 587).  Neither h5py nor
hdf5storage is available on the target image, so this restates just enough of the HDF5 file format
(superblock v0/v1, v1 object headers, v1 group B-trees + local heaps, contiguous and chunked
layouts, the shuffle / deflate / fletcher32 filters) to read one numeric dataset from the root group.
SURVEY.md 8(f) rank 2.  Pure Python + numpy + zlib; host-side I/O, not on the timed path.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _H5:
    def __init__(self, data: bytes):
        self.b = data
        sb = -1
        off = 0
        while off < len(data):                # the superblock sits at 0, 512, 1024, ... (MATLAB: 512)
            if data[off:off + 8] == _SIG:
                sb = off
                break
            off = 512 if off == 0 else off * 2
        if sb < 0:
            raise ValueError("not an HDF5 / MATLAB v7.3 file")
        ver = data[sb + 8]
        if ver not in (0, 1):
            raise ValueError("unsupported HDF5 superblock version %d" % ver)
        self.O, self.L = data[sb + 13], data[sb + 14]
        if self.O != 8 or self.L != 8:
            raise ValueError("only 8-byte offsets/lengths are supported")
        p = sb + 24 + (4 if ver == 1 else 0)
        self.base = self.u64(p)               # all file addresses are relative to this
        p += 4 * 8
        # root group symbol table entry
        self.root_hdr = self.u64(p + 8)
        cache = self.u32(p + 16)
        self.root_btree = self.u64(p + 24) if cache == 1 else None
        self.root_heap = self.u64(p + 32) if cache == 1 else None

    # raw accessors: `a` is an absolute position, `A(x)` converts a file address
    def u16(self, a): return struct.unpack_from("<H", self.b, a)[0]
    def u32(self, a): return struct.unpack_from("<I", self.b, a)[0]
    def u64(self, a): return struct.unpack_from("<Q", self.b, a)[0]
    def A(self, addr): return addr + self.base

    def heap_name(self, heap_addr, off):
        h = self.A(heap_addr)
        assert self.b[h:h + 4] == b"HEAP"
        seg = self.A(self.u64(h + 24))
        end = self.b.index(b"\x00", seg + off)
        return self.b[seg + off:end].decode("ascii")

    def group_entries(self, btree_addr, heap_addr) -> Dict[str, int]:
        """name -> object header address, walking a v1 group B-tree."""
        out: Dict[str, int] = {}
        t = self.A(btree_addr)
        assert self.b[t:t + 4] == b"TREE" and self.b[t + 4] == 0
        level, used = self.b[t + 5], self.u16(t + 6)
        p = t + 8 + 16
        for i in range(used):
            child = self.u64(p + 8)           # key_i (8) then child_i (8)
            p += 16
            if level > 0:
                out.update(self.group_entries(child, heap_addr))
            else:
                s = self.A(child)
                assert self.b[s:s + 4] == b"SNOD"
                n = self.u16(s + 6)
                for k in range(n):
                    e = s + 8 + 40 * k
                    out[self.heap_name(heap_addr, self.u64(e))] = self.u64(e + 8)
        return out

    def messages(self, hdr_addr):
        """(type, absolute data position, size) of every v1 object-header message."""
        h = self.A(hdr_addr)
        if self.b[h] != 1:
            raise ValueError("only version-1 object headers are supported")
        nmsg, size = self.u16(h + 2), self.u32(h + 8)
        blocks = [(h + 16, size)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(msgs) < nmsg:
                mtype, msize = self.u16(p), self.u16(p + 2)
                dpos = p + 8
                if mtype == 0x10:              # continuation block
                    blocks.append((self.A(self.u64(dpos)), self.u64(dpos + 8)))
                msgs.append((mtype, dpos, msize))
                p = dpos + ((msize + 7) & ~7)
        return msgs

    def read_dataset(self, hdr_addr) -> np.ndarray:
        dims = dtype = layout = None
        filters = []
        for mtype, p, size in self.messages(hdr_addr):
            if mtype == 0x01:                  # dataspace
                ver, rank, flags = self.b[p], self.b[p + 1], self.b[p + 2]
                q = p + (8 if ver == 1 else 4)
                dims = [self.u64(q + 8 * i) for i in range(rank)]
            elif mtype == 0x03:                # datatype
                cls = self.b[p] & 0x0F
                bits0 = self.b[p + 1]
                esize = self.u32(p + 4)
                if bits0 & 1:
                    raise ValueError("big-endian data is not supported")
                if cls == 1:
                    dtype = {4: np.float32, 8: np.float64}[esize]
                elif cls == 0:
                    signed = (bits0 >> 3) & 1
                    dtype = {1: (np.uint8, np.int8), 2: (np.uint16, np.int16), 4: (np.uint32, np.int32),
                             8: (np.uint64, np.int64)}[esize][signed]
                else:
                    raise ValueError("unsupported datatype class %d" % cls)
            elif mtype == 0x08:                # data layout (version 3)
                if self.b[p] != 3:
                    raise ValueError("unsupported data layout version %d" % self.b[p])
                lclass = self.b[p + 1]
                if lclass == 1:
                    layout = ("contiguous", self.u64(p + 2), self.u64(p + 10))
                elif lclass == 2:
                    nd = self.b[p + 2]
                    btree = self.u64(p + 3)
                    cdims = [self.u32(p + 11 + 4 * i) for i in range(nd)]
                    layout = ("chunked", btree, cdims)
                elif lclass == 0:
                    layout = ("compact", p + 4, self.u16(p + 2))
            elif mtype == 0x0B:                # filter pipeline
                ver, nf = self.b[p], self.b[p + 1]
                q = p + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = self.u16(q)
                    if ver == 1 or fid >= 256:
                        nlen = self.u16(q + 2)
                        q += 2
                    else:
                        nlen = 0
                    ncd = self.u16(q + 4)
                    q += 6
                    q += (nlen + 7) & ~7 if ver == 1 else nlen
                    cd = [self.u32(q + 4 * i) for i in range(ncd)]
                    q += 4 * ncd
                    if ver == 1 and ncd % 2:
                        q += 4
                    filters.append((fid, cd))
        if dims is None or dtype is None or layout is None:
            raise ValueError("object is not a simple numeric dataset")
        n = int(np.prod(dims)) if dims else 1
        esz = np.dtype(dtype).itemsize
        if layout[0] == "contiguous":
            a = self.A(layout[1])
            return np.frombuffer(self.b, dtype=dtype, count=n, offset=a).reshape(dims).copy()
        if layout[0] == "compact":
            return np.frombuffer(self.b, dtype=dtype, count=n, offset=layout[1]).reshape(dims).copy()
        out = np.zeros(dims, dtype=dtype)
        cdims = layout[2][:-1]
        for offs, addr, csize, mask in self._chunks(layout[1], len(layout[2])):
            raw = self.b[self.A(addr):self.A(addr) + csize]
            for i in reversed(range(len(filters))):
                if mask & (1 << i):
                    continue
                fid, cd = filters[i]
                if fid == 3:                    # fletcher32: checksum appended
                    raw = raw[:-4]
                elif fid == 1:                  # deflate
                    raw = zlib.decompress(raw)
                elif fid == 2:                  # shuffle
                    k = cd[0] if cd else esz
                    raw = np.frombuffer(raw, dtype=np.uint8).reshape(k, -1).T.tobytes()
                else:
                    raise ValueError("unsupported HDF5 filter %d" % fid)
            chunk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims))).reshape(cdims)
            sl = tuple(slice(o, min(o + c, d)) for o, c, d in zip(offs, cdims, dims))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks(self, btree_addr, nd):
        t = self.A(btree_addr)
        assert self.b[t:t + 4] == b"TREE" and self.b[t + 4] == 1
        level, used = self.b[t + 5], self.u16(t + 6)
        p = t + 8 + 16
        ksz = 8 + 8 * nd
        for _ in range(used):
            csize, mask = self.u32(p), self.u32(p + 4)
            offs = [self.u64(p + 8 + 8 * i) for i in range(nd - 1)]
            child = self.u64(p + ksz)
            p += ksz + 8
            if level > 0:
                yield from self._chunks(child, nd)
            else:
                yield offs, child, csize, mask


def loadmat(path: str) -> Dict[str, np.ndarray]:
    """`{name: array}` for every numeric dataset in the root group, in MATLAB / hdf5storage
    orientation (HDF5 stores MATLAB arrays with reversed dimension order)."""
    with open(path, "rb") as f:
        h5 = _H5(f.read())
    if h5.root_btree is None:
        for mtype, p, _ in h5.messages(h5.root_hdr):
            if mtype == 0x11:
                h5.root_btree, h5.root_heap = h5.u64(p), h5.u64(p + 8)
    if h5.root_btree is None:
        raise ValueError("root group has no symbol table")
    out = {}
    for name, hdr in h5.group_entries(h5.root_btree, h5.root_heap).items():
        if name.startswith("#"):
            continue
        try:
            arr = h5.read_dataset(hdr)
        except ValueError:
            continue
        out[name] = arr.transpose()            # reversed dimension order -> MATLAB orientation
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Writer: the caller loop's result file (Demo_Test.py:93-95: `h5io.savemat(path, {'salmap': pred_mat})`, uint8
# `[H, W, 1, F]`).  MATLAB v7.3 = a 512-byte text header + an HDF5 file whose datasets carry a `MATLAB_class` attribute
# and store the array with reversed dimension order.  Written with the same structures the reader above parses (and the
# reference's own .mat files use): superblock v0, v1 object headers, one v1 group B-tree node + local heap + symbol
# node for the root group, contiguous data, no filters.
_MATLAB_CLASS = {"uint8": "uint8", "int8": "int8", "uint16": "uint16", "int16": "int16", "uint32": "uint32",
                 "int32": "int32", "uint64": "uint64", "int64": "int64", "float32": "single", "float64": "double"}


def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _datatype_msg(dt: np.dtype) -> bytes:
    if dt.kind == "f":
        if dt.itemsize == 4:        # IEEE little-endian: sign bit 31, exponent 23 / 8 bits bias 127, mantissa 0 / 23 bits
            return bytes.fromhex("11201f00") + struct.pack("<IHHBBBBI", 4, 0, 32, 23, 8, 0, 23, 127)
        return bytes.fromhex("11203f00") + struct.pack("<IHHBBBBI", 8, 0, 64, 52, 11, 0, 52, 1023)
    signed = 0x08 if dt.kind == "i" else 0x00
    return bytes([0x10, signed, 0, 0]) + struct.pack("<IHH", dt.itemsize, 0, 8 * dt.itemsize)


def _string_attr(name: str, value: str) -> bytes:
    """Attribute message v1: a scalar fixed-length (null-padded) ASCII string."""
    nm = name.encode("ascii") + b"\x00"
    val = value.encode("ascii")
    dtype = bytes([0x13, 0x01, 0x00, 0x00]) + struct.pack("<I", len(val))       # class 3 (string), null-pad, ASCII
    dspace = bytes([1, 0, 0, 0, 0, 0, 0, 0])                                    # v1, rank 0 (scalar)
    return struct.pack("<BxHHH", 1, len(nm), len(dtype), len(dspace)) + _pad8(nm) + _pad8(dtype) + _pad8(dspace) + val


def savemat(path: str, mdict: Dict[str, np.ndarray]) -> None:
    """`hdf5storage.savemat(path, mdict)` for numeric arrays (the result writer of Demo_Test.py:93-95).  At most 8
    variables (one symbol-table node)."""
    names = sorted(mdict)
    if not names or len(names) > 8:
        raise ValueError("savemat writes 1..8 variables")
    arrs = {}
    for n in names:
        a = np.asarray(mdict[n])
        if a.dtype.name not in _MATLAB_CLASS or not n.isidentifier() or len(n) > 63:
            raise ValueError("variable %r: unsupported dtype %s or name" % (n, a.dtype))
        arrs[n] = a.astype(a.dtype.newbyteorder("<"), copy=False)
    SB, BT, K_LEAF, K_INT = 96, 544, 4, 16
    root_hdr = SB
    btree = root_hdr + 16 + 24
    heap = btree + BT
    # local heap data segment: "" at 0, then the names, 8-byte aligned, then one free block
    seg = bytearray(8)
    name_off = {}
    for n in names:
        name_off[n] = len(seg)
        seg += _pad8(n.encode("ascii") + b"\x00")
    free_off = len(seg)
    seg += struct.pack("<QQ", 1, 16)                       # free block: no next (1), its own size
    heap_seg = heap + 32
    pos = heap_seg + len(seg)
    # dataset object headers
    hdrs, hdr_addr, data_addr = {}, {}, {}
    bodies = {}
    for n in names:
        a = arrs[n]
        dims = a.shape[::-1] if a.ndim else ()
        body = _msg(0x01, bytes([1, len(dims), 0, 0, 0, 0, 0, 0]) + b"".join(struct.pack("<Q", d) for d in dims))
        body += _msg(0x03, _datatype_msg(a.dtype), flags=1)
        body += _msg(0x05, bytes([2, 2, 0, 1]) + struct.pack("<I", 0))          # fill value v2: late allocation, default fill
        bodies[n] = body
        hdr_addr[n] = pos
        pos += 16 + len(body) + 8 + 24 + len(_msg(0x0C, _string_attr("MATLAB_class", _MATLAB_CLASS[a.dtype.name])))
    snod = pos
    pos += 8 + 2 * K_LEAF * 40
    for n in names:
        data_addr[n] = pos
        pos += (arrs[n].nbytes + 7) & ~7
    eof = pos
    for n in names:
        a = arrs[n]
        body = bodies[n] + _msg(0x08, bytes([3, 1]) + struct.pack("<QQ", data_addr[n], a.nbytes))
        body += _msg(0x0C, _string_attr("MATLAB_class", _MATLAB_CLASS[a.dtype.name]))
        hdrs[n] = struct.pack("<BxHII4x", 1, 5, 1, len(body)) + body
    out = bytearray()
    text = "MATLAB 7.3 MAT-file, Platform: iip_uavsal_saliency_amd.matio, Created on: %s HDF5 schema 1.00 ." % __import__("time").strftime("%a %b %d %H:%M:%S %Y")
    out += text.encode("ascii")[:116].ljust(116, b" ") + bytes(8) + bytes([0x00, 0x02]) + b"IM"
    out += bytes(512 - len(out))
    base = 512
    sb = bytearray(_SIG)
    sb += bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", K_LEAF, K_INT, 0)
    sb += struct.pack("<QQQQ", base, _UNDEF, base + eof, _UNDEF)
    sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", btree, heap)
    assert len(sb) == SB
    out += sb
    out += struct.pack("<BxHII4x", 1, 1, 1, 24) + _msg(0x11, struct.pack("<QQ", btree, heap))
    node = bytearray(b"TREE" + bytes([0, 0]) + struct.pack("<H", 1) + struct.pack("<QQ", _UNDEF, _UNDEF))
    node += struct.pack("<QQQ", 0, snod, name_off[names[-1]])             # key 0, child 0, key 1 = the largest name
    out += bytes(node).ljust(BT, b"\x00")
    out += b"HEAP" + bytes(4) + struct.pack("<QQQ", len(seg), free_off, heap_seg)
    out += seg
    for n in names:
        assert len(out) - base == hdr_addr[n]
        out += hdrs[n]
    assert len(out) - base == snod
    sn = bytearray(b"SNOD" + bytes([1, 0]) + struct.pack("<H", len(names)))
    for n in names:
        sn += struct.pack("<QQII16x", name_off[n], hdr_addr[n], 0, 0)
    out += bytes(sn).ljust(8 + 2 * K_LEAF * 40, b"\x00")
    for n in names:
        assert len(out) - base == data_addr[n]
        out += _pad8(arrs[n].tobytes(order="F"))       # column-major == the transposed array in C order
    assert len(out) - base == eof
    with open(path, "wb") as f:
        f.write(bytes(out))
