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
