"""Minimal pure-Python HDF5 reader / writer for the files DOLFINx's XDMFFile writes next to its .xdmf (host side, no GPU).

The reference reads and writes its meshes as XDMF with the heavy data in HDF5 (/root/reference/examples/01_obstacle_problem/
obstacle_pg.py:64-65 `XDMFFile(..., "r").read_mesh(name="mesh")`; generate_mesh_gmsh.py:40-43 and src/lvpp/mesh_generation.py:158-168
`write_mesh` / `write_meshtags`).  Neither h5py nor a Python HDF5 binding exists in this environment, so the subset of the format
those files use is read here directly (HDF5 File Format Specification 3.0):

* superblock version 0 or 1 (what libhdf5 writes with its default "earliest" format bounds, i.e. what DOLFINx and h5py produce);
* old-style groups: symbol-table message -> version-1 B-tree of group nodes -> symbol nodes (SNOD) -> names in a local heap;
* version-1 object headers with continuation blocks;
* datasets with CONTIGUOUS or COMPACT layout (data-layout message version 3; DOLFINx writes contiguous, unfiltered datasets
  through parallel HDF5) of little-endian integers (1/2/4/8 bytes, signed or not) and IEEE floats (4/8 bytes).
Anything else (chunked / filtered datasets, version-2 superblocks with fractal-heap groups) raises NotImplementedError naming the
feature.  `write` produces files of the same subset (superblock 0, one symbol node per group), which libhdf5's own `h5dump` reads
(tools/make_h5_fixtures.py checks that where the tool exists); the reader is tested against fixtures written by the real libhdf5
(tests/golden/*.h5, tools/make_h5_fixtures.c).
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    """Read-only view: f["/Mesh/mesh/geometry"] -> numpy array; f.keys("/Mesh") -> child names."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.b = fh.read()
        b = self.b
        base = b.find(_SIG)
        if base != 0:
            raise ValueError(f"{path}: not an HDF5 file (no signature at offset 0)")
        ver = b[8]
        if ver > 1:
            raise NotImplementedError(f"{path}: superblock version {ver} (new-style groups) - written with libver='latest'; re-save "
                                      "with the default format bounds")
        self.so, self.sl = b[13], b[14]  # size of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise NotImplementedError(f"{path}: {self.so}-byte offsets / {self.sl}-byte lengths")
        p = 24 + (4 if ver == 1 else 0)
        self.base_addr = self._u(p, 8)
        p += 4 * 8  # base, free-space info, end of file, driver info
        self.root = self._symbol_entry(p)

    # -- primitives ----------------------------------------------------------------------------------
    def _u(self, off, n):
        return int.from_bytes(self.b[off: off + n], "little")

    def _symbol_entry(self, p):
        return {"name_off": self._u(p, 8), "header": self._u(p + 8, 8), "cache": self._u(p + 16, 4),
                "btree": self._u(p + 24, 8), "heap": self._u(p + 32, 8)}

    def _messages(self, addr):
        """(type, data bytes) of a version-1 object header, continuation blocks followed."""
        b = self.b
        if b[addr] != 1:
            if b[addr: addr + 4] == b"OHDR":
                raise NotImplementedError("version-2 object headers (libver='latest')")
            raise ValueError(f"object header version {b[addr]} at {addr}")
        nmsg = self._u(addr + 2, 2)
        size = self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                data = b[p + 8: p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:  # continuation
                    blocks.append((int.from_bytes(data[:8], "little"), int.from_bytes(data[8:16], "little")))
                out.append((mtype, data))
        return out

    def _heap_name(self, heap, off):
        b = self.b
        if b[heap: heap + 4] != b"HEAP":
            raise ValueError("local heap signature")
        seg = self._u(heap + 24, 8)
        e = b.index(b"\0", seg + off)
        return b[seg + off: e].decode()

    def _children(self, btree, heap):
        """name -> symbol table entry, walking the group B-tree."""
        b = self.b
        out = {}
        if b[btree: btree + 4] != b"TREE":
            raise ValueError("B-tree signature")
        level, used = b[btree + 5], self._u(btree + 6, 2)
        p = btree + 8 + 16  # past the sibling pointers
        for k in range(used):
            child = self._u(p + 8 + k * 16, 8)  # key_k (8), child_k (8), ...
            if level > 0:
                out.update(self._children(child, heap))
            else:
                if b[child: child + 4] != b"SNOD":
                    raise ValueError("symbol node signature")
                n = self._u(child + 6, 2)
                for i in range(n):
                    e = self._symbol_entry(child + 8 + 40 * i)
                    out[self._heap_name(heap, e["name_off"])] = e
        return out

    def _group_tables(self, entry):
        if entry["cache"] == 1:
            return entry["btree"], entry["heap"]
        for t, d in self._messages(entry["header"]):
            if t == 0x0011:
                return int.from_bytes(d[:8], "little"), int.from_bytes(d[8:16], "little")
        raise KeyError("not a group")

    def _resolve(self, path):
        e = self.root
        for part in [q for q in path.split("/") if q]:
            kids = self._children(*self._group_tables(e))
            if part not in kids:
                raise KeyError(f"{path}: no object {part!r} (found {sorted(kids)})")
            e = kids[part]
        return e

    # -- public --------------------------------------------------------------------------------------
    def keys(self, path="/"):
        return sorted(self._children(*self._group_tables(self._resolve(path))))

    def __getitem__(self, path):
        e = self._resolve(path)
        shape = dtype = layout = None
        for t, d in self._messages(e["header"]):
            if t == 0x0001:  # dataspace
                ver, rank = d[0], d[1]
                p = 8 if ver == 1 else 4
                shape = tuple(int.from_bytes(d[p + 8 * k: p + 8 * k + 8], "little") for k in range(rank))
            elif t == 0x0003:  # datatype
                cls, size = d[0] & 0x0F, int.from_bytes(d[4:8], "little")
                if d[1] & 1:
                    raise NotImplementedError("big-endian datasets")
                if cls == 0:
                    dtype = np.dtype(("<i" if d[1] & 8 else "<u") + str(size))
                elif cls == 1:
                    dtype = np.dtype("<f" + str(size))
                else:
                    raise NotImplementedError(f"datatype class {cls}")
            elif t == 0x0008:  # data layout
                if d[0] != 3:
                    raise NotImplementedError(f"data layout message version {d[0]}")
                if d[1] == 1:
                    layout = ("contiguous", int.from_bytes(d[2:10], "little"), int.from_bytes(d[10:18], "little"))
                elif d[1] == 0:
                    n = int.from_bytes(d[2:4], "little")
                    layout = ("compact", d[4: 4 + n])
                else:
                    raise NotImplementedError("chunked datasets (DOLFINx writes contiguous ones; re-save without chunking / filters)")
            elif t == 0x000B:
                raise NotImplementedError("filtered (compressed) datasets")
        if shape is None or dtype is None or layout is None:
            raise KeyError(f"{path}: not a dataset")
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "compact":
            raw = layout[1]
        else:
            addr = layout[1]
            if addr == _UNDEF:
                return np.zeros(shape, dtype=dtype)
            raw = self.b[self.base_addr + addr: self.base_addr + addr + n * dtype.itemsize]
        return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).copy()


# ---------------------------------------------------------------------------------------------------
def _pad8(n):
    return (n + 7) & ~7


def write(path, datasets: dict):
    """Write {"/Mesh/mesh/geometry": array, ...} as an HDF5 file of the subset described in the module docstring (superblock 0,
    old-style groups with one symbol node each, contiguous little-endian datasets of int32/int64/float32/float64)."""
    tree: dict = {}
    for name, arr in datasets.items():
        node = tree
        parts = [q for q in name.split("/") if q]
        for q in parts[:-1]:
            node = node.setdefault(q, {})
            if not isinstance(node, dict):
                raise ValueError(f"{name}: {q} is a dataset")
        node[parts[-1]] = np.ascontiguousarray(arr)
    buf = bytearray(96)  # superblock, filled at the end

    def alloc(n):
        off = len(buf)
        buf.extend(b"\0" * _pad8(n))
        return off

    def msg(t, data, flags=0):
        data = data + b"\0" * (_pad8(len(data)) - len(data))
        return struct.pack("<HHB3x", t, len(data), flags) + data

    def header(msgs):
        body = b"".join(msgs)
        off = alloc(16 + len(body))
        buf[off: off + 16] = struct.pack("<BxHII4x", 1, len(msgs), 1, len(body))
        buf[off + 16: off + 16 + len(body)] = body
        return off

    def dataset(a):
        if a.dtype.kind == "f":
            size = a.dtype.itemsize
            prec, eloc, esize, mloc, msize, bias = (64, 52, 11, 0, 52, 1023) if size == 8 else (32, 23, 8, 0, 23, 127)
            dt = struct.pack("<BBBBI", 0x11, 0x20, prec - 1, 0, size) + struct.pack("<HHBBBBI", 0, prec, eloc, esize, mloc, msize, bias)
        elif a.dtype.kind in "iu":
            size = a.dtype.itemsize
            dt = struct.pack("<BBBBI", 0x10, 0x08 if a.dtype.kind == "i" else 0, 0, 0, size) + struct.pack("<HH", 0, 8 * size)
        else:
            raise NotImplementedError(f"dtype {a.dtype}")
        a = a.astype(a.dtype.newbyteorder("<"), copy=False)
        raw = a.tobytes()
        daddr = alloc(len(raw)) if raw else _UNDEF
        if raw:
            buf[daddr: daddr + len(raw)] = raw
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
        layout = struct.pack("<BBQQ", 3, 1, daddr, len(raw))
        fill = struct.pack("<BBBB", 2, 2, 2, 0)  # fill value message v2: allocate late, write if set, undefined
        return header([msg(0x0001, space), msg(0x0003, dt, 1), msg(0x0005, fill, 1), msg(0x0008, layout)])

    def group(node):
        """-> (object header address, B-tree address, heap address)"""
        names = sorted(node)
        if len(names) > 8:
            raise NotImplementedError("more than 8 objects in one group")
        heap_data = bytearray(b"\0" * 8)  # offset 0: the empty name
        entries = []
        for nm in names:
            child = node[nm]
            noff = len(heap_data)
            heap_data.extend(nm.encode() + b"\0")
            heap_data.extend(b"\0" * (_pad8(len(heap_data)) - len(heap_data)))
            if isinstance(child, dict):
                hdr, bt, hp = group(child)
                entries.append(struct.pack("<QQII", noff, hdr, 1, 0) + struct.pack("<QQ", bt, hp))
            else:
                entries.append(struct.pack("<QQII", noff, dataset(child), 0, 0) + b"\0" * 16)
        free_off = len(heap_data)
        heap_data.extend(struct.pack("<QQ", 1, 16))  # one free block: next = 1 (none), size 16
        seg = alloc(len(heap_data))
        buf[seg: seg + len(heap_data)] = heap_data
        heap = alloc(32)
        buf[heap: heap + 32] = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, seg)
        snod = alloc(8 + 40 * 8)
        buf[snod: snod + 8] = b"SNOD" + struct.pack("<BxH", 1, len(entries))
        for i, e in enumerate(entries):
            buf[snod + 8 + 40 * i: snod + 8 + 40 * (i + 1)] = e
        # B-tree node of group type, level 0, one child; room for 2K = 32 children as libhdf5 expects (internal K = 16)
        bt = alloc(24 + (2 * 16 + 1) * 8 + 2 * 16 * 8)
        last = struct.unpack("<Q", entries[-1][:8])[0] if entries else 0
        buf[bt: bt + 24] = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, _UNDEF, _UNDEF)
        buf[bt + 24: bt + 48] = struct.pack("<QQQ", 0, snod, last)
        hdr = header([msg(0x0011, struct.pack("<QQ", bt, heap))])
        return hdr, bt, heap

    hdr, bt, heap = group(tree)
    eof = len(buf)
    sb = _SIG + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
    sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", bt, heap)
    assert len(sb) == 96
    buf[:96] = sb
    with open(path, "wb") as fh:
        fh.write(bytes(buf))
