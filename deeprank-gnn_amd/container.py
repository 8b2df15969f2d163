"""Native on-disk container of graph data ("DRGS", version 1): a named set of raw little-endian arrays.

Why: the reference keeps graphs in HDF5 (writer ``Graph.nx2h5`` reference Graph.py:61-139, reader
``HDF5DataSet.load_one_graph`` reference DataSet.py:231-366) and needs h5py to touch them; the MI355X
image has no h5py and the hot path wants flat arrays it can upload with one copy.  One file format serves

* ``tree/<mol>/<dataset path>``  -- a lossless mirror of the reference's HDF5 tree (same group / dataset names,
  dtypes and shapes, byte-string datasets included): what ``GraphStore`` reads and writes, what
  ``tools/hdf5_to_native.py`` / ``tools/native_to_hdf5.py`` convert from / to ``.hdf5`` in an environment with h5py;
* ``set/<name>``   -- the concatenated graph-major arrays of a ``ResidentGraphSet`` (x, local edge_index, edge_attr,
  clusters, targets, int64 offset tables): the image that is uploaded to HBM as it is;
* ``topo/<name>``  -- the cached per-graph topology of that set: the int32 workspace (CSR / CSC of the input graph,
  consecutive depth-0 clusters + member lists, pooled CSR / CSC, depth-1 clusters + member lists) and the f32
  workspace (edge weights in CSR order, summed pooled weights), with the layout offsets they were written with.

Layout:  8-byte magic ``DRGSET\\0\\1`` | uint64 header length | header (UTF-8 JSON) | zero padding to 64 bytes |
sections, each starting on a 64-byte boundary.  Header: ``{"version": 1, "meta": {...}, "sections": {name: {"dtype":
numpy dtype string, "shape": [...], "offset": byte offset from the start of the file, "nbytes": n}}}``.

This module imports numpy and json only (no torch, nothing of the package): the converters load it by file path
from a Python that has h5py but no torch.
"""
import json
import struct

import numpy as np

MAGIC = b"DRGSET\x00\x01"
ALIGN = 64

__all__ = ["write_container", "read_container", "read_header", "MAGIC"]


def _pad(n):
    return (-n) % ALIGN


def write_container(path, sections, meta=None):
    """``sections``: {name: ndarray} (any numeric or fixed-width byte-string dtype); ``meta``: JSON-able dict."""
    names = list(sections)
    arrays = []
    for name in names:
        a = np.asarray(sections[name])
        if a.dtype == object:
            raise TypeError("section %r has object dtype" % name)
        if a.dtype.kind == "U":
            a = np.char.encode(a, "utf-8")
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        # (ascontiguousarray would turn a 0-d dataset into shape (1,): keep the shape)
        arrays.append(np.ascontiguousarray(a).reshape(a.shape))
    # two passes: the header's size depends on the offsets' digits, so lay out with a generous fixed header area
    table = {n: {"dtype": a.dtype.str, "shape": list(a.shape), "offset": 0, "nbytes": int(a.nbytes)}
             for n, a in zip(names, arrays)}
    header = {"version": 1, "meta": meta or {}, "sections": table}
    probe = json.dumps(header).encode("utf-8")
    area = len(probe) + 24 * len(names) + 64                 # room for the final offsets
    start = 16 + area
    start += _pad(start)
    off = start
    for n, a in zip(names, arrays):
        table[n]["offset"] = off
        off += a.nbytes
        off += _pad(off)
    blob = json.dumps(header).encode("utf-8")
    assert len(blob) <= area
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(blob)))
        f.write(blob)
        f.write(b"\0" * (start - 16 - len(blob)))
        pos = start
        for n, a in zip(names, arrays):
            assert pos == table[n]["offset"]
            f.write(a.tobytes())
            pos += a.nbytes
            pad = _pad(pos)
            f.write(b"\0" * pad)
            pos += pad
    return header


def read_header(path):
    with open(path, "rb") as f:
        magic = f.read(8)
        if magic != MAGIC:
            raise ValueError("%s is not a DRGS container (bad magic %r)" % (path, magic))
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n).decode("utf-8"))
    if header.get("version") != 1:
        raise ValueError("%s: unsupported container version %r" % (path, header.get("version")))
    return header


def read_container(path, prefix=None, mmap=False):
    """(meta, {name: ndarray}) of the sections whose name starts with ``prefix`` (all when None).  ``mmap``: arrays
    are read-only views of the file instead of copies."""
    header = read_header(path)
    out = {}
    raw = np.memmap(path, dtype=np.uint8, mode="r") if mmap else None
    with open(path, "rb") as f:
        for name, d in header["sections"].items():
            if prefix is not None and not name.startswith(prefix):
                continue
            dt = np.dtype(d["dtype"])
            count = int(np.prod(d["shape"], dtype=np.int64)) if d["shape"] else 1
            if mmap:
                a = raw[d["offset"]:d["offset"] + d["nbytes"]].view(dt)
            else:
                f.seek(d["offset"])
                a = np.frombuffer(f.read(d["nbytes"]), dtype=dt, count=count).copy() if d["nbytes"] else np.zeros(0, dt)
            out[name] = a.reshape(d["shape"])
    return header["meta"], out
