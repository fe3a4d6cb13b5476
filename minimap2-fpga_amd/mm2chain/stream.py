"""Anchor streams on disk (MM2ANCH1, layout in csrc/anchor_stream.c): numpy reader/writer and the C-side entry points."""
import ctypes as C
import struct

import numpy as np

from . import _native as N
from .params import make_params

HDR = 128
MAGIC = b"MM2ANCH1"


def write(path, par, offsets, anchors, min_cnt=3, min_sc=40):
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    base = int(off[0]) if off.size else 0
    total = int(off[-1]) - base if off.size else 0
    h = struct.pack("<8sIIqq9if", MAGIC, 1, HDR, off.size - 1, total, par.max_dist_x, par.max_dist_y, par.bw, par.max_skip, par.max_iter,
                    par.is_cdna, par.n_segs, min_cnt, min_sc, par.gap_scale)
    with open(path, "wb") as fp:
        fp.write(h + b"\0" * (HDR - len(h)))
        fp.write((off - base).tobytes())
        fp.write(a[base:base + total].tobytes())


def read(path):
    """returns (params, min_cnt, min_sc, offsets int64, anchors uint64 [n,2])"""
    with open(path, "rb") as fp:
        raw = fp.read()
    magic, ver, hb, n_tasks, total, mdx, mdy, bw, ms, mi, cdna, nseg, min_cnt, min_sc, gs = struct.unpack_from("<8sIIqq9if", raw, 0)
    if magic != MAGIC or ver != 1:
        raise ValueError("not an MM2ANCH1 stream")
    off = np.frombuffer(raw, dtype=np.int64, count=n_tasks + 1, offset=hb).copy()
    a = np.frombuffer(raw, dtype=np.uint64, count=2 * total, offset=hb + 8 * (n_tasks + 1)).reshape(total, 2).copy()
    return make_params(mdx, mdy, bw, ms, mi, gs, cdna, nseg), min_cnt, min_sc, off, a


def _from_c(st):
    n, tot = st.n_tasks, st.total
    off = np.ctypeslib.as_array(st.offsets, shape=(n + 1,)).copy()
    a = np.ctypeslib.as_array(C.cast(st.anchors, C.POINTER(C.c_uint64)), shape=(tot, 2)).copy() if tot else np.zeros((0, 2), np.uint64)
    par = make_params(st.par.max_dist_x, st.par.max_dist_y, st.par.bw, st.par.max_skip, st.par.max_iter, st.par.gap_scale,
                      st.par.is_cdna, st.par.n_segs)
    return par, st.min_cnt, st.min_sc, off, a


def read_c(path):
    """the same through mm2c_stream_read"""
    lib = N.load()
    st = N.Stream()
    N.check(lib.mm2c_stream_read(str(path).encode(), C.byref(st)), "mm2c_stream_read")
    out = _from_c(st)
    lib.mm2c_stream_free(C.byref(st))
    return out


def write_c(path, par, offsets, anchors, min_cnt=3, min_sc=40):
    lib = N.load()
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    a = np.ascontiguousarray(anchors).view(np.uint64).reshape(-1, 2)
    N.check(lib.mm2c_stream_write(str(path).encode(), C.byref(par), min_cnt, min_sc, off.size - 1, off.ctypes.data_as(C.c_void_p),
                                  a.ctypes.data_as(C.c_void_p)), "mm2c_stream_write")


def from_seed_dump(path, par, min_cnt=3, min_sc=40):
    """import `minimap2 --print-seeds` RS / SD lines (map.c:298-303) through mm2c_stream_from_seed_dump"""
    lib = N.load()
    st = N.Stream()
    N.check(lib.mm2c_stream_from_seed_dump(str(path).encode(), C.byref(par), min_cnt, min_sc, C.byref(st)), "mm2c_stream_from_seed_dump")
    out = _from_c(st)
    lib.mm2c_stream_free(C.byref(st))
    return out
