"""ctypes loader for the CPU oracle (oracle/o2v_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (obj2voxel_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libo2v_oracle.so")

TRI_MATERIALLESS, TRI_UNTEXTURED, TRI_TEXTURED = 1, 2, 3
STRATEGY_MAX, STRATEGY_BLEND = 0, 1


class _Texture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("channels", C.c_uint32), ("wrap", C.c_uint32)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("candidates", "culled", "clips", "hits", "pieces", "splits", "leaves")]


def build(force=False):
    src = os.path.join(_HERE, "o2v_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.o2v_oracle_voxelize.restype = C.c_int64
        _lib.o2v_oracle_voxelize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_uint32))]
        _lib.o2v_oracle_free.argtypes = [C.POINTER(C.c_uint32)]
        _lib.o2v_oracle_mesh_transform.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib.o2v_oracle_get_stats.argtypes = [C.POINTER(_Stats)]
        _lib.o2v_oracle_set_threads.argtypes = [C.c_int]
        _lib.o2v_oracle_get_phase_seconds.argtypes = [C.c_void_p]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def voxelize(verts, resolution, *, uvs=None, types=None, colors=None, texids=None, textures=(),
             supersampling=1, strategy=STRATEGY_MAX, unit_transform=None, bounds=None, zslab=(0, 0)):
    """Run the oracle. verts: float32 [T,9] (or [T,3,3]). Returns uint32 [V,4] = (x,y,z,argb), unsorted.

    textures: sequence of (pixels uint8 [h,w,c], wrap) tuples.
    """
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    T = verts.shape[0]
    uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(T, 6)
    types = None if types is None else np.ascontiguousarray(types, dtype=np.uint32).reshape(T)
    colors = None if colors is None else np.ascontiguousarray(colors, dtype=np.float32).reshape(T, 3)
    texids = None if texids is None else np.ascontiguousarray(texids, dtype=np.int32).reshape(T)
    unit = None if unit_transform is None else np.ascontiguousarray(unit_transform, dtype=np.int32).reshape(9)
    bnd = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.float32).reshape(6)
    keep = []
    tex_arr = (_Texture * max(1, len(textures)))()
    for i, (pix, wrap) in enumerate(textures):
        pix = np.ascontiguousarray(pix, dtype=np.uint8)
        keep.append(pix)
        h, w, c = pix.shape
        tex_arr[i] = _Texture(pix.ctypes.data, w, h, c, int(wrap))
    out = C.POINTER(C.c_uint32)()
    n = lib().o2v_oracle_voxelize(_ptr(verts), _ptr(uvs), _ptr(types), _ptr(colors), _ptr(texids), T,
                                  C.cast(tex_arr, C.c_void_p), resolution, supersampling, strategy, _ptr(unit),
                                  _ptr(bnd), zslab[0], zslab[1], C.byref(out))
    if n == 0:
        return np.zeros((0, 4), dtype=np.uint32)
    res = np.ctypeslib.as_array(out, shape=(n, 4)).copy()
    lib().o2v_oracle_free(out)
    return res


def set_threads(n):
    """Chunk-parallel worker threads (default 1). Results are independent of the thread count."""
    lib().o2v_oracle_set_threads(int(n))


def release():
    """Frees the per-thread state (voxelizers, output lists) the harness keeps between calls."""
    lib().o2v_oracle_release()


def phase_seconds():
    """Wall seconds of the last voxelize(): (prelude: copy + bounds + transform + chunk binning, the chunk loop = the
    reference's algorithm, joining the threads' output lists)."""
    out = (C.c_double * 3)()
    lib().o2v_oracle_get_phase_seconds(out)
    return tuple(float(x) for x in out)


def stats():
    s = _Stats()
    lib().o2v_oracle_get_stats(C.byref(s))
    return {n: getattr(s, n) for n, _ in _Stats._fields_}


def mesh_transform(bounds, sample_res, unit_transform=None):
    """Returns float32[12]: row-major 3x3 then translation (obj2voxel.cpp:370-402)."""
    bnd = np.ascontiguousarray(bounds, dtype=np.float32).reshape(6)
    unit = None if unit_transform is None else np.ascontiguousarray(unit_transform, dtype=np.int32).reshape(9)
    out = np.zeros(12, dtype=np.float32)
    lib().o2v_oracle_mesh_transform(_ptr(bnd), sample_res, _ptr(unit), _ptr(out))
    return out
