"""ctypes binding of the drop-in public C API (include/obj2voxel.h), plus Python equivalents of the callback
adapters the reference's tests use (test/testutil.hpp:42-172), so the parity tests read like the reference's."""
import ctypes as C

import numpy as np

from ._lib import lib

ERR_OK, ERR_NO_INPUT, ERR_NO_OUTPUT, ERR_NO_RESOLUTION = 0, 1, 2, 3
ERR_OPEN_INPUT, ERR_OPEN_OUTPUT, ERR_VOXEL_WRITE, ERR_DOUBLE_VOXELIZATION, ERR_DEVICE = 4, 5, 6, 7, 8
MAX_STRATEGY, BLEND_STRATEGY = 0, 1
UV_CLAMP, UV_WRAP = 0, 1
LOG_SILENT, LOG_ERROR, LOG_WARNING, LOG_INFO, LOG_DEBUG = 0, 1, 2, 3, 4

TRIANGLE_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p)
VOXEL_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t)
LOG_CB = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_char_p, C.c_ubyte)

# name -> (restype, argtypes): all 35 entry points of include/obj2voxel.h
SIGNATURES = {
    "obj2voxel_alloc": (C.c_void_p, []),
    "obj2voxel_free": (None, [C.c_void_p]),
    "obj2voxel_set_log_level": (None, [C.c_ubyte]),
    "obj2voxel_set_log_callback": (None, [C.c_void_p, C.c_void_p]),
    "obj2voxel_get_log_level": (C.c_ubyte, []),
    "obj2voxel_set_resolution": (None, [C.c_void_p, C.c_uint32]),
    "obj2voxel_set_supersampling": (None, [C.c_void_p, C.c_uint32]),
    "obj2voxel_set_color_strategy": (None, [C.c_void_p, C.c_ubyte]),
    "obj2voxel_set_texture": (None, [C.c_void_p, C.c_void_p]),
    "obj2voxel_set_input_file": (None, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "obj2voxel_set_input_callback": (None, [C.c_void_p, TRIANGLE_CB, C.c_void_p]),
    "obj2voxel_set_output_file": (None, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "obj2voxel_set_output_memory": (None, [C.c_void_p, C.c_char_p]),
    "obj2voxel_set_output_callback": (None, [C.c_void_p, VOXEL_CB, C.c_void_p]),
    "obj2voxel_set_parallel": (None, [C.c_void_p, C.c_bool]),
    "obj2voxel_set_unit_transform": (None, [C.c_void_p, C.POINTER(C.c_int)]),
    "obj2voxel_set_mesh_boundaries": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "obj2voxel_get_resolution": (C.c_uint32, [C.c_void_p]),
    "obj2voxel_get_chunk_size": (C.c_uint32, [C.c_void_p]),
    "obj2voxel_get_output_memory": (C.POINTER(C.c_ubyte), [C.c_void_p, C.POINTER(C.c_size_t)]),
    "obj2voxel_set_triangle_basic": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "obj2voxel_set_triangle_colored": (None, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "obj2voxel_set_triangle_textured": (None, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    "obj2voxel_texture_alloc": (C.c_void_p, []),
    "obj2voxel_texture_free": (None, [C.c_void_p]),
    "obj2voxel_texture_load_from_file": (C.c_bool, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "obj2voxel_texture_load_from_memory": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]),
    "obj2voxel_texture_load_pixels": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "obj2voxel_teture_set_uv_mode": (None, [C.c_void_p, C.c_ubyte]),
    "obj2voxel_texture_get_meta": (None, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_size_t)]),
    "obj2voxel_texture_get_pixels": (None, [C.c_void_p, C.c_void_p]),
    "obj2voxel_run_worker": (None, [C.c_void_p]),
    "obj2voxel_stop_workers": (None, [C.c_void_p]),
    "obj2voxel_get_worker_count": (C.c_uint32, [C.c_void_p]),
    "obj2voxel_voxelize": (C.c_ubyte, [C.c_void_p]),
}

_bound = None


def api():
    global _bound
    if _bound is None:
        L = lib()
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _bound = L
    return _bound


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class TriangleInput:
    """Feeds [T, 9] vertices (optionally uvs + texture handle, or colours) through the triangle callback
    (reference test/testutil.hpp:42-66)."""

    def __init__(self, verts, uvs=None, texture=None, colors=None):
        self.verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 6)
        self.colors = None if colors is None else np.ascontiguousarray(colors, dtype=np.float32).reshape(-1, 3)
        self.texture = texture
        self.index = 0
        self.callback = TRIANGLE_CB(self._next)

    def _next(self, _data, tri):
        if self.index >= len(self.verts):
            return False
        i = self.index
        self.index += 1
        a = api()
        if self.uvs is not None and self.texture is not None:
            a.obj2voxel_set_triangle_textured(tri, _fptr(self.verts[i]), _fptr(self.uvs[i]), self.texture)
        elif self.colors is not None:
            a.obj2voxel_set_triangle_colored(tri, _fptr(self.verts[i]), _fptr(self.colors[i]))
        else:
            a.obj2voxel_set_triangle_basic(tri, _fptr(self.verts[i]))
        return True


class CountingOutput:
    """reference test/testutil.hpp:123-131"""

    def __init__(self, fail_after=None):
        self.voxel_count = 0
        self.calls = 0
        self.fail_after = fail_after
        self.callback = VOXEL_CB(self._write)

    def _write(self, _data, _voxels, count):
        self.calls += 1
        self.voxel_count += count
        return not (self.fail_after is not None and self.calls > self.fail_after)


class CollectingOutput:
    """Keeps every (x, y, z, argb) record."""

    def __init__(self):
        self.chunks = []
        self.callback = VOXEL_CB(self._write)

    def _write(self, _data, voxels, count):
        self.chunks.append(np.ctypeslib.as_array(voxels, shape=(count, 4)).copy())
        return True

    def voxels(self):
        return np.concatenate(self.chunks) if self.chunks else np.zeros((0, 4), np.uint32)
