#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry point: obj2voxel_voxelize() with a triangle callback in and a voxel
callback out (what the reference's CLI and tests call). Includes the callback pulls, H2D of 76 B/triangle, context
creation + first-touch of the dense grid, the device pipeline, D2H of 16 B/voxel and the sink callbacks.
This is NOT bench.py's `value` (which is HBM-resident); it is the number DESIGN.md section 6 asks to be noted.

The triangle callback here is a C function inside a tiny helper library (a Python ctypes callback per triangle would
measure Python, not the library)."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HELPER_SRC = r'''
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#include "obj2voxel.h"
typedef struct { const float *verts; size_t n, i; } feed;
typedef struct { size_t voxels, calls; } count;
bool feed_next(void *d, obj2voxel_triangle *t) {
    feed *f = (feed *) d;
    if (f->i >= f->n) return false;
    obj2voxel_set_triangle_basic(t, f->verts + 9 * f->i++);
    return true;
}
typedef struct { const float *verts, *uvs; obj2voxel_texture *texture; size_t n, i; } feed_tex;
bool feed_next_textured(void *d, obj2voxel_triangle *t) {
    feed_tex *f = (feed_tex *) d;
    if (f->i >= f->n) return false;
    obj2voxel_set_triangle_textured(t, f->verts + 9 * f->i, f->uvs + 6 * f->i, f->texture);
    f->i++;
    return true;
}
bool count_write(void *d, uint32_t *v, size_t n) { count *c = (count *) d; (void) v; c->voxels += n; c->calls++; return true; }
'''


def _helper():
    import obj2voxel_amd
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "helper.c")
    open(src, "w").write(HELPER_SRC)
    so = os.path.join(tmp, "libhelper.so")
    libdir = os.path.dirname(obj2voxel_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", so,
                           "-L", libdir, "-lobj2voxel_amd", "-Wl,-rpath," + libdir])
    return C.CDLL(so)


def measure_published(reps=4, new_session_first=True):
    """The one workload the reference publishes a wall time for (README.adoc:177-178, img/terminal_screenshot.png: 19 392
    textured triangles at r = 8192, MAX, VL32: 20.3 M voxels in 1.82 s end to end on the author's CPU), on its stand-in
    (meshes.readme_blade: 19 320 textured triangles): obj2voxel_voxelize() with a C triangle callback in and the VL32 memory
    sink out.  The first call makes a new device session (the cached one is released first): it allocates the dense grids."""
    import numpy as np
    from obj2voxel_amd import capi, meshes
    import obj2voxel_amd
    a = capi.api()
    helper = _helper()

    class FeedTex(C.Structure):
        _fields_ = [("verts", C.c_void_p), ("uvs", C.c_void_p), ("texture", C.c_void_p), ("n", C.c_size_t), ("i", C.c_size_t)]

    verts, uvs = meshes.readme_blade()
    verts, uvs = np.ascontiguousarray(verts), np.ascontiguousarray(uvs)
    pix = np.ascontiguousarray(meshes.checker_texture(1024, 32))
    tex = a.obj2voxel_texture_alloc()
    assert a.obj2voxel_texture_load_pixels(tex, pix.ctypes.data, 1024, 1024, 3)
    level = a.obj2voxel_get_log_level()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    a.obj2voxel_set_input_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if new_session_first:
        C.CDLL(obj2voxel_amd.LIB_PATH).o2v_release_cached_device_memory()
    times, size = [], C.c_size_t(0)
    for _ in range(reps):
        feed = FeedTex(verts.ctypes.data, uvs.ctypes.data, tex, len(verts), 0)
        inst = a.obj2voxel_alloc()
        a.obj2voxel_set_input_callback(inst, C.cast(helper.feed_next_textured, C.c_void_p), C.byref(feed))
        a.obj2voxel_set_output_memory(inst, b"vl32")
        a.obj2voxel_set_resolution(inst, 8192)
        a.obj2voxel_set_color_strategy(inst, capi.MAX_STRATEGY)
        t0 = time.perf_counter()
        err = a.obj2voxel_voxelize(inst)
        dt = time.perf_counter() - t0
        assert err == 0, err
        size = C.c_size_t(0)
        assert bool(a.obj2voxel_get_output_memory(inst, C.byref(size)))
        a.obj2voxel_free(inst)
        times.append(dt)
    a.obj2voxel_texture_free(tex)
    a.obj2voxel_set_log_level(level)
    return {"triangles": len(verts), "resolution": 8192, "voxels": size.value // 16, "output_bytes": size.value, "wall_s": times}


def measure(nv=467, res=1024, reps=3, debug=False):
    import numpy as np
    from obj2voxel_amd import capi, meshes
    import obj2voxel_amd
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, "helper.c")
    open(src, "w").write(HELPER_SRC)
    so = os.path.join(tmp, "libhelper.so")
    libdir = os.path.dirname(obj2voxel_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", so,
                           "-L", libdir, "-lobj2voxel_amd", "-Wl,-rpath," + libdir])
    a = capi.api()
    helper = C.CDLL(so)

    class Feed(C.Structure):
        _fields_ = [("verts", C.c_void_p), ("n", C.c_size_t), ("i", C.c_size_t)]

    class Count(C.Structure):
        _fields_ = [("voxels", C.c_size_t), ("calls", C.c_size_t)]

    verts = np.ascontiguousarray(meshes.uv_sphere(nv))
    level = a.obj2voxel_get_log_level()
    a.obj2voxel_set_log_level(4 if debug else capi.LOG_SILENT)   # 4: prints the library's per-phase wall times
    a.obj2voxel_set_input_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    a.obj2voxel_set_output_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    times = []
    cnt = Count(0, 0)
    for rep in range(reps):
        if rep and rep == int(os.environ.get("O2V_CAPI_NEW_SESSION_AT", "0")):
            # what a first call costs once the HIP runtime is up: the cached device session goes back to the system
            C.CDLL(obj2voxel_amd.LIB_PATH).o2v_release_cached_device_memory()
        feed = Feed(verts.ctypes.data, len(verts), 0)
        cnt = Count(0, 0)
        inst = a.obj2voxel_alloc()
        a.obj2voxel_set_input_callback(inst, C.cast(helper.feed_next, C.c_void_p), C.byref(feed))
        a.obj2voxel_set_output_callback(inst, C.cast(helper.count_write, C.c_void_p), C.byref(cnt))
        a.obj2voxel_set_resolution(inst, res)
        t0 = time.perf_counter()
        err = a.obj2voxel_voxelize(inst)
        dt = time.perf_counter() - t0
        a.obj2voxel_free(inst)
        assert err == 0, err
        times.append(dt)
    a.obj2voxel_set_log_level(level)
    return {"triangles": len(verts), "resolution": res, "voxels": cnt.voxels, "sink_calls": cnt.calls, "wall_s": times}


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 467
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    r = measure(nv, res, reps, debug=os.environ.get("O2V_CAPI_DEBUG") == "1")
    best = min(r["wall_s"])
    print(json.dumps({"entry_point": "obj2voxel_voxelize (callback in, callback out)", "triangles": r["triangles"],
                      "resolution": res, "voxels": r["voxels"], "sink_calls": r["sink_calls"],
                      "wall_s": [round(t, 4) for t in r["wall_s"]], "best_mvoxels_per_s": round(r["voxels"] / best / 1e6, 2),
                      "best_mtris_per_s": round(r["triangles"] / best / 1e6, 2)}))


if __name__ == "__main__":
    main()
