"""The reference's own API-level tests (test/main.cpp:128-252), restated against the drop-in C API."""
import ctypes as C

import numpy as np
import pytest

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu


def _expected_unit_cube(res):
    return 8 + 12 * (res - 2) + 6 * (res - 2) * (res - 2)


def _instance(a, verts):
    from obj2voxel_amd import capi
    inst = a.obj2voxel_alloc()
    inp = capi.TriangleInput(verts)
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    return inst, inp


def test_unit_cube_produces_expected_voxel_count():  # test/main.cpp:128-156
    from obj2voxel_amd import capi
    a = capi.api()
    inst, inp = _instance(a, meshes.unit_cube())
    out = capi.CountingOutput()
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 64)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    assert out.voxel_count == _expected_unit_cube(64) == 23816


def test_unit_cube_produces_expected_byte_count():  # test/main.cpp:158-179
    from obj2voxel_amd import capi
    a = capi.api()
    inst, inp = _instance(a, meshes.unit_cube())
    a.obj2voxel_set_output_memory(inst, b"vl32")
    a.obj2voxel_set_resolution(inst, 64)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    size = C.c_size_t(0)
    ptr = a.obj2voxel_get_output_memory(inst, C.byref(size))
    assert bool(ptr)
    assert size.value == _expected_unit_cube(64) * 16
    raw = np.ctypeslib.as_array(ptr, shape=(size.value,)).copy()
    a.obj2voxel_free(inst)
    rec = raw.view(">u4").reshape(-1, 4)  # VL32: big-endian x, y, z, argb (README.adoc:233-252)
    assert rec[:, :3].max() == 63 and (rec[:, 3] == 0xFFFFFFFF).all()


def test_unit_cube_multiple_chunks():  # test/main.cpp:194-208
    from obj2voxel_amd import capi
    a = capi.api()
    inst, inp = _instance(a, meshes.unit_cube())
    res = a.obj2voxel_get_chunk_size(inst) * 2
    a.obj2voxel_set_resolution(inst, res)
    assert a.obj2voxel_get_resolution(inst) == res == 128
    out = capi.CountingOutput()
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    assert out.voxel_count == 96776


@pytest.mark.parametrize("res,expected", [(32, 3072), (128, 49152)])  # test/main.cpp:225-252
def test_three_planes(res, expected):
    from obj2voxel_amd import capi
    a = capi.api()
    inst, inp = _instance(a, meshes.three_planes())
    out = capi.CountingOutput()
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, res)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    assert out.voxel_count == expected == 3 * res * res


def test_double_voxelization_and_sink_failure():
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    inst, inp = _instance(a, meshes.unit_cube())
    out = capi.CountingOutput(fail_after=0)  # sink reports failure on its first write
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 32)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_VOXEL_WRITE  # obj2voxel.cpp:509-512
    assert a.obj2voxel_voxelize(inst) == capi.ERR_DOUBLE_VOXELIZATION  # obj2voxel.cpp:604-606
    a.obj2voxel_free(inst)
    a.obj2voxel_set_log_level(capi.LOG_INFO)


def test_textured_callback_path_matches_oracle(oracle):
    from obj2voxel_amd import capi
    a = capi.api()
    v, uv = meshes.uv_sphere(10, with_uv=True)
    pix = meshes.checker_texture(64, 8)
    tex = a.obj2voxel_texture_alloc()
    assert a.obj2voxel_texture_load_pixels(tex, pix.ctypes.data, 64, 64, 3)
    inst = a.obj2voxel_alloc()
    inp = capi.TriangleInput(v, uvs=uv, texture=tex)
    out = capi.CollectingOutput()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 96)
    a.obj2voxel_set_color_strategy(inst, capi.BLEND_STRATEGY)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    a.obj2voxel_texture_free(tex)
    T = len(v)
    want = oracle.voxelize(v, 96, uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32),
                           textures=[(pix, 1)], strategy=1)
    assert np.array_equal(meshes.sorted_voxels(out.voxels()), meshes.sorted_voxels(want))


def test_log_callback_receives_messages():
    """reference obj2voxel.cpp:664-677: a log callback that returns true consumes the message."""
    from obj2voxel_amd import capi
    a = capi.api()
    seen = []
    cb = capi.LOG_CB(lambda _d, msg, level: (seen.append((level, msg.decode())), True)[1])
    a.obj2voxel_set_log_callback.argtypes = [capi.LOG_CB, C.c_void_p]
    a.obj2voxel_set_log_callback(cb, None)
    a.obj2voxel_set_log_level(capi.LOG_DEBUG)
    try:
        inst, inp = _instance(a, meshes.unit_cube())
        out = capi.CountingOutput()
        a.obj2voxel_set_output_callback(inst, out.callback, None)
        a.obj2voxel_set_resolution(inst, 16)
        assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
        a.obj2voxel_free(inst)
    finally:
        a.obj2voxel_set_log_callback.argtypes = [C.c_void_p, C.c_void_p]
        a.obj2voxel_set_log_callback(None, None)
        a.obj2voxel_set_log_level(capi.LOG_INFO)
    assert a.obj2voxel_get_log_level() == capi.LOG_INFO
    assert any("Cached model with 12 triangles" in m for _, m in seen)
    assert any(level == capi.LOG_DEBUG for level, _ in seen)


@pytest.mark.parametrize("res,ss,textured", [(100_000, 1, False), (40_000, 2, True)])
def test_sample_resolution_above_65535_in_xy_tiles(oracle, res, ss, textured):
    """The reference takes any uint32 resolution (include/obj2voxel.h:130-138: u32 coordinates, 64-bit Morton keys,
    src/util.hpp:185-196).  Here voxel coordinates travel in 16-bit fields relative to a pass' box, so obj2voxel_voxelize() cuts a
    grid of more than 65 535 samples per axis into x / y tiles (as it cuts one too thick for the memory into z-slabs).  A thin
    ribbon along the diagonal of the z ~ 0 plane - it crosses every x and y, the mesh's box is a few layers thick - at a
    sample resolution of 100 000 (2 x 2 tiles, occupancy only) and of 2 x 40 000 with a texture and BLEND (the weighted route,
    supersampled): every record equals the oracle's."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_ERROR)
    v = meshes.diagonal_strip(400, width=6e-5)
    T = len(v)
    uv = np.ascontiguousarray((v.reshape(T, 3, 3)[:, :, :2] * 37.0).reshape(T, 6), dtype=np.float32)
    pix = meshes.checker_texture(64, 8)
    tex = None
    try:
        inst = a.obj2voxel_alloc()
        if textured:
            tex = a.obj2voxel_texture_alloc()
            assert a.obj2voxel_texture_load_pixels(tex, pix.ctypes.data, 64, 64, 3)
            inp = capi.TriangleInput(v, uvs=uv, texture=tex)
        else:
            inp = capi.TriangleInput(v)
        out = capi.CollectingOutput()
        a.obj2voxel_set_input_callback(inst, inp.callback, None)
        a.obj2voxel_set_output_callback(inst, out.callback, None)
        a.obj2voxel_set_resolution(inst, res)
        a.obj2voxel_set_supersampling(inst, ss)
        a.obj2voxel_set_color_strategy(inst, capi.BLEND_STRATEGY if textured else capi.MAX_STRATEGY)
        assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
        a.obj2voxel_free(inst)
    finally:
        if tex:
            a.obj2voxel_texture_free(tex)
        a.obj2voxel_set_log_level(capi.LOG_INFO)
        # (a 50 GB grid: give it back before the next test)
        C.CDLL(__import__("obj2voxel_amd").LIB_PATH).o2v_release_cached_device_memory()
    kw = dict(uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32), textures=[(pix, 1)], strategy=1) if textured else {}
    want = oracle.voxelize(v, res, supersampling=ss, **kw)
    got = out.voxels()
    assert len(got) == len(want) > 50_000
    assert int(got[:, 0].max()) > 0.99 * res and int(got[:, 1].max()) > 0.99 * res     # (beyond the first tile)
    assert np.array_equal(meshes.sorted_voxels(got), meshes.sorted_voxels(want))
    if textured:
        assert len(np.unique(got[:, 3])) > 2


@pytest.mark.parametrize("mode", [0, 1])
def test_argb_texture_and_uv_mode_through_the_api(oracle, mode):
    """4-channel textures are ARGB (include/obj2voxel.h:317-320); obj2voxel_teture_set_uv_mode picks clamp / wrap."""
    from obj2voxel_amd import capi
    a = capi.api()
    v, uv = meshes.uv_sphere(9, with_uv=True)
    uv = uv * 2.2 - 0.6
    rgb = meshes.checker_texture(32, 4)
    argb = np.ascontiguousarray(np.concatenate([np.full((32, 32, 1), 77, np.uint8), rgb], axis=2))
    tex = a.obj2voxel_texture_alloc()
    assert a.obj2voxel_texture_load_pixels(tex, argb.ctypes.data, 32, 32, 4)
    a.obj2voxel_teture_set_uv_mode(tex, mode)
    inst = a.obj2voxel_alloc()
    inp = capi.TriangleInput(v, uvs=uv, texture=tex)
    out = capi.CollectingOutput()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 72)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    a.obj2voxel_texture_free(tex)
    T = len(v)
    want = oracle.voxelize(v, 72, uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32),
                           textures=[(argb, mode)])
    assert np.array_equal(meshes.sorted_voxels(out.voxels()), meshes.sorted_voxels(want))


def test_colored_triangles_render_white_like_the_reference():
    """obj2voxel_set_triangle_colored stores the colour but leaves the type MATERIALLESS (obj2voxel.cpp:828-837)."""
    from obj2voxel_amd import capi
    a = capi.api()
    v = meshes.uv_sphere(6)
    inst = a.obj2voxel_alloc()
    inp = capi.TriangleInput(v, colors=meshes.triangle_colors(len(v)))
    out = capi.CollectingOutput()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 40)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    assert (out.voxels()[:, 3] == 0xFFFFFFFF).all()


def test_concurrent_instances_and_worker_pool(oracle):
    """Two host threads voxelize different meshes at once (the process-wide device context serves one of them, the other
    gets a temporary context), one of them in the reference CLI's worker-pool mode (src/main.cpp:149-194: workers parked
    in obj2voxel_run_worker, set_parallel(true), stop_workers afterwards). Both results equal the oracle's."""
    import threading
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    meshes_in = [meshes.uv_sphere(20), meshes.uv_sphere(14, radius=0.7, center=(0.2, 0.1, 0.0))]
    resolutions = [112, 96]
    results, errors = [None, None], [None, None]

    def job(k):
        inst, inp = _instance(a, meshes_in[k])
        out = capi.CollectingOutput()
        a.obj2voxel_set_output_callback(inst, out.callback, None)
        a.obj2voxel_set_resolution(inst, resolutions[k])
        workers = []
        if k == 0:
            a.obj2voxel_set_parallel(inst, True)
            workers = [threading.Thread(target=a.obj2voxel_run_worker, args=(inst,)) for _ in range(2)]
            for w in workers:
                w.start()
        errors[k] = a.obj2voxel_voxelize(inst)
        if k == 0:
            a.obj2voxel_stop_workers(inst)
            for w in workers:
                w.join(timeout=30)
            assert not any(w.is_alive() for w in workers)
        a.obj2voxel_free(inst)
        results[k] = out.voxels()

    for _ in range(3):   # a few rounds: the cache hand-over happens in either order
        threads = [threading.Thread(target=job, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert errors == [capi.ERR_OK, capi.ERR_OK]
        for k in range(2):
            want = oracle.voxelize(meshes_in[k], resolutions[k])
            assert np.array_equal(meshes.sorted_voxels(results[k]), meshes.sorted_voxels(want))
    a.obj2voxel_set_log_level(capi.LOG_INFO)


def test_streamed_upload_optional_arrays_appear_late(oracle):
    """obj2voxel_voxelize() streams the triangle source into the device's staging blocks (131 072 triangles each): a mesh
    of more than one block whose first textured triangle arrives in the second block - uvs, texture ids and types then
    exist on the device only from that commit on and the triangles before it get the defaults."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    plain = meshes.uv_sphere(190)                                   # 143 640 triangles: more than one block
    tv, tuv = meshes.uv_sphere(40, radius=0.5, center=(0.2, 0.1, -0.3), with_uv=True)
    tex_pixels = meshes.checker_texture(64, 8)
    tex = a.obj2voxel_texture_alloc()
    assert a.obj2voxel_texture_load_pixels(tex, tex_pixels.ctypes.data_as(C.POINTER(C.c_ubyte)), 64, 64, 3)

    class Mixed(capi.TriangleInput):
        def __init__(self):
            super().__init__(np.concatenate([plain, tv]))
            self.n_plain = len(plain)

        def _next(self, _data, tri):
            if self.index >= len(self.verts):
                return False
            i = self.index
            self.index += 1
            v = self.verts[i].ctypes.data_as(C.POINTER(C.c_float))
            if i < self.n_plain:
                a.obj2voxel_set_triangle_basic(tri, v)
            else:
                a.obj2voxel_set_triangle_textured(tri, v, tuv[i - self.n_plain].ctypes.data_as(C.POINTER(C.c_float)), tex)
            return True

    inp, out = Mixed(), capi.CollectingOutput()
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 200)
    a.obj2voxel_set_color_strategy(inst, 1)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    a.obj2voxel_texture_free(tex)
    verts = np.concatenate([plain, tv])
    T, n0 = len(verts), len(plain)
    uvs = np.concatenate([np.zeros((n0, 6), np.float32), tuv])
    types = np.concatenate([np.full(n0, 1, np.uint32), np.full(len(tv), 3, np.uint32)])
    want = oracle.voxelize(verts, 200, uvs=uvs, types=types, texids=np.zeros(T, np.int32), textures=[(tex_pixels, 1)], strategy=1)
    assert np.array_equal(meshes.sorted_voxels(out.voxels()), meshes.sorted_voxels(want))


def _voxelize_collect(a, verts, res, bounds=None, strategy=0):
    from obj2voxel_amd import capi
    inst, inp = _instance(a, verts)
    out = capi.CollectingOutput()
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, res)
    a.obj2voxel_set_color_strategy(inst, strategy)
    if bounds is not None:
        arr = (C.c_float * 6)(*bounds)
        a.obj2voxel_set_mesh_boundaries(inst, arr)
    err = a.obj2voxel_voxelize(inst)
    a.obj2voxel_free(inst)
    assert err == capi.ERR_OK
    return meshes.sorted_voxels(out.voxels())


def test_grid_voxelized_as_consecutive_slabs_when_it_does_not_fit(oracle, monkeypatch):
    """obj2voxel_voxelize() runs the grid as z-slabs when its dense grids exceed the device memory (here forced: slabs of 8
    layers); records, counts and the sink protocol are those of the single pass."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    v = meshes.uv_sphere(24)
    whole = _voxelize_collect(a, v, 72)
    monkeypatch.setenv("O2V_TEST_SLAB_LAYERS", "8")
    for strategy in (0, 1):
        slabbed = _voxelize_collect(a, v, 72, strategy=strategy)
        assert np.array_equal(slabbed[:, :3], whole[:, :3])
        assert np.array_equal(slabbed, meshes.sorted_voxels(oracle.voxelize(v, 72, strategy=strategy)))
    a.obj2voxel_set_log_level(capi.LOG_INFO)


def test_resolution_16384_runs_in_slabs_on_one_gpu(oracle):
    """A grid far beyond what fits densely (16384^3 cells x 12 bytes = 53 TB): the library picks the slab thickness from the
    free device memory.  The mesh sits in the far corner of user bounds that make the mesh transform x -> x + 0.5, so that
    the oracle's chunk maps stay small; compared bit for bit, occupancy and colours."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    S = 16384
    rng = np.random.default_rng(77)
    c = np.array([S - 130.0, S - 140.0, S - 120.0]) + 100.0 * rng.random((3000, 1, 3))
    v = (c + np.exp(rng.uniform(np.log(0.5), np.log(30.0), size=(3000, 1, 1))) * (rng.random((3000, 3, 3)) - 0.5))
    v = np.clip(v, 0.0, S - 1.0).astype(np.float32).reshape(-1, 9)
    bounds = meshes.stress_bounds(S)
    got = _voxelize_collect(a, v, S, bounds=bounds, strategy=1)
    a.obj2voxel_set_log_level(capi.LOG_INFO)
    want = meshes.sorted_voxels(oracle.voxelize(v, S, bounds=bounds, strategy=1))
    assert len(want) > 50_000 and got[:, :3].max() > 16300
    assert got.shape == want.shape and np.array_equal(got, want)


def test_callback_that_sets_nothing_repeats_the_previous_triangle(oracle):
    """The object handed to the triangle callback is reused (reference obj2voxel.cpp:585: `CachedTriangle triangle{}` outside
    the loop), so a callback that returns true without calling a setter caches the previous triangle again - also when the
    setters write straight into the upload staging block."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    v = meshes.uv_sphere(12)
    state = {"i": 0, "calls": 0}

    def feed(_data, tri):
        state["calls"] += 1
        if state["calls"] % 3 == 0:
            return True                     # nothing set: the previous triangle once more
        if state["i"] >= len(v):
            return False
        a.obj2voxel_set_triangle_basic(tri, capi._fptr(v[state["i"]]))
        state["i"] += 1
        return True
    cb = capi.TRIANGLE_CB(feed)
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_input_callback(inst, cb, None)
    out = capi.CollectingOutput()
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 64)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
    a.obj2voxel_free(inst)
    a.obj2voxel_set_log_level(capi.LOG_INFO)
    assert state["calls"] > len(v) * 1.4
    # duplicates of material-less triangles change nothing: the sphere's voxels
    assert np.array_equal(meshes.sorted_voxels(out.voxels()), meshes.sorted_voxels(oracle.voxelize(v, 64)))
