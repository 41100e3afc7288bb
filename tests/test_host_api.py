"""Host-side logic of the drop-in C API that needs no GPU: exported symbols, precondition error codes
(reference test/main.cpp:68-118), instance/texture plumbing, worker compatibility, sinks on an empty model."""
import ctypes as C
import os
import re
import threading
import time

import numpy as np
import pytest

from obj2voxel_amd import meshes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def a():
    import obj2voxel_amd
    if not os.path.exists(obj2voxel_amd.LIB_PATH):
        obj2voxel_amd.build()
    from obj2voxel_amd import capi
    api = capi.api()
    api.obj2voxel_set_log_level(capi.LOG_SILENT)
    yield api
    api.obj2voxel_set_log_level(capi.LOG_INFO)


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:obj2voxel|o2v)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(a):
    from obj2voxel_amd import capi
    public = _declared_functions("obj2voxel.h")
    assert len(public) == 35 and set(public) == set(capi.SIGNATURES)  # reference include/obj2voxel.h:89-406
    assert "obj2voxel_teture_set_uv_mode" in public  # the reference's spelling (include/obj2voxel.h:350)
    device = _declared_functions("o2v_hip.h")
    assert len(device) >= 12
    for name in public + device:
        assert hasattr(a, name), name


def test_error_on_missing_input(a):  # test/main.cpp:68-83
    from obj2voxel_amd import capi
    out = capi.CountingOutput()
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 1)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_NO_INPUT
    a.obj2voxel_free(inst)


def test_error_on_missing_output(a):  # test/main.cpp:85-100
    from obj2voxel_amd import capi
    inp = capi.TriangleInput(meshes.single_triangle())
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_resolution(inst, 1)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_NO_OUTPUT
    a.obj2voxel_free(inst)


def test_error_on_missing_resolution(a):  # test/main.cpp:102-118
    from obj2voxel_amd import capi
    inp = capi.TriangleInput(meshes.single_triangle())
    out = capi.CountingOutput()
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_NO_RESOLUTION
    a.obj2voxel_free(inst)


def test_settings_round_trip(a):
    inst = a.obj2voxel_alloc()
    assert a.obj2voxel_get_resolution(inst) == 0
    a.obj2voxel_set_resolution(inst, 128)
    a.obj2voxel_set_supersampling(inst, 2)
    assert a.obj2voxel_get_resolution(inst) == 128  # output resolution, not the sample resolution
    assert a.obj2voxel_get_chunk_size(inst) == 64
    a.obj2voxel_free(inst)


def test_empty_model_finalizes_sinks_without_a_gpu(a, tmp_path):
    """obj2voxel.cpp:590-594: a model without triangles writes an empty voxel model and returns OK."""
    from obj2voxel_amd import capi
    for kind in ("memory", "file", "callback"):
        inst = a.obj2voxel_alloc()
        inp = capi.TriangleInput(np.zeros((0, 9), np.float32))
        a.obj2voxel_set_input_callback(inst, inp.callback, None)
        a.obj2voxel_set_resolution(inst, 32)
        path = tmp_path / "empty.vl32"
        out = capi.CountingOutput()
        if kind == "memory":
            a.obj2voxel_set_output_memory(inst, b"ply")
        elif kind == "file":
            path_bytes = str(path).encode()  # borrowed by the API until voxelize
            a.obj2voxel_set_output_file(inst, path_bytes, None)
        else:
            a.obj2voxel_set_output_callback(inst, out.callback, None)
        assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
        assert a.obj2voxel_voxelize(inst) == capi.ERR_DOUBLE_VOXELIZATION
        if kind == "memory":
            size = C.c_size_t(0)
            ptr = a.obj2voxel_get_output_memory(inst, C.byref(size))
            assert bool(ptr) and size.value == 300  # PLY header is exactly 300 bytes (README.adoc:236-237)
            header = bytes(np.ctypeslib.as_array(ptr, shape=(300,)))
            assert header.startswith(b"ply\nformat binary_big_endian 1.0\nelement vertex 0")
            assert header.endswith(b"end_header\n")
        elif kind == "file":
            assert path.exists() and path.stat().st_size == 0
        else:
            size = C.c_size_t(7)
            assert not a.obj2voxel_get_output_memory(inst, C.byref(size)) and size.value == 7
            assert out.voxel_count == 0
        a.obj2voxel_free(inst)


def test_unopenable_files_map_to_error_codes(a, tmp_path):
    from obj2voxel_amd import capi
    inst = a.obj2voxel_alloc()
    missing = str(tmp_path / "missing.stl").encode()
    a.obj2voxel_set_input_file(inst, missing, None)
    a.obj2voxel_set_output_memory(inst, b"vl32")
    a.obj2voxel_set_resolution(inst, 8)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OPEN_INPUT
    a.obj2voxel_free(inst)
    inst = a.obj2voxel_alloc()
    inp = capi.TriangleInput(np.zeros((0, 9), np.float32))
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    bad_out = str(tmp_path / "no_such_dir" / "x.vl32").encode()
    a.obj2voxel_set_output_file(inst, bad_out, None)
    a.obj2voxel_set_resolution(inst, 8)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_OPEN_OUTPUT
    a.obj2voxel_free(inst)


def test_texture_pixels_round_trip(a):
    from obj2voxel_amd import capi
    tex = a.obj2voxel_texture_alloc()
    pix = meshes.checker_texture(16, 4)
    assert a.obj2voxel_texture_load_pixels(tex, pix.ctypes.data, 16, 16, 3)
    w, h, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    a.obj2voxel_texture_get_meta(tex, C.byref(w), C.byref(h), C.byref(c))
    assert (w.value, h.value, c.value) == (16, 16, 3)
    back = np.zeros_like(pix)
    a.obj2voxel_texture_get_pixels(tex, back.ctypes.data)
    assert np.array_equal(back, pix)
    a.obj2voxel_teture_set_uv_mode(tex, capi.UV_CLAMP)
    assert not a.obj2voxel_texture_load_from_memory(tex, pix.ctypes.data, 10, b"png")  # not a PNG
    a.obj2voxel_texture_free(tex)


def test_png_decoder(a):
    import struct
    import zlib
    from obj2voxel_amd import capi  # noqa: F401
    rgb = meshes.checker_texture(8, 2)
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(8))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 2, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    tex = a.obj2voxel_texture_alloc()
    assert a.obj2voxel_texture_load_from_memory(tex, png, len(png), b"png")
    w, h, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    a.obj2voxel_texture_get_meta(tex, C.byref(w), C.byref(h), C.byref(c))
    assert (w.value, h.value, c.value) == (8, 8, 4)
    argb = np.zeros((8, 8, 4), np.uint8)
    a.obj2voxel_texture_get_pixels(tex, argb.ctypes.data)
    assert (argb[..., 0] == 255).all() and np.array_equal(argb[..., 1:], rgb)
    a.obj2voxel_texture_free(tex)


def test_worker_entry_points_keep_their_contract(a):
    """reference obj2voxel.cpp:957-1003: run_worker registers and blocks until stop_workers."""
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_parallel(inst, True)
    threads = [threading.Thread(target=a.obj2voxel_run_worker, args=(inst,)) for _ in range(3)]
    for t in threads:
        t.start()
    deadline = time.time() + 5
    while a.obj2voxel_get_worker_count(inst) != 3 and time.time() < deadline:
        time.sleep(0.01)
    assert a.obj2voxel_get_worker_count(inst) == 3
    a.obj2voxel_stop_workers(inst)
    for t in threads:
        t.join(5)
        assert not t.is_alive()
    assert a.obj2voxel_get_worker_count(inst) == 0
    a.obj2voxel_run_worker(inst)  # returns immediately once workers were stopped
    a.obj2voxel_free(inst)


def test_gpu_path_fails_loudly_without_a_device(a):
    """No CPU fallback: on a box without a GPU the product path reports OBJ2VOXEL_ERR_DEVICE / raises."""
    from obj2voxel_amd import capi, hip
    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    inp = capi.TriangleInput(meshes.unit_cube())
    out = capi.CountingOutput()
    inst = a.obj2voxel_alloc()
    a.obj2voxel_set_input_callback(inst, inp.callback, None)
    a.obj2voxel_set_output_callback(inst, out.callback, None)
    a.obj2voxel_set_resolution(inst, 16)
    assert a.obj2voxel_voxelize(inst) == capi.ERR_DEVICE
    assert out.voxel_count == 0
    a.obj2voxel_free(inst)
    with pytest.raises(hip.DeviceError):
        hip.DeviceVoxelizer(0)


def test_header_declares_the_error_codes_the_library_returns():
    """The reference's codes 0..7 (include/obj2voxel.h:63-79) plus this build's extension OBJ2VOXEL_ERR_DEVICE = 8: a
    caller compiled against the header can name every code obj2voxel_voxelize() returns."""
    from obj2voxel_amd import capi
    text = open(os.path.join(ROOT, "include", "obj2voxel.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"obj2voxel_error_t (OBJ2VOXEL_ERR_[A-Z_]+) = (\d+);", text)}
    assert sorted(codes.values()) == list(range(9))
    assert codes["OBJ2VOXEL_ERR_DEVICE"] == capi.ERR_DEVICE == 8
    assert codes["OBJ2VOXEL_ERR_DOUBLE_VOXELIZATION"] == capi.ERR_DOUBLE_VOXELIZATION == 7


def test_cli_version_and_help_need_no_device():
    """-V / --version (reference src/main.cpp:296-298,320-340), --80 and -h of the command line front end."""
    import subprocess
    import obj2voxel_amd
    cli = os.path.join(os.path.dirname(obj2voxel_amd.LIB_PATH), "obj2voxel-amd")
    if not os.path.exists(cli):
        obj2voxel_amd.build()
    for flag in ("-V", "--version"):
        r = subprocess.run([cli, flag], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "Version:" in r.stdout and "1.3.5" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([cli, "--80", "-h"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--version" in r.stdout
