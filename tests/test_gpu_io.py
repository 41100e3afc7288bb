"""The steps either side of the hot path: file inputs (binary STL, OBJ + MTL +
PNG) and list-format outputs (VL32, PLY, XYZRGB; file and memory), end to end through the public C API, checked
against the oracle fed with the same triangles. Formats: reference README.adoc:210-264."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu


def _png_rgb(rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def _run_files(a, in_path, out_spec, res, strategy=0):
    """out_spec: ('file', path) or ('memory', type). Returns (error code, bytes)."""
    inst = a.obj2voxel_alloc()
    # the API borrows the path strings until voxelize (reference obj2voxel.cpp:714-720): keep them alive
    in_bytes, out_bytes = str(in_path).encode(), str(out_spec[1]).encode()
    a.obj2voxel_set_input_file(inst, in_bytes, None)
    if out_spec[0] == "file":
        a.obj2voxel_set_output_file(inst, out_bytes, None)
    else:
        a.obj2voxel_set_output_memory(inst, out_bytes)
    a.obj2voxel_set_resolution(inst, res)
    a.obj2voxel_set_color_strategy(inst, strategy)
    err = a.obj2voxel_voxelize(inst)
    data = b""
    if out_spec[0] == "memory" and err == 0:
        size = C.c_size_t(0)
        ptr = a.obj2voxel_get_output_memory(inst, C.byref(size))
        data = bytes(np.ctypeslib.as_array(ptr, shape=(size.value,))) if size.value else b""
    a.obj2voxel_free(inst)
    if out_spec[0] == "file" and err == 0:
        data = open(out_spec[1], "rb").read()
    return err, data


def _vl32_to_voxels(data):
    return np.frombuffer(data, dtype=">u4").astype(np.uint32).reshape(-1, 4)


def test_binary_stl_to_vl32_ply_xyzrgb(tmp_path, oracle):
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    v = meshes.uv_sphere(9)
    stl = tmp_path / "sphere.stl"
    with open(stl, "wb") as f:
        f.write(b"binary stl".ljust(80, b" ") + struct.pack("<I", len(v)))
        for t in v:
            f.write(struct.pack("<12fH", 0, 0, 0, *t.tolist(), 0))
    want = meshes.sorted_voxels(oracle.voxelize(v, 72))

    err, data = _run_files(a, stl, ("file", tmp_path / "out.vl32"), 72)
    assert err == capi.ERR_OK and len(data) == 16 * len(want)
    assert np.array_equal(meshes.sorted_voxels(_vl32_to_voxels(data)), want)

    err, data = _run_files(a, stl, ("memory", "ply"), 72)
    assert err == capi.ERR_OK and len(data) == 300 + 16 * len(want)   # README.adoc:236-237
    header = data[:300].decode()
    assert header.startswith("ply\nformat binary_big_endian 1.0\nelement vertex ") and header.endswith("end_header\n")
    assert int(header.split("element vertex ")[1].split("\n")[0]) == len(want)
    assert np.array_equal(meshes.sorted_voxels(_vl32_to_voxels(data[300:])), want)

    err, data = _run_files(a, stl, ("file", tmp_path / "out.xyzrgb"), 72)
    assert err == capi.ERR_OK
    rows = np.array([[int(t) for t in line.split()] for line in data.decode().splitlines()], dtype=np.uint32)
    argb = 0xFF000000 | (rows[:, 3] << 16) | (rows[:, 4] << 8) | rows[:, 5]
    got = np.concatenate([rows[:, :3], argb[:, None]], axis=1).astype(np.uint32)
    assert np.array_equal(meshes.sorted_voxels(got), want)

    a.obj2voxel_set_log_level(capi.LOG_INFO)


def _parse_qef(data):
    lines = data.decode().splitlines()
    assert lines[:3] == ["Qubicle Exchange Format", "Version 0.2", "www.minddesk.com"]
    size = [int(t) for t in lines[3].split()]
    n_col = int(lines[4])
    colors = np.array([[float(t) for t in ln.split()] for ln in lines[5:5 + n_col]])
    rows = np.array([[int(t) for t in ln.split()] for ln in lines[5 + n_col:]], dtype=np.int64).reshape(-1, 5)
    return size, colors, rows


def _parse_vox(data):
    """MagicaVoxel .vox (version 150): returns (models [(size, xyzi uint8 [n, 4])], translations, palette [256, 4])."""
    assert data[:4] == b"VOX " and struct.unpack("<I", data[4:8])[0] == 150
    assert data[8:12] == b"MAIN"
    n_content, n_children = struct.unpack("<II", data[12:20])
    assert n_content == 0 and 20 + n_children == len(data)
    pos, sizes, models, trans, palette = 20, [], [], {}, None

    def rd_str(b, o):
        n = struct.unpack("<I", b[o:o + 4])[0]
        return b[o + 4:o + 4 + n].decode(), o + 4 + n

    def rd_dict(b, o):
        n = struct.unpack("<I", b[o:o + 4])[0]
        o += 4
        d = {}
        for _ in range(n):
            k, o = rd_str(b, o)
            v, o = rd_str(b, o)
            d[k] = v
        return d, o
    while pos < len(data):
        cid = data[pos:pos + 4]
        nc, nch = struct.unpack("<II", data[pos + 4:pos + 12])
        body = data[pos + 12:pos + 12 + nc]
        assert nch == 0
        pos += 12 + nc
        if cid == b"SIZE":
            sizes.append(struct.unpack("<III", body))
        elif cid == b"XYZI":
            n = struct.unpack("<I", body[:4])[0]
            assert len(body) == 4 + 4 * n
            models.append((sizes[-1], np.frombuffer(body[4:], np.uint8).reshape(-1, 4)))
        elif cid == b"nTRN":
            node = struct.unpack("<I", body[:4])[0]
            _, o = rd_dict(body, 4)
            child, _res, _layer, frames = struct.unpack("<IiiI", body[o:o + 16])
            fr, o = rd_dict(body, o + 16)
            assert frames == 1 and o == len(body)
            if "_t" in fr:
                trans[child] = [int(t) for t in fr["_t"].split()]
        elif cid == b"nSHP":
            node = struct.unpack("<I", body[:4])[0]
            _, o = rd_dict(body, 4)
            n_models, model_id = struct.unpack("<II", body[o:o + 8])
            assert n_models == 1
            trans[("model", model_id)] = trans.pop(node)
        elif cid == b"RGBA":
            palette = np.frombuffer(body, np.uint8).reshape(256, 4)
        else:
            assert cid == b"nGRP", cid
    return models, trans, palette


def test_palette_formats_qef_and_vox(tmp_path, oracle):
    """QEF (text, unlimited colour table, face-visibility mask) and MagicaVoxel VOX
    (binary chunks, <= 255 colours, 256^3 models placed by a scene graph). The files are parsed back and compared with
    the oracle's voxels: positions exactly; colours exactly when they fit the table, else within the quantiser's error."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    v, uv = meshes.uv_sphere(10, with_uv=True)
    T = len(v)
    (tmp_path / "tex.png").write_bytes(_png_rgb(meshes.checker_texture(64, 8)))
    (tmp_path / "m.mtl").write_text("newmtl few\nKd 0.5 0.25 1\nnewmtl many\nKd 1 1 1\nmap_Kd tex.png\n")

    def write_obj(path, material):
        lines = ["mtllib m.mtl", "usemtl " + material]
        for t in range(T):
            for k in range(3):
                lines.append("v %r %r %r" % tuple(float(x) for x in v[t, k * 3:k * 3 + 3]))
                lines.append("vt %r %r" % tuple(float(x) for x in uv[t, k * 2:k * 2 + 2]))
            i = 3 * t + 1
            lines.append(f"f {i}/{i} {i + 1}/{i + 1} {i + 2}/{i + 2}")
        path.write_text("\n".join(lines) + "\n")
    write_obj(tmp_path / "few.obj", "few")
    write_obj(tmp_path / "many.obj", "many")
    res = 40
    few = meshes.sorted_voxels(oracle.voxelize(v, res, types=np.full(T, 2, np.uint32),
                                               colors=np.tile(np.array([0.5, 0.25, 1.0], np.float32), (T, 1))))

    # QEF, one colour
    err, data = _run_files(a, tmp_path / "few.obj", ("memory", "qef"), res)
    assert err == capi.ERR_OK
    size, colors, rows = _parse_qef(data)
    assert size == [res, res, res] and len(colors) == 1 and len(rows) == len(few)
    assert np.allclose(colors[0] * 255, [(few[0, 3] >> 16) & 255, (few[0, 3] >> 8) & 255, few[0, 3] & 255], atol=1e-3)
    got = np.concatenate([rows[:, :3], np.full((len(rows), 1), few[0, 3])], axis=1).astype(np.uint32)
    assert np.array_equal(meshes.sorted_voxels(got), few)
    filled = {tuple(p) for p in rows[:, :3].tolist()}
    for x, y, z, _, mask in rows[::37].tolist():          # face mask: a face is visible iff the neighbour is empty
        want = sum(bit for bit, d in ((2, (-1, 0, 0)), (4, (1, 0, 0)), (8, (0, 1, 0)), (16, (0, -1, 0)), (32, (0, 0, 1)),
                                      (64, (0, 0, -1))) if (x + d[0], y + d[1], z + d[2]) not in filled)
        assert mask == want

    # VOX, one colour, single model
    err, data = _run_files(a, tmp_path / "few.obj", ("file", tmp_path / "few.vox"), res)
    assert err == capi.ERR_OK
    models, trans, palette = _parse_vox(data)
    assert len(models) == 1 and models[0][0] == (res, res, res) and not trans
    xyzi = models[0][1]
    assert (xyzi[:, 3] == 1).all()
    assert palette[0].tolist() == [(few[0, 3] >> 16) & 255, (few[0, 3] >> 8) & 255, few[0, 3] & 255, 255]
    got = np.concatenate([xyzi[:, :3].astype(np.uint32), np.full((len(xyzi), 1), few[0, 3], np.uint32)], axis=1)
    assert np.array_equal(meshes.sorted_voxels(got), few)

    # VOX at 300^3: several 256^3 models placed by the scene graph; a texture with 2 colours blended at cell borders
    res = 300
    many = meshes.sorted_voxels(oracle.voxelize(v, res, uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32),
                                                textures=[(meshes.checker_texture(64, 8), 1)], strategy=1))
    err, data = _run_files(a, tmp_path / "many.obj", ("memory", "vox"), res, strategy=1)
    assert err == capi.ERR_OK
    models, trans, palette = _parse_vox(data)
    assert len(models) == 4            # the sphere's surface misses the blocks beyond 256 on two or three axes
    parts = []
    n_colors = len(np.unique(many[:, 3]))
    for k, (size, xyzi) in enumerate(models):
        t = np.array(trans[("model", k)])
        origin = t - np.array(size) // 2
        assert set(origin.tolist()) <= {0, 256} and all(s in (256, 44) for s in size)
        assert (xyzi[:, 3] >= 1).all()
        rgba = palette[xyzi[:, 3].astype(int) - 1].astype(np.uint32)
        argb = (rgba[:, 3] << 24) | (rgba[:, 0] << 16) | (rgba[:, 1] << 8) | rgba[:, 2]
        parts.append(np.concatenate([xyzi[:, :3].astype(np.uint32) + origin.astype(np.uint32), argb[:, None]], axis=1))
    got = meshes.sorted_voxels(np.concatenate(parts).astype(np.uint32))
    assert np.array_equal(got[:, :3], many[:, :3])
    ga = (got[:, 3:4] >> np.array([16, 8, 0], np.uint32)) & 255
    wa = (many[:, 3:4] >> np.array([16, 8, 0], np.uint32)) & 255
    if n_colors <= 255:
        assert np.array_equal(got[:, 3], many[:, 3])
    else:
        assert np.abs(ga.astype(int) - wa.astype(int)).mean() < 4.0     # median cut to 255 colours

    # QEF keeps every colour
    err, data = _run_files(a, tmp_path / "many.obj", ("memory", "qef"), 64, strategy=1)
    assert err == capi.ERR_OK
    want64 = meshes.sorted_voxels(oracle.voxelize(v, 64, uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32),
                                                  textures=[(meshes.checker_texture(64, 8), 1)], strategy=1))
    size, colors, rows = _parse_qef(data)
    rgb = np.rint(colors[rows[:, 3]] * 255).astype(np.uint32)
    argb = 0xFF000000 | (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]
    got = np.concatenate([rows[:, :3].astype(np.uint32), argb[:, None].astype(np.uint32)], axis=1)
    assert np.array_equal(meshes.sorted_voxels(got), want64)
    assert len(colors) == len(np.unique(want64[:, 3]))
    a.obj2voxel_set_log_level(capi.LOG_INFO)


def test_obj_with_materials_and_png_texture(tmp_path, oracle):
    """OBJ subset the reference consumes through tinyobjloader (src/io.cpp:244-312): v / vt / f, usemtl with Kd
    (UNTEXTURED) or map_Kd (TEXTURED), faces without material (MATERIALLESS), polygon fan triangulation."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    v, uv = meshes.uv_sphere(8, with_uv=True)
    T = len(v)
    tex = meshes.checker_texture(32, 4)
    (tmp_path / "tex.png").write_bytes(_png_rgb(tex))
    (tmp_path / "m.mtl").write_text("newmtl red\nKd 0.8 0.25 0.125\nnewmtl checker\nKd 1 1 1\nmap_Kd tex.png\n")
    kind = np.arange(T) % 3          # 0: no material, 1: red, 2: textured
    lines = ["mtllib m.mtl"]
    for t in range(T):
        for k in range(3):
            lines.append("v %r %r %r" % tuple(float(x) for x in v[t, k * 3:k * 3 + 3]))
            lines.append("vt %r %r" % tuple(float(x) for x in uv[t, k * 2:k * 2 + 2]))
    cur = None
    for t in range(T):
        want_m = [None, "red", "checker"][kind[t]]
        if want_m != cur and want_m is not None:
            lines.append("usemtl " + want_m)
            cur = want_m
        i = 3 * t + 1
        if kind[t] == 0:
            continue  # material-less faces must come before the first usemtl; emitted below
        lines.append(f"f {i}/{i} {i + 1}/{i + 1} {i + 2}/{i + 2}" if kind[t] == 2 else f"f {i} {i + 1} {i + 2}")
    # faces without a material first (OBJ has no "unset material" statement)
    head = [f"f {3 * t + 1} {3 * t + 2} {3 * t + 3}" for t in range(T) if kind[t] == 0]
    obj = tmp_path / "mesh.obj"
    obj.write_text("\n".join(lines[:1 + 6 * T] + head + lines[1 + 6 * T:]) + "\n")

    # the same triangle sequence for the oracle: material-less first, then in file order
    order = [t for t in range(T) if kind[t] == 0] + [t for t in range(T) if kind[t] != 0]
    types = np.array([[1, 2, 3][kind[t]] for t in order], np.uint32)
    colors = np.tile(np.array([0.8, 0.25, 0.125], np.float32), (T, 1))
    uvs = np.array([uv[t] if kind[t] == 2 else np.zeros(6, np.float32) for t in order], np.float32)
    argb = np.concatenate([np.full((32, 32, 1), 255, np.uint8), tex], axis=2)
    want = oracle.voxelize(v[order], 80, uvs=uvs, types=types, colors=colors, texids=np.zeros(T, np.int32),
                           textures=[(argb, 1)], strategy=1)
    err, data = _run_files(a, obj, ("memory", "vl32"), 80, strategy=1)
    a.obj2voxel_set_log_level(capi.LOG_INFO)
    assert err == capi.ERR_OK
    assert np.array_equal(meshes.sorted_voxels(_vl32_to_voxels(data)), meshes.sorted_voxels(want))


def test_cli_front_end(tmp_path, oracle):
    """The command line front end (reference src/main.cpp:115-202 call sequence, :224-262 axis permutation) on a
    binary STL: -r, -s blend, -p zYx (axes permuted, y flipped), -u, -j 2; output compared with the oracle."""
    import os
    import subprocess
    import obj2voxel_amd
    cli = os.path.join(os.path.dirname(obj2voxel_amd.LIB_PATH), "obj2voxel-amd")
    assert os.path.exists(cli), "build with __graft_entry__.build()"
    v = meshes.uv_sphere(8, radius=0.8, center=(0.1, 0.0, -0.2))
    v[:, 1::3] *= 0.6  # not a cube-filling shape, so the permutation is visible
    stl = tmp_path / "in.stl"
    with open(stl, "wb") as f:
        f.write(b"binary stl".ljust(80, b" ") + struct.pack("<I", len(v)))
        for t in v:
            f.write(struct.pack("<12fH", 0, 0, 0, *t.tolist(), 0))
    out = tmp_path / "out.vl32"
    r = subprocess.run([cli, str(stl), str(out), "-r", "48", "-s", "blend", "-p", "zYx", "-u", "-j", "2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = _vl32_to_voxels(out.read_bytes())
    # -p zYx: row i of the unit transform selects axis perm[i]; capital = flipped (main.cpp:224-262)
    unit = [0, 0, 1, 0, -1, 0, 1, 0, 0]
    want = oracle.voxelize(v, 48, strategy=1, supersampling=2, unit_transform=unit)
    assert np.array_equal(meshes.sorted_voxels(got), meshes.sorted_voxels(want))
    bad = subprocess.run([cli, str(stl), str(out)], capture_output=True, text=True)
    assert bad.returncode == 1  # resolution is required


def test_obj_fallback_texture_for_faces_without_material(tmp_path, oracle):
    """reference src/io.cpp:280-288: a face with uv coordinates but no material uses the instance's fallback texture
    (obj2voxel_set_texture); without one it is material-less."""
    from obj2voxel_amd import capi
    a = capi.api()
    a.obj2voxel_set_log_level(capi.LOG_SILENT)
    v, uv = meshes.uv_sphere(7, with_uv=True)
    T = len(v)
    lines = []
    for t in range(T):
        for k in range(3):
            lines.append("v %r %r %r" % tuple(float(x) for x in v[t, k * 3:k * 3 + 3]))
            lines.append("vt %r %r" % tuple(float(x) for x in uv[t, k * 2:k * 2 + 2]))
    # quads are not used here; negative (relative) indices are: -3/-3 -2/-2 -1/-1 right after each vertex triple
    body = []
    for t in range(T):
        body += lines[6 * t:6 * t + 6] + ["f -3/-3 -2/-2 -1/-1"]
    obj = tmp_path / "m.obj"
    obj.write_text("\n".join(body) + "\n")
    pix = meshes.checker_texture(16, 4)
    tex = a.obj2voxel_texture_alloc()
    assert a.obj2voxel_texture_load_pixels(tex, pix.ctypes.data, 16, 16, 3)
    for with_tex in (True, False):
        inst = a.obj2voxel_alloc()
        in_b = str(obj).encode()
        a.obj2voxel_set_input_file(inst, in_b, None)
        a.obj2voxel_set_output_memory(inst, b"vl32")
        a.obj2voxel_set_resolution(inst, 56)
        if with_tex:
            a.obj2voxel_set_texture(inst, tex)
        assert a.obj2voxel_voxelize(inst) == capi.ERR_OK
        size = C.c_size_t(0)
        ptr = a.obj2voxel_get_output_memory(inst, C.byref(size))
        got = _vl32_to_voxels(bytes(np.ctypeslib.as_array(ptr, shape=(size.value,))))
        a.obj2voxel_free(inst)
        if with_tex:
            want = oracle.voxelize(v, 56, uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32),
                                   textures=[(pix, 1)])
        else:
            want = oracle.voxelize(v, 56)
        assert np.array_equal(meshes.sorted_voxels(got), meshes.sorted_voxels(want)), with_tex
    a.obj2voxel_texture_free(tex)
    a.obj2voxel_set_log_level(capi.LOG_INFO)
