"""Host-side voxel sinks without a GPU: a small C++ harness links against the
library's sink factory (o2v::open_memory_sink, obj2voxel_amd/csrc/o2v_io.hpp), feeds it a synthetic voxel list and
the files are parsed back here. Formats: reference README.adoc:210-264; QEF 0.2 and MagicaVoxel .vox 150 as published."""
import os
import subprocess

import numpy as np
import pytest

from obj2voxel_amd import meshes
from tests.test_gpu_io import _parse_qef, _parse_vox

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include "o2v_io.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
// usage: harness FORMAT RESOLUTION IN.bin OUT  (IN.bin: uint32 quadruples x, y, z, argb)
int main(int argc, char **argv)
{
    using namespace o2v;
    if (argc != 5) return 2;
    const FileFormat f = detect_format(nullptr, argv[1]);
    std::vector<uint8_t> in;
    if (!read_whole_file(argv[3], in)) return 3;
    std::vector<uint32_t> v(in.size() / 4);
    for (size_t i = 0; i < v.size(); ++i) v[i] = ((const uint32_t *) in.data())[i];
    auto sink = open_memory_sink(f, (uint32_t) std::atoi(argv[2]));
    if (!sink) return 4;
    // two batches, like the voxel callback would deliver them
    const size_t n = v.size() / 4, half = n / 2;
    sink->write(v.data(), half);
    sink->write(v.data() + half * 4, n - half);
    sink->finalize();
    if (!sink->can_write() || sink->written != n) return 5;
    std::FILE *out = std::fopen(argv[4], "wb");
    const o2v::ByteBuffer *mem = sink->memory();
    if (!out || !mem) return 6;
    std::fwrite(mem->bytes, 1, mem->size, out);
    std::fclose(out);
    return 0;
}
'''


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    import obj2voxel_amd
    obj2voxel_amd.build()
    d = tmp_path_factory.mktemp("io_harness")
    src = d / "harness.cpp"
    src.write_text(HARNESS)
    exe = d / "harness"
    libdir = os.path.dirname(obj2voxel_amd.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "obj2voxel_amd", "csrc"), str(src), "-o", str(exe),
                           "-L", libdir, "-lobj2voxel_amd", "-Wl,-rpath," + libdir])
    return exe


def _encode(harness, tmp_path, fmt, res, voxels):
    a, b = tmp_path / "in.bin", tmp_path / ("out." + fmt)
    np.ascontiguousarray(voxels, dtype=np.uint32).tofile(a)
    subprocess.check_call([str(harness), fmt, str(res), str(a), str(b)])
    return b.read_bytes()


def _voxels(res, n, n_colors, seed):
    rng = np.random.default_rng(seed)
    pos = rng.permutation(res ** 3)[:n]
    xyz = np.stack([pos % res, (pos // res) % res, pos // (res * res)], axis=1)
    palette = 0xFF000000 | rng.integers(0, 1 << 24, size=n_colors, dtype=np.uint64)
    argb = palette[rng.integers(0, n_colors, size=n)]
    return np.concatenate([xyz, argb[:, None]], axis=1).astype(np.uint32)


def test_list_formats(harness, tmp_path):
    v = _voxels(40, 3000, 500, 1)
    want = meshes.sorted_voxels(v)
    data = _encode(harness, tmp_path, "vl32", 40, v)
    assert len(data) == 16 * len(v)
    got = np.frombuffer(data, dtype=">u4").astype(np.uint32).reshape(-1, 4)
    assert np.array_equal(meshes.sorted_voxels(got), want)
    data = _encode(harness, tmp_path, "ply", 40, v)
    assert len(data) == 300 + 16 * len(v) and data[:4] == b"ply\n" and data[:300].endswith(b"end_header\n")
    assert int(data[:300].decode().split("element vertex ")[1].split("\n")[0]) == len(v)
    assert np.array_equal(meshes.sorted_voxels(np.frombuffer(data[300:], dtype=">u4").astype(np.uint32).reshape(-1, 4)), want)
    data = _encode(harness, tmp_path, "xyzrgb", 40, v)
    rows = np.array([[int(t) for t in ln.split()] for ln in data.decode().splitlines()], dtype=np.uint32)
    argb = 0xFF000000 | (rows[:, 3] << 16) | (rows[:, 4] << 8) | rows[:, 5]
    assert np.array_equal(meshes.sorted_voxels(np.concatenate([rows[:, :3], argb[:, None]], axis=1).astype(np.uint32)), want)


def test_qef_keeps_every_colour_and_marks_visible_faces(harness, tmp_path):
    v = _voxels(24, 5000, 700, 2)         # dense enough that many faces are hidden
    size, colors, rows = _parse_qef(_encode(harness, tmp_path, "qef", 24, v))
    assert size == [24, 24, 24] and len(colors) == len(np.unique(v[:, 3])) and len(rows) == len(v)
    rgb = np.rint(colors[rows[:, 3]] * 255).astype(np.uint32)
    got = np.concatenate([rows[:, :3].astype(np.uint32), (0xFF000000 | (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2])[:, None]],
                         axis=1).astype(np.uint32)
    assert np.array_equal(meshes.sorted_voxels(got), meshes.sorted_voxels(v))
    filled = {tuple(p) for p in v[:, :3].tolist()}
    for x, y, z, _, mask in rows.tolist():
        want = sum(bit for bit, d in ((2, (-1, 0, 0)), (4, (1, 0, 0)), (8, (0, 1, 0)), (16, (0, -1, 0)), (32, (0, 0, 1)),
                                      (64, (0, 0, -1))) if (x + d[0], y + d[1], z + d[2]) not in filled)
        assert mask == want


def test_vox_exact_palette_models_and_quantisation(harness, tmp_path):
    # <= 255 colours: exact; one model up to 256^3
    v = _voxels(200, 4000, 255, 3)
    models, trans, palette = _parse_vox(_encode(harness, tmp_path, "vox", 200, v))
    assert len(models) == 1 and models[0][0] == (200, 200, 200) and not trans
    xyzi = models[0][1]
    rgba = palette[xyzi[:, 3].astype(int) - 1].astype(np.uint32)
    argb = (rgba[:, 3] << 24) | (rgba[:, 0] << 16) | (rgba[:, 1] << 8) | rgba[:, 2]
    got = np.concatenate([xyzi[:, :3].astype(np.uint32), argb[:, None]], axis=1).astype(np.uint32)
    assert np.array_equal(meshes.sorted_voxels(got), meshes.sorted_voxels(v))

    # 600^3 -> 27 blocks of at most 256^3, > 255 colours -> median cut
    v = _voxels(600, 20000, 5000, 4)
    models, trans, palette = _parse_vox(_encode(harness, tmp_path, "vox", 600, v))
    assert 1 < len(models) <= 27
    parts = []
    for k, (size, xyzi) in enumerate(models):
        origin = np.array(trans[("model", k)]) - np.array(size) // 2
        assert set(origin.tolist()) <= {0, 256, 512} and all(sz in (256, 88) for sz in size)
        assert (xyzi[:, 3] >= 1).all() and (xyzi[:, :3] < np.array(size)).all()
        rgba = palette[xyzi[:, 3].astype(int) - 1].astype(np.uint32)
        argb = (rgba[:, 3] << 24) | (rgba[:, 0] << 16) | (rgba[:, 1] << 8) | rgba[:, 2]
        parts.append(np.concatenate([xyzi[:, :3].astype(np.uint32) + origin.astype(np.uint32), argb[:, None]], axis=1))
    got, want = meshes.sorted_voxels(np.concatenate(parts).astype(np.uint32)), meshes.sorted_voxels(v)
    assert np.array_equal(got[:, :3], want[:, :3])
    ga = ((got[:, 3:4] >> np.array([16, 8, 0], np.uint32)) & 255).astype(int)
    wa = ((want[:, 3:4] >> np.array([16, 8, 0], np.uint32)) & 255).astype(int)
    err = np.abs(ga - wa)
    assert err.mean() < 12 and err.max() < 64          # 5000 random colours into 255 boxes of the RGB cube
    assert len(np.unique(got[:, 3])) <= 255


SOURCE_HARNESS = r'''
#include "o2v_io.hpp"
#include <cstdio>
#include <cstring>
// usage: src_harness (stl|obj) IN OUT : dumps every triangle as 9 v + 6 t + type + 3 color + has_texture (20 x 4 bytes)
int main(int argc, char **argv)
{
    using namespace o2v;
    if (argc != 4) return 2;
    std::unique_ptr<TriangleSource> src = std::strcmp(argv[1], "stl") == 0 ? open_stl_file(argv[2]) : open_obj_file(argv[2], nullptr);
    if (!src) return 3;
    std::FILE *out = std::fopen(argv[3], "wb");
    if (!out) return 4;
    while (const HostTriangle *t = src->next()) {
        float rec[20];
        std::memcpy(rec, t->v, 36);
        std::memcpy(rec + 9, t->t, 24);
        const uint32_t type = t->type, has = t->texture != nullptr;
        std::memcpy(rec + 15, &type, 4);
        std::memcpy(rec + 16, t->color, 12);
        std::memcpy(rec + 19, &has, 4);
        std::fwrite(rec, 4, 20, out);
    }
    std::fclose(out);
    return 0;
}
'''


@pytest.fixture(scope="module")
def src_harness(tmp_path_factory):
    import obj2voxel_amd
    obj2voxel_amd.build()
    d = tmp_path_factory.mktemp("src_harness")
    src = d / "src_harness.cpp"
    src.write_text(SOURCE_HARNESS)
    exe = d / "src_harness"
    libdir = os.path.dirname(obj2voxel_amd.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "obj2voxel_amd", "csrc"), str(src), "-o", str(exe),
                           "-L", libdir, "-lobj2voxel_amd", "-Wl,-rpath," + libdir])
    return exe


def _read_triangles(exe, kind, path, tmp_path):
    out = tmp_path / "tris.bin"
    subprocess.check_call([str(exe), kind, str(path), str(out)])
    raw = np.fromfile(out, dtype=np.float32).reshape(-1, 20)
    return raw[:, :9], raw[:, 9:15], raw[:, 15].view(np.uint32), raw[:, 16:19], raw[:, 19].view(np.uint32)


def test_binary_stl_source(src_harness, tmp_path):
    """reference src/io.cpp:395-435 (50-byte records after an 80-byte header and a count); every triangle MATERIALLESS."""
    import struct
    v = meshes.uv_sphere(6)
    stl = tmp_path / "m.stl"
    with open(stl, "wb") as f:
        f.write(b"x".ljust(80, b" ") + struct.pack("<I", len(v)))
        for t in v:
            f.write(struct.pack("<12fH", 0, 0, 0, *t.tolist(), 0))
    verts, _, types, _, has_tex = _read_triangles(src_harness, "stl", stl, tmp_path)
    assert np.array_equal(verts, np.reshape(v, (-1, 9))) and (types == 1).all() and not has_tex.any()


def test_obj_source_materials_polygons_and_png(src_harness, tmp_path):
    """The OBJ subset the reference consumes through tinyobjloader (src/io.cpp:244-312,351-393): v / vt / f with negative
    and v/vt/vn indices, polygon fans, usemtl with Kd (UNTEXTURED) or map_Kd PNG (TEXTURED), faces before any usemtl
    (MATERIALLESS)."""
    from tests.test_gpu_io import _png_rgb
    (tmp_path / "t.png").write_bytes(_png_rgb(meshes.checker_texture(8, 2)))
    (tmp_path / "m.mtl").write_text("newmtl red\nKd 0.5 0.25 0.125\nnewmtl tex\nKd 1 1 1\nmap_Kd t.png\n")
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "vn 0 0 1\n"
        "f 1 2 3\n"                      # no material yet
        "usemtl red\n"
        "f 1 2 3 4\n"                    # quad -> fan of two triangles
        "f -1 -5 -4\n"                   # negative indices: v5, v1, v2
        "usemtl tex\n"
        "f 1/1/1 2/2/1 3/3/1\n"
        "f 1/1 3/3 4/4\n")
    verts, uvs, types, colors, has_tex = _read_triangles(src_harness, "obj", tmp_path / "m.obj", tmp_path)
    P = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    tri = lambda *i: np.concatenate([P[k] for k in i])
    assert len(verts) == 6
    assert np.array_equal(verts[0], tri(0, 1, 2)) and types[0] == 1
    assert np.array_equal(verts[1], tri(0, 1, 2)) and np.array_equal(verts[2], tri(0, 2, 3))
    assert (types[1:4] == 2).all() and np.allclose(colors[1:4], [0.5, 0.25, 0.125])
    assert np.array_equal(verts[3], tri(4, 0, 1))
    assert (types[4:6] == 3).all() and has_tex[4:6].all() and not has_tex[:4].any()
    assert np.array_equal(uvs[4], [0, 0, 1, 0, 1, 1]) and np.array_equal(uvs[5], [0, 0, 1, 1, 0, 1])
