"""o2v_mesh_load_file (include/o2v_hip.h): a triangle file read into the device C-ABI's flat arrays by the library's own
OBJ / STL readers - the route bench.py takes for real assets under $O2V_ASSETS.  No GPU needed."""
import struct

import numpy as np

from obj2voxel_amd import hip, meshes
from tests.test_gpu_io import _png_rgb


def test_stl_round_trip(tmp_path):
    v = meshes.uv_sphere(6)
    stl = tmp_path / "m.stl"
    with open(stl, "wb") as f:
        f.write(b"binary stl".ljust(80, b" ") + struct.pack("<I", len(v)))
        for t in v:
            f.write(struct.pack("<12fH", 0, 0, 0, *t.tolist(), 0))
    verts, mat, textures = hip.load_mesh_file(stl)
    assert np.array_equal(verts, v) and mat == {} and textures == []


def test_obj_with_materials(tmp_path):
    v, uv = meshes.uv_sphere(5, with_uv=True)
    T = len(v)
    tex = meshes.checker_texture(16, 4)
    (tmp_path / "tex.png").write_bytes(_png_rgb(tex))
    (tmp_path / "m.mtl").write_text("newmtl red\nKd 0.8 0.25 0.125\nnewmtl checker\nKd 1 1 1\nmap_Kd tex.png\n")
    lines = ["mtllib m.mtl"]
    for t in range(T):
        for k in range(3):
            lines.append("v %r %r %r" % tuple(float(x) for x in v[t, k * 3:k * 3 + 3]))
            lines.append("vt %r %r" % tuple(float(x) for x in uv[t, k * 2:k * 2 + 2]))
    half = T // 2
    lines.append("usemtl red")
    lines += [f"f {3 * t + 1} {3 * t + 2} {3 * t + 3}" for t in range(half)]
    lines.append("usemtl checker")
    lines += [f"f {3 * t + 1}/{3 * t + 1} {3 * t + 2}/{3 * t + 2} {3 * t + 3}/{3 * t + 3}" for t in range(half, T)]
    obj = tmp_path / "mesh.obj"
    obj.write_text("\n".join(lines) + "\n")
    verts, mat, textures = hip.load_mesh_file(obj)
    assert np.array_equal(verts, v)
    assert np.array_equal(mat["types"], np.array([2] * half + [3] * (T - half), np.uint32))
    assert np.allclose(mat["colors"][:half], [0.8, 0.25, 0.125])
    assert np.array_equal(mat["uvs"][half:], uv[half:]) and np.all(mat["texids"] == 0)
    assert len(textures) == 1 and textures[0][0].shape == (16, 16, 4) and textures[0][1] == 1
    assert np.array_equal(textures[0][0][..., 1:], tex) and np.all(textures[0][0][..., 0] == 255)


def test_build_id_is_a_source_hash():
    bid = hip.build_id()
    assert len(bid) == 16 and int(bid, 16) >= 0


def test_build_id_matches_the_device_sources():
    """The id embedded in the library is the hash the Makefile computes over the device sources: a library whose device
    object was not rebuilt after a source change would carry counters' trust it does not deserve (bench.py compares it with
    the id recorded in profiles/current.json)."""
    import glob
    import hashlib
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "obj2voxel_amd", "csrc")
    files = [os.path.join(csrc, "o2v_device.hip")] + sorted(glob.glob(os.path.join(csrc, "o2v_dev_*.hpp"))) + [os.path.join(csrc, "o2v_math.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    assert hip.build_id() == h.hexdigest()[:16]
