#!/usr/bin/env python3
"""Device-pipeline timings on workload shapes other than bench.py's (BASELINE.json configs and stress shapes).
Developer tool: prints one JSON line per case; inputs resident in HBM, results left in HBM, like bench.py."""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes


ONLY = sys.argv[1] if len(sys.argv) > 1 else None   # substring of the case name: run only the matching cases


def run(dv, name, verts, res, steps=5, **kw):
    if ONLY and ONLY not in name:
        return
    uvs, types, colors, texids, textures = (kw.get(k) for k in ("uvs", "types", "colors", "texids", "textures"))
    if textures:
        dv.set_textures(textures)
    dv.set_triangles(verts, uvs=uvs, types=types, colors=colors, texids=texids)
    args = dict(strategy=kw.get("strategy", 0), supersampling=kw.get("supersampling", 1), read=False)
    dv.voxelize(res, **args)
    dv.voxelize(res, **args)
    t0 = time.perf_counter()
    for _ in range(steps):
        n = dv.voxelize(res, **args)
    dt = (time.perf_counter() - t0) / steps
    tm, st = dv.timings(), dv.stats()
    print(json.dumps({"case": name, "tris": len(verts), "res": res, "voxels": n, "ms": round(dt * 1e3, 3),
                      "mvox_s": round(n / dt / 1e6, 1), "stages_ms": {k: round(v, 3) for k, v in tm.items() if k != "passes"},
                      "leaves": st["leaves"], "candidates": st["candidates"], "hits": st["hits"], "passes": tm["passes"]}), flush=True)


def main():
    dv = hip.DeviceVoxelizer(0)
    tex = [(meshes.checker_texture(1024, 32), 1)]
    run(dv, "unit cube @1024 (12 aligned triangles)", meshes.unit_cube(), 1024)
    run(dv, "box room 8x8 quads @2048", meshes.box_room(8), 2048)
    v = meshes.uv_sphere(12)
    run(dv, "sphere nv=12 @1024 (subdivision heavy)", v, 1024)
    v, uv = meshes.uv_sphere(39, with_uv=True)
    T = len(v)
    run(dv, "config2: sphere nv=39 @512 textured BLEND", v, 512, uvs=uv, types=np.full(T, 3, np.uint32),
        texids=np.zeros(T, np.int32), textures=tex, strategy=1)
    v = meshes.uv_sphere(467)
    T = len(v)
    run(dv, "config3 variant: nv=467 @1024 coloured BLEND", v, 1024, types=np.full(T, 2, np.uint32),
        colors=meshes.triangle_colors(T), strategy=1)
    v, uv = meshes.uv_sphere(467, with_uv=True)
    run(dv, "config3 variant: nv=467 @1024 textured MAX", v, 1024, uvs=uv, types=np.full(T, 3, np.uint32),
        texids=np.zeros(T, np.int32), textures=tex, strategy=0)
    room = meshes.box_room(16)
    sph, suv = meshes.uv_sphere(255, radius=0.3, center=(0.5, 0.45, 0.55), with_uv=True)
    v = np.concatenate([room, sph])
    uv = np.concatenate([np.tile(np.array([0, 0, 1, 0, 1, 1], np.float32), (len(room), 1)), suv])
    T = len(v)
    run(dv, "config4: room+sphere (%d tris) @2048 x2 supersampling textured BLEND" % T, v, 2048, uvs=uv,
        types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32), textures=tex, strategy=1, supersampling=2, steps=3)
    run(dv, "config4 with MAX: room+sphere (%d tris) @2048 x2 supersampling textured MAX" % T, v, 2048, uvs=uv,
        types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32), textures=tex, strategy=0, supersampling=2, steps=3)


if __name__ == "__main__":
    main()
