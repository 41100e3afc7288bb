#!/usr/bin/env python3
"""Runs one named workload of the device pipeline for a few steps (inputs resident in HBM, results left in HBM) and prints
one JSON line; the unit rocprofv3 wraps in tools/profile_all.sh and the cases tools/bench_configs.py prints.
usage: run_workload.py NAME [--steps K] [--warmup W]     NAME: see WORKLOADS"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from obj2voxel_amd import hip, meshes


def _sphere(nv, **kw):
    return lambda: (meshes.uv_sphere(nv), {}, None, kw)


def _coloured(nv):
    def make():
        v = meshes.uv_sphere(nv)
        T = len(v)
        return v, dict(types=np.full(T, 2, np.uint32), colors=meshes.triangle_colors(T)), None, {}
    return make


def _textured(nv):
    def make():
        v, uv = meshes.uv_sphere(nv, with_uv=True)
        T = len(v)
        return v, dict(uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32)), [(meshes.checker_texture(1024, 32), 1)], {}
    return make


def _sponza():
    room = meshes.box_room(16)
    sph, suv = meshes.uv_sphere(255, radius=0.3, center=(0.5, 0.45, 0.55), with_uv=True)
    v = np.concatenate([room, sph])
    uv = np.concatenate([np.tile(np.array([0, 0, 1, 0, 1, 1], np.float32), (len(room), 1)), suv])
    T = len(v)
    return v, dict(uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32)), [(meshes.checker_texture(1024, 32), 1)], {}


# name: (mesh factory, resolution, voxelize keywords, description)
WORKLOADS = {
    "config2": (_sphere(467), 1024, dict(strategy=0), "BASELINE configs[2] stand-in: nv=467 @1024^3 MATERIALLESS MAX (the bench workload)"),
    "config2_blend": (_coloured(467), 1024, dict(strategy=1), "configs[2] mesh with per-triangle colours, BLEND: pool -> counting sort -> replay"),
    "config2_textured_max": (_textured(467), 1024, dict(strategy=0), "configs[2] mesh textured, MAX: direct path with pick records"),
    "config1": (_textured(39), 512, dict(strategy=1), "BASELINE configs[1] stand-in: nv=39 (5928 tris) @512^3 textured BLEND"),
    "config3": (_sponza, 2048, dict(strategy=1, supersampling=2), "BASELINE configs[3] stand-in: room + sphere (262 092 textured tris) @2048^3 x2 SS, BLEND"),
    "config3_max": (_sponza, 2048, dict(strategy=0, supersampling=2), "configs[3] stand-in with MAX"),
    "cube1024": (lambda: (meshes.unit_cube(), {}, None, {}), 1024, dict(strategy=0), "unit cube @1024^3 (12 aligned triangles)"),
    "room2048": (lambda: (meshes.box_room(8), {}, None, {}), 2048, dict(strategy=0), "box room 8x8 quads per wall @2048^3"),
    "lowpoly1024": (_sphere(12), 1024, dict(strategy=0), "sphere nv=12 @1024^3 (subdivision heavy)"),
}


def run(name, steps=5, warmup=2, dv=None):
    make, res, kw, text = WORKLOADS[name]
    verts, mat, textures, _ = make()
    own = dv is None
    if own:
        dv = hip.DeviceVoxelizer(0)
    if textures:
        dv.set_textures(textures)
    dv.set_triangles(verts, **mat)
    for _ in range(warmup):
        dv.voxelize(res, read=False, **kw)
    acc = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        n = dv.voxelize(res, read=False, **kw)
        for k, v in dv.timings().items():
            acc[k] = acc.get(k, 0.0) + v
    dt = (time.perf_counter() - t0) / steps
    st = dv.stats()
    out = {"workload": name, "what": text, "tris": len(verts), "res": res, "voxels": int(n), "ms": round(dt * 1e3, 3),
           "mvox_s": round(n / dt / 1e6, 1), "stages_ms": {k: round(v / steps, 4) for k, v in acc.items() if k.endswith("_ms")},
           "passes": dv.timings()["passes"], "stats": st}
    if own:
        dv.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    print(json.dumps(run(a.name, a.steps, a.warmup)), flush=True)
