"""Generates tests/golden/oracle_regression.npz with the CPU oracle (oracle/o2v_oracle.c).

These are regression vectors of the oracle itself (small, full (x,y,z,argb) dumps, sorted by z,y,x).
They are NOT reference outputs: the reference cannot be built in this image (voxel-io is absent), and the
only results its tests pin are the four voxel counts asserted in tests/test_oracle_golden.py.
Run:  python -m tests.golden.make_golden
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from obj2voxel_amd import meshes  # noqa: E402


def _sphere_colored(o, res, strategy, nv=9):
    s = meshes.uv_sphere(nv)
    T = len(s)
    return o.voxelize(s, res, types=np.full(T, o.TRI_UNTEXTURED), colors=meshes.triangle_colors(T), strategy=strategy)


def _sphere_textured(o, res, strategy, ss=1):
    s, uv = meshes.uv_sphere(9, with_uv=True)
    T = len(s)
    return o.voxelize(s, res, uvs=uv, types=np.full(T, o.TRI_TEXTURED), texids=np.zeros(T, np.int32),
                      textures=[(meshes.checker_texture(64, 8), 1)], strategy=strategy, supersampling=ss)


CASES = {
    "cube32_max": lambda o: o.voxelize(meshes.unit_cube(), 32),
    "planes32_max": lambda o: o.voxelize(meshes.three_planes(), 32),
    "sphere9_c48_max": lambda o: _sphere_colored(o, 48, o.STRATEGY_MAX),
    "sphere9_c48_blend": lambda o: _sphere_colored(o, 48, o.STRATEGY_BLEND),
    "sphere9_c100_max": lambda o: _sphere_colored(o, 100, o.STRATEGY_MAX),       # two chunks per axis, subdivision
    "sphere9_t64_blend": lambda o: _sphere_textured(o, 64, o.STRATEGY_BLEND),
    "sphere9_t32_ss2_max": lambda o: _sphere_textured(o, 32, o.STRATEGY_MAX, ss=2),  # documented downscale semantics
    "soup_c40_blend": lambda o: (lambda m: o.voxelize(m, 40, types=np.full(len(m), o.TRI_UNTEXTURED),
                                                     colors=meshes.triangle_colors(len(m)),
                                                     strategy=o.STRATEGY_BLEND))(meshes.random_soup(60, seed=3)),
}


def run_case(o, name):
    return CASES[name](o)


if __name__ == "__main__":
    from oracle import oracle as o
    o.build()
    out = {name: meshes.sorted_voxels(run_case(o, name)) for name in CASES}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz")
    np.savez_compressed(path, **out)
    for k, v in out.items():
        print(k, v.shape)
