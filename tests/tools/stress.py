"""Developer tool (test infrastructure: it uses the oracle): seeded random parity beyond the fixed cases of tests/test_gpu_fuzz.py,
with the call shapes round 6 added - x / y tiles (o2v_hip_params::x_begin ..), z-slabs inside them, and tessellated material-less
surfaces whose root stage is left out (Params::solo_roots) - device vs. oracle, record for record.

usage: python tests/tools/stress.py [--seeds A B] [--minutes M]        (prints one line per failure and a summary; exit 1 if any)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from obj2voxel_amd import hip, meshes  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_gpu_fuzz import _case  # noqa: E402


def _tile(rng, res):
    """A random x or y range of the output grid, begin a multiple of 4 (or the whole axis)."""
    if rng.random() < 0.3 or res < 8:
        return (0, 0)
    b = int(rng.integers(0, max(res // 4, 1))) * 4
    b = min(b, ((res - 1) // 4) * 4)
    e = int(rng.integers(b + 1, res + 1))
    return (b, e)


def _inside(vox, xt, yt, zt, res):
    m = np.ones(len(vox), bool)
    for axis, t in ((0, xt), (1, yt), (2, zt)):
        if tuple(t) != (0, 0):
            m &= (vox[:, axis] >= t[0]) & (vox[:, axis] < t[1])
    return vox[m]


def soup_case(dv, seed):
    v, res, kw, mat, textures = _case(seed)
    rng = np.random.default_rng(77_000 + seed)
    xt, yt = _tile(rng, res), _tile(rng, res)
    dv.set_textures(textures)
    dv.set_triangles(v, **mat)
    got = meshes.sorted_voxels(dv.voxelize(res, xtile=xt, ytile=yt, **kw))
    want = oracle.voxelize(v, res, textures=textures, **mat, **kw)
    want = meshes.sorted_voxels(_inside(want, xt, yt, (0, 0), res))
    return np.array_equal(got, want), dict(kind="soup", seed=seed, res=res, xtile=xt, ytile=yt, kw={k: kw[k] for k in kw if k != "unit_transform"},
                                           got=len(got), want=len(want))


def surface_case(dv, seed):
    """A tessellated material-less surface, triangles a few voxels across (with and without the root stage), random tile and slab."""
    rng = np.random.default_rng(91_000 + seed)
    nv = int(rng.integers(12, 160))
    scale = rng.random(3) * 0.8 + 0.2
    v = meshes.uv_sphere(nv, radius=1.0) * np.tile(scale, 3).astype(np.float32)
    per_voxel = float(rng.choice([1.5, 2.5, 3.5, 4.5, 6.0, 9.0]))          # triangle size in voxels (around the solo limit of 5)
    res = int(min(max(per_voxel * 2.0 * nv / np.pi, 8), 700))            # a triangle is ~ pi res / (2 nv) voxels across
    ss = int(rng.choice([1, 1, 2]))
    xt, yt = _tile(rng, res), _tile(rng, res)
    zt = _tile(rng, res) if rng.random() < 0.5 else (0, 0)
    if zt != (0, 0) and zt[0] >= zt[1]:
        zt = (0, 0)
    dv.set_triangles(v)
    got = meshes.sorted_voxels(dv.voxelize(res, supersampling=ss, xtile=xt, ytile=yt, zslab=zt))
    st = dv.stats()
    want = oracle.voxelize(v, res, supersampling=ss)
    want = meshes.sorted_voxels(_inside(want, xt, yt, zt, res))
    return np.array_equal(got, want), dict(kind="surface", seed=seed, nv=nv, res=res, ss=ss, xtile=xt, ytile=yt, zslab=zt, got=len(got), want=len(want),
                                           bypassed=st["bypassed_leaves"], leaves=st["leaves"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=(5000, 5400))
    ap.add_argument("--minutes", type=float, default=8.0)
    a = ap.parse_args()
    oracle.build()
    oracle.set_threads(8)
    dv = hip.DeviceVoxelizer(0)
    t0 = time.time()
    n = bad = 0
    try:
        for seed in range(a.seeds[0], a.seeds[1]):
            for fn in (soup_case, surface_case):
                ok, info = fn(dv, seed)
                n += 1
                if not ok:
                    bad += 1
                    print("MISMATCH", info, flush=True)
            if time.time() - t0 > a.minutes * 60:
                break
    finally:
        dv.close()
    print(f"stress: {n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
