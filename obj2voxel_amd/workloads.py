"""Named workloads of the device pipeline: the stand-ins of BASELINE.json's configurations and the
route variants bench.py times beside its headline, plus the real assets when $O2V_ASSETS holds them.

Used by bench.py (routes), tools/run_workload.py (the unit rocprofv3 wraps) and the tests.  One step of a workload = one pass
of the whole device pipeline over triangles already resident in HBM, the (x, y, z, argb) records left in HBM.
"""
import os
import time

import numpy as np

from . import hip, meshes


def _sphere(nv):
    return lambda: (meshes.uv_sphere(nv), {}, None)


def _coloured(nv):
    def make():
        v = meshes.uv_sphere(nv)
        T = len(v)
        return v, dict(types=np.full(T, 2, np.uint32), colors=meshes.triangle_colors(T)), None
    return make


def _scan():
    v = meshes.scan_like()
    T = len(v)
    return v, dict(types=np.full(T, 2, np.uint32), colors=meshes.triangle_colors(T)), None


def _textured(nv):
    def make():
        v, uv = meshes.uv_sphere(nv, with_uv=True)
        T = len(v)
        return v, dict(uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32)), [(meshes.checker_texture(1024, 32), 1)]
    return make


def _blade():
    v, uv = meshes.readme_blade()
    T = len(v)
    return v, dict(uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32)), [(meshes.checker_texture(1024, 32), 1)]


def _sponza():
    room = meshes.box_room(16)
    sph, suv = meshes.uv_sphere(255, radius=0.3, center=(0.5, 0.45, 0.55), with_uv=True)
    v = np.concatenate([room, sph])
    uv = np.concatenate([np.tile(np.array([0, 0, 1, 0, 1, 1], np.float32), (len(room), 1)), suv])
    T = len(v)
    return v, dict(uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32)), [(meshes.checker_texture(1024, 32), 1)]


# name: (mesh factory -> (verts, materials, textures), resolution, voxelize keywords, description)
WORKLOADS = {
    "config2": (_sphere(467), 1024, dict(strategy=0), "BASELINE configs[2] stand-in: uv-sphere nv=467 (870 488 tris) @1024^3, MATERIALLESS, MAX (the bench headline)"),
    "config2_colored_max": (_coloured(467), 1024, dict(strategy=0), "configs[2] mesh with per-triangle colours, MAX: direct path, colour by winner"),
    "config2_blend": (_coloured(467), 1024, dict(strategy=1), "configs[2] mesh with per-triangle colours, BLEND: pool -> counting sort -> ordered replay"),
    "config2_textured_max": (_textured(467), 1024, dict(strategy=0), "configs[2] mesh textured, MAX: k_voxelize<true>, direct path with pick records"),
    "scan_colored_max": (_scan, 1024, dict(strategy=0), "irregular mesh: adaptively refined, noise-displaced icosphere (908 288 tris, areas spread 380 : 1, 5 % slivers) @1024^3, coloured, MAX"),
    "scan_blend": (_scan, 1024, dict(strategy=1), "the same irregular mesh, coloured, BLEND"),
    "config1": (_textured(39), 512, dict(strategy=1), "BASELINE configs[1] stand-in: uv-sphere nv=39 (5 928 tris) @512^3, textured, BLEND"),
    "config3": (_sponza, 2048, dict(strategy=1, supersampling=2), "BASELINE configs[3] stand-in: box room + sphere (262 092 textured tris) @2048^3 x2 supersampling, BLEND"),
    "config3_max": (_sponza, 2048, dict(strategy=0, supersampling=2), "configs[3] stand-in with MAX"),
    "readme8192": (_blade, 8192, dict(strategy=0), "stand-in of the reference README's showcase run: 19 320 textured triangles (a long thin ellipsoid) @8192^3, MAX - "
                   "the dense grids cover the mesh's voxel bounding box (8192 x ~670 x ~670 cells), not the cube"),
    "cube1024": (lambda: (meshes.unit_cube(), {}, None), 1024, dict(strategy=0), "unit cube @1024^3 (12 aligned triangles)"),
    "room2048": (lambda: (meshes.box_room(8), {}, None), 2048, dict(strategy=0), "box room 8x8 quads per wall @2048^3"),
    "lowpoly1024": (_sphere(12), 1024, dict(strategy=0), "sphere nv=12 @1024^3 (subdivision heavy)"),
}

# the routes bench.py times after its headline (N = 1), in this order
BENCH_ROUTES = ("config2_colored_max", "config2_blend", "config2_textured_max", "scan_colored_max", "config1", "config3", "readme8192")

# real assets: file stem under $O2V_ASSETS -> (BASELINE configuration it belongs to, resolution, voxelize keywords)
ASSETS = {
    "spot": ("configs[1]", 512, dict(strategy=1)),
    "dragon": ("configs[2]", 1024, dict(strategy=0)),
    "sponza": ("configs[3]", 2048, dict(strategy=1, supersampling=2)),
}


def asset_path(stem):
    """$O2V_ASSETS/<stem>.obj (or .stl) if it exists, else None."""
    root = os.environ.get("O2V_ASSETS")
    if not root:
        return None
    for ext in (".obj", ".stl"):
        p = os.path.join(root, stem + ext)
        if os.path.isfile(p):
            return p
    return None


def load(name):
    """(verts, materials, textures, resolution, keywords, description) of a named workload or of 'asset:<stem>'."""
    if name.startswith("asset:"):
        stem = name.split(":", 1)[1]
        path = asset_path(stem)
        if path is None:
            raise FileNotFoundError(f"$O2V_ASSETS holds no {stem}.obj / {stem}.stl")
        cfg, res, kw = ASSETS[stem]
        verts, mat, textures = hip.load_mesh_file(path)
        what = "MAX" if kw.get("strategy", 0) == 0 else "BLEND"
        ss = kw.get("supersampling", 1)
        text = (f"BASELINE {cfg}: real asset {os.path.basename(path)} ({len(verts)} tris) @{res}^3" +
                (f" x{ss} supersampling" if ss > 1 else "") + f", materials as in the file, {what}")
        return verts, mat, textures or None, res, kw, text
    make, res, kw, text = WORKLOADS[name]
    verts, mat, textures = make()
    return verts, mat, textures, res, kw, text


def run(name, steps=5, warmup=2, dv=None, kernel_steps=0, loaded=None):
    """Times `steps` passes of the workload (wall clock around the calls, each of which waits for the device) and, in
    `kernel_steps` further passes, the individual kernels (O2V_HIP_FLAG_KERNEL_TIMES: event pairs around every launch)."""
    verts, mat, textures, res, kw, text = loaded if loaded is not None else load(name)
    own = dv is None
    if own:
        dv = hip.DeviceVoxelizer(0)
    try:
        dv.set_textures(textures or [])
        dv.set_triangles(verts, **mat)
        for _ in range(warmup):
            dv.voxelize(res, read=False, **kw)
        n = 0
        t0 = time.perf_counter()
        for _ in range(steps):   # the timed region: the steps and nothing else
            n = dv.voxelize(res, read=False, **kw)
        dt = (time.perf_counter() - t0) / max(steps, 1)
        k2_ms = dv.timings()["voxelize_ms"]   # the clip kernel's own duration in the last timed step
        # per-stage device times: read in a few further steps (a library call and a dictionary per step)
        acc = {}
        stage_steps = min(3, max(steps, 1))
        for _ in range(stage_steps):
            dv.voxelize(res, read=False, stage_times=True, **kw)   # (an event between the stages: not in the timed steps)
            for k, v in dv.timings().items():
                if isinstance(v, (int, float)):
                    acc[k] = acc.get(k, 0.0) + v
        st = dv.stats()
        passes = dv.timings()["passes"]
        kernels = {}
        for _ in range(kernel_steps):
            dv.voxelize(res, read=False, kernel_times=True, **kw)
            for k, (ms, launches) in dv.kernel_times().items():
                e = kernels.setdefault(k, [0.0, 0])
                e[0] += ms
                e[1] += launches
        out = {"workload": name, "what": text, "tris": len(verts), "res": res, "supersampling": kw.get("supersampling", 1),
               "strategy": "BLEND" if kw.get("strategy", 0) else "MAX", "textured": bool(textures), "voxels": int(n),
               "ms": round(dt * 1e3, 4), "mvox_s": round(n / dt / 1e6, 1) if dt > 0 else None, "mtris_s": round(len(verts) / dt / 1e6, 2) if dt > 0 else None,
               "stages_ms": {k: round(v / stage_steps, 4) for k, v in acc.items() if k.endswith("_ms")}, "k2_ms_timed": round(k2_ms, 4),
               "passes": passes, "stats": st,
               "build_id": hip.build_id()}
        if kernel_steps:
            out["kernels_ms"] = {k: {"ms": round(ms / kernel_steps, 4), "launches": launches // kernel_steps} for k, (ms, launches) in kernels.items()}
        return out
    finally:
        if own:
            dv.close()


def kernel_algorithmic_bytes(stats, textured, strategy_blend, kernels=None):
    """Algorithmic bytes per launch of the pipeline's main kernels from the device statistics of a
    run: what each kernel has to read and write once, without re-reads, write amplification or cache-line granularity."""
    L, tiles, H, V, T = stats["leaves"], stats["tiles"], stats["hits"], stats["voxels"], stats["triangles"]
    D, slots, Hd, jobs = stats["dirty_bricks"], stats["pool_slots"], stats["direct_hits"], stats["jobs"]
    B = stats["bricks"]
    cpb = stats["grid_cells"] // max(B, 1)
    Hp = H - Hd
    rec = 24 if textured else 16
    direct = Hd > 0
    pick = 32 if (direct and textured) else 0
    out = {
        "k_bounds": 36 * T,
        "k_expand_roots": (36 + (24 if textured else 0)) * T + 96 * L + 8 * tiles,
        # per pooled hit: the counter atomic and the record - in the brick's slab for a cell's first eight hits, in the pool (32
        # bytes) for the later ones of crowded cells, which `slots` counts
        ("k_voxelize<true>" if textured else "k_voxelize<false>"): 96 * L + 8 * tiles + 16 * jobs + (8 + 1 + pick) * Hd + (4 + rec) * Hp + 32 * (slots if Hp else 0),
    }
    if Hp:
        # (D = the bricks listed before k_voxelize: every brick a leaf's box touches)
        out["k_mark_bricks"] = 12 * L + D
        out["k_scan_bricks"] = 4 * cpb * D + 16 * V
        out["k_scatter"] = (32 + 4 + rec) * slots
        out["k_reset_bricks"] = 4 * cpb * D
        out["k_resolve<6>" if textured else "k_resolve<4>"] = 16 * V + rec * Hp + 16 * V   # (all tiers together: cells + records + output)
    if direct and (stats.get("certain_hits") or stats.get("bypassed_leaves")):
        # occupancy-only mode: root triangles of one tile have no Leaf / Tile record (both kernels read their 36 bytes); a job
        # record per voxel job, written, read by the filter, the live ones written and read once more; a byte and a flag per
        # hit; 64 bytes per dirty brick
        Lb = stats.get("bypassed_leaves", 0)
        out.pop("k_voxelize<false>", None)
        out["k_expand_roots"] = 36 * T + 96 * (L - Lb) + 8 * (tiles - Lb)
        out["k_voxelize_occ"] = 36 * T + 96 * (L - Lb) + 8 * (tiles - Lb) + 16 * jobs + 16 * (jobs - stats.get("skipped_jobs", 0)) + 2 * H
        out["k_emit_occ"] = 2 * cpb * D + 16 * V
        if kernels is not None and "k_count_roots" in kernels:
            # (a tessellated surface: k_count_roots reads the vertex array; k_expand_roots only the blocks it lists - unknown here,
            # as a rule none - and writes the records of what is not bypassed)
            out["k_count_roots"] = 36 * T
            out["k_expand_roots"] = 96 * (L - Lb) + 8 * (tiles - Lb)
    elif direct:
        out["k_emit_max"] = 8 * cpb * D + 32 * V + 16 * V
    return {k: int(v) for k, v in out.items()}
