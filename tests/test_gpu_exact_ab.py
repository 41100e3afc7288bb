"""Fast vs. exact clip kernel on the device, at GPU scale.

k_voxelize carries the only arithmetic that can change occupancy and is not the reference's: the separating-axis row test
(row_span), the bounding-box plane masks with their margins (piece_masks) and the single-plane rule
(obj2voxel_amd/csrc/o2v_dev_k2_voxelize.hpp).  They only remove work - by argument.  O2V_HIP_FLAG_EXACT_CLIP
(include/o2v_hip.h) switches all of it off: every candidate of a leaf's clamped AABB goes through the plane-distance cull
and all six planes of splitTriangle's classification, as reference src/voxelization.cpp:383-424,446-470 does.  Here both
modes run on the same >= 1 M-triangle workloads - far beyond what the CPU oracle reaches in a test - and must produce
identical (x, y, z, argb) records; a sample of every workload is also compared with the oracle.
"""
import numpy as np
import pytest

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu

T_FULL = 1_000_000

# name, soup kind, triangles, resolution, supersampling, strategy, materials, z slab (output layers) or None
#   materials: "none" = MATERIALLESS (k_voxelize<false>, direct MAX path), "color" = UNTEXTURED, "tex" = a third of the
#   triangles textured (k_voxelize<true>; with MAX the pick variant), the others coloured / materialless
WORKLOADS = [
    ("small_1024_max", "small", 1_200_000, 1024, 1, 0, "none", None),
    ("sliver_2048_blend", "sliver", T_FULL, 2048, 1, 1, "color", None),
    ("huge_1024_max_tex", "huge", T_FULL, 1024, 1, 0, "tex", None),
    ("planar_1024_blend_tex", "planar", T_FULL, 1024, 1, 1, "tex", None),
    ("mixed_2048_ss2_max", "mixed", T_FULL, 2048, 2, 0, "color", None),
    ("mixed_4096_slab_blend_tex", "mixed", T_FULL, 4096, 1, 1, "tex", (1792, 2304)),
    ("planar_1024_ss2_max", "planar", T_FULL, 512, 2, 0, "none", None),
    # far from the origin (float32 has 8 .. 9 fraction bits): the margins that scale with the coordinates
    ("mixed_32768_far_slab_blend", "mixed", 400_000, 32768, 1, 1, "color", (20000, 20008)),
]


def _materials(kind, T, seed):
    rng = np.random.default_rng(900 + seed)
    if kind == "none":
        return {}, []
    if kind == "color":
        return dict(types=np.full(T, 2, np.uint32), colors=rng.random((T, 3)).astype(np.float32)), []
    types = rng.integers(1, 4, size=T).astype(np.uint32)
    mat = dict(types=types, colors=rng.random((T, 3)).astype(np.float32),
               uvs=(rng.random((T, 6)) * 2.5 - 0.7).astype(np.float32), texids=rng.integers(0, 2, size=T).astype(np.int32))
    textures = [(rng.integers(0, 256, size=(37, 53, 3)).astype(np.uint8), 1), (rng.integers(0, 256, size=(16, 8, 4)).astype(np.uint8), 0)]
    return mat, textures


def _subset(mat, idx):
    return {k: v[idx] for k, v in mat.items()}


@pytest.mark.parametrize("name,kind,T,res,ss,strategy,materials,zslab", WORKLOADS, ids=[w[0] for w in WORKLOADS])
def test_fast_equals_exact(oracle, name, kind, T, res, ss, strategy, materials, zslab):
    from obj2voxel_amd import hip
    seed = [w[0] for w in WORKLOADS].index(name)
    S = res * ss
    z_range = None if zslab is None else (zslab[0] * ss - 40.0, zslab[1] * ss + 40.0)
    if zslab is not None and zslab[1] - zslab[0] < 64:
        z_range = (zslab[0] * ss - 6.0, zslab[1] * ss + 6.0)   # a thin slab: keep most triangles inside it
    v = meshes.stress_soup(kind, T, S, seed=seed, z_range=z_range)
    mat, textures = _materials(materials, len(v), seed)
    kw = dict(supersampling=ss, strategy=strategy, bounds=meshes.stress_bounds(S))
    if zslab is not None:
        kw["zslab"] = zslab
    d = hip.DeviceVoxelizer(0)   # own context: the grids of the large resolutions are released again
    try:
        d.set_textures(textures)
        d.set_triangles(v, **mat)
        fast = meshes.sorted_voxels(d.voxelize(res, **kw))
        st_fast = d.stats()
        exact = meshes.sorted_voxels(d.voxelize(res, exact_clip=True, **kw))
        st_exact = d.stats()
        # the switch really changes the kernel's work: without the row test every candidate of the AABBs is a job candidate
        assert st_exact["jobs"] > st_fast["jobs"], (st_fast, st_exact)
        # (occupancy-only mode - the fast side of a mesh without materials - does not run the jobs of voxels that are marked
        # already, so it counts fewer hits, by at most the jobs it skipped)
        assert st_fast["hits"] <= st_exact["hits"] <= st_fast["hits"] + st_fast["skipped_jobs"], (name, st_fast, st_exact)
        assert len(fast) > (1_000_000 if T >= 1_000_000 else 200_000), len(fast)
        assert fast.shape == exact.shape, (name, fast.shape, exact.shape)
        assert np.array_equal(fast, exact), name
        # ... and a sample against the oracle (every 40th triangle: what the CPU finishes in seconds)
        idx = np.arange(0, len(v), 40)
        d.set_triangles(v[idx], **_subset(mat, idx))
        got = meshes.sorted_voxels(d.voxelize(res, **kw))
    finally:
        d.close()
    oracle.set_threads(32)
    try:
        want = meshes.sorted_voxels(oracle.voxelize(v[idx], res, textures=textures, **_subset(mat, idx), **kw))
    finally:
        oracle.set_threads(1)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.array_equal(got, want), name


def test_exact_flag_via_environment(monkeypatch):
    """O2V_EXACT_CLIP=1 forces the same mode for callers that cannot set the flag (obj2voxel_voxelize(), the CLI)."""
    from obj2voxel_amd import hip
    v = meshes.stress_soup("mixed", 40_000, 256, seed=11)
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(v)
        kw = dict(bounds=meshes.stress_bounds(256))
        fast = meshes.sorted_voxels(d.voxelize(256, **kw))
        jobs_fast = d.stats()["jobs"]
        monkeypatch.setenv("O2V_EXACT_CLIP", "1")
        exact = meshes.sorted_voxels(d.voxelize(256, **kw))
        jobs_env = d.stats()["jobs"]
        monkeypatch.delenv("O2V_EXACT_CLIP")
        flagged = meshes.sorted_voxels(d.voxelize(256, exact_clip=True, **kw))
        assert d.stats()["jobs"] == jobs_env > jobs_fast
    finally:
        d.close()
    assert np.array_equal(fast, exact) and np.array_equal(fast, flagged)


def _sorted_hits(h):
    order = np.lexsort((h[:, 4], h[:, 3], h[:, 2], h[:, 1], h[:, 0]))
    return h[order]


@pytest.mark.parametrize("uv_scale", [2.0 ** -60, 2.0 ** -22, 1.0, 2.0 ** 19, 2.0 ** 21], ids=["uv2^-60", "uv2^-22", "uv1", "uv2^19", "uv2^21"])
def test_hit_records_equal_in_exact_mode(uv_scale):
    """Not only the voxels: every (leaf, voxel) hit's weight and uv mean, bit for bit.  The clip loop's shortened divisions
    (third(), the lean forms of o2v_dev_arith.hpp) and its work-removal rules are all off in exact mode, so the two runs'
    hit records - cell, triangle, leaf order key, w, u, v - must be the same multiset.  The uv scales put the running uv mean
    on both sides of the lean forms' limits (|uv| <= 2^20 for the lean path at all; quotients below 2^-50 repeat the division
    the long way), the tiny triangles near the origin have areas below 2^-30 (no lean path either)."""
    from obj2voxel_amd import hip
    S, T = 512, 100_000
    v = meshes.stress_soup("mixed", T, S, seed=23)
    rng = np.random.default_rng(77)
    # 2 000 triangles of ~1e-4 x 1e-7 voxels inside the first voxels (areas ~1e-12 .. 1e-11 < 2^-30)
    tiny_c = rng.random((2000, 1, 3)) * 3.0 + 0.2
    tiny = (tiny_c + np.concatenate([np.zeros((2000, 1, 3)), rng.random((2000, 1, 3)) * 1e-4, rng.random((2000, 1, 3)) * 1e-7 + 1e-8], axis=1)).astype(np.float32)
    v = np.concatenate([v.reshape(-1, 9), tiny.reshape(-1, 9)])
    n = len(v)
    types = rng.integers(1, 4, size=n).astype(np.uint32)
    uvs = ((rng.random((n, 6)) * 2.5 - 0.7) * uv_scale).astype(np.float32)
    uvs[rng.random(n) < 0.05] = 0.0          # +0 numerators
    mat = dict(types=types, colors=rng.random((n, 3)).astype(np.float32), uvs=uvs, texids=rng.integers(0, 2, size=n).astype(np.int32))
    textures = [(rng.integers(0, 256, size=(37, 53, 3)).astype(np.uint8), 1), (rng.integers(0, 256, size=(16, 8, 4)).astype(np.uint8), 0)]
    kw = dict(strategy=1, bounds=meshes.stress_bounds(S))
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_textures(textures)
        d.set_triangles(v, **mat)
        n_fast = d.voxelize(S, read=False, **kw)
        fast = _sorted_hits(d.hits())
        n_exact = d.voxelize(S, read=False, exact_clip=True, **kw)
        exact = _sorted_hits(d.hits())
    finally:
        d.close()
    assert n_fast == n_exact and len(fast) == len(exact) > 600_000, (n_fast, n_exact, len(fast), len(exact))
    same = (fast == exact).all(axis=1)
    assert same.all(), (int((~same).sum()), fast[~same][:3], exact[~same][:3])


@pytest.mark.parametrize("kind,ss", [("mixed", 1), ("planar", 2)], ids=["mixed", "planar_ss2"])
def test_hit_weights_equal_in_exact_mode_untextured(kind, ss):
    """The same for k_voxelize<false>, whose fast mode settles pieces with one plane left from their classification alone
    (single_plane: the number of final pieces, hence the weight, without cutting): every hit's weight, bit for bit."""
    from obj2voxel_amd import hip
    res = 512
    S = res * ss
    v = meshes.stress_soup(kind, 200_000, S, seed=31)
    rng = np.random.default_rng(78)
    mat = dict(types=np.full(len(v), 2, np.uint32), colors=rng.random((len(v), 3)).astype(np.float32))
    kw = dict(strategy=1, supersampling=ss, bounds=meshes.stress_bounds(S))
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(v, **mat)
        d.voxelize(res, read=False, **kw)
        fast = _sorted_hits(d.hits())
        d.voxelize(res, read=False, exact_clip=True, **kw)
        exact = _sorted_hits(d.hits())
    finally:
        d.close()
    assert len(fast) == len(exact) > 1_000_000, (len(fast), len(exact))
    same = (fast == exact).all(axis=1)
    assert same.all(), (int((~same).sum()), fast[~same][:3], exact[~same][:3])


def test_root_stage_left_out_for_small_triangles(monkeypatch, oracle):
    """Occupancy only, every triangle less than five voxels across: k_expand_roots is not launched, k_voxelize_occ makes and counts
    the root leaves (Params::solo_roots).  Same voxels and the same leaf / tile / candidate statistics as with the root stage
    (O2V_NO_SOLO_ROOTS=1), on the whole grid and on a z-slab; checked against the oracle."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(150)              # 89 400 triangles, ~3.4 voxels across at 320^3
    want = meshes.sorted_voxels(oracle.voxelize(v, 320))
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(v)
        runs = {}
        for solo in (True, False):
            if solo:
                monkeypatch.delenv("O2V_NO_SOLO_ROOTS", raising=False)
            else:
                monkeypatch.setenv("O2V_NO_SOLO_ROOTS", "1")
            whole = meshes.sorted_voxels(d.voxelize(320))
            st, kt = d.stats(), None
            d.voxelize(320, read=False, kernel_times=True)
            kt = d.kernel_times()
            slab = meshes.sorted_voxels(d.voxelize(320, zslab=(96, 200)))
            runs[solo] = (whole, {k: st[k] for k in ("leaves", "tiles", "bypassed_leaves", "candidates", "voxels")}, set(kt), slab, d.timings()["passes"])
        assert "k_expand_roots" not in runs[True][2] and "k_expand_roots" in runs[False][2]
        assert "k_voxelize_occ" in runs[True][2]
        assert np.array_equal(runs[True][0], want) and np.array_equal(runs[False][0], want)
        assert runs[True][1] == runs[False][1] and runs[True][1]["bypassed_leaves"] == len(v)
        assert np.array_equal(runs[True][3], runs[False][3]) and np.array_equal(runs[True][3], want[(want[:, 2] >= 96) & (want[:, 2] < 200)])
        assert runs[True][4] == 1
    finally:
        d.close()


def test_root_stage_comes_back_when_a_triangle_needs_it(monkeypatch, oracle):
    """The premise of solo_roots is checked per triangle on the device: with the hint overridden (test hook) on a mesh of large
    triangles, the first pass reports kErrSoloRoots and the call repeats it with k_expand_roots - same voxels as the oracle."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(12)               # triangles tens of voxels across: subdivided
    want = meshes.sorted_voxels(oracle.voxelize(v, 200))
    monkeypatch.setenv("O2V_TEST_FORCE_SOLO_ROOTS", "1")
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(v)
        got = meshes.sorted_voxels(d.voxelize(200))
        assert d.timings()["passes"] >= 2
        assert np.array_equal(got, want)
        got = meshes.sorted_voxels(d.voxelize(200))      # remembered for this mesh and these settings: one pass, with the root stage
        assert d.timings()["passes"] == 1 and np.array_equal(got, want)
    finally:
        d.close()
