"""Parity of the HIP path against the CPU oracle, through the device C-ABI (include/o2v_hip.h).

Bar: bit-exact occupancy and bit-exact ARGB for both strategies (the device replays every voxel's hits in the
reference's sequential order, so even BLEND is expected to match exactly; BASELINE.json allows 1 LSB per
channel for blended colour, asserted as the fallback bound below).
"""
import numpy as np
import pytest

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dv():
    from obj2voxel_amd import hip
    d = hip.DeviceVoxelizer(0)
    yield d
    d.close()


def _compare(got, want, blend_tolerance=False):
    got, want = meshes.sorted_voxels(got), meshes.sorted_voxels(want)
    assert got.shape == want.shape, f"voxel count {got.shape[0]} != oracle {want.shape[0]}"
    assert np.array_equal(got[:, :3], want[:, :3]), "occupancy differs"
    if np.array_equal(got[:, 3], want[:, 3]):
        return
    if blend_tolerance:
        ga = got[:, 3:4] >> np.array([24, 16, 8, 0], dtype=np.uint32) & 255
        wa = want[:, 3:4] >> np.array([24, 16, 8, 0], dtype=np.uint32) & 255
        assert np.abs(ga.astype(int) - wa.astype(int)).max() <= 1
    else:
        bad = np.flatnonzero(got[:, 3] != want[:, 3])
        raise AssertionError(f"{len(bad)} colours differ, first: {got[bad[0]]} vs {want[bad[0]]}")


def _run_both(dv, oracle, verts, res, **kw):
    from obj2voxel_amd import hip  # noqa: F401
    uvs, types, colors = kw.get("uvs"), kw.get("types"), kw.get("colors")
    texids, textures = kw.get("texids"), kw.get("textures", ())
    if textures:
        dv.set_textures(list(textures))
    dv.set_triangles(verts, uvs=uvs, types=types, colors=colors, texids=texids)
    run = dict(supersampling=kw.get("supersampling", 1), strategy=kw.get("strategy", 0),
               unit_transform=kw.get("unit_transform"), bounds=kw.get("bounds"), zslab=kw.get("zslab", (0, 0)))
    got = dv.voxelize(res, **run)
    want = oracle.voxelize(verts, res, uvs=uvs, types=types, colors=colors, texids=texids, textures=textures, **run)
    return got, want


@pytest.mark.parametrize("res", [32, 64, 128])
def test_unit_cube(dv, oracle, res):
    got, want = _run_both(dv, oracle, meshes.unit_cube(), res)
    _compare(got, want)
    assert len(got) == 8 + 12 * (res - 2) + 6 * (res - 2) ** 2  # reference test/main.cpp:120-126


@pytest.mark.parametrize("res", [32, 128])
def test_three_planes(dv, oracle, res):
    got, want = _run_both(dv, oracle, meshes.three_planes(), res)
    _compare(got, want)
    assert len(got) == 3 * res * res  # reference test/main.cpp:225-252


@pytest.mark.parametrize("strategy", [0, 1])
@pytest.mark.parametrize("nv,res", [(9, 48), (9, 100), (12, 256), (40, 200), (5, 300)])
def test_colored_sphere(dv, oracle, nv, res, strategy):
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(nv)
    T = len(v)
    got, want = _run_both(dv, oracle, v, res, types=np.full(T, hip.TRI_UNTEXTURED, np.uint32),
                          colors=meshes.triangle_colors(T), strategy=strategy)
    _compare(got, want, blend_tolerance=False)


@pytest.mark.parametrize("strategy", [0, 1])
@pytest.mark.parametrize("wrap", [0, 1])
def test_textured_sphere(dv, oracle, strategy, wrap):
    from obj2voxel_amd import hip
    v, uv = meshes.uv_sphere(14, with_uv=True)
    uv = uv * 1.5 - 0.2  # leave [0,1] so that wrap / clamp matter
    T = len(v)
    tex = [(meshes.checker_texture(64, 8), wrap)]
    got, want = _run_both(dv, oracle, v, 160, uvs=uv, types=np.full(T, hip.TRI_TEXTURED, np.uint32),
                          texids=np.zeros(T, np.int32), textures=tex, strategy=strategy)
    _compare(got, want)


def test_mixed_materials_two_textures(dv, oracle):
    v, uv = meshes.uv_sphere(12, with_uv=True)
    T = len(v)
    types = (np.arange(T) % 3 + 1).astype(np.uint32)
    argb = np.concatenate([np.full((32, 32, 1), 255, np.uint8), meshes.checker_texture(32, 4)], axis=2)
    tex = [(meshes.checker_texture(64, 8), 1), (argb, 0)]
    got, want = _run_both(dv, oracle, v, 128, uvs=uv, types=types, colors=meshes.triangle_colors(T),
                          texids=(np.arange(T) % 2).astype(np.int32), textures=tex, strategy=1)
    _compare(got, want)


@pytest.mark.parametrize("strategy", [0, 1])
def test_random_soup(dv, oracle, strategy):
    from obj2voxel_amd import hip
    v = meshes.random_soup(400, seed=7, scale=0.5)
    T = len(v)
    got, want = _run_both(dv, oracle, v, 150, types=np.full(T, hip.TRI_UNTEXTURED, np.uint32),
                          colors=meshes.triangle_colors(T), strategy=strategy)
    _compare(got, want)


def test_box_room_large_aligned(dv, oracle):
    got, want = _run_both(dv, oracle, meshes.box_room(2), 256)
    _compare(got, want)


@pytest.mark.parametrize("strategy", [0, 1])
def test_supersampling(dv, oracle, strategy):
    from obj2voxel_amd import hip
    v, uv = meshes.uv_sphere(12, with_uv=True)
    T = len(v)
    got, want = _run_both(dv, oracle, v, 64, uvs=uv, types=np.full(T, hip.TRI_TEXTURED, np.uint32),
                          texids=np.zeros(T, np.int32), textures=[(meshes.checker_texture(64, 8), 1)],
                          strategy=strategy, supersampling=2)
    _compare(got, want)


def test_unit_transform_and_bounds(dv, oracle):
    v = meshes.uv_sphere(10, radius=0.7, center=(0.1, -0.2, 0.3))
    got, want = _run_both(dv, oracle, v, 90, unit_transform=[0, 0, 1, 0, -1, 0, 1, 0, 0],
                          bounds=[-1, -1, -1, 1, 1, 1])
    _compare(got, want)


@pytest.mark.parametrize("unit", [[1, 0, 0, 0, 1, 0, 0, 0, 1], [0, -1, 0, 0, 0, -1, -1, 0, 0], [0, 0, 1, -1, 0, 0, 0, 1, 0], [-1, 0, 0, 0, -1, 0, 0, 0, -1]])
@pytest.mark.parametrize("mode", ["blend_ss2", "max", "occupancy"])
def test_grid_covers_the_mesh_box_wherever_it_lies(dv, oracle, unit, mode):
    """The dense grids are allocated for the mesh's voxel bounding box, not the cube (o2v_hip_voxelize: grid_box).  A long thin
    mesh, turned and mirrored by the unit transform so that its box lies along every axis and at either end of the grid (mirrored
    axes put it at the far end: a grid origin above zero in x, y and z), whole and in z-slabs that cut the box or miss it."""
    from obj2voxel_amd import hip
    v, uv = meshes.readme_blade()
    v, uv = v[::3], uv[::3]
    T = len(v)
    res = 300
    if mode == "blend_ss2":
        kw = dict(uvs=uv, types=np.full(T, hip.TRI_TEXTURED, np.uint32), texids=np.zeros(T, np.int32),
                  textures=[(meshes.checker_texture(64, 8), 1)], strategy=1, supersampling=2)
    elif mode == "max":
        kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=0)
    else:
        kw = {}
    got, want = _run_both(dv, oracle, v, res, unit_transform=unit, **kw)
    _compare(got, want)
    assert len(want) > 3000
    st = dv.stats()
    assert st["grid_cells"] < res ** 3 // 20, st        # a twelfth of the cube's width in two axes
    # slabs: one inside the box, one cutting its end, one that misses it (no voxels, no error)
    zs = np.sort(meshes.sorted_voxels(want)[:, 2])
    zmid = int(zs[len(zs) // 2])
    parts = []
    for zslab in ((0, max(zmid - 5, 1)), (max(zmid - 5, 1), zmid + 7), (zmid + 7, res)):
        g, w = _run_both(dv, oracle, v, res, unit_transform=unit, zslab=zslab, **kw)
        _compare(g, w)
        parts.append(g)
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), meshes.sorted_voxels(got))


def test_no_max_grid_for_a_mesh_of_large_triangles(dv, oracle):
    """MAX strategy: the direct path's 64-bit grid (two thirds of the grids' memory) is not allocated when three quarters of
    the triangles are large enough to be subdivided anyway (grid_modes: the triangles' extents are known since the upload) -
    the sort-and-replay route computes the same voxels.  A fine mesh keeps it."""
    from obj2voxel_amd import hip
    for nv, res, direct in ((12, 384, False), (60, 96, True)):
        v = meshes.uv_sphere(nv)
        T = len(v)
        kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=0)
        got, want = _run_both(dv, oracle, v, res, **kw)
        _compare(got, want)
        st = dv.stats()
        assert (st["direct_hits"] > 0) == direct, st
        per_cell = st["grid_bytes"] / st["grid_cells"]
        assert (per_cell > 12.0) == direct and per_cell > 4.0, (per_cell, st)   # 4 + 8 bytes per cell (+ flags), or 4


@pytest.mark.parametrize("with_large", [False, True])
def test_count_roots_ahead_of_expand_roots(oracle, monkeypatch, with_large):
    """A tessellated material-less surface in z-slabs: k_count_roots counts the one-tile root triangles and lists the blocks that
    hold anything else for k_expand_roots (o2v_hip_voxelize; O2V_COUNT_ROOTS=1 takes that route on the whole grid as well).  The
    voxels are those of the route without it and the oracle's; `with_large`: a few triangles that need subdivision (listed blocks)."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(260)                      # 270 k triangles, ~2.4 voxels across at 400^3
    if with_large:
        big = meshes.uv_sphere(6) * 0.5            # 140 triangles, ~50 voxels across: fewer than one in 512
        v = np.concatenate([v[:100_000], big[:70], v[100_000:], big[70:]]).astype(np.float32)
    res = 400
    want = meshes.sorted_voxels(oracle.voxelize(v, res))
    outs = {}
    # (a mesh without a triangle of five voxels would skip the root stage altogether - Params::solo_roots,
    # tests/test_gpu_exact_ab.py::test_root_stage_left_out_for_small_triangles; this test is about the stage itself)
    monkeypatch.setenv("O2V_NO_SOLO_ROOTS", "1")
    for mode in ("default", "always", "never"):
        monkeypatch.delenv("O2V_COUNT_ROOTS", raising=False)
        monkeypatch.delenv("O2V_NO_COUNT_ROOTS", raising=False)
        if mode == "always":
            monkeypatch.setenv("O2V_COUNT_ROOTS", "1")
        if mode == "never":
            monkeypatch.setenv("O2V_NO_COUNT_ROOTS", "1")
        d = hip.DeviceVoxelizer(0)
        try:
            d.set_triangles(v)
            whole = meshes.sorted_voxels(d.voxelize(res, kernel_times=True))
            ran_whole = "k_count_roots" in d.kernel_times()
            st = d.stats()
            cuts, bnd = d.plan_slabs(res, 3)
            parts = []
            ran_slab = True
            for r in range(3):
                parts.append(d.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, kernel_times=True))
                ran_slab = ran_slab and "k_count_roots" in d.kernel_times()
        finally:
            d.close()
        assert np.array_equal(whole, want), mode
        assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), want), mode
        assert ran_whole == (mode == "always") and ran_slab == (mode != "never"), (mode, ran_whole, ran_slab)
        assert st["bypassed_leaves"] > 0.99 * 270_000 and st["leaves"] >= st["bypassed_leaves"] + (100 if with_large else 0), st
        outs[mode] = st
    assert outs["always"]["leaves"] == outs["never"]["leaves"] and outs["always"]["candidates"] == outs["never"]["candidates"]


def test_non_multiple_of_four_resolution(dv, oracle):
    got, want = _run_both(dv, oracle, meshes.uv_sphere(8), 77)
    _compare(got, want)


def test_degenerate_and_tiny_triangles(dv, oracle):
    v = np.array([[0, 0, 0, 1, 1, 1, 0, 0, 0],            # zero area (two equal vertices)
                  [0, 0, 0, 0.5, 0.5, 0.5, 1, 1, 1],      # collinear
                  [0.2, 0.2, 0.2, 0.2001, 0.2, 0.2, 0.2, 0.2001, 0.2],  # sub-voxel
                  [0, 0, 0, 1, 0, 0, 0, 1, 0],
                  [0, 0, 1, 1, 0, 1, 1, 1, 0.3]], dtype=np.float32)
    got, want = _run_both(dv, oracle, v, 64)
    _compare(got, want)


def test_slabs_union_equals_whole(dv, oracle):
    """z-slab sharding: every slab equals the matching subset of the full run."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(16)
    T = len(v)
    kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=1)
    full, want = _run_both(dv, oracle, v, 128, **kw)
    _compare(full, want)
    parts = []
    for z in range(0, 128, 32):
        got, want_slab = _run_both(dv, oracle, v, 128, zslab=(z, z + 32), **kw)
        _compare(got, want_slab)
        parts.append(got)
    _compare(np.concatenate(parts), full)


@pytest.mark.parametrize("n_slabs", [2, 3, 8])
def test_planned_slabs_union_equals_whole_and_balance(dv, oracle, n_slabs):
    """o2v_hip_plan_slabs: the work-balanced cuts partition [0, res); the union of the planned slabs equals the whole
    result bit for bit (and the oracle's); the slabs' hit counts are balanced to a few per cent on a mesh whose
    equal-height slabs are not (a sphere with its poles on the y axis and a dense patch of extra triangles)."""
    from obj2voxel_amd import hip
    res = 192
    sphere = meshes.uv_sphere(90)
    patch = meshes.uv_sphere(60, radius=0.08, center=(0.0, 0.0, 0.7))  # a dense blob near the top of z
    v = np.concatenate([np.reshape(sphere, (-1, 9)), np.reshape(patch, (-1, 9))])
    T = len(v)
    kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=1)
    got, want = _run_both(dv, oracle, v, res, **kw)
    _compare(got, want)
    whole_hits = dv.stats()["hits"]
    cuts, bnd = dv.plan_slabs(res, n_slabs)
    assert cuts[0] == 0 and cuts[-1] == res and all(a < b for a, b in zip(cuts, cuts[1:]))
    assert np.array_equal(bnd, np.concatenate([v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)]).astype(np.float32))
    parts, hits, work = [], [], []
    for k in range(n_slabs):
        parts.append(dv.voxelize(res, strategy=1, zslab=(cuts[k], cuts[k + 1]), bounds=bnd))
        hits.append(dv.stats()["hits"])
        work.append(dv.stats()["hits"] + 4.0 * dv.stats()["leaves"])   # what the plan balances: hits + 4 per leaf (k_zhist)
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), meshes.sorted_voxels(got))
    assert sum(hits) == whole_hits
    assert max(work) < 1.08 * sum(work) / n_slabs, (cuts, hits, work)
    equal = []
    for k in range(n_slabs):
        z0, z1 = k * res // n_slabs, (k + 1) * res // n_slabs
        dv.voxelize(res, strategy=1, zslab=(z0, z1), read=False)
        equal.append(dv.stats()["hits"] + 4.0 * dv.stats()["leaves"])
    assert max(equal) > max(work)


def test_planned_slabs_edge_cases(dv):
    """Empty mesh -> equal heights; one layer per slab; supersampling; a single triangle (all work in one layer)."""
    from obj2voxel_amd import hip
    dv.set_triangles(np.zeros((0, 9), np.float32))
    cuts, _ = dv.plan_slabs(64, 4)
    assert cuts == [0, 16, 32, 48, 64]
    dv.set_triangles(meshes.uv_sphere(12))
    cuts, _ = dv.plan_slabs(16, 16)
    assert cuts == list(range(17))
    cuts, _ = dv.plan_slabs(100, 4, supersampling=2)
    assert cuts[0] == 0 and cuts[-1] == 100 and all(a < b for a, b in zip(cuts, cuts[1:]))
    assert abs(cuts[2] - 50) <= 1                      # the sphere is symmetric in z
    flat = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1, 1, 0, 1, 0, 1, 1]], np.float32)  # two z-flat triangles
    dv.set_triangles(flat)
    cuts, _ = dv.plan_slabs(64, 4)
    assert cuts[0] == 0 and cuts[-1] == 64 and all(a < b for a, b in zip(cuts, cuts[1:]))
    with pytest.raises(hip.DeviceError):
        dv.plan_slabs(8, 9)                             # more slabs than layers


def test_context_reuse_is_idempotent(dv, oracle):
    v = meshes.uv_sphere(9)
    a, want = _run_both(dv, oracle, v, 80)
    b = dv.voxelize(80)
    c = dv.voxelize(40)
    d = dv.voxelize(80)
    _compare(a, want)
    _compare(b, a)
    _compare(d, a)
    _compare(c, oracle.voxelize(v, 40))


def test_golden_fixtures(dv, oracle):
    """Committed vectors (tests/golden/oracle_regression.npz, produced by the oracle) against the device."""
    import os
    from tests.golden.make_golden import CASES

    class DeviceAsOracle:
        TRI_UNTEXTURED, TRI_TEXTURED, STRATEGY_MAX, STRATEGY_BLEND = 2, 3, 0, 1

        @staticmethod
        def voxelize(verts, res, uvs=None, types=None, colors=None, texids=None, textures=(), **kw):
            if textures:
                dv.set_textures(list(textures))
            dv.set_triangles(verts, uvs=uvs, types=types, colors=colors, texids=texids)
            return dv.voxelize(res, **kw)

    data = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_regression.npz"))
    for name in CASES:
        _compare(CASES[name](DeviceAsOracle), data[name])


def test_full_size_properties(dv):
    """BASELINE.json-size grids (1024^3) through size-independent properties the reference's tests define."""
    res = 1024
    dv.set_triangles(meshes.unit_cube())
    vox = dv.voxelize(res)
    assert len(vox) == 8 + 12 * (res - 2) + 6 * (res - 2) ** 2
    xyz = vox[:, :3]
    assert ((xyz == 0) | (xyz == res - 1)).any(axis=1).all()
    assert len(np.unique(xyz[:, 0].astype(np.uint64) << 40 | xyz[:, 1].astype(np.uint64) << 20 | xyz[:, 2])) == len(vox)
    dv.set_triangles(meshes.three_planes())
    assert dv.voxelize(res, read=False) == 3 * res * res
    # slab union == whole, determinism
    v = meshes.uv_sphere(120)
    dv.set_triangles(v)
    whole = meshes.sorted_voxels(dv.voxelize(res))
    again = meshes.sorted_voxels(dv.voxelize(res))
    assert np.array_equal(whole, again)
    parts = [dv.voxelize(res, zslab=(z, z + 256)) for z in range(0, res, 256)]
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), whole)


@pytest.mark.parametrize("strategy", [0, 1])
def test_long_hit_lists(dv, oracle, strategy):
    """Many triangles per voxel: exercises the cooperative resolve tiers (wavefront and workgroup LDS sorts and
    the global-memory sort for lists longer than 2048 hits)."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(60)          # 14160 triangles
    T = len(v)
    kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=strategy)
    for res in (24, 6, 2):            # up to thousands of hits per voxel at the coarse end
        got, want = _run_both(dv, oracle, v, res, **kw)
        _compare(got, want)


@pytest.mark.parametrize("strategy", [0, 1])
def test_long_hit_lists_with_supersampling(dv, oracle, strategy):
    """Crowded cells whose hits are spread over the eight sub-voxels of 2x supersampling: the cooperative tiers fold every
    sub-voxel's chain on its own and combine the eight results in ascending order (downscale, voxelization.hpp:82-85) -
    BLEND runs the sub-voxels' chains side by side on the lanes of one wavefront (blend_chain_lds).  Textured and coloured
    triangles mixed, so the chains' colours differ per group."""
    from obj2voxel_amd import hip
    v, uv = meshes.uv_sphere(60, with_uv=True)          # 14160 triangles
    T = len(v)
    types = np.where(np.arange(T) % 3 == 0, hip.TRI_TEXTURED, hip.TRI_UNTEXTURED).astype(np.uint32)
    kw = dict(uvs=uv, types=types, colors=meshes.triangle_colors(T), texids=np.zeros(T, np.int32),
              textures=[(meshes.checker_texture(64, 8), 1)], strategy=strategy, supersampling=2)
    for res in (20, 8, 3, 1):            # tens to thousands of hits per output voxel
        got, want = _run_both(dv, oracle, v, res, **kw)
        _compare(got, want)


@pytest.mark.parametrize("strategy", [0, 1])
def test_big_and_huge_hit_lists(dv, oracle, strategy):
    """39 600 triangles into 8 cells (~5-10 k hits each: the dynamic-LDS tier, 2049..8192 hits) and into a single
    cell (> 8192 hits: the global-memory sort)."""
    from obj2voxel_amd import hip
    v = meshes.uv_sphere(100)
    T = len(v)
    kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=strategy)
    for res in (3, 2, 1):
        got, want = _run_both(dv, oracle, v, res, **kw)
        _compare(got, want)
        st = dv.stats()
        assert st["hits"] >= T
    assert st["voxels"] == 1 and st["hits"] > 8192


def test_baseline_config2_spot_512_blend_textured(dv, oracle):
    """BASELINE.json configs[1]: 'Spot cow at 512^3, weighted-blend, 1xMI355X' with the survey's stand-in
    (uv-sphere nv=39 -> 5928 triangles, textured). Full size, bit-exact against the oracle."""
    from obj2voxel_amd import hip
    v, uv = meshes.uv_sphere(39, with_uv=True)
    T = len(v)
    oracle.set_threads(8)
    try:
        got, want = _run_both(dv, oracle, v, 512, uvs=uv, types=np.full(T, hip.TRI_TEXTURED, np.uint32),
                              texids=np.zeros(T, np.int32), textures=[(meshes.checker_texture(256, 16), 1)], strategy=1)
    finally:
        oracle.set_threads(1)
    _compare(got, want)
    assert len(got) > 1_200_000


def test_baseline_config4_like_supersampled_room(dv, oracle):
    """BASELINE.json configs[3] ('Sponza textured at 2048^3 with 2x supersampling') at a size the oracle finishes in
    seconds: large axis-aligned textured quads + a sphere, 2x supersampling, BLEND. Bit-exact."""
    from obj2voxel_amd import hip
    room = meshes.box_room(3)
    sph, suv = meshes.uv_sphere(24, radius=0.3, center=(0.5, 0.45, 0.55), with_uv=True)
    v = np.concatenate([room, sph])
    ruv = np.tile(np.array([0, 0, 1, 0, 1, 1], np.float32), (len(room), 1))
    uv = np.concatenate([ruv, suv])
    T = len(v)
    got, want = _run_both(dv, oracle, v, 160, uvs=uv, types=np.full(T, hip.TRI_TEXTURED, np.uint32),
                          texids=np.zeros(T, np.int32), textures=[(meshes.checker_texture(128, 8), 1)], strategy=1,
                          supersampling=2)
    _compare(got, want)


def test_full_size_supersampled_2048(dv):
    """BASELINE.json configs[3] at its full grid (2048^3 output, 4096^3 samples): size-independent properties.
    A cube's 2x-supersampled result equals the plain one, runs are deterministic and the
    z-slabs tile the result."""
    res = 2048
    dv.set_triangles(meshes.unit_cube())
    plain = meshes.sorted_voxels(dv.voxelize(res))
    ss = meshes.sorted_voxels(dv.voxelize(res, supersampling=2))
    assert len(plain) == 8 + 12 * (res - 2) + 6 * (res - 2) ** 2
    assert np.array_equal(plain, ss)
    v = np.concatenate([meshes.box_room(4), meshes.uv_sphere(200, radius=0.3, center=(0.5, 0.5, 0.5))])
    dv.set_triangles(v)
    whole = meshes.sorted_voxels(dv.voxelize(res, supersampling=2, strategy=1))
    parts = [dv.voxelize(res, supersampling=2, strategy=1, zslab=(z, z + 512)) for z in range(0, res, 512)]
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), whole)
    assert np.array_equal(meshes.sorted_voxels(dv.voxelize(res, supersampling=2, strategy=1)), whole)


@pytest.mark.parametrize("coloured", [False, True])
def test_4096_grid_in_eight_slabs(dv, coloured):
    """BASELINE.json configs[4] layout: a 4096^3 grid split into 8 z-slabs, here run one after the other on a single GPU.
    Material-less (the configs[4] mesh): one byte per cell, 8.6 GB per slab; with colours and BLEND the 32-bit counter grid,
    34 GB per slab.  The slab counts of the unit cube must add up to the reference's closed form (test/main.cpp:120-126) and
    every voxel must lie in its slab."""
    res, n = 4096, 8
    v = meshes.unit_cube()
    kw = dict(types=np.full(len(v), 2, np.uint32), colors=meshes.triangle_colors(len(v))) if coloured else {}
    dv.set_triangles(v, **kw)
    total = 0
    for r in range(n):
        z0, z1 = r * res // n, (r + 1) * res // n
        vox = dv.voxelize(res, zslab=(z0, z1), strategy=1 if coloured else 0)
        assert ((vox[:, 2] >= z0) & (vox[:, 2] < z1)).all()
        total += len(vox)
    assert total == 8 + 12 * (res - 2) + 6 * (res - 2) ** 2
    st = dv.stats()
    assert st["grid_bytes"] > (34e9 if coloured else 8.5e9) and st["grid_bytes"] < (40e9 if coloured else 10e9)


def test_buffer_overflow_regrow_paths(oracle, monkeypatch):
    """Every device buffer (leaves, tiles, big-leaf list, subdivision queue, hit pool, occupied cells, sort scratch)
    starts tiny (test hook O2V_TEST_TINY_BUFFERS): the pipeline must detect each overflow, grow and re-run until the
    result is complete - and bit-identical to the oracle."""
    from obj2voxel_amd import hip
    monkeypatch.setenv("O2V_TEST_TINY_BUFFERS", "1")
    d = hip.DeviceVoxelizer(0)
    try:
        v = np.concatenate([meshes.uv_sphere(7), meshes.box_room(1) * 0.9 + 0.05])   # subdivision + big aligned leaves
        T = len(v)
        kw = dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=1)
        d.set_triangles(v, **kw_mat(kw))
        got = d.voxelize(200, strategy=1)
        assert d.timings()["passes"] > 1
        _compare(got, oracle.voxelize(v, 200, **kw))
        again = d.voxelize(200, strategy=1)          # capacities persist: one pass now
        assert d.timings()["passes"] == 1
        _compare(again, got)
        # thousands of hits in one cell with tiny buffers: the global-sort tier's scratch is allocated on demand
        dense = meshes.uv_sphere(40)
        Td = len(dense)
        kd = dict(types=np.full(Td, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(Td), strategy=1)
        d.set_triangles(dense, **kw_mat(kd))
        _compare(d.voxelize(2, strategy=1), oracle.voxelize(dense, 2, **kd))
    finally:
        d.close()


def kw_mat(kw):
    return {k: v for k, v in kw.items() if k in ("types", "colors", "uvs", "texids")}


@pytest.mark.parametrize("supersampling", [1, 2])
def test_direct_max_path_mixes_whole_and_subdivided_triangles(dv, oracle, supersampling):
    """MAX strategy without textured triangles: hits of unsplit triangles go straight into the 64-bit max grid, hits of
    subdivided ones through the pool / sort / replay, and both meet in the same cells - small triangles (one leaf each)
    scattered over a few large ones (subdivided), equal weights included (duplicates: ties go to the lower index)."""
    from obj2voxel_amd import hip
    rng = np.random.default_rng(77)
    big = rng.random((6, 3, 3)).astype(np.float32)                                # subdivided at this resolution
    c = rng.random((900, 1, 3)).astype(np.float32)
    small = np.clip(c + 0.03 * (rng.random((900, 3, 3)).astype(np.float32) - 0.5), 0, 1)
    v = np.concatenate([small[:450], big, small[450:], small[:40]]).reshape(-1, 9)   # duplicates at the end
    T = len(v)
    types = np.where(np.arange(T) % 3 == 0, hip.TRI_MATERIALLESS, hip.TRI_UNTEXTURED).astype(np.uint32)
    kw = dict(types=types, colors=rng.random((T, 3)).astype(np.float32), strategy=0, supersampling=supersampling)
    got, want = _run_both(dv, oracle, v, 96, **kw)
    _compare(got, want)
    st = dv.stats()
    assert 0 < st["direct_hits"] < st["hits"]            # both routes were taken
    # the same mesh with BLEND never uses the direct path
    kw["strategy"] = 1
    got, want = _run_both(dv, oracle, v, 96, **kw)
    _compare(got, want)
    assert dv.stats()["direct_hits"] == 0


@pytest.mark.parametrize("supersampling,wrap", [(1, 1), (2, 0)])
def test_direct_max_path_with_textures(dv, oracle, supersampling, wrap):
    """MAX strategy with textured triangles: direct hits also leave a {cell, key, colour} record and k_pick gives every
    cell the colour of the record that won. A fine textured sphere (unsplit triangles) mixed with a few large textured
    and coloured triangles (subdivided: their winners come from the replay tiers)."""
    from obj2voxel_amd import hip
    rng = np.random.default_rng(5)
    v, uv = meshes.uv_sphere(48, with_uv=True)
    big = (rng.random((5, 9)) * 1.6 - 0.8).astype(np.float32)
    big_uv = rng.random((5, 6)).astype(np.float32)
    verts = np.concatenate([np.reshape(v, (-1, 9)), big])
    uvs = np.concatenate([np.reshape(uv, (-1, 6)), big_uv])
    T = len(verts)
    types = np.full(T, hip.TRI_TEXTURED, np.uint32)
    types[::7] = hip.TRI_UNTEXTURED
    types[-2] = hip.TRI_MATERIALLESS
    kw = dict(uvs=uvs, types=types, colors=rng.random((T, 3)).astype(np.float32), texids=(np.arange(T) % 2).astype(np.int32),
              textures=[(meshes.checker_texture(64, 8), wrap), (rng.integers(0, 256, (9, 5, 4)).astype(np.uint8), 1 - wrap)],
              strategy=0, supersampling=supersampling)
    got, want = _run_both(dv, oracle, verts, 112, **kw)
    _compare(got, want)
    st = dv.stats()
    assert 0 < st["direct_hits"] < st["hits"]


@pytest.mark.parametrize("mode", ["none", "few"])
@pytest.mark.parametrize("strategy", [0, 1])
def test_hit_slabs_are_a_budget_not_a_requirement(oracle, monkeypatch, mode, strategy):
    """A cell's first eight hits go straight into the slab of its brick (k_voxelize -> Params::slabs), the rest through the pool
    and the counting sort.  The slabs are sized from a budget: with none at all (O2V_NO_SLABS=1, a fresh context) every hit is
    pooled, with four slabs (the tiny-buffer hook) a handful of bricks are inline and all the others pooled - both must give
    the oracle's voxels, like the default.  Textured mesh with subdivided triangles, cells from 1 to hundreds of hits, 2 x
    supersampling; MAX takes the route through the pool for the leaves of subdivided triangles only."""
    from obj2voxel_amd import hip
    if mode == "none":
        monkeypatch.setenv("O2V_NO_SLABS", "1")
    else:
        monkeypatch.setenv("O2V_TEST_TINY_BUFFERS", "1")
    v, uv = meshes.uv_sphere(24, with_uv=True)
    big = meshes.box_room(2) * 0.8 + 0.1
    v = np.concatenate([v * 0.4 + 0.5, big, meshes.uv_sphere(30, radius=0.02, center=(0.5, 0.5, 0.93))])
    T = len(v)
    uvs = np.concatenate([uv, np.tile(np.array([0, 0, 1, 0, 1, 1], np.float32), (T - len(uv), 1))])
    types = np.full(T, hip.TRI_TEXTURED, np.uint32)
    types[1::3] = hip.TRI_UNTEXTURED
    kw = dict(uvs=uvs, types=types, colors=meshes.triangle_colors(T), texids=np.zeros(T, np.int32))
    tex = [(meshes.checker_texture(64, 8), 1)]
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_textures(tex)
        d.set_triangles(v, **kw)
        got = d.voxelize(160, strategy=strategy, supersampling=2)
        _compare(got, oracle.voxelize(v, 160, strategy=strategy, supersampling=2, textures=tex, **kw))
    finally:
        d.close()


@pytest.mark.parametrize("route", ["occupancy", "coloured_max", "blend_ss2", "textured_blend"])
def test_xy_tiles_partition_the_grid(dv, oracle, route):
    """o2v_hip_params::x_begin .. y_end: an x / y tile of the output grid, like the z slab - triangles that miss it are dropped,
    leaves are clamped to it, the 16-bit coordinate fields are relative to its box.  Every output voxel belongs to exactly one
    tile: the records of 3 x 2 tiles (and a z slab inside one) concatenated equal the whole grid's and the oracle's, on every
    route.  Large and small triangles mixed, so that subdivided leaves straddle the tile borders."""
    from obj2voxel_amd import hip
    v, uv = meshes.uv_sphere(40, with_uv=True)
    big, buv = meshes.uv_sphere(5, radius=0.7, center=(0.1, -0.1, 0.05), with_uv=True)
    v, uv = np.concatenate([v, big]).astype(np.float32), np.concatenate([uv, buv]).astype(np.float32)
    T = len(v)
    res = 200
    kw = {"occupancy": dict(),
          "coloured_max": dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=0),
          "blend_ss2": dict(types=np.full(T, hip.TRI_UNTEXTURED, np.uint32), colors=meshes.triangle_colors(T), strategy=1, supersampling=2),
          "textured_blend": dict(uvs=uv, types=np.full(T, hip.TRI_TEXTURED, np.uint32), texids=np.zeros(T, np.int32),
                                 textures=[(meshes.checker_texture(64, 8), 1)], strategy=1)}[route]
    got, want = _run_both(dv, oracle, v, res, **kw)
    _compare(got, want)
    want = meshes.sorted_voxels(want)
    run = dict(supersampling=kw.get("supersampling", 1), strategy=kw.get("strategy", 0))
    parts = []
    xcuts, ycuts = (0, 64, 132, 200), (0, 100, 200)
    for iy in range(2):
        for ix in range(3):
            xt, yt = (xcuts[ix], xcuts[ix + 1]), (ycuts[iy], ycuts[iy + 1])
            part = dv.voxelize(res, xtile=xt, ytile=yt, **run)
            assert len(part) == 0 or (part[:, 0].min() >= xt[0] and part[:, 0].max() < xt[1] and part[:, 1].min() >= yt[0] and part[:, 1].max() < yt[1])
            parts.append(part)
    assert np.array_equal(meshes.sorted_voxels(np.concatenate(parts)), want)
    # a z slab inside a tile
    part = meshes.sorted_voxels(dv.voxelize(res, xtile=(64, 132), ytile=(100, 200), zslab=(50, 120), **run))
    m = (want[:, 0] >= 64) & (want[:, 0] < 132) & (want[:, 1] >= 100) & (want[:, 2] >= 50) & (want[:, 2] < 120)
    assert np.array_equal(part, want[m])
    with pytest.raises(hip.DeviceError):
        dv.voxelize(res, xtile=(2, 100))      # a tile begins at a multiple of 4
