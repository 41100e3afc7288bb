"""Pins the CPU oracle on every result the reference's own tests hold for the hot path.

Reference: test/main.cpp:120-126 (expected unit cube count), :128-156 (cube @64), :194-208 (cube @128 = two
chunks per axis), :225-237 (three planes @32), :239-252 (three planes @128).
"""
import numpy as np
import pytest

from obj2voxel_amd import meshes


def expected_unit_cube_voxels(r):
    # test/main.cpp:120-126
    return 8 + 12 * (r - 2) + 6 * (r - 2) * (r - 2)


@pytest.mark.parametrize("res", [64, 128])
def test_unit_cube_known_answer(oracle, res):
    vox = oracle.voxelize(meshes.unit_cube(), res)
    assert len(vox) == expected_unit_cube_voxels(res)
    assert {64: 23816, 128: 96776}[res] == len(vox)
    # every voxel lies on the cube surface and is white/opaque (MATERIALLESS, triangle.hpp:186)
    xyz = vox[:, :3]
    on_surface = ((xyz == 0) | (xyz == res - 1)).any(axis=1)
    assert on_surface.all()
    assert (vox[:, 3] == 0xFFFFFFFF).all()
    assert len(np.unique(xyz, axis=0)) == len(vox)


@pytest.mark.parametrize("res", [32, 128])
def test_three_planes_known_answer(oracle, res):
    vox = oracle.voxelize(meshes.three_planes(), res)
    assert len(vox) == 3 * res * res
    # x = 0.5 maps to exactly res/2 (0.25 + 0.5*(res-0.5)), whose voxel AABB is the single slice res/2
    assert np.unique(vox[:, 0]).tolist() == [0, res // 2, res - 1]


def test_empty_mesh(oracle):
    assert len(oracle.voxelize(np.zeros((0, 9), np.float32), 32)) == 0


def test_slab_filter_is_subset(oracle):
    s = meshes.uv_sphere(10)
    full = meshes.sorted_voxels(oracle.voxelize(s, 96))
    parts = [oracle.voxelize(s, 96, zslab=(z, z + 32)) for z in (0, 32, 64)]
    got = meshes.sorted_voxels(np.concatenate(parts))
    assert np.array_equal(full, got)


def test_oracle_regression_vectors(oracle):
    """Drift detector: vectors in tests/golden/oracle_regression.npz were produced by THIS oracle
    (tests/golden/make_golden.py); they are not reference outputs."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "oracle_regression.npz")
    data = np.load(path)
    from tests.golden.make_golden import CASES, run_case
    for name in CASES:
        vox = meshes.sorted_voxels(run_case(oracle, name))
        assert np.array_equal(vox, data[name]), name


def test_baseline_config1_spot_64_max_cpu_path(oracle):
    """BASELINE.json configs[0]: 'Spot cow (~6k tris) at 64^3, max-blend, CPU reference path (plumbing, no GPU)' with
    the survey's stand-in mesh (uv-sphere nv=39 -> 5928 triangles): the CPU oracle alone, and its chunk-parallel mode
    gives the identical result for any thread count."""
    v = meshes.uv_sphere(39)
    assert len(v) == 5928
    T = len(v)
    kw = dict(types=np.full(T, 2, np.uint32), colors=meshes.triangle_colors(T), strategy=0)
    one = meshes.sorted_voxels(oracle.voxelize(v, 64, **kw))
    oracle.set_threads(4)
    try:
        four = meshes.sorted_voxels(oracle.voxelize(v, 64, **kw))
    finally:
        oracle.set_threads(1)
    assert np.array_equal(one, four)
    assert len(np.unique(one[:, :3], axis=0)) == len(one) > 15000


def test_per_thread_state_survives_between_calls(oracle):
    """The harness keeps every thread's voxelizer and output list between calls (oracle/o2v_oracle.c: g_vz_cache): other
    meshes, strategies, supersampling and thread counts in between must not change a result, nor must releasing the state."""
    from obj2voxel_amd import meshes
    import numpy as np
    a = meshes.uv_sphere(24)
    b, buv = meshes.uv_sphere(9, with_uv=True)
    tex = [(meshes.checker_texture(32, 4), 1)]
    bkw = dict(uvs=buv, types=np.full(len(b), 3, np.uint32), texids=np.zeros(len(b), np.int32), textures=tex)

    def run_a(threads):
        oracle.set_threads(threads)
        return meshes.sorted_voxels(oracle.voxelize(a, 96, strategy=1, supersampling=2))

    oracle.release()
    want = run_a(1)
    try:
        oracle.set_threads(5)
        oracle.voxelize(b, 128, strategy=1, **bkw)              # textured, uv stamps in use
        assert np.array_equal(run_a(5), want)
        oracle.set_threads(3)
        oracle.voxelize(meshes.unit_cube(), 64, strategy=0)     # axis-aligned, MAX
        assert np.array_equal(run_a(8), want)
        oracle.release()
        assert np.array_equal(run_a(2), want)
    finally:
        oracle.set_threads(1)


def test_sparse_chunk_binning_equals_the_dense_tables(oracle, monkeypatch):
    """Grids of more than 2^27 chunks (resolutions beyond ~32 000: the x / y tile tests) bin the triangles by a sorted list of
    (chunk, triangle) pairs instead of dense per-chunk tables; forced on small grids here, the two must agree record for record
    (order included: ascending chunks, ascending triangles inside a chunk)."""
    import numpy as np
    from obj2voxel_amd import meshes
    v, uv = meshes.uv_sphere(14, with_uv=True)
    T = len(v)
    kw = dict(uvs=uv, types=np.where(np.arange(T) % 2 == 0, 3, 2).astype(np.uint32), colors=meshes.triangle_colors(T),
              texids=np.zeros(T, np.int32), textures=[(meshes.checker_texture(32, 4), 1)])
    for res, ss, strategy, zslab in ((96, 1, 1, (0, 0)), (70, 2, 0, (0, 0)), (130, 1, 1, (40, 90))):
        monkeypatch.delenv("O2V_ORACLE_SPARSE_BINS", raising=False)
        dense = oracle.voxelize(v, res, supersampling=ss, strategy=strategy, zslab=zslab, **kw)
        monkeypatch.setenv("O2V_ORACLE_SPARSE_BINS", "1")
        sparse = oracle.voxelize(v, res, supersampling=ss, strategy=strategy, zslab=zslab, **kw)
        assert len(dense) > 1000 and np.array_equal(meshes.sorted_voxels(dense), meshes.sorted_voxels(sparse))
    monkeypatch.delenv("O2V_ORACLE_SPARSE_BINS", raising=False)
    # a thin strip in the z = const plane of a 100 000^3 grid: only the sparse path can hold its chunk lists
    strip = meshes.diagonal_strip(400, width=6e-5)
    vox = oracle.voxelize(strip, 100_000)
    assert len(vox) > 100_000 and int(vox[:, 0].max()) > 99_000 and int(vox[:, 1].max()) > 99_000 and int(vox[:, 2].max()) < 16
