"""A 4096^3 dense grid on ONE MI355X (BASELINE.json: "288 GB HBM makes 4096^3 RGBA dense feasible").  A material-less mesh
takes one byte per cell (69 GB); with colours and BLEND the 32-bit counter grid takes 275 GB of the 288 GB HBM.  Runs last
(file name) and in its own context so that no other test holds device memory; the 275 GB case is skipped if the allocation
does not fit."""
import numpy as np
import pytest

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu


def test_unit_cube_at_4096_on_one_gpu():
    from obj2voxel_amd import hip
    hip._bind().o2v_release_cached_device_memory()
    d = hip.DeviceVoxelizer(0)
    try:
        v = meshes.unit_cube()
        d.set_triangles(v)
        n = d.voxelize(4096, read=False)
        assert n == 8 + 12 * 4094 + 6 * 4094 ** 2  # reference test/main.cpp:120-126 at resolution 4096
        assert 68e9 < d.stats()["grid_bytes"] < 72e9
        part = d.voxelize(4096, zslab=(4000, 4096))   # same context, a thin slab: reuses the clean grid
        assert ((part[:, 2] >= 4000) & (part[:, 2] < 4096)).all()
        assert len(np.unique(part[:, :3], axis=0)) == len(part)
    finally:
        d.close()
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(v, types=np.full(len(v), 2, np.uint32), colors=meshes.triangle_colors(len(v)))
        try:
            n = d.voxelize(4096, strategy=1, read=False)
        except hip.DeviceError as e:
            if "code 4" in str(e):
                pytest.skip("not enough free HBM for a 275 GB grid: " + str(e))
            raise
        assert n == 8 + 12 * 4094 + 6 * 4094 ** 2
        assert d.stats()["grid_bytes"] > 270e9
    finally:
        d.close()
