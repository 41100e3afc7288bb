import sys; sys.path.insert(0,'.')
from obj2voxel_amd import hip, meshes
dv = hip.DeviceVoxelizer(0)
dv.set_triangles(meshes.unit_cube())
for res, kw in ((4096, {}), (70000, {}), (4096, dict(zslab=(0, 512)))):
    try:
        n = dv.voxelize(res, read=False, **kw)
        print(res, kw, "ok", n, dv.stats()["grid_bytes"] / 1e9, "GB")
    except hip.DeviceError as e:
        print(res, kw, "DeviceError:", e)
