"""Developer tool: run one parity case (device vs oracle) by name; use with O2V_DEBUG_SYNC=1 and `timeout`."""
import sys
sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes
from oracle import oracle

nv, res, strategy = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
v = meshes.uv_sphere(nv)
T = len(v)
ty = np.full(T, 2, np.uint32)
col = meshes.triangle_colors(T)
dv = hip.DeviceVoxelizer(0)
dv.set_triangles(v, types=ty, colors=col)
g = meshes.sorted_voxels(dv.voxelize(res, strategy=strategy))
print('device', len(g), dv.stats(), dv.timings(), flush=True)
w = meshes.sorted_voxels(oracle.voxelize(v, res, types=ty, colors=col, strategy=strategy))
print('equal', np.array_equal(g, w))
