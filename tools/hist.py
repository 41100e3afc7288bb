"""Developer tool: hits-per-cell histogram of the bench workload."""
import os
os.environ.setdefault("O2V_NO_DIRECT_MAX", "1")  # the hit lists are only kept on the sort-and-replay route
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes
nv, res = int(sys.argv[1]), int(sys.argv[2])
zslab = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
dv = hip.DeviceVoxelizer(0)
dv.set_triangles(meshes.uv_sphere(nv))
dv.voxelize(res, zslab=zslab, read=False)
dv.voxelize(res, zslab=zslab, read=False)
print(dv.timings())
h = np.zeros(32, np.uint64)
dv._L.o2v_hip_debug_hits_histogram.argtypes = [C.c_void_p, C.c_void_p]
dv._L.o2v_hip_debug_hits_histogram(dv._ctx, h.ctypes.data)
print(dv.stats())
for b, n in enumerate(h):
    if n:
        print(f"hits <= {1 << b:6d}: {int(n)} cells")
