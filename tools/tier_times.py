#!/usr/bin/env python3
"""Developer tool: per-kernel times of a workload with the resolve tiers serialised on one stream (O2V_DEBUG_SYNC=1), and
the histogram of pooled hits per brick.  usage: python tools/tier_times.py [workload ...]"""
import ctypes as C
import json
import os
import sys

os.environ["O2V_DEBUG_SYNC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from obj2voxel_amd import hip, workloads  # noqa: E402

for name in (sys.argv[1:] or ["config2_blend"]):
    verts, mat, textures, res, kw, text = workloads.load(name)
    dv = hip.DeviceVoxelizer(0)
    dv.set_textures(textures or [])
    dv.set_triangles(verts, **mat)
    dv.voxelize(res, read=False, **kw)
    dv.voxelize(res, read=False, kernel_times=True, **kw)
    kt = dv.kernel_times()
    h = np.zeros(32, np.uint64)
    dv._L.o2v_hip_debug_hits_histogram.argtypes = [C.c_void_p, C.c_void_p]
    dv._L.o2v_hip_debug_hits_histogram(dv._ctx, h.ctypes.data)
    print(json.dumps({"workload": name, "stats": dv.stats(), "kernels_ms": {k: round(v[0], 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
                      "bricks_by_log2_hits": [int(x) for x in h[:20]]}), flush=True)
    dv.close()
