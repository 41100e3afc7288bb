#!/usr/bin/env python3
"""Developer tool: event counts of k_voxelize's clip loop on the bench workload, from the instrumented library
(make -C obj2voxel_amd/csrc instr -> libobj2voxel_amd_instr.so, loaded through O2V_LIB).
usage: O2V_LIB=obj2voxel_amd/libobj2voxel_amd_instr.so python tools/instrument.py [nv res [textured]]"""
import json
import os
import sys

sys.path.insert(0, '.')
import numpy as np
from obj2voxel_amd import hip, meshes

NAMES = ["wave_iterations", "lane_events", "acc_passes_all(event)", "whole_keep(event)", "iterations with <= 16 active lanes",
         "whole_discard(event)", "iterations with <= 32 active lanes", "cut(event)", "cut: first piece final (incl. settled)", "cut: first piece settled by its single plane",
         "cut: second piece final (incl. settled)", "cut: second piece settled by its single plane", "cycles: staging + barriers (sum over waves)", "cycles: phase 2 loop",
         "cycles: phase 1", "cycles: whole kernel"]
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 467
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
textured = len(sys.argv) > 3
dv = hip.DeviceVoxelizer(0)
if textured:
    v, uv = meshes.uv_sphere(nv, with_uv=True)
    T = len(v)
    dv.set_textures([(meshes.checker_texture(1024, 32), 1)])
    dv.set_triangles(v, uvs=uv, types=np.full(T, 3, np.uint32), texids=np.zeros(T, np.int32))
else:
    dv.set_triangles(meshes.uv_sphere(nv))
dv.voxelize(res, read=False)
n = dv.voxelize(res, read=False)
c = dv.debug_counters()
st = dv.stats()
out = {"voxels": int(n), "hits": st["hits"], "candidates": st["candidates"], "timings": dv.timings()}
out["events"] = {NAMES[i]: int(c[i]) for i in range(16)}
print(json.dumps(out, indent=1))
