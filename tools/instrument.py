#!/usr/bin/env python3
"""Developer tool: event counts of k_voxelize's clip loop on named workloads (obj2voxel_amd/workloads.py), from the
instrumented library (make -C obj2voxel_amd/csrc instr -> libobj2voxel_amd_instr.so, loaded through O2V_LIB).
usage: O2V_LIB=obj2voxel_amd/libobj2voxel_amd_instr.so python tools/instrument.py [workload ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from obj2voxel_amd import hip, workloads  # noqa: E402

NAMES = ["wave_iterations", "lane_events", "acc_passes_all(event)", "whole_keep(event)", "iterations with <= 16 active lanes",
         "whole_discard(event)", "iterations with <= 32 active lanes", "cut(event)", "cut: first piece final (incl. settled)", "cycles: phase 1 inside file_survivor (sum over waves)",
         "phase 1: groups of survivors taken from the ring", "phase 1: survivors taken from the ring", "cycles: staging + barriers (sum over waves)", "cycles: phase 2 loop",
         "cycles: phase 1", "cycles: whole kernel"]
for name in (sys.argv[1:] or ["config2"]):
    verts, mat, textures, res, kw, text = workloads.load(name)
    dv = hip.DeviceVoxelizer(0)
    dv.set_textures(textures or [])
    dv.set_triangles(verts, **mat)
    dv.voxelize(res, read=False, **kw)
    n = dv.voxelize(res, read=False, **kw)
    c = dv.debug_counters()
    st = dv.stats()
    out = {"workload": name, "voxels": int(n), "stats": st, "timings": dv.timings()}
    out["events"] = {NAMES[i]: int(c[i]) for i in range(16)}
    ev = out["events"]
    if st["jobs"]:
        out["per_job"] = {"lane_events": round(ev["lane_events"] / st["jobs"], 2), "cuts": round(ev["cut(event)"] / st["jobs"], 2)}
        out["lanes_per_iteration"] = round(ev["lane_events"] / max(ev["wave_iterations"], 1), 1)
    print(json.dumps(out), flush=True)
    dv.close()
