#!/usr/bin/env python3
"""Device-pipeline timings on every workload of tools/run_workload.py (BASELINE.json configs and stress shapes).
Developer tool: prints one JSON line per case; inputs resident in HBM, results left in HBM, like bench.py.
usage: bench_configs.py [substring of the workload names to run]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from obj2voxel_amd import hip
from tools import run_workload

only = sys.argv[1] if len(sys.argv) > 1 else None
dv = hip.DeviceVoxelizer(0)
for name in run_workload.WORKLOADS:
    if only and only not in name:
        continue
    r = run_workload.run(name, steps=3 if name.startswith("config3") else 5, dv=dv)
    r.pop("stats")
    print(json.dumps(r), flush=True)
