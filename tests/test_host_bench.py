"""bench.py without a GPU: the workload series and the JSON line's contract, from recorded device statistics."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")
STATS = {"triangles": 870488, "leaves": 870488, "tiles": 870488, "candidates": 32604084, "jobs": 12806306, "hits": 12054853,
         "voxels": 4936186, "grid_cells": 1073741824, "bricks": 16777216, "dirty_bricks": 308336, "pool_slots": 0,
         "direct_hits": 12054853}
STAGES = {"bounds_ms": 0.024, "expand_ms": 0.081, "voxelize_ms": 0.95, "scan_ms": 0.02, "resolve_ms": 0.078, "total_ms": 1.153,
          "plan_ms": 0.0, "collective_ms": 0.0}


def test_workload_series():
    assert bench.workload_for(1) == ("config2", 1024, 467)            # BASELINE configs[2]: the configuration the metric is quoted on
    assert bench.workload_for(8) == ("config4", 4096, 3536)           # BASELINE configs[4] on 8 GPUs
    for n in (2, 4):
        name, res, nv = bench.workload_for(n)
        assert name == "weak" and res % (2 * n) == 0
        assert abs(res * res - 1024 * 1024 * n) / (1024 * 1024 * n) < 0.01   # surface voxels (~ res^2) grow with N
        assert abs(nv * nv - 467 * 467 * n) / (467 * 467 * n) < 0.01         # ... and so do the triangles (~ nv^2)
    assert bench.workload_for(8, "weak")[0] == "weak"


class _Comm:
    kind, world = "rccl", 2


def _line(n, stats, name="config2", res=1024, nv=467, comm=None):
    args = argparse.Namespace(steps=20, warmup=3, no_capi=True, no_cpu_baseline=True)
    run = {"name": name, "res": res, "nv": nv, "T": stats["triangles"], "verts": None, "voxels": stats["voxels"] * n,
           "seconds_per_step": 1.18e-3, "stages_ms": dict(STAGES), "stats": dict(stats)}
    out = bench.report(args, n, run, None, comm)
    json.dumps(out)  # serialisable
    for k in REQUIRED:
        assert k in out, k
    assert out["n_gpus"] == n and out["higher_is_better"] is True and out["unit"] == "Mvoxels/s"
    assert str(res) in out["metric"]
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    return out


def test_line_of_the_profiled_workload_uses_the_measured_counters():
    cur = json.load(open(bench.PROFILE_SUMMARY))
    out = _line(1, dict(STATS, **{k: v for k, v in cur["workload_stats"].items() if not k.startswith("_")}))
    r = out["roofline"]
    assert r["bound"] == "valu" and not r.get("estimated") and r["kernel"] == "k_voxelize<false>"
    assert r["valu_instructions_per_launch"] == int(cur["kernels"]["k_voxelize<false>"]["sq"]["SQ_INSTS_VALU"])
    assert out["pipeline"]["measured_traffic_bytes"] > 0 and out["roofline_hbm_view"]["bound"] == "hbm"
    assert abs(out["value"] - cur["workload_stats"]["voxels"] / 1.18e-3 / 1e6) < 1


def test_other_workloads_get_an_estimate_not_the_counters():
    stats = dict(STATS, jobs=STATS["jobs"] // 2, triangles=400000)
    out = _line(1, stats)
    assert out["roofline"]["bound"] == "valu" and out["roofline"]["estimated"] is True
    assert out["pipeline"]["measured_traffic_bytes"] is None
    two = _line(2, STATS, name="weak", res=1448, nv=660, comm=_Comm())
    assert two["roofline"]["bound"] == "valu" and two["roofline"]["estimated"] is True
    assert two["config"]["collectives"]["backend"] == "rccl" and two["config"]["parallelism"] == "zslab2"
