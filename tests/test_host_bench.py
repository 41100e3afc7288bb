"""bench.py without a GPU: the workload series and the JSON line's contract, from recorded device statistics."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402

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
    assert bench.workload_for(8, "config4") == ("config4", 4096, 3536)   # BASELINE configs[4]: timed beside the series at N = 8
    for n in (2, 4, 8):   # one series: `value` at every N is the same kind of job ("scaling": "weak")
        name, res, nv = bench.workload_for(n)
        assert name == "weak" and res % (2 * n) == 0
        assert abs(res * res - 1024 * 1024 * n) / (1024 * 1024 * n) < 0.01   # surface voxels (~ res^2) grow with N
        assert abs(nv * nv - 467 * 467 * n) / (467 * 467 * n) < 0.01         # ... and so do the triangles (~ nv^2)
    assert bench.workload_for(8, "weak")[0] == "weak"


class _Comm:
    kind, world = "rccl", 2


def _line(n, stats, name="config2", res=1024, nv=467, comm=None):
    args = argparse.Namespace(steps=20, warmup=3, no_capi=True, no_cpu_baseline=True, no_routes=True, route_steps=2)
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
    # (the clip kernel of the headline route: k_voxelize_occ since round 5, k_voxelize<false> in older summaries)
    vk = "k_voxelize_occ" if "k_voxelize_occ" in cur["kernels"] else "k_voxelize<false>"
    assert r["bound"] == "valu" and r["kernel"] == vk
    # counters of another build of the device code are labelled, not passed off as this kernel's
    from obj2voxel_amd import hip
    assert bool(r.get("stale")) == (cur.get("build_id") != hip.build_id())
    assert bool(r.get("estimated")) == bool(r.get("stale"))
    sq = cur["kernels"][vk]["sq"]
    assert r["valu_instructions_per_launch"] == int(sq["SQ_INSTS_VALU"])
    # the direct evidence of the bound, recomputable from the committed counters with the formulas the line states
    assert abs(r["active_lane_fraction"] - sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_INSTS_VALU"] / 64.0) < 1e-3
    assert abs(r["valu_busy"] - sq["SQ_ACTIVE_INST_VALU"] * 4.0 / bench.N_SIMDS / (cur["kernels"][vk]["avg_us"] * 1e-6 * bench.CLOCK_GHZ * 1e9)) < 1e-3
    assert 0.3 < r["valu_busy"] <= 1.05 and set(r["formulas"]) >= {"achieved", "peak", "active_lane_fraction", "valu_busy", "counters"}
    assert out["pipeline"]["measured_traffic_bytes"] > 0 and out["roofline_hbm_view"]["bound"] == "hbm"
    assert abs(out["value"] - cur["workload_stats"]["voxels"] / 1.18e-3 / 1e6) < 1
    # the mix-weighted ceiling: recomputable from profiles/ alone (histogram x measured issue costs; instruction count; time)
    # (a model, kept apart from the measured fields: static histogram x microbenchmark)
    isa = json.load(open(bench.ISA_HIST))
    if vk in isa:
        m = isa[vk]["mix_cycles_per_valu"]
        mm = r["mix_model"]
        assert 2.0 < m < 5.0 and abs(mm["mix_cycles_per_valu_instruction"] - m) < 1e-2
        assert abs(mm["mix_ceiling"] - bench.N_SIMDS * bench.CLOCK_GHZ / m) < 0.1
        assert abs(mm["frac_of_mix_ceiling"] - r["achieved"] / mm["mix_ceiling"]) < 1e-3
        assert mm["frac_of_mix_ceiling"] > r["frac"]          # (the 2-cycle peak is the looser bound)
        assert bool(mm["stale"]) == (isa.get("build_id") != hip.build_id())


def test_isa_histogram_prices_every_opcode_of_the_clip_loop():
    """tools/isa_hist.py: the committed histogram names the rates file it was priced with, and every opcode's price is one of
    that file's measured loops (w4 column)."""
    isa = json.load(open(bench.ISA_HIST))
    rates = json.load(open(os.path.join(bench.ROOT, isa["rates_file"])))["results"]
    for k in ("k_voxelize<false>", "k_voxelize<true>"):
        e = isa[k]
        total = 0.0
        for op, d in e["valu_by_opcode"].items():
            assert d["priced_as"] in rates and abs(rates[d["priced_as"]]["w4"] - d["cycles"]) < 1e-9, op
            total += d["count"] * d["cycles"]
        assert abs(total / e["valu"] - e["mix_cycles_per_valu"]) < 1e-6


def test_other_workloads_get_an_estimate_not_the_counters():
    stats = dict(STATS, jobs=STATS["jobs"] // 2, triangles=400000)
    out = _line(1, stats)
    assert out["roofline"]["bound"] == "valu" and out["roofline"]["estimated"] is True
    assert out["pipeline"]["measured_traffic_bytes"] is None
    two = _line(2, STATS, name="weak", res=1448, nv=660, comm=_Comm())
    assert two["roofline"]["bound"] == "valu" and two["roofline"]["estimated"] is True
    assert two["config"]["collectives"]["backend"] == "rccl" and two["config"]["parallelism"] == "zslab2"
    assert two["config"]["collectives"]["rccl_world_size"] == 2 and two["scaling"] == "weak"


def _full_line(n=1):
    """A line as heavy as the real one: routes, stages, capi_wall, published_workload, a cpu_baseline with every diagnostic."""
    cur = json.load(open(bench.PROFILE_SUMMARY))
    stats = dict(STATS, **{k: v for k, v in cur["workload_stats"].items() if not k.startswith("_")})
    out = _line(n, stats, comm=_Comm() if n > 1 else None, **({"name": "weak", "res": 2896, "nv": 1321} if n > 1 else {}))
    out["routes"] = [{"workload": f"route{i}", "what": "x" * 200, "kernels_ms": {f"k{j}": 0.1 for j in range(30)}} for i in range(8)]
    out["capi_wall"] = {"ms": 4.1, "ms_all": [4.1] * 8, "what": "y" * 300}
    out["published_workload"] = {"workload": "z" * 300, "later_calls_s": [0.03] * 4}
    out["cli_wall"] = {"headline": {"wall_s": 0.5}, "readme": {"wall_s": 0.6}}
    out["cpu_baseline"] = {"value": 21.2, "unit": "Mvoxels/s", "cores": 32, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                           "hardware_threads": 256, "median_s_by_threads": {str(k): 0.3 for k in (256, 128, 64, 32, 16)}, "value_1_thread": 1.61,
                           "runs_s": [0.23] * 3, "phases_s": {"prelude": 0.01, "chunk_loop": 0.2, "join": 0.02},
                           "sample": "the full workload (870488 tris at 1024^3 -> 4936186 voxels): median of 3 runs with 32 threads (the best of "
                                     "[16, 32, 64, 128, 256]) over 64^3 chunks after one warm-up run, and one run with 1 thread",
                           "matches_gpu_voxel_count": True}
    if n > 1:
        se = bench.strong_entry({"seconds": 4.0e-3, "voxels": stats["voxels"] * n, "cuts": list(range(n + 1))},
                                {"seconds_per_step": 0.6e-3, "voxels": stats["voxels"] * n}, n)
        out["strong_scaling_same_job"] = se
        out["config4"] = {"workload": bench.WORKLOAD_TEXT["config4"].format(nv=3536, T=49999040, res=4096, n=n), "metric": "Mvoxels/sec at 4096^3 grid",
                          "value": 60000.0, "mtris_per_s": 30000.0, "ms_per_step": 1.3, "voxels": 79000000, "stages_ms_rank0": dict(STAGES),
                          "strong_scaling_same_job": se, "answers": "w" * 200}
        out["config"]["upload"] = {"bytes": 1, "what": "v" * 200}
    return out


def test_stdout_line_is_compact_whatever_the_details_hold():
    """The driver keeps a bounded tail of stdout and parses its last line: round 5's 24.7 KB line came out unparsed.  The line
    has a hard budget and a fixed key set; everything else is in the sidecar."""
    out = _full_line(1)
    assert len(json.dumps(out)) > 8000          # (the details are what they were)
    s = bench.compact_line(out)
    assert len(s) < 4096 and "\n" not in s
    line = json.loads(s)
    assert set(line) <= set(bench.LINE_KEYS)
    for k in REQUIRED + ("cpu_baseline",):
        assert k in line, k
    assert set(line["config"]) == {"workload", "resolution", "triangles", "voxels", "parallelism", "collectives"}
    assert "configs[2]" in line["config"]["workload"] and line["dtype"] == "f32" and line["vs_baseline"] is None
    r = line["roofline"]
    assert set(r) <= set(bench.ROOFLINE_KEYS) and {"bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "traffic"} <= set(r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "cpu_model", "value_1_thread", "sample"}
    for gone in ("routes", "stages", "pipeline", "capi_wall", "published_workload", "cli_wall", "kernels_ms", "stats", "steady_state"):
        assert gone not in line


def test_stdout_line_of_an_eight_gpu_run_is_compact_and_carries_both_curves():
    out = _full_line(8)
    s = bench.compact_line(out)
    line = json.loads(s)
    assert len(s) < 4096 and line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["config"]["collectives"]["rccl_world_size"] == 2 or line["config"]["collectives"]["rccl_world_size"] == 8
    assert line["strong_scaling_same_job"]["speedup"] > 0 and line["config4"]["strong_scaling_same_job"]["speedup"] > 0
    assert "upload" not in line["config"] and "answers" not in line["config4"]


def test_stdout_line_budget_sheds_text_then_optional_keys(monkeypatch):
    out = _full_line(8)
    out["cpu_baseline"] = dict(value=1.0, unit="Mvoxels/s", cores=1, kind="port", cpu_model="c" * 3000, value_1_thread=1.0, sample="s" * 3000)
    out["config"]["workload"] = "w" * 3000
    s = bench.compact_line(out)
    assert len(s) < 4096
    line = json.loads(s)
    for k in REQUIRED:
        assert k in line, k
    monkeypatch.setattr(bench, "LINE_BUDGET", 600)      # nothing left to shed: loud, not a silently oversized line
    with pytest.raises(RuntimeError):
        bench.compact_line(out)


def test_emit_writes_the_details_beside_the_line(tmp_path, capsys, monkeypatch):
    out = _full_line(1)
    monkeypatch.setenv("O2V_BENCH_DETAILS", str(tmp_path / "d.json"))
    bench.emit(out)
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    assert json.loads(last) == bench.compact(out) and len(last) < 4096
    assert json.load(open(tmp_path / "d.json")) == json.loads(json.dumps(out))
    assert "bench details: " in cap.err and "routes" in cap.err


def test_same_job_strong_scaling_entry():
    """bench.py --gpus N prints, beside the weak-scaling `value`, what the N GPUs gain on the SAME job (its slabs one after the
    other on one GPU against the N-GPU step): the figure DESIGN.md section 5 holds against BASELINE's '>= 6x at 8 GPUs'."""
    run = {"seconds_per_step": 2.5e-3, "voxels": 79_000_000}
    e = bench.strong_entry({"seconds": 20.0e-3, "voxels": 79_000_000, "cuts": list(range(9))}, run, 8)
    assert e["speedup"] == 8.0 and e["efficiency"] == 1.0 and e["voxels_match"] is True and e["one_gpu_ms"] == 20.0 and e["n_gpu_ms"] == 2.5
    assert bench.strong_entry(None, run, 8) is None


def test_route_entry_contract():
    """A route of the N = 1 line: live kernel times + the committed counters of the same workload (or none)."""
    stats = dict(STATS, pool_slots=STATS["hits"], direct_hits=0)
    r = {"workload": "config2_blend", "what": "text", "tris": stats["triangles"], "res": 1024, "supersampling": 1, "strategy": "BLEND",
         "textured": False, "voxels": stats["voxels"], "ms": 2.0, "mvox_s": 2468.0, "mtris_s": 435.0, "passes": 1, "stages_ms": dict(STAGES),
         "stats": stats, "kernels_ms": {"k_voxelize<false>": {"ms": 1.0, "launches": 1}, "k_scatter": {"ms": 0.3, "launches": 1}}}
    prof = {"build_id": "0" * 16, "kernels": {}, "workloads": {"config2_blend": {"workload_stats": {"triangles": stats["triangles"], "voxels": stats["voxels"]},
            "kernels": {"k_scatter": {"launches_per_step": 1.0, "hbm_bytes": 1.2e9, "sq": {"SQ_INSTS_VALU": 5e6, "SQ_THREAD_CYCLES_VALU": 3e8}}}}}}
    e = bench.route_entry(r, prof)
    json.dumps(e)
    assert e["dominant_kernel"]["kernel"] == "k_voxelize<false>" and e["dominant_kernel"]["traffic_bytes"] is None
    sc = [k for k in e["top_kernels"] if k["kernel"] == "k_scatter"][0]
    assert sc["traffic_bytes"] == int(1.2e9) and abs(sc["hbm_frac_traffic"] - 1.2e9 / 0.3e-3 / 8e12) < 1e-3
    assert sc["stale"] is True            # the profile's build id is not the running library's
    assert sc["algorithmic_bytes"] == 32 * stats["pool_slots"] + 20 * stats["hits"]
    # a profile measured on another mesh is not used at all
    prof["workloads"]["config2_blend"]["workload_stats"]["voxels"] += 1
    assert bench.route_entry(r, prof)["top_kernels"][1]["traffic_bytes"] is None
