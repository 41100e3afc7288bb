#!/usr/bin/env python3
"""Turns the rocprofv3 outputs that tools/profile_all.sh leaves under gpurun_out/prof/ into the committed summaries under
profiles/<round>/ and profiles/current.json (read by bench.py: measured HBM traffic and VALU instruction counts per
kernel of the bench command).   usage: collect_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
N_SIMDS, CLOCK_GHZ = 256 * 4, 2.4   # as bench.py
src = "gpurun_out/prof"
out_dir = os.path.join("profiles", rnd)
os.makedirs(out_dir, exist_ok=True)


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    return hits[0] if hits else None


def kname(s):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", s)
    if not m:
        return None
    targs = re.sub(r"[\s]|(?<=\d)u", "", m.group(2) or "").replace("(bool)1", "true").replace("(bool)0", "false")
    return m.group(1) + targs


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not path:
        return acc
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


summary = {}
for w in ("config2", "config2_colored_max", "config2_blend", "config2_textured_max", "scan_colored_max", "config1", "config3", "readme8192"):
    stats = find(f"{src}/{w}_stats/**/s_kernel_stats.csv")
    if not stats:
        continue
    shutil.copy(stats, os.path.join(out_dir, f"{w}_kernel_stats.csv"))
    rows = [(kname(r["Name"]), r) for r in csv.DictReader(open(stats))]
    # k_voxelize runs exactly once per pass of the pipeline (a step is one pass unless a buffer had to grow in the warm-up)
    passes = sum(int(r["Calls"]) for k, r in rows if k and k.startswith("k_voxelize")) or 1
    kernels = {}
    for k, r in rows:
        if not k:
            continue
        kernels[k] = {"calls": int(r["Calls"]), "launches_per_step": round(int(r["Calls"]) / passes, 2),
                      "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "share_pct": float(r["Percentage"])}
    fetch, write = counters(find(f"{src}/{w}_FETCH_SIZE/**/p_counter_collection.csv")), counters(find(f"{src}/{w}_WRITE_SIZE/**/p_counter_collection.csv"))
    for k in kernels:
        f = fetch.get(k, {}).get("FETCH_SIZE")
        wr = write.get(k, {}).get("WRITE_SIZE")
        if f and wr:
            f, wr = sum(f) / len(f) * 1024, sum(wr) / len(wr) * 1024   # the counters are in KiB
            kernels[k].update({"fetch_size_raw_bytes": round(f), "write_size_bytes": round(wr), "hbm_bytes": round(2 * f + wr)})
    sq = {}
    for part in ("sq1", "sq2", "sq3"):
        for k, d in counters(find(f"{src}/{w}_{part}/**/p_counter_collection.csv")).items():
            sq.setdefault(k, {}).update({c: round(sum(v) / len(v)) for c, v in d.items()})
    for k, d in sq.items():
        if k in kernels and d.get("SQ_INSTS_VALU", 0) > 1e6:
            us = kernels[k]["avg_us"]
            d["derived"] = {
                # every figure from the counters of this file and the kernel's average duration (avg_us), one formula each:
                "active_lanes_per_valu_instruction": round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"], 1),
                "active_lane_fraction": round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"] / 64.0, 3),
                # SQ_INSTS_VALU / avg_us against one wave64 VALU instruction per 2 cycles per SIMD (1024 SIMDs x 2.4 GHz / 2)
                "valu_issue_fraction_of_peak_at_2_cycles": round(d["SQ_INSTS_VALU"] / (us * 1e-6) / (N_SIMDS * CLOCK_GHZ * 1e9 / 2.0), 4),
                # SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD's VALU is executing, summed over the SIMDs:
                # x 4 / 1024 SIMDs / (avg_us x 2.4 GHz) = the fraction of the kernel's duration the VALUs are busy
                "valu_busy": round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMDS / (us * 1e-6 * CLOCK_GHZ * 1e9), 3) if d.get("SQ_ACTIVE_INST_VALU") else None,
                "simd_cycles_per_valu_instruction": round(us * 1e-6 * CLOCK_GHZ * 1e9 * N_SIMDS / d["SQ_INSTS_VALU"], 2)}
            kernels[k]["sq"] = d
    line = None
    lp = os.path.join(src, f"{w}_line.json")
    if os.path.exists(lp) and os.path.getsize(lp):
        line = json.loads(open(lp).read())
    step_traffic = sum(v["hbm_bytes"] * v["launches_per_step"] for v in kernels.values() if "hbm_bytes" in v)
    doc = {"_comment": "rocprofv3 on MI355X, one workload of tools/profile_all.sh. kernel times: --kernel-trace --stats; HBM bytes per "
                       "launch: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (--kernel-trace only), counters in KiB, "
                       "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half "
                       "the bytes of wide coalesced reads; calibrated in round 1 on k_bounds = 36 B x triangles read and "
                       "k_reset_bricks = the dirty bricks written); sq: two or three --pmc passes of 8 SQ counters, averages per launch "
                       "(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_CYCLES and SQ_BUSY_CYCLES are summed over the 32 shader engines).",
           "workload": w, "command": "python bench.py --no-cpu-baseline --no-capi --no-routes" if w == "config2" else f"python tools/run_workload.py {w}",
           "result_line": line, "hbm_bytes_per_step_all_kernels": round(step_traffic), "kernels": kernels}
    json.dump(doc, open(os.path.join(out_dir, f"{w}_profile.json"), "w"), indent=1)
    summary[w] = doc
    print(f"== {w}: {round(step_traffic / 1e6, 1)} MB of HBM traffic per step")
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"]):
        if v["avg_us"] * v["launches_per_step"] >= 5:
            print("   %-30s x%-5s avg %9.1f us  %6.1f MB  %s" % (k, v["launches_per_step"], v["avg_us"], v.get("hbm_bytes", 0) / 1e6,
                                                             ("VALU %.1f M, lanes %.1f" % (v["sq"]["SQ_INSTS_VALU"] / 1e6, v["sq"]["derived"]["active_lanes_per_valu_instruction"])) if "sq" in v else ""))
if "config2" in summary:
    bid_path = os.path.join(src, "build_id.txt")
    build_id = open(bid_path).read().strip() if os.path.exists(bid_path) else None
    cur = {"source": f"profiles/{rnd}/config2_profile.json (rocprofv3 passes of `python bench.py --no-cpu-baseline --no-capi --no-routes`)",
           # hash of the device sources of the profiled library (o2v_hip_build_id): bench.py labels these counters stale if
           # the running library was built from other sources
           "build_id": build_id,
           "kernels": summary["config2"]["kernels"]}
    # device statistics of the profiled workload (bench.py uses `jobs` to scale the instruction count to other workloads)
    stats = (summary["config2"].get("result_line") or {}).get("stats")
    if stats:
        cur["workload_stats"] = stats
    # the other routes (tools/run_workload.py NAME): only what bench.py reads, per kernel
    keep = ("launches_per_step", "avg_us", "hbm_bytes", "fetch_size_raw_bytes", "write_size_bytes")
    cur["workloads"] = {}
    for w, doc in summary.items():
        if w == "config2":
            continue
        ks = {}
        for k, v in doc["kernels"].items():
            e = {f: v[f] for f in keep if f in v}
            if "sq" in v:
                e["sq"] = {c: v["sq"][c] for c in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "derived") if c in v["sq"]}
            ks[k] = e
        cur["workloads"][w] = {"source": f"profiles/{rnd}/{w}_profile.json", "kernels": ks,
                               "workload_stats": (doc.get("result_line") or {}).get("stats")}
    json.dump(cur, open("profiles/current.json", "w"), indent=1)
for extra in ("predict_scaling_8_weak.jsonl", "predict_scaling_8_config4.jsonl", "valu_rates.json"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(out_dir, extra))
if os.path.exists(os.path.join(src, "bench_line.json")) and os.path.getsize(os.path.join(src, "bench_line.json")):
    shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(out_dir, "bench_line.json"))
