#!/usr/bin/env python3
"""Turns the rocprofv3 outputs that tools/profile_all.sh leaves under gpurun_out/ (prof_final, pmcf_FETCH_SIZE, pmcf_WRITE_SIZE, pmcf_sq, bench_final.json)
into the committed summaries under profiles/<round>/ and profiles/traffic.json (read by bench.py)."""
import collections
import csv
import json
import os
import re
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join("profiles", rnd)
os.makedirs(out_dir, exist_ok=True)
import glob


def find(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


shutil.copy(find("gpurun_out/prof_final/**/f_kernel_stats.csv"), os.path.join(out_dir, "kernel_stats.csv"))
if os.path.exists("gpurun_out/predict_scaling_8.jsonl"):
    shutil.copy("gpurun_out/predict_scaling_8.jsonl", os.path.join(out_dir, "predict_scaling_8.jsonl"))
shutil.copy("gpurun_out/bench_final.json", os.path.join(out_dir, "bench_line.json"))


def kname(s):
    m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", s)
    return m.group(1) if m else None


res = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(find(f"gpurun_out/pmcf_{cname}/**/p_counter_collection.csv"))):
        k = kname(r["Kernel_Name"])
        if k and r["Counter_Name"] == cname:
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[cname] = sum(v) / len(v)
out = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (separate runs: --pmc FETCH_SIZE, --pmc WRITE_SIZE, with "
                   "--kernel-trace only), bench workload (uv-sphere nv=467 @1024^3). Counters are in KiB. Correction per "
                   "MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so "
                   "fetch is doubled; calibrated here on k_bounds (31.3 MB read) and k_reset_bricks (1 KiB per dirty brick "
                   "written, WRITE_SIZE exact).", "kernels": {}}
for k, v in res.items():
    f, w = v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024
    out["kernels"][k] = {"fetch_size_raw_bytes": round(f), "write_size_bytes": round(w), "hbm_bytes_corrected": round(2 * f + w)}
K = out["kernels"]


def hbm(k):
    return K.get(k, {}).get("hbm_bytes_corrected", 0)  # a kernel the workload never launches contributes nothing


out["k_voxelize"] = hbm("k_voxelize<false>")
out["k_scan_flags+k_scan_bricks+k_scatter+k_reset_bricks"] = sum(hbm(k) for k in ("k_scan_bricks", "k_scatter", "k_reset_bricks"))
out["k_resolve*+k_emit_max"] = sum(hbm(k) for k in K if k.startswith("k_resolve") or k in ("k_emit_max", "k_scan_flags"))
out["k_expand_roots+k_expand_nodes"] = hbm("k_expand_roots")
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
json.dump(out, open(os.path.join(out_dir, "pmc_hbm_traffic.json"), "w"), indent=1)

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(find("gpurun_out/pmcf_sq/**/p_counter_collection.csv"))):
    k = kname(r["Kernel_Name"])
    if k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
sq = {"_comment": "rocprofv3 --pmc pass (8 SQ counters, --kernel-trace only), bench workload, averages per launch. "
                  "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)."}
for k in ("k_voxelize<false>", "k_emit_max", "k_expand_roots", "k_scatter"):
    if k not in acc:
        continue
    d = {c: round(sum(v) / len(v)) for c, v in acc[k].items()}
    if d.get("SQ_INSTS_VALU"):
        d["derived"] = {"valu_active_fraction_per_wave": round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], 3),
                        "active_lanes_per_valu_instruction": round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"], 1)}
    sq[k] = d
json.dump(sq, open(os.path.join(out_dir, "sq_counters.json"), "w"), indent=1)
for r in csv.DictReader(open(os.path.join(out_dir, "kernel_stats.csv"))):
    print("%-34s calls=%-4s avg_us=%9.1f  %s%%" % (kname(r["Name"]) or r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(json.dumps(sq["k_voxelize<false>"]))
