#!/usr/bin/env python3
"""Developer tool: the kernels of one step of a workload on a time axis (start offset, duration, queue), from a
rocprofv3 --kernel-trace run - shows what overlaps and where the stream waits.
usage (on the GPU box): python tools/timeline.py <workload> [steps]"""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "config2_blend"
    steps = sys.argv[2] if len(sys.argv) > 2 else "3"
    out = tempfile.mkdtemp(prefix="o2v_tl_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--",
                    sys.executable, os.path.join(ROOT, "tools", "run_workload.py"), workload, "--steps", steps, "--warmup", "1"],
                   cwd="/tmp", env=env, check=True, capture_output=True)
    path = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        if m:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Queue_Id", "?")))
    rows.sort()
    # the last step: from the last k_bounds on
    first = max(i for i, r in enumerate(rows) if r[2] == "k_bounds")
    t0 = rows[first][0]
    print(f"{workload}: last step, times in us")
    for s, e, n, q in rows[first:]:
        print(f"  {(s - t0) / 1e3:9.1f}  +{(e - s) / 1e3:8.1f}  -> {(e - t0) / 1e3:9.1f}   q{q}  {n}")


if __name__ == "__main__":
    main()
