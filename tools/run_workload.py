#!/usr/bin/env python3
"""Runs one named workload of the device pipeline for a few steps (inputs resident in HBM, results left in HBM) and prints
one JSON line; the unit rocprofv3 wraps in tools/profile_all.sh.  The workloads live in obj2voxel_amd/workloads.py.
usage: run_workload.py NAME [--steps K] [--warmup W] [--kernel-steps J]     NAME: a key of workloads.WORKLOADS or asset:<stem>"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from obj2voxel_amd import workloads  # noqa: E402

WORKLOADS = workloads.WORKLOADS
run = workloads.run

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kernel-steps", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(run(a.name, a.steps, a.warmup, kernel_steps=a.kernel_steps)), flush=True)
