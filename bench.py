#!/usr/bin/env python3
"""bench.py -- throughput of the voxelization hot path on N MI355X GPUs of one node.

Metric (BASELINE.json): Mvoxels/s (output voxels / second) at a 1024^3 grid, with Mtris/s beside it.

Workload at N=1: BASELINE.json configs[2] ("Stanford Dragon (~870k tris) at 1024^3, 1xMI355X"), which is the
configuration the metric is quoted on ("at 1024^3 grid").  The asset is not in the reference tree and there is
no network, so the stand-in is the deterministic UV sphere of SURVEY.md section 8d: nv = 467 -> 870 488
triangles, MATERIALLESS, MAX strategy, resolution 1024.

One step = one pass of the whole device pipeline (bounds -> transform -> exact subdivision -> AABB walk + clip
-> per-voxel combine: for this workload (MAX, no textures) a 64-bit atomic max per hit and one emission pass over the
dirty bricks) over triangles already resident in HBM, leaving the (x, y, z, argb) records in HBM.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), the grid is split into N z-slabs,
every rank voxelizes its slab from the replicated triangle list (triangles are binned to slabs on the device
by AABB; no data-path collective is needed: SURVEY.md section 8e).  The slab cuts are work-balanced: each step
every rank runs o2v_hip_plan_slabs (a z-histogram of predicted hits over the replicated triangles) and takes its
own slab, so the plan's cost is inside the timed region.  Scaling is weak: the job grows with N so that triangles
and output voxels per GPU stay fixed on average (resolution 1024*sqrt(N), nv = 467*sqrt(N)).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def workload_for(n_gpus):
    s = math.sqrt(n_gpus)
    res = int(round(1024 * s / (2 * n_gpus))) * 2 * n_gpus  # even slabs of equal height
    nv = int(round(467 * s))
    return res, nv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the N > 1 code path on a single-GPU box)")
    ap.add_argument("--same-device", action="store_true", help="debugging: every rank uses GPU 0")
    ap.add_argument("--equal-slabs", action="store_true", help="N > 1: equal-height z-slabs instead of the work-balanced plan")
    ap.add_argument("--resolution", type=int, default=0, help="override (debugging only; invalidates the metric)")
    ap.add_argument("--nv", type=int, default=0, help="override (debugging only; invalidates the metric)")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: the HIP runtime torch bundles must be the one that gets loaded

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
        n = world
    dist = None
    dev_index = 0
    if n > 1:
        import torch.distributed as dist
        # one process per GPU; if the launcher narrowed each rank's visibility to a single device, that device is 0
        visible = torch.cuda.device_count()
        dev_index = 0 if (args.same_device or local_rank >= visible) else local_rank
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)

    from obj2voxel_amd import hip, meshes, slab as slabs

    res, nv = workload_for(n)
    if args.resolution:
        res = args.resolution
    if args.nv:
        nv = args.nv
    verts = meshes.uv_sphere(nv)
    T = len(verts)
    z0, z1 = slabs.slab_range(rank, n, res)

    dv = hip.DeviceVoxelizer(dev_index if n > 1 else 0)
    dv.set_triangles(verts)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if n == 1:
            return dv.voxelize(res, read=False)
        if args.equal_slabs:
            return dv.voxelize(res, zslab=(z0, z1), read=False)
        # every rank derives the same work-balanced cuts from the replicated triangles (o2v_hip_plan_slabs, part of the
        # timed step: a new mesh needs a new plan), then voxelizes its own slab; the plan's bounds save a second pass
        cuts, bnd = dv.plan_slabs(res, n)
        return dv.voxelize(res, zslab=(cuts[rank], cuts[rank + 1]), bounds=bnd, read=False)

    for _ in range(args.warmup):
        step()
    stage_names = ("bounds_ms", "expand_ms", "voxelize_ms", "scan_ms", "resolve_ms", "total_ms")
    stage_sum = {k: 0.0 for k in stage_names}
    barrier()
    t0 = time.perf_counter()
    count = 0
    for _ in range(args.steps):
        count = step()
        tm = dv.timings()
        for k in stage_names:
            stage_sum[k] += tm[k]
    barrier()
    elapsed = time.perf_counter() - t0
    stats = dv.stats()

    total_voxels, max_elapsed = slabs.reduce_job(dist, count, elapsed,
                                                  device="cuda" if (dist is not None and args.backend == "nccl") else None)

    if rank == 0:
        ms_per_step = max_elapsed / args.steps * 1e3
        value = total_voxels / (max_elapsed / args.steps) / 1e6
        stage_avg = {k: stage_sum[k] / args.steps for k in stage_names}
        # dominant kernel and its algorithmic bytes per launch (DESIGN.md section "Kernels and rooflines")
        L, tiles, H, V = stats["leaves"], stats["tiles"], stats["hits"], stats["voxels"]
        B, D, slots = stats["bricks"], stats["dirty_bricks"], stats["pool_slots"]
        # algorithmic bytes per launch of each stage
        REC = 16  # sorted record: 16 bytes for a mesh without textured triangles (this workload), else 24
        Hd = stats["direct_hits"]  # MAX without textures: hits that went straight into the 64-bit max grid
        Hp = H - Hd                # hits that took the pool -> counting sort -> ordered replay route
        if Hd:
            # direct MAX path: the dirty bricks / voxels reported belong to the 64-bit grid (2 KiB per brick)
            alg_bytes = {
                "expand_ms": 36 * T + 96 * L + 8 * tiles,
                "voxelize_ms": 96 * L + 8 * tiles + (8 + 1) * Hd + (32 + 4 + 1) * Hp,
                "scan_ms": B + 32 * slots + (4 + REC) * Hp,
                "resolve_ms": REC * Hp + B + 2 * 2048 * D + 16 * V,
            }
        else:
            alg_bytes = {
                "expand_ms": 36 * T + 96 * L + 8 * tiles,
                "voxelize_ms": 96 * L + 8 * tiles + (32 + 4 + 1) * H,
                "scan_ms": B + 1024 * D + (16 + 4) * V + 32 * slots + (4 + REC) * H + 1024 * D,
                "resolve_ms": 16 * V + REC * H + 16 * V,
            }
        kernel_of = {"expand_ms": "k_expand_roots+k_expand_nodes", "voxelize_ms": "k_voxelize",
                     "scan_ms": "k_scan_flags+k_scan_bricks+k_scatter+k_reset_bricks", "resolve_ms": "k_resolve*+k_emit_max"}
        dom = max(alg_bytes, key=lambda k: stage_avg[k])
        achieved = alg_bytes[dom] / (stage_avg[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if n == 1 and os.path.exists(tpath):  # the committed PMC pass measured the N = 1 workload
            try:
                traffic = json.load(open(tpath)).get(kernel_of[dom])
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kernel_of[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes": alg_bytes[dom], "kernel_ms": round(stage_avg[dom], 4)}
        # the fixed whole-pipeline numerator of SURVEY.md section 8d: 8*G^3 + 16*V + 76*T
        b_alg = 8 * res * res * res + 16 * total_voxels + 76 * T
        pipeline = {"b_alg_bytes": b_alg, "device_ms": round(stage_avg["total_ms"], 4),
                    "gbs": round(b_alg / (stage_avg["total_ms"] * 1e-3) / 1e9, 1),
                    "frac_of_hbm_peak": round(b_alg / (stage_avg["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "stages_ms": {k: round(v, 4) for k, v in stage_avg.items()}}
        out = {
            "metric": "Mvoxels/sec at 1024^3 grid", "value": round(value, 2), "unit": "Mvoxels/s", "n_gpus": n,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mtris_per_s": round(T / (max_elapsed / args.steps) / 1e6, 2),
            "config": {"workload": f"uv-sphere nv={nv} ({T} tris, Stanford Dragon stand-in) at {res}^3, MATERIALLESS, "
                                   f"MAX, {n} z-slab(s)", "resolution": res, "triangles": T, "voxels": total_voxels,
                       "parallelism": f"zslab{n}"},
            "roofline": roofline, "pipeline": pipeline,
        }
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(verts, res, total_voxels)
        print(json.dumps(out), flush=True)
    dv.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(verts, res, expect_voxels):
    """The CPU oracle (a port of the reference algorithm, oracle/o2v_oracle.c) timed on this host's cores on the
    same workload, chunk-parallel like the reference's worker pool. Baseline only, not the optimisation target."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    oracle.build()
    oracle.set_threads(cores)
    t0 = time.perf_counter()
    vox = oracle.voxelize(verts, res)
    dt = time.perf_counter() - t0
    oracle.set_threads(1)
    return {"value": round(len(vox) / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": f"the full workload once ({len(verts)} tris at {res}^3 -> {len(vox)} voxels, {dt:.1f} s wall, "
                      f"{cores} threads over 64^3 chunks)", "matches_gpu_voxel_count": len(vox) == expect_voxels}


if __name__ == "__main__":
    main()
