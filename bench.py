#!/usr/bin/env python3
"""bench.py -- throughput of the voxelization hot path on N MI355X GPUs of one node.

Metric (BASELINE.json): Mvoxels/s (output voxels / second), with Mtris/s beside it, "at 1024^3 grid; 1/2/4/8 MI355X".

Workloads (the assets BASELINE.json names are not in the reference tree and there is no network; the stand-ins are the
deterministic meshes of SURVEY.md section 8d):
  N = 1      BASELINE.json configs[2] "Stanford Dragon (~870k tris) at 1024^3": uv-sphere nv = 467 -> 870 488 triangles,
             MATERIALLESS, MAX, resolution 1024 - the configuration the metric is quoted on.
  N = 2,4,8  the same job grown with N so that triangles and output voxels per GPU stay fixed on average (weak scaling:
             `value` at every N belongs to one series, "scaling": "weak"): resolution 1024 * sqrt(N), nv = 467 * sqrt(N).
             Beside it, in the same run: "strong_scaling_same_job" - the N-GPU job's planned slabs one after the other on
             rank 0's GPU against the N-GPU step (the speed-up of THAT job over one GPU).
  N = 8      additionally BASELINE.json configs[4] "Synthetic 50M-tri tessellated sphere at 4096^3, z-slab split across
             8xMI355X" (uv-sphere nv = 3536 -> 49 999 040 triangles at 4096^3), under "config4", with its own same-job
             strong scaling: the number that answers BASELINE's ">= 6x at 8 GPUs" (DESIGN.md section 5).
             (--workload weak|config4 picks the main job for any N > 1.)

N = 1 also times, after the headline, the other routes of the pipeline on their own workloads (obj2voxel_amd/workloads.py:
the configs[2] mesh coloured with MAX, with BLEND, textured with MAX; the configs[1] and configs[3] stand-ins) and reports them
under "routes", each with its kernels' live event times and the dominant kernel's fraction of the HBM and VALU-issue peaks.
If $O2V_ASSETS holds spot.obj / dragon.obj / sponza.obj those are run as well: dragon.obj replaces the
stand-in as the headline workload (config.workload says so), the others appear under "routes".

One step = one pass of the whole device pipeline over triangles already resident in HBM, the (x, y, z, argb) records left
in HBM.  N = 1: o2v_hip_voxelize (bounds -> transform -> exact subdivision -> AABB walk + clip -> per-voxel combine ->
records).  N > 1: one process per GPU (torch.distributed; backend nccl = RCCL), every rank holding the triangle list, and
one step = o2v_hip_voxelize_sharded: the bounds and work-histogram passes sharded over the ranks and combined with RCCL
all-reduces, block extents and slab counts all-gathered, each rank voxelizing its work-balanced z-slab.  No voxel data
crosses GPUs (every output voxel is owned by exactly one slab).
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8 TB/s spec
CLOCK_GHZ = 2.4                            # nominal (the SE cycle counters read 2.1 - 2.15 GHz under the clip kernel)
N_SIMDS = 256 * 4
VALU_PEAK_GINSTR = N_SIMDS * CLOCK_GHZ / 2.0   # wave64 VALU instructions / ns: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles = 1228.8 G/s
PROFILE_SUMMARY = os.path.join(ROOT, "profiles", "current.json")   # rocprofv3 PMC summary of this command (tools/collect_profiles.py)


def load_profile():
    try:
        return json.load(open(PROFILE_SUMMARY))
    except Exception:
        return None


ISA_HIST = os.path.join(ROOT, "profiles", "current_isa.json")   # tools/isa_hist.py --json: instruction histogram of the clip loop x measured issue costs


def mix_ceiling(kernel):
    """The mix-weighted issue ceiling of k_voxelize<...>: its clip loop's instruction histogram (from the compiled gfx950 code,
    tools/isa_hist.py) priced with the measured SIMD cycles per instruction of each opcode class (profiles/*/valu_rates.json)
    gives the cycles an average VALU instruction of THIS kernel costs; SIMDs x clock / that is the rate at which the kernel's
    own mix can issue at best.  None if the histogram is missing or was made from other device sources."""
    try:
        h = json.load(open(ISA_HIST))
        e = h.get(kernel)
        if not e or not e.get("mix_cycles_per_valu"):
            return None
        try:
            from obj2voxel_amd import hip
            running = hip.build_id()
        except Exception:
            running = None
        m = float(e["mix_cycles_per_valu"])
        return {"mix_cycles_per_valu_instruction": round(m, 3), "ceiling_ginstr": round(N_SIMDS * CLOCK_GHZ / m, 1),
                "stale": bool(running is not None and h.get("build_id") != running),
                "source": "profiles/current_isa.json (tools/isa_hist.py: clip loop histogram x " + str(h.get("rates_file")) + ")"}
    except (OSError, ValueError):
        return None


def valu_busy(sq, avg_us):
    """Fraction of a kernel's duration its VALUs are executing, from the committed counters alone: SQ_ACTIVE_INST_VALU counts
    quad-cycles summed over the SIMDs, so x 4 / 1024 SIMDs / (the kernel's average duration in the same rocprofv3 runs x the
    2.4 GHz nominal clock).  Near 1: bound by instruction issue (only fewer instructions or fuller lanes help); well below:
    the wavefronts wait (memory, barriers, tails)."""
    if not sq or not sq.get("SQ_ACTIVE_INST_VALU") or not avg_us:
        return None
    return round(sq["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMDS / (avg_us * 1e-6 * CLOCK_GHZ * 1e9), 3)


def profile_for(prof, workload, stats=None):
    """(kernels, stale) of the committed PMC summary for a named workload: `kernels` is None if the summary does not hold
    the workload or was measured on another mesh; stale = the summary was recorded with a library built from other device
    sources than the running one (o2v_hip_build_id), i.e. its counters describe other kernels."""
    if not prof:
        return None, False
    entry = prof if workload == "config2" else (prof.get("workloads") or {}).get(workload)
    if not entry or not entry.get("kernels"):
        return None, False
    ref = entry.get("workload_stats") or {}
    if stats is not None and ref and (ref.get("triangles"), ref.get("voxels")) != (stats.get("triangles"), stats.get("voxels")):
        return None, False
    try:
        from obj2voxel_amd import hip
        running = hip.build_id()
    except Exception:
        running = None
    stale = bool(prof.get("build_id")) and running is not None and prof["build_id"] != running
    if not prof.get("build_id"):
        stale = True   # a summary from before build ids were recorded cannot be matched to a library
    return entry["kernels"], stale


def workload_for(n_gpus, kind="auto"):
    """(name, resolution, nv).  kind: auto | weak | config4"""
    if n_gpus == 1:
        return "config2", 1024, 467
    if kind == "config4":
        return "config4", 4096, 3536
    s = math.sqrt(n_gpus)
    return "weak", int(round(1024 * s / (2 * n_gpus))) * 2 * n_gpus, int(round(467 * s))


WORKLOAD_TEXT = {
    "config2": "BASELINE configs[2] stand-in: uv-sphere nv={nv} ({T} tris, Stanford Dragon stand-in) at {res}^3, MATERIALLESS, MAX",
    "config4": "BASELINE configs[4]: uv-sphere nv={nv} ({T} tris, the 50M-triangle tessellated sphere) at {res}^3, MATERIALLESS, "
               "MAX, {n} work-balanced z-slabs",
    "weak": "weak-scaling series of configs[2]: uv-sphere nv={nv} ({T} tris) at {res}^3, MATERIALLESS, MAX, {n} work-balanced z-slabs",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-capi", action="store_true", help="skip the obj2voxel_voxelize() wall-time leg")
    ap.add_argument("--no-routes", action="store_true", help="skip the other routes / assets timed after the headline (N = 1)")
    ap.add_argument("--route-steps", type=int, default=5)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise "
                    "the N > 1 code path on a single-GPU box: the collectives then run over host memory)")
    ap.add_argument("--same-device", action="store_true", help="debugging: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--workload", default="auto", choices=["auto", "weak", "config4"])
    ap.add_argument("--resolution", type=int, default=0, help="override (debugging only; invalidates the metric)")
    ap.add_argument("--nv", type=int, default=0, help="override (debugging only; invalidates the metric)")
    ap.add_argument("--config4-resolution", type=int, default=0, help="override of the N = 8 companion job (tests only; invalidates it)")
    ap.add_argument("--config4-nv", type=int, default=0, help="override of the N = 8 companion job (tests only; invalidates it)")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: the HIP runtime (and RCCL) torch bundles must be the copies that get loaded

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

    from obj2voxel_amd import hip, meshes, slab as slabs, workloads

    dv = hip.DeviceVoxelizer(dev_index if n > 1 else 0)
    comm = None
    if n > 1:
        if args.backend == "nccl":
            # the library's own RCCL communicator, on its own stream: rank 0 makes the id, torch ships it
            box = [hip.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            try:
                comm = hip.Comm.rccl(box[0], rank, n, dev_index)
            except Exception as e:   # noqa: BLE001 - any rank without a communicator moves every rank to the fallback
                print(f"bench: rank {rank}: RCCL communicator failed ({e}); using torch.distributed collectives", file=sys.stderr)
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:
                    comm.close()
                comm = hip.Comm.torch_distributed(dist)
        else:
            comm = hip.Comm.torch_distributed(dist)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def time_workload(name, res, nv, steps, warmup, loaded=None):
        kw, text = {}, None
        if loaded is not None:     # a real asset (N = 1): (verts, materials, textures, resolution, keywords, description)
            verts, mat, textures, res, kw, text = loaded
            dv.set_textures(textures or [])
            dv.set_triangles(verts, **mat)
        else:
            verts = meshes.uv_sphere(nv)
            dv.set_triangles(verts)
        if n == 1:
            step = lambda **t: (dv.voxelize(res, read=False, **kw, **t), None)   # noqa: E731
        else:
            def step(**t):
                count, counts, cuts = dv.voxelize_sharded(comm, res, read=False, **t)
                return count, counts
        for _ in range(warmup):
            step()
        names = ("bounds_ms", "expand_ms", "voxelize_ms", "scan_ms", "resolve_ms", "total_ms", "plan_ms", "collective_ms")
        barrier()
        t0 = time.perf_counter()
        count = 0
        for _ in range(steps):   # the timed region: the steps and nothing else
            count, counts = step()
        barrier()
        elapsed = time.perf_counter() - t0
        total_voxels, max_elapsed = slabs.reduce_job(dist, count, elapsed,
                                                      device="cuda" if (dist is not None and args.backend == "nccl") else None)
        # the clip kernel's duration in the timed steps (two hipEvents on the kernel's own dispatch, recorded in every step: read
        # here for the last one; the roofline's kernel_ms)
        k2_ms_timed = dv.timings()["voxelize_ms"]
        # per-stage device times and, for N > 1, the collectives' shares: a few further steps outside the timed region with
        # O2V_HIP_FLAG_STAGE_TIMES (an event between the stages costs ~4 us of device time each: not in the timed steps)
        stage_steps = min(5, max(steps, 1))
        acc = {k: 0.0 for k in names}
        parts = [0.0] * 5
        for _ in range(stage_steps):
            step(stage_times=True)
            tm = dv.timings()
            for k in names:
                acc[k] += tm[k]
            parts = [a + b for a, b in zip(parts, tm["collective_parts_ms"])]
        run = {"name": name, "res": res, "nv": nv, "T": len(verts), "verts": verts, "voxels": total_voxels, "text": text, "kw": kw,
               "collective_parts_ms": [x / stage_steps for x in parts],
               "seconds_per_step": max_elapsed / steps, "stages_ms": {k: acc[k] / stage_steps for k in names}, "stats": dv.stats(),
               "k2_ms_timed": k2_ms_timed}
        if n == 1:
            # per-kernel times: two further steps with an event pair around every launch (outside the timed region: the
            # brackets cost a few microseconds per launch)
            kernels = {}
            for _ in range(2):
                dv.voxelize(res, read=False, kernel_times=True, **kw)
                for k, (ms, launches) in dv.kernel_times().items():
                    e = kernels.setdefault(k, [0.0, 0])
                    e[0] += ms
                    e[1] += launches
            run["kernels_ms"] = {k: {"ms": round(ms / 2, 4), "launches": launches // 2} for k, (ms, launches) in kernels.items()}
        return run

    name, res, nv = workload_for(n, args.workload)
    if args.resolution:
        res = args.resolution
    if args.nv:
        nv = args.nv
    loaded = None
    if n == 1 and not (args.resolution or args.nv) and workloads.asset_path("dragon"):
        loaded = workloads.load("asset:dragon")      # BASELINE configs[2] names this asset: it is the headline when present
        name = "asset:dragon"
    main_run = time_workload(name, res, nv, args.steps, args.warmup, loaded)
    routes = None
    if n == 1 and not args.no_routes and not (args.resolution or args.nv):
        names = list(workloads.BENCH_ROUTES)
        if loaded is not None:
            names.insert(0, "config2")               # the stand-in beside the real asset
        names += [f"asset:{stem}" for stem in ("spot", "sponza") if workloads.asset_path(stem)]
        prof = load_profile()
        routes = []
        for r in names:
            try:
                routes.append(route_entry(workloads.run(r, steps=args.route_steps, warmup=2, dv=dv, kernel_steps=2), prof))
            except Exception as e:   # noqa: BLE001 - a route that cannot run must not take the headline down
                routes.append({"workload": r, "error": f"{type(e).__name__}: {e}"})
    upload = None
    if n > 1 and main_run["verts"] is not None:
        upload = upload_comparison(main_run["verts"], dv, dist, args.backend, rank, barrier)
    strong = None
    if n > 1:
        strong = same_job_on_one_gpu(dv, dist, rank, main_run, n, barrier)
    companion = None
    if n == 8 and name == "weak" and args.workload == "auto":
        main_run["verts"] = None
        cname, cres, cnv = workload_for(n, "config4")
        cres, cnv = args.config4_resolution or cres, args.config4_nv or cnv
        companion = time_workload(cname, cres, cnv, max(args.steps // 2, 3), 2)
        companion["strong"] = same_job_on_one_gpu(dv, dist, rank, companion, n, barrier)
        companion["verts"] = None   # 1.8 GB of host memory

    if rank == 0:
        out = report(args, n, main_run, dv, comm)
        if routes is not None:
            out["routes"] = routes
        if upload is not None:
            out["config"]["upload"] = upload
        if strong is not None:
            out["strong_scaling_same_job"] = strong_entry(strong, main_run, n)
        if companion:
            sec = companion["seconds_per_step"]
            out["config4"] = {
                "workload": WORKLOAD_TEXT["config4"].format(nv=companion["nv"], T=companion["T"], res=companion["res"], n=n),
                "metric": f"Mvoxels/sec at {companion['res']}^3 grid", "value": round(companion["voxels"] / sec / 1e6, 2),
                "mtris_per_s": round(companion["T"] / sec / 1e6, 2), "ms_per_step": round(sec * 1e3, 4),
                "voxels": companion["voxels"], "stages_ms_rank0": {k: round(v, 4) for k, v in companion["stages_ms"].items()},
                "strong_scaling_same_job": strong_entry(companion["strong"], companion, n),
                "answers": "BASELINE.json north_star '>= 6x at 8 GPUs': strong_scaling_same_job.speedup of this job (DESIGN.md section 5)"}
        emit(out)
    dv.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


LINE_BUDGET = 4096                         # bytes of the one stdout line (the driver keeps a bounded tail of stdout and parses its last line)
DETAILS_FILE = os.path.join(ROOT, "bench_details.json")

# The stdout line: the contract's keys and nothing else.  Everything else report() gathers (routes, stages, pipeline, kernels_ms,
# capi_wall, published_workload, cli_wall, formulas, mix_model, stats ...) goes to the sidecar bench_details.json and to stderr.
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
             "data", "mtris_per_s", "config", "roofline", "cpu_baseline", "strong_scaling_same_job", "config4", "build_id", "details")
CONFIG_KEYS = ("workload", "resolution", "triangles", "voxels", "parallelism", "collectives")
COLLECTIVE_KEYS = ("backend", "world", "rccl_world_size", "collective_ms_rank0")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "traffic", "valu_busy", "active_lane_fraction",
                 "estimated", "stale", "source")
CPU_KEYS = ("value", "unit", "cores", "kind", "cpu_model", "value_1_thread", "sample")
STRONG_KEYS = ("one_gpu_ms", "n_gpu_ms", "speedup", "efficiency", "voxels_match")
CONFIG4_KEYS = ("workload", "value", "mtris_per_s", "ms_per_step", "voxels", "strong_scaling_same_job")


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d} if isinstance(d, dict) else d


def compact(out):
    """The object of the stdout line: the bench contract's keys (+ the two multi-GPU curves), nested objects cut to their
    contract fields.  Never more: five rounds of diagnostics appended to this line made it 24.7 KB and the driver's record of
    round 5 came out unparsed."""
    line = _pick(out, LINE_KEYS)
    cfg = _pick(out.get("config") or {}, CONFIG_KEYS)
    if isinstance(cfg.get("collectives"), dict):
        cfg["collectives"] = _pick(cfg["collectives"], COLLECTIVE_KEYS)
    line["config"] = cfg
    line["roofline"] = _pick(out.get("roofline"), ROOFLINE_KEYS)
    if isinstance(line["roofline"], dict) and isinstance(line["roofline"].get("traffic"), float):
        line["roofline"]["traffic"] = int(line["roofline"]["traffic"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _pick(out["cpu_baseline"], CPU_KEYS)
    if out.get("strong_scaling_same_job"):
        line["strong_scaling_same_job"] = _pick(out["strong_scaling_same_job"], STRONG_KEYS)
    if out.get("config4"):
        c4 = _pick(out["config4"], CONFIG4_KEYS)
        if c4.get("strong_scaling_same_job"):
            c4["strong_scaling_same_job"] = _pick(c4["strong_scaling_same_job"], STRONG_KEYS)
        line["config4"] = c4
    line["details"] = "bench_details.json (next to bench.py) and stderr"
    return line


def compact_line(out):
    """compact(out) as one JSON line of less than LINE_BUDGET bytes: free text is shortened first, then optional keys go."""
    line = compact(out)

    def dump():
        return json.dumps(line, separators=(",", ":"))
    s = dump()
    for keep in (240, 120, 60):    # free text first: every string of the line cut to `keep` characters
        if len(s) < LINE_BUDGET:
            break

        def shorten(o):
            for k, v in list(o.items()):
                if isinstance(v, dict):
                    shorten(v)
                elif isinstance(v, str) and len(v) > keep:
                    o[k] = v[:keep - 3] + "..."
        shorten(line)
        s = dump()
    for key in ("build_id", "mtris_per_s", "config4", "strong_scaling_same_job"):   # then the keys beyond the contract
        if len(s) < LINE_BUDGET:
            break
        line.pop(key, None)
        s = dump()
    if len(s) >= LINE_BUDGET:
        raise RuntimeError(f"bench line of {len(s)} bytes exceeds its budget of {LINE_BUDGET}")
    return s


def emit(out):
    """Rank 0: everything to the sidecar and to stderr, the compact line - the LAST thing written to stdout - for the driver."""
    try:
        with open(os.environ.get("O2V_BENCH_DETAILS", DETAILS_FILE), "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    except OSError as e:
        print(f"bench: details file not written ({e})", file=sys.stderr)
    print("bench details: " + json.dumps(out), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def same_job_on_one_gpu(dv, dist, rank, run, n, barrier):
    """The N-GPU job on ONE GPU: its N planned z-slabs one after the other on rank 0's device (the slab plan unsharded, then every
    slab with the plan's bounds: what o2v_hip_voxelize_sharded does on N GPUs at once), median of three repetitions.  The
    other ranks wait.  Returns None on the other ranks."""
    import time as _t
    out = None
    barrier()
    if rank == 0:
        res = run["res"]
        reps = []
        for rep in range(4):
            t0 = _t.perf_counter()
            cuts, bnd = dv.plan_slabs(res, n)
            total = 0
            for r in range(n):
                total += dv.voxelize(res, zslab=(cuts[r], cuts[r + 1]), bounds=bnd, read=False)
            if rep:   # (the first repetition sizes the buffers)
                reps.append(_t.perf_counter() - t0)
        out = {"seconds": statistics.median(reps), "voxels": int(total), "cuts": [int(c) for c in cuts]}
    barrier()
    return out


def strong_entry(strong, run, n):
    """The same job on 1 and on N GPUs (bench.py same_job_on_one_gpu): what the N GPUs gain on THAT job."""
    if not strong:
        return None
    t_n = run["seconds_per_step"]
    return {"one_gpu_ms": round(strong["seconds"] * 1e3, 4), "n_gpu_ms": round(t_n * 1e3, 4), "speedup": round(strong["seconds"] / t_n, 3),
            "efficiency": round(strong["seconds"] / t_n / n, 3), "voxels_match": strong["voxels"] == run["voxels"],
            "what": f"the job's {n} planned z-slabs one after the other on rank 0's GPU (plan + slabs, median of 3) / the {n}-GPU step"}


def upload_comparison(verts, dv, dist, backend, rank, barrier):
    """SURVEY.md section 8e asks to measure, on a multi-GPU node, how the triangle list reaches the GPUs: every rank copying it
    over its own PCIe link (what the sharded run does: o2v_hip_set_triangles per rank) against one host-to-device copy on rank
    0 followed by an RCCL broadcast over xGMI (here: torch.distributed's broadcast of the same bytes, backend nccl = RCCL).
    Wall times, max over ranks, best of 3."""
    import time as _t
    import torch
    out = {"bytes": int(verts.nbytes), "what": "triangle list to every GPU: per-rank H2D (o2v_hip_set_triangles) vs H2D on rank 0 + RCCL broadcast"}
    best = None
    for _ in range(3):
        barrier()
        t0 = _t.perf_counter()
        dv.set_triangles(verts)
        torch.cuda.synchronize()
        barrier()
        dt = _t.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["h2d_per_rank_ms"] = round(best * 1e3, 3)
    if backend == "nccl":
        try:
            host = torch.from_numpy(verts)
            dev = torch.empty_like(host, device="cuda")
            best = None
            for _ in range(3):
                barrier()
                t0 = _t.perf_counter()
                if rank == 0:
                    dev.copy_(host, non_blocking=False)
                dist.broadcast(dev, src=0)
                torch.cuda.synchronize()
                barrier()
                dt = _t.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out["h2d_rank0_plus_rccl_broadcast_ms"] = round(best * 1e3, 3)
            del dev
        except Exception as e:  # noqa: BLE001
            out["h2d_rank0_plus_rccl_broadcast_ms"] = None
            out["broadcast_error"] = f"{type(e).__name__}: {e}"
    return out


def kernel_view(name, ms, launches, alg_bytes, prof_kernels, stale):
    """One kernel of a run: live event time, algorithmic bytes, and - from the committed PMC summary of the same workload -
    measured HBM traffic and VALU instructions, each as a fraction of its peak over the live time."""
    row = {"kernel": name, "ms": round(ms, 4), "launches": launches, "algorithmic_bytes": alg_bytes,
           "hbm_frac_algorithmic": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (alg_bytes and ms > 0) else None,
           "traffic_bytes": None, "hbm_frac_traffic": None, "valu_instructions": None, "valu_frac": None}
    k = (prof_kernels or {}).get(name)
    if k and ms > 0:
        per_step = k.get("launches_per_step", 1) or 1
        if k.get("hbm_bytes") is not None:
            row["traffic_bytes"] = int(k["hbm_bytes"] * per_step)
            row["hbm_frac_traffic"] = round(row["traffic_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        sq = k.get("sq") or {}
        if sq.get("SQ_INSTS_VALU"):
            row["valu_instructions"] = int(sq["SQ_INSTS_VALU"] * per_step)
            row["valu_frac"] = round(row["valu_instructions"] / (ms * 1e-3) / 1e9 / VALU_PEAK_GINSTR, 4)
            if sq.get("SQ_THREAD_CYCLES_VALU"):
                row["active_lane_fraction"] = round(sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_INSTS_VALU"] / 64.0, 3)
            vb = valu_busy(sq, k.get("avg_us"))
            if vb is not None:
                row["valu_busy"] = vb
            mc = mix_ceiling(name)
            if mc:
                # against the rate the kernel's own instruction mix can issue at (see mix_ceiling), not the 2-cycle peak
                row["valu_frac_of_mix_ceiling"] = round(row["valu_instructions"] / (ms * 1e-3) / 1e9 / mc["ceiling_ginstr"], 4)
                row["mix_cycles_per_valu_instruction"] = mc["mix_cycles_per_valu_instruction"]
                if mc["stale"]:
                    row["mix_stale"] = True
        row["counters"] = "profiles/current.json"
        if stale:
            row["stale"] = True   # recorded with a library built from other device sources: not this kernel's counters
    return row


def route_entry(r, prof):
    """The bench line's entry for one workload run by obj2voxel_amd.workloads.run()."""
    from obj2voxel_amd import workloads
    kern, stale = profile_for(prof, r["workload"], r["stats"])
    alg = workloads.kernel_algorithmic_bytes(r["stats"], r["textured"], r["strategy"] == "BLEND", r.get("kernels_ms"))
    rows = [kernel_view(k, v["ms"], v["launches"], alg.get(k), kern, stale) for k, v in (r.get("kernels_ms") or {}).items()]
    rows.sort(key=lambda x: -x["ms"])
    return {"workload": r["workload"], "what": r["what"], "triangles": r["tris"], "resolution": r["res"], "supersampling": r["supersampling"],
            "strategy": r["strategy"], "voxels": r["voxels"], "ms_per_step": r["ms"], "mvoxels_per_s": r["mvox_s"], "mtris_per_s": r["mtris_s"],
            "passes": r["passes"], "stages_ms": r["stages_ms"], "dominant_kernel": rows[0] if rows else None, "top_kernels": rows[:4],
            "kernels_ms": {x["kernel"]: x["ms"] for x in rows}}


def report(args, n, run, dv, comm):
    T, res, nv, V = run["T"], run["res"], run["nv"], run["voxels"]
    sec = run["seconds_per_step"]
    stages_ms, st = run["stages_ms"], run["stats"]
    prof, stale = None, False
    if n == 1 and run["name"] == "config2":   # the committed PMC passes measured this workload
        prof = load_profile()
        kern_ok, stale = profile_for(prof, "config2", st)
        if kern_ok is None:
            prof = None   # --resolution / --nv changed the workload: the counters were not measured on this one
    kern = (prof or {}).get("kernels", {})

    # ---- per-stage accounting: algorithmic bytes per launch and measured HBM traffic (PMC) ------
    L, tiles, H = st["leaves"], st["tiles"], st["hits"]
    B, D, slots, Hd = st["bricks"], st["dirty_bricks"], st["pool_slots"], st["direct_hits"]
    Vr = st["voxels"]                      # this rank's voxels
    Hp = H - Hd                            # hits that took the pool -> counting sort -> ordered replay route
    REC = 16                               # sorted record: 16 bytes for a mesh without textured triangles (these workloads)
    CPB = st["grid_cells"] // max(st["bricks"], 1)   # cells per brick of the dense grids (o2v_dev_common.hpp: kBrickCells)
    direct = Hd > 0
    occ = "k_emit_occ" in (run.get("kernels_ms") or {}) or bool(st.get("certain_hits"))   # occupancy-only mode (material-less mesh)
    alg = {
        "bounds": 36 * T,
        "expand": 36 * T + 96 * L + 8 * tiles,
        # leaves and tiles staged, a job record written and read per surviving candidate, then per hit: 64-bit atomic + flag
        # byte (direct) or pool record + counter atomic + flag byte (pooled)
        # (a pooled hit: the counter atomic + its record, in the brick's slab - the first eight of a cell - or in the pool)
        "voxelize": 96 * L + 8 * tiles + 16 * st["jobs"] + (8 + 1) * Hd + (REC + 4) * Hp + (32 * slots if Hp else 0),
        # counting sort of the pooled hits (nothing to do when every hit was direct)
        # the listed bricks' counters read, the occupied cells filed; the pool's overflow hits (slots) scattered; counters reset
        "scan": (4 * CPB * D + 16 * Vr + (32 + 4 + REC) * slots + 4 * CPB * D) if Hp else 0,
        # replay of the pooled hits + emission of the 64-bit grid: flag map, the dirty bricks (8 bytes per cell) read, the occupied
        # 32-byte lane groups zeroed (at most one per voxel), records written
        "resolve": (16 * Vr + REC * Hp if Hp else 0) + ((B + 8 * CPB * D + 32 * Vr + 16 * Vr) if direct else 16 * Vr),
    }
    stage_kernels = {
        "bounds": ["k_init", "k_bounds", "k_setup"],
        "expand": ["k_count_roots", "k_expand_roots", "k_expand_nodes", "k_expand_big", "k_mark_bricks"],
        "voxelize": ["k_voxelize_occ", "k_voxelize<false>", "k_voxelize<true>"],
        "scan": ["k_scan_flags", "k_scan_bricks", "k_scatter", "k_reset_bricks"],
        "resolve": ["k_resolve<4>", "k_resolve<6>", "k_resolve_inline_list<4>", "k_resolve_inline_list<6>", "k_resolve_list16<4>", "k_resolve_list16<6>", "k_resolve_wave<32>", "k_resolve_wave<64>", "k_resolve_tiers",
                    "k_resolve_sorted", "k_resolve_big", "k_resolve_huge", "k_pick", "k_emit_max"],
    }
    if direct and not Hp:   # the max grid's flag scan runs in the "scan" interval, the emission in "resolve"
        stage_kernels["scan"] = ["k_scan_flags"]
        stage_kernels["resolve"] = ["k_emit_max"]
        alg["scan"] = 2 * B
        alg["resolve"] -= B
    if occ:
        # one byte per cell: a job record per voxel job (written, read by the filter, the live ones written and read again), a
        # byte + a flag per hit; the emission reads and zeroes 64 bytes per dirty brick and writes the records.  Root triangles
        # that are one leaf of one tile (here: all) have no Leaf / Tile records: both kernels read their 36 bytes.
        Lb = st.get("bypassed_leaves", 0)
        alg["expand"] = 36 * T + 96 * (L - Lb) + 8 * (tiles - Lb)
        if run.get("kernels_ms") and "k_expand_roots" not in run["kernels_ms"] and "k_count_roots" not in run["kernels_ms"]:
            alg["expand"] = 0   # no root stage at all (every triangle less than five voxels across: k_voxelize_occ makes and counts the leaves)
        alg["voxelize"] = 36 * T + 96 * (L - Lb) + 8 * (tiles - Lb) + 16 * st["jobs"] + 16 * (st["jobs"] - st.get("skipped_jobs", 0)) + 2 * H
        stage_kernels["resolve"] = ["k_emit_occ"]
        alg["resolve"] = 2 * CPB * D + 16 * Vr
    bound_of = {"bounds": "hbm", "expand": "hbm", "voxelize": "valu", "scan": "hbm", "resolve": "hbm"}
    stages = []
    for name in ("bounds", "expand", "voxelize", "scan", "resolve"):
        ms = stages_ms[name + "_ms"]
        if name == "voxelize" and run.get("k2_ms_timed"):
            ms = run["k2_ms_timed"]   # the kernel's own duration in the last timed step (not the stage interval of the extra steps)
        traffic = sum(kern[k]["hbm_bytes"] * kern[k].get("launches_per_step", 1) for k in stage_kernels[name] if k in kern) if kern else None
        row = {"stage": name, "kernels": [k for k in stage_kernels[name] if not kern or k in kern], "ms": round(ms, 4), "bound": bound_of[name],
               "algorithmic_bytes": int(alg[name]), "traffic_bytes": traffic,
               "gbs_algorithmic": round(alg[name] / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
               "gbs_traffic": round(traffic / (ms * 1e-3) / 1e9, 1) if (traffic and ms > 0) else None}
        stages.append(row)

    # ---- roofline of the dominant kernel ----------------------------------------------------------------------------------
    dom = max(stages, key=lambda r: r["ms"])
    kms = run.get("kernels_ms") or {}
    cands = ("k_voxelize<true>", "k_voxelize_occ", "k_voxelize<false>")
    vox_kernel = (next((k for k in cands if k in kms), None) or next((k for k in cands if k in kern), None)
                  or ("k_voxelize_occ" if occ else "k_voxelize<false>"))
    dom_kernel = vox_kernel if dom["stage"] == "voxelize" else "+".join(dom["kernels"])
    hbm_view = {"bound": "hbm", "kernel": dom_kernel, "achieved": dom["gbs_algorithmic"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round((dom["gbs_algorithmic"] or 0) / HBM_PEAK_GBS, 5), "traffic": dom["traffic_bytes"],
                "algorithmic_bytes": dom["algorithmic_bytes"], "kernel_ms": dom["ms"]}
    roofline = hbm_view
    sq = kern.get(dom_kernel, {}).get("sq") if kern else None
    if dom["stage"] == "voxelize" and not sq:
        # Not the profiled workload (N > 1, another mesh): the kernel is bound the same way, by instruction issue; its
        # instruction count is estimated from the profiled one in proportion to the voxel jobs (labelled as an estimate).
        try:
            ref = json.load(open(PROFILE_SUMMARY))
            ref_sq = (ref["kernels"].get("k_voxelize_occ") or ref["kernels"]["k_voxelize<false>"])["sq"]
            ref_jobs = ref["workload_stats"]["jobs"]
            if ref_jobs and st.get("jobs"):
                scale = st["jobs"] / ref_jobs
                ginstr = ref_sq["SQ_INSTS_VALU"] * scale / (dom["ms"] * 1e-3) / 1e9
                roofline = {"bound": "valu", "kernel": dom_kernel, "achieved": round(ginstr, 1), "peak": VALU_PEAK_GINSTR,
                            "unit": "G wave64-instr/s", "frac": round(ginstr / VALU_PEAK_GINSTR, 4), "estimated": True,
                            "valu_instructions_per_launch": int(ref_sq["SQ_INSTS_VALU"] * scale), "traffic": None, "kernel_ms": dom["ms"],
                            "source": "SQ_INSTS_VALU of the profiled N = 1 bench workload (profiles/current.json) x this rank's "
                                      "voxel jobs / that workload's voxel jobs"}
        except Exception:
            pass
    if dom["stage"] == "voxelize" and sq and sq.get("SQ_INSTS_VALU"):
        # The clip loop is float32 VALU work: what bounds it is the rate at which the SIMDs issue wave64 VALU instructions
        # (one per 2 cycles per SIMD), not HBM.  Instruction count: rocprofv3 --pmc SQ_INSTS_VALU of this command
        # (profiles/current.json); time: this run's hipEvent time of the stage.
        ginstr = sq["SQ_INSTS_VALU"] / (dom["ms"] * 1e-3) / 1e9
        roofline = {"bound": "valu", "kernel": dom_kernel, "achieved": round(ginstr, 1), "peak": VALU_PEAK_GINSTR,
                    "unit": "G wave64-instr/s", "frac": round(ginstr / VALU_PEAK_GINSTR, 4),
                    "active_lane_fraction": round(sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_INSTS_VALU"] / 64.0, 3) if sq.get("SQ_THREAD_CYCLES_VALU") else None,
                    "valu_instructions_per_launch": int(sq["SQ_INSTS_VALU"]), "traffic": dom["traffic_bytes"], "kernel_ms": dom["ms"],
                    # SIMD cycles per VALU instruction at the nominal clock.  The peak above is the rate of the cheapest class
                    # (v_mul / v_add / v_mov: 2 cycles); compares, selects, fma, min / max cost 4 (profiles/r03/valu_rates.json),
                    # so a figure near 4 means the VALU pipes are full for the instructions this kernel is made of.
                    "simd_cycles_per_valu_instruction": round(dom["ms"] * 1e-3 * CLOCK_GHZ * 1e9 * N_SIMDS / sq["SQ_INSTS_VALU"], 2),
                    # the direct evidence of the bound, from the committed counters alone (see valu_busy())
                    "valu_busy": valu_busy(sq, kern.get(dom_kernel, {}).get("avg_us")),
                    "profiled_kernel_us": kern.get(dom_kernel, {}).get("avg_us"),
                    "formulas": {"achieved": "SQ_INSTS_VALU / kernel_ms", "peak": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction",
                                 "active_lane_fraction": "SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64",
                                 "valu_busy": "SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (profiled_kernel_us x 2.4 GHz)",
                                 "counters": "profiles/current.json -> kernels[kernel].sq (rocprofv3 --pmc passes of this command); kernel_ms: two hipEvents on the kernel's own dispatch (hipExtLaunchKernelGGL), last timed step"},
                    # the instruction count is the committed one (deterministic for a build; `stale` if the build id differs), the time is live
                    "source": "counters: committed rocprofv3 --pmc summary profiles/current.json (same build id unless `stale`); kernel_ms: live hipEvents",
                    "counters_from": (prof or {}).get("source", "profiles/current.json")}
        mc = mix_ceiling(dom_kernel)
        if mc:
            # A model, not a measurement: the compiled kernel's STATIC instruction histogram priced with a microbenchmark's issue
            # cost per opcode class (tools/isa_hist.py x tools/ubench/valu_rates.hip).  valu_busy above is the measured statement.
            roofline["mix_model"] = {"mix_cycles_per_valu_instruction": mc["mix_cycles_per_valu_instruction"], "mix_ceiling": mc["ceiling_ginstr"],
                                     "frac_of_mix_ceiling": round(ginstr / mc["ceiling_ginstr"], 4), "source": mc["source"], "stale": bool(mc["stale"]),
                                     "caveat": "static histogram (every instruction of the loop counted once) x single-pattern microbenchmark"}
        if stale:
            # the summary was recorded with a library built from other device sources (o2v_hip_build_id differs): the count is
            # another kernel's - kept for orientation, labelled, to be re-measured (tools/profile_all.sh)
            roofline["stale"] = True
            roofline["estimated"] = True
    measured_total = sum(r["traffic_bytes"] for r in stages if r["traffic_bytes"]) if kern else None
    # the fixed whole-pipeline numerator of SURVEY.md section 8d (a dense 32-bit grid cleared and compacted): 8*G^3 + 16*V + 76*T.
    # The bricked grid touches only the dirty bricks, so this is a figure of merit for the design, not a bandwidth measurement.
    b_alg = 8 * res * res * res + 16 * V + 76 * T
    dev_ms = stages_ms["total_ms"]
    pipeline = {"device_ms": round(dev_ms, 4), "stages_ms": {k: round(v, 4) for k, v in stages_ms.items()},
                "algorithmic_bytes_touched": int(sum(alg.values())), "measured_traffic_bytes": measured_total,
                "gbs_measured_traffic": round(measured_total / (dev_ms * 1e-3) / 1e9, 1) if measured_total else None,
                "frac_of_hbm_peak_measured": round(measured_total / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if measured_total else None,
                "survey_numerator_bytes": b_alg, "survey_numerator_gbs": round(b_alg / (dev_ms * 1e-3) / 1e9 / n, 1),
                "survey_numerator_frac_of_hbm_peak": round(b_alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS / n, 4)}

    out = {
        "metric": f"Mvoxels/sec at {res}^3 grid", "value": round(V / sec / 1e6, 2), "unit": "Mvoxels/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "mtris_per_s": round(T / sec / 1e6, 2),
        "config": {"workload": run.get("text") or WORKLOAD_TEXT[run["name"]].format(nv=nv, T=T, res=res, n=n), "resolution": res, "triangles": T,
                   "voxels": V, "parallelism": f"zslab{n}",
                   "collectives": None if comm is None else {
                       "backend": comm.kind, "world": comm.world, "rccl_world_size": comm.world if comm.kind == "rccl" else None,
                       "plan_ms_rank0": round(stages_ms["plan_ms"], 4),
                       "collective_ms_rank0": round(stages_ms["collective_ms"], 4),
                       # device time of each collective on rank 0 (it includes waiting for the slowest rank to arrive)
                       # (measured in the extra steps made with O2V_HIP_FLAG_STAGE_TIMES: timing a collective is a wait on the host;
                       # the ranks' readiness words and the mesh bounds travel in one max-reduce, the partial histograms and the block extents in one all-gather)
                       "per_collective_ms_rank0": dict(zip(("ready_and_bounds_allreduce_28B", "histogram_and_block_extents_allgather",
                                                            "slab_counts_allgather"),
                                                           (lambda q: [round(q[0] + q[1], 4), round(q[2] + q[3], 4), round(q[4], 4)])(run.get("collective_parts_ms", [0.0] * 5))))}},
        "roofline": roofline, "roofline_hbm_view": hbm_view if roofline is not hbm_view else None, "stages": stages, "pipeline": pipeline,
        "steady_state": "steps 2.. of one uploaded mesh: buffers sized, counters zeroed behind the previous step, and the coloured-MAX routes reuse "
                        "the previous step's finding that no hit is pooled (two launches left out); a single-use obj2voxel_instance pays the first "
                        "step's price - see capi_wall.first_call_ms",
        "stats": {k: int(st[k]) for k in ("triangles", "leaves", "tiles", "bypassed_leaves", "candidates", "jobs", "skipped_jobs", "certain_hits", "hits", "voxels", "bricks", "dirty_bricks", "grid_cells", "grid_bytes")
                  if k in st},
    }
    if run.get("kernels_ms"):
        out["kernels_ms"] = {k: v["ms"] for k, v in sorted(run["kernels_ms"].items(), key=lambda kv: -kv[1]["ms"])}
    try:
        from obj2voxel_amd import hip
        out["build_id"] = hip.build_id()
        out["profile_build_id"] = (load_profile() or {}).get("build_id")
    except Exception:
        pass
    if n == 1 and not args.no_capi:
        out["capi_wall"] = capi_wall(467, 1024) if run["name"] != "config2" else capi_wall(nv, res)   # (always the stand-in mesh)
    if n == 1 and not args.no_capi:
        out["published_workload"] = published_workload()
    if n == 1 and not args.no_capi:
        out["cli_wall"] = cli_wall()
    if n == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(run["verts"], res, V, (run.get("kw") or {}).get("supersampling", 1))
    return out


def capi_wall(nv, res):
    """SURVEY.md section 8d metric (ii): wall time of the drop-in entry point obj2voxel_voxelize() (the only timing the
    reference itself reports, src/main.cpp:268-269,377-379) with a C triangle callback in and a counting C voxel callback
    out: callback pulls + H2D + device pipeline + D2H + sink calls.  PCIe-inclusive, so it is never `value`."""
    try:
        from tools import bench_capi
        r = bench_capi.measure(nv, res, reps=8, debug=os.environ.get("O2V_CAPI_DEBUG") == "1")
        later = sorted(r["wall_s"][1:])
        med = statistics.median(later)
        return {"ms": round(med * 1e3, 3), "ms_min": round(later[0] * 1e3, 3), "ms_max": round(later[-1] * 1e3, 3),
                "first_call_ms": round(r["wall_s"][0] * 1e3, 3), "ms_all": [round(t * 1e3, 3) for t in r["wall_s"]],
                "mvoxels_per_s": round(r["voxels"] / med / 1e6, 1), "mvoxels_per_s_best": round(r["voxels"] / later[0] / 1e6, 1),
                "what": "obj2voxel_voxelize(): triangle callback in, voxel callback out; median (min, max) of the seven calls after the first "
                        "(the first creates the device session and allocates the dense grids)"}
    except Exception as e:  # the helper needs gcc; the bench line must not depend on it
        return {"error": str(e)}


def cli_wall():
    """Process-level wall time of the command line front end on the headline stand-in (binary STL, -r 1024) and on the stand-in of
    the reference README's showcase run (OBJ + MTL + PNG, -r 8192): process start -> exit with the output file closed, every run
    a new process - the only kind of figure the reference publishes (README.adoc:177-178: 1.82 s) and what a CLI user gets
    (tools/bench_cli.py).  Goes to the details file, not to the stdout line."""
    try:
        from tools import bench_cli
        return bench_cli.measure("both", reps=3)
    except Exception as e:  # noqa: BLE001 - the line must not depend on it
        return {"error": f"{type(e).__name__}: {e}"}


def published_workload():
    """The only workload the reference publishes a number for (README.adoc:177-178, img/terminal_screenshot.png): 19 392
    textured triangles at r = 8192, MAX, VL32 -> 20.3 M voxels in 1.82 s end to end (the author's CPU).
    Timed here on its stand-in (obj2voxel_amd.meshes.readme_blade: 19 320 textured triangles, a long thin ellipsoid) through
    obj2voxel_voxelize() with the VL32 memory sink: wall time of the first call in a new device session and of later calls."""
    try:
        from tools import bench_capi
        from obj2voxel_amd import workloads
        r = bench_capi.measure_published(reps=5)
        later = sorted(r["wall_s"][1:])
        dev = workloads.run("readme8192", steps=5, warmup=3)   # (its own context: the first steps size the buffers and allocate the grids)
        st = dev["stats"]
        return {"workload": workloads.WORKLOADS["readme8192"][3], "entry_point": "obj2voxel_voxelize(): C triangle callback in, VL32 memory sink out",
                "triangles": r["triangles"], "resolution": 8192, "voxels": r["voxels"], "output_bytes": r["output_bytes"],
                "first_call_s": round(r["wall_s"][0], 4), "later_calls_s": [round(t, 4) for t in r["wall_s"][1:]],
                "median_s": round(statistics.median(later), 4), "mvoxels_per_s": round(r["voxels"] / statistics.median(later) / 1e6, 1),
                "device_pipeline_ms": dev["ms"], "passes": dev["passes"], "grid_bytes": int(st["grid_bytes"]), "grid_cells": int(st["grid_cells"]),
                "cube_cells": 8192 ** 3, "z_slabs": 1 if st["grid_cells"] else None,
                "reference_published_s": 1.82, "reference_published_voxels": 20_300_000,
                "reference_note": "README.adoc:177-178: the author's machine, another model of the same size class (19 392 triangles -> 20.3 M voxels); "
                                  "not measured here, beside it for orientation only"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(verts, res, expect_voxels, supersampling=1):
    """The CPU oracle (a port of the reference algorithm, oracle/o2v_oracle.c) timed on this host's cores on the
    same workload, chunk-parallel like the reference's worker pool: once with one thread (the whole workload), and three
    times each (median) with one thread per hardware thread down to one per sixteen of them - the best count is the baseline.
    Baseline only, not the optimisation target."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    oracle.build()

    def once(threads):
        oracle.set_threads(threads)
        t0 = time.perf_counter()
        vox = oracle.voxelize(verts, res, supersampling=supersampling)   # (occupancy does not depend on materials)
        return len(vox), time.perf_counter() - t0, oracle.phase_seconds()

    # One thread per hardware thread is not necessarily the fastest on a 256-thread host (two SMT siblings share a core's
    # caches and ports, and the workload is 1 500 surface chunks): the baseline is the best of five thread counts, each warmed up once (the harness keeps its
    # per-thread arrays between calls, like the reference's worker threads) and run three times (median).
    tried = {}
    for threads in sorted({max(cores // d, 1) for d in (1, 2, 4, 8, 16)}, reverse=True):
        once(threads)
        r3 = [once(threads) for _ in range(3)]
        tried[threads] = (statistics.median(t for _, t, _ in r3), r3)
    best = min(tried, key=lambda k: tried[k][0])
    med, runs = tried[best]
    n_vox = runs[0][0]
    phases = sorted(runs, key=lambda r: r[1])[1][2]   # of the median run: prelude, chunk loop (the algorithm), output join
    n1, t1, phases1 = once(1)
    oracle.set_threads(1)
    return {"value": round(n_vox / med / 1e6, 3), "unit": "Mvoxels/s", "cores": best, "kind": "port", "cpu_model": cpu_model(),
            "hardware_threads": cores, "median_s_by_threads": {str(k): round(v[0], 3) for k, v in tried.items()},
            "value_1_thread": round(n1 / t1 / 1e6, 3),
            "runs_s": [round(t, 3) for _, t, _ in runs], "run_1_thread_s": round(t1, 2),
            # the harness around the algorithm is parallel too (copy, bounds, transform, chunk binning, output join); what is
            # left of it and the Python call are the serial part
            "phases_s": {"prelude": round(phases[0], 4), "chunk_loop": round(phases[1], 4), "join": round(phases[2], 4)},
            "chunk_loop_mvoxels_per_s": round(n_vox / phases[1] / 1e6, 2) if phases[1] > 0 else None,
            "phases_1_thread_s": {"prelude": round(phases1[0], 4), "chunk_loop": round(phases1[1], 4), "join": round(phases1[2], 4)},
            "sample": f"the full workload ({len(verts)} tris at {res}^3 -> {n_vox} voxels): median of 3 runs with {best} threads (the best of {sorted(tried)}) "
                      f"over 64^3 chunks after one warm-up run, and one run with 1 thread",
            "matches_gpu_voxel_count": n_vox == expect_voxels}


if __name__ == "__main__":
    main()
