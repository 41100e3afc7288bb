"""bench.py on the GPU: the routes of the N = 1 line, per-kernel event times, and the $O2V_ASSETS hook (SURVEY.md section 8d:
"if $O2V_ASSETS/{spot,dragon,sponza}.obj exist on the GPU box, also run those")."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from obj2voxel_amd import meshes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_obj(path, v):
    lines = []
    for t in v:
        for k in range(3):
            lines.append("v %r %r %r" % tuple(float(x) for x in t[k * 3:k * 3 + 3]))
    lines += [f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}" for i in range(len(v))]
    path.write_text("\n".join(lines) + "\n")


def _stdout_line(r):
    """What the driver parses: the LAST line of stdout - one compact JSON object of the contract's keys, under 4 KB."""
    import bench
    lines = r.stdout.strip().splitlines()
    last = lines[-1]
    assert len(last) < bench.LINE_BUDGET, len(last)
    assert sum(1 for ln in lines if ln.startswith("{")) == 1, "one JSON line on stdout"
    line = json.loads(last)
    assert set(line) <= set(bench.LINE_KEYS), sorted(set(line) - set(bench.LINE_KEYS))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in line, k
    assert set(line["roofline"]) <= set(bench.ROOFLINE_KEYS) and set(line["config"]) <= set(bench.CONFIG_KEYS)
    return line


def test_default_bench_stdout_is_the_compact_line(tmp_path):
    """`python bench.py --steps 2` as the driver runs it (every leg on: routes, capi, published workload, cpu baseline): the last
    stdout line is the compact object with `roofline` and `cpu_baseline`; the details are in the sidecar and on stderr."""
    env = dict(os.environ, O2V_BENCH_DETAILS=str(tmp_path / "details.json"))
    env.pop("O2V_ASSETS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--route-steps", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _stdout_line(r)
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["dtype"] == "f32" and line["vs_baseline"] is None
    assert "configs[2]" in line["config"]["workload"] and line["config"]["resolution"] == 1024 and line["config"]["voxels"] == 4936186
    rf = line["roofline"]
    assert rf["kernel"] == "k_voxelize_occ" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["kernel_ms"] > 0
    cb = line["cpu_baseline"]
    assert set(cb) == set(bench_keys("CPU_KEYS")) and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    det = json.load(open(tmp_path / "details.json"))
    assert {"routes", "stages", "pipeline", "capi_wall", "published_workload", "kernels_ms", "stats"} <= set(det)
    assert "bench details: " in r.stderr


def bench_keys(name):
    import bench
    return getattr(bench, name)


def test_kernel_times_cover_the_pipeline():
    from obj2voxel_amd import hip
    d = hip.DeviceVoxelizer(0)
    try:
        v = meshes.uv_sphere(60)
        d.set_triangles(v, types=np.full(len(v), 2, np.uint32), colors=meshes.triangle_colors(len(v)))
        d.voxelize(256, strategy=1, read=False)
        assert d.kernel_times() == {}                       # only on request
        d.voxelize(256, strategy=1, read=False, kernel_times=True)
        kt = d.kernel_times()
        tm = d.timings()
    finally:
        d.close()
    for k in ("k_bounds", "k_expand_roots", "k_voxelize<false>", "k_scan_bricks", "k_scatter", "k_resolve<4>"):
        assert k in kt and kt[k][0] > 0 and kt[k][1] >= 1, (k, kt)
    assert abs(kt["k_voxelize<false>"][0] - tm["voxelize_ms"]) < 0.05 + 0.2 * tm["voxelize_ms"]


def test_stage_times_only_on_request():
    """o2v_hip_timings: the clip kernel's duration is measured in every call (events on its own dispatch), the stage intervals and
    total_ms only with O2V_HIP_FLAG_STAGE_TIMES (an event between two kernels costs device time)."""
    from obj2voxel_amd import hip
    d = hip.DeviceVoxelizer(0)
    try:
        d.set_triangles(meshes.uv_sphere(120))
        d.voxelize(512, read=False)
        d.voxelize(512, read=False)
        plain = d.timings()
        d.voxelize(512, read=False, stage_times=True)
        staged = d.timings()
    finally:
        d.close()
    assert plain["total_ms"] == 0 and plain["bounds_ms"] == 0 and plain["resolve_ms"] == 0
    assert plain["voxelize_ms"] > 0
    assert all(staged[k] > 0 for k in ("bounds_ms", "expand_ms", "voxelize_ms", "scan_ms", "resolve_ms", "total_ms"))
    assert staged["total_ms"] >= staged["voxelize_ms"]
    # the same kernel on the same input, measured both ways
    assert abs(plain["voxelize_ms"] - staged["voxelize_ms"]) < 0.02 + 0.25 * staged["voxelize_ms"], (plain, staged)


def test_bench_line_routes_and_assets(tmp_path):
    _write_obj(tmp_path / "dragon.obj", meshes.uv_sphere(40))
    _write_obj(tmp_path / "spot.obj", meshes.uv_sphere(16))
    env = dict(os.environ, O2V_ASSETS=str(tmp_path), O2V_BENCH_DETAILS=str(tmp_path / "details.json"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--route-steps", "1",
                        "--no-cpu-baseline", "--no-capi"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    short = _stdout_line(r)
    line = json.load(open(tmp_path / "details.json"))          # routes, kernels_ms ...: the sidecar, not the stdout line
    assert short["config"]["workload"] == line["config"]["workload"] and short["value"] == line["value"]
    assert "dragon.obj" in line["config"]["workload"] and line["config"]["triangles"] == len(meshes.uv_sphere(40))
    names = [x["workload"] for x in line["routes"]]
    assert names[0] == "config2" and "asset:spot" in names and "asset:sponza" not in names
    for want in ("config2_colored_max", "config2_blend", "config2_textured_max", "config1", "config3", "readme8192"):
        assert want in names
    for x in line["routes"]:
        assert "error" not in x, x
        assert x["ms_per_step"] > 0 and x["voxels"] > 0 and x["dominant_kernel"]["ms"] > 0
        assert x["dominant_kernel"]["kernel"] in x["kernels_ms"]
    blend = line["routes"][names.index("config2_blend")]
    assert {"k_scatter", "k_scan_bricks", "k_resolve<4>"} <= set(blend["kernels_ms"])
    assert line["build_id"] and "kernels_ms" in line


def test_bench_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """The N > 1 bench plumbing on a single-GPU box: two processes (torch.distributed launch, gloo), both on GPU 0, the
    library's collectives through host-memory callbacks.  The line carries the world size, the per-collective times and the
    upload comparison (the RCCL broadcast leg needs the nccl backend and is absent here)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--same-device", "--workload", "weak", "--resolution", "256", "--nv", "120"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, O2V_BENCH_DETAILS=str(tmp_path / "details.json")))
    assert r.returncode == 0, r.stderr[-3000:]
    short = _stdout_line(r)
    assert short["n_gpus"] == 2 and short["strong_scaling_same_job"]["voxels_match"] is True
    assert short["config"]["collectives"]["world"] == 2 and "upload" not in short["config"]
    line = json.load(open(tmp_path / "details.json"))
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "zslab2" and line["value"] > 0
    col = line["config"]["collectives"]
    assert col["world"] == 2 and col["backend"] == "callbacks"
    assert set(col["per_collective_ms_rank0"]) == {"ready_and_bounds_allreduce_28B", "histogram_and_block_extents_allgather", "slab_counts_allgather"}
    assert all(v > 0 for v in col["per_collective_ms_rank0"].values()), col
    assert sum(col["per_collective_ms_rank0"].values()) > 0
    up = line["config"]["upload"]
    assert up["h2d_per_rank_ms"] > 0 and "h2d_rank0_plus_rccl_broadcast_ms" not in up
    # both curves: the weak-scaling series' value, and what the two "GPUs" gain on this very job (here they share one device,
    # so the figure itself means nothing; the job's slabs run one after the other must give the same voxels)
    assert line["scaling"] == "weak" and col["rccl_world_size"] is None
    ss = line["strong_scaling_same_job"]
    assert ss["voxels_match"] is True and ss["one_gpu_ms"] > 0 and ss["n_gpu_ms"] > 0 and abs(ss["speedup"] - ss["one_gpu_ms"] / ss["n_gpu_ms"]) < 0.01


def test_bench_eight_ranks_on_one_gpu_prints_a_compact_line_with_both_curves(tmp_path):
    """The first real 8-GPU run must not be lost to the line: eight processes (torch.distributed launch, gloo, all on GPU 0, the
    jobs scaled down by the test-only overrides) go through everything `bench.py --gpus 8` does - the weak-scaling job, its
    same-job strong scaling, the upload comparison and the configs[4] companion job with its own strong scaling - and rank 0's
    LAST stdout line is the compact object: n_gpus 8, both curves, config4, under 4 KB; nobody else prints to stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--same-device", "--resolution", "256", "--nv", "120", "--config4-resolution", "384", "--config4-nv", "200"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, O2V_BENCH_DETAILS=str(tmp_path / "details.json")))
    assert r.returncode == 0, r.stderr[-3000:]
    line = _stdout_line(r)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["parallelism"] == "zslab8" and line["value"] > 0
    assert line["config"]["collectives"]["world"] == 8 and "rccl_world_size" in line["config"]["collectives"]
    ss = line["strong_scaling_same_job"]
    assert ss["voxels_match"] is True and ss["speedup"] > 0
    c4 = line["config4"]
    assert c4["value"] > 0 and c4["voxels"] > 0 and c4["strong_scaling_same_job"]["voxels_match"] is True
    det = json.load(open(tmp_path / "details.json"))
    assert det["config4"]["answers"] and det["config"]["upload"]["h2d_per_rank_ms"] > 0
