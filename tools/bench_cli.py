#!/usr/bin/env python3
"""Process-level wall time of the command line front end: `obj2voxel-amd MODEL out.vl32 -r N`, process start to exit with the
output file closed - what a user of the reference's CLI sees, and the only kind of number the reference publishes
(README.adoc:177-178 / img/terminal_screenshot.png: 1.82 s for 19 392 textured triangles at -r 8192; the CLI prints it
itself, src/main.cpp:268-269,377-379).  Every run is a new process: HIP runtime start, device session, allocation of the
dense grids, file parsing, device pipeline, read-back, file writing - a CLI run never sees bench.py's steady state.

Workloads: the headline stand-in as a binary STL (870 488 triangles, -r 1024) and the stand-in of the README's showcase run as
OBJ + MTL + PNG (19 320 textured triangles, -r 8192).  Files go to a directory in /dev/shm (or $TMPDIR): the figure is the
program, not the disk.

usage: tools/bench_cli.py [headline|readme|both] [--reps N]      (prints one JSON object)"""
import json
import os
import re
import shutil
import statistics
import struct
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_stl(path, verts):
    import numpy as np
    rec = np.zeros(len(verts), dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", 9), ("a", "<u2")]))
    rec["v"] = verts
    with open(path, "wb") as f:
        f.write(b"binary stl".ljust(80, b" ") + struct.pack("<I", len(verts)))
        f.write(rec.tobytes())


def png_rgb(rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_textured_obj(directory, verts, uvs, texture):
    """OBJ + MTL + PNG of a fully textured mesh (one vertex / vt triple per triangle, as the tests write it)."""
    with open(os.path.join(directory, "tex.png"), "wb") as f:
        f.write(png_rgb(texture))
    with open(os.path.join(directory, "model.mtl"), "w") as f:
        f.write("newmtl skin\nKd 1 1 1\nmap_Kd tex.png\n")
    lines = ["mtllib model.mtl"]
    for t in range(len(verts)):
        for k in range(3):
            lines.append("v %.9g %.9g %.9g" % tuple(float(x) for x in verts[t, k * 3:k * 3 + 3]))
            lines.append("vt %.9g %.9g" % tuple(float(x) for x in uvs[t, k * 2:k * 2 + 2]))
    lines.append("usemtl skin")
    for t in range(len(verts)):
        i = 3 * t + 1
        lines.append(f"f {i}/{i} {i + 1}/{i + 1} {i + 2}/{i + 2}")
    path = os.path.join(directory, "model.obj")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def cli_path():
    import obj2voxel_amd
    return os.path.join(os.path.dirname(obj2voxel_amd.LIB_PATH), "obj2voxel-amd")


PHASE = re.compile(r"host phases: (.*)")


def run_cli(model, out, res, reps, extra=()):
    """`reps` processes one after the other; per run: wall (perf_counter around the process), the CLI's own figure, the output
    size; of the LAST run (made with -v) the library's host phases."""
    cli = cli_path()
    walls, own, phases = [], [], []
    size = 0
    for rep in range(reps + 1):
        verbose = rep == reps          # one extra run with -v: its phases are reported, its wall time is not
        if os.path.exists(out):
            os.remove(out)
        cmd = [cli, model, out, "-r", str(res), *extra] + (["-v"] if verbose else [])
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)} -> {r.returncode}: {(r.stdout + r.stderr)[-400:]}")
        if verbose:
            phases = PHASE.findall(r.stdout + r.stderr)
            continue
        walls.append(dt)
        m = re.search(r"Done! \(([0-9.]+) s\)", r.stdout)
        own.append(float(m.group(1)) if m else None)
        size = os.path.getsize(out)
    return {"wall_s": [round(t, 4) for t in walls], "wall_s_median": round(statistics.median(walls), 4), "wall_s_min": round(min(walls), 4),
            "cli_reported_s": own, "output_bytes": size, "voxels": size // 16, "host_phases_of_a_verbose_run": phases,
            "command": f"obj2voxel-amd {os.path.basename(model)} {os.path.basename(out)} -r {res}" + ("".join(" " + e for e in extra))}


def measure(which="both", reps=3):
    from obj2voxel_amd import meshes
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    work = tempfile.mkdtemp(prefix="o2v_cli_", dir=base)
    out = {"what": "process-level wall time of the CLI (process start -> exit, output file closed), every run a new process; "
                   "median of `reps` runs; files in " + ("/dev/shm" if base else "the temporary directory"), "reps": reps}
    try:
        if which in ("headline", "both"):
            verts = meshes.uv_sphere(467)
            stl = os.path.join(work, "headline.stl")
            write_stl(stl, verts)
            e = run_cli(stl, os.path.join(work, "headline.vl32"), 1024, reps)
            e.update({"workload": "BASELINE configs[2] stand-in as a binary STL", "triangles": len(verts), "resolution": 1024,
                      "input_bytes": os.path.getsize(stl), "mvoxels_per_s": round(e["voxels"] / e["wall_s_median"] / 1e6, 1)})
            out["headline"] = e
        if which in ("readme", "both"):
            verts, uvs = meshes.readme_blade()
            obj = write_textured_obj(work, verts, uvs, meshes.checker_texture(1024, 32))
            e = run_cli(obj, os.path.join(work, "readme.vl32"), 8192, reps)
            e.update({"workload": "stand-in of the reference README's showcase run (README.adoc:177-178) as OBJ + MTL + PNG", "triangles": len(verts),
                      "resolution": 8192, "input_bytes": os.path.getsize(obj), "mvoxels_per_s": round(e["voxels"] / e["wall_s_median"] / 1e6, 1),
                      "reference_published_s": 1.82, "reference_published_voxels": 20_300_000,
                      "reference_note": "the author's CPU and model (19 392 triangles -> 20.3 M voxels); beside it for orientation only"})
            out["readme"] = e
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


if __name__ == "__main__":
    which = next((a for a in sys.argv[1:] if a in ("headline", "readme", "both")), "both")
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    print(json.dumps(measure(which, reps), indent=1))
