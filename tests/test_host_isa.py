"""Static check of the compiled gfx950 code of k_voxelize (no GPU needed, hipcc cross-compiles).

The clip loop prefetches every lane's next job record with LDS-DMA (global_load_lds_dword: global memory -> the lane's LDS
slot, no destination register) and waits for it explicitly right before the slot is read (o2v_dev_k2_voxelize.hpp: take_job /
job_record).  hipcc's own s_waitcnt placement for LDS-DMA was seen to guard the wrong LDS reads across the loop's back edge,
hence the explicit wait; this test pins the shape of the compiled code: the DMA loads are there, the only inline asm of the
kernel besides the selects with a pinned encoding (vsel, o2v_dev_arith.hpp) is that wait, and the reads of the slot follow it
in the same basic block.  A second test pins what vsel is for: no long runs of two-operand v_cndmask in the clip loop."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "obj2voxel_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "o2v_device.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function",
           "--cuda-device-only", "-S", "o2v_device.hip", "-o", str(out)]
    subprocess.run(cmd, cwd=SRC, check=True, capture_output=True)
    return out.read_text().splitlines()


def _kernel_start(device_asm, variant):
    """variant: 'Lb0E' = k_voxelize<false>, 'Lb1E' = k_voxelize<true>, '_occ' = k_voxelize_occ (the occupancy-only route's kernel)"""
    pat = r"^_ZN\S*k_voxelize_occ\S*:" if variant == "_occ" else r"^_ZN\S*k_voxelizeI" + variant + r"\S*:"
    return next(i for i, l in enumerate(device_asm) if re.match(pat, l))


@pytest.mark.parametrize("variant", ["Lb0E", "Lb1E", "_occ"])
def test_job_prefetch_is_lds_dma_with_an_explicit_wait(device_asm, variant):
    start = _kernel_start(device_asm, variant)
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].startswith(".Lfunc_end"))
    body = [l.strip() for l in device_asm[start:end]]
    dma = [i for i, l in enumerate(body) if l.startswith("global_load_lds_dword")]
    assert len(dma) == 4, dma          # two dwords per record; issued before the loop and in the refill
    # no load of the record into registers by hand (the scheme this replaces), no other inline asm in the kernel
    blocks = [i for i, l in enumerate(body) if "#ASMSTART" in l]
    other = [i for i in blocks if not body[i + 1].startswith("v_cndmask_b32_e64")]
    assert len(other) == 1, [body[i + 1] for i in other]
    i = other[0]
    assert body[i + 1] == "s_waitcnt vmcnt(0)" and "#ASMEND" in body[i + 2]
    # the slot reads come after the wait, before control flow leaves the block
    rest = body[i + 3:i + 40]
    stop = next((k for k, l in enumerate(rest) if l.startswith(("s_cbranch", "s_branch", ".LBB"))), len(rest))
    reads = [l for l in rest[:stop] if l.startswith("ds_read_b32")]
    assert len(reads) >= 2, rest[:stop]


@pytest.mark.parametrize("variant", ["Lb0E", "Lb1E", "_occ"])
def test_no_runs_of_two_operand_selects(device_asm, variant):
    """A VOP2 v_cndmask directly after another costs 16+ cycles of its SIMD on gfx950 (profiles/r03/valu_rates.json); the
    pieces of the clip loop are selected with the three-operand encoding (vsel), so no run of more than two remains."""
    start = _kernel_start(device_asm, variant)
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].startswith(".Lfunc_end"))
    ops = [l.split()[0] for l in (x.strip() for x in device_asm[start:end]) if l and not l.startswith((";", ".")) and not l.endswith(":")]
    longest = run = 0
    for op in ops:
        run = run + 1 if op == "v_cndmask_b32_e32" else 0
        longest = max(longest, run)
    assert longest <= 2, longest


@pytest.mark.parametrize("variant,min_alignbit", [("Lb0E", 20), ("Lb1E", 12), ("_occ", 20)])
def test_piece_masks_are_built_from_sign_bits(device_asm, variant, min_alignbit):
    """piece_masks reads its conditions off the sign bits of differences: one v_alignbit_b32 per condition instead of a
    compare and a select (profiles/r04/NOTES.md: -4 .. -7 % on the clip kernel).  If the compiler turns that back into
    compares, the instruction mix the bench line's mix ceiling is priced with changes: this pins the form."""
    start = _kernel_start(device_asm, variant)
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].startswith(".Lfunc_end"))
    ops = [l.split()[0] for l in (x.strip() for x in device_asm[start:end]) if l and not l.startswith((";", ".")) and not l.endswith(":")]
    assert sum(op == "v_alignbit_b32" for op in ops) >= min_alignbit, sum(op == "v_alignbit_b32" for op in ops)


def test_wavefront_prefix_sums_use_dpp(device_asm):
    """wave_inclusive_scan (o2v_dev_common.hpp): four row_shr adds and the two row broadcasts instead of six ds_bpermute steps -
    phase 1 of the clip kernels runs one such scan per 64 candidate rows and waits on its dependent chains
    (profiles/r05/NOTES.md).  Pins that the compiler emits the DPP forms on gfx950."""
    start = _kernel_start(device_asm, "_occ")
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].startswith(".Lfunc_end"))
    body = [l.strip() for l in device_asm[start:end]]
    assert sum("row_bcast:15" in l for l in body) >= 1 and sum("row_bcast:31" in l for l in body) >= 1
    assert sum("row_shr:8" in l for l in body) >= 1


def _kernel_body(device_asm, name):
    start = next(i for i, l in enumerate(device_asm) if re.match(r"^_ZN\S*" + name + r"\S*:", l))
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].startswith(".Lfunc_end"))
    return [l.strip() for l in device_asm[start:end]]


def test_flag_scan_has_no_barrier_in_its_loop(device_asm):
    """k_scan_flags (o2v_dev_k5_scan_scatter.hpp): every wavefront scans its own share of the dirty map - list positions from a DPP
    prefix sum, staging in its own LDS list - so the only workgroup barriers are the two around the final reservation, and there is
    no LDS atomic (the barrier version's list positions).  Four 16-byte loads per lane are issued together."""
    body = _kernel_body(device_asm, "k_scan_flags")
    assert sum(l.startswith("s_barrier") for l in body) <= 2, [l for l in body if l.startswith("s_barrier")]
    assert not any(l.startswith("ds_add") for l in body)
    assert sum("row_bcast:31" in l for l in body) >= 1
    assert sum(l.startswith("global_load_dwordx4") for l in body) >= 4


def test_count_roots_runs_without_lds_staging(device_asm):
    """k_count_roots (o2v_dev_k1_expand.hpp): a wavefront per block of 256 triangles, the 36 loads of a lane's four triangles in
    flight together, no LDS staging and no barrier in the loop (the one barrier pair belongs to the final sums)."""
    body = _kernel_body(device_asm, "k_count_roots")
    assert sum(l.startswith("s_barrier") for l in body) <= 2
    assert not any(l.startswith("ds_write_b32") or l.startswith("ds_read_b32") for l in body)
    assert sum(l.startswith("global_load_dword ") or l.startswith("global_load_dword\t") for l in body) >= 36
