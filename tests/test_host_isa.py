"""Static check of the compiled gfx950 code of k_voxelize (no GPU needed, hipcc cross-compiles).

The clip loop issues its job prefetch by hand: one `asm` statement loads the next record into the registers that carry
it around the loop, another waits and reads them when the record is consumed (o2v_dev_k2_voxelize.hpp: take_job /
job_record).  That is only sound if the compiler never copies or reuses those registers between the two statements - a
copy would read them before the load has landed.  The parity tests on the GPU would show it at once; this test shows it
at build time: inside the loop, nothing but the two asm statements may touch the registers."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "obj2voxel_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _registers(text):
    """VGPR numbers an operand string mentions (v7, v[4:5])."""
    regs = set()
    for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(n) for n in re.findall(r"\bv(\d+)\b", text))
    return regs


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "o2v_device.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function",
           "--cuda-device-only", "-S", "o2v_device.hip", "-o", str(out)]
    subprocess.run(cmd, cwd=SRC, check=True, capture_output=True)
    return out.read_text().splitlines()


@pytest.mark.parametrize("variant", ["Lb0E", "Lb1E"])  # k_voxelize<false>, k_voxelize<true>
def test_prefetch_registers_are_only_touched_by_the_asm_statements(device_asm, variant):
    start = next(i for i, l in enumerate(device_asm) if re.match(r"^_ZN\S*k_voxelizeI" + variant + r"\S*:", l))
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].startswith(".Lfunc_end"))
    body = device_asm[start:end]
    # the asm statements: (first line, last line, text)
    blocks, i = [], 0
    while i < len(body):
        if "#ASMSTART" in body[i]:
            j = next(k for k in range(i, len(body)) if "#ASMEND" in body[k])
            blocks.append((i, j, " ".join(body[i + 1:j])))
            i = j
        i += 1
    loads = [b for b in blocks if "global_load_dwordx2" in b[2]]
    reads = [b for b in blocks if "s_waitcnt vmcnt(0)" in b[2]]
    assert len(loads) == 2 and len(reads) == 1, (len(loads), len(reads))  # before the loop, in the refill; the consumer
    dest = [re.search(r"global_load_dwordx2 (v\[\d+:\d+\])", b[2]).group(1) for b in loads]
    assert dest[0] == dest[1], dest
    carried = _registers(dest[0])
    sources = set()
    for m in re.finditer(r"v_mov_b32 v\d+, (v\d+)", reads[0][2]):
        sources |= _registers(m.group(1))
    assert sources == carried, (sources, carried)
    # the clip loop: the innermost loop that contains the consumer
    header = None
    for k in range(reads[0][0], -1, -1):
        m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", body[k])
        if m:
            header = m.group(1)
            break
    assert header
    inside = [k for k, l in enumerate(body) if ("Header=" + header + " ") in l or l.startswith(".L" + header + ":")]
    lo, hi = min(inside), max(inside)
    asm_lines = set()
    for a, b, _ in blocks:
        asm_lines.update(range(a, b + 1))
    offenders = []
    for k in range(lo, hi + 1):
        line = body[k].split(";")[0]
        if k in asm_lines or not line.strip() or line.lstrip().startswith("."):
            continue
        if _registers(line) & carried:
            offenders.append(body[k].strip())
    assert not offenders, offenders[:5]
