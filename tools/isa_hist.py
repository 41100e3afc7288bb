#!/usr/bin/env python3
"""Instruction histogram of k_voxelize's clip loop (phase 2) from the compiled gfx950 code, priced with the measured issue
costs of profiles/<round>/valu_rates.json: the "mix-weighted issue ceiling" bench.py quotes beside roofline.frac.

No GPU needed (hipcc cross-compiles).  The clip loop is found structurally: the innermost loop (a backward branch to a label)
that contains the kernel's one hand-written `s_waitcnt vmcnt(0)` (job_record(), o2v_dev_k2_voxelize.hpp), i.e. the
`for (;;)` of phase 2.  The histogram is static - every instruction of the loop body once - which is the mix a wavefront
issues when all of an iteration's branches are taken by some lane (the usual case: cut, accumulate and refill all occur in
most iterations of a full wavefront); the blocks of the append section (flush_results) are inside the loop too and are
counted, although they run once per ~48 hits.

usage: isa_hist.py [--rates profiles/r03/valu_rates.json] [--asm file.s] [--json out.json] [-v]
"""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "obj2voxel_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# Which measured loop of tools/ubench/valu_rates.hip prices an opcode (w4 column: four wavefronts per SIMD, as k_voxelize
# runs).  Opcodes that were not measured take the class of their closest measured relative; the mapping is printed with -v.
# (several candidates: the first one the rates file holds is used - round 3's file has fewer loops than round 4's)
PRICE_KEYS = [
    (r"^v_(mul|add|sub|subrev)_f32", ["ind_mul_f32"]),
    (r"^v_(or|xor)_b32", ["ind_or_b32", "ind_and_b32"]),
    (r"^v_(and|not)_b32", ["ind_and_b32"]),
    (r"^v_(add|sub|subrev)(_co)?_u32", ["ind_add_u32"]),
    (r"^v_mov_b32_dpp|_dpp$", ["ind_dpp_mov"]),
    (r"^v_mov_b32", ["ind_mov"]),
    (r"^v_lshlrev_b32", ["ind_lshlrev_b32", "ind_and_b32"]),
    (r"^v_(lshrrev|ashrrev)_b32", ["ind_lshrrev_b32", "ind_and_b32"]),
    (r"^v_accvgpr", ["ind_mov"]),
    (r"^v_rcp_|^v_rsq_|^v_sqrt_|^v_exp_|^v_log_", ["ind_rcp_f32"]),
    (r"^v_cmpx?_\w+_(u|i)(16|32|64)", ["ind_cmp_u32", "ind_cmp_sgpr"]),
    (r"^v_cmp|^v_cmpx", ["ind_cmp_sgpr"]),
    (r"^v_cndmask_b32", ["ind_cndmask_e64"]),  # (runs of the two-operand form are excluded by tests/test_host_isa.py)
    # (round 5: the 4.3 cycles of `ind_fma_f32` were its three sources sitting in one register bank; with the sources in distinct
    # banks - `bank_fma_distinct`, what the register allocator mostly achieves - v_fma_f32 issues at the rate of v_mul / v_add)
    (r"^v_(fma|fmac|mad|mac)_f32", ["bank_fma_distinct", "ind_fma_f32"]),
    (r"^v_(max|min)3_f32|^v_med3", ["ind_max3_f32", "ind_min3", "ind_max_f32"]),
    (r"^v_(max|min)_f32", ["ind_max_f32"]),
    (r"^v_(fma|mul|add)_f64", ["ind_mul_f64"]),
    (r"^v_cvt_f64_f32", ["ind_cvt_f64_f32"]),
    (r"^v_cvt_f32_f64", ["ind_cvt_f32_f64"]),
    (r"^v_cvt_f32_(u|i)32|^v_cvt_(u|i)32_f32", ["ind_cvt_f32_u32", "ind_cvt_f32_f64"]),
    (r"^v_cvt_", ["ind_cvt_f32_f64"]),
    (r"^v_div_scale", ["ind_div_scale"]),
    (r"^v_div_fixup", ["ind_div_fixup"]),
    (r"^v_div_fmas", ["dep_div_fmas"]),
    (r"^v_readlane|^v_readfirstlane|^v_writelane", ["ind_readlane"]),
    (r"^v_or3_b32|^v_bitop3", ["ind_or3_b32", "ind_lshl_or"]),
    (r"^v_and_or_b32", ["ind_and_or_b32", "ind_lshl_or"]),
    (r"^v_bfe_|^v_bfi_", ["ind_bfe_u32", "ind_lshl_or"]),
    (r"^v_mbcnt", ["ind_mbcnt", "ind_lshl_or"]),
    (r"^v_lshl_add_u64|^v_lshlrev_b64|^v_mad_u64", ["ind_lshl_add_u64", "ind_lshl_or"]),
    (r"^v_mul_u32_u24|^v_mad_u32_u24|^v_mul_lo|^v_mul_hi", ["ind_mul_u32_u24", "ind_lshl_or"]),
    (r"^v_alignbit", ["ind_alignbit"]),
    (r"^v_pk_", ["ind_pk_mul_f32"]),
]
DEFAULT_KEY = "ind_lshl_or"  # the 4-cycle class


def compile_asm(extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function",
           "--cuda-device-only", "-S", "o2v_device.hip", "-o", out, *extra]
    subprocess.run(cmd, cwd=SRC, check=True, capture_output=True)
    return out


def kernel_body(lines, variant):
    """variant: 'Lb0E' = k_voxelize<false>, 'Lb1E' = k_voxelize<true>, '_occ' = k_voxelize_occ"""
    pat = r"^_ZN\S*k_voxelize_occ\S*:" if variant == "_occ" else r"^_ZN\S*k_voxelizeI" + variant + r"\S*:"
    start = next(i for i, l in enumerate(lines) if re.match(pat, l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [l.strip() for l in lines[start + 1:end]]


def is_inst(l):
    return bool(l) and not l.startswith((";", ".", "#")) and not l.endswith(":") and not l.startswith("//")


def clip_loop(body):
    """Line indices of the clip loop: the blocks the compiler's loop annotations attribute to the innermost loop around the
    hand-written wait (`.LBBn_m: ; in Loop: Header=BBn_h Depth=d`, the header block itself, and loops nested in it)."""
    # (the kernel's last hand-written wait is the one at the job start inside the loop; an earlier one fills the pipeline)
    marker = [i for i, l in enumerate(body) if l == "s_waitcnt vmcnt(0)" and "#ASMSTART" in body[i - 1]][-1]
    # block starts and the loop header each block is attributed to
    blocks = []  # (line, label, header or None, is_header)
    for i, l in enumerate(body):
        m = re.match(r"^\.(LBB\d+_\d+):\s*(;.*)?$", l)
        if not m:
            continue
        ann = " ".join(x for x in [m.group(2) or ""] + [body[k] for k in range(i + 1, min(i + 6, len(body))) if body[k].startswith(";")])
        h = re.search(r"in Loop: Header=(BB\d+_\d+)", ann)
        is_header = "Loop Header" in ann
        blocks.append((i, m.group(1)[1:], h.group(1) if h else None, is_header))
    cur = max((b for b in blocks if b[0] <= marker), key=lambda b: b[0])
    header = cur[1] if cur[3] else cur[2]
    if header is None:
        raise RuntimeError("clip loop not found")
    # loops nested inside it: blocks whose header's own parent chain reaches `header`
    parent = {}
    for i, lab, h, is_h in blocks:
        if is_h:
            anns = [body[k] for k in range(i, min(i + 8, len(body))) if "Parent Loop" in body[k]]
            par = [re.search(r"Parent Loop (BB\d+_\d+)", a).group(1) for a in anns]
            parent[lab] = par  # outermost first
    inside = {header} | {lab for lab, par in parent.items() if header in par}
    idx = []
    for n, (i, lab, h, is_h) in enumerate(blocks):
        end = blocks[n + 1][0] if n + 1 < len(blocks) else len(body)
        if (is_h and lab in inside) or (not is_h and h in inside):
            idx.extend(range(i, end))
    return idx


def price_key(op, rates=None):
    for pat, keys in PRICE_KEYS:
        if re.search(pat, op):
            for key in keys:
                if rates is None or key in rates:
                    return key
    return DEFAULT_KEY


def histogram(body, idx):
    ops = collections.Counter()
    for i in idx:
        l = body[i]
        if is_inst(l):
            ops[l.split()[0]] += 1
    return ops


def summarize(ops, rates):
    valu = {o: n for o, n in ops.items() if o.startswith("v_")}
    cost = {}
    for o in valu:
        k = price_key(o, rates)
        r = rates.get(k) or rates[DEFAULT_KEY]
        cost[o] = (k, r["w4"])
    n_valu = sum(valu.values())
    cycles = sum(n * cost[o][1] for o, n in valu.items())
    classes = collections.Counter()
    for o, n in valu.items():
        classes["cheap (<3 cycles)" if cost[o][1] < 3.0 else ("slow (>6 cycles)" if cost[o][1] > 6.0 else "4-cycle")] += n
    return {
        "valu": n_valu,
        "salu": sum(n for o, n in ops.items() if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier"))),
        "branch": sum(n for o, n in ops.items() if o.startswith(("s_cbranch", "s_branch"))),
        "lds": sum(n for o, n in ops.items() if o.startswith("ds_")),
        "vmem": sum(n for o, n in ops.items() if o.startswith(("global_", "buffer_", "scratch_", "flat_"))),
        "valu_classes": dict(classes),
        "mix_cycles_per_valu": cycles / n_valu if n_valu else None,
        "valu_by_opcode": {o: {"count": n, "priced_as": cost[o][0], "cycles": cost[o][1]} for o, n in sorted(valu.items(), key=lambda x: -x[1])},
    }


def analyze(asm_path, rates):
    lines = open(asm_path).read().splitlines()
    out = {}
    for variant, name in (("Lb0E", "k_voxelize<false>"), ("Lb1E", "k_voxelize<true>"), ("_occ", "k_voxelize_occ")):
        body = kernel_body(lines, variant)
        s = summarize(histogram(body, clip_loop(body)), rates)
        whole = summarize(histogram(body, range(len(body))), rates)
        s["whole_kernel_valu"] = whole["valu"]
        s["whole_kernel_mix_cycles_per_valu"] = whole["mix_cycles_per_valu"]
        out[name] = s
    return out


def source_build_id():
    """The hash the Makefile gives the library (o2v_hip_build_id): sha256 of the device sources, first 16 hex digits."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(SRC, "o2v_device.hip")] + sorted(glob.glob(os.path.join(SRC, "o2v_dev_*.hpp"))) + [os.path.join(SRC, "o2v_math.h")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r04", "valu_rates.json"))
    ap.add_argument("--asm")
    ap.add_argument("--json")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    rates = json.load(open(a.rates))["results"]
    asm = a.asm or compile_asm()
    res = analyze(asm, rates)
    res["rates_file"] = os.path.relpath(a.rates, ROOT)
    res["build_id"] = source_build_id()
    res["note"] = ("static histogram of the clip loop (phase 2 of k_voxelize), priced with the w4 column of the rates file; "
                   "mix_cycles_per_valu x (VALU instructions of a launch) / (SIMDs x clock) is the mix-weighted minimum issue time. "
                   "A model: the histogram is static (every instruction of the loop counted once, whatever its branch executes "
                   "how often), and k_voxelize_occ spends most of its time outside this loop (phase 1): for it use "
                   "whole_kernel_mix_cycles_per_valu, with the same caveat")
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    for k in ("k_voxelize<false>", "k_voxelize<true>", "k_voxelize_occ"):
        s = res[k]
        print(f"{k}: clip loop {s['valu']} VALU ({s['valu_classes']}), {s['salu']} SALU, {s['branch']} branches, {s['lds']} LDS, "
              f"{s['vmem']} VMEM; mix {s['mix_cycles_per_valu']:.2f} cycles per VALU instruction (whole kernel: {s['whole_kernel_valu']} VALU, "
              f"{s['whole_kernel_mix_cycles_per_valu']:.2f})")
        if a.v:
            for o, d in list(s["valu_by_opcode"].items()):
                print(f"   {d['count']:5d}  {o:28s} {d['cycles']:.2f}  ({d['priced_as']})")


if __name__ == "__main__":
    main()
