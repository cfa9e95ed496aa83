#!/usr/bin/env python3
"""Static instruction census of the gfx950 code of one kernel, by loop and by instruction class.

    tools/isa_census.py [--kernel k_rates] [--variant 'ILb0ELb0E'] [--out profiles/rNN_isa_census.txt]

Compiles csrc/c2ray_hip.hip with `hipcc --save-temps` (cross-compiles without a GPU), takes the kernel's
text from the .s file and counts, for the whole kernel and for every loop LLVM annotates ("in Loop:
Header=BBx_y Depth=d"), the instructions of each class:

    f64     v_add/mul/fma/fmac_f64, v_max/min_f64, v_div_*      (full rate: 4 cycles per wave64 instruction)
    trans   v_rcp/rsq/sqrt_f64                                  (quarter rate)
    cvt     v_cvt_*, v_frexp_*, v_ldexp_*
    cmp     v_cmp_*
    mov     v_mov_*, v_cndmask_*, v_readlane/readfirstlane, v_accvgpr_*
    int     every other VALU instruction (address / exponent bookkeeping)
    vmem    global_/scratch_/buffer_/flat_ loads and stores
    lds     ds_*
    smem    s_load_*, s_buffer_load_*
    salu    every other s_* instruction except branches / waitcnt / nop
    branch  s_branch, s_cbranch_*
    wait    s_waitcnt, s_nop

A static count is an upper bound of one trip through a loop body (both sides of every divergent branch are
counted); the dynamic figure per band iteration comes from the SQ_INSTS_VALU counter (tools/pmc_summary.py).
"""
from __future__ import annotations

import argparse
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "c2-ray3dm1d_helium_amd" / "csrc" / "c2ray_hip.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]
CLASSES = ["f64", "trans", "cvt", "cmp", "mov", "int", "vmem", "lds", "smem", "salu", "branch", "wait"]


def classify(op: str) -> str | None:
    if op.startswith("v_"):
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
            return "trans"
        if re.match(r"v_(add|mul|fma|fmac|max|min|div_scale|div_fmas|div_fixup)_f64", op):
            return "f64"
        if re.match(r"v_(cvt|frexp|ldexp|trunc|floor|rndne|fract)_", op):
            return "cvt"
        if op.startswith("v_cmp"):
            return "cmp"
        if re.match(r"v_(mov|cndmask|readlane|readfirstlane|writelane|accvgpr)", op):
            return "mov"
        return "int"
    if re.match(r"(global|scratch|buffer|flat)_", op):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if re.match(r"s_(load|buffer_load|scratch_load|store)", op):
        return "smem"
    if re.match(r"s_(branch|cbranch|setpc|swappc|endpgm|call)", op):
        return "branch"
    if re.match(r"s_(waitcnt|nop|sleep|barrier)", op):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return None


def kernel_text(asm: str, kernel: str, variant: str) -> tuple[str, list[str]]:
    lines = asm.splitlines()
    start = None
    name = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w*" + re.escape(kernel) + r"\w*):", ln)
        if m and (not variant or variant in m.group(1)):
            start, name = i, m.group(1)
            break
    if start is None:
        raise SystemExit(f"kernel {kernel} ({variant}) not found")
    out = []
    for ln in lines[start + 1:]:
        if ln.strip().startswith(".end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            break
        out.append(ln)
    return name, out


def census(body: list[str]):
    """-> (per-loop counters, block list).  Loop key: (header, depth); ("kernel", 0) holds everything."""
    loops = collections.defaultdict(collections.Counter)
    blocks = []
    parent = {}             # loop -> enclosing loop
    loop_of_header = {}     # ".LBBx_y" -> (header, depth)
    cur_loop = None
    cur_block = None
    pending_parent = None
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)$", ln)
        s = ln.strip()
        if m:
            cur_block = [m.group(1), m.group(2).strip(), collections.Counter()]
            blocks.append(cur_block)
            cur_loop = None
            pending_parent = None
            s = ";" + m.group(2)
        if s.startswith(";") and cur_block is not None:
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", s)
            if mm:
                cur_loop = (".L" + mm.group(1), int(mm.group(2)))
            mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", s)
            if mm:
                pending_parent = (".L" + mm.group(1), int(mm.group(2)))
            mm = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", s)
            if mm:
                cur_loop = (cur_block[0], int(mm.group(1)))
                loop_of_header[cur_block[0]] = cur_loop
                if pending_parent is not None:
                    parent[cur_loop] = pending_parent
            continue
        if not s or s.startswith("."):
            continue
        cl = classify(s.split()[0])
        if cl is None:
            continue
        loops[("kernel", 0)][cl] += 1
        if cur_block is not None:
            cur_block[2][cl] += 1
        lp = cur_loop
        while lp is not None:   # an inner loop's instructions also belong to the loops around it
            loops[lp][cl] += 1
            lp = parent.get(lp)
    return loops, blocks


def fmt(c: collections.Counter) -> str:
    valu = sum(c[k] for k in ("f64", "trans", "cvt", "cmp", "mov", "int"))
    return f"VALU {valu:5d} (" + " ".join(f"{k} {c[k]}" for k in ("f64", "trans", "cvt", "cmp", "mov", "int")) + ")  " + \
           " ".join(f"{k} {c[k]}" for k in ("vmem", "lds", "smem", "salu", "branch", "wait"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_rates")
    ap.add_argument("--variant", default="ILb0ELb0E", help="substring of the mangled name (template arguments)")
    ap.add_argument("--asm", help="use this .s file instead of compiling")
    ap.add_argument("--out")
    ap.add_argument("--blocks", action="store_true", help="also list every basic block")
    ap.add_argument("--scratch-scan", action="store_true", help="list every kernel of the library that has a private segment")
    a = ap.parse_args()
    if a.asm:
        asm = Path(a.asm).read_text()
    else:
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run(["hipcc", *FLAGS, "--save-temps", "-o", f"{td}/lib.so", str(SRC)], cwd=td,
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(r.stderr)
            asm = next(Path(td).glob("*gfx950.s")).read_text()
    if a.scratch_scan:
        kname = None
        for ln in asm.splitlines():
            m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
            if m:
                kname = m.group(1)
            m = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", ln)
            if m and int(m.group(1)) > 0:
                print(f"{int(m.group(1)):6d} bytes  {kname}")
        return
    name, body = kernel_text(asm, a.kernel, a.variant)
    loops, blocks = census(body)
    meta = {}
    for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size"):
        m = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", "\n".join(body))
        if m:
            meta[key] = int(m.group(1))
    spills = sum(1 for ln in body if "scratch_" in ln and "Spill" in ln or "Reload" in ln)
    # every access to the private segment, spill or not: an array the compiler could not keep in registers (a pointer
    # selected at run time, an index it could not unroll) shows up here and nowhere else
    scratch = sum(1 for ln in body if re.match(r"\s*scratch_(load|store)", ln))
    lines = [f"kernel {name}", f"registers {meta}  spill/reload instructions {spills}  scratch loads+stores {scratch}", ""]
    lines.append(f"{'whole kernel':28s} {fmt(loops[('kernel', 0)])}")
    for (hdr, depth), c in sorted(((k, v) for k, v in loops.items() if k[0] != "kernel"), key=lambda kv: (kv[0][1], kv[0][0])):
        lines.append(f"loop {hdr:14s} depth {depth}   {fmt(c)}")
    if a.blocks:
        lines.append("")
        for label, note, c in blocks:
            if sum(c.values()):
                lines.append(f"  {label:12s} {fmt(c)}   {note[:70]}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if a.out:
        Path(a.out).write_text(text)


if __name__ == "__main__":
    main()
