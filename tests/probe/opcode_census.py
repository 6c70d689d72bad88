"""Opcode census of one fused-kernel instantiation from its `hipcc -S` listing.

    python tests/probe/opcode_census.py [--unit 12] [--phase-prof] [--defs "-DX=1 ..."] [--out profiles/r05a_opcode_census.txt]

Compiles kernel unit `--unit` of diffusion_edf_amd/csrc/dedf_kernels.hip to gfx950 assembly (device only, the product's flags) and counts
the instructions of the tile loop by class.  With --phase-prof the unit is built with -DDEDF_PHASE_PROF: the DEDF_STAMP markers of
dedf_edge.h become `s_memtime` reads, which split the straight-line listing into the pipeline regions of edge_tile (the same regions
tests/phase_prof.py reports cycles for), so every class is attributed to a region.  The per-edge fallback front of the table-reading
kernel (the blocks a tile only runs when its lengths leave the radial table) is reported separately: it is in the listing, not in the
executed path.

Static counts of straight-line code = executed counts per tile here: inside the tile the only loops are the row copies of the prologue
and the exec-masked stores, everything else is unrolled.
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "diffusion_edf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-slp-vectorize", "--cuda-device-only", "-S"]

# (class, regex on the mnemonic); first match wins
CLASSES = [
    ("mfma", r"^v_mfma"),
    ("accvgpr_read", r"^v_accvgpr_read"),
    ("accvgpr_write", r"^v_accvgpr_write"),
    ("split: cvt_pk_f16", r"^v_cvt_pk(rtz)?_f16_f32|^v_cvt_f16_f32"),
    ("split/unsplit: fma_mix", r"^v_fma_mix"),
    ("dpp (fmac/mov/max/min ... _dpp)", r"_dpp$"),
    ("fma/fmac/mac f32", r"^v_(fma|fmac|mac|mad)_f32|^v_fmaak_f32|^v_fmamk_f32"),
    ("mul f32", r"^v_mul_f32"),
    ("add/sub f32", r"^v_(add|sub|subrev)_f32"),
    ("pk f32 (v_pk_*)", r"^v_pk_"),
    ("max/min/med f32", r"^v_(max|min|med3)_(f32|num_f32)|^v_max3|^v_min3"),
    ("transcendental (exp/log/rcp/rsq/sqrt/sin/cos)", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"),
    ("cvt other", r"^v_cvt_"),
    ("mov b32/b64", r"^v_mov_b(32|64)"),
    ("and_or (operand ties)", r"^v_and_or_b32"),
    ("cndmask", r"^v_cndmask"),
    ("cmp", r"^v_cmp|^v_cmpx"),
    ("int alu (add/sub/mul/shift/and/or/bfe/lshl_add ...)", r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|lshlrev|lshrrev|ashrrev|add3|lshl_add|lshl_or|or3|and_b|not|perm|alignbit|min_[iu]|max_[iu])"),
    ("readlane/readfirstlane/writelane", r"^v_read|^v_writelane"),
    ("other valu", r"^v_"),
    ("lds read", r"^ds_read|^ds_load"),
    ("lds write", r"^ds_write|^ds_store"),
    ("ds_bpermute/permute/swizzle", r"^ds_bpermute|^ds_permute|^ds_swizzle"),
    ("buffer/global load", r"^(buffer|global|flat)_load"),
    ("buffer/global store/atomic", r"^(buffer|global|flat)_(store|atomic)"),
    ("scratch", r"^scratch_"),
    ("s_waitcnt", r"^s_waitcnt"),
    ("s_nop", r"^s_nop"),
    ("s_load / s_buffer_load", r"^s_(buffer_)?load"),
    ("branch", r"^s_c?branch"),
    ("other salu", r"^s_"),
]
CLASSES = [(n, re.compile(r)) for n, r in CLASSES]
VALU_PREFIX = "v_"


def classify(mn: str) -> str:
    for n, r in CLASSES:
        if r.search(mn):
            return n
    return "other"


def is_valu(mn: str) -> bool:
    return mn.startswith("v_") and not mn.startswith("v_mfma")


def parse_kernel(path: str, want: str):
    """-> list of (line_no, label_or_None, mnemonic_or_None) of the first kernel whose mangled name contains `want`"""
    lines = open(path).read().splitlines()
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and want in ln.split(":")[0] and ":" in ln:
            start = i
            break
    if start is None:
        raise SystemExit(f"no kernel matching {want!r} in {path}")
    out = []
    for i in range(start + 1, len(lines)):
        ln = lines[i]
        if ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
            break
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if s.startswith(".LBB") and s.split()[0].endswith(":"):
                out.append((i + 1, s.split(":")[0], None))
            continue
        mn = s.split()[0]
        if mn.endswith(":"):
            continue
        out.append((i + 1, None, mn))
    return out


def blocks_of(items):
    """split into basic blocks: [(label, first_line, [mnemonics], ends_with_branch_to)]"""
    blocks, cur, lab, first = [], [], "entry", items[0][0] if items else 0
    for ln, label, mn in items:
        if label is not None:
            if cur:
                blocks.append((lab, first, cur))
            cur, lab, first = [], label, ln
        else:
            cur.append(mn)
    if cur:
        blocks.append((lab, first, cur))
    return blocks


def table(counter: collections.Counter, title: str, width: int = 60) -> list[str]:
    tot = sum(counter.values())
    valu = sum(v for k, v in counter.items() if k not in ("mfma",) and not k.startswith(("lds", "ds_", "buffer", "scratch", "s_", "branch", "other salu", "other")) or k == "other valu")
    rows = [f"{title}: {tot} instructions, {counter.get('mfma', 0)} MFMA, {valu} VALU ({valu / max(counter.get('mfma', 0), 1):.1f} per MFMA)"]
    for n, _ in CLASSES:
        if counter.get(n):
            rows.append(f"    {n:<{width}} {counter[n]:>6}")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", type=int, default=12)
    ap.add_argument("--kernel", default="k_edge", help="substring of the mangled kernel name")
    ap.add_argument("--phase-prof", action="store_true")
    ap.add_argument("--defs", default="")
    ap.add_argument("--asm", default="", help="use this listing instead of compiling")
    ap.add_argument("--out", default="")
    ap.add_argument("--min-block", type=int, default=150, help="basic blocks below this size are folded into 'small blocks'")
    a = ap.parse_args()
    asm = a.asm
    if not asm:
        asm = f"/tmp/census_k{a.unit}{'_prof' if a.phase_prof else ''}.s"
        cmd = [HIPCC] + FLAGS + [f"-DDEDF_KUNIT={a.unit}"] + (["-DDEDF_PHASE_PROF"] if a.phase_prof else []) + a.defs.split() + [os.path.join(CSRC, "dedf_kernels.hip"), "-o", asm]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    items = parse_kernel(asm, a.kernel)
    out = [f"# opcode census: unit {a.unit} ({a.kernel}), {'-DDEDF_PHASE_PROF ' if a.phase_prof else ''}{a.defs}".rstrip(), f"# listing: {len(items)} lines of code"]
    meta = open(asm).read()
    for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size"):
        m = re.search(re.escape(key) + r":\s*(\d+)", meta)
        if m:
            out.append(f"# {key[1:]} = {m.group(1)}")
    if a.phase_prof:
        # regions between consecutive s_memtime reads, in listing order (DEDF_STAMP(i): see dedf_edge.h)
        seg, segs = collections.Counter(), []
        for ln, label, mn in items:
            if mn is None:
                continue
            if mn == "s_memtime":
                segs.append(seg)
                seg = collections.Counter()
            else:
                seg[classify(mn)] += 1
        segs.append(seg)
        for i, s in enumerate(segs):
            out += table(s, f"segment {i} (code between s_memtime #{i} and #{i + 1})")
        tot = collections.Counter()
        for s in segs:
            tot.update(s)
        out += table(tot, "whole listing")
    else:
        blocks = blocks_of(items)
        small = collections.Counter()
        tot = collections.Counter()
        for lab, first, mns in blocks:
            c = collections.Counter(classify(m) for m in mns)
            tot.update(c)
            if len(mns) >= a.min_block:
                out += table(c, f"block {lab} @ line {first}")
            else:
                small.update(c)
        out += table(small, f"all blocks below {a.min_block} instructions")
        out += table(tot, "whole listing")
        # runs of consecutive MFMAs (no VALU between them): a lone in-order wave issues nothing while a run drains
        runs, run = collections.Counter(), 0
        for ln, label, mn in items:
            if mn is None:
                continue
            if mn.startswith("v_mfma"):
                run += 1
            elif is_valu(mn):
                if run:
                    runs[run] += 1
                run = 0
        if run:
            runs[run] += 1
        out.append("MFMA runs without a VALU instruction between them (length: count): " + ", ".join(f"{k}: {v}" for k, v in sorted(runs.items())))
    text = "\n".join(out) + "\n"
    if a.out:
        with open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
