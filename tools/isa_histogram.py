#!/usr/bin/env python3
"""Static histogram of the VALU instruction classes in the loops of the shipped gfx950 kernels, priced with the issue
costs measured by tools/valu_issue.hip (profiles/round2/valu_issue*_mi355x.jsonl).

    python tools/isa_histogram.py [--kernel SUBSTR] [--json OUT] [--flags "-DX=1 ..."]

Classes (cycles per wave64 instruction on one SIMD, shader clock; `cdna_hip_programming.md` quotes 2 for v_fma_f32):
    fast   2.2   v_fma/v_fmac/v_mul/v_add/v_sub _f32 with VGPR, inline-constant or literal operands (neg/abs/clamp/omod
                 are free), v_mov_b32, v_and/or/xor_b32, v_lshrrev/ashrrev, v_add/sub_u32
    slow   4.1   every VALU op with an SGPR operand, v_max/min/med3/max3/min3, v_cvt_*, v_floor/fract/trunc/rndne,
                 v_cmp*, v_cndmask, v_lshlrev, v_mad_u32_u24, v_mul_u32_u24, v_mul_lo/hi, v_add3/lshl_add/and_or/bfe/perm,
                 v_ldexp, v_cube*, DPP forms, v_pk_*_f32 (2 results)
    trans  8.1   v_exp/log/sqrt/rsq/rcp/sin/cos (f32 and f16); blocks both pipes and the next ~3 fast ops do not pair
Pricing of a block: 8.1 T + 2.0 per fast op poisoned by a preceding trans (<= 3 each) + max(4.1 S, 2.2 (S + F)):
slow ops run on one 16-lane pipe while fast ops of other waves use the second one (measured: max:fma 1:1 -> 2.2 per
instruction, 3:1 -> 3.2).  "spec" pricing is the guide's: 2 cycles for every non-transcendental op, 8 for transcendentals.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "godot_atmosphere_shader_amd", "csrc")

C_FAST, C_SLOW, C_TRANS = 2.2, 4.1, 8.1
POISON_OPS, POISON_EXTRA = 3, 2.0

TRANS = re.compile(r"^v_(exp|log|sqrt|rsq|rcp|sin|cos)_(f32|f16|legacy_f32)")
FAST_OPS = re.compile(
    r"^v_(fma_f32|fmac_f32|fmaak_f32|fmamk_f32|mul_f32|add_f32|sub_f32|subrev_f32|mov_b32|and_b32|or_b32|xor_b32|"
    r"lshrrev_b32|ashrrev_i32|add_u32|sub_u32|subrev_u32|not_b32|mul_legacy_f32)(_e32|_e64)?$")


def classify(op: str, operands: str) -> str:
    if not op.startswith("v_"):
        return "other"
    if op.startswith(("v_readfirstlane", "v_readlane", "v_writelane")):
        return "slow"
    if TRANS.match(op):
        return "trans"
    if "dpp" in op or "sdwa" in op or "quad_perm" in operands or "row_" in operands:
        return "slow"
    # SGPR / VCC / EXEC used as a DATA operand makes the op slow (v_cndmask's mask is part of its own cost)
    srcs = operands.split(",")[1:] if "," in operands else []
    uses_sgpr = any(re.match(r"^\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0)\|?\s*$", t) for t in srcs)
    if FAST_OPS.match(op) and not uses_sgpr:
        return "fast"
    return "slow"


def disassemble(flags):
    src = os.path.join(CSRC, "atmo_kernels.hip")
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S",
           "--cuda-device-only", src, "-o", out] + flags
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return names
    p = subprocess.run([tool] + names, capture_output=True, text=True)
    return p.stdout.split("\n")[:len(names)] if p.returncode == 0 else names


def parse_kernels(asm: str):
    kernels, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):  # not the first s_endpgm: kernels with early exits have several
            cur = None
            continue
        cur.append(line)
    return kernels


def blocks_of(lines):
    """Split into labelled basic blocks; returns [(label, [ (op, operands) ]) ] and the backward-branch targets."""
    blocks, cur, label = [], [], "entry"
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                blocks.append((label, cur))
                label, cur = m.group(1), []
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        parts = s.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    blocks.append((label, cur))
    return blocks


def find_loops(blocks):
    """Natural loops by label order: a branch to an earlier-or-same label closes a loop spanning those blocks."""
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    loops = []
    for i, (lab, ins) in enumerate(blocks):
        for op, operands in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = operands.strip()
                if tgt in index and index[tgt] <= i:
                    loops.append((index[tgt], i))
    # keep innermost-first ordering, unique
    return sorted(set(loops), key=lambda ab: (ab[1] - ab[0], ab[0]))


def histogram(ins):
    h = {"fast": 0, "slow": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0, "ops": {}}
    seq = []
    for op, operands in ins:
        if op.startswith("v_"):
            c = classify(op, operands)
            h[c] += 1
            seq.append(c)
            key = re.sub(r"_e(32|64)$", "", op) + (" [sgpr]" if c == "slow" and FAST_OPS.match(op) else "")
            h["ops"][key] = h["ops"].get(key, 0) + 1
        elif op.startswith("s_"):
            h["salu"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            h["vmem"] += 1
        elif op.startswith("ds_"):
            h["lds"] += 1
    # fast ops within POISON_OPS VALU instructions after a trans do not pair with another wave's
    poisoned, since = 0, 99
    for c in seq:
        if c == "trans":
            since = 0
        else:
            if c == "fast" and since < POISON_OPS:
                poisoned += 1
            since += 1
    h["poisoned_fast"] = poisoned
    return h


def price(h):
    valu = h["fast"] + h["slow"] + h["trans"]
    spec = 2.0 * (h["fast"] + h["slow"]) + 8.0 * h["trans"]
    measured = C_TRANS * h["trans"] + POISON_EXTRA * h["poisoned_fast"] + max(C_SLOW * h["slow"], C_FAST * (h["slow"] + h["fast"]))
    return {"valu": valu, "cycles_spec": spec, "cycles_measured_model": measured,
            "cycles_per_valu_spec": spec / valu if valu else 0.0, "cycles_per_valu_measured_model": measured / valu if valu else 0.0}


def analyse(asm: str, select: str | None):
    kernels = parse_kernels(asm)
    names = list(kernels)
    pretty = dict(zip(names, demangle(names)))
    out = {}
    for name, lines in kernels.items():
        pn = pretty[name]
        if select and select not in pn:
            continue
        blocks = blocks_of(lines)
        loops = find_loops(blocks)
        whole = histogram([i for _, ins in blocks for i in ins])
        entry = {"whole_kernel_static": dict({k: v for k, v in whole.items() if k != "ops"}, **price(whole)), "loops": []}
        for (a, b) in loops:
            ins = [i for _, bl in blocks[a:b + 1] for i in bl]
            h = histogram(ins)
            if h["fast"] + h["slow"] + h["trans"] < 8:
                continue
            entry["loops"].append(dict(blocks=f"{blocks[a][0]}..{blocks[b][0]}", n_blocks=b - a + 1,
                                       **{k: v for k, v in h.items() if k != "ops"}, **price(h),
                                       top_slow=sorted(((k, v) for k, v in h["ops"].items()
                                                        if classify(k.split()[0], "v0, s0" if "[sgpr]" in k else "v0, v0") != "fast"
                                                        and not TRANS.match(k)), key=lambda kv: -kv[1])[:12],
                                       trans_ops={k: v for k, v in h["ops"].items() if TRANS.match(k)}))
        out[pn] = entry
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=None, help="substring of the demangled kernel name, e.g. 'atmo_render_kernel<4, 8>'")
    ap.add_argument("--json", default=None)
    ap.add_argument("--flags", default="", help="extra hipcc flags (A/B builds)")
    ap.add_argument("--asm", default=None, help="use an existing .s file instead of compiling")
    args = ap.parse_args()
    asm = open(args.asm).read() if args.asm else disassemble(args.flags.split())
    res = analyse(asm, args.kernel)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)
    for k, e in res.items():
        w = e["whole_kernel_static"]
        print(f"{k}\n  static: {w['valu']} VALU (fast {w['fast']}, slow {w['slow']}, trans {w['trans']}), {w['salu']} SALU, {w['vmem']} VMEM, {w['lds']} LDS")
        for lp in e["loops"]:
            print(f"  loop {lp['blocks']:24s} blocks {lp['n_blocks']:2d}: VALU {lp['valu']:4d} = fast {lp['fast']:3d} + slow {lp['slow']:3d} + trans {lp['trans']:2d}"
                  f" | salu {lp['salu']:3d} vmem {lp['vmem']:2d} lds {lp['lds']:2d} | cycles spec {lp['cycles_spec']:6.0f} model {lp['cycles_measured_model']:6.0f}"
                  f" ({lp['cycles_per_valu_measured_model']:.2f}/inst)")
            print("      slow: " + ", ".join(f"{k} x{v}" for k, v in lp["top_slow"]))
            if lp["trans_ops"]:
                print("      trans: " + ", ".join(f"{k} x{v}" for k, v in lp["trans_ops"].items()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
