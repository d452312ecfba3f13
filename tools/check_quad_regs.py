#!/usr/bin/env python3
"""Static check behind the whole-quad exchange of the declared-sampler kernels (atmo_kernels.hip, QuadRegs).

The exchange blocks run a few instructions in whole-quad mode (s_wqm_b64 exec, exec): lanes the compiler believes inactive WRITE the
blocks' destination registers.  That is only safe if those registers hold nothing else, for any lane, anywhere in the kernel -- which
the source arranges by making them read-write operands whose live range spans the kernel.  This script verifies it in the ISA hipcc
emits: in every kernel that contains an exchange block, the VGPRs written inside the blocks are written by NO instruction outside them; and the VGPRs the blocks READ (the lane's march position,
which helper lanes read as well) are written inside the march's loop nest by nothing but the march's own position update (round 5).

It also verifies that no kernel of the file has a stack frame (ScratchSize 0, no scratch_load / scratch_store).

    python tools/check_quad_regs.py [-DFLAG ...]        exit code 0 = every kernel passes; prints one line per kernel
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "godot_atmosphere_shader_amd", "csrc", "atmo_kernels.hip")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only"]


def vgprs(tok: str):
    """VGPR numbers named by one operand token: v12 or v[12:15]."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def written(line: str):
    """VGPRs an instruction writes: the first operand of VALU / vector-memory-load / LDS-read instructions."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    op, rest = parts
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()  # compares write SGPRs / VCC; readlane writes an SGPR
    if not op.startswith(("v_", "buffer_load", "global_load", "flat_load", "ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "scratch_load",
                          "global_atomic", "buffer_atomic", "ds_add_rtn", "ds_max_rtn")):
        return set()
    return vgprs(rest.split(",")[0].strip())


def read_regs(line: str):
    """VGPRs an instruction reads: every operand after the first (the first as well for stores and DPP moves' sources are operands 2..)."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    toks = [t.strip().split()[0] for t in parts[1].split(",") if t.strip()]
    regs = set()
    for t in toks[1:]:
        regs |= vgprs(t.strip("|-"))
    return regs


_BB = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")


def check_inputs(body: str, private: set):
    """the helper lanes a whole-quad block re-enables READ the block's input VGPRs (the lane's own march position) -- lanes the
    compiler believes inactive.  That is only right if those registers hold the position for EVERY lane that marches, i.e. if inside the march
    loop nest they are written by nothing but the march's own position update (`v_add_f32 R, step, R`, executed by all marching lanes): a copy
    made inside a divergent region (a split live range: `v_mov_b32 R', R` under a narrowed EXEC, the block then reading R') would leave the helper
    lanes with a stale R'.  Returns (input registers, offending lines)."""
    lines = body.split("\n")
    # basic blocks: (first line, in a loop?)
    starts = [i for i, ln in enumerate(lines) if _BB.match(ln.strip())]
    def in_loop(i):  # the label line and the comment lines that continue it
        txt = lines[i]
        j = i + 1
        while j < len(lines) and lines[j].strip().startswith(";") and not _BB.match(lines[j].strip()):
            txt += lines[j]
            j += 1
        return "Loop" in txt
    blocks_at, inside, inputs = [], False, set()
    for i, raw in enumerate(lines):
        ln = raw.strip()
        if ln.startswith(";;#ASMSTART"):
            inside, cur, first = True, [], i
        elif ln.startswith(";;#ASMEND"):
            inside = False
            if any("s_wqm_b64" in c for c in cur):
                blocks_at.append(first)
                for c in cur:
                    inputs |= read_regs(c) - private
        elif inside:
            cur.append(ln)
    if not blocks_at:
        return set(), []
    # the march's loop nest: the run of basic blocks annotated as inside a loop around the first .. last exchange block
    bb_of = lambda line: max([k for k in range(len(starts)) if starts[k] <= line], default=0)
    lo, hi = bb_of(blocks_at[0]), bb_of(blocks_at[-1])
    while lo > 0 and in_loop(starts[lo - 1]):
        lo -= 1
    while hi + 1 < len(starts) and in_loop(starts[hi + 1]):
        hi += 1
    first_line = starts[lo]
    last_line = starts[hi + 1] if hi + 1 < len(starts) else len(lines)
    bad, inside, recent = [], False, []
    for i in range(first_line, last_line):
        ln = lines[i].strip()
        if ln.startswith(";;#ASMSTART"):
            inside = True
        elif ln.startswith(";;#ASMEND"):
            inside = False
        if inside or not ln or ln.startswith((";", ".")) or ln.endswith(":"):
            continue
        recent = (recent + [ln])[-16:]
        w = written(ln) & inputs
        if not w:
            continue
        op = ln.split(None, 1)[0]
        ok = op.startswith("v_add_f32") and w <= read_regs(ln)          # the position update reads what it writes ...
        if not ok and op.startswith("v_add_f32"):
            # ... or, with two lanes per ray (two steps per iteration: p = (p + d) + d), reads the first half of the update a few instructions up
            for prev in reversed(recent[:-1]):
                if written(prev) & read_regs(ln):
                    ok = prev.split(None, 1)[0].startswith("v_add_f32") and w <= read_regs(prev)
                    break
        if not ok:
            bad.append(ln)
    return inputs, bad


def check(asm_text: str):
    results = {}
    for m in re.finditer(r"^(_ZN4atmo\w+):[^\n]*\n(.*?)\n\s*s_endpgm", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "s_wqm_b64" not in body:
            continue
        inside, private, outside_lines = False, set(), []
        block, blocks = [], 0
        for raw in body.split("\n"):
            line = raw.strip()
            if line.startswith(";;#ASMSTART"):
                inside, block = True, []
                continue
            if line.startswith(";;#ASMEND"):
                inside = False
                if any("s_wqm_b64" in b for b in block):
                    blocks += 1
                    for b in block:
                        private |= written(b)
                else:
                    outside_lines.extend(block)
                continue
            if not line or line.startswith((";", ".")) or line.endswith(":"):
                continue
            (block if inside else outside_lines).append(line)
        bad = [ln for ln in outside_lines if written(ln) & private]
        inputs, bad_inputs = check_inputs(body, private)
        results[name] = (blocks, sorted(private), bad, sorted(inputs), bad_inputs)
    return results


def main(argv):
    out = os.path.join(tempfile.mkdtemp(prefix="quadregs_"), "k.s")
    subprocess.run(["hipcc"] + HIPCC_FLAGS + [SRC, "-o", out] + argv, check=True, stderr=subprocess.DEVNULL)
    res = check(open(out).read())
    names = subprocess.run(["c++filt"] + list(res), capture_output=True, text=True).stdout.split("\n") if res else []
    ok = bool(res)
    for (name, (blocks, private, bad, inputs, bad_inputs)), nice in zip(res.items(), names):
        nice = nice.replace("void atmo::", "").replace("(atmo::RenderConsts)", "")
        print(f"{nice:44s} {blocks} exchange blocks, {len(private)} private VGPRs: " + ("ok" if not bad else f"{len(bad)} OUTSIDE WRITES, e.g. {bad[0]}")
              + f"; {len(inputs)} input VGPRs {['v%d' % r for r in inputs]}: "
              + ("only the position update writes them in the march" if not bad_inputs else f"{len(bad_inputs)} OTHER WRITES IN THE MARCH, e.g. {bad_inputs[0]}"))
        ok = ok and not bad and blocks > 0 and not bad_inputs and 3 <= len(inputs) <= 7   # the position, and the coverage rotation where it lives in VGPRs
    if not res:
        print("no kernel contains an exchange block")
    # and, for every kernel of the file: no stack.  A frame object -- even a dead spill slot of a rematerialised kernel-argument tuple, with no
    # scratch instruction left in the ISA -- makes the kernel descriptor ask for a private segment (round 4: seven kernels, 36 bytes).
    text = open(out).read()
    scratch = re.findall(r"^; ScratchSize: (\d+)", text, re.M)
    spilled = [m for m in re.finditer(r"^\s*(scratch_(?:load|store)\w*)", text, re.M)]
    nonzero = [int(x) for x in scratch if int(x)]
    print(f"{len(scratch)} kernels, {len(nonzero)} with a stack frame" + (f" (bytes: {nonzero})" if nonzero else "") + f", {len(spilled)} scratch instructions")
    ok = ok and not nonzero and not spilled and len(scratch) > 0
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
