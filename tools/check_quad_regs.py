#!/usr/bin/env python3
"""Static check behind the whole-quad exchange of the declared-sampler kernels (atmo_kernels.hip, QuadRegs).

The exchange blocks run a few instructions in whole-quad mode (s_wqm_b64 exec, exec): lanes the compiler believes inactive WRITE the
blocks' destination registers.  That is only safe if those registers hold nothing else, for any lane, anywhere in the kernel -- which
the source arranges by making them read-write operands whose live range spans the kernel.  This script verifies it in the ISA hipcc
emits: in every kernel that contains an exchange block, the VGPRs written inside the blocks are written by NO instruction outside them.

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
        results[name] = (blocks, sorted(private), bad)
    return results


def main(argv):
    out = os.path.join(tempfile.mkdtemp(prefix="quadregs_"), "k.s")
    subprocess.run(["hipcc"] + HIPCC_FLAGS + [SRC, "-o", out] + argv, check=True, stderr=subprocess.DEVNULL)
    res = check(open(out).read())
    names = subprocess.run(["c++filt"] + list(res), capture_output=True, text=True).stdout.split("\n") if res else []
    ok = bool(res)
    for (name, (blocks, private, bad)), nice in zip(res.items(), names):
        nice = nice.replace("void atmo::", "").replace("(atmo::RenderConsts)", "")
        print(f"{nice:44s} {blocks} exchange blocks, {len(private)} private VGPRs: " + ("ok" if not bad else f"{len(bad)} OUTSIDE WRITES, e.g. {bad[0]}"))
        ok = ok and not bad and blocks > 0
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
