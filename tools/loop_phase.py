#!/usr/bin/env python3
"""Where the view loop of the direct-light kernels sits in the instruction stream (round 6, profiles/round6/ab_loop_phase.txt): the draw is 8.5-11 % slower
unless the loop's first instruction -- the target of its backward branch -- lies 12 bytes into a 32-byte block.

    python tools/loop_phase.py [libatmo_hip.so] [kernel substring, default "atmo_render_kernelILi4ELi8ELi1E"]

Prints, per matching kernel, the address of the loop header relative to the kernel's start, its offset in the 32-byte block, and the loop's size."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FAST_PHASE = 12   # bytes into the 32-byte block


def device_code_object(lib, out):
    """The gfx950 code object inside a HIP shared library (the .hip_fatbin section is a clang offload bundle)."""
    fat = out + ".fatbin"
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={out}"],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def view_loops(lib, pattern="atmo_render_kernelILi4ELi8ELi1E"):
    """[(kernel symbol, header offset from the kernel's start, header address mod 32, loop bytes)] for the loop of each matching kernel that holds the
    seven-root cluster of the 8-step light march."""
    with tempfile.TemporaryDirectory(prefix="phase_") as tmp:
        co = device_code_object(lib, os.path.join(tmp, "dev.co"))
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", co], check=True, capture_output=True, text=True).stdout
        names = sorted({l.split()[-1] for l in syms.splitlines() if " FUNC " in l and pattern in l})
        rows = []
        for name in names:
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={name}", co], check=True, capture_output=True, text=True).stdout
            ins = [(int(m.group(2), 16), m.group(1)) for m in re.finditer(r"^\s+(\S[^\n]*?)\s+// ([0-9A-F]{12}):", dis, re.M)]
            if not ins:
                continue
            start = ins[0][0]
            for addr, text in ins:
                m = re.match(r"s_cbranch_scc[01] (\d+)", text)
                if not m or int(m.group(1)) < 32768:
                    continue
                target = addr + 4 + (int(m.group(1)) - 65536) * 4
                body = [t for a, t in ins if target <= a <= addr]
                if sum(t.startswith("v_sqrt_f32") for t in body) >= 7:
                    rows.append((name, target - start, target % 32, addr + 4 - target))
        return rows


GEO_TWIN = "atmo_render_kernelILi260ELi8ELi1E"   # <KF_LIGHT_DIRECT | KF_GEO, 8, 1>: the same loop behind the geometric order's lookup (ATMO_LOOP_PAD_GEO)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "godot_atmosphere_shader_amd", "libatmo_hip.so")
    for name, off, phase, size in (view_loops(lib, sys.argv[2]) if len(sys.argv) > 2 else view_loops(lib) + view_loops(lib, GEO_TWIN)):
        print(f"{name}: view loop at +0x{off:x}, {size} bytes, header {phase} bytes into its 32-byte block ({'the fast position' if phase == FAST_PHASE else 'a SLOW position'})")
