#!/usr/bin/env python3
"""VGPRs / SGPRs / LDS / stack of the render kernels, read from the .amdhsa descriptors hipcc emits (hipcc -S of atmo_kernels.hip).

    python tools/kernel_resources.py [k.s]         without an argument: compiles the file to ISA first (~20 s)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "godot_atmosphere_shader_amd", "csrc", "atmo_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only"]


def main(argv):
    if argv:
        path = argv[0]
    else:
        path = os.path.join(tempfile.mkdtemp(prefix="kres_"), "k.s")
        subprocess.run(["hipcc"] + FLAGS + [SRC, "-o", path], check=True, stderr=subprocess.DEVNULL)
    text = open(path).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel _ZN4atmo18atmo_render_kernel(_s80)?ILi(\d+)ELi(\d+)ELi(\d+)EEEvNS_12RenderConstsE\n(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(5)
        g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        v, s, lds, priv = g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size")
        vr = (v + 7) // 8 * 8
        sg = (s + 6 + 15) // 16 * 16   # + VCC, FLAT_SCRATCH, XNACK_MASK; 16-granular
        waves = min(8, 512 // vr, 800 // (sg + 16))
        if lds:
            waves = min(waves, (160 * 1024 // lds) * 2 // 4)
        rows.append((int(m.group(2)), int(m.group(3)), int(m.group(4)), v, s, lds, priv, waves))
    for f, l, sp, v, s, lds, priv, waves in sorted(rows):
        print(f"atmo_render_kernel<{f}, {l}, {sp}>  VGPRs {v:3d}  SGPRs {s:3d}  LDS {lds:6d} B  stack {priv:4d} B  ~{waves} waves/SIMD")


if __name__ == "__main__":
    main(sys.argv[1:])
