#!/usr/bin/env python3
"""Where the float copies of the cloud-texture footprints (ATMO_F4 bit 0 = cubemap, bit 1 = shape volume) stop paying: kernel ms
by texture size, 1920x1080, pose P_space.  gpurun -- 'python tools/texsize_probe.py'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures  # noqa: E402

params = demo_params()
POSE = os.environ.get("POSE", "P_space")
cases = [(256, 64), (512, 64), (1024, 64), (256, 96), (256, 128), (128, 32)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for cube_n, shape_n in cases:
    tex = demo_textures(cube_n, shape_n)
    for wl in ("clouds_high", "clouds_high_rm"):
        best = {}
        for rnd in range(2):
            for f4 in ("0", "1", "2", "3"):
                os.environ["ATMO_F4"] = f4
                r = bench.run_workload(torch, S, wl, 1920, 1080, POSE, 60, 8, tex, params, 0, with_frame_stats=False)
                best[f4] = min(best.get(f4, 9.0), r["kernel_avg_ms"])
        print(f"{POSE} cube {cube_n:4d} shape {shape_n:3d} {wl:15s} " + "  ".join(f"F4={k}: {v:.4f}" for k, v in best.items()), flush=True)
os.environ.pop("ATMO_F4", None)
