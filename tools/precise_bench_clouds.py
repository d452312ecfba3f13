import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from godot_atmosphere_shader_amd import scene as S
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures
tex, params = demo_textures(), demo_params()
for wl, w, h in (("clouds_high", 1920, 1080), ("clouds_high_rm", 1920, 1080), ("clouds_high_rm", 3840, 2160)):
    for rnd in range(2):
        for pa in (False, True):
            r = bench.run_workload(torch, S, wl, w, h, "P_space", 60, 8, tex, params, 0, with_frame_stats=False, node_extra=dict(precise_atmosphere=pa))
            print(wl, w, "reference-order atmosphere" if pa else "default", round(r["kernel_avg_ms"], 4), "ms", r.get("kernel"), flush=True)
