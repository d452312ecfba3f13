#!/usr/bin/env python3
"""Prints the max |HIP - fp32 oracle| per config and pose (needs an MI355X). Used to watch the parity margin."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); import numpy as np, torch
from godot_atmosphere_shader_amd.demo import CONFIGS, demo_frame, demo_params, demo_textures, make_node
from godot_atmosphere_shader_amd import scene as S
from oracle.oracle import Oracle

o = Oracle("f32")
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 216)
tex, params = demo_textures(), demo_params()
worst = 0.0
for name in CONFIGS:
    row = []
    for pose in ["P_space", "P_ground", "P_limb", "P_clouds", "P_night"]:
        cam = S.Camera.from_pose(w, h, pose)
        depth = S.depth_ground_sphere(cam)
        node = make_node(name, tex, params)
        got = node.render(cam, torch.from_numpy(depth).cuda()).cpu().numpy()
        cfg = CONFIGS[name][1]
        lut = node.read_optical_depth() if not (cfg.get("lite") or cfg.get("light_steps")) else None
        node.close()
        want, _ = o.render(params, dict(tex, optical_depth=lut), CONFIGS[name][1], demo_frame(cam), depth, nthreads=os.cpu_count())
        e = float(np.abs(got - want).max())
        worst = max(worst, e)
        row.append(f"{pose}={e:.2e}")
    print(f"{name:24s}", " ".join(row))
print("worst", f"{worst:.3e}")
