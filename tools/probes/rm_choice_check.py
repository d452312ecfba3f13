#!/usr/bin/env python3
"""The per-tile-class choice of round 6 (heavier tiles on the lit-sample-queue kernel beside the in-place kernel) must not change a bit: clouds_high_rm frames
drawn repeatedly (so that the tile order and with it the choice are in force) under ATMO_RM_TILE_CHOICE=1 and =0.  One process per setting."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    from godot_atmosphere_shader_amd import scene as S
    from godot_atmosphere_shader_amd.demo import demo_textures, make_node
    tex = demo_textures()
    out = {}
    for pose, (w, h) in (("P_space", (1920, 1080)), ("P_limb", (1280, 720)), ("P_clouds", (960, 540)), ("P_space", (3840, 2160))):
        cam = S.Camera.from_pose(w, h, pose)
        depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
        node = make_node("clouds_high_rm", tex)
        for _ in range(24):
            img = node.render(cam, depth)
            torch.cuda.synchronize()
        out[f"{pose}_{w}x{h}"] = img.cpu().numpy()
        print(pose, w, h, node.kernel_name, "split stats", node.split_stats(), flush=True)
        node.close()
    np.savez(sys.argv[2], **out)
else:
    for v in ("1", "0"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f"/tmp/rmc_{v}.npz"], check=True, env=dict(os.environ, ATMO_RM_TILE_CHOICE=v, ATMO_HEAVY_SPLIT="0"))
    a, b = np.load("/tmp/rmc_1.npz"), np.load("/tmp/rmc_0.npz")
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print(f"{len(a.files) - len(bad)} of {len(a.files)} frames bit-identical with and without the tile-class choice", bad)
    sys.exit(1 if bad else 0)
