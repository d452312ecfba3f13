#!/usr/bin/env python3
"""How much of a short render kernel is not shading?  Times the shipped8 configuration (8 view steps, baked LUT) at
1920x1080 for a camera that sees the planet (P_space: 60 % of the rays shade) and for one that looks away from it (every
ray leaves after the exact prologue and stores zeros): the second number is launch + wave start-up + prologue + the 33 MB
store stream; "+cleared" = atmo_set_target_cleared(ctx, 1), discarded fragments store nothing.  python tools/floor_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures, make_node  # noqa: E402


def timed(node, cam, depth, n=300):
    out = node.render(cam, depth)
    frame = node.prepare_frame(cam)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, float((out[..., 3] > 0).float().mean())


def main():
    w, h = 1920, 1080
    tex = demo_textures()
    for cfg, kw in (("no_clouds_8", {}), ("no_clouds_8", dict(target_cleared=True)), ("no_clouds_32_lut", {}), ("no_clouds_32_lut", dict(target_cleared=True)),
                    ("no_clouds_32x8_direct", {}), ("no_clouds_32x8_direct", dict(target_cleared=True))):
        node = make_node(cfg, tex, **kw)
        cfg = cfg + ("+cleared" if kw else "")
        for name, cam in (("P_space", S.Camera.from_pose(w, h, "P_space")),
                          ("away", S.Camera(w, h, eye=(0.0, 0.0, 1000.0), target=(0.0, 1000.0, 1000.0), far=4000.0))):
            depth = torch.from_numpy(S.depth_far(cam)).cuda()
            ms, shaded = timed(node, cam, depth)
            print(f"{cfg:32s} {name:8s} {ms * 1000:7.1f} us per frame   shaded fraction {shaded:.2f}")
        node.close()
    # the store stream alone
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        out.zero_()
    e0.record()
    for _ in range(300):
        out.zero_()
    e1.record()
    torch.cuda.synchronize()
    print(f"{'memset 33 MB':32s} {'':8s} {e0.elapsed_time(e1) / 300 * 1000:7.1f} us")


if __name__ == "__main__":
    main()
