#!/usr/bin/env python3
"""VERDICT r5 next #6, candidate 1: what is there to gain from not LAUNCHING the tiles no ray of which can hit the atmosphere?
An upper bound, measured with what exists: the full draw under atmo_set_target_cleared (sure-miss waves store nothing: ~25 VALU and gone) against a
tile-list draw (atmo_render_tiles) of exactly the tiles that hold a shaded pixel, heaviest first, for the three cloudless kernels and two cloud kernels.
    gpurun -- 'python tools/probes/hit_tiles_probe.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures, make_node  # noqa: E402


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    tex = demo_textures()
    for cfg in ("no_clouds_8", "no_clouds_32_lut", "no_clouds_32x8_direct", "clouds_high", "clouds_high_rm"):
        for pose, (w, h) in (("P_space", (1920, 1080)), ("P_space", (3840, 2160)), ("P_limb", (1920, 1080))):
            cam = S.Camera.from_pose(w, h, pose)
            depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
            node = make_node(cfg, tex, target_cleared=True)
            for _ in range(3):
                cost, tw, th = node.measure_tile_costs(cam, depth)
            out = node.render(cam, depth)
            torch.cuda.synchronize()
            ty, tx = cost.shape
            shaded = (out.abs().sum(dim=-1) > 0).cpu().numpy()
            pad = np.zeros((ty * th, tx * tw), dtype=bool)
            pad[:h, :w] = shaded
            tile_hit = pad.reshape(ty, th, tx, tw).any(axis=(1, 3))
            idx = np.flatnonzero(tile_hit.reshape(-1))
            idx = idx[np.argsort(-cost.reshape(-1)[idx], kind="stable")]     # heaviest first, as the feedback order has them
            tiles = torch.from_numpy(idx.astype(np.int32)).cuda()
            frame = node.prepare_frame(cam)
            stream = torch.cuda.current_stream().cuda_stream
            n = 300 if "cloud" not in cfg.replace("no_clouds", "") else 60
            t_full = timed(lambda: node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream), n)
            t_list = timed(lambda: node.render_tiles_prepared(frame, depth.data_ptr(), out.data_ptr(), tiles.data_ptr(), tiles.numel(), stream), n)
            node.close()
            print(f"{cfg:24s} {pose:8s} {w}x{h}: full draw (cleared target, feedback order) {t_full * 1e3:8.1f} us   only the {idx.size} of {ty * tx} tiles that shade "
                  f"{t_list * 1e3:8.1f} us   ({(t_list / t_full - 1) * 100:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
