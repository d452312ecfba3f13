#!/usr/bin/env python3
"""Round 6: how fine must the cost sort of the tile order be?  The library sorts tiles into 32 half-octave classes of the measured wave duration (row-major inside a
class); profiles/round6/xcd_order.txt showed a tile-list draw in EXACT cost order 2.9 % faster than the library's draw of clouds_high at 1920x1080.  Same path for all
arms here (atmo_render_tiles): the library's class function at 32 (half octaves), 64 (quarter octaves) and 128 classes, and the exact order.
    gpurun -- 'python tools/probes/order_granularity_probe.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures, make_node  # noqa: E402


def timed(fn, n, reps=5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return float(np.median(ts))


def classes(cost, per_octave):
    """tile_cost_class of atmo_kernels.hip generalised: per_octave classes per octave of the duration, from 2^8 cycles up; heaviest class first."""
    c = np.maximum(cost.astype(np.float64), 1.0)
    q = np.floor((np.log2(c) - 8.0) * per_octave).astype(np.int64)
    return -np.clip(q, 0, 16 * per_octave - 1)


def main():
    tex = demo_textures()
    for cfg in ("clouds_high", "clouds_high_rm", "no_clouds_32x8_direct", "no_clouds_32_lut"):
        for pose, (w, h) in (("P_space", (1920, 1080)), ("P_ground", (1920, 1080)), ("P_space", (3840, 2160))):
            cam = S.Camera.from_pose(w, h, pose)
            depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
            node = make_node(cfg, tex)
            for _ in range(3):
                cost, tw, th = node.measure_tile_costs(cam, depth)
            out = node.render(cam, depth)
            torch.cuda.synchronize()
            flat = np.asarray(cost).reshape(-1).astype(np.int64)
            frame = node.prepare_frame(cam)
            stream = torch.cuda.current_stream().cuda_stream
            n = 200 if "cloud" not in cfg.replace("no_clouds", "") else 50
            lists = {"32 classes": np.argsort(classes(flat, 2), kind="stable"), "64": np.argsort(classes(flat, 4), kind="stable"), "128": np.argsort(classes(flat, 8), kind="stable"),
                     "exact": np.argsort(-flat, kind="stable")}
            if os.environ.get("PROBE_WITHIN_CLASS"):   # second question: the order INSIDE a class (the library: row-major) -- Morton order, and 2-tile-high row pairs
                ty, tx = cost.shape
                yy, xx = np.divmod(np.arange(flat.size), tx)
                def part1by1(v):
                    v = v.astype(np.int64) & 0xFFFF
                    v = (v | (v << 8)) & 0x00FF00FF
                    v = (v | (v << 4)) & 0x0F0F0F0F
                    v = (v | (v << 2)) & 0x33333333
                    return (v | (v << 1)) & 0x55555555
                morton = part1by1(xx) | (part1by1(yy) << 1)
                pairs = (yy // 2) * (2 * tx) + xx * 2 + (yy & 1)          # two tile rows at a time, column by column
                cls = classes(flat, 4)
                lists = {"32 classes": lists["32 classes"], "64": lists["64"], "64, Morton inside": np.lexsort((morton, cls)), "64, row pairs inside": np.lexsort((pairs, cls)),
                         "64, columns inside": np.lexsort((xx * ty + yy, cls))}
            res = {}
            for rnd in range(2):   # two interleaved passes
                for name, lst in lists.items():
                    tiles = torch.from_numpy(lst.astype(np.int32)).cuda()
                    t = timed(lambda: node.render_tiles_prepared(frame, depth.data_ptr(), out.data_ptr(), tiles.data_ptr(), tiles.numel(), stream), n)
                    res.setdefault(name, []).append(t)
            t_lib = timed(lambda: node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream), n)
            node.close()
            base = min(res["32 classes"])
            print(f"{cfg:22s} {pose:8s} {w}x{h}: library draw {t_lib * 1e3:7.1f} us | tile lists: " + "   ".join(f"{k} {min(v) * 1e3:7.1f} ({(min(v) / base - 1) * 100:+.1f} %)" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
