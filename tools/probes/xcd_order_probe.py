#!/usr/bin/env python3
"""Round 6: does an XCD-aware tile order pay?  Workgroup b of a launch runs on XCD b % 8, each XCD has its own 4 MB L2, and under the cost-sorted order
neighbouring tiles -- which share cubemap / LUT footprints -- are dealt to eight different L2s, so every XCD streams the whole texture set once per frame
(fabric traffic 1.8x the algorithmic bytes on the 1080p cloud frames).  Measured with what exists, the tile-list draw (atmo_render_tiles):
  list A = all tiles, heaviest first (what the feedback order is);
  list B = the tiles in row-major order cut into 8 contiguous runs of equal total cost (screen bands), each run heaviest first, interleaved so that run k's
           tiles sit at list positions 8 i + k (XCD k), padded with out-of-grid indices (which shade nothing);
  list C = the same with the runs cut along 2 x 4 screen blocks instead of bands.
    gpurun -- 'python tools/probes/xcd_order_probe.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures, make_node  # noqa: E402


def timed(fn, n):
    for _ in range(10):
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


def interleave(runs, sentinel):
    m = max(len(r) for r in runs)
    out = np.full((m, 8), sentinel, dtype=np.int64)
    for k, r in enumerate(runs):
        out[:len(r), k] = r
    return out.reshape(-1)


def equal_cost_runs(seq, cost, parts=8):
    """seq: tile indices in a locality-preserving sequence; cut into `parts` contiguous runs of (nearly) equal total cost, each sorted heaviest first."""
    c = cost[seq].astype(np.float64) + 1.0
    cum = np.cumsum(c)
    cuts = np.searchsorted(cum, cum[-1] * np.arange(1, parts) / parts)
    runs = np.split(seq, cuts)
    return [r[np.argsort(-cost[r], kind="stable")] for r in runs]


def main():
    tex = demo_textures()
    for cfg in ("clouds_high", "clouds_high_rm", "no_clouds_32x8_direct", "no_clouds_8"):
        for pose, (w, h) in (("P_space", (1920, 1080)), ("P_space", (3840, 2160)), ("P_ground", (1920, 1080))):
            cam = S.Camera.from_pose(w, h, pose)
            depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
            node = make_node(cfg, tex)
            for _ in range(3):
                cost, tw, th = node.measure_tile_costs(cam, depth)
            out = node.render(cam, depth)
            torch.cuda.synchronize()
            ty, tx = cost.shape
            flat = np.asarray(cost).reshape(-1).astype(np.int64)
            n_tiles = flat.size
            sentinel = n_tiles + 7
            a = np.argsort(-flat, kind="stable")
            rowmajor = np.arange(n_tiles)
            b = interleave(equal_cost_runs(rowmajor, flat), sentinel)
            # 2 x 4 screen blocks: tiles ordered block by block (row-major inside a block), then cut by equal cost
            yy, xx = np.divmod(rowmajor, tx)
            blk = (yy * 2 // ty) * 4 + (xx * 4 // tx)
            c = interleave(equal_cost_runs(rowmajor[np.argsort(blk, kind="stable")], flat), sentinel)
            frame = node.prepare_frame(cam)
            stream = torch.cuda.current_stream().cuda_stream
            n = 200 if "cloud" not in cfg.replace("no_clouds", "") else 50
            t_full = timed(lambda: node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream), n)
            res = []
            for lst in (a, b, c):
                tiles = torch.from_numpy(lst.astype(np.int32)).cuda()
                res.append(timed(lambda: node.render_tiles_prepared(frame, depth.data_ptr(), out.data_ptr(), tiles.data_ptr(), tiles.numel(), stream), n))
            node.close()
            print(f"{cfg:22s} {pose:8s} {w}x{h}: library draw {t_full * 1e3:7.1f} us | tile lists: heaviest first {res[0] * 1e3:7.1f}   8 bands, one per XCD {res[1] * 1e3:7.1f} "
                  f"({(res[1] / res[0] - 1) * 100:+.1f} %)   8 blocks {res[2] * 1e3:7.1f} ({(res[2] / res[0] - 1) * 100:+.1f} %)   [{len(b) - n_tiles} / {len(c) - n_tiles} padding entries of {n_tiles}]", flush=True)


if __name__ == "__main__":
    main()
