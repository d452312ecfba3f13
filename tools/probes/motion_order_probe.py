#!/usr/bin/env python3
"""what does a moving camera cost the cloud kernels, and how much of it is the tile ORDER?  An orbit sequence (bench.motion_cameras,
a new pose every frame), every frame timed alone (HIP events) under four orders, heavy-tile split off (ATMO_HEAVY_SPLIT=0: the order alone):
  none      row-major (atmo_set_tile_feedback 0)
  perfect   heaviest first by the costs measured ON THIS VERY FRAME (an upper bound for any predictor: atmo_measure_tile_costs, then atmo_render_tiles)
  lag1      heaviest first by the costs measured on the PREVIOUS frame (what a one-frame-lag sort has, without any window)
  lag1+d    the same, every tile's cost replaced by the maximum over its 3 x 3 neighbourhood (a one-tile window)
and the mis-ranking of lag1: where the heaviest 5 % of this frame's tiles stood in the previous frame's order.
    gpurun -- 'python tools/probes/motion_order_probe.py [workload] [deg_per_frame] [W H]'"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ATMO_HEAVY_SPLIT", "0")
import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures, make_node  # noqa: E402


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def dilate(cost):
    p = np.pad(cost, 1, mode="edge")
    out = cost.copy()
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, p[dy:dy + cost.shape[0], dx:dx + cost.shape[1]])
    return out


SHIFT_ARMS = bool(os.environ.get("PROBE_SHIFT_ARMS"))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "clouds_high_rm"
    deg = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
    kind = sys.argv[5] if len(sys.argv) > 5 else "orbit"
    config_name = bench.WORKLOADS[wl][0]
    tex = demo_textures()
    n = 24
    cams = bench.motion_cameras(S, w, h, (kind, deg), n)
    device = torch.device("cuda", 0)
    depths = [bench.depth_ground_sphere_torch(torch, S, c, device) for c in cams]
    node = make_node(config_name, tex, tile_feedback=0)
    out = torch.empty((h, w, 4), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    frames = [node.prepare_frame(c) for c in cams]
    costs = []
    for k in range(n):
        for _ in range(2):
            cost, tw, th = node.measure_tile_costs(cams[k], depths[k])
        costs.append(cost.astype(np.int64))
    for _ in range(30):   # clocks
        node.render_prepared(frames[0], depths[0].data_ptr(), out.data_ptr(), stream)
    rows, shift_rows = [], []
    for k in range(1, n):
        def order_of(c):
            flat = c.reshape(-1)
            return torch.from_numpy(np.argsort(-flat, kind="stable").astype(np.int32)).cuda()
        o_perf, o_lag, o_lagd = order_of(costs[k]), order_of(costs[k - 1]), order_of(dilate(costs[k - 1]))
        t_none = timed(lambda: node.render_prepared(frames[k], depths[k].data_ptr(), out.data_ptr(), stream))
        t = {}
        arms = [("perfect", o_perf), ("lag1", o_lag), ("lag1+d", o_lagd)]
        if SHIFT_ARMS and k >= 3:
            # round 6 (later): a THREE-frame-old cost map (the side-stream sort's real lag), as it is, under a window as wide as the planet centre's motion
            # over those frames, and TRANSLATED by that motion (the centre projected through both cameras; whole tiles) with and without a one-tile window
            def centre_px(cam):
                v = cam.projection @ (cam.view @ np.array([0.0, 0.0, 0.0, 1.0]))
                return np.array([(v[0] / v[3] * 0.5 + 0.5) * w, (1.0 - (v[1] / v[3] * 0.5 + 0.5)) * h])
            d = centre_px(cams[k]) - centre_px(cams[k - 3])
            tx_, ty_ = int(round(d[0] / tw)), int(round(d[1] / th))
            old = costs[k - 3]
            def window(c, rx, ry):
                pd = np.pad(c, ((ry, ry), (rx, rx)), mode="edge")
                o = c.copy()
                for dy in range(2 * ry + 1):
                    for dx in range(2 * rx + 1):
                        o = np.maximum(o, pd[dy:dy + c.shape[0], dx:dx + c.shape[1]])
                return o
            shifted = np.zeros_like(old)
            H_, W_ = old.shape
            ys, xs = np.mgrid[0:H_, 0:W_]
            sy, sx = ys - ty_, xs - tx_
            ok = (sy >= 0) & (sy < H_) & (sx >= 0) & (sx < W_)
            shifted[ok] = old[sy[ok], sx[ok]]
            arms += [("lag3", order_of(old)), ("lag3+w", order_of(window(old, abs(tx_) + 1, abs(ty_) + 1))), ("lag3 shifted", order_of(shifted)),
                     ("lag3 shifted+d", order_of(dilate(shifted)))]
        for name, o in arms:
            t[name] = timed(lambda: node.render_tiles_prepared(frames[k], depths[k].data_ptr(), out.data_ptr(), o.data_ptr(), o.numel(), stream))
        # mis-ranking: rank (in the previous frame's order) of this frame's heaviest 5 %
        flat_k, flat_p = costs[k].reshape(-1), costs[k - 1].reshape(-1)
        top = np.argsort(-flat_k)[: max(1, flat_k.size // 20)]
        rank_prev = np.empty(flat_p.size, dtype=np.int64)
        rank_prev[np.argsort(-flat_p, kind="stable")] = np.arange(flat_p.size)
        late = rank_prev[top] / flat_p.size
        rows.append((t_none, t["perfect"], t["lag1"], t["lag1+d"], float(np.median(late)), float(np.percentile(late, 95)), float(late.max()),
                     float(np.corrcoef(flat_k, flat_p)[0, 1])))
        if SHIFT_ARMS and k >= 3:
            shift_rows.append((t["lag3"], t["lag3+w"], t["lag3 shifted"], t["lag3 shifted+d"]))
    r = np.array(rows)
    node.close()
    # the library's own path on the SAME frames: a context with the tile-order feedback on, the sequence drawn frame by frame (forwards, then again: the
    # second pass is timed; every frame a new pose, so the order in force is the one the in-stream / side-stream machinery made from the frames before)
    lib_ms = []
    node = make_node(config_name, tex)
    for p in range(2):
        lib_ms = []
        for k in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            node.render_prepared(frames[k], depths[k].data_ptr(), out.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            lib_ms.append(e0.elapsed_time(e1))
    stats = node.feedback_stats()
    node.close()
    lib = float(np.median(lib_ms[1:]))
    print(f"{wl} {w}x{h} {kind} {deg:g} deg/frame, {n - 1} frames, kernel ms per frame (median over the frames; each frame best of 3), heavy-tile split off:")
    print(f"   row-major {np.median(r[:, 0]):.4f}   perfect order {np.median(r[:, 1]):.4f}   previous frame's order {np.median(r[:, 2]):.4f}   ... with a one-tile window {np.median(r[:, 3]):.4f}")
    if shift_rows:
        q = np.median(np.array(shift_rows), axis=0)
        print(f"   a 3-frame-old cost map: as it is {q[0]:.4f}   under a window as wide as the planet centre's motion {q[1]:.4f}   TRANSLATED by that motion {q[2]:.4f}   ... and a one-tile window {q[3]:.4f}")
    print(f"   the library's own path (feedback on, frame by frame, incl. its in-stream sort kernels) {lib:.4f}   {stats}")
    print(f"   this frame's heaviest 5 % of the tiles in the previous frame's order: median position {np.median(r[:, 4]) * 100:.1f} % of the list, 95th percentile {np.median(r[:, 5]) * 100:.1f} %, "
          f"last one {np.median(r[:, 6]) * 100:.1f} %; correlation of consecutive cost maps {np.median(r[:, 7]):.3f}")


if __name__ == "__main__":
    main()
