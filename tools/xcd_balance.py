#!/usr/bin/env python3
"""How evenly the heaviest-first tile order spreads the measured work over the 8 XCDs (workgroup i of a launch runs on XCD i mod 8):
per-XCD sums of the measured tile costs under (a) row-major order, (b) the class sort the kernels use (32 half-octave classes, row-major
inside a class), (c) the same with a per-class rotation chosen to balance the XCDs.   gpurun -- 'python tools/xcd_balance.py [workload W H pose]'"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node  # noqa: E402


def cost_class(c):
    c = np.asarray(c, dtype=np.int64)
    msb = np.where(c > 0, np.floor(np.log2(np.maximum(c, 1))).astype(np.int64), 0)
    half = np.where(msb > 0, (c >> np.maximum(msb - 1, 0)) & 1, 0)
    q = np.clip(msb * 2 + half - 16, 0, 31)
    return np.where(c == 0, 31, 31 - q)


def spread(order, cost):
    loads = np.array([cost[order[x::8]].sum() for x in range(8)], dtype=np.float64)
    return loads, (loads.max() / loads.mean() - 1.0) * 100.0


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "clouds_high_rm"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    pose = sys.argv[4] if len(sys.argv) > 4 else "P_space"
    config_name, _ = bench.WORKLOADS[wl]
    node = make_node(config_name, demo_textures(), demo_params(), **bench.node_kwargs(wl))
    cam = S.Camera.from_pose(w, h, pose)
    depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
    for _ in range(3):
        node.measure_row_costs(cam, depth)
    cost = node._last_tile_costs.reshape(-1).astype(np.int64)
    node.close()
    n = cost.size
    cls = cost_class(cost)
    rowmajor = np.arange(n)
    sorted_order = np.argsort(cls, kind="stable")  # classes 0 (heaviest) .. 31, row-major inside
    print(f"{wl} {w}x{h} {pose}: {n} tiles, cost sum {cost.sum()}, max {cost.max()}, classes used {len(np.unique(cls))}")
    for name, order in (("row-major", rowmajor), ("class sort", sorted_order)):
        loads, s = spread(order, cost)
        print(f"  {name:12s} XCD loads / mean: " + " ".join(f"{v / loads.mean():.3f}" for v in loads) + f"   max over mean {s:+.2f} %")
    # per-class rotation, greedy from the heaviest class
    out = np.empty(n, dtype=np.int64)
    loads = np.zeros(8)
    base = 0
    for c in range(32):
        idx = sorted_order[cls[sorted_order] == c]
        m = idx.size
        if m == 0:
            continue
        best = None
        for r in range(min(8, m)):
            pos = base + (np.arange(m) + r) % m
            add = np.bincount(pos % 8, weights=cost[idx], minlength=8)
            worst = (loads + add).max()
            if best is None or worst < best[0]:
                best = (worst, r, add)
        _, r, add = best
        out[base + (np.arange(m) + r) % m] = idx
        loads += add
        base += m
    l2, s2 = spread(out, cost)
    print(f"  {'rotated':12s} XCD loads / mean: " + " ".join(f"{v / l2.mean():.3f}" for v in l2) + f"   max over mean {s2:+.2f} %")


if __name__ == "__main__":
    main()
