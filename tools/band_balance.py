#!/usr/bin/env python3
"""Do row bands cut from MEASURED tile costs (atmo_measure_tile_costs) take equal time?  For each workload and world size: cut the
1920x1080 frame by measured cost, by the analytic estimate (bench.cloud_row_cost) and by equal row counts, render every band alone
(what one GPU of N would do), time it with HIP events, and print max / mean of the band times (1.00 = perfectly balanced; the
slowest band sets the frame time of a strong-scaling run).   gpurun -- 'python tools/band_balance.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node  # noqa: E402
from godot_atmosphere_shader_amd.sharding import balanced_row_bands, band_rect, row_bands  # noqa: E402


def band_ms(node, cam, depth, band, reps=30):
    rect = band_rect(cam.width, band)
    if band[1] <= band[0]:
        return 0.0
    out = torch.empty((band[1] - band[0], cam.width, 4), dtype=torch.float32, device="cuda")
    frame = node.prepare_frame(cam, rect=rect)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(8):
        node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    w, h = 1920, 1080
    tex, params = demo_textures(), demo_params()
    print(f"# {w}x{h}, band time max / mean over the bands (kernel ms of the slowest band)")
    print(f"# {'workload':16s} {'pose':9s} {'N':>2s} {'measured':>18s} {'analytic':>18s} {'equal rows':>18s}")
    for wl in ("clouds_high_rm", "clouds_high", "direct32x8"):
        for pose in ("P_space", "P_limb"):
            config_name = bench.WORKLOADS[wl][0]
            cam = S.Camera.from_pose(w, h, pose)
            depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
            node = make_node(config_name, tex, params)
            for _ in range(3):
                rows = node.measure_row_costs(cam, depth)
            analytic = bench.cloud_row_cost(np, S, cam, "clouds" in wl)
            for world in (2, 4, 8):
                cells = []
                for bands in (balanced_row_bands(rows, world), balanced_row_bands(analytic, world), row_bands(h, world)):
                    t = np.array([band_ms(node, cam, depth, b) for b in bands])
                    cells.append(f"{t.max() / t.mean():5.2f} ({t.max():.4f})")
                print(f"  {wl:16s} {pose:9s} {world:2d} " + " ".join(f"{c:>18s}" for c in cells), flush=True)
            node.close()


if __name__ == "__main__":
    main()
