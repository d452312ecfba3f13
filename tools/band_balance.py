#!/usr/bin/env python3
"""Do the shares of ONE 1920x1080 frame take equal time on N GPUs?  For each workload and world size: cut the frame into row bands by
measured cost (atmo_measure_tile_costs), by the analytic estimate (bench.cloud_row_cost) and by equal row counts, and -- round 4 -- deal
its 16-row tile strips to the ranks longest-processing-time-first (sharding.lpt_strips; each share drawn as ONE tile-list launch,
atmo_render_tiles, heaviest tile first); render every share alone (what one GPU of N would do), time it with HIP events, and print
max / mean of the share times (1.00 = perfectly balanced) and the slowest share's kernel ms = the frame time of the sharded draw.
    gpurun -- 'python tools/band_balance.py [lod0]'      (lod0: the cloud workloads with the level-0 cubemap sampler, as in round 3)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node  # noqa: E402
from godot_atmosphere_shader_amd.sharding import balanced_row_bands, band_rect, heavy_tiles, lpt_strips, row_bands  # noqa: E402


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


def tiles_ms(node, cam, depth, tiles, reps=30, n_heavy=0):
    if tiles.size == 0:
        return 0.0
    out = torch.empty((cam.height, cam.width, 4), dtype=torch.float32, device="cuda")
    t = torch.from_numpy(tiles.astype(np.int32)).cuda()
    frame = node.prepare_frame(cam)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(8):
        node.render_tiles_prepared(frame, depth.data_ptr(), out.data_ptr(), t.data_ptr(), t.numel(), stream, n_heavy=n_heavy)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        node.render_tiles_prepared(frame, depth.data_ptr(), out.data_ptr(), t.data_ptr(), t.numel(), stream, n_heavy=n_heavy)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lod0 = "lod0" in sys.argv[1:]
    split2 = "split2" in sys.argv[1:]   # the shares drawn with two lanes per ray (LOD-0 kernels only): half as long wavefronts, 13-29 % more work
    w, h = 1920, 1080
    tex, params = demo_textures(), demo_params()
    print(f"# {w}x{h}, band time max / mean over the bands (kernel ms of the slowest band)")
    if split2:
        print("# two lanes per ray (atmo_set_lane_split 2) in every draw of this table")
    print(f"# cubemap sampler of the cloud workloads: {'level 0 (atmo_set_sampler_lod 0)' if lod0 else 'as declared (linear-mipmap, implicit LOD: the default)'}")
    print(f"# {'workload':16s} {'pose':9s} {'N':>2s} {'measured':>18s} {'analytic':>18s} {'equal rows':>18s} {'LPT tile strips':>18s} {'+ heavy tiles split':>22s}   whole frame")
    for wl in (("clouds_high_rm",) if split2 else ("clouds_high_rm", "clouds_high", "direct32x8")):
        for pose in ("P_space", "P_limb"):
            config_name = bench.WORKLOADS[wl][0]
            cam = S.Camera.from_pose(w, h, pose)
            depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
            node = make_node(config_name, tex, params, **(dict(cubemap_lod=False) if lod0 else {}), **(dict(lane_split=2) if split2 else {}))
            for _ in range(3):
                rows = node.measure_row_costs(cam, depth)
            cost = node._last_tile_costs
            whole = band_ms(node, cam, depth, (0, h))
            analytic = bench.cloud_row_cost(np, S, cam, "clouds" in wl)
            for world in (2, 4, 8):
                cells = []
                for bands in (balanced_row_bands(rows, world), balanced_row_bands(analytic, world), row_bands(h, world)):
                    t = np.array([band_ms(node, cam, depth, b) for b in bands])
                    cells.append(f"{t.max() / t.mean():5.2f} ({t.max():.4f})")
                _, tiles = lpt_strips(cost, world)
                t = np.array([tiles_ms(node, cam, depth, tl) for tl in tiles])
                cells.append(f"{t.max() / t.mean():5.2f} ({t.max():.4f})")
                # round 5: the same shares with their heavy tiles on two lanes per ray (atmo_render_tiles_split; declared-sampler cloud kernels)
                nh = [heavy_tiles(cost.reshape(-1)[tl]) for tl in tiles]
                t = np.array([tiles_ms(node, cam, depth, tl, n_heavy=k) for tl, k in zip(tiles, nh)])
                cells.append(f"{t.max() / t.mean():5.2f} ({t.max():.4f}) h{max(nh)}")
                print(f"  {wl:16s} {pose:9s} {world:2d} " + " ".join(f"{c:>18s}" for c in cells[:4]) + f" {cells[4]:>22s}" + f"   {whole:.4f}", flush=True)
            node.close()


if __name__ == "__main__":
    main()
