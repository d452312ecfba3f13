#!/usr/bin/env python3
"""Round 6, late: what a GEOMETRIC tile order would give the cloudless kernels under a moving camera.  Their cost map is geometry -- a tile either holds rays that hit the
atmosphere shell or it does not -- so an order can be computed from the camera alone, with no lag: hit tiles first (row-major), all-miss tiles last.  Here the lists are made
on the host from a render of every pose (which tiles shade), and a 64-pose sequence is drawn back to back: row-major (feedback off), the library's feedback path, and the
tile-list draw with each pose's own hit-first list.
    gpurun -- 'python tools/probes/geo_order_probe.py'"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures, make_node  # noqa: E402


def main():
    tex = demo_textures()
    w, h = 1920, 1080
    device = torch.device("cuda", 0)
    n = 64
    for wl in (os.environ.get("PROBE_WORKLOADS", "direct32x8,lut32,shipped8").split(",")):
        config_name = bench.WORKLOADS[wl][0]
        n = 64 if "cloud" not in wl else 32
        for motion in (("orbit", 0.0), ("orbit", 1.0), ("pan", 1.0), ("pan", 3.0)):
            cams = bench.motion_cameras(S, w, h, motion, n)
            depths = [bench.depth_ground_sphere_torch(torch, S, c, device) for c in cams]
            out = torch.empty((h, w, 4), dtype=torch.float32, device=device)
            stream = torch.cuda.current_stream().cuda_stream
            res = {}
            for arm in ("row-major", "feedback", "hit tiles first"):
                node = make_node(config_name, tex) if arm == "feedback" else make_node(config_name, tex, tile_feedback=0)
                frames = [node.prepare_frame(c) for c in cams]
                lists = None
                if arm == "hit tiles first":
                    lists = []
                    cost, tw, th = node.measure_tile_costs(cams[0], depths[0])
                    ty, tx = cost.shape
                    for k in range(n):
                        node.render_prepared(frames[k], depths[k].data_ptr(), out.data_ptr(), stream)
                        torch.cuda.synchronize()
                        shaded = (out.abs().sum(dim=-1) > 0)
                        pad = torch.zeros((ty * th, tx * tw), dtype=torch.bool, device=device)
                        pad[:h, :w] = shaded
                        hit = pad.reshape(ty, th, tx, tw).any(dim=3).any(dim=1).reshape(-1)
                        idx = torch.arange(ty * tx, device=device)
                        lists.append(torch.cat([idx[hit], idx[~hit]]).to(torch.int32).contiguous())

                def seq():
                    for k in range(n):
                        if lists is None:
                            node.render_prepared(frames[k], depths[k].data_ptr(), out.data_ptr(), stream)
                        else:
                            node.render_tiles_prepared(frames[k], depths[k].data_ptr(), out.data_ptr(), lists[k].data_ptr(), lists[k].numel(), stream)
                for _ in range(3):
                    seq()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    seq()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / n)
                res[arm] = best
                node.close()
            print(f"{wl:11s} {motion[0]} {motion[1]:g} deg/frame, {n} poses back to back, ms per frame: " + "   ".join(f"{k} {v:.4f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
