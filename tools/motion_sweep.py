#!/usr/bin/env python3
"""Tile-order feedback with a MOVING camera: rate with feedback on / off, camera still, orbiting, panning.

    gpurun -- 'python tools/motion_sweep.py [--workloads direct32x8,clouds_high_rm] [--periods 8,2,1] [--steps 192]'

One process, so every cell of the table is measured on the same box at the same clocks; the loops are bench.py's own
(time_workload: a new pose every step, all frames and depth buffers prepared first, host at most 2 frames ahead of the GPU).
"period" = every n-th draw records tile costs and starts a sort (ATMO_TILE_FEEDBACK_PERIOD, read by atmo_create).
Prints the table that is committed as profiles/round3/ab_tile_feedback_motion.txt."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="direct32x8,clouds_high_rm,clouds_high")
    ap.add_argument("--periods", default="8,2,1")
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--motions", default="static,orbit:0.1,orbit:1,orbit:5,pan:0.1,pan:1,pan:5")
    args = ap.parse_args()
    tex, params = demo_textures(), demo_params()
    motions = [("orbit", 0.0) if m == "static" else bench.parse_motion(m) for m in args.motions.split(",")]
    periods = [int(p) for p in args.periods.split(",")]
    print(f"# {args.width}x{args.height}, {args.steps} timed steps per cell, Mrays/s (kernel ms); gain = on / off - 1")
    print(f"# {'workload':16s} {'motion':10s} {'feedback off':>20s} " + " ".join(f"{'on, period ' + str(p):>28s}" for p in periods))
    for wl in args.workloads.split(","):
        for motion in motions:
            def cell(fb, period=None):
                if period is not None:
                    os.environ["ATMO_TILE_FEEDBACK_PERIOD"] = str(period)
                r = bench.run_workload(torch, S, wl, args.width, args.height, "P_space", args.steps, 8, tex, params, 0,
                                       with_frame_stats=False, motion=motion, node_extra=dict(tile_feedback=fb))
                st = r["feedback_stats"]
                return r["Mrays/s"], r["kernel_avg_ms"], st["ordered_draws"], st["sorts"]
            off = cell(0)
            ons = [cell(1, p) for p in periods]
            off2 = cell(0)  # bracket: the box's drift over the row
            base = 0.5 * (off[0] + off2[0])
            name = "static" if motion[1] == 0.0 else f"{motion[0]}:{motion[1]:g}"
            print(f"  {wl:16s} {name:10s} {base:9.0f} ({0.5 * (off[1] + off2[1]):.4f}) " +
                  " ".join(f"{r:9.0f} ({k:.4f}) {100.0 * (r / base - 1.0):+6.1f} % [{od} ordered, {so} sorts]" for r, k, od, so in ons), flush=True)
    os.environ.pop("ATMO_TILE_FEEDBACK_PERIOD", None)


if __name__ == "__main__":
    main()
