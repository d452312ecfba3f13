#!/usr/bin/env python3
"""How much of a kernel's time at 1080p is a shortage of waves?  Renders K copies of the same frame concurrently on K
streams (K contexts) and compares with K back-to-back renders: if K concurrent frames finish much faster than K x one
frame, the single frame does not fill the chip (too few waves per SIMD to pair instructions and hide latency)."""
import sys, time
import torch
sys.path.insert(0, ".")
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node
from godot_atmosphere_shader_amd import scene as S

wl = sys.argv[1] if len(sys.argv) > 1 else "clouds_high_rm"
pose = sys.argv[2] if len(sys.argv) > 2 else "P_space"
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
tex, params = demo_textures(), demo_params()
cam = S.Camera.from_pose(w, h, pose)
depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
for K in (1, 2, 4):
    nodes = [make_node(wl, tex, params) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    outs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(K)]
    frames = [n.prepare_frame(cam) for n in nodes]
    def run(reps):
        for _ in range(reps):
            for n, s, o, f in zip(nodes, streams, outs, frames):
                n.render_prepared(f, depth.data_ptr(), o.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
    run(5)
    t0 = time.perf_counter(); run(30); dt = (time.perf_counter() - t0) / 30
    print(f"{wl} {pose} {w}x{h}: {K} concurrent frame(s): {dt*1e3:.4f} ms per round = {dt*1e3/K:.4f} ms per frame")
    for n in nodes: n.close()
