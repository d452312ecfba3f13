#!/usr/bin/env python3
"""what a per-tile choice between the in-place raymarched light (atmo_render_kernel<51, 0, 1>, the shipped form) and the lit-sample
queue (the -DATMO_RM_INPLACE=0 build's kernel of the same name) could buy: every tile's cost (its longest wavefront, shader cycles:
atmo_measure_tile_costs) under both kernels on the same frame -- sum of the per-tile minima against either sum.  Two processes (the library is chosen
at load time): this script starts itself once per library.     gpurun -- 'python tools/probes/rm_tile_choice_probe.py'"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [("P_space", 1920, 1080), ("P_space", 3840, 2160), ("P_limb", 1920, 1080), ("P_space", 1280, 720)]


def child(out_path):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from godot_atmosphere_shader_amd import scene as S
    from godot_atmosphere_shader_amd.demo import demo_textures, make_node
    tex = demo_textures()
    res = {}
    for pose, w, h in CASES:
        cam = S.Camera.from_pose(w, h, pose)
        depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
        node = make_node("clouds_high_rm", tex, tile_feedback=0)
        for _ in range(4):
            cost, tw, th = node.measure_tile_costs(cam, depth)
        res[f"{pose}_{w}x{h}"] = dict(cost=cost.astype(np.int64).tolist(), tw=tw, th=th, kernel=node.kernel_name)
        node.close()
    json.dump(res, open(out_path, "w"))


def main():
    import numpy as np
    outs = {}
    for name, lib in (("inplace", None), ("queue", os.path.join(ROOT, "godot_atmosphere_shader_amd", "libatmo_hip_rmq.so"))):
        env = dict(os.environ)
        if lib:
            env["ATMO_HIP_LIB"] = lib
        path = f"/tmp/rm_tile_{name}.json"
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], check=True, env=env)
        outs[name] = json.load(open(path))
    for key in outs["inplace"]:
        a = np.array(outs["inplace"][key]["cost"], dtype=np.float64).reshape(-1)
        b = np.array(outs["queue"][key]["cost"], dtype=np.float64).reshape(-1)
        m = np.minimum(a, b)
        busy = (a > 0.02 * a.max())
        print(f"{key}: tiles {a.size} ({int(busy.sum())} above 2 % of the heaviest); sum of tile costs  in place {a.sum():.3e}  queue {b.sum():.3e} ({(b.sum() / a.sum() - 1) * 100:+.1f} %)  "
              f"per-tile minimum {m.sum():.3e} ({(m.sum() / a.sum() - 1) * 100:+.1f} % against in place); queue cheaper in {int((b < a)[busy].sum())} busy tiles, by > 10 % in {int((b < 0.9 * a)[busy].sum())}; "
              f"heaviest tile: in place {a.max():.3e}, queue {b.max():.3e}")
        # where the queue wins: by the in-place cost's decile
        order = np.argsort(a)
        dec = np.array_split(order[a[order] > 0.02 * a.max()], 10)
        print("   by decile of the in-place tile cost (light -> heavy): queue / in place = " + " ".join(f"{b[d].sum() / a[d].sum():.2f}" for d in dec if d.size))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
