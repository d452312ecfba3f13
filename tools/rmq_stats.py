#!/usr/bin/env python3
"""How full the light batches of the lit-sample queue run (march_clouds_rm_queue), from a diagnostic build:

    tools/ab_build.sh rmqstats -DATMO_WAVE_TRACE=1 -DATMO_RMQ_STATS=1
    gpurun -- 'ATMO_HIP_LIB=$PWD/godot_atmosphere_shader_amd/libatmo_hip_rmqstats.so python tools/rmq_stats.py [workload[@lod0] W H pose]'
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "clouds_high_rm"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    pose = sys.argv[4] if len(sys.argv) > 4 else "P_space"
    config_name, _ = bench.WORKLOADS[wl.split("@")[0]]
    kw = dict(bench.node_kwargs(wl.split("@")[0]))
    if wl.endswith("@lod0"):   # default: the declared cubemap sampler; name@lod0: level 0 only
        kw["cubemap_lod"] = False
    node = make_node(config_name, demo_textures(), demo_params(), **kw)
    cam = S.Camera.from_pose(w, h, pose)
    depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
    out = None
    for _ in range(6):
        out = node.render(cam, depth, out=out)
        torch.cuda.synchronize()
    fn = node._lib.atmo_debug_wave_trace
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    max_waves = 4 * ((w + 15) // 16) * ((h + 7) // 8)
    buf = np.zeros((max_waves, 4), dtype=np.uint64)
    n = fn(node._ctx, buf.ctypes.data_as(C.c_void_p), max_waves)
    assert n == max_waves, (n, max_waves)
    st = buf.reshape(-1)[-64:].astype(np.float64)
    waves, lanes, calls, lit, offered, evals, ticks_b, ticks_all, wave_lanes = st[:9]
    print(f"{wl} {w}x{h} {pose}: kernel {node.kernel_name}")
    print(f"  waves that march {waves:.0f}, marching lanes per such wave {lanes / waves:.1f} of 64")
    print(f"  density evaluations of the march {evals:.0f}, lit samples {lit:.0f} = {100 * lit / evals:.1f} % of them")
    print(f"  light batches {calls:.0f} ({calls / waves:.1f} per wave); filled {100 * lit / offered:.1f} % of the marching lanes, "
          f"{100 * lit / wave_lanes:.1f} % of the wave's 64 lanes")
    print(f"  time inside phase B (lighting) {100 * ticks_b / ticks_all:.1f} % of the cloud march's wave-time")
    if st[40]:
        print(f"  declared sampler: {st[40]:.0f} coverage samples, lambda = 0 (level 0 alone) for {100 * st[41] / st[40]:.1f} % of them; "
              f"{st[42]:.0f} wave executions, every lane at lambda = 0 in {100 * st[43] / st[42]:.1f} % of them")
    if st[44]:
        print(f"  level-0 certificate: held for {100 * st[45] / st[44]:.1f} % of {st[44]:.0f} samples; for every lane in {100 * st[47] / st[46]:.1f} % of {st[46]:.0f} wave executions")
    names = ["entered", "inside the layer (coverage sample)", "past the coverage early-outs (shape sample)"]
    for ph, label in ((0, "march"), (1, "light taps")):
        for stage in range(3):
            lanes_s, waves_s = st[16 + 8 * ph + 2 * stage], st[16 + 8 * ph + 2 * stage + 1]
            if waves_s:
                print(f"  density evaluation, {label:10s} {names[stage]:45s} lanes {lanes_s:12.0f}  wave executions {waves_s:10.0f}  "
                      f"= {lanes_s / waves_s:5.1f} lanes per execution")


if __name__ == "__main__":
    main()
