"""Heavy tiles on two lanes per ray: ms per draw and how many tiles were split, per threshold (ATMO_HEAVY_SPLIT_RATIO) and workload.
    python tools/heavy_split_probe.py [width height]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, demo_textures, make_node  # noqa: E402

tex, params = demo_textures(), demo_params()
SIZES = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(1280, 720), (1920, 1080), (3840, 2160)]
MODES = os.environ.get("PROBE_MODES", "off,all,t1.3:0.25,t1.3:0.4,t2:0.25,off").split(",")
for wl, (w, h) in [(a, b) for a in os.environ.get("PROBE_WORKLOADS", "clouds_high_rm").split(",") for b in SIZES]:
    for pose in os.environ.get("PROBE_POSES", "P_space,P_limb,P_ground,P_clouds,P_night").split(","):
        cam = S.Camera.from_pose(w, h, pose)
        depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
        out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        for mode in MODES:
            for k in ("ATMO_HEAVY_SPLIT", "ATMO_HEAVY_SPLIT_TRIGGER", "ATMO_HEAVY_SPLIT_RATIO"):
                os.environ.pop(k, None)
            if mode != "default":
                os.environ["ATMO_HEAVY_SPLIT"] = "0" if mode in ("off", "all") else "1"
            kw = dict(lane_split=2) if mode == "all" else {}
            if ":" in mode:
                os.environ["ATMO_HEAVY_SPLIT_TRIGGER"], os.environ["ATMO_HEAVY_SPLIT_RATIO"] = mode[1:].split(":")
            node = make_node(wl, tex, params, **kw)
            frame = node.prepare_frame(cam)
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(40):
                node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), s)
                torch.cuda.synchronize()
            best = 1e9
            K = 60
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(K):
                    node.render_prepared(frame, depth.data_ptr(), out.data_ptr(), s)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / K)
            n, last = C.c_uint(), C.c_uint()
            node._lib.atmo_get_split_stats(node._ctx, C.byref(n), C.byref(last))
            print(f"{wl:15s} {pose:8s} {w}x{h} {mode:10s}: {best:.4f} ms per draw ({node.kernel_name}), {n.value:3d} draws split, {last.value:5d} heavy tiles", flush=True)
            node.close()
