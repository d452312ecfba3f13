"""Fetch ablation: kernel time with every texture gather in place against a DIAGNOSTIC build that replaces the gathers by a
constant (tools/ab_build.sh ablate -DATMO_ABLATE_FETCH=1), on frames whose textures ARE that constant -- so both builds draw the same picture through
the same control flow.  The coverage bias sweeps the cloud layer from nearly empty to overcast (the share of samples that reach the shape filter / are
lit changes with it); the frame checksum printed per line must agree between the two builds.

    ATMO_HIP_LIB=.../libatmo_hip_ablate.so python tools/fetch_ablation_probe.py     (and once without ATMO_HIP_LIB)"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_params, make_node  # noqa: E402

w, h = 1920, 1080
C = np.float32(0.50196081399917603)
tex = dict(blue_noise=S.make_blue_noise(), shape=np.full((64, 64, 64), 128, dtype=np.uint8), cubemap=np.full((6, 256, 256), 128, dtype=np.uint8))
arm = "ablated" if "ablate" in os.environ.get("ATMO_HIP_LIB", "") else "gathers"
cases = [("no_clouds_32_lut", {}, None), ("no_clouds_8", {}, None)]
for wl, kw in (("clouds_high", dict(cubemap_lod=False)), ("clouds_high", {}), ("clouds_high_rm", dict(cubemap_lod=False)), ("clouds_high_rm", {})):
    for bias in (0.10, 0.16, 0.22, 0.40):
        cases.append((wl, kw, bias))
for wl, kw, bias in cases:
    params = demo_params() if bias is None else demo_params(u_cloud_coverage_bias=bias)
    for pose in ("P_space",):
        cam = S.Camera.from_pose(w, h, pose)
        depth = torch.from_numpy(S.depth_ground_sphere(cam)).cuda()
        node = make_node(wl, tex, params, **kw)
        node._bake_if_needed()
        node.set_shader_parameter("u_optical_depth_texture", np.full((256, 256), C, dtype=np.float32))
        out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        frame = node.prepare_frame(cam)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(30):
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
        img = out.cpu().numpy()
        lit = float((img[..., 3] > 0.5).mean())
        print(f"{arm:8s} {wl:17s} {node.kernel_name:30s} bias {bias if bias is not None else '-':>5}: {best:.4f} ms  crc {zlib.crc32(img.tobytes()):08x}  alpha>0.5 on {lit:.3f} of the frame", flush=True)
        node.close()
