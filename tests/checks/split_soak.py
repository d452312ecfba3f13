"""Soak of round 5's two-lanes-per-ray forms of the declared-sampler kernels: the random scenes of tests/test_gpu_parity.py::test_parity_random_scenes
(random planets, cameras inside / outside the layer, suns, parameter sets, cubemaps of 16 .. 1024 texels incl. one that is not on the fast path, shape
volumes that are not powers of two), re-rendered at frame sizes with thousands of tiles,
  (a) one lane per ray, row-major order (atmo_set_tile_feedback 0)                -- the reference frame,
  (b) the WHOLE frame on two lanes per ray (atmo_set_lane_split 2: <49, 0, 2> / <51, 0, 2>),
  (c) the mixed draw: tile-order feedback on, the heavy tiles FORCED onto the lane-split kernel beside the rest (ATMO_HEAVY_SPLIT=2, a threshold
      that splits up to a third of the tiles), eight draws so that orders and class totals arrive.
Every frame of (b) and (c) must equal (a) bit for bit.      python tests/checks/split_soak.py [n_scenes] [first_seed]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from godot_atmosphere_shader_amd import scene as S, PlanetAtmosphere, load_shader
import test_gpu_parity as T

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
sizes = [(1280, 720), (960, 540), (1920, 1080), (640, 360), (1600, 900)]
variants = [("planet_atmosphere_clouds_high_rm", dict()), ("planet_atmosphere_clouds_high", dict()), ("planet_atmosphere_clouds_high_rm", dict(cloud_steps=24)),
            ("planet_atmosphere_clouds_high", dict(cloud_steps=9))]
bad, engaged = [], 0
for seed in range(first, first + n):
    rng = np.random.default_rng(1000 + seed)
    params, cam0, sun = T._random_scene(rng, seed)
    w, h = sizes[seed % len(sizes)]
    iv = cam0.inv_view
    cam = S.Camera(w, h, iv[:3, 3], iv[:3, 3] - iv[:3, 2], up=iv[:3, 1], fovy_deg=cam0.fovy_deg, near=cam0.near, far=cam0.far, reverse_z=cam0.reverse_z)
    tex = dict(blue_noise=S.make_blue_noise(seed + 1), shape=S.make_shape_texture([64, 32, 24, 48][seed % 4], seed=seed, cells=4),
               cubemap=S.make_coverage_cubemap([256, 64, 1024, 128, 17][seed % 5], seed=seed))
    shader, kw = variants[seed % len(variants)]
    depth = torch.from_numpy(S.depth_ground_sphere(cam, radius=params["u_planet_radius"]) if seed % 3 else S.depth_far(cam)).cuda()

    def node_for(mode):
        env = {"a": {"ATMO_HEAVY_SPLIT": "0"}, "b": {"ATMO_HEAVY_SPLIT": "0"}, "c": {"ATMO_HEAVY_SPLIT": "2", "ATMO_HEAVY_SPLIT_RATIO": "0.05"}}[mode]
        os.environ.update(env)
        node = PlanetAtmosphere(blue_noise=tex["blue_noise"], lane_split=2 if mode == "b" else 1, tile_feedback=1 if mode == "c" else 0, **kw)
        for k in env:
            os.environ.pop(k, None)
        node.custom_shader = load_shader(shader)
        node.planet_radius, node.atmosphere_height, node.sun_path = params["u_planet_radius"], params["u_atmosphere_height"], sun
        for k, v in params.items():
            if k not in ("u_planet_radius", "u_atmosphere_height", "u_cloud_coverage_rotation", "u_world_to_model_matrix"):
                node.set(f"shader_params/{k}", v)
        node._process(0.0, cam, time=0.0)
        node.set_shader_parameter("u_cloud_coverage_rotation", np.asarray(params["u_cloud_coverage_rotation"], dtype=np.float32))
        node.set_shader_parameter("u_cloud_shape_texture", tex["shape"])
        node.set_shader_parameter("u_cloud_coverage_cubemap", tex["cubemap"])
        return node

    a = node_for("a")
    want = a.render(cam, depth).clone()
    a.close()
    b = node_for("b")
    got_b = b.render(cam, depth).clone()
    name_b = b.kernel_name
    b.close()
    c = node_for("c")
    ok_c = True
    for k in range(8):
        out = torch.full_like(want, float("nan"))
        c.render(cam, depth, out=out)
        torch.cuda.synchronize()
        ok_c = ok_c and bool(((out == want) | (out.isnan() & want.isnan())).all())
    nsplit = C.c_uint()
    c._lib.atmo_get_split_stats(c._ctx, C.byref(nsplit), None)
    engaged += int(nsplit.value > 0)
    c.close()
    ok_b = bool(((got_b == want) | (got_b.isnan() & want.isnan())).all())
    if not (ok_b and ok_c):
        bad.append(seed)
        print(f"seed {seed}: {shader} {kw} {w}x{h}: whole frame on two lanes {'==' if ok_b else '!='} one lane ({name_b}); mixed draw {'==' if ok_c else '!='} plain draw")
print(f"{n} scenes from seed {first}: {n - len(bad)} bit-identical in both forms, {len(bad)} differ {bad}; the mixed draw engaged in {engaged} scenes")
sys.exit(1 if bad else 0)
