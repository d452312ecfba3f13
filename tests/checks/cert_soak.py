"""Soak of the declared sampler's level-0 certificate: the random scenes of tests/test_gpu_parity.py::test_parity_random_scenes, re-rendered at frame
sizes where certain and uncertain samples mix, with the certificate (default) and without it (ATMO_LOD0_CERT=0, read at atmo_create): every frame
pair must be bit-identical.      python tests/checks/cert_soak.py [n_scenes] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from godot_atmosphere_shader_amd import scene as S, PlanetAtmosphere, load_shader
import test_gpu_parity as T

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
sizes = [(1280, 720), (1920, 1080), (960, 540), (2560, 1440), (640, 360)]
variants = [("planet_atmosphere_clouds", dict(cloud_steps=8)), ("planet_atmosphere_clouds_high", dict()), ("planet_atmosphere_clouds_high_rm", dict(cloud_steps=24)),
            ("planet_atmosphere_clouds_high_rm", dict()), ("planet_atmosphere_v1_clouds_high", dict())]
bad, differs = [], 0
for seed in range(first, first + n):
    rng = np.random.default_rng(1000 + seed)
    params, cam0, sun = T._random_scene(rng, seed)
    w, h = sizes[seed % len(sizes)]
    iv = cam0.inv_view
    cam = S.Camera(w, h, iv[:3, 3], iv[:3, 3] - iv[:3, 2], up=iv[:3, 1], fovy_deg=cam0.fovy_deg, near=cam0.near, far=cam0.far, reverse_z=cam0.reverse_z)
    tex = dict(blue_noise=S.make_blue_noise(seed + 1), shape=S.make_shape_texture([64, 32, 24, 48][seed % 4], seed=seed, cells=4),
               cubemap=S.make_coverage_cubemap([256, 64, 1024, 128, 16][seed % 5], seed=seed))
    shader, kw = variants[seed % len(variants)]
    depth = torch.from_numpy(S.depth_ground_sphere(cam, radius=params["u_planet_radius"]) if seed % 3 else S.depth_far(cam)).cuda()
    frames = []
    for mode in ("1", "0", "lod0"):
        if mode != "lod0":
            os.environ["ATMO_LOD0_CERT"] = mode
        node = PlanetAtmosphere(blue_noise=tex["blue_noise"], cubemap_lod=None if mode != "lod0" else False, **kw)
        os.environ.pop("ATMO_LOD0_CERT", None)
        node.custom_shader = load_shader(shader)
        node.planet_radius, node.atmosphere_height, node.sun_path = params["u_planet_radius"], params["u_atmosphere_height"], sun
        for k, v in params.items():
            if k not in ("u_planet_radius", "u_atmosphere_height", "u_cloud_coverage_rotation", "u_world_to_model_matrix"):
                node.set(f"shader_params/{k}", v)
        node._process(0.0, cam, time=0.0)
        node.set_shader_parameter("u_cloud_coverage_rotation", np.asarray(params["u_cloud_coverage_rotation"], dtype=np.float32))
        node.set_shader_parameter("u_cloud_shape_texture", tex["shape"])
        node.set_shader_parameter("u_cloud_coverage_cubemap", tex["cubemap"])
        frames.append(node.render(cam, depth).clone())
        node.close()
    same = torch.equal(frames[0], frames[1]) or bool(((frames[0] == frames[1]) | (frames[0].isnan() & frames[1].isnan())).all())
    differs += int(not torch.equal(frames[0], frames[2]))
    if not same:
        bad.append(seed)
        print(f"seed {seed}: certificate on != off  ({shader} {w}x{h}, max |diff| {float((frames[0] - frames[1]).abs().nan_to_num().max()):.3e})")
print(f"{n} scenes from seed {first}: {n - len(bad)} bit-identical with and without the certificate, {len(bad)} differ {bad}; "
      f"{differs} of them differ from the level-0 sampler's frame (lambda > 0 somewhere)")
sys.exit(1 if bad else 0)
