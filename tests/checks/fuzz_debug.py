import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
from godot_atmosphere_shader_amd import scene as S, PlanetAtmosphere, load_shader
from godot_atmosphere_shader_amd.planet_atmosphere import make_frame
from oracle.oracle import Oracle
import test_gpu_parity as T
o32 = Oracle("f32"); o64 = Oracle("f64")
for seed in (55, 91):
    rng = np.random.default_rng(1000 + seed)
    params, cam, sun = T._random_scene(rng, seed)
    tex = dict(blue_noise=S.make_blue_noise(seed + 1))
    depth = S.depth_ground_sphere(cam, radius=params["u_planet_radius"]) if seed % 3 else S.depth_far(cam)
    p2 = dict(params, u_atmosphere_modulate=tuple(S.srgb_to_linear(params["u_atmosphere_modulate"]).tolist()),
              u_atmosphere_ambient_color=tuple(S.srgb_to_linear(params["u_atmosphere_ambient_color"]).tolist()))
    print("seed", seed, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in params.items() if k in ("u_planet_radius", "u_atmosphere_height", "u_density", "u_scattering_strength", "u_sphere_depth_factor")})
    for vs, ls in ((64, 5), (64, 8), (16, 5), (32, 8), (64, 0)):
        kw = dict(view_steps=vs) if ls == 0 else dict(view_steps=vs, light_mode="direct", light_steps=ls)
        ocfg = dict(view_steps=vs) if ls == 0 else dict(view_steps=vs, light_steps=ls)
        node = PlanetAtmosphere(blue_noise=tex["blue_noise"], **kw)
        node.custom_shader = load_shader("planet_atmosphere_no_clouds")
        node.planet_radius, node.atmosphere_height, node.sun_path = params["u_planet_radius"], params["u_atmosphere_height"], sun
        for k, v in params.items():
            if k.startswith("u_") and k not in ("u_planet_radius", "u_atmosphere_height", "u_cloud_coverage_rotation", "u_world_to_model_matrix") and "cloud" not in k:
                node.set(f"shader_params/{k}", v)
        node._process(0.0, cam, time=0.0)
        got = T._gpu_render(node, cam, depth)
        lut = node.read_optical_depth() if ls == 0 else None
        node.close()
        fr = make_frame(cam, np.eye(4), sun)
        w32, _ = o32.render(p2, dict(tex, optical_depth=lut), ocfg, fr, depth, nthreads=8)
        w64, _ = o64.render(p2, dict(tex, optical_depth=lut), ocfg, fr, depth, nthreads=8)
        e32 = np.abs(got - w32); e64 = np.abs(got - w64); eo = np.abs(w32 - w64)
        i = np.unravel_index(np.argmax(e32), e32.shape)
        print(f"  view {vs:2d} light {ls}: |hip-o32| {e32.max():.3e} at {i} (values hip {got[i]:.6f} o32 {w32[i]:.6f} o64 {w64[i]:.6f})  |hip-o64| {e64.max():.3e}  |o32-o64| {eo.max():.3e}")
