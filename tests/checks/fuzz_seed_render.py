"""Renders one scene of tests/test_gpu_parity.py::test_parity_random_scenes (declared cubemap sampler, or "lod0") and saves the frame and the
oracle's: python tests/checks/fuzz_seed_render.py <seed> <out.npz> [lod0]   (ATMO_HIP_LIB selects another library build, e.g. last round's)"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
from godot_atmosphere_shader_amd import scene as S, PlanetAtmosphere, load_shader
from godot_atmosphere_shader_amd.planet_atmosphere import make_frame
from oracle.oracle import Oracle
import test_gpu_parity as T

seed, out = int(sys.argv[1]), sys.argv[2]
lod0 = len(sys.argv) > 3 and sys.argv[3] == "lod0"
rng = np.random.default_rng(1000 + seed)
params, cam, sun = T._random_scene(rng, seed)
shape_n = [64, 32, 24, 48][seed % 4]
cube_n = [256, 64, 17, 128][seed % 4]
tex = dict(blue_noise=S.make_blue_noise(seed + 1), shape=S.make_shape_texture(shape_n, seed=seed, cells=4),
           cubemap=None if seed % 5 == 4 else S.make_coverage_cubemap(cube_n, seed=seed))
variants = [
    ("planet_atmosphere_no_clouds", dict(view_steps=16), dict(view_steps=16)),
    ("planet_atmosphere_no_clouds", dict(view_steps=64, light_steps=5), dict(view_steps=64, light_mode="direct", light_steps=5)),
    ("planet_atmosphere_clouds", dict(view_steps=8, cloud_steps=8), dict(cloud_steps=8)),
    ("planet_atmosphere_clouds_high_rm", dict(view_steps=8, cloud_steps=24, cloud_light_rm=1), dict(cloud_steps=24)),
    ("planet_atmosphere_clouds_high", dict(view_steps=12, cloud_steps=64, light_steps=8), dict(view_steps=12, light_mode="direct", light_steps=8)),
    ("planet_atmosphere_clouds_high_rm", dict(view_steps=8, cloud_steps=64, cloud_light_rm=1, light_steps=3), dict(light_mode="direct", light_steps=3)),
]
shader, ocfg, kw = variants[seed % len(variants)]
depth = S.depth_ground_sphere(cam, radius=params["u_planet_radius"]) if seed % 3 else S.depth_far(cam)
node = PlanetAtmosphere(blue_noise=tex["blue_noise"], cubemap_lod=not lod0, **kw)
node.custom_shader = load_shader(shader)
node.planet_radius, node.atmosphere_height, node.sun_path = params["u_planet_radius"], params["u_atmosphere_height"], sun
for k, v in params.items():
    if k not in ("u_planet_radius", "u_atmosphere_height", "u_cloud_coverage_rotation", "u_world_to_model_matrix"):
        node.set(f"shader_params/{k}", v)
node._process(0.0, cam, time=0.0)
node.set_shader_parameter("u_cloud_coverage_rotation", np.asarray(params["u_cloud_coverage_rotation"], dtype=np.float32))
node.set_shader_parameter("u_cloud_shape_texture", tex["shape"])
node.set_shader_parameter("u_cloud_coverage_cubemap", tex["cubemap"])
got = T._gpu_render(node, cam, depth)
name = node.kernel_name
lut = node.read_optical_depth() if "light_steps" not in ocfg else None
node.close()
o = Oracle("f32")
oparams = dict(params, u_atmosphere_modulate=tuple(S.srgb_to_linear(params["u_atmosphere_modulate"]).tolist()),
               u_atmosphere_ambient_color=tuple(S.srgb_to_linear(params["u_atmosphere_ambient_color"]).tolist()))
want, hits = o.render(oparams, dict(tex, optical_depth=lut, cubemap=tex["cubemap"] if lod0 else o.cubemap_mip_chain(tex["cubemap"])),
                      ocfg if lod0 else dict(ocfg, cube_lod=1), make_frame(cam, np.eye(4), sun), depth, nthreads=8)
err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
bad = np.argwhere(err.max(axis=-1) > 1e-4)
print(f"seed {seed} {'lod0' if lod0 else 'declared'} {shader} {name}: u_cloud_density_scale {params['u_cloud_density_scale']:.1f} H {params['u_atmosphere_height']:.3g}; viewport {cam.width}x{cam.height}, max err {err.max():.3e}, {len(bad)} pixels over 1e-4:", bad[:12].tolist())
for y, x in bad[:6]:
    print("   ", (int(y), int(x)), "got", got[y, x], "want", want[y, x])
np.savez(out, got=got, want=want)
