"""The reference's demo scene (addons/zylann.atmosphere/demo/planet_atmosphere_test.tscn:96-114) with seeded synthetic
textures, and the named shader configurations used by tests, bench.py and smoke(): one place that says what
"clouds_high at the demo parameters" means for both the product binding and the checker's config dicts."""
from __future__ import annotations

import math

import numpy as np

from . import scene as S
from .planet_atmosphere import make_frame

ROT = 0.3  # fixed cloud-coverage rotation angle

# name -> (shader variant file, step-count config {view_steps, cloud_steps, cloud_light_rm, light_steps, lite}, PlanetAtmosphere kwargs)
CONFIGS = {
    "no_clouds_8": ("planet_atmosphere_no_clouds", dict(view_steps=8), {}),
    "no_clouds_32_lut": ("planet_atmosphere_no_clouds", dict(view_steps=32), dict(view_steps=32)),
    "no_clouds_32x8_direct": ("planet_atmosphere_no_clouds", dict(view_steps=32, light_steps=8),
                              dict(view_steps=32, light_mode="direct", light_steps=8)),
    "clouds": ("planet_atmosphere_clouds", dict(view_steps=8, cloud_steps=32), {}),
    "clouds_high": ("planet_atmosphere_clouds_high", dict(view_steps=8, cloud_steps=64), {}),
    "clouds_high_rm": ("planet_atmosphere_clouds_high_rm", dict(view_steps=8, cloud_steps=64, cloud_light_rm=1), {}),
    # ATMOSPHERE_LITE variants
    "v1_no_clouds": ("planet_atmosphere_v1_no_clouds", dict(view_steps=16, lite=1), {}),
    "v1_clouds": ("planet_atmosphere_v1_clouds", dict(view_steps=16, lite=1, cloud_steps=32), {}),
    "v1_clouds_high": ("planet_atmosphere_v1_clouds_high", dict(view_steps=16, lite=1, cloud_steps=64), {}),
}

_tex_cache = {}


def demo_textures(cube_n=256, shape_n=64):
    key = (cube_n, shape_n)
    if key not in _tex_cache:
        _tex_cache[key] = dict(blue_noise=S.make_blue_noise(), shape=S.make_shape_texture(shape_n),
                               cubemap=S.make_coverage_cubemap(cube_n))
    return dict(_tex_cache[key])


def demo_params(**over):
    c, s = math.cos(ROT), math.sin(ROT)
    p = dict(S.DEMO_SHADER_PARAMS, u_planet_radius=S.DEMO_PLANET_RADIUS, u_atmosphere_height=S.DEMO_ATMOSPHERE_HEIGHT,
             u_cloud_coverage_rotation=(c, s, -s, c),
             u_world_to_model_matrix=(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1))
    p.update(over)
    return p


def demo_frame(cam, rect=None):
    return make_frame(cam, np.eye(4), S.DEMO_SUN_POSITION, 0.0, rect)


def make_node(config_name, textures, params=None, device=0, **extra):
    """A PlanetAtmosphere set up like the demo scene for one of CONFIGS."""
    from .planet_atmosphere import PlanetAtmosphere, load_shader

    shader, _, kw = CONFIGS[config_name]
    node = PlanetAtmosphere(device=device, blue_noise=textures["blue_noise"], **kw, **extra)
    node.custom_shader = load_shader(shader)
    params = params or demo_params()
    node.planet_radius = params["u_planet_radius"]
    node.atmosphere_height = params["u_atmosphere_height"]
    node.sun_path = S.DEMO_SUN_POSITION
    from .planet_atmosphere import _SOURCE_COLOR, LinearColor

    for k, v in params.items():
        if k in ("u_planet_radius", "u_atmosphere_height", "u_cloud_coverage_rotation", "u_world_to_model_matrix"):
            continue
        if k in _SOURCE_COLOR:
            v = LinearColor(v)  # the scene dictionaries (scene.DEMO_SHADER_PARAMS, the oracle's inputs) hold linear colours
        node.set(f"shader_params/{k}", v)
    node._process(0.0, None, time=0.0)
    node.set_shader_parameter("u_cloud_coverage_rotation", np.asarray(params["u_cloud_coverage_rotation"], dtype=np.float32))
    if "shape" in textures:
        node.set_shader_parameter("u_cloud_shape_texture", textures["shape"])
    if textures.get("cubemap") is not None:
        node.set_shader_parameter("u_cloud_coverage_cubemap", textures["cubemap"])
    return node
