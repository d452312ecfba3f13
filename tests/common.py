"""Shared scene construction for the tests (the demo scene lives in godot_atmosphere_shader_amd/demo.py)."""
from godot_atmosphere_shader_amd import demo as _demo
from godot_atmosphere_shader_amd.demo import CONFIGS, ROT, demo_frame, demo_params, demo_textures  # noqa: F401

TOL = 1e-4  # BASELINE.json north_star: <= 1e-4 max per-channel deviation

# The coverage cubemap's sampler.  "declared" = what cloud_funcs.gdshaderinc:15,45 declares (no filter hint: linear-mipmap with the
# implicit LOD of the 2 x 2 pixel quad) -- the library's default, and since round 5 the default of every test: a new test gets the
# kernels the library ships by default.  "lod0" = level 0 only (atmo_set_sampler_lod 0), the convention of rounds 1-3, which a test
# has to ASK for.
SAMPLERS = ("declared", "lod0")


def has_clouds(config_name):
    return bool(CONFIGS[config_name][1].get("cloud_steps"))


def kernel_flags(node):
    """FLAGS of atmo_render_kernel<FLAGS, LSTEPS, SPLIT>: bit 5 (32) = the declared-sampler (implicit LOD) kernels."""
    return int(node.kernel_name.split("<")[1].split(",")[0])


def make_node(config_name, textures, params=None, device=0, sampler="declared", **extra):
    """demo.make_node with the cubemap sampler stated by name.  A cloud variant with a mip-able cubemap bound and the precise cloud
    mode (the default) must then run the declared-sampler kernels -- checked here, so no parity case can pass on the other family."""
    assert sampler in SAMPLERS, sampler
    if "cubemap_lod" not in extra:
        extra["cubemap_lod"] = None if sampler == "declared" else False
    node = _demo.make_node(config_name, textures, params, device=device, **extra)
    if has_clouds(config_name) and extra["cubemap_lod"] in (None, False):
        cube = textures.get("cubemap")
        chain = cube is not None and not isinstance(cube, (list, tuple)) and extra.get("precise_clouds", True)
        want = bool(chain) and extra["cubemap_lod"] is None
        assert bool(kernel_flags(node) & 32) == want, (node.kernel_name, sampler)
    return node


def oracle_inputs(oracle, cfg, textures, lut=None, declared=True):
    """(config dict, texture dict) for Oracle.render under the given sampler: the declared sampler gets the 2 x 2-box mip chain
    (noise_cubemap.gd:107,135) and cube_lod=1, like the kernels that generate the chain on the device."""
    cfg, tex = dict(cfg), dict(textures, optical_depth=lut)
    if declared and cfg.get("cloud_steps") and tex.get("cubemap") is not None and not isinstance(tex["cubemap"], (list, tuple)):
        cfg["cube_lod"] = 1
        tex["cubemap"] = oracle.cubemap_mip_chain(tex["cubemap"])
    return cfg, tex
