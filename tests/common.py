"""Shared scene construction for the tests (the demo scene lives in godot_atmosphere_shader_amd/demo.py)."""
from godot_atmosphere_shader_amd import demo as _demo
from godot_atmosphere_shader_amd.demo import CONFIGS, ROT, demo_frame, demo_params, demo_textures  # noqa: F401

TOL = 1e-4  # BASELINE.json north_star: <= 1e-4 max per-channel deviation


def make_node(config_name, textures, params=None, device=0, **extra):
    """demo.make_node with the cubemap sampler STATED: the parity cases written in rounds 1-3 hold the LOD-0 kernels against the
    oracle's LOD-0 sampler (config key cube_lod absent) and say so here; the cases for the library's default -- the declared
    linear-mipmap sampler -- pass cubemap_lod=None or True and give the oracle the mip chain and cube_lod=1."""
    extra.setdefault("cubemap_lod", False)
    return _demo.make_node(config_name, textures, params, device=device, **extra)
