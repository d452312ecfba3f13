"""MI355X-native (gfx950) implementation of the per-pixel atmosphere / volumetric-cloud raymarch of
Zylann/godot_atmosphere_shader, behind the reference's `PlanetAtmosphere` / `shader_params` surface.

  csrc/                 hand-written HIP kernels + the C ABI of include/atmo.h (libatmo_hip.so)
  planet_atmosphere.py  host-side mirror of addons/zylann.atmosphere/planet_atmosphere.gd
  noise_cubemap.py      host-side mirror of addons/zylann.atmosphere/noise_cubemap.gd (generation on the GPU)
  demo.py               the reference's demo scene + named shader configurations (tests, bench.py, smoke())
  scene.py              synthetic inputs (camera, depth, jitter, cloud textures) for tests and bench
  sharding.py           row-band / viewport sharding across the GPUs of a node + RCCL gather
"""
from .planet_atmosphere import (  # noqa: F401
    DefaultShader, PlanetAtmosphere, Shader, SHADERS, Transform2D, atmosphere_vertex, load_shader, make_frame,
)

from .noise_cubemap import NoiseCubemap, SeededValueNoise  # noqa: F401,E402

__all__ = ["NoiseCubemap", "SeededValueNoise", "PlanetAtmosphere", "Shader", "SHADERS", "DefaultShader", "Transform2D", "load_shader",
           "atmosphere_vertex", "make_frame"]
