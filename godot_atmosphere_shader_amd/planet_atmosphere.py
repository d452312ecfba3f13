"""Host-side mirror of the reference's `PlanetAtmosphere` node (addons/zylann.atmosphere/
planet_atmosphere.gd), driving the gfx950 kernels through the C ABI of include/atmo.h.

The reference host is GDScript; Godot is not available here, so the host side is Python with the same
names, argument meaning and (silent) error behaviour:

  reference (planet_atmosphere.gd)                     here
  ---------------------------------------------------  ------------------------------------------
  planet_radius / atmosphere_height (:20-33,230-253)   properties, trigger a LUT re-bake
  sun_path (:36-41)                                    `sun_path`: anything with `.global_position`, or an xyz
  custom_shader (:44-49,118-141)                       `custom_shader`: a `Shader` from `load_shader()`
  clouds_rotation_speed, force_fullscreen (:52-54)     same
  set/get_shader_parameter (:175-180)                  same (unknown names are kept and ignored, as Godot does)
  set/get_shader_param (:164-172)                      same, with a DeprecationWarning
  _get/_set "shader_params/<name>" (:200-218)          `get()` / `set()`; u_density triggers a re-bake (:79-81)
  _get_property_list (:185-197)                        `get_property_list()`
  _process (:285-341)                                  `_process(delta, camera, time)`: per-frame uniforms
  (the draw itself: Godot renderer)                    `render(camera, depth, out, rect, stream)`

The fragment work runs only on the GPU: `render` raises if libatmo_hip.so or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import math
import time as _time
import warnings

import numpy as np

from . import _native as N
from .scene import col_major, srgb_to_linear

MODE_NEAR = 0
MODE_FAR = 1
SWITCH_MARGIN_RATIO = 1.1

_SHADER_DIR = "res://addons/zylann.atmosphere/shaders/"

# uniforms each shader variant declares (used by get_property_list and by the baker's "does it use the LUT" test)
_V2_UNIFORMS = [
    "u_planet_radius", "u_atmosphere_height", "u_sun_position", "u_density", "u_optical_depth_texture",
    "u_scattering_strength", "u_scattering_wavelengths", "u_atmosphere_modulate", "u_atmosphere_ambient_color",
    "u_clip_mode", "u_sphere_depth_factor", "u_blue_noise_texture",
]
_V1_UNIFORMS = [
    "u_planet_radius", "u_atmosphere_height", "u_sun_position", "u_density", "u_day_color0", "u_day_color1",
    "u_night_color0", "u_night_color1", "u_day_night_transition_scale", "u_clip_mode", "u_sphere_depth_factor",
    "u_blue_noise_texture",
]
_CLOUD_UNIFORMS = [
    "u_cloud_density_scale", "u_cloud_bottom", "u_cloud_top", "u_cloud_blend", "u_world_to_model_matrix",
    "u_cloud_shape_texture", "u_cloud_shape_invert", "u_cloud_coverage_bias", "u_cloud_shape_factor",
    "u_cloud_shape_scale", "u_cloud_coverage_cubemap", "u_cloud_coverage_rotation",
]

# GDShader defaults
SHADER_DEFAULTS = {
    "u_planet_radius": 1.0, "u_atmosphere_height": 0.1, "u_sun_position": (0.0, 0.0, 0.0), "u_density": 0.2,
    "u_scattering_strength": 20.0, "u_scattering_wavelengths": (700.0, 530.0, 440.0),
    "u_atmosphere_modulate": (1.0, 1.0, 1.0), "u_atmosphere_ambient_color": (0.0, 0.0, 0.002),
    "u_clip_mode": False, "u_sphere_depth_factor": 0.0, "u_cloud_density_scale": 50.0, "u_cloud_bottom": 0.2,
    "u_cloud_top": 0.5, "u_cloud_blend": 0.5, "u_cloud_shape_invert": 0.0, "u_cloud_coverage_bias": 0.0,
    "u_cloud_shape_factor": 0.8, "u_cloud_shape_scale": 1.0,
    # `source_color` uniforms hold the sRGB values written in the shader / shown by the inspector; `_forward` applies
    # the engine's sRGB -> linear conversion on upload (atmosphere_funcs_v2.gdshaderinc:10-11, _v1.gdshaderinc:8-12)
    "u_day_color0": (0.5, 0.8, 1.0, 1.0), "u_day_color1": (0.5, 0.8, 1.0, 1.0),
    "u_night_color0": (0.2, 0.4, 0.8, 1.0), "u_night_color1": (0.2, 0.4, 0.8, 1.0),
    "u_day_night_transition_scale": 2.0,
}

_FLOAT_COUNTS = {
    "u_planet_radius": 1, "u_atmosphere_height": 1, "u_sun_position": 3, "u_density": 1, "u_scattering_strength": 1,
    "u_scattering_wavelengths": 3, "u_atmosphere_modulate": 3, "u_atmosphere_ambient_color": 3, "u_clip_mode": 1,
    "u_sphere_depth_factor": 1, "u_cloud_density_scale": 1, "u_cloud_bottom": 1, "u_cloud_top": 1, "u_cloud_blend": 1,
    "u_world_to_model_matrix": 16, "u_cloud_shape_invert": 1, "u_cloud_coverage_bias": 1, "u_cloud_shape_factor": 1,
    "u_cloud_shape_scale": 1, "u_cloud_coverage_rotation": 4,
    "u_day_color0": 4, "u_day_color1": 4, "u_night_color0": 4, "u_night_color1": 4, "u_day_night_transition_scale": 1,
}
# uniforms declared `source_color`: Godot converts their rgb from sRGB to linear when the material uploads them
_SOURCE_COLOR = frozenset(["u_atmosphere_modulate", "u_atmosphere_ambient_color", "u_day_color0", "u_day_color1",
                           "u_night_color0", "u_night_color1"])


class LinearColor(tuple):
    """A colour value that is ALREADY linear: `set_shader_parameter(name, LinearColor(rgb))` uploads it unchanged
    (the opt-out of the `source_color` conversion, for hosts that keep linear colours)."""

    def __new__(cls, *v):
        if len(v) == 1 and not isinstance(v[0], (int, float)):
            v = tuple(v[0])
        return super().__new__(cls, tuple(float(x) for x in v))


_TEXTURES = {
    "u_optical_depth_texture": N.TEX_2D_R32F, "u_blue_noise_texture": N.TEX_2D_R8,
    "u_cloud_shape_texture": N.TEX_3D_R8, "u_cloud_coverage_cubemap": N.TEX_CUBE_R8,
}


class Shader:
    """Stands for one of the reference's .gdshader variant files: a set of #defines
    (shaders/planet_atmosphere_*.gdshader:4-7)."""

    def __init__(self, name, variant, view_steps, cloud_steps, cloud_light_rm, lite=False):
        self.name = name
        self.variant = variant
        self.lite = lite                      # ATMOSPHERE_LITE
        self.view_steps = view_steps          # ATMOSPHERE_RAYMARCH_STEPS
        self.cloud_steps = cloud_steps        # CLOUDS_MAX_RAYMARCH_STEPS (0: CLOUDS_ENABLED undefined)
        self.cloud_light_rm = cloud_light_rm  # CLOUDS_RAYMARCHED_LIGHTING
        self.resource_path = _SHADER_DIR + name + ".gdshader"

    def get_shader_uniform_list(self):
        names = list(_V1_UNIFORMS if self.lite else _V2_UNIFORMS) + (list(_CLOUD_UNIFORMS) if self.cloud_steps else [])
        return [{"name": n} for n in names]

    def __repr__(self):
        return f"Shader({self.name})"


SHADERS = {
    "planet_atmosphere_no_clouds": Shader("planet_atmosphere_no_clouds", N.VARIANT_NO_CLOUDS, 8, 0, False),
    "planet_atmosphere_clouds": Shader("planet_atmosphere_clouds", N.VARIANT_CLOUDS, 8, 32, False),
    "planet_atmosphere_clouds_high": Shader("planet_atmosphere_clouds_high", N.VARIANT_CLOUDS_HIGH, 8, 64, False),
    "planet_atmosphere_clouds_high_rm": Shader("planet_atmosphere_clouds_high_rm", N.VARIANT_CLOUDS_HIGH_RM, 8, 64, True),
    "planet_atmosphere_v1_no_clouds": Shader("planet_atmosphere_v1_no_clouds", N.VARIANT_V1_NO_CLOUDS, 16, 0, False, lite=True),
    "planet_atmosphere_v1_clouds": Shader("planet_atmosphere_v1_clouds", N.VARIANT_V1_CLOUDS, 16, 32, False, lite=True),
    "planet_atmosphere_v1_clouds_high": Shader("planet_atmosphere_v1_clouds_high", N.VARIANT_V1_CLOUDS_HIGH, 16, 64, False, lite=True),
}
DefaultShader = SHADERS["planet_atmosphere_no_clouds"]  # planet_atmosphere.gd:13-14


def load_shader(path: str) -> Shader:
    """`preload("./shaders/<name>.gdshader")`: accepts a res:// path, a file name or a bare variant name.
    README.md:35 calls the raymarched-lighting variant `..._clouds_high_m`; the file is `..._clouds_high_rm`."""
    name = path.rsplit("/", 1)[-1]
    if name.endswith(".gdshader"):
        name = name[: -len(".gdshader")]
    if name == "planet_atmosphere_clouds_high_m":
        name = "planet_atmosphere_clouds_high_rm"
    if name not in SHADERS:
        raise FileNotFoundError(path)
    return SHADERS[name]


class Transform2D:
    """Just enough of Godot's Transform2D for u_cloud_coverage_rotation (planet_atmosphere.gd:340-341)."""

    def __init__(self, x=(1.0, 0.0), y=(0.0, 1.0)):
        self.x, self.y = tuple(x), tuple(y)  # basis columns

    def rotated(self, angle: float) -> "Transform2D":
        c, s = math.cos(angle), math.sin(angle)
        return Transform2D((c, s), (-s, c))

    def as_mat2_col_major(self):
        return np.array([self.x[0], self.x[1], self.y[0], self.y[1]], dtype=np.float32)


def _mat4_vec4_f32(m: np.ndarray, v) -> np.ndarray:
    """fp32 mat4*vec4 summed left to right (what the vertex stage computes)."""
    m = np.asarray(m, dtype=np.float32)
    v = [np.float32(x) for x in v]
    out = np.empty(4, dtype=np.float32)
    for r in range(4):
        out[r] = ((m[r, 0] * v[0] + m[r, 1] * v[1]) + m[r, 2] * v[2]) + m[r, 3] * v[3]
    return out


def atmosphere_vertex(view_matrix, model_matrix, sun_position):
    """The per-draw constants of atmosphere_vertex (shaders/include/planet_atmosphere_main.gdshaderinc:101-103):
    (v_planet_center_viewspace, v_sun_center_viewspace), fp32."""
    world_pos = _mat4_vec4_f32(model_matrix, (0.0, 0.0, 0.0, 1.0))
    planet = _mat4_vec4_f32(view_matrix, world_pos)[:3]
    sun = _mat4_vec4_f32(view_matrix, (sun_position[0], sun_position[1], sun_position[2], 1.0))[:3]
    return planet.copy(), sun.copy()


def make_frame(camera, model_matrix, sun_position, time=0.0, rect=None) -> dict:
    """Frame description shared by the product binding and the test oracle: plain dict of numpy values."""
    planet, sun = atmosphere_vertex(camera.view, model_matrix, sun_position)
    w, h = camera.width, camera.height
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, w, h)
    return dict(
        inv_projection_matrix=col_major(camera.inv_projection), inv_view_matrix=col_major(camera.inv_view),
        viewport_w=w, viewport_h=h, planet_center_viewspace=planet, sun_center_viewspace=sun, time=float(time),
        rect=(int(x0), int(y0), int(x1), int(y1)),
    )


def _to_native_frame(frame: dict) -> N.AtmoFrame:
    f = N.AtmoFrame()
    f.inv_projection_matrix[:] = [float(x) for x in frame["inv_projection_matrix"]]
    f.inv_view_matrix[:] = [float(x) for x in frame["inv_view_matrix"]]
    f.viewport_w, f.viewport_h = int(frame["viewport_w"]), int(frame["viewport_h"])
    f.planet_center_viewspace[:] = [float(x) for x in frame["planet_center_viewspace"]]
    f.sun_center_viewspace[:] = [float(x) for x in frame["sun_center_viewspace"]]
    f.time = float(frame.get("time", 0.0))
    f.x0, f.y0, f.x1, f.y1 = frame.get("rect", (0, 0, f.viewport_w, f.viewport_h))
    return f


class PlanetAtmosphere:
    """See module docstring.  One instance owns one AtmoContext on one GPU."""

    # parameters assigned internally (planet_atmosphere.gd:68-77)
    _api_shader_params = {
        "u_planet_radius": True, "u_atmosphere_height": True, "u_clip_mode": True, "u_sun_position": True,
        "u_world_to_model_matrix": True, "u_blue_noise_texture": True, "u_cloud_coverage_rotation": True,
        "u_optical_depth_texture": True,
    }
    _shader_params_affecting_optical_depth = {"u_density": True}  # planet_atmosphere.gd:79-81

    def __init__(self, device: int = 0, light_mode: str = "lut", light_steps: int = 0,
                 view_steps: int | None = None, cloud_steps: int | None = None, blue_noise=None,
                 precise_clouds: bool = True, precise_atmosphere: bool = False, double_precision: bool = False, lane_split: int = 0,
                 tile_feedback: int = -1, cubemap_lod: bool | None = None, target_cleared: bool = False):
        self._lib = N.load()
        self._device = int(device)
        self._light_mode = {"lut": N.LIGHT_LUT, "direct": N.LIGHT_DIRECT}[light_mode]
        self._light_steps = int(light_steps)
        self._view_steps_override = view_steps    # macro override of ATMOSPHERE_RAYMARCH_STEPS
        self._cloud_steps_override = cloud_steps  # macro override of CLOUDS_MAX_RAYMARCH_STEPS
        self._precise_clouds = bool(precise_clouds)  # atmo_set_precision: bit-faithful cloud density (default); False = fast mode
        self._precise_atmosphere = bool(precise_atmosphere)  # atmo_set_precision 2: the v2 atmosphere march of a no-cloud variant in reference order
        self._double_precision = bool(double_precision)  # `#define DOUBLE_PRECISION` (main:25): engine negates INV_VIEW origin
        self._lane_split = int(lane_split)  # atmo_set_lane_split: 0 auto, 1 / 2 lanes per ray
        self._tile_feedback = int(tile_feedback)  # atmo_set_tile_feedback: -1 default (on), 0 off, 1 on
        self._target_cleared = bool(target_cleared)  # atmo_set_target_cleared: discarded fragments write nothing (the shader's `discard`)
        # atmo_set_sampler_lod: None = as the shader declares the samplerCube (linear-mipmap: implicit LOD from the 2x2 pixel quad when a mip
        # chain is bound), True = the same, required (an error if the draw cannot use it), False = level 0 only
        self._cubemap_lod = None if cubemap_lod is None else bool(cubemap_lod)
        self._ctx = C.c_void_p()
        self._planet_radius = 1.0
        self._atmosphere_height = 0.1
        self._sun_path = None
        self._custom_shader = None
        self._shader = DefaultShader
        self.clouds_rotation_speed = 1.0  # degrees per second
        self.force_fullscreen = False
        self.global_transform = np.eye(4)
        self._mode = MODE_FAR
        self._uses_baked_optical_depth = False
        self._bake_pending = False
        self._params = {}  # the ShaderMaterial's parameter dictionary
        self._start_time = _time.monotonic()
        self._create_context()
        # defaults for the builtin shader (planet_atmosphere.gd:105-108)
        self.set_shader_parameter("u_sun_position", (5000.0, 0.0, 0.0))
        if blue_noise is not None:
            self.set_shader_parameter("u_blue_noise_texture", blue_noise)
        self.set_shader_parameter("u_clip_mode", 0.0)
        # _ready (planet_atmosphere.gd:111-115)
        self.set_shader_parameter("u_planet_radius", self._planet_radius)
        self.set_shader_parameter("u_atmosphere_height", self._atmosphere_height)
        self._sync_bake_flag()

    # ---- context management --------------------------------------------------------------------
    def _create_context(self):
        sh = self._shader
        vs = self._view_steps_override or sh.view_steps
        cs = (self._cloud_steps_override or sh.cloud_steps) if sh.cloud_steps else 0
        ctx = C.c_void_p()
        rc = self._lib.atmo_create(self._device, sh.variant, vs, cs, self._light_mode, self._light_steps, C.byref(ctx))
        N.check(None, rc)
        self._ctx = ctx
        N.check(ctx, self._lib.atmo_set_precision(ctx, 2 if self._precise_atmosphere else (1 if self._precise_clouds else 0)))
        N.check(ctx, self._lib.atmo_set_host_double_precision(ctx, 1 if self._double_precision else 0))
        N.check(ctx, self._lib.atmo_set_lane_split(ctx, self._lane_split))
        N.check(ctx, self._lib.atmo_set_tile_feedback(ctx, self._tile_feedback))
        if self._target_cleared:  # (an A/B library older than round 4 has no such entry point: only asked for when wanted)
            N.check(ctx, self._lib.atmo_set_target_cleared(ctx, 1))
        N.check(ctx, self._lib.atmo_set_sampler_lod(ctx, -1 if self._cubemap_lod is None else (1 if self._cubemap_lod else 0)))

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.atmo_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def kernel_name(self) -> str:
        return self._lib.atmo_kernel_name(self._ctx).decode()

    # ---- exported properties -------------------------------------------------------------------
    @property
    def planet_radius(self):
        return self._planet_radius

    @planet_radius.setter
    def planet_radius(self, v):
        self.set_planet_radius(v)

    @property
    def atmosphere_height(self):
        return self._atmosphere_height

    @atmosphere_height.setter
    def atmosphere_height(self, v):
        self.set_atmosphere_height(v)

    @property
    def sun_path(self):
        return self._sun_path

    @sun_path.setter
    def sun_path(self, v):
        self._sun_path = v

    @property
    def custom_shader(self):
        return self._custom_shader

    @custom_shader.setter
    def custom_shader(self, v):
        self.set_custom_shader(v)

    def set_planet_radius(self, new_radius: float):  # planet_atmosphere.gd:230-238
        if self._planet_radius == new_radius:
            return
        self._planet_radius = max(float(new_radius), 0.0)
        self.set_shader_parameter("u_planet_radius", self._planet_radius)
        if self._uses_baked_optical_depth:
            self._request_bake_optical_depth()

    def set_atmosphere_height(self, new_height: float):  # planet_atmosphere.gd:245-253
        if self._atmosphere_height == new_height:
            return
        self._atmosphere_height = max(float(new_height), 0.0)
        self.set_shader_parameter("u_atmosphere_height", self._atmosphere_height)
        if self._uses_baked_optical_depth:
            self._request_bake_optical_depth()

    def set_custom_shader(self, shader):  # planet_atmosphere.gd:118-141
        if isinstance(shader, str):
            shader = load_shader(shader)
        self._custom_shader = shader
        new = DefaultShader if shader is None else shader
        if new is not self._shader:
            self._shader = new
            self.close()
            self._create_context()
            for k, v in list(self._params.items()):  # the material keeps its parameters across shader changes
                if k == "u_optical_depth_texture" and isinstance(v, str):
                    self._bake_pending = True  # device-baked LUT: re-bake into the new context
                    continue
                self._forward(k, v)
        self._sync_bake_flag()

    def _sync_bake_flag(self):
        uses = any(u["name"] == "u_optical_depth_texture" for u in self._shader.get_shader_uniform_list())
        uses = uses and self._light_mode == N.LIGHT_LUT
        self._uses_baked_optical_depth = uses
        if uses:
            self._request_bake_optical_depth()

    # ---- shader parameters -----------------------------------------------------------------------
    def set_shader_param(self, param_name, value):  # planet_atmosphere.gd:164-166
        warnings.warn("set_shader_param is deprecated, use set_shader_parameter", DeprecationWarning, stacklevel=2)
        self.set_shader_parameter(param_name, value)

    def get_shader_param(self, param_name):  # planet_atmosphere.gd:170-172
        warnings.warn("get_shader_param is deprecated, use get_shader_parameter", DeprecationWarning, stacklevel=2)
        return self.get_shader_parameter(param_name)

    def set_shader_parameter(self, param_name: str, value):  # planet_atmosphere.gd:175-176
        self._params[param_name] = value
        self._forward(param_name, value)

    def get_shader_parameter(self, param_name: str):  # planet_atmosphere.gd:179-180
        return self._params.get(param_name)

    def _forward(self, name: str, value):
        """ShaderMaterial -> RenderingServer uniform upload.  Unknown names are ignored silently (Godot)."""
        if name in _TEXTURES:
            self._upload_texture(name, value)
            return
        n = _FLOAT_COUNTS.get(name)
        if n is None or value is None:
            return
        if isinstance(value, Transform2D):
            arr = value.as_mat2_col_major()
        elif isinstance(value, (bool, int, float, np.floating, np.integer)):
            arr = np.array([float(value)], dtype=np.float32)
        else:
            a = np.asarray(value, dtype=np.float64)
            if a.shape == (4, 4) or a.shape == (2, 2):
                arr = col_major(a)
            else:
                arr = a.reshape(-1).astype(np.float32)
        if arr.size != n:
            raise ValueError(f"{name} takes {n} floats, got {arr.size}")
        if name in _SOURCE_COLOR and not isinstance(value, LinearColor):
            a = np.asarray(arr, dtype=np.float64).copy()
            a[:3] = srgb_to_linear(a[:3])  # alpha (v1 colours) is not converted
            arr = a.astype(np.float32)
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        rc = self._lib.atmo_set_param_f32(self._ctx, name.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), n)
        N.check(self._ctx, rc)

    def _upload_texture(self, name: str, value):
        kind = _TEXTURES[name]
        if hasattr(value, "get_images"):  # a NoiseCubemap resource
            value = value.get_images()
        if value is None:
            rc = self._lib.atmo_set_texture(self._ctx, name.encode(), kind, 0, 0, 0, 0, None, N.MEM_HOST, None)
            N.check(self._ctx, rc)
            return
        mips = 1
        if name == "u_optical_depth_texture":
            a = np.ascontiguousarray(value, dtype=np.float32)
            h, w = a.shape
            d = 1
        elif name == "u_blue_noise_texture":
            a = np.ascontiguousarray(value, dtype=np.uint8)
            h, w = a.shape
            d = 1
        elif name == "u_cloud_shape_texture":
            a = np.ascontiguousarray(value, dtype=np.uint8)
            d, h, w = a.shape
        else:
            # a cubemap is (6, n, n) = level 0 (its mip chain is generated on the device, as Image.generate_mipmaps
            # does in noise_cubemap.gd:135), or a list of levels [(6, n, n), (6, n/2, n/2), ...] given explicitly
            if isinstance(value, (list, tuple)):
                levels = [np.ascontiguousarray(v, dtype=np.uint8) for v in value]
                d, h, w = levels[0].shape
                mips = len(levels)
                a = np.concatenate([lv.reshape(-1) for lv in levels])
            else:
                a = np.ascontiguousarray(value, dtype=np.uint8)
                d, h, w = a.shape
                mips = 0
        rc = self._lib.atmo_set_texture(self._ctx, name.encode(), kind, w, h, d, mips, a.ctypes.data_as(C.c_void_p), N.MEM_HOST, None)
        N.check(self._ctx, rc)

    # Object.get / Object.set with the "shader_params/<name>" convention (planet_atmosphere.gd:200-218)
    def get(self, key: str):
        if key.startswith("shader_params/"):
            param_name = key[len("shader_params/"):]
            value = self.get_shader_parameter(param_name)
            if value is None:
                value = SHADER_DEFAULTS.get(param_name)
            return value
        return getattr(self, key, None)

    def set(self, key: str, value):
        if key.startswith("shader_params/"):
            param_name = key[len("shader_params/"):]
            self.set_shader_parameter(param_name, value)
            if self._uses_baked_optical_depth and param_name in self._shader_params_affecting_optical_depth:
                self._request_bake_optical_depth()
            return
        setattr(self, key, value)

    def get_property_list(self):  # planet_atmosphere.gd:185-197
        props = []
        for p in self._shader.get_shader_uniform_list():
            if p["name"] in self._api_shader_params:
                continue
            props.append({"name": "shader_params/" + p["name"]})
        return props

    def get_configuration_warnings(self):  # planet_atmosphere.gd:221-227
        if self._sun_path is None:
            return ["The path to the sun is not assigned."]
        return []

    # ---- optical depth bake ----------------------------------------------------------------------
    def _request_bake_optical_depth(self):
        """planet_atmosphere.gd:144-150.  The reference defers the bake by two frames through a SubViewport
        (optical_depth_baker.gd:67-85); here it is one kernel enqueued before the next draw."""
        self._bake_pending = True

    def _bake_if_needed(self, stream=None):
        if self._bake_pending and self._uses_baked_optical_depth:
            rc = self._lib.atmo_bake_optical_depth(self._ctx, C.c_void_p(stream or 0))
            N.check(self._ctx, rc)
            self._params["u_optical_depth_texture"] = "<baked on device>"
        self._bake_pending = False

    def read_optical_depth(self, with_rgba8: bool = False, stream=None):
        """The baked LUT as the reference's baker would hand it to ImageTexture (FORMAT_RF), optionally with the
        RGBA8 packing of optical_depth.gdshader:33-43."""
        self._bake_if_needed(stream)
        w, h = C.c_int(0), C.c_int(0)
        N.check(self._ctx, self._lib.atmo_get_texture_size(self._ctx, b"u_optical_depth_texture", C.byref(w), C.byref(h), None, None))
        lut = np.empty((h.value, w.value), dtype=np.float32)
        rgba8 = np.empty((h.value, w.value, 4), dtype=np.uint8) if with_rgba8 else None
        rc = self._lib.atmo_read_optical_depth(
            self._ctx, lut.ctypes.data_as(C.c_void_p),
            rgba8.ctypes.data_as(C.c_void_p) if with_rgba8 else None, w.value * h.value, C.c_void_p(stream or 0))
        N.check(self._ctx, rc)
        return (lut, rgba8) if with_rgba8 else lut

    # ---- per frame -------------------------------------------------------------------------------
    def _sun_position(self):
        s = self._sun_path
        if s is None:
            return None
        if hasattr(s, "global_position"):
            return tuple(float(x) for x in s.global_position)
        return tuple(float(x) for x in s)

    def _set_mode(self, mode: int):  # planet_atmosphere.gd:261-282 (mesh swap is rasteriser-only)
        if mode == self._mode:
            return
        self._mode = mode
        self.set_shader_parameter("u_clip_mode", 1.0 if mode == MODE_NEAR else 0.0)

    def _process(self, delta: float = 0.0, camera=None, time: float | None = None):
        """planet_atmosphere.gd:285-341: near/far switch and the per-frame uniforms."""
        cam_pos = np.zeros(3)
        cam_near = 0.1
        if camera is not None:
            cam_pos = np.asarray(camera.inv_view)[:3, 3]
            cam_near = camera.near
        atmo_clip_distance = 1.75 * (self._planet_radius + self._atmosphere_height + cam_near) * SWITCH_MARGIN_RATIO
        d = float(np.linalg.norm(np.asarray(self.global_transform)[:3, 3] - cam_pos))
        self._set_mode(MODE_NEAR if (d < atmo_clip_distance or self.force_fullscreen) else MODE_FAR)

        sun = self._sun_position()
        if sun is not None:
            self.set_shader_parameter("u_sun_position", sun)
        # planet_atmosphere.gd:335 calls Transform3D.inverse(), which assumes an orthonormal basis (transpose + rotated origin); the general
        # inverse used here equals it for every rigid transform and stays a true inverse when the node is scaled (what affine_inverse() gives)
        self.set_shader_parameter("u_world_to_model_matrix", np.linalg.inv(np.asarray(self.global_transform, dtype=np.float64)))
        if time is None:
            time = _time.monotonic() - self._start_time
        self.set_shader_parameter("u_cloud_coverage_rotation",
                                  Transform2D().rotated(time * math.radians(self.clouds_rotation_speed)))

    def make_frame(self, camera, time: float = 0.0, rect=None) -> dict:
        sun = self.get_shader_parameter("u_sun_position")
        if sun is None:
            sun = (0.0, 0.0, 0.0)
        return make_frame(camera, self.global_transform, sun, time, rect)

    def render(self, camera, depth, out=None, rect=None, stream=None, time: float = 0.0):
        """One draw: shades `rect` (default: whole viewport) of the camera's viewport.

        depth: CUDA float32 tensor (H, W), Godot reversed-Z depth.  out: CUDA float32 tensor
        (rect_h, rect_w, 4), allocated when None.  Work is enqueued on `stream` (a torch stream, a raw
        hipStream_t int, or None for torch's current stream).  Returns `out`."""
        import torch

        frame = self.make_frame(camera, time, rect)
        x0, y0, x1, y1 = frame["rect"]
        if not (isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous()):
            raise TypeError("depth must be a contiguous CUDA float32 tensor")
        if tuple(depth.shape) != (camera.height, camera.width):
            raise ValueError("depth must have shape (viewport_h, viewport_w)")
        if out is None:
            out = torch.empty((y1 - y0, x1 - x0, 4), dtype=torch.float32, device=depth.device)
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (y1 - y0, x1 - x0, 4)):
            raise ValueError("out must be a contiguous CUDA float32 tensor of shape (rect_h, rect_w, 4)")
        if stream is None:
            stream = torch.cuda.current_stream(depth.device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self.render_raw(frame, depth.data_ptr(), out.data_ptr(), stream)
        return out

    def prepare_frame(self, camera, time: float = 0.0, rect=None) -> N.AtmoFrame:
        """The native per-frame argument block for `render_prepared` (build once per camera pose)."""
        return _to_native_frame(self.make_frame(camera, time, rect))

    def render_prepared(self, native_frame: N.AtmoFrame, depth_ptr: int, out_ptr: int, stream: int = 0):
        """Enqueue one draw with a frame from `prepare_frame` on raw device addresses: the per-step host cost is
        one ctypes call (what a render loop that re-draws an unchanged camera would do)."""
        self._bake_if_needed(stream)
        rc = self._lib.atmo_render(self._ctx, C.byref(native_frame), C.c_void_p(depth_ptr), C.c_void_p(out_ptr),
                                   C.c_void_p(stream or 0))
        N.check(self._ctx, rc)

    def render_composite(self, camera, depth, scene_rgba, rect=None, stream=None, time: float = 0.0):
        """The draw including the renderer's blend stage: shades `rect` and alpha-blends the result over
        `scene_rgba` (CUDA float32 (H, W, 4), the scene colour buffer) in place, as Godot's blend_mix does with
        ALBEDO/ALPHA; discarded fragments leave the scene untouched.  Returns `scene_rgba`."""
        import torch

        frame = self.make_frame(camera, time, rect)
        if not (isinstance(scene_rgba, torch.Tensor) and scene_rgba.is_cuda and scene_rgba.dtype == torch.float32
                and scene_rgba.is_contiguous() and tuple(scene_rgba.shape) == (camera.height, camera.width, 4)):
            raise ValueError("scene_rgba must be a contiguous CUDA float32 tensor of shape (viewport_h, viewport_w, 4)")
        if not (isinstance(depth, torch.Tensor) and depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous()
                and tuple(depth.shape) == (camera.height, camera.width)):
            raise TypeError("depth must be a contiguous CUDA float32 tensor of shape (viewport_h, viewport_w)")
        if stream is None:
            stream = torch.cuda.current_stream(depth.device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self._bake_if_needed(stream)
        nf = _to_native_frame(frame)
        rc = self._lib.atmo_render_composite(self._ctx, C.byref(nf), C.c_void_p(depth.data_ptr()),
                                             C.c_void_p(scene_rgba.data_ptr()), C.c_void_p(stream or 0))
        N.check(self._ctx, rc)
        return scene_rgba

    def render_raw(self, frame: dict, depth_ptr: int, out_ptr: int, stream: int = 0):
        """`render` on raw device addresses (what a non-torch host would call)."""
        self._bake_if_needed(stream)
        nf = _to_native_frame(frame)
        rc = self._lib.atmo_render(self._ctx, C.byref(nf), C.c_void_p(depth_ptr), C.c_void_p(out_ptr), C.c_void_p(stream or 0))
        N.check(self._ctx, rc)

    def render_tiles_prepared(self, native_frame: N.AtmoFrame, depth_ptr: int, out_ptr: int, tiles_ptr: int, n_tiles: int, stream: int = 0,
                              n_heavy: int = 0):
        """atmo_render_tiles: draw only the listed tiles (uint32 indices in device memory, row-major in the grid of
        measure_tile_costs) of the frame's rect, in list order; `out_ptr` is addressed like render_prepared's.
        n_heavy > 0 (atmo_render_tiles_split): the list's first n_heavy tiles on two lanes per ray beside the rest (sharding.heavy_tiles)."""
        self._bake_if_needed(stream)
        if n_heavy:
            rc = self._lib.atmo_render_tiles_split(self._ctx, C.byref(native_frame), C.c_void_p(depth_ptr), C.c_void_p(out_ptr), C.c_void_p(tiles_ptr),
                                                   int(n_tiles), int(n_heavy), C.c_void_p(stream or 0))
        else:
            rc = self._lib.atmo_render_tiles(self._ctx, C.byref(native_frame), C.c_void_p(depth_ptr), C.c_void_p(out_ptr), C.c_void_p(tiles_ptr),
                                             int(n_tiles), C.c_void_p(stream or 0))
        N.check(self._ctx, rc)

    def measure_tile_costs(self, camera, depth, rect=None, stream=None, time: float = 0.0):
        """One draw through atmo_measure_tile_costs: ((tiles_y, tiles_x) uint32 costs -- every tile's longest wavefront in shader cycles --,
        tile_w, tile_h).  sharding.lpt_strips deals them to the GPUs of a node."""
        rows = self.measure_row_costs(camera, depth, rect=rect, stream=stream, time=time)
        del rows
        return self._last_tile_costs, self._last_tile_size[0], self._last_tile_size[1]

    def measure_row_costs(self, camera, depth, rect=None, stream=None, time: float = 0.0):
        """Measured cost of every pixel row of `rect` (default: the whole viewport): one draw through atmo_measure_tile_costs,
        each tile's cost (longest wavefront, shader cycles) spread over its pixel rows and summed along the row.  Feed it to
        sharding.balanced_row_bands to cut a viewport into row bands of equal WORK for several GPUs."""
        import torch

        frame = self.make_frame(camera, time, rect)
        x0, y0, x1, y1 = frame["rect"]
        nf = _to_native_frame(frame)
        if stream is None:
            stream = torch.cuda.current_stream(depth.device).cuda_stream
        elif hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        self._bake_if_needed(stream)
        scratch = torch.empty((y1 - y0, x1 - x0, 4), dtype=torch.float32, device=depth.device)
        gx, gy, tw, th = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        args = (self._ctx, C.byref(nf), C.c_void_p(depth.data_ptr()), C.c_void_p(scratch.data_ptr()), C.c_void_p(stream or 0))
        N.check(self._ctx, self._lib.atmo_measure_tile_costs(*args, None, 0, C.byref(gx), C.byref(gy), C.byref(tw), C.byref(th)))
        cost = np.zeros((gy.value, gx.value), dtype=np.uint32)
        N.check(self._ctx, self._lib.atmo_measure_tile_costs(*args, cost.ctypes.data_as(C.c_void_p), cost.size, C.byref(gx), C.byref(gy),
                                                            C.byref(tw), C.byref(th)))
        self._last_tile_costs = cost  # (tiles_y, tiles_x) uint32, for diagnostics (tools/xcd_balance.py)
        self._last_tile_size = (tw.value, th.value)
        rows = np.repeat(cost.astype(np.float64).sum(axis=1) / th.value, th.value)[: y1 - y0]
        return rows

    def split_stats(self) -> dict:
        """atmo_get_split_stats: draws so far whose heaviest tiles were drawn with two lanes per ray beside the rest, and how many tiles the last
        such draw split (the raymarched-light kernel under the declared sampler, when a draw is as long as its heaviest wavefront)."""
        n, last = C.c_uint(0), C.c_uint(0)
        N.check(self._ctx, self._lib.atmo_get_split_stats(self._ctx, C.byref(n), C.byref(last)))
        return {"split_draws": int(n.value), "heavy_tiles_last": int(last.value)}

    def feedback_stats(self) -> dict:
        """atmo_get_feedback_stats (diagnostics of the tile-order feedback)."""
        st, od, so, rc = C.c_int(0), C.c_uint(0), C.c_uint(0), C.c_uint(0)
        N.check(self._ctx, self._lib.atmo_get_feedback_stats(self._ctx, C.byref(st), C.byref(od), C.byref(so), C.byref(rc)))
        return dict(states=st.value, ordered_draws=od.value, sorts=so.value, recycled=rc.value)

    # ---- kernel timing (HIP events on the launch stream) ---------------------------------------------
    def set_timing(self, enable, every: int = 1):
        """enable: False/0 off; True = bracket every `every`-th launch with HIP events."""
        k = 0 if not enable else max(1, int(every))
        N.check(self._ctx, self._lib.atmo_set_timing(self._ctx, k))

    def get_timing(self):
        n, ms = C.c_int(0), C.c_double(0.0)
        N.check(self._ctx, self._lib.atmo_get_timing(self._ctx, C.byref(n), C.byref(ms)))
        return n.value, ms.value
