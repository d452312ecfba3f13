"""ctypes binding of the C ABI in include/atmo.h (libatmo_hip.so).

There is no CPU fallback: if the library is missing or no gfx950 device is present, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

ATMO_OK, ATMO_E_NAME, ATMO_E_ARG, ATMO_E_STATE, ATMO_E_HIP, ATMO_E_NO_DEVICE = range(6)
VARIANT_NO_CLOUDS, VARIANT_CLOUDS, VARIANT_CLOUDS_HIGH, VARIANT_CLOUDS_HIGH_RM = range(4)
VARIANT_V1_NO_CLOUDS, VARIANT_V1_CLOUDS, VARIANT_V1_CLOUDS_HIGH = 4, 5, 6
LIGHT_LUT, LIGHT_DIRECT = 0, 1
TEX_2D_R32F, TEX_2D_R8, TEX_3D_R8, TEX_CUBE_R8 = range(4)
MEM_HOST, MEM_DEVICE = 0, 1
ABI_VERSION = 4

# every symbol include/atmo.h declares: the surface a host binds (each replaces a reference interface)
CORE_SYMBOLS = (
    "atmo_abi_version", "atmo_device_count", "atmo_create", "atmo_destroy", "atmo_set_param_f32", "atmo_get_param_f32",
    "atmo_set_texture", "atmo_get_texture_size", "atmo_set_sampler_lod", "atmo_bake_optical_depth",
    "atmo_generate_noise_cubemap", "atmo_read_optical_depth", "atmo_render", "atmo_render_composite", "atmo_render_tiles", "atmo_render_tiles_split", "atmo_measure_tile_costs", "atmo_set_precision",
    "atmo_set_host_double_precision", "atmo_set_target_cleared", "atmo_set_tile_feedback", "atmo_last_error_string",
)
# every symbol include/atmo_debug.h declares: experiment knobs and diagnostics (tests, bench.py, tools/)
DEBUG_SYMBOLS = (
    "atmo_set_lane_split", "atmo_debug_motion_px", "atmo_get_feedback_stats", "atmo_set_timing", "atmo_get_timing", "atmo_host_layout_cubemap", "atmo_host_layout_shape",
    "atmo_host_layout_lut", "atmo_host_cubemap_mip", "atmo_read_texture_layout", "atmo_selftest_exact_math", "atmo_debug_marched_optical_depth", "atmo_debug_log2_cr", "atmo_kernel_name", "atmo_build_id",
    "atmo_get_host_wait_stats", "atmo_get_split_stats", "atmo_debug_create_host_only", "atmo_debug_frame_constants",
)
EXPORTED_SYMBOLS = CORE_SYMBOLS + DEBUG_SYMBOLS


class AtmoFrame(C.Structure):
    _fields_ = [
        ("inv_projection_matrix", C.c_float * 16),
        ("inv_view_matrix", C.c_float * 16),
        ("viewport_w", C.c_int32),
        ("viewport_h", C.c_int32),
        ("planet_center_viewspace", C.c_float * 3),
        ("sun_center_viewspace", C.c_float * 3),
        ("time", C.c_float),
        ("x0", C.c_int32),
        ("y0", C.c_int32),
        ("x1", C.c_int32),
        ("y1", C.c_int32),
    ]


class AtmoError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libatmo_hip error {code}: {message}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libatmo_hip.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the gfx950 HIP extension has not been built "
            "(python -m godot_atmosphere_shader_amd.build). There is no CPU fallback.")
    # PyTorch ships its own HIP runtime; whichever libamdhip64 is loaded first serves the whole process, and
    # loading /opt/rocm's copy (through this library) before torch's leaves the later-initialised side without a
    # device.  torch is this package's plumbing layer, so let it load its runtime first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, cp, ip, fp = C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float)
    sig = {
        "atmo_abi_version": (ip, []),
        "atmo_device_count": (ip, []),
        "atmo_create": (ip, [ip, ip, ip, ip, ip, ip, C.POINTER(vp)]),
        "atmo_destroy": (ip, [vp]),
        "atmo_set_param_f32": (ip, [vp, cp, fp, ip]),
        "atmo_get_param_f32": (ip, [vp, cp, fp, ip]),
        "atmo_set_texture": (ip, [vp, cp, ip, ip, ip, ip, ip, vp, ip, vp]),
        "atmo_get_texture_size": (ip, [vp, cp, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]),
        "atmo_set_sampler_lod": (ip, [vp, ip]),
        "atmo_read_texture_layout": (ip, [vp, cp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]),
        "atmo_host_cubemap_mip": (ip, [vp, ip, vp]),
        "atmo_generate_noise_cubemap": (ip, [vp, ip, C.c_uint32, C.c_float, ip, C.c_float, fp, ip, vp, C.POINTER(C.c_double)]),
        "atmo_bake_optical_depth": (ip, [vp, vp]),
        "atmo_read_optical_depth": (ip, [vp, vp, vp, ip, vp]),
        "atmo_render": (ip, [vp, C.POINTER(AtmoFrame), vp, vp, vp]),
        "atmo_render_composite": (ip, [vp, C.POINTER(AtmoFrame), vp, vp, vp]),
        "atmo_render_tiles": (ip, [vp, C.POINTER(AtmoFrame), vp, vp, vp, ip, vp]),
        "atmo_measure_tile_costs": (ip, [vp, C.POINTER(AtmoFrame), vp, vp, vp, vp, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]),
        "atmo_set_precision": (ip, [vp, ip]),
        "atmo_set_host_double_precision": (ip, [vp, ip]),
        "atmo_set_lane_split": (ip, [vp, ip]),
        "atmo_set_tile_feedback": (ip, [vp, ip]),
        "atmo_set_target_cleared": (ip, [vp, ip]),
        "atmo_debug_motion_px": (C.c_float, [C.POINTER(AtmoFrame), C.POINTER(AtmoFrame), C.c_float, ip]),
        "atmo_get_feedback_stats": (ip, [vp, C.POINTER(ip), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
        "atmo_set_timing": (ip, [vp, ip]),
        "atmo_get_timing": (ip, [vp, C.POINTER(ip), C.POINTER(C.c_double)]),
        "atmo_selftest_exact_math": (ip, [vp, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "atmo_debug_marched_optical_depth": (ip, [vp, ip, vp, vp, ip, vp]),
        "atmo_debug_log2_cr": (ip, [vp, ip, vp, vp]),
        "atmo_host_layout_cubemap": (ip, [vp, ip, vp]),
        "atmo_host_layout_shape": (ip, [vp, ip, vp]),
        "atmo_host_layout_lut": (ip, [vp, ip, ip, vp]),
        "atmo_kernel_name": (cp, [vp]),
        "atmo_build_id": (cp, []),
        "atmo_get_host_wait_stats": (ip, [vp, C.POINTER(C.c_uint)]),
        "atmo_get_split_stats": (ip, [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
        "atmo_render_tiles_split": (ip, [vp, C.POINTER(AtmoFrame), vp, vp, vp, ip, ip, vp]),
        "atmo_debug_create_host_only": (ip, [ip, ip, ip, ip, ip, C.POINTER(vp)]),
        "atmo_debug_frame_constants": (ip, [vp, C.POINTER(AtmoFrame), ip, C.POINTER(C.c_float), ip, C.POINTER(ip)]),
        "atmo_last_error_string": (cp, [vp]),
    }
    # ATMO_HIP_LIB names an A/B build (tools/ab_build_commit.sh: possibly an OLDER commit's library): entry points it lacks are skipped
    # (callers of those guard with hasattr) and its ABI version is not held against it.  The in-tree library must match exactly.
    ab_build = bool(os.environ.get("ATMO_HIP_LIB"))
    for name, (res, args) in sig.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if ab_build:
                continue
            raise RuntimeError(f"libatmo_hip.so does not export {name}; rebuild it")
        fn.restype = res
        fn.argtypes = args
    have = lib.atmo_abi_version()
    if have != ABI_VERSION:
        # An A/B library may be older, but only a version whose AtmoFrame layout and shared entry points are KNOWN to be what this binding
        # declares may be driven with it, and the caller has to name it: ATMO_HIP_LIB_ABI=3 (ABI 3 = round 3: the same AtmoFrame, no
        # atmo_set_target_cleared / atmo_render_tiles, atmo_set_sampler_lod without the -1 mode).  Anything else could corrupt memory silently.
        allowed = {3}
        named = os.environ.get("ATMO_HIP_LIB_ABI", "")
        if not (ab_build and named.isdigit() and int(named) == have and have in allowed):
            raise RuntimeError(f"libatmo_hip.so has ABI version {have}, this binding is written for {ABI_VERSION}; rebuild it"
                               + (f" (an older A/B library needs ATMO_HIP_LIB_ABI={have}; known-compatible: {sorted(allowed)})" if ab_build else ""))
    _lib = lib
    return lib


def check(ctx, code: int) -> None:
    if code != ATMO_OK:
        msg = load().atmo_last_error_string(ctx)
        raise AtmoError(code, msg.decode() if msg else "")
