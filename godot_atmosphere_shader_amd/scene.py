"""Synthetic scene inputs for the atmosphere raymarch: camera/projection matrices, depth buffers,
the jitter table, the 3-D cloud-shape tile and the coverage cubemap.

Everything is generated from integer hashes in numpy (no RNG state, no transcendental functions in
the texel values), so the same bytes come out on every machine; the golden fixtures under
tests/golden/ store a checksum of each generated texture to detect drift.

Conventions ("engine-behaviour assumptions"): matrices are 4x4 numpy arrays in
mathematical row/column layout (M[row, col]); `col_major()` flattens to the GLSL/Godot column-major
order the C ABI takes.  `inv_projection` maps (SCREEN_UV*2-1, depth, 1) to view space: SCREEN_UV
origin top-left, depth reversed (1 = near plane, 0 = far plane), view space looks down -Z with +Y up.

Demo-scene values come from the reference's demo (addons/zylann.atmosphere/demo/
planet_atmosphere_test.tscn:96-114, demo/flying_avatar.tscn:10-13).
"""
from __future__ import annotations

import math
import zlib

import numpy as np

# --------------------------------------------------------------------------------------------
# matrices
# --------------------------------------------------------------------------------------------


def col_major(m: np.ndarray) -> np.ndarray:
    """4x4 (or 2x2) M[row, col] -> flat float32 column-major (GLSL memory order)."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.reshape(-1).astype(np.float32))


def perspective(fovy_deg: float, aspect: float, near: float, far: float, reverse_z: bool = True) -> np.ndarray:
    """View -> clip, Vulkan-style as Godot 4.3 hands it to shaders: y down, reversed z in [0,1].
    reverse_z=False: Godot 4.2 and older (the reference's `REVERSE_Z` define commented out, main:21-22):
    forward z in [0,1], 0 = near plane, 1 = far plane."""
    f = 1.0 / math.tan(math.radians(fovy_deg) * 0.5)
    gl = np.array(
        [
            [f / aspect, 0, 0, 0],
            [0, f, 0, 0],
            [0, 0, (far + near) / (near - far), 2.0 * far * near / (near - far)],
            [0, 0, -1, 0],
        ],
        dtype=np.float64,
    )
    # y flip, z_gl in [-1,1] -> reversed [1,0] (or forward [0,1])
    zs = -0.5 if reverse_z else 0.5
    fix = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, zs, 0.5], [0, 0, 0, 1]], dtype=np.float64)
    return fix @ gl


def look_at(eye, target, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """Camera global transform (= INV_VIEW_MATRIX): -Z looks at target."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    upv = np.asarray(up, dtype=np.float64)
    right = np.cross(fwd, upv)
    if np.linalg.norm(right) < 1e-9:
        right = np.cross(fwd, np.array([1.0, 0.0, 0.0]))
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    m = np.eye(4)
    m[:3, 0] = right
    m[:3, 1] = true_up
    m[:3, 2] = -fwd
    m[:3, 3] = eye
    return m


def srgb_to_linear(c):
    """Godot's `source_color` conversion applied to colour uniforms before upload."""
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


# --------------------------------------------------------------------------------------------
# integer-hash noise
# --------------------------------------------------------------------------------------------


def _hash_u32(x: np.ndarray) -> np.ndarray:
    """lowbias32-style avalanche on uint32 arrays (wrapping arithmetic)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iy, iz, seed: int) -> np.ndarray:
    """Lattice value in [0,1) from integer coordinates."""
    with np.errstate(over="ignore"):
        h = (ix.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (iy.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ (
            iz.astype(np.uint32) * np.uint32(0xC2B2AE3D)
        ) ^ np.uint32(seed & 0xFFFFFFFF)
        h = _hash_u32(h)
    return (h >> np.uint32(8)).astype(np.float64) / float(1 << 24)


def value_noise3(p: np.ndarray, seed: int, period: int | None = None) -> np.ndarray:
    """Smooth value noise at points p[..., 3]; lattice wraps with `period` when given."""
    p = np.asarray(p, dtype=np.float64)
    fl = np.floor(p)
    f = p - fl
    w = f * f * (3.0 - 2.0 * f)
    i0 = fl.astype(np.int64)
    i1 = i0 + 1
    if period is not None:
        i0 = np.mod(i0, period)
        i1 = np.mod(i1, period)
    x0, y0, z0 = i0[..., 0], i0[..., 1], i0[..., 2]
    x1, y1, z1 = i1[..., 0], i1[..., 1], i1[..., 2]
    wx, wy, wz = w[..., 0], w[..., 1], w[..., 2]

    def L(a, b, c):
        return _lattice(a, b, c, seed)

    c00 = L(x0, y0, z0) * (1 - wx) + L(x1, y0, z0) * wx
    c10 = L(x0, y1, z0) * (1 - wx) + L(x1, y1, z0) * wx
    c01 = L(x0, y0, z1) * (1 - wx) + L(x1, y0, z1) * wx
    c11 = L(x0, y1, z1) * (1 - wx) + L(x1, y1, z1) * wx
    c0 = c00 * (1 - wy) + c10 * wy
    c1 = c01 * (1 - wy) + c11 * wy
    return c0 * (1 - wz) + c1 * wz


def fbm3(p: np.ndarray, seed: int, octaves: int, base_period: int | None = None, gain: float = 0.5) -> np.ndarray:
    total = np.zeros(np.asarray(p).shape[:-1], dtype=np.float64)
    amp, norm, freq = 1.0, 0.0, 1
    for o in range(octaves):
        per = None if base_period is None else base_period * freq
        total += amp * value_noise3(np.asarray(p, dtype=np.float64) * freq, seed + 1013 * o, per)
        norm += amp
        amp *= gain
        freq *= 2
    return total / norm


def _to_u8(v: np.ndarray, contrast: float = 1.0) -> np.ndarray:
    v = (v - 0.5) * contrast + 0.5
    return np.clip(np.floor(v * 255.0 + 0.5), 0, 255).astype(np.uint8)


def make_blue_noise(seed: int = 1) -> np.ndarray:
    """256x256 R8 jitter table with a flat histogram (every level 256 times), like the reference's
    blue_noise.png (256x256, all 256 levels, mean 127.5).  White rather than blue spectrum: the
    spectrum only matters visually."""
    idx = np.arange(65536, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = _hash_u32(idx * np.uint32(0x9E3779B1) ^ np.uint32(seed))
    order = np.argsort(h, kind="stable")
    out = np.empty(65536, dtype=np.uint8)
    out[order] = (np.arange(65536) >> 8).astype(np.uint8)
    return out.reshape(256, 256)


def make_shape_texture(n: int = 64, seed: int = 7, cells: int = 8, octaves: int = 3) -> np.ndarray:
    """n^3 R8 tileable fBm value noise, indexed [z, y, x] (stand-in for the demo's seamless
    NoiseTexture3D, planet_atmosphere_test.tscn:55-57)."""
    g = (np.arange(n, dtype=np.float64) + 0.5) / n * cells
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    p = np.stack([x, y, z], axis=-1)
    return _to_u8(fbm3(p, seed, octaves, base_period=cells), contrast=1.6)


_FACE_AXES = (
    # Vulkan cube face table: direction = major*ma + sc*S + tc*T
    ((1, 0, 0), (0, 0, -1), (0, -1, 0)),   # +X: sc=-z tc=-y
    ((-1, 0, 0), (0, 0, 1), (0, -1, 0)),   # -X: sc=+z tc=-y
    ((0, 1, 0), (1, 0, 0), (0, 0, 1)),     # +Y: sc=+x tc=+z
    ((0, -1, 0), (1, 0, 0), (0, 0, -1)),   # -Y: sc=+x tc=-z
    ((0, 0, 1), (1, 0, 0), (0, -1, 0)),    # +Z: sc=+x tc=-y
    ((0, 0, -1), (-1, 0, 0), (0, -1, 0)),  # -Z: sc=-x tc=-y
)


def cube_texel_directions(n: int) -> np.ndarray:
    """Unit direction of every texel centre, shape (6, n, n, 3), faces +X,-X,+Y,-Y,+Z,-Z.
    Same texel->direction mapping as the reference's NoiseCubemap generator
    (addons/zylann.atmosphere/noise_cubemap.gd:110-128), which follows the Vulkan face table."""
    c = (np.arange(n, dtype=np.float64) + 0.5) / n * 2.0 - 1.0
    tc, sc = np.meshgrid(c, c, indexing="ij")  # row = t, column = s
    out = np.empty((6, n, n, 3), dtype=np.float64)
    for f, (major, s_ax, t_ax) in enumerate(_FACE_AXES):
        d = (np.asarray(major, dtype=np.float64)[None, None, :]
             + sc[..., None] * np.asarray(s_ax, dtype=np.float64)
             + tc[..., None] * np.asarray(t_ax, dtype=np.float64))
        out[f] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    return out


def make_coverage_cubemap(n: int = 256, seed: int = 11, scale=(3.0, 6.0, 3.0), octaves: int = 4) -> np.ndarray:
    """6 x n x n R8 coverage: density = noise(dir * scale) per texel (NoiseCubemap analogue;
    seeded value noise, NOT FastNoiseLite)."""
    d = cube_texel_directions(n) * np.asarray(scale, dtype=np.float64)
    return _to_u8(fbm3(d + 100.0, seed, octaves), contrast=2.2)


def checksum(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------
# demo scene
# --------------------------------------------------------------------------------------------

DEMO_PLANET_RADIUS = 100.0
DEMO_ATMOSPHERE_HEIGHT = 8.0
DEMO_SUN_POSITION = (0.0, 0.0, 478.677)  # Sun node z=598.677 + DirectionalLight child z=-120
DEMO_CAMERA = dict(fovy_deg=75.0, near=0.1, far=800.0)

# shader_params of the demo's PlanetAtmosphere node; colours already converted sRGB -> linear.
DEMO_SHADER_PARAMS = {
    "u_density": 0.5,
    "u_scattering_strength": 1.0,
    "u_scattering_wavelengths": (700.0, 530.0, 440.0),
    "u_atmosphere_modulate": tuple(srgb_to_linear((1.0, 0.980392, 0.964706)).tolist()),
    "u_atmosphere_ambient_color": tuple(srgb_to_linear((0.0196078, 0.0196078, 0.0431373)).tolist()),
    "u_sphere_depth_factor": 0.0,
    "u_cloud_density_scale": 2.0,
    "u_cloud_bottom": 0.2,
    "u_cloud_top": 0.6,
    "u_cloud_blend": 0.5,
    "u_cloud_shape_invert": 1.0,
    "u_cloud_coverage_bias": 0.0,
    "u_cloud_shape_factor": 0.5,
    "u_cloud_shape_scale": 0.1,
}

POSES = {
    # camera of the demo scene: Avatar (0,0,156.425) + Camera child (0.357289, 0.105603, 1.49554), looking -Z
    "P_space": dict(eye=(0.357289, 0.105603, 157.92054), target=(0.357289, 0.105603, 0.0)),
    # 1 unit above the ground, looking along the horizon towards the sun-lit side
    "P_ground": dict(eye=(0.0, 101.0, 0.0), target=(0.0, 101.0, 100.0), up=(0.0, 1.0, 0.0)),
    # grazing view of the limb from low orbit
    "P_limb": dict(eye=(0.0, 112.0, 30.0), target=(0.0, 60.0, 130.0)),
    # inside the cloud layer
    "P_clouds": dict(eye=(0.0, 103.0, 5.0), target=(30.0, 104.0, 60.0)),
    # night side, sun behind the planet
    "P_night": dict(eye=(20.0, 10.0, -170.0), target=(0.0, 0.0, 0.0)),
}


def orbit_pose(k: int, n: int, radius: float = 158.0) -> dict:
    """k-th of n camera poses on an orbit around the planet (config 4: one viewport per GPU)."""
    a = 2.0 * math.pi * k / max(n, 1)
    eye = (radius * math.sin(a), 12.0 * math.sin(2.0 * a + 0.3), radius * math.cos(a))
    return dict(eye=eye, target=(0.0, 0.0, 0.0))


class Camera:
    def __init__(self, width: int, height: int, eye, target, up=(0.0, 1.0, 0.0), fovy_deg=75.0, near=0.1, far=800.0,
                 reverse_z: bool = True):
        self.width, self.height = int(width), int(height)
        self.near, self.far, self.fovy_deg = near, far, fovy_deg
        self.reverse_z = bool(reverse_z)
        self.projection = perspective(fovy_deg, width / height, near, far, reverse_z)
        self.inv_projection = np.linalg.inv(self.projection)
        self.inv_view = look_at(eye, target, up)
        self.view = np.linalg.inv(self.inv_view)

    @classmethod
    def from_pose(cls, width, height, pose, **kw):
        if isinstance(pose, str):
            pose = POSES[pose]
        args = dict(DEMO_CAMERA)
        args.update(kw)
        return cls(width, height, pose["eye"], pose["target"], pose.get("up", (0.0, 1.0, 0.0)), **args)

    def pixel_view_dirs(self) -> np.ndarray:
        """Unnormalised view-space ray direction per pixel centre, (H, W, 3), float64."""
        xs = (np.arange(self.width) + 0.5) / self.width * 2.0 - 1.0
        ys = (np.arange(self.height) + 0.5) / self.height * 2.0 - 1.0
        gx, gy = np.meshgrid(xs, ys)
        near_z = np.ones_like(gx) if self.reverse_z else np.zeros_like(gx)  # a point on the near plane
        ndc = np.stack([gx, gy, near_z, np.ones_like(gx)], axis=-1)
        v = ndc @ self.inv_projection.T
        return v[..., :3] / v[..., 3:4]


def depth_far(cam: Camera) -> np.ndarray:
    """Empty scene: the far plane everywhere (reversed-Z depth 0; forward-Z depth 1)."""
    return np.full((cam.height, cam.width), 0.0 if cam.reverse_z else 1.0, dtype=np.float32)


def depth_ground_sphere(cam: Camera, center_world=(0.0, 0.0, 0.0), radius: float = DEMO_PLANET_RADIUS) -> np.ndarray:
    """Depth buffer of an opaque sphere (the demo's Ground mesh): nonlinear depth in the camera's convention, the far
    plane (reversed-Z 0 / forward-Z 1) where missed."""
    d = cam.pixel_view_dirs()
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    c = (cam.view @ np.array([*center_world, 1.0]))[:3]
    b = -(d @ c)  # dot(oc, d), oc = -c
    qc2 = c @ c - b * b
    h = radius * radius - qc2
    hit = h >= 0.0
    t = -b - np.sqrt(np.where(hit, h, 0.0))
    hit &= t > cam.near
    z_view = d[..., 2] * t  # negative in front of the camera
    p = cam.projection
    zc = p[2, 2] * z_view + p[2, 3]
    wc = p[3, 2] * z_view + p[3, 3]
    depth = np.where(hit, zc / np.where(hit, wc, 1.0), 0.0 if cam.reverse_z else 1.0)
    return depth.astype(np.float32)
