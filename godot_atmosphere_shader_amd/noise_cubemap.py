"""Host-side mirror of the reference's `NoiseCubemap` resource (addons/zylann.atmosphere/noise_cubemap.gd):
a procedural cubemap projecting 3-D noise, used as `u_cloud_coverage_cubemap`.

  reference (noise_cubemap.gd)                 here
  -------------------------------------------  ------------------------------------------------------------
  noise : Noise (:9-22)                         `noise`: a `SeededValueNoise` (stands for the engine's FastNoiseLite)
  resolution, clampi(1, 4096) (:25-33)          `resolution`
  scale : Vector3 = (100,100,100) (:38-45)      `scale`
  _request_update / call_deferred (:61-64)      `_request_update()`; the deferred `_update` runs on `process_deferred()`
                                                or on first access to the images
  _update -> _generate_images (:67-81,101-140)  one kernel launch (`atmo_generate_noise_cubemap`), not a CPU triple loop
  generate_importable_image (:93-97,143-155)    same: 3 x 2 atlas of the six sides

The arithmetic of `Noise.get_noise_3dv` is Godot engine code (FastNoiseLite) and not part of the reference tree, so
the noise here is this package's own seeded fractal value noise; what is kept from the reference is the
texel -> direction mapping, the `scale`, the 0.5 + 0.5 * n remap and the L8 store.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

_M32 = np.uint32(0xFFFFFFFF)


def _hash_u32(x):
    x = x.astype(np.uint32, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iy, iz, seed):
    with np.errstate(over="ignore"):
        h = (ix.astype(np.int32).view(np.uint32) * np.uint32(0x9E3779B1)) ^ (iy.astype(np.int32).view(np.uint32) * np.uint32(0x85EBCA77)) \
            ^ (iz.astype(np.int32).view(np.uint32) * np.uint32(0xC2B2AE3D)) ^ np.uint32(seed & 0xFFFFFFFF)
    return (_hash_u32(h) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


class SeededValueNoise:
    """Stands for the `Noise` resource: fractal value noise in [-1, 1].  `get_noise_3dv` mirrors the script-callable
    `Noise.get_noise_3dv` (float32 numpy, the same arithmetic the device kernel performs per texel); cubemap
    generation itself never goes through it -- `NoiseCubemap._generate_images` launches the kernel."""

    def __init__(self, seed: int = 0, frequency: float = 0.01, fractal_octaves: int = 4, fractal_gain: float = 0.5):
        self._listeners = []
        self.seed, self.frequency, self.fractal_octaves, self.fractal_gain = seed, frequency, fractal_octaves, fractal_gain

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if not k.startswith("_"):
            for cb in list(getattr(self, "_listeners", [])):
                cb()

    def connect_changed(self, cb):
        self._listeners.append(cb)

    def disconnect_changed(self, cb):
        if cb in self._listeners:
            self._listeners.remove(cb)

    def _value(self, p, seed):
        f32 = np.float32
        fl = np.floor(p)
        t = p - fl
        w = t * t * (f32(3.0) - f32(2.0) * t)
        i0 = fl.astype(np.int32)
        i1 = i0 + np.int32(1)
        x0, y0, z0, x1, y1, z1 = i0[..., 0], i0[..., 1], i0[..., 2], i1[..., 0], i1[..., 1], i1[..., 2]
        wx, wy, wz = w[..., 0], w[..., 1], w[..., 2]
        one = f32(1.0)
        c00 = _lattice(x0, y0, z0, seed) * (one - wx) + _lattice(x1, y0, z0, seed) * wx
        c10 = _lattice(x0, y1, z0, seed) * (one - wx) + _lattice(x1, y1, z0, seed) * wx
        c01 = _lattice(x0, y0, z1, seed) * (one - wx) + _lattice(x1, y0, z1, seed) * wx
        c11 = _lattice(x0, y1, z1, seed) * (one - wx) + _lattice(x1, y1, z1, seed) * wx
        c0 = c00 * (one - wy) + c10 * wy
        c1 = c01 * (one - wy) + c11 * wy
        return c0 * (one - wz) + c1 * wz

    def get_noise_3dv(self, p) -> np.ndarray:
        p = np.asarray(p, dtype=np.float32)
        f32 = np.float32
        total = np.zeros(p.shape[:-1], dtype=np.float32)
        amp, norm, freq = f32(1.0), f32(0.0), f32(self.frequency)
        for o in range(int(self.fractal_octaves)):
            total = total + amp * self._value(p * freq, (self.seed + 1013 * o) & 0xFFFFFFFF)
            norm = f32(norm + amp)
            amp = f32(amp * f32(self.fractal_gain))
            freq = f32(freq * f32(2.0))
        return f32(2.0) * (total / norm) - f32(1.0)


def generate_importable_image(images: np.ndarray) -> np.ndarray:
    """noise_cubemap.gd:143-155: 3 x 2 atlas, side = x + 3 * y."""
    n = images.shape[1]
    atlas = np.empty((2 * n, 3 * n), dtype=np.uint8)
    for y in range(2):
        for x in range(3):
            atlas[y * n:(y + 1) * n, x * n:(x + 1) * n] = images[x + y * 3]
    return atlas


class NoiseCubemap:
    """See module docstring.  Generation runs on the GPU through `atmo_generate_noise_cubemap`."""

    def __init__(self, device: int = 0, noise: SeededValueNoise | None = None, resolution: int = 256, scale=(100.0, 100.0, 100.0)):
        self._lib = N.load()
        self._device = device
        self._ctx = None
        self._noise = None
        self._resolution = 256
        self._scale = (100.0, 100.0, 100.0)
        self._update_scheduled = False
        self._images = None
        self._changed = []
        self.last_kernel_ms = None
        self.resolution = resolution
        self.scale = scale
        self.noise = noise if noise is not None else SeededValueNoise()  # noise_cubemap.gd:51-54
        self._request_update()

    # ---- properties ------------------------------------------------------------------------------------
    @property
    def noise(self):
        return self._noise

    @noise.setter
    def noise(self, value):  # noise_cubemap.gd:13-22
        if self._noise is not None:
            self._noise.disconnect_changed(self._on_noise_changed)
        self._noise = value
        if self._noise is not None:
            self._noise.connect_changed(self._on_noise_changed)
            self._request_update()

    @property
    def resolution(self):
        return self._resolution

    @resolution.setter
    def resolution(self, value):  # noise_cubemap.gd:28-33
        r = min(max(int(value), 1), 4096)
        if r != self._resolution:
            self._resolution = r
            self._request_update()

    @property
    def scale(self):
        return self._scale

    @scale.setter
    def scale(self, value):  # noise_cubemap.gd:41-45
        value = tuple(float(v) for v in value)
        if value != self._scale:
            self._scale = value
            self._request_update()

    def connect_changed(self, cb):
        self._changed.append(cb)

    def _on_noise_changed(self):
        self._request_update()

    def _request_update(self):  # noise_cubemap.gd:61-64
        self._update_scheduled = True

    def process_deferred(self):
        """Runs the deferred `_update` if one is scheduled (Godot's call_deferred at idle time)."""
        if self._update_scheduled:
            self._update()

    def _update(self):  # noise_cubemap.gd:67-81
        if self._noise is None:
            self._update_scheduled = False
            return
        self._images = self._generate_images(self._resolution, self._noise, self._scale)
        self._update_scheduled = False
        for cb in list(self._changed):
            cb()

    def _context(self):
        if self._ctx is None:
            ctx = C.c_void_p()
            N.check(None, self._lib.atmo_create(self._device, N.VARIANT_NO_CLOUDS, 0, 0, N.LIGHT_LUT, 0, C.byref(ctx)))
            self._ctx = ctx
        return self._ctx

    def _generate_images(self, resolution, noise, scale) -> np.ndarray:  # noise_cubemap.gd:101-140, on the device
        out = np.empty((6, resolution, resolution), dtype=np.uint8)
        sc = (C.c_float * 3)(*[float(v) for v in scale])
        ms = C.c_double(0.0)
        rc = self._lib.atmo_generate_noise_cubemap(
            self._context(), resolution, int(noise.seed) & 0xFFFFFFFF, float(noise.frequency), int(noise.fractal_octaves),
            float(noise.fractal_gain), sc, 0, out.ctypes.data_as(C.c_void_p), C.byref(ms))
        N.check(self._ctx, rc)
        self.last_kernel_ms = ms.value
        return out

    def get_images(self) -> np.ndarray:
        self.process_deferred()
        return self._images

    def get_layer_data(self, side: int) -> np.ndarray:
        return self.get_images()[side]

    def generate_importable_image(self) -> np.ndarray:  # noise_cubemap.gd:93-97
        return generate_importable_image(self.get_images())

    def close(self):
        if self._ctx is not None:
            self._lib.atmo_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
