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

class SeededValueNoise:
    """Stands for the `Noise` resource: the settings of this package's fractal value noise in [-1, 1] (seed, frequency,
    octaves, gain) plus Godot's `changed` signal.  The arithmetic lives in the device kernel only
    (`atmo_noise_cubemap_kernel`); a float32 numpy statement of it for the CPU tests is tests/noise_host.py."""

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
