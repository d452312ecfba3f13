"""Texture units for tests/golden/gdshader_vm.py: what `texture()` / `texelFetch()` return for each sampler the reference
declares.  Filtering is engine / hardware behaviour, not reference text; these follow the conventions stated in DESIGN.md
section 2 and oracle/atmo_oracle.h, in float32, written for whole lane arrays.  TEST INFRASTRUCTURE (fixture generation)."""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _mix(a, b, t):
    return a * (F32(1.0) - t) + b * t


def _floor_int(x):
    with np.errstate(all="ignore"):
        f = np.floor(x)
        i = np.where(np.isfinite(f), f, 0.0).astype(np.int64)
    return f, i


class LutTexture:
    """sampler2D, repeat_disable, linear: the optical-depth LUT (R32F), clamp to edge, texel centres at (i + 0.5) / N."""

    def __init__(self, lut):
        self.t = np.ascontiguousarray(lut, dtype=F32)

    def texture(self, c):
        h, w = self.t.shape
        x, y = c[0] * F32(w) - F32(0.5), c[1] * F32(h) - F32(0.5)
        xf, xi = _floor_int(x)
        yf, yi = _floor_int(y)
        fx, fy = x - xf, y - yf
        i0, i1 = np.clip(xi, 0, w - 1), np.clip(xi + 1, 0, w - 1)
        j0, j1 = np.clip(yi, 0, h - 1), np.clip(yi + 1, 0, h - 1)
        t = self.t
        return _mix(_mix(t[j0, i0], t[j0, i1], fx), _mix(t[j1, i0], t[j1, i1], fx), fy)


class DepthTexture:
    """hint_depth_texture sampled at SCREEN_UV: the depth of the pixel the fragment covers."""

    def __init__(self, depth):
        self.t = np.ascontiguousarray(depth, dtype=F32)

    def texture(self, c):
        h, w = self.t.shape
        _, xi = _floor_int(c[0] * F32(w))
        _, yi = _floor_int(c[1] * F32(h))
        return self.t[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)]


class ByteTexture2D:
    """sampler2D, filter_nearest, repeat_enable over an R8 image: texelFetch returns byte / 255."""

    def __init__(self, texels):
        self.t = np.ascontiguousarray(texels, dtype=np.uint8)

    def texel_fetch(self, c, lod):
        h, w = self.t.shape
        return self.t[c[1].astype(np.int64) % h, c[0].astype(np.int64) % w].astype(F32) / F32(255.0)


class ShapeTexture:
    """sampler3D, repeat_enable, linear over an R8 volume indexed [z, y, x]: trilinear, mix order x, y, z."""

    def __init__(self, texels):
        self.t = np.ascontiguousarray(texels, dtype=np.uint8)

    def texture(self, c):
        n = self.t.shape[0]
        q = [c[k] * F32(n) - F32(0.5) for k in range(3)]
        fl = [_floor_int(v) for v in q]
        f = [q[k] - fl[k][0] for k in range(3)]
        i0 = [fl[k][1] % n for k in range(3)]
        i1 = [(fl[k][1] + 1) % n for k in range(3)]

        def s(i, j, k):
            return self.t[k, j, i].astype(F32) / F32(255.0)

        c00 = _mix(s(i0[0], i0[1], i0[2]), s(i1[0], i0[1], i0[2]), f[0])
        c10 = _mix(s(i0[0], i1[1], i0[2]), s(i1[0], i1[1], i0[2]), f[0])
        c01 = _mix(s(i0[0], i0[1], i1[2]), s(i1[0], i0[1], i1[2]), f[0])
        c11 = _mix(s(i0[0], i1[1], i1[2]), s(i1[0], i1[1], i1[2]), f[0])
        return _mix(_mix(c00, c10, f[1]), _mix(c01, c11, f[1]), f[2])


class CubeTexture:
    """samplerCube over an R8 cubemap, level 0, bilinear, seamless.  `padded`: (6, n + 2, n + 2) uint8, the faces with a
    one-texel apron holding the texel reached by folding over the cube edge (corners: mean of the three corner texels) --
    built by the caller from the checker's `cube_texel`."""

    def __init__(self, padded):
        self.t = np.ascontiguousarray(padded, dtype=np.uint8)

    def texture(self, d):
        n = self.t.shape[1] - 2
        x, y, z = d[0], d[1], d[2]
        ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
        fz = (az >= ax) & (az >= ay)          # z wins ties over y over x (Vulkan)
        fy = ~fz & (ay >= ax)
        face = np.where(fz, np.where(z >= 0, 4, 5), np.where(fy, np.where(y >= 0, 2, 3), np.where(x >= 0, 0, 1)))
        sc = np.choose(face, [-z, z, x, x, x, -x])
        tc = np.choose(face, [-y, -y, z, -z, -y, -y])
        ma = np.where(fz, az, np.where(fy, ay, ax))
        with np.errstate(all="ignore"):
            s = F32(0.5) * (sc / ma + F32(1.0))
            t = F32(0.5) * (tc / ma + F32(1.0))
        u, v = s * F32(n) - F32(0.5), t * F32(n) - F32(0.5)
        uf, ui = _floor_int(u)
        vf, vi = _floor_int(v)
        fx, fy_ = u - uf, v - vf
        i0, j0 = np.clip(ui, -1, n - 1), np.clip(vi, -1, n - 1)

        def s8(i, j):
            return self.t[face, j + 1, i + 1].astype(F32) / F32(255.0)

        return _mix(_mix(s8(i0, j0), s8(i0 + 1, j0), fx), _mix(s8(i0, j0 + 1), s8(i0 + 1, j0 + 1), fx), fy_)


def pad_cubemap(faces, cube_texel):
    """(6, n, n) uint8 -> (6, n + 2, n + 2) with the seamless apron; cube_texel(f, i, j) for i, j in [-1, n]."""
    n = faces.shape[1]
    out = np.zeros((6, n + 2, n + 2), dtype=np.uint8)
    out[:, 1:-1, 1:-1] = faces
    for f in range(6):
        for k in range(-1, n + 1):
            for (i, j) in ((k, -1), (k, n), (-1, k), (n, k)):
                out[f, j + 1, i + 1] = cube_texel(f, i, j)
    return out
