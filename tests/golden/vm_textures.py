"""Texture units for tests/golden/gdshader_vm.py: what `texture()` / `texelFetch()` return for each sampler the reference
declares.  Filtering is engine / hardware behaviour, not reference text; these follow the conventions stated in DESIGN.md
section 2 and oracle/atmo_oracle.h, in float32, written for whole lane arrays.  TEST INFRASTRUCTURE (fixture generation)."""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _mix(a, b, t):
    return a * (F32(1.0) - t) + b * t


def _quantize(f, bits):
    """Sensitivity runs only: a filter weight held to `bits` fractional bits, as texture units do (0 = exact float32)."""
    if not bits:
        return f
    q = F32(1 << bits)
    return (np.floor(f * q + F32(0.5)) / q).astype(F32)


def _floor_int(x):
    with np.errstate(all="ignore"):
        f = np.floor(x)
        i = np.where(np.isfinite(f), f, 0.0).astype(np.int64)
    return f, i


class LutTexture:
    """sampler2D, repeat_disable, linear: the optical-depth LUT (R32F), clamp to edge, texel centres at (i + 0.5) / N."""

    def __init__(self, lut, quantize_bits=0):
        self.t = np.ascontiguousarray(lut, dtype=F32)
        self.q = quantize_bits

    def texture(self, c):
        h, w = self.t.shape
        x, y = c[0] * F32(w) - F32(0.5), c[1] * F32(h) - F32(0.5)
        xf, xi = _floor_int(x)
        yf, yi = _floor_int(y)
        fx, fy = _quantize(x - xf, self.q), _quantize(y - yf, self.q)
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

    def __init__(self, texels, quantize_bits=0):
        self.t = np.ascontiguousarray(texels, dtype=np.uint8)
        self.q = quantize_bits

    def texture(self, c):
        n = self.t.shape[0]
        q = [c[k] * F32(n) - F32(0.5) for k in range(3)]
        fl = [_floor_int(v) for v in q]
        f = [_quantize(q[k] - fl[k][0], self.q) for k in range(3)]
        i0 = [fl[k][1] % n for k in range(3)]
        i1 = [(fl[k][1] + 1) % n for k in range(3)]

        def s(i, j, k):
            return self.t[k, j, i].astype(F32) / F32(255.0)

        c00 = _mix(s(i0[0], i0[1], i0[2]), s(i1[0], i0[1], i0[2]), f[0])
        c10 = _mix(s(i0[0], i1[1], i0[2]), s(i1[0], i1[1], i0[2]), f[0])
        c01 = _mix(s(i0[0], i0[1], i1[2]), s(i1[0], i0[1], i1[2]), f[0])
        c11 = _mix(s(i0[0], i1[1], i1[2]), s(i1[0], i1[1], i1[2]), f[0])
        return _mix(_mix(c00, c10, f[1]), _mix(c01, c11, f[1]), f[2])


# Vulkan cube face table: direction = major + sc * S + tc * T (faces +X, -X, +Y, -Y, +Z, -Z)
_MAJOR = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
_S_AX = np.array([[0, 0, -1], [0, 0, 1], [1, 0, 0], [1, 0, 0], [1, 0, 0], [-1, 0, 0]], dtype=np.float64)
_T_AX = np.array([[0, -1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, -1, 0], [0, -1, 0]], dtype=np.float64)


def _select_face(d):
    """Vulkan face selection of directions d (3, N): z wins ties over y over x.  Returns face, sc, tc, |ma|."""
    x, y, z = d[0], d[1], d[2]
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    fz = (az >= ax) & (az >= ay)
    fy = ~fz & (ay >= ax)
    face = np.where(fz, np.where(z >= 0, 4, 5), np.where(fy, np.where(y >= 0, 2, 3), np.where(x >= 0, 0, 1)))
    sc = np.choose(face, [-z, z, x, x, x, -x])
    tc = np.choose(face, [-y, -y, z, -z, -y, -y])
    ma = np.where(fz, az, np.where(fy, ay, ax))
    return face, sc, tc, ma


def _face_frame(face, v):
    """(sc, tc, signed major) of arbitrary vectors v (3, N) in the frame of the given faces: the table's linear maps."""
    x, y, z = v[0], v[1], v[2]
    sc = np.choose(face, [-z, z, x, x, x, -x])
    tc = np.choose(face, [-y, -y, z, -z, -y, -y])
    ma = np.choose(face, [x, -x, y, -y, z, -z])
    return sc, tc, ma


def seamless_apron(faces):
    """(6, n, n) uint8 -> (6, n + 2, n + 2): the faces with a one-texel apron, stated from the cube's GEOMETRY alone (no call
    into the checker under test):
      * an apron texel beside an edge is the texel of the neighbouring face that contains the 3-D direction of the apron
        texel's own centre -- the point of this face's plane one texel beyond the edge, (sc, tc) = 2 (i + 0.5) / n - 1 with
        i = -1 or n -- found by selecting the face of that direction again and truncating to a texel there;
      * an apron CORNER texel is the mean, rounded to nearest, of the three texels that touch that vertex of the cube (one on
        each of the three faces meeting there): a seamless sampler has only three texels to filter at a cube corner."""
    faces = np.ascontiguousarray(faces, dtype=np.uint8)
    n = faces.shape[1]
    out = np.zeros((6, n + 2, n + 2), dtype=np.uint8)
    out[:, 1:-1, 1:-1] = faces
    c = 2.0 * (np.arange(-1, n + 1, dtype=np.float64) + 0.5) / n - 1.0   # plane coordinate of texel column / row -1 .. n
    for f in range(6):
        for edge in range(4):
            k = np.arange(n)
            if edge == 0:
                i, j = np.full(n, -1), k
            elif edge == 1:
                i, j = np.full(n, n), k
            elif edge == 2:
                i, j = k, np.full(n, -1)
            else:
                i, j = k, np.full(n, n)
            d = (_MAJOR[f][:, None] + c[i + 1][None, :] * _S_AX[f][:, None] + c[j + 1][None, :] * _T_AX[f][:, None])
            g, sc, tc, ma = _select_face(d)
            assert np.all(g != f)
            i2 = np.clip(np.floor((sc / ma + 1.0) * 0.5 * n).astype(np.int64), 0, n - 1)
            j2 = np.clip(np.floor((tc / ma + 1.0) * 0.5 * n).astype(np.int64), 0, n - 1)
            out[f, j + 1, i + 1] = faces[g, j2, i2]
        for (ci, cj) in ((-1, -1), (n, -1), (-1, n), (n, n)):
            vertex = _MAJOR[f] + np.sign(c[ci + 1]) * _S_AX[f] + np.sign(c[cj + 1]) * _T_AX[f]   # (+-1, +-1, +-1)
            total = 0
            for g in range(6):
                if _MAJOR[g] @ vertex <= 0:
                    continue                                     # the three faces on the vertex's side of each axis
                sc, tc = _S_AX[g] @ vertex, _T_AX[g] @ vertex    # +-1: which corner texel of face g touches the vertex
                total += int(faces[g, 0 if tc < 0 else n - 1, 0 if sc < 0 else n - 1])
            out[f, cj + 1, ci + 1] = (2 * total + 3) // 6        # round(total / 3)
    return out


def mip_chain(faces):
    """Image.generate_mipmaps on L8 cube faces (noise_cubemap.gd:107,135): 2 x 2 box, (a + b + c + d + 2) >> 2, down to 1 x 1."""
    levels = [np.ascontiguousarray(faces, dtype=np.uint8)]
    while levels[-1].shape[1] > 1:
        a = levels[-1].astype(np.uint32)
        m = a.shape[1] // 2
        a = a[:, :2 * m, :2 * m]
        levels.append(((a[:, 0::2, 0::2] + a[:, 0::2, 1::2] + a[:, 1::2, 0::2] + a[:, 1::2, 1::2] + 2) >> 2).astype(np.uint8))
    return levels


def _bilinear_seamless(padded, face, s, t, quantize_bits=0):
    n = padded.shape[1] - 2
    u, v = s * F32(n) - F32(0.5), t * F32(n) - F32(0.5)
    uf, ui = _floor_int(u)
    vf, vi = _floor_int(v)
    fx, fy = _quantize((u - uf).astype(F32), quantize_bits), _quantize((v - vf).astype(F32), quantize_bits)
    i0, j0 = np.clip(ui, -1, n - 1), np.clip(vi, -1, n - 1)

    def s8(i, j):
        return padded[face, j + 1, i + 1].astype(F32) / F32(255.0)

    return _mix(_mix(s8(i0, j0), s8(i0 + 1, j0), fx), _mix(s8(i0, j0 + 1), s8(i0 + 1, j0 + 1), fx), fy)


class CubeTexture:
    """samplerCube over an R8 cubemap, level 0, bilinear, seamless.  `padded`: (6, n + 2, n + 2) uint8 from seamless_apron()."""

    def __init__(self, padded, quantize_bits=0):
        self.t = np.ascontiguousarray(padded, dtype=np.uint8)
        self.q = quantize_bits

    def texture(self, d):
        face, sc, tc, ma = _select_face(d)
        with np.errstate(all="ignore"):
            s = F32(0.5) * (sc / ma + F32(1.0))
            t = F32(0.5) * (tc / ma + F32(1.0))
        return _bilinear_seamless(self.t, face, s, t, self.q)


class CubeTextureLod:
    """samplerCube with the engine's default filter for a spatial shader -- linear-mipmap, IMPLICIT level of detail
    (cloud_funcs.gdshaderinc:15,45 declare no filter hint; noise_cubemap.gd:107,135 builds the mip chain).

    The interpreter holds all lanes of a `texture()` call at once, so the derivatives are what the fragment pipeline's are:
    differences between the lanes of a 2 x 2 pixel quad AT THE SAME CALL.  `quad` = (qx, qy): for every lane the lane index of
    its horizontal / vertical quad partner, -1 when that pixel is outside the viewport.  `reach`: lanes that reach this call
    (Machine.bi_texture); a partner that does not reach it contributes a zero derivative.  Rule (stated convention of this
    build; Vulkan 1.3 "Cube Map Derivative Transformation" and "Scale Factor Operation"):
      * the partner's direction is expressed in the frame of the face selected by the lane's OWN direction; a partner beyond
        that face's half space (major component <= 0 there) gives no usable derivative;
      * s' - s = 0.5 (dsc ma - sc dma) / (ma ma')  with d* = partner - self in that frame (the exact difference of the two
        projections, written without cancellation), likewise t;
      * rho^2 = n^2 (ds^2 + dt^2) per axis, lambda = 0.5 log2(max(rho_x^2, rho_y^2)) clamped to [0, levels - 1];
      * result = mix(level floor(lambda), level floor(lambda) + 1, fract(lambda)), each level bilinear + seamless.
    `alt`: sensitivity runs only -- "f64_plain" evaluates s' - s as the plain difference of the two projections in float64; "fast_log2" takes
    lambda from a piecewise-linear log2 as llvmpipe does (profiles/round5/mesa_pin.txt, section 11); "unmasked" lets a quad partner that did NOT reach
    the call contribute the coordinate its lane holds anyway (what an implementation that runs all four lanes of a quad under an execution mask
    differences against); "fast_log2+unmasked" both."""

    needs_quad = True

    def __init__(self, levels, quad, quantize_bits=0, alt=None):
        self.levels = [seamless_apron(lv) for lv in levels]
        self.n0 = levels[0].shape[1]
        self.qx, self.qy = (np.asarray(q, dtype=np.int64) for q in quad)
        self.q = quantize_bits
        self.alt = alt

    def texture_quad(self, d, reach):
        d = d.astype(F32)
        nl = len(self.levels)
        face, sc, tc, ma = _select_face(d)
        with np.errstate(all="ignore"):
            s = F32(0.5) * (sc / ma + F32(1.0))
            t = F32(0.5) * (tc / ma + F32(1.0))
            rho2 = np.zeros(d.shape[1], dtype=F32)
            n2 = F32(self.n0) * F32(self.n0)
            for q in (self.qx, self.qy):
                ok = (q >= 0) & (reach[np.clip(q, 0, None)] | (self.alt is not None and "unmasked" in self.alt))  # "unmasked": sensitivity runs only, see below
                dv = (d[:, np.clip(q, 0, None)] - d).astype(F32)
                dsc, dtc, dma = _face_frame(face, dv)
                ma2 = (ma + dma).astype(F32)
                ok &= ma2 > 0
                if self.alt == "f64_plain":
                    psc, ptc, pma = _face_frame(face, d[:, np.clip(q, 0, None)].astype(np.float64))
                    ds = 0.5 * (psc / pma + 1.0) - 0.5 * (sc.astype(np.float64) / ma + 1.0)
                    dt = 0.5 * (ptc / pma + 1.0) - 0.5 * (tc.astype(np.float64) / ma + 1.0)
                    r2 = ((ds * ds + dt * dt) * float(n2)).astype(F32)
                else:
                    inv = (F32(0.5) / (ma * ma2)).astype(F32)
                    ds = ((dsc * ma - sc * dma) * inv).astype(F32)
                    dt = ((dtc * ma - tc * dma) * inv).astype(F32)
                    r2 = ((ds * ds + dt * dt) * n2).astype(F32)
                rho2 = np.where(ok, np.fmax(rho2, r2), rho2)
            if self.alt is not None and "fast_log2" in self.alt:  # sensitivity runs only: llvmpipe's level-of-detail unit takes log2 piecewise linear (exponent + mantissa - 1)
                m, e = np.frexp(np.where(rho2 > 0, rho2, F32(1.0)).astype(np.float64))
                lam = np.where(rho2 > 0, F32(0.5) * ((e - 1) + (2.0 * m - 1.0)).astype(F32), F32(0.0)).astype(F32)
            else:
                lam = np.where(rho2 > 0, F32(0.5) * np.log2(np.where(rho2 > 0, rho2, F32(1.0)).astype(np.float64)).astype(F32), F32(0.0)).astype(F32)  # log2 rounded once
        lam = np.fmin(np.fmax(lam, F32(0.0)), F32(nl - 1))
        lf = np.floor(lam)
        lo = lf.astype(np.int64)
        hi = np.minimum(lo + 1, nl - 1)
        fr = (lam - lf).astype(F32)
        out = np.zeros(d.shape[1], dtype=F32)
        for level in range(nl):
            use_lo, use_hi = lo == level, (hi == level) & (hi != lo) & (fr != 0)
            if not (use_lo.any() or use_hi.any()):
                continue
            v = _bilinear_seamless(self.levels[level], face, s, t, self.q)
            out = np.where(use_lo, np.where((hi == lo) | (fr == 0), v, v * (F32(1.0) - fr)), out)
            out = np.where(use_hi, out + v * fr, out)
        return out.astype(F32)


def pad_cubemap(faces, cube_texel=None):
    """The seamless apron.  `cube_texel` (the checker's own edge rule) is accepted for old callers and IGNORED: since round 3
    the interpreter's cubemap edges are stated here, from the cube's geometry (seamless_apron), not taken from the oracle."""
    return seamless_apron(faces)
