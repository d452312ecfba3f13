// atmo_layout.h -- the device texture layouts (DESIGN.md "Data layout in HBM"), one definition for both sides:
// the re-layout kernels of atmo_kernels.hip evaluate these functions per output element on the GPU, and the host
// helpers of the C ABI (atmo_host_layout_*: the CPU checker of tests/test_host_logic.py) loop over them.
//
//   LUT    (w+2) x (h+2) fp32 with a clamp-to-edge apron (u_optical_depth_texture is `repeat_disable`, v2:7); the kernels
//          sample a footprint copy of it (4 floats per bilinear footprint, built on the device from the apron buffer)
//   shape  n^3 uint32 "xy footprints": word (i,j,k) = T(i,j,k), T(i+1,j,k), T(i,j+1,k), T(i+1,j+1,k), repeat wrap
//   cube   per mip level, 6 x (n+1)^2 uint32 footprints of the faces padded with a seamless apron: the border texel is
//          the one reached by folding over the cube edge, a corner the rounded mean of the three corner texels
//   mips   level l+1 = 2x2 box of level l on L8, (a + b + c + d + 2) >> 2 -- what Image.generate_mipmaps computes for
//          FORMAT_L8 (noise_cubemap.gd:135; engine arithmetic, a stated convention of this build)
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define ATMO_HD __host__ __device__ inline
#else
#define ATMO_HD inline
#endif

namespace atmo {

// Vulkan cube face table (same as noise_cubemap.gd:110-128): direction = major * ma + S * sc + T * tc.
// Returns component `axis` (0..2) of the face's major / s / t axis.
ATMO_HD int cube_axis(int f, int which /*0 major, 1 s, 2 t*/, int axis) {
    // rows: +X, -X, +Y, -Y, +Z, -Z; columns: major xyz, s xyz, t xyz
    switch (f * 9 + which * 3 + axis) {
    case 0: return 1;    // +X major.x
    case 5: return -1;   // +X s = -z
    case 7: return -1;   // +X t = -y
    case 9: return -1;   // -X major.x
    case 14: return 1;   // -X s = +z
    case 16: return -1;  // -X t = -y
    case 19: return 1;   // +Y major.y
    case 21: return 1;   // +Y s = +x
    case 26: return 1;   // +Y t = +z
    case 28: return -1;  // -Y major.y
    case 30: return 1;   // -Y s = +x
    case 35: return -1;  // -Y t = -z
    case 38: return 1;   // +Z major.z
    case 39: return 1;   // +Z s = +x
    case 43: return -1;  // +Z t = -y
    case 47: return -1;  // -Z major.z
    case 48: return -1;  // -Z s = -x
    case 52: return -1;  // -Z t = -y
    default: return 0;
    }
}

// Integer cube-surface coordinates: texel centre (i,j) of face f sits at 2*i+1-n, 2*j+1-n in the face plane and at n on
// the major axis (units of half texels).  Stepping one texel off the face keeps the in-plane coordinate at the edge
// (+-n) and moves the major-axis coordinate in by one texel (n - 1), which is a texel centre of the neighbouring face.
ATMO_HD uint8_t cube_fold(const uint8_t *faces, int n, int f, int i, int j) {
    int sc = 2 * i + 1 - n, tc = 2 * j + 1 - n, ma = n;
    if (i < 0) { sc = -n; ma = n - 1; } else if (i >= n) { sc = n; ma = n - 1; }
    if (j < 0) { tc = -n; ma = n - 1; } else if (j >= n) { tc = n; ma = n - 1; }
    int p[3];
    for (int a = 0; a < 3; ++a) p[a] = cube_axis(f, 0, a) * ma + cube_axis(f, 1, a) * sc + cube_axis(f, 2, a) * tc;
    int f2 = 0;  // which face is this point on?  exactly one coordinate has magnitude n
    for (int g = 0; g < 6; ++g) {
        const int m = cube_axis(g, 0, 0) * p[0] + cube_axis(g, 0, 1) * p[1] + cube_axis(g, 0, 2) * p[2];
        if (m == n) { f2 = g; break; }
    }
    const int s2 = cube_axis(f2, 1, 0) * p[0] + cube_axis(f2, 1, 1) * p[1] + cube_axis(f2, 1, 2) * p[2];
    const int t2 = cube_axis(f2, 2, 0) * p[0] + cube_axis(f2, 2, 1) * p[1] + cube_axis(f2, 2, 2) * p[2];
    int i2 = (s2 + n - 1) / 2, j2 = (t2 + n - 1) / 2;
    i2 = i2 < 0 ? 0 : (i2 > n - 1 ? n - 1 : i2);
    j2 = j2 < 0 ? 0 : (j2 > n - 1 ? n - 1 : j2);
    return faces[((size_t)f2 * n + j2) * n + i2];
}

// texel (i,j), i,j in [-1, n], of face f padded with the seamless apron
ATMO_HD uint8_t cube_padded_texel(const uint8_t *faces, int n, int f, int i, int j) {
    const bool oi = (i < 0 || i >= n), oj = (j < 0 || j >= n);
    if (!oi && !oj) return faces[((size_t)f * n + j) * n + i];
    if (oi && oj) {
        const int ci = i < 0 ? 0 : n - 1, cj = j < 0 ? 0 : n - 1;
        const int a = faces[((size_t)f * n + cj) * n + ci];
        const int b = cube_fold(faces, n, f, i, cj);
        const int c = cube_fold(faces, n, f, ci, j);
        return (uint8_t)((a + b + c + 1) / 3);
    }
    return cube_fold(faces, n, f, i, j);
}

// footprint word (i,j), i,j in [0, n], of face f: padded texels (i-1,j-1), (i,j-1), (i-1,j), (i,j) in bytes 0..3
ATMO_HD uint32_t cube_footprint_word(const uint8_t *faces, int n, int f, int i, int j) {
    return (uint32_t)cube_padded_texel(faces, n, f, i - 1, j - 1) | ((uint32_t)cube_padded_texel(faces, n, f, i, j - 1) << 8) |
           ((uint32_t)cube_padded_texel(faces, n, f, i - 1, j) << 16) | ((uint32_t)cube_padded_texel(faces, n, f, i, j) << 24);
}

// footprint word (i,j,k) of an n^3 repeat-wrapped volume
ATMO_HD uint32_t shape_footprint_word(const uint8_t *t, int n, int i, int j, int k) {
    const int i1 = (i + 1) % n, j1 = (j + 1) % n;
    const uint8_t *r0 = t + ((size_t)k * n + j) * n, *r1 = t + ((size_t)k * n + j1) * n;
    return (uint32_t)r0[i] | ((uint32_t)r0[i1] << 8) | ((uint32_t)r1[i] << 16) | ((uint32_t)r1[i1] << 24);
}

// apron element (i,j), i in [0, w+2), j in [0, h+2), of a w x h LUT
ATMO_HD float lut_apron_value(const float *lut, int w, int h, int i, int j) {
    const int ci = i - 1 < 0 ? 0 : (i - 1 >= w ? w - 1 : i - 1);
    const int cj = j - 1 < 0 ? 0 : (j - 1 >= h ? h - 1 : j - 1);
    return lut[(size_t)cj * w + ci];
}

// one texel of mip level l+1 (n/2 per side) from level l (n per side), face f
ATMO_HD uint8_t cube_mip_texel(const uint8_t *level, int n, int f, int i, int j) {
    const uint8_t *p = level + ((size_t)f * n + 2 * j) * n + 2 * i;
    return (uint8_t)(((int)p[0] + (int)p[1] + (int)p[n] + (int)p[n + 1] + 2) >> 2);
}

// number of levels of a full chain and element offsets of the packed per-level arrays
ATMO_HD int cube_full_mip_count(int n) {
    int l = 1;
    while (n > 1) { n >>= 1; ++l; }
    return l;
}
ATMO_HD size_t cube_level_texels(int n, int level) { const size_t m = (size_t)(n >> level); return 6 * m * m; }
ATMO_HD size_t cube_level_footprints(int n, int level) { const size_t m = (size_t)(n >> level) + 1; return 6 * m * m; }

}  // namespace atmo
