/*
 * atmo_oracle.c -- scalar CPU restatement of the reference's per-pixel atmosphere/cloud raymarch.
 * TEST INFRASTRUCTURE ONLY (see atmo_oracle.h).  Pinned to the executed reference text by tests/test_reference_exec.py (see atmo_oracle.h).
 *
 * Evaluation order follows the GDShader source statement by statement; build with
 * -ffp-contract=off so nothing is fused.  "ref:" comments cite /root/reference/addons/
 * zylann.atmosphere/shaders/ (I/ = include/).
 */
#include "atmo_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F64
typedef double REAL;
#define SFX(name) name##_f64
#define R_SQRT sqrt
#define R_EXP exp
#define R_LOG2 log2
#define R_FLOOR floor
#define R_FABS fabs
#else
typedef float REAL;
#define SFX(name) name##_f32
#define R_SQRT sqrtf
#define R_EXP expf
#define R_LOG2 log2f
#define R_FLOOR floorf
#define R_FABS fabsf
#endif

#define K(x) ((REAL)(x))

typedef struct { REAL x, y; } vec2;
typedef struct { REAL x, y, z; } vec3;
typedef struct { REAL x, y, z, w; } vec4;

/* ---- GLSL built-ins, spelled out -------------------------------------------------------- */
static inline REAL r_min(REAL a, REAL b) { return b < a ? b : a; }
static inline REAL r_max(REAL a, REAL b) { return a < b ? b : a; }
static inline REAL r_clamp(REAL x, REAL lo, REAL hi) { return r_min(r_max(x, lo), hi); }
static inline REAL r_mix(REAL a, REAL b, REAL t) { return a * (K(1.0) - t) + b * t; }
static inline REAL r_smoothstep(REAL e0, REAL e1, REAL x) {
    REAL t = r_clamp((x - e0) / (e1 - e0), K(0.0), K(1.0));
    return t * t * (K(3.0) - K(2.0) * t);
}
static inline vec3 v3(REAL x, REAL y, REAL z) { vec3 v = {x, y, z}; return v; }
static inline vec3 v3_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v3_scale(vec3 a, REAL s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline REAL v3_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline REAL v3_length(vec3 a) { return R_SQRT(v3_dot(a, a)); }
static inline vec3 v3_normalize(vec3 a) { return v3_scale(a, K(1.0) / R_SQRT(v3_dot(a, a))); }

/* column-major mat4 (m[col*4+row]) times vec4, summed left to right */
static inline vec4 m4_mul_v4(const REAL *m, vec4 v) {
    vec4 r;
    r.x = m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w;
    r.y = m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w;
    r.z = m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w;
    r.w = m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w;
    return r;
}
/* C = A*B, column-major, inner sum left to right */
static void m4_mul_m4(const REAL *a, const REAL *b, REAL *c) {
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            c[col * 4 + row] = a[0 * 4 + row] * b[col * 4 + 0] + a[1 * 4 + row] * b[col * 4 + 1] +
                               a[2 * 4 + row] * b[col * 4 + 2] + a[3 * 4 + row] * b[col * 4 + 3];
}

/* ref: I/util.gdshaderinc:49-59 */
static inline REAL pow2(REAL x) { return x * x; }
static inline REAL pow4(REAL x) { return x * x * x * x; }

/* ref: I/util.gdshaderinc:20-40 -- x = first hit, y = second hit, equal if not hit */
static vec2 ray_sphere(vec3 center, REAL radius, vec3 ray_origin, vec3 ray_dir) {
    vec3 oc = v3_sub(ray_origin, center);
    REAL b = v3_dot(oc, ray_dir);
    vec3 qc = v3_sub(oc, v3_scale(ray_dir, b));
    REAL h = radius * radius - v3_dot(qc, qc);
    vec2 r;
    if (h < K(0.0)) {
        r.x = K(1000000.0);
        r.y = K(1000000.0);
        return r;
    }
    h = R_SQRT(h);
    r.x = -b - h;
    r.y = -b + h;
    return r;
}

/* ref: I/util.gdshaderinc:61-69 */
static vec4 blend_colors(vec4 self, vec4 over) {
    REAL sa = K(1.0) - over.w;
    REAL a = self.w * sa + over.w;
    vec4 r;
    if (a == K(0.0)) {
        r.x = r.y = r.z = r.w = K(0.0);
    } else {
        r.x = (self.x * self.w * sa + over.x * over.w) / a;
        r.y = (self.y * self.w * sa + over.y * over.w) / a;
        r.z = (self.z * self.w * sa + over.z * over.w) / a;
        r.w = a;
    }
    return r;
}

/* ---- uniforms widened to REAL ---------------------------------------------------------- */
typedef struct {
    REAL planet_radius, atmosphere_height, density, scattering_strength;
    vec3 wavelengths, modulate, ambient;
    REAL sphere_depth_factor;
    REAL cloud_density_scale, cloud_bottom, cloud_top, cloud_blend;
    REAL world_to_model[16];
    REAL cloud_shape_invert, cloud_coverage_bias, cloud_shape_factor, cloud_shape_scale;
    REAL cov_rot[4];
    vec3 day0, day1, night0, night1;
    REAL day_night_transition_scale;
    OracleTextures tex;
    OracleConfig cfg;
} Ctx;

static void ctx_init(Ctx *c, const OracleParams *p, const OracleTextures *t, const OracleConfig *cfg) {
    memset(c, 0, sizeof(*c));
    c->planet_radius = p->u_planet_radius;
    c->atmosphere_height = p->u_atmosphere_height;
    c->density = p->u_density;
    c->scattering_strength = p->u_scattering_strength;
    c->wavelengths = v3(p->u_scattering_wavelengths[0], p->u_scattering_wavelengths[1], p->u_scattering_wavelengths[2]);
    c->modulate = v3(p->u_atmosphere_modulate[0], p->u_atmosphere_modulate[1], p->u_atmosphere_modulate[2]);
    c->ambient = v3(p->u_atmosphere_ambient_color[0], p->u_atmosphere_ambient_color[1], p->u_atmosphere_ambient_color[2]);
    c->sphere_depth_factor = p->u_sphere_depth_factor;
    c->cloud_density_scale = p->u_cloud_density_scale;
    c->cloud_bottom = p->u_cloud_bottom;
    c->cloud_top = p->u_cloud_top;
    c->cloud_blend = p->u_cloud_blend;
    for (int i = 0; i < 16; ++i) c->world_to_model[i] = p->u_world_to_model_matrix[i];
    c->cloud_shape_invert = p->u_cloud_shape_invert;
    c->cloud_coverage_bias = p->u_cloud_coverage_bias;
    c->cloud_shape_factor = p->u_cloud_shape_factor;
    c->cloud_shape_scale = p->u_cloud_shape_scale;
    for (int i = 0; i < 4; ++i) c->cov_rot[i] = p->u_cloud_coverage_rotation[i];
    c->day0 = v3(p->u_day_color0[0], p->u_day_color0[1], p->u_day_color0[2]);
    c->day1 = v3(p->u_day_color1[0], p->u_day_color1[1], p->u_day_color1[2]);
    c->night0 = v3(p->u_night_color0[0], p->u_night_color0[1], p->u_night_color0[2]);
    c->night1 = v3(p->u_night_color1[0], p->u_night_color1[1], p->u_night_color1[2]);
    c->day_night_transition_scale = p->u_day_night_transition_scale;
    if (t) c->tex = *t;
    if (cfg) c->cfg = *cfg;
}

/* ref: I/atmosphere_common.gdshaderinc:12-24.  `height` is the distance from the planet centre. */
static REAL get_atmosphere_density(const Ctx *c, REAL height) {
    REAL sd = height - c->planet_radius;
    REAL h = r_clamp(sd / c->atmosphere_height, K(0.0), K(1.0));
    REAL y = K(1.0) - h;
    REAL density = y * y * y * c->density;
    return density;
}

/* ---- software samplers (conventions stated in atmo_oracle.h) ---------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* texture(sampler2D repeat_disable, uv).r on an R32F image, bilinear */
static REAL sample_lut(const OracleTextures *t, REAL u, REAL v) {
    const int w = t->lut_w, h = t->lut_h;
    REAL x = u * K(w) - K(0.5);
    REAL y = v * K(h) - K(0.5);
    REAL xf = R_FLOOR(x), yf = R_FLOOR(y);
    REAL fx = x - xf, fy = y - yf;
    int i0 = clampi((int)xf, 0, w - 1), i1 = clampi((int)xf + 1, 0, w - 1);
    int j0 = clampi((int)yf, 0, h - 1), j1 = clampi((int)yf + 1, 0, h - 1);
    REAL t00 = t->optical_depth[j0 * w + i0], t10 = t->optical_depth[j0 * w + i1];
    REAL t01 = t->optical_depth[j1 * w + i0], t11 = t->optical_depth[j1 * w + i1];
    return r_mix(r_mix(t00, t10, fx), r_mix(t01, t11, fx), fy);
}

static inline int wrapi(int v, int n) { int m = v % n; return m < 0 ? m + n : m; }

/* texture(sampler3D repeat_enable, p).r on an R8 image, trilinear */
static REAL sample_shape(const OracleTextures *t, vec3 p) {
    const int n = t->shape_n;
    REAL x = p.x * K(n) - K(0.5), y = p.y * K(n) - K(0.5), z = p.z * K(n) - K(0.5);
    REAL xf = R_FLOOR(x), yf = R_FLOOR(y), zf = R_FLOOR(z);
    REAL fx = x - xf, fy = y - yf, fz = z - zf;
    int i0 = wrapi((int)xf, n), i1 = wrapi((int)xf + 1, n);
    int j0 = wrapi((int)yf, n), j1 = wrapi((int)yf + 1, n);
    int k0 = wrapi((int)zf, n), k1 = wrapi((int)zf + 1, n);
    const uint8_t *d = t->shape;
#define S3(i, j, k) (K(d[((k) * n + (j)) * n + (i)]) / K(255.0))
    REAL c00 = r_mix(S3(i0, j0, k0), S3(i1, j0, k0), fx);
    REAL c10 = r_mix(S3(i0, j1, k0), S3(i1, j1, k0), fx);
    REAL c01 = r_mix(S3(i0, j0, k1), S3(i1, j0, k1), fx);
    REAL c11 = r_mix(S3(i0, j1, k1), S3(i1, j1, k1), fx);
#undef S3
    return r_mix(r_mix(c00, c10, fy), r_mix(c01, c11, fy), fz);
}

/* Inverse of the Vulkan cube face table: point on face f at face-plane (sc,tc), major-axis magnitude ma. */
static void cube_face_point(int f, double sc, double tc, double ma, double *o) {
    switch (f) {
    case 0: o[0] = ma;  o[1] = -tc; o[2] = -sc; break; /* +X: sc=-z tc=-y */
    case 1: o[0] = -ma; o[1] = -tc; o[2] = sc;  break; /* -X: sc=+z tc=-y */
    case 2: o[0] = sc;  o[1] = ma;  o[2] = tc;  break; /* +Y: sc=+x tc=+z */
    case 3: o[0] = sc;  o[1] = -ma; o[2] = -tc; break; /* -Y: sc=+x tc=-z */
    case 4: o[0] = sc;  o[1] = -tc; o[2] = ma;  break; /* +Z: sc=+x tc=-y */
    default: o[0] = -sc; o[1] = -tc; o[2] = -ma; break; /* -Z: sc=-x tc=-y */
    }
}

/* Vulkan face selection (z wins ties over y over x); returns face, writes sc, tc, |ma|. */
static int cube_select_d(const double *d, double *sc, double *tc, double *ma) {
    double ax = fabs(d[0]), ay = fabs(d[1]), az = fabs(d[2]);
    if (az >= ax && az >= ay) {
        *ma = az;
        if (d[2] >= 0) { *sc = d[0]; *tc = -d[1]; return 4; }
        *sc = -d[0]; *tc = -d[1]; return 5;
    }
    if (ay >= ax) {
        *ma = ay;
        if (d[1] >= 0) { *sc = d[0]; *tc = d[2]; return 2; }
        *sc = d[0]; *tc = -d[2]; return 3;
    }
    *ma = ax;
    if (d[0] >= 0) { *sc = -d[2]; *tc = -d[1]; return 0; }
    *sc = d[2]; *tc = -d[1]; return 1;
}

/* one mip level of the cubemap: 6 faces of n^2 texels */
typedef struct { const uint8_t *faces; int n; } CubeLevel;

static CubeLevel cube_level(const OracleTextures *t, int level) {
    CubeLevel L;
    size_t off = 0;
    for (int l = 0; l < level; ++l) { size_t m = (size_t)(t->cube_n >> l); off += 6 * m * m; }
    L.faces = t->cubemap + off;
    L.n = t->cube_n >> level;
    return L;
}

static int cube_texel_raw(CubeLevel L, int f, int i, int j) {
    return L.faces[(f * L.n + j) * L.n + i];
}

/* Texel across ONE edge: fold (i,j) (exactly one of them is -1 or n) over that edge. */
static int cube_texel_fold(CubeLevel t, int f, int i, int j) {
    const int n = t.n;
    int ci = clampi(i, 0, n - 1), cj = clampi(j, 0, n - 1);
    /* face-plane coordinates of the clamped texel centre, pushed onto the edge in the folded axis */
    double sc = (i < 0) ? -1.0 : (i >= n) ? 1.0 : (2.0 * (ci + 0.5) / n - 1.0);
    double tc = (j < 0) ? -1.0 : (j >= n) ? 1.0 : (2.0 * (cj + 0.5) / n - 1.0);
    double p[3], s2, t2, ma;
    cube_face_point(f, sc, tc, 1.0 - 1.0 / n, p);
    int f2 = cube_select_d(p, &s2, &t2, &ma);
    int i2 = clampi((int)floor((s2 / ma + 1.0) * 0.5 * n), 0, n - 1);
    int j2 = clampi((int)floor((t2 / ma + 1.0) * 0.5 * n), 0, n - 1);
    return cube_texel_raw(t, f2, i2, j2);
}

/* Seamless texel fetch, (i,j) in [-1, n]; returns byte value (corner: mean of 3 rounded to nearest). */
static int cube_texel(CubeLevel t, int f, int i, int j) {
    const int n = t.n;
    int oi = (i < 0 || i >= n), oj = (j < 0 || j >= n);
    if (!oi && !oj) return cube_texel_raw(t, f, i, j);
    if (oi && oj) {
        int ci = clampi(i, 0, n - 1), cj = clampi(j, 0, n - 1);
        int a = cube_texel_raw(t, f, ci, cj);
        int b = cube_texel_fold(t, f, i, cj);
        int c = cube_texel_fold(t, f, ci, j);
        return (a + b + c + 1) / 3;
    }
    return cube_texel_fold(t, f, i, j);
}

/* Vulkan face selection of `dir` (z wins ties over y over x): face, sc, tc, ma = |major component| */
static int cube_select(vec3 dir, REAL *sc, REAL *tc, REAL *ma) {
    REAL ax = R_FABS(dir.x), ay = R_FABS(dir.y), az = R_FABS(dir.z);
    if (az >= ax && az >= ay) {
        *ma = az;
        if (dir.z >= K(0.0)) { *sc = dir.x; *tc = -dir.y; return 4; }
        *sc = -dir.x; *tc = -dir.y; return 5;
    }
    if (ay >= ax) {
        *ma = ay;
        if (dir.y >= K(0.0)) { *sc = dir.x; *tc = dir.z; return 2; }
        *sc = dir.x; *tc = -dir.z; return 3;
    }
    *ma = ax;
    if (dir.x >= K(0.0)) { *sc = -dir.z; *tc = -dir.y; return 0; }
    *sc = dir.z; *tc = -dir.y; return 1;
}

/* (sc, tc, signed major) of an arbitrary vector in the frame of face f: the linear maps cube_select applies on that face */
static void cube_face_frame(int f, vec3 v, REAL *sc, REAL *tc, REAL *ma) {
    switch (f) {
    case 0: *sc = -v.z; *tc = -v.y; *ma = v.x; break;
    case 1: *sc = v.z;  *tc = -v.y; *ma = -v.x; break;
    case 2: *sc = v.x;  *tc = v.z;  *ma = v.y; break;
    case 3: *sc = v.x;  *tc = -v.z; *ma = -v.y; break;
    case 4: *sc = v.x;  *tc = -v.y; *ma = v.z; break;
    default: *sc = -v.x; *tc = -v.y; *ma = -v.z; break;
    }
}

/* bilinear, seamless sample of one level at face coordinates (s, tt) in [0,1] */
static REAL cube_bilinear(CubeLevel L, int f, REAL s, REAL tt) {
    const int n = L.n;
    REAL x = s * K(n) - K(0.5), y = tt * K(n) - K(0.5);
    REAL xf = R_FLOOR(x), yf = R_FLOOR(y);
    REAL fx = x - xf, fy = y - yf;
    int i0 = clampi((int)xf, -1, n - 1), j0 = clampi((int)yf, -1, n - 1);
    REAL t00 = K(cube_texel(L, f, i0, j0)) / K(255.0);
    REAL t10 = K(cube_texel(L, f, i0 + 1, j0)) / K(255.0);
    REAL t01 = K(cube_texel(L, f, i0, j0 + 1)) / K(255.0);
    REAL t11 = K(cube_texel(L, f, i0 + 1, j0 + 1)) / K(255.0);
    return r_mix(r_mix(t00, t10, fx), r_mix(t01, t11, fx), fy);
}

/* texture(samplerCube, dir).r on an R8 cubemap, LOD 0, bilinear, seamless */
static REAL sample_cube(const OracleTextures *t, vec3 dir) {
    if (t->cubemap == NULL) return K(1.0);
    REAL sc, tc, ma;
    int f = cube_select(dir, &sc, &tc, &ma);
    REAL s = K(0.5) * (sc / ma + K(1.0));
    REAL tt = K(0.5) * (tc / ma + K(1.0));
    return cube_bilinear(cube_level(t, 0), f, s, tt);
}

/*
 * texture(samplerCube, dir).r with the IMPLICIT level of detail of a linear-mipmap sampler (cloud_funcs.gdshaderinc:15,45;
 * engine behaviour, stated convention of this build -- OracleConfig.cube_lod = 1):
 *   - derivatives are finite differences inside the 2x2 pixel quad: dir_x / dir_y are the cube directions the horizontal
 *     / vertical quad neighbour passes to the SAME texture() call (same march step, same light tap); a neighbour that
 *     does not reach the call (outside the viewport, discarded, cloud gates false) contributes a zero derivative;
 *   - they are transformed to the face selected by `dir` (Vulkan 1.3, "Cube Map Derivative Transformation"), written in
 *     the cancellation-free exact form  s' - s = 0.5 (dsc ma - sc dma) / (ma ma'),  d* = neighbour - self in the face frame;
 *   - rho_x^2 = n^2 (ds_x^2 + dt_x^2), likewise y; lambda = 0.5 log2(max(rho_x^2, rho_y^2)) clamped to [0, levels - 1];
 *   - the result is mix(level floor(lambda), level floor(lambda) + 1, fract(lambda)), each level sampled bilinear + seamless.
 */
static REAL sample_cube_lod(const OracleTextures *t, vec3 dir, const vec3 *nb_dir, const int *nb_valid) {
    if (t->cubemap == NULL) return K(1.0);
    const int levels = t->cube_mips > 0 ? t->cube_mips : 1;
    REAL sc, tc, ma;
    int f = cube_select(dir, &sc, &tc, &ma);
    REAL s = K(0.5) * (sc / ma + K(1.0));
    REAL tt = K(0.5) * (tc / ma + K(1.0));
    REAL rho2 = K(0.0);
    for (int a = 0; a < 2; ++a) {
        if (!nb_valid[a]) continue;
        vec3 dv = v3_sub(nb_dir[a], dir);
        REAL dsc, dtc, dma;
        cube_face_frame(f, dv, &dsc, &dtc, &dma);
        REAL ma2 = ma + dma;
        if (!(ma2 > K(0.0))) continue; /* neighbour direction beyond the face's half space: no usable derivative */
        REAL inv = K(0.5) / (ma * ma2);
        REAL ds = (dsc * ma - sc * dma) * inv;
        REAL dt = (dtc * ma - tc * dma) * inv;
        REAL r2 = (ds * ds + dt * dt) * (K(t->cube_n) * K(t->cube_n));
        rho2 = r_max(rho2, r2);
    }
    REAL lambda = rho2 > K(0.0) ? K(0.5) * R_LOG2(rho2) : K(0.0);
    lambda = r_clamp(lambda, K(0.0), K(levels - 1));
    REAL lf = R_FLOOR(lambda);
    int lo = (int)lf, hi = lo + 1 < levels ? lo + 1 : lo;
    REAL fr = lambda - lf;
    REAL v0 = cube_bilinear(cube_level(t, lo), f, s, tt);
    if (hi == lo || fr == K(0.0)) return v0;
    REAL v1 = cube_bilinear(cube_level(t, hi), f, s, tt);
    return r_mix(v0, v1, fr);
}

/* ---- atmosphere (v2) ------------------------------------------------------------------- */

/* ref: I/atmosphere_funcs_v2.gdshaderinc:14-29 */
static REAL get_baked_optical_depth(const Ctx *c, vec3 pos, vec3 dir, vec3 planet_center) {
    REAL height = v3_length(v3_sub(pos, planet_center)) - c->planet_radius;
    REAL height_ratio = r_clamp(height / c->atmosphere_height, K(0.0), K(1.0));
    vec3 up = v3_normalize(v3_sub(pos, planet_center));
    REAL uvx = K(0.5) + K(0.5) * v3_dot(up, dir);
    return sample_lut(&c->tex, uvx, height_ratio);
}

/*
 * "Direct" light mode (not in the shipped shaders; SURVEY.md 8d config 1, "N view x M light steps"):
 * the quantity the LUT tabulates -- ref: optical_depth.gdshader:17-31 (get_optical_depth) with the
 * chord of :56-65 -- evaluated inline from the 3-D sample position with `steps` left-Riemann samples.
 */
static REAL get_marched_optical_depth(const Ctx *c, vec3 pos, vec3 dir, vec3 planet_center, int steps) {
    vec2 rs = ray_sphere(planet_center, c->planet_radius + c->atmosphere_height, pos, dir);
    REAL ray_len = rs.y - r_max(rs.x, K(0.0));
    REAL step_len = ray_len / K(steps);
    REAL optical_depth = K(0.0);
    for (int i = 0; i < steps; ++i) {
        vec3 p = v3_add(pos, v3_scale(v3_scale(dir, step_len), K(i)));
        REAL d = v3_length(v3_sub(p, planet_center));
        REAL density = get_atmosphere_density(c, d);
        optical_depth += density * step_len * c->density;
    }
    return optical_depth;
}

/* ref: I/atmosphere_funcs_v2.gdshaderinc:32-101 */
static vec4 compute_atmosphere_v2(const Ctx *c, vec3 ray_origin, vec3 ray_dir, vec3 planet_center,
                                  REAL t_begin, REAL t_end, vec3 sun_dir, REAL jitter) {
    const int steps = c->cfg.view_steps;
    vec3 coeff;
    coeff.x = pow4(K(400.0) / c->wavelengths.x) * c->scattering_strength;
    coeff.y = pow4(K(400.0) / c->wavelengths.y) * c->scattering_strength;
    coeff.z = pow4(K(400.0) / c->wavelengths.z) * c->scattering_strength;

    REAL step_len = (t_end - t_begin) / K(steps);
    vec3 total_light = v3(K(0.0), K(0.0), K(0.0));
    REAL view_ray_optical_depth = K(0.0);
    REAL alpha = K(0.0);
    vec3 pos0 = v3_add(ray_origin, v3_scale(ray_dir, t_begin));
    vec3 pos = pos0;

    for (int i = 0; i < steps; ++i) {
        REAL sun_ray_optical_depth = (c->cfg.light_steps > 0)
            ? get_marched_optical_depth(c, pos, sun_dir, planet_center, c->cfg.light_steps)
            : get_baked_optical_depth(c, pos, sun_dir, planet_center);

        REAL height = v3_length(v3_sub(pos, planet_center));
        REAL local_density = get_atmosphere_density(c, height) * c->density;
        view_ray_optical_depth += local_density * step_len;

        REAL od = sun_ray_optical_depth + view_ray_optical_depth;
        vec3 transmittance = v3(R_EXP(-od * coeff.x), R_EXP(-od * coeff.y), R_EXP(-od * coeff.z));

        total_light.x += local_density * step_len * transmittance.x * coeff.x;
        total_light.y += local_density * step_len * transmittance.y * coeff.y;
        total_light.z += local_density * step_len * transmittance.z * coeff.z;

        REAL vtransmittance = R_EXP(-local_density * step_len);
        alpha += (K(1.0) - vtransmittance) * (K(1.0) - alpha);

        pos = v3_add(pos, v3_scale(ray_dir, step_len));
    }

    total_light.x = r_clamp(total_light.x + c->ambient.x, K(0.0), K(1.0));
    total_light.y = r_clamp(total_light.y + c->ambient.y, K(0.0), K(1.0));
    total_light.z = r_clamp(total_light.z + c->ambient.z, K(0.0), K(1.0));

    alpha = r_clamp(alpha + jitter * K(0.02), K(0.0), K(0.99));

    vec4 r;
    r.x = total_light.x * c->modulate.x;
    r.y = total_light.y * c->modulate.y;
    r.z = total_light.z * c->modulate.z;
    r.w = alpha;
    return r;
}

/* ---- atmosphere (v1 "lite") ------------------------------------------------------------- */

/* ref: I/atmosphere_funcs_v1.gdshaderinc:15-45 */
static REAL get_atmo_factor(const Ctx *c, vec3 ray_origin, vec3 ray_dir, vec3 planet_center,
                            REAL t_begin, REAL t_end, vec3 sun_dir, REAL *light_factor) {
    const int steps = c->cfg.view_steps;
    REAL inv_steps = K(1.0) / K(steps);
    REAL step_len = (t_end - t_begin) * inv_steps;
    vec3 stepv = v3_scale(ray_dir, step_len);
    vec3 pos = v3_add(ray_origin, v3_scale(ray_dir, t_begin));
    REAL factor = K(1.0);
    REAL light_sum = K(0.0);
    for (int i = 0; i < steps; ++i) {
        vec3 rel = v3_sub(pos, planet_center);
        REAL d = v3_length(rel);
        vec3 up = v3(rel.x / d, rel.y / d, rel.z / d);
        REAL density = get_atmosphere_density(c, d);
        REAL light = r_clamp(K(1.2) * v3_dot(sun_dir, up) + K(0.5), K(0.0), K(1.0));
        light = light * light;
        light_sum += light * inv_steps;
        factor *= (K(1.0) - density * step_len);
        pos = v3_add(pos, stepv);
    }
    *light_factor = light_sum;
    return K(1.0) - factor;
}

/* ref: I/atmosphere_funcs_v1.gdshaderinc:48-63 */
static vec4 compute_atmosphere_v1(const Ctx *c, vec3 ray_origin, vec3 ray_dir, vec3 planet_center,
                                  REAL t_begin, REAL t_end, vec3 sun_dir) {
    REAL light_factor;
    REAL atmo_factor = get_atmo_factor(c, ray_origin, ray_dir, planet_center, t_begin, t_end, sun_dir, &light_factor);
    vec3 night_col = v3(r_mix(c->night0.x, c->night1.x, atmo_factor), r_mix(c->night0.y, c->night1.y, atmo_factor),
                        r_mix(c->night0.z, c->night1.z, atmo_factor));
    vec3 day_col = v3(r_mix(c->day0.x, c->day1.x, atmo_factor), r_mix(c->day0.y, c->day1.y, atmo_factor),
                      r_mix(c->day0.z, c->day1.z, atmo_factor));
    REAL day_factor = r_clamp(light_factor * c->day_night_transition_scale, K(0.0), K(1.0));
    vec4 r;
    r.x = r_mix(night_col.x, day_col.x, day_factor);
    r.y = r_mix(night_col.y, day_col.y, day_factor);
    r.z = r_mix(night_col.z, day_col.z, day_factor);
    r.w = r_clamp(atmo_factor, K(0.0), K(1.0));
    return r;
}

/* ---- clouds ---------------------------------------------------------------------------- */

/* ref: I/cloud_funcs.gdshaderinc:18-23 */
typedef struct { REAL bottom_height, top_height, density_scale, ground_height; } CloudSettings;

/* ref: I/cloud_funcs.gdshaderinc:25-29 */
static REAL height_curve(REAL x) { return K(1.0) - pow2(K(2.0) * x - K(1.0)); }

/* ref: I/cloud_funcs.gdshaderinc:31-68 with CLOUDS_ALWAYS_LOW_QUALITY forced (main:49) => detail = 0.5 */
/* The 2x2-quad neighbours of the shaded pixel for the implicit cubemap LOD: their sample position at the current call */
typedef struct { int valid[2]; vec3 pos[2]; } QuadNb;

static REAL get_density(const Ctx *c, vec3 pos_world, const CloudSettings *s, const QuadNb *nb) {
    REAL height = v3_length(pos_world) - s->bottom_height;
    REAL height_ratio = height / (s->top_height - s->bottom_height);
    REAL hc = r_max(height_curve(height_ratio), K(0.0));

    /* mat2 * vec2, column-major: (m0*x + m2*z, m1*x + m3*z) */
    REAL cx = c->cov_rot[0] * pos_world.x + c->cov_rot[2] * pos_world.z;
    REAL cy = c->cov_rot[1] * pos_world.x + c->cov_rot[3] * pos_world.z;
    REAL coverage;
    if (nb != NULL && c->cfg.cube_lod) {
        vec3 nd[2];
        for (int a = 0; a < 2; ++a) {
            vec3 q = nb->pos[a];
            nd[a] = v3(c->cov_rot[0] * q.x + c->cov_rot[2] * q.z, q.y, c->cov_rot[1] * q.x + c->cov_rot[3] * q.z);
        }
        coverage = sample_cube_lod(&c->tex, v3(cx, pos_world.y, cy), nd, nb->valid);
    } else {
        coverage = sample_cube(&c->tex, v3(cx, pos_world.y, cy));
    }
    coverage = coverage - K(0.25) * height_ratio + c->cloud_coverage_bias;

    REAL shape = r_mix(K(0.5), sample_shape(&c->tex, v3_scale(pos_world, c->cloud_shape_scale)), c->cloud_shape_factor);
    REAL detail = K(0.5);
    if (c->cloud_shape_invert == K(1.0)) shape = K(1.0) - shape;

    REAL density = (shape - K(0.2) * detail + r_mix(K(-1.2), K(1.5), coverage)) * hc;
    density = density * K(50.0) - K(20.0);
    return r_clamp(density, K(0.0), K(1.0));
}

/* ref: I/cloud_funcs.gdshaderinc:78-90 */
static REAL get_planet_shadow(vec3 pos, vec3 sun_dir) {
    vec3 n = v3_normalize(pos);
    REAL dp = -(n.x * sun_dir.x) + -(n.y * sun_dir.y) + -(n.z * sun_dir.z); /* dot(n, -sun_dir) */
    return r_smoothstep(K(-0.3), K(0.3), dp);
}

/* ref: I/cloud_funcs.gdshaderinc:92-102 */
static REAL get_light_cheap(vec3 pos_world, vec3 ray_dir, vec3 sun_dir, REAL alpha, const CloudSettings *s) {
    REAL height = v3_length(pos_world) - s->bottom_height;
    REAL height_ratio = height / (s->top_height - s->bottom_height);
    REAL light = height_ratio;
    REAL dp = v3_dot(ray_dir, sun_dir);
    REAL p16 = K(0.0);
    if (dp > K(0.0)) { REAL p2 = dp * dp, p4 = p2 * p2, p8 = p4 * p4; p16 = p8 * p8; }
    return light + r_max(p16, K(0.0)) * (K(1.0) - alpha);
}

/* ref: I/cloud_funcs.gdshaderinc:104-151 */
static REAL get_light_raymarched(const Ctx *c, vec3 pos0, vec3 sun_dir, const CloudSettings *s, const QuadNb *nb) {
    const int steps = 6;
    REAL reach = (s->top_height - s->bottom_height) * K(0.15);
    REAL pos0_height = v3_length(pos0) - s->bottom_height;
    REAL pos0_height_ratio = pos0_height / (s->top_height - s->bottom_height);
    REAL inv_steps = K(1.0) / K(steps);
    REAL step_len = reach * inv_steps;
    REAL alpha = K(0.0);
    for (int i = 0; i < steps; ++i) {
        vec3 pos = v3_add(pos0, v3_scale(sun_dir, K(i) * step_len));
        QuadNb tap, *tp = NULL;
        if (nb != NULL) { /* the quad neighbours evaluate the same tap from their own pos0 */
            tap = *nb;
            for (int a = 0; a < 2; ++a) tap.pos[a] = v3_add(nb->pos[a], v3_scale(sun_dir, K(i) * step_len));
            tp = &tap;
        }
        REAL density = get_density(c, pos, s, tp); /* both alpha0 branches identical (detail forced low) */
        density *= step_len * s->density_scale;
        REAL transmittance = R_EXP(-density);
        alpha += (K(1.0) - transmittance) * (K(1.0) - alpha);
        step_len *= K(1.2);
    }
    REAL light0 = pos0_height_ratio * K(0.2);
    return r_mix(K(1.0), light0, alpha);
}

/* ref: I/cloud_funcs.gdshaderinc:153-167 */
static REAL get_light(const Ctx *c, vec3 pos, vec3 ray_dir, vec3 sun_dir, REAL alpha, const CloudSettings *s, const QuadNb *nb) {
    REAL light = c->cfg.cloud_light_rm ? get_light_raymarched(c, pos, sun_dir, s, nb)
                                       : get_light_cheap(pos, ray_dir, sun_dir, alpha, s);
    REAL shadow_amount = get_planet_shadow(pos, sun_dir);
    light = light * r_mix(K(1.0), K(0.002), shadow_amount);
    return light;
}

/* ref: I/cloud_funcs.gdshaderinc:175-247 */
/* where a pixel's cloud march starts and how it advances (model space); valid = 0 when the pixel does not march */
typedef struct { int valid; vec3 pos0, dd; } MarchRay;

static MarchRay cloud_march_ray(const Ctx *c, vec3 ray_origin, vec3 ray_dir, REAL t_begin, REAL t_end, REAL jitter,
                                const CloudSettings *s) {
    /* the first lines of raymarch_cloud (clouds:186-213), for a quad neighbour */
    const int steps = c->cfg.cloud_steps;
    REAL march_distance_space = K(0.5) * R_SQRT(K(1.0) - pow2(s->ground_height / s->top_height)) * s->bottom_height;
    REAL march_distance_ground = K(3.0) * march_distance_space;
    REAL max_d = r_mix(march_distance_ground, march_distance_space,
                       r_smoothstep(s->bottom_height, s->top_height * K(1.05), v3_length(ray_origin)));
    t_end = t_begin + r_min(t_end - t_begin, max_d);
    REAL step_len = (t_end - t_begin) * (K(1.0) / K(steps));
    MarchRay m;
    m.valid = 1;
    m.pos0 = v3_add(v3_add(ray_origin, v3_scale(ray_dir, jitter * step_len)), v3_scale(ray_dir, t_begin));
    m.dd = v3_scale(ray_dir, step_len);
    return m;
}

static vec2 raymarch_cloud(const Ctx *c, vec3 ray_origin, vec3 ray_dir, REAL t_begin, REAL t_end,
                           REAL jitter, vec3 sun_dir, const CloudSettings *s, const MarchRay *nbray) {
    const int steps = c->cfg.cloud_steps;
    REAL march_distance_space = K(0.5) * R_SQRT(K(1.0) - pow2(s->ground_height / s->top_height)) * s->bottom_height;
    REAL march_distance_ground = K(3.0) * march_distance_space;
    REAL tmin = s->bottom_height;
    REAL tmax = s->top_height * K(1.05);
    REAL max_d = r_mix(march_distance_ground, march_distance_space,
                       r_smoothstep(tmin, tmax, v3_length(ray_origin)));
    t_end = t_begin + r_min(t_end - t_begin, max_d);

    REAL inv_steps = K(1.0) / K(steps);
    REAL step_len = (t_end - t_begin) * inv_steps;
    REAL total_transmittance = K(1.0);
    REAL total_light = K(0.0);
    REAL alpha = K(0.0);
    /* ray_origin + jitter * step_len * ray_dir + ray_dir * t_begin */
    vec3 pos = v3_add(v3_add(ray_origin, v3_scale(ray_dir, jitter * step_len)), v3_scale(ray_dir, t_begin));

    QuadNb nb, *nbp = NULL;
    if (nbray != NULL && c->cfg.cube_lod) {
        for (int a = 0; a < 2; ++a) { nb.valid[a] = nbray[a].valid; nb.pos[a] = nbray[a].pos0; }
        nbp = &nb;
    }
    for (int i = 0; i < steps; ++i) {
        REAL light = get_light(c, pos, ray_dir, sun_dir, alpha, s, nbp);
        REAL density = get_density(c, pos, s, nbp);
        density *= s->density_scale;
        REAL transmittance = R_EXP(-density * step_len);
        total_transmittance *= transmittance;
        total_transmittance = r_max(total_transmittance, K(0.005));
        total_light += light * density * step_len * total_transmittance;
        alpha += (K(1.0) - transmittance) * (K(1.0) - alpha);
        pos = v3_add(pos, v3_scale(ray_dir, step_len));
        if (nbp != NULL)
            for (int a = 0; a < 2; ++a) nb.pos[a] = v3_add(nb.pos[a], nbray[a].dd);
    }
    vec2 r = {total_light, alpha};
    return r;
}

/* ref: I/cloud_funcs.gdshaderinc:249-324 */
/* The part of render_clouds (clouds:249-301) in front of the march: shell hits, the gate, the transform to model space.
 * Returns 0 when the pixel does not march. */
typedef struct { vec3 origin, dir, sun; vec2 cloud_rs; CloudSettings cs; } CloudRay;

static int cloud_gate(const Ctx *c, vec3 planet_center_vs, vec3 ray_origin, vec3 ray_dir, REAL linear_depth,
                      const REAL *inv_view, vec3 sun_dir, CloudRay *out) {
    REAL clouds_bottom = c->planet_radius + c->cloud_bottom * c->atmosphere_height;
    REAL clouds_top = c->planet_radius + c->cloud_top * c->atmosphere_height;
    vec2 rs_top = ray_sphere(planet_center_vs, clouds_top, ray_origin, ray_dir);
    if (rs_top.x != rs_top.y) {
        vec2 rs_bottom = ray_sphere(planet_center_vs, clouds_bottom, ray_origin, ray_dir);
        vec2 cloud_rs = rs_top;
        cloud_rs.x = r_max(cloud_rs.x, K(0.0));
        cloud_rs.y = r_min(cloud_rs.y, linear_depth);
        if (cloud_rs.x < linear_depth && (linear_depth > rs_bottom.y || rs_bottom.x > K(0.0))) {
            REAL m[16];
            m4_mul_m4(c->world_to_model, inv_view, m);
            vec4 o4 = {ray_origin.x, ray_origin.y, ray_origin.z, K(1.0)};
            vec4 d4 = {ray_dir.x, ray_dir.y, ray_dir.z, K(0.0)};
            vec4 s4 = {sun_dir.x, sun_dir.y, sun_dir.z, K(0.0)};
            vec4 ow = m4_mul_v4(m, o4), dw = m4_mul_v4(m, d4), sw = m4_mul_v4(m, s4);
            out->origin = v3(ow.x, ow.y, ow.z);
            out->dir = v3(dw.x, dw.y, dw.z);
            out->sun = v3(sw.x, sw.y, sw.z);
            out->cloud_rs = cloud_rs;
            out->cs.bottom_height = clouds_bottom;
            out->cs.top_height = clouds_top;
            out->cs.density_scale = c->cloud_density_scale;
            out->cs.ground_height = c->planet_radius;
            return 1;
        }
    }
    return 0;
}

/* ref: I/cloud_funcs.gdshaderinc:249-324 */
static void render_clouds(const Ctx *c, vec3 *out_albedo, REAL *out_alpha, vec3 planet_center_vs,
                          vec3 ray_origin, vec3 ray_dir, REAL linear_depth, const REAL *inv_view,
                          vec3 sun_dir, REAL jitter, const MarchRay *nbray) {
    CloudRay cr;
    if (cloud_gate(c, planet_center_vs, ray_origin, ray_dir, linear_depth, inv_view, sun_dir, &cr)) {
        vec2 rr = raymarch_cloud(c, cr.origin, cr.dir, cr.cloud_rs.x, cr.cloud_rs.y, jitter, cr.sun, &cr.cs, nbray);
        REAL cl = rr.x, ca = rr.y;
        vec4 self = {out_albedo->x, out_albedo->y, out_albedo->z, *out_alpha};
        vec4 over = {cl, cl, cl, ca};
        vec4 ab = blend_colors(self, over);
        vec4 add = {out_albedo->x + cl * ca, out_albedo->y + cl * ca, out_albedo->z + cl * ca,
                    r_max(*out_alpha, ca)};
        out_albedo->x = r_mix(ab.x, add.x, c->cloud_blend);
        out_albedo->y = r_mix(ab.y, add.y, c->cloud_blend);
        out_albedo->z = r_mix(ab.z, add.z, c->cloud_blend);
        *out_alpha = r_mix(ab.w, add.w, c->cloud_blend);
    }
}

/* ---- fragment driver ------------------------------------------------------------------- */

/* The per-pixel set-up of atmosphere_fragment (main:128-169): everything in front of compute_atmosphere. */
typedef struct { int hit; vec3 ray_origin, ray_dir, center, sun_dir; REAL t_begin, t_end, linear_depth, jitter; } FragSetup;

static FragSetup fragment_setup(const Ctx *c, const OracleFrame *f, const REAL *inv_p, const REAL *inv_v,
                                int px, int py, REAL nonlinear_depth) {
    FragSetup o;
    REAL vw = K(f->viewport_w), vh = K(f->viewport_h);
    REAL uvx = (K(px) + K(0.5)) / vw, uvy = (K(py) + K(0.5)) / vh;
    vec4 ndc = {uvx * K(2.0) - K(1.0), uvy * K(2.0) - K(1.0), nonlinear_depth, K(1.0)};
    vec4 view_coords = m4_mul_v4(inv_p, ndc);
    vec4 world_coords = m4_mul_v4(inv_v, view_coords);
    vec3 pos_world = v3(world_coords.x / world_coords.w, world_coords.y / world_coords.w, world_coords.z / world_coords.w);
    vec4 origin4 = {K(0.0), K(0.0), K(0.0), K(1.0)};
    vec4 cam4 = m4_mul_v4(inv_v, origin4);
    vec3 cam_pos_world = v3(cam4.x, cam4.y, cam4.z);
    REAL linear_depth = v3_length(v3_sub(cam_pos_world, pos_world));

    o.ray_origin = v3(K(0.0), K(0.0), K(0.0));
    o.ray_dir = v3_normalize(v3_sub(v3(view_coords.x, view_coords.y, view_coords.z), o.ray_origin));
    o.center = v3(f->planet_center_viewspace[0], f->planet_center_viewspace[1], f->planet_center_viewspace[2]);
    vec3 sun_c = v3(f->sun_center_viewspace[0], f->sun_center_viewspace[1], f->sun_center_viewspace[2]);

    REAL atmosphere_radius = c->planet_radius + c->atmosphere_height;
    vec2 rs_atmo = ray_sphere(o.center, atmosphere_radius, o.ray_origin, o.ray_dir);
    o.hit = rs_atmo.x != rs_atmo.y;
    o.t_begin = o.t_end = o.jitter = K(0.0);
    o.linear_depth = linear_depth;
    o.sun_dir = v3(K(0.0), K(0.0), K(0.0));
    if (o.hit) {
        o.t_begin = r_max(rs_atmo.x, K(0.0));
        o.t_end = r_max(rs_atmo.y, K(0.0));
        vec2 rs_ground = ray_sphere(o.center, c->planet_radius, o.ray_origin, o.ray_dir);
        REAL gd = K(10000000.0);
        if (rs_ground.x != rs_ground.y) gd = rs_ground.x;
        o.linear_depth = r_mix(linear_depth, gd, c->sphere_depth_factor);
        o.t_end = r_min(o.t_end, o.linear_depth);
        o.sun_dir = v3_normalize(v3_sub(sun_c, o.center));
        REAL jx = vw * uvx, jy = vh * uvy;
        int ji = ((int)jx) & 0xff, jj = ((int)jy) & 0xff;
        o.jitter = K(c->tex.blue_noise[jj * 256 + ji]) / K(255.0);
    }
    return o;
}

/* Where the cloud march of pixel (px, py) starts and how it advances: what a 2x2-quad neighbour contributes to the
 * implicit cubemap LOD (sample_cube_lod).  valid = 0: outside the viewport, discarded, or the cloud gates are false. */
static MarchRay pixel_march_ray(const Ctx *c, const OracleFrame *f, const REAL *inv_p, const REAL *inv_v,
                                int px, int py, const float *depth) {
    MarchRay m;
    memset(&m, 0, sizeof(m));
    if (px < 0 || py < 0 || px >= f->viewport_w || py >= f->viewport_h) return m;
    FragSetup fs = fragment_setup(c, f, inv_p, inv_v, px, py, K(depth[(size_t)py * f->viewport_w + px]));
    if (!fs.hit) return m;
    CloudRay cr;
    if (!cloud_gate(c, fs.center, fs.ray_origin, fs.ray_dir, fs.linear_depth, inv_v, fs.sun_dir, &cr)) return m;
    return cloud_march_ray(c, cr.origin, cr.dir, cr.cloud_rs.x, cr.cloud_rs.y, fs.jitter, &cr.cs);
}

/* ref: I/planet_atmosphere_main.gdshaderinc:106-197.  Returns 1 if the fragment is kept, 0 on discard. */
static int atmosphere_fragment(const Ctx *c, const OracleFrame *f, const REAL *inv_p, const REAL *inv_v,
                               int px, int py, const float *depth, REAL *rgba) {
    FragSetup fs = fragment_setup(c, f, inv_p, inv_v, px, py, K(depth[(size_t)py * f->viewport_w + px]));
    if (fs.hit) {
        /* main:172-179 */
        vec4 atmosphere = c->cfg.lite
            ? compute_atmosphere_v1(c, fs.ray_origin, fs.ray_dir, fs.center, fs.t_begin, fs.t_end, fs.sun_dir)
            : compute_atmosphere_v2(c, fs.ray_origin, fs.ray_dir, fs.center, fs.t_begin, fs.t_end, fs.sun_dir, fs.jitter);
        vec3 albedo = v3(atmosphere.x, atmosphere.y, atmosphere.z);
        REAL alpha = atmosphere.w;

        if (c->cfg.cloud_steps > 0) {
            MarchRay nb[2], *nbp = NULL;
            if (c->cfg.cube_lod && c->tex.cubemap != NULL && c->tex.cube_mips > 1) {
                /* the quad partners: (px ^ 1, py) and (px, py ^ 1), absolute viewport coordinates */
                nb[0] = pixel_march_ray(c, f, inv_p, inv_v, px ^ 1, py, depth);
                nb[1] = pixel_march_ray(c, f, inv_p, inv_v, px, py ^ 1, depth);
                nbp = nb;
            }
            render_clouds(c, &albedo, &alpha, fs.center, fs.ray_origin, fs.ray_dir, fs.linear_depth, inv_v, fs.sun_dir, fs.jitter, nbp);
        }

        rgba[0] = albedo.x; rgba[1] = albedo.y; rgba[2] = albedo.z; rgba[3] = alpha;
        return 1;
    }
    rgba[0] = rgba[1] = rgba[2] = rgba[3] = K(0.0);
    return 0;
}

/* ---- exported entry points ------------------------------------------------------------- */

typedef struct {
    const Ctx *ctx;
    const OracleFrame *frame;
    REAL inv_p[16], inv_v[16];
    const float *depth;
    REAL *out;
    int x0, y0, x1, y1; /* rect, output is (y1-y0) rows of (x1-x0) RGBA */
    int *next_row;      /* shared row counter: threads take rows dynamically (hit and miss rows cost very differently) */
    long hits;
} Job;

static void *render_rows(void *arg) {
    Job *j = (Job *)arg;
    const int rw = j->x1 - j->x0;
    long hits = 0;
    for (;;) {
        const int y = __atomic_fetch_add(j->next_row, 1, __ATOMIC_RELAXED);
        if (y >= j->y1) break;
        for (int x = j->x0; x < j->x1; ++x) {
            REAL *o = j->out + ((size_t)(y - j->y0) * rw + (x - j->x0)) * 4;
            hits += atmosphere_fragment(j->ctx, j->frame, j->inv_p, j->inv_v, x, y, j->depth, o);
        }
    }
    j->hits = hits;
    return NULL;
}

/*
 * Render the rect [x0,x1) x [y0,y1) of the viewport.  depth = full-viewport nonlinear depth buffer
 * (viewport_h rows of viewport_w floats).  out = (y1-y0)*(x1-x0) RGBA REAL, row-major.
 * nthreads workers take rows from a shared counter.  Returns the number of non-discarded pixels.
 */
long SFX(oracle_render)(const OracleParams *p, const OracleTextures *t, const OracleConfig *cfg,
                        const OracleFrame *f, const float *depth, REAL *out,
                        int x0, int y0, int x1, int y1, int nthreads) {
    Ctx ctx;
    ctx_init(&ctx, p, t, cfg);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    if (nthreads > (y1 - y0)) nthreads = (y1 - y0) > 0 ? (y1 - y0) : 1;
    Job *jobs = (Job *)malloc(sizeof(Job) * (size_t)nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    int next_row = y0;
    for (int k = 0; k < nthreads; ++k) {
        Job *j = &jobs[k];
        j->ctx = &ctx; j->frame = f; j->depth = depth; j->out = out;
        for (int i = 0; i < 16; ++i) { j->inv_p[i] = f->inv_projection_matrix[i]; j->inv_v[i] = f->inv_view_matrix[i]; }
        if (cfg->double_precision) {
            /* ref: I/planet_atmosphere_main.gdshaderinc:118-125 (#ifdef DOUBLE_PRECISION) */
            j->inv_v[12] *= K(-1.0); j->inv_v[13] *= K(-1.0); j->inv_v[14] *= K(-1.0);
        }
        j->x0 = x0; j->y0 = y0; j->x1 = x1; j->y1 = y1;
        j->next_row = &next_row;
        j->hits = 0;
    }
    if (nthreads == 1) {
        render_rows(&jobs[0]);
    } else {
        for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, render_rows, &jobs[k]);
        for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    }
    long hits = 0;
    for (int k = 0; k < nthreads; ++k) hits += jobs[k].hits;
    free(jobs);
    free(th);
    return hits;
}

/* ref: optical_depth.gdshader:17-31 */
static REAL bake_get_optical_depth(const Ctx *c, vec2 ray_origin, vec2 ray_dir, REAL ray_len, int steps) {
    REAL step_len = ray_len / K(steps);
    REAL optical_depth = K(0.0);
    for (int i = 0; i < steps; ++i) {
        vec2 pos;
        pos.x = ray_origin.x + ray_dir.x * step_len * K(i);
        pos.y = ray_origin.y + ray_dir.y * step_len * K(i);
        REAL d = R_SQRT(pos.x * pos.x + pos.y * pos.y);
        REAL density = get_atmosphere_density(c, d);
        optical_depth += density * step_len * c->density;
    }
    return optical_depth;
}

/*
 * ref: optical_depth.gdshader:45-68 (fragment) for every texel of a w x h target, UV = texel centre.
 * steps = 64 in the reference (:18).  Output: h rows of w floats (what FORMAT_RF reinterpretation of
 * the RGBA8 viewport yields, optical_depth_baker.gd:75-77), always fp32 -- the f64 twin rounds at the end.
 */
void SFX(oracle_bake_optical_depth)(float planet_radius, float atmosphere_height, float density,
                                    int w, int h, int steps, float *out) {
    OracleParams p;
    memset(&p, 0, sizeof(p));
    p.u_planet_radius = planet_radius;
    p.u_atmosphere_height = atmosphere_height;
    p.u_density = density;
    Ctx c;
    ctx_init(&c, &p, NULL, NULL);
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            REAL u = (K(i) + K(0.5)) / K(w), v = (K(j) + K(0.5)) / K(h);
            vec2 ray_dir;
            ray_dir.y = K(2.0) * u - K(1.0);
            ray_dir.x = R_SQRT(K(1.0) - ray_dir.y * ray_dir.y);
            REAL height_ratio = v;
            vec2 pos = {K(0.0), c.planet_radius + c.atmosphere_height * height_ratio};
            vec2 rs = ray_sphere(v3(K(0.0), K(0.0), K(0.0)), c.planet_radius + c.atmosphere_height,
                                 v3(pos.x, pos.y, K(0.0)), v3(ray_dir.x, ray_dir.y, K(0.0)));
            REAL distance_through_atmosphere = rs.y - r_max(rs.x, K(0.0));
            out[j * w + i] = (float)bake_get_optical_depth(&c, pos, ray_dir, distance_through_atmosphere, steps);
        }
}

/* ---- NoiseCubemap generator ------------------------------------------------------ */
#ifndef ORACLE_F64
/*
 * ref: /root/reference/addons/zylann.atmosphere/noise_cubemap.gd:101-140 (_generate_images): for every side and
 * texel, pos2d -> +X direction -> per-side swizzle (:110-128), density = 0.5 + 0.5 * noise(pos * scale) (:130),
 * stored as L8 (:107,134).  The reference's noise is Godot's FastNoiseLite (engine code, not in the tree): PARITY
 * UNPINNED there; this build defines its own "seeded value noise" below, and the restatement pins the
 * texel->direction mapping, the scale, the 0.5+0.5*n remap and the L8 store.  fp32 only, unfused, IEEE sqrt/div,
 * so the device kernel reproduces the bytes exactly.
 * Conventions stated by the build: Vector math in fp32; L8 store = (uint8) clamp(v * 255, 0, 255), truncating
 * (Godot's Image::set_pixel for FORMAT_L8); mipmaps are not generated (the path samples LOD 0 only).
 */
static inline uint32_t nz_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
static inline float nz_lattice(int32_t ix, int32_t iy, int32_t iz, uint32_t seed) {
    uint32_t h = ((uint32_t)ix * 0x9E3779B1u) ^ ((uint32_t)iy * 0x85EBCA77u) ^ ((uint32_t)iz * 0xC2B2AE3Du) ^ seed;
    return (float)(nz_hash(h) >> 8) * (1.0f / 16777216.0f);
}
static float nz_value(float px, float py, float pz, uint32_t seed) {
    float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    float tx = px - fx, ty = py - fy, tz = pz - fz;
    float wx = tx * tx * (3.0f - 2.0f * tx), wy = ty * ty * (3.0f - 2.0f * ty), wz = tz * tz * (3.0f - 2.0f * tz);
    int32_t x0 = (int32_t)fx, y0 = (int32_t)fy, z0 = (int32_t)fz, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    float c00 = nz_lattice(x0, y0, z0, seed) * (1.0f - wx) + nz_lattice(x1, y0, z0, seed) * wx;
    float c10 = nz_lattice(x0, y1, z0, seed) * (1.0f - wx) + nz_lattice(x1, y1, z0, seed) * wx;
    float c01 = nz_lattice(x0, y0, z1, seed) * (1.0f - wx) + nz_lattice(x1, y0, z1, seed) * wx;
    float c11 = nz_lattice(x0, y1, z1, seed) * (1.0f - wx) + nz_lattice(x1, y1, z1, seed) * wx;
    float c0 = c00 * (1.0f - wy) + c10 * wy;
    float c1 = c01 * (1.0f - wy) + c11 * wy;
    return c0 * (1.0f - wz) + c1 * wz;
}
/* Noise.get_noise_3d analogue: fractal value noise in [-1, 1] */
float oracle_noise_get_3d(float x, float y, float z, uint32_t seed, float frequency, int octaves, float gain) {
    float total = 0.0f, amp = 1.0f, norm = 0.0f, freq = frequency;
    for (int o = 0; o < octaves; ++o) {
        total += amp * nz_value(x * freq, y * freq, z * freq, seed + 1013u * (uint32_t)o);
        norm += amp;
        amp *= gain;
        freq *= 2.0f;
    }
    return 2.0f * (total / norm) - 1.0f;
}
/* ref: noise_cubemap.gd:110-128 -- direction of texel (x, y) of `side` */
void oracle_noise_cubemap_direction(int resolution, int side, int x, int y, float *dir3) {
    float half = 0.5f * (float)resolution;
    float p2x = ((float)x + 0.5f) / half - 1.0f;
    float p2y = ((float)(resolution - y - 1) + 0.5f) / half - 1.0f;
    float vx = 1.0f, vy = p2y, vz = -p2x;
    float len = sqrtf(vx * vx + vy * vy + vz * vz);
    vx /= len; vy /= len; vz /= len;
    float ox, oy, oz;
    switch (side) {
    case 0: ox = vx;  oy = vy;  oz = vz;  break;   /* +X */
    case 1: ox = -vx; oy = vy;  oz = -vz; break;   /* -X */
    case 2: ox = -vz; oy = vx;  oz = -vy; break;   /* +Y */
    case 3: ox = -vz; oy = -vx; oz = vy;  break;   /* -Y */
    case 4: ox = -vz; oy = vy;  oz = vx;  break;   /* +Z */
    default: ox = vz; oy = vy;  oz = -vx; break;   /* -Z */
    }
    dir3[0] = ox; dir3[1] = oy; dir3[2] = oz;
}
/* ref: noise_cubemap.gd:101-140.  out: 6 faces of resolution^2 bytes, row y, column x. */
void oracle_noise_cubemap(int resolution, uint32_t seed, float frequency, int octaves, float gain,
                          const float *scale3, uint8_t *out) {
    for (int side = 0; side < 6; ++side)
        for (int y = 0; y < resolution; ++y)
            for (int x = 0; x < resolution; ++x) {
                float d[3];
                oracle_noise_cubemap_direction(resolution, side, x, y, d);
                float n = oracle_noise_get_3d(d[0] * scale3[0], d[1] * scale3[1], d[2] * scale3[2], seed, frequency, octaves, gain);
                float density = 0.5f + 0.5f * n;
                float v = density * 255.0f;
                v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
                out[((size_t)side * resolution + y) * resolution + x] = (uint8_t)v;
            }
}
/* ref: noise_cubemap.gd:143-155 (_generate_importable_image): 3 x 2 atlas, side = x + 3*y */
void oracle_noise_cubemap_atlas(int resolution, const uint8_t *faces, uint8_t *atlas) {
    const int aw = 3 * resolution;
    for (int sy = 0; sy < 2; ++sy)
        for (int sx = 0; sx < 3; ++sx) {
            const uint8_t *f = faces + (size_t)(sx + sy * 3) * resolution * resolution;
            for (int y = 0; y < resolution; ++y)
                memcpy(atlas + ((size_t)(sy * resolution + y)) * aw + (size_t)sx * resolution, f + (size_t)y * resolution, (size_t)resolution);
        }
}
#endif

/* ---- single-function probes for the known-answer tests --------------------------------- */

void SFX(oracle_ray_sphere)(const REAL *center, REAL radius, const REAL *origin, const REAL *dir, REAL *out2) {
    vec2 r = ray_sphere(v3(center[0], center[1], center[2]), radius, v3(origin[0], origin[1], origin[2]),
                        v3(dir[0], dir[1], dir[2]));
    out2[0] = r.x; out2[1] = r.y;
}

REAL SFX(oracle_get_atmosphere_density)(float planet_radius, float atmosphere_height, float density, REAL height) {
    OracleParams p;
    memset(&p, 0, sizeof(p));
    p.u_planet_radius = planet_radius; p.u_atmosphere_height = atmosphere_height; p.u_density = density;
    Ctx c;
    ctx_init(&c, &p, NULL, NULL);
    return get_atmosphere_density(&c, height);
}

/* the direct light march (get_marched_optical_depth) for n samples: pos3 / dir3 arrays, planet centre at the origin */
void SFX(oracle_marched_optical_depth)(float planet_radius, float atmosphere_height, float density, int n, const REAL *pos3,
                                       const REAL *dir3, int steps, REAL *out) {
    OracleParams p;
    memset(&p, 0, sizeof(p));
    p.u_planet_radius = planet_radius; p.u_atmosphere_height = atmosphere_height; p.u_density = density;
    Ctx c;
    ctx_init(&c, &p, NULL, NULL);
    for (int i = 0; i < n; ++i)
        out[i] = get_marched_optical_depth(&c, v3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]),
                                           v3(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]), v3(K(0.0), K(0.0), K(0.0)), steps);
}

void SFX(oracle_blend_colors)(const REAL *self4, const REAL *over4, REAL *out4) {
    vec4 s = {self4[0], self4[1], self4[2], self4[3]}, o = {over4[0], over4[1], over4[2], over4[3]};
    vec4 r = blend_colors(s, o);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}

REAL SFX(oracle_sample_lut)(const OracleTextures *t, REAL u, REAL v) { return sample_lut(t, u, v); }
REAL SFX(oracle_sample_shape)(const OracleTextures *t, const REAL *p) { return sample_shape(t, v3(p[0], p[1], p[2])); }
REAL SFX(oracle_sample_cube)(const OracleTextures *t, const REAL *d) { return sample_cube(t, v3(d[0], d[1], d[2])); }
REAL SFX(oracle_sample_cube_lod)(const OracleTextures *t, const REAL *d, const REAL *nb6, const int *valid2) {
    vec3 nb[2] = {v3(nb6[0], nb6[1], nb6[2]), v3(nb6[3], nb6[4], nb6[5])};
    return sample_cube_lod(t, v3(d[0], d[1], d[2]), nb, valid2);
}
int SFX(oracle_cube_texel)(const OracleTextures *t, int f, int i, int j) { return cube_texel(cube_level(t, 0), f, i, j); }

REAL SFX(oracle_get_cloud_density)(const OracleParams *p, const OracleTextures *t, const REAL *pos_model) {
    OracleConfig cfg = {8, 8, 0, 0, 0, 0, 0};
    Ctx c;
    ctx_init(&c, p, t, &cfg);
    CloudSettings cs;
    cs.bottom_height = c.planet_radius + c.cloud_bottom * c.atmosphere_height;
    cs.top_height = c.planet_radius + c.cloud_top * c.atmosphere_height;
    cs.density_scale = c.cloud_density_scale;
    cs.ground_height = c.planet_radius;
    return get_density(&c, v3(pos_model[0], pos_model[1], pos_model[2]), &cs, NULL);
}

#ifndef ORACLE_F64
/* ref: optical_depth.gdshader:33-43 followed by the RGBA8 store: byte k of the float's bit pattern,
 * little-endian, as value/255 quantised back to the byte.  Returns the 4 stored bytes. */
void oracle_encode_float_to_viewport(float h, uint8_t *rgba8) {
    uint32_t u;
    memcpy(&u, &h, 4);
    for (int k = 0; k < 4; ++k) {
        float ch = (float)((u >> (8 * k)) & 255u) / 255.0f;       /* shader output colour */
        rgba8[k] = (uint8_t)(int)floorf(ch * 255.0f + 0.5f);       /* UNORM8 store */
    }
}
/* ref: optical_depth_baker.gd:75-77 -- reinterpret the RGBA8 bytes as FORMAT_RF */
float oracle_decode_viewport_to_float(const uint8_t *rgba8) {
    uint32_t u = (uint32_t)rgba8[0] | ((uint32_t)rgba8[1] << 8) | ((uint32_t)rgba8[2] << 16) | ((uint32_t)rgba8[3] << 24);
    float h;
    memcpy(&h, &u, 4);
    return h;
}
#endif
