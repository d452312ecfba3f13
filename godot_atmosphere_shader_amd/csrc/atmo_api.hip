// atmo_api.hip -- host side of the C ABI declared in include/atmo.h.
//
// Holds the uniform table (the reference's `shader_params` names), owns the device copies of the four
// textures, evaluates the per-frame (pixel-independent) expressions of the shader once per launch in
// fp32 in the reference's operation order (this file is built with -ffp-contract=off, so nothing is
// fused), and enqueues the gfx950 kernels of atmo_kernels.hip.  There is no CPU fallback: without a
// HIP device every compute entry point fails with ATMO_E_NO_DEVICE / ATMO_E_HIP.
#include "../../include/atmo.h"
#include "atmo_device.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

struct ParamDesc {
    const char *name;
    int count;
    size_t offset;
};

// The reference's uniforms with their GDShader defaults.  `source_color` defaults are stored linear
// (include/atmo.h: "colours are linear"), i.e. after the engine's sRGB -> linear conversion of the written value.
struct Params {
    float u_planet_radius = 1.0f;                            // planet_common.gdshaderinc:4
    float u_atmosphere_height = 0.1f;                        // :5
    float u_sun_position[3] = {0, 0, 0};                     // :6 (host-side only: feeds sun_center_viewspace)
    float u_density = 0.2f;                                  // atmosphere_common.gdshaderinc:10
    float u_scattering_strength = 20.0f;                     // atmosphere_funcs_v2.gdshaderinc:8
    float u_scattering_wavelengths[3] = {700, 530, 440};     // :9
    float u_atmosphere_modulate[3] = {1, 1, 1};              // :10
    float u_atmosphere_ambient_color[3] = {0, 0, 0.002f / 12.92f};  // :11 `source_color` vec3(0,0,0.002), linear
    float u_clip_mode = 0.0f;                                // main:55 (rasteriser only; stored, unused)
    float u_sphere_depth_factor = 0.0f;                      // main:60
    float u_cloud_density_scale = 50.0f;                     // cloud_funcs.gdshaderinc:5
    float u_cloud_bottom = 0.2f;                             // :6
    float u_cloud_top = 0.5f;                                // :7
    float u_cloud_blend = 0.5f;                              // :8
    float u_world_to_model_matrix[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // :9
    float u_cloud_shape_invert = 0.0f;                       // :11
    float u_cloud_coverage_bias = 0.0f;                      // :12
    float u_cloud_shape_factor = 0.8f;                       // :13
    float u_cloud_shape_scale = 1.0f;                        // :14
    float u_cloud_coverage_rotation[4] = {1, 0, 0, 1};       // :16
    // atmosphere_funcs_v1.gdshaderinc:8-12 (`source_color` defaults, already linear; alpha unused by the shader)
    float u_day_color0[4] = {0.21404114f, 0.60382734f, 1.0f, 1.0f};
    float u_day_color1[4] = {0.21404114f, 0.60382734f, 1.0f, 1.0f};
    float u_night_color0[4] = {0.03310477f, 0.13286832f, 0.60382734f, 1.0f};
    float u_night_color1[4] = {0.03310477f, 0.13286832f, 0.60382734f, 1.0f};
    float u_day_night_transition_scale = 2.0f;
};

#define PD(field, n) {#field, n, offsetof(Params, field)}
const ParamDesc kParams[] = {
    PD(u_planet_radius, 1), PD(u_atmosphere_height, 1), PD(u_sun_position, 3), PD(u_density, 1),
    PD(u_scattering_strength, 1), PD(u_scattering_wavelengths, 3), PD(u_atmosphere_modulate, 3),
    PD(u_atmosphere_ambient_color, 3), PD(u_clip_mode, 1), PD(u_sphere_depth_factor, 1),
    PD(u_cloud_density_scale, 1), PD(u_cloud_bottom, 1), PD(u_cloud_top, 1), PD(u_cloud_blend, 1),
    PD(u_world_to_model_matrix, 16), PD(u_cloud_shape_invert, 1), PD(u_cloud_coverage_bias, 1),
    PD(u_cloud_shape_factor, 1), PD(u_cloud_shape_scale, 1), PD(u_cloud_coverage_rotation, 4),
    PD(u_day_color0, 4), PD(u_day_color1, 4), PD(u_night_color0, 4), PD(u_night_color1, 4),
    PD(u_day_night_transition_scale, 1),
};
#undef PD

const ParamDesc *find_param(const char *name) {
    if (!name) return nullptr;
    for (const ParamDesc &d : kParams)
        if (std::strcmp(d.name, name) == 0) return &d;
    return nullptr;
}

struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
};

thread_local std::string g_create_error = "";

}  // namespace

struct AtmoContext {
    int device = 0;
    int variant = 0;
    int flags = 0;
    int view_steps = 8, cloud_steps = 0, light_steps = 0;
    Params p;
    DeviceBuffer lut, blue, shape, cube;
    int lut_w = 0, lut_h = 0, shape_n = 0, cube_n = 0;
    int host_double_precision = 0;  // DOUBLE_PRECISION (main:25,118-125)
    int lane_split = 0;             // 0 = choose per launch by size, 1 = one lane per ray, 2 = two lanes per ray
    int last_split = 1;             // what the most recent launch used (atmo_kernel_name)
    // tile order with cost feedback (atmo_set_tile_feedback): -1 = by variant (clouds_high_rm on), 0 off, 1 on
    int tile_feedback = -1;
    DeviceBuffer tile_cost[2], tile_order[2];          // double-buffered: draw N uses [N & 1]
    int fb_tiles_x = 0, fb_tiles_y = 0, fb_split = 0;  // launch grid the buffers belong to
    unsigned fb_n = 0;                                 // draws of that grid so far
    bool fb_cost_valid[2] = {false, false};            // tile_cost[k] holds the costs of an enqueued draw
    bool fb_order_valid[2] = {false, false};           // tile_order[k] was (or is being) written by the sort kernel
    hipStream_t fb_stream = nullptr;                   // the sort kernel runs here, beside the draw
    hipEvent_t fb_ev_draw[2] = {nullptr, nullptr}, fb_ev_order[2] = {nullptr, nullptr};
    int timing = 0;          // 0 off; k >= 1: bracket every k-th launch with HIP events
    int launch_counter = 0;
    int timed_launches = 0;
    double timed_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // events not yet read back
    std::string err;
};

namespace {

int fail(AtmoContext *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(AtmoContext *ctx, hipError_t e, const char *what) {
    return fail(ctx, ATMO_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int dev_alloc(AtmoContext *ctx, DeviceBuffer &b, size_t bytes) {
    if (b.ptr && b.bytes == bytes) return ATMO_OK;
    if (b.ptr) { (void)hipFree(b.ptr); b.ptr = nullptr; b.bytes = 0; }
    if (bytes == 0) return ATMO_OK;
    HIP_TRY(ctx, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return ATMO_OK;
}

void dev_free(DeviceBuffer &b) {
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

// ---- cubemap apron -----------------------------------------------------------------------------------
// The kernel samples each face as an (n+2)^2 image whose border row/column holds the texels that lie
// across the cube edge (seamless filtering) and whose corners hold the mean of the three faces' corner
// texels.  Faces +X,-X,+Y,-Y,+Z,-Z; Vulkan face table (same as noise_cubemap.gd:110-128).
struct FaceBasis { int major[3], s[3], t[3]; };
const FaceBasis kFaces[6] = {
    {{1, 0, 0}, {0, 0, -1}, {0, -1, 0}}, {{-1, 0, 0}, {0, 0, 1}, {0, -1, 0}},
    {{0, 1, 0}, {1, 0, 0}, {0, 0, 1}},   {{0, -1, 0}, {1, 0, 0}, {0, 0, -1}},
    {{0, 0, 1}, {1, 0, 0}, {0, -1, 0}},  {{0, 0, -1}, {-1, 0, 0}, {0, -1, 0}},
};

// Integer cube-surface coordinates: texel centre (i,j) of face f sits at 2*i+1-n, 2*j+1-n in the face
// plane and at n on the major axis (units of half texels).  Stepping one texel off the face keeps the
// in-plane coordinate at the edge (+-n) and moves the major-axis coordinate in by one texel (n-1 ... in
// half-texel units: n - 1), which is a texel centre of the neighbouring face.
uint8_t cube_fold(const uint8_t *faces, int n, int f, int i, int j) {
    const FaceBasis &fb = kFaces[f];
    int sc = 2 * i + 1 - n, tc = 2 * j + 1 - n, ma = n;
    if (i < 0) { sc = -n; ma = n - 1; } else if (i >= n) { sc = n; ma = n - 1; }
    if (j < 0) { tc = -n; ma = n - 1; } else if (j >= n) { tc = n; ma = n - 1; }
    int p[3];
    for (int a = 0; a < 3; ++a) p[a] = fb.major[a] * ma + fb.s[a] * sc + fb.t[a] * tc;
    // which face is this point on?  exactly one coordinate has magnitude n
    int f2 = -1;
    for (int g = 0; g < 6 && f2 < 0; ++g) {
        const FaceBasis &gb = kFaces[g];
        int m = gb.major[0] * p[0] + gb.major[1] * p[1] + gb.major[2] * p[2];
        if (m == n) f2 = g;
    }
    const FaceBasis &gb = kFaces[f2];
    int s2 = gb.s[0] * p[0] + gb.s[1] * p[1] + gb.s[2] * p[2];
    int t2 = gb.t[0] * p[0] + gb.t[1] * p[1] + gb.t[2] * p[2];
    int i2 = (s2 + n - 1) / 2, j2 = (t2 + n - 1) / 2;
    if (i2 < 0) i2 = 0; if (i2 > n - 1) i2 = n - 1;
    if (j2 < 0) j2 = 0; if (j2 > n - 1) j2 = n - 1;
    return faces[((size_t)f2 * n + j2) * n + i2];
}

void build_cube_apron(const uint8_t *faces, int n, std::vector<uint8_t> &out) {
    const int st = n + 2;
    out.assign((size_t)6 * st * st, 0);
    for (int f = 0; f < 6; ++f)
        for (int j = -1; j <= n; ++j)
            for (int i = -1; i <= n; ++i) {
                const bool oi = (i < 0 || i >= n), oj = (j < 0 || j >= n);
                int v;
                if (!oi && !oj) {
                    v = faces[((size_t)f * n + j) * n + i];
                } else if (oi && oj) {
                    const int ci = i < 0 ? 0 : n - 1, cj = j < 0 ? 0 : n - 1;
                    const int a = faces[((size_t)f * n + cj) * n + ci];
                    const int b = cube_fold(faces, n, f, i, cj);
                    const int c = cube_fold(faces, n, f, ci, j);
                    v = (a + b + c + 1) / 3;
                } else {
                    v = cube_fold(faces, n, f, i, j);
                }
                out[((size_t)f * st + (j + 1)) * st + (i + 1)] = (uint8_t)v;
            }
}

// ---- device texture layouts (see atmo_kernels.hip "samplers") ---------------------------------------------

// 6 x (n+1)^2 words: word (i,j) of a face = padded texels (i,j), (i+1,j), (i,j+1), (i+1,j+1) in bytes 0..3
void build_cube_footprints(const std::vector<uint8_t> &padded, int n, std::vector<uint32_t> &out) {
    const int ps = n + 2, fs = n + 1;
    out.assign((size_t)6 * fs * fs, 0u);
    for (int f = 0; f < 6; ++f)
        for (int j = 0; j < fs; ++j)
            for (int i = 0; i < fs; ++i) {
                const uint8_t *p = &padded[((size_t)f * ps + j) * ps + i];
                out[((size_t)f * fs + j) * fs + i] =
                    (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[ps] << 16) | ((uint32_t)p[ps + 1] << 24);
            }
}

// n^3 words: word (i,j,k) = T(i,j,k), T(i+1,j,k), T(i,j+1,k), T(i+1,j+1,k) with repeat wrap
void build_shape_footprints(const uint8_t *t, int n, std::vector<uint32_t> &out) {
    out.assign((size_t)n * n * n, 0u);
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j) {
            const int j1 = (j + 1) % n;
            const uint8_t *r0 = t + ((size_t)k * n + j) * n, *r1 = t + ((size_t)k * n + j1) * n;
            for (int i = 0; i < n; ++i) {
                const int i1 = (i + 1) % n;
                out[((size_t)k * n + j) * n + i] =
                    (uint32_t)r0[i] | ((uint32_t)r0[i1] << 8) | ((uint32_t)r1[i] << 16) | ((uint32_t)r1[i1] << 24);
            }
        }
}

// (h+2) x (w+2) floats with a clamp-to-edge apron
void build_lut_apron(const float *lut, int w, int h, std::vector<float> &out) {
    const int st = w + 2;
    out.assign((size_t)st * (h + 2), 0.0f);
    for (int j = -1; j <= h; ++j) {
        const int cj = j < 0 ? 0 : (j >= h ? h - 1 : j);
        for (int i = -1; i <= w; ++i) {
            const int ci = i < 0 ? 0 : (i >= w ? w - 1 : i);
            out[(size_t)(j + 1) * st + (i + 1)] = lut[(size_t)cj * w + ci];
        }
    }
}

// ---- per-frame constants, evaluated like a scalar fp32 run of the shader would ----------------------------
inline float pow2f(float x) { return x * x; }
inline float pow4f(float x) { return x * x * x * x; }
inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline float clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
inline float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

void fill_consts(const AtmoContext *ctx, const AtmoFrame *f, const float *depth, float *rgba, atmo::RenderConsts &rc) {
    const Params &p = ctx->p;
    std::memset(&rc, 0, sizeof(rc));
    std::memcpy(rc.inv_p, f->inv_projection_matrix, sizeof(rc.inv_p));
    std::memcpy(rc.inv_v, f->inv_view_matrix, sizeof(rc.inv_v));
    const float *V = f->inv_view_matrix;
    // inv_view * (0,0,0,1), summed left to right (main:136)
    for (int r = 0; r < 3; ++r) rc.cam_pos_world[r] = V[0 + r] * 0.0f + V[4 + r] * 0.0f + V[8 + r] * 0.0f + V[12 + r] * 1.0f;
    rc.vw = (float)f->viewport_w;
    rc.vh = (float)f->viewport_h;
    rc.w = f->viewport_w; rc.h = f->viewport_h;
    rc.x0 = f->x0; rc.y0 = f->y0; rc.x1 = f->x1; rc.y1 = f->y1;
    for (int i = 0; i < 3; ++i) rc.center[i] = f->planet_center_viewspace[i];
    {   // sun_dir = normalize(sun_center_vs - planet_center_vs)  (main:164)
        float d[3];
        for (int i = 0; i < 3; ++i) d[i] = f->sun_center_viewspace[i] - f->planet_center_viewspace[i];
        float inv = 1.0f / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int i = 0; i < 3; ++i) rc.sun_dir[i] = d[i] * inv;
    }
    rc.planet_radius = p.u_planet_radius;
    rc.atmosphere_height = p.u_atmosphere_height;
    rc.atmosphere_radius = p.u_planet_radius + p.u_atmosphere_height;
    rc.density = p.u_density;
    rc.sphere_depth_factor = p.u_sphere_depth_factor;
    for (int i = 0; i < 3; ++i) {
        rc.coeff[i] = pow4f(400.0f / p.u_scattering_wavelengths[i]) * p.u_scattering_strength;  // v2:47-51
        rc.ambient[i] = p.u_atmosphere_ambient_color[i];
        rc.modulate[i] = p.u_atmosphere_modulate[i];
    }
    rc.view_steps = ctx->view_steps;
    rc.light_steps = ctx->light_steps;
    for (int i = 0; i < 3; ++i) {
        rc.day0[i] = p.u_day_color0[i]; rc.day1[i] = p.u_day_color1[i];
        rc.night0[i] = p.u_night_color0[i]; rc.night1[i] = p.u_night_color1[i];
    }
    rc.day_night_transition_scale = p.u_day_night_transition_scale;

    // clouds (cloud_funcs.gdshaderinc:260-261, 285-294, 186-206, 108-115)
    rc.clouds_bottom = p.u_planet_radius + p.u_cloud_bottom * p.u_atmosphere_height;
    rc.clouds_top = p.u_planet_radius + p.u_cloud_top * p.u_atmosphere_height;
    rc.cloud_thickness = rc.clouds_top - rc.clouds_bottom;
    rc.inv_cloud_thickness = 1.0f / rc.cloud_thickness;
    rc.cloud_density_scale = p.u_cloud_density_scale;
    rc.cloud_blend = p.u_cloud_blend;
    rc.coverage_bias = p.u_cloud_coverage_bias;
    rc.shape_factor = p.u_cloud_shape_factor;
    rc.shape_scale = p.u_cloud_shape_scale;
    rc.shape_invert = (p.u_cloud_shape_invert == 1.0f) ? 1 : 0;
    {   // range of `shape - 0.2 * detail` (clouds:52-58, detail = 0.5) over tex in [0, 1 + 2^-20]: every fp32 step of the
        // mix / invert is monotone in tex, so the two end points bound it whatever the sign of u_cloud_shape_factor
        const float f = p.u_cloud_shape_factor;
        const float t_hi = 1.0f + 9.5367431640625e-07f;
        float a = 0.5f * (1.0f - f) + 0.0f * f, b = 0.5f * (1.0f - f) + t_hi * f;
        if (rc.shape_invert) { a = 1.0f - a; b = 1.0f - b; }
        rc.shape_lo01 = std::fmin(a, b) - 0.1f;
        rc.shape_hi01 = std::fmax(a, b) - 0.1f;
    }
    std::memcpy(rc.cov_rot, p.u_cloud_coverage_rotation, sizeof(rc.cov_rot));
    const float *A = p.u_world_to_model_matrix;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            rc.view_to_model[col * 4 + row] = A[0 * 4 + row] * V[col * 4 + 0] + A[1 * 4 + row] * V[col * 4 + 1] +
                                              A[2 * 4 + row] * V[col * 4 + 2] + A[3 * 4 + row] * V[col * 4 + 3];
    const float *M = rc.view_to_model;
    for (int r = 0; r < 3; ++r) {
        rc.origin_model[r] = M[0 + r] * 0.0f + M[4 + r] * 0.0f + M[8 + r] * 0.0f + M[12 + r] * 1.0f;
        rc.sun_dir_model[r] = M[0 + r] * rc.sun_dir[0] + M[4 + r] * rc.sun_dir[1] + M[8 + r] * rc.sun_dir[2] + M[12 + r] * 0.0f;
    }
    {
        const float ground = p.u_planet_radius, top = rc.clouds_top, bottom = rc.clouds_bottom;
        const float space = 0.5f * std::sqrt(1.0f - pow2f(ground / top)) * bottom;
        const float groundd = 3.0f * space;
        const float len = std::sqrt(rc.origin_model[0] * rc.origin_model[0] + rc.origin_model[1] * rc.origin_model[1] +
                                    rc.origin_model[2] * rc.origin_model[2]);
        rc.max_d = mixf(groundd, space, smoothstepf(bottom, top * 1.05f, len));
    }
    rc.cloud_steps = ctx->cloud_steps;
    rc.inv_cloud_steps = ctx->cloud_steps > 0 ? 1.0f / (float)ctx->cloud_steps : 0.0f;
    {
        // get_light_raymarched (clouds:104-151): step_len grows x1.2 after each tap
        const float reach = (rc.clouds_top - rc.clouds_bottom) * 0.15f;
        const float inv_steps = 1.0f / 6.0f;
        float step_len = reach * inv_steps;
        for (int i = 0; i < 6; ++i) {
            rc.rm_offset[i] = (float)i * step_len;
            rc.rm_weight[i] = step_len * p.u_cloud_density_scale;
            step_len *= 1.2f;
        }
    }
    rc.lut = (const float *)ctx->lut.ptr; rc.lut_w = ctx->lut_w; rc.lut_h = ctx->lut_h;
    rc.blue = (const uint8_t *)ctx->blue.ptr;
    rc.shape = (const uint32_t *)ctx->shape.ptr; rc.shape_n = ctx->shape_n;
    rc.cube = (const uint32_t *)ctx->cube.ptr; rc.cube_n = ctx->cube_n;
    rc.depth = depth;
    rc.out = (float4 *)rgba;
    rc.out_pitch = f->x1 - f->x0;
    rc.out_x0 = f->x0;
    rc.out_y0 = f->y0;
    rc.composite = 0;
}

// Lanes per ray for one launch (atmo_set_lane_split; ATMO_LANE_SPLIT overrides for A/B runs).  Two lanes per ray double
// the wave count at the price of a duplicated per-pixel prologue and regrouped view sums; measured on MI355X it pays
// only where a few very long waves set the kernel time (clouds_high_rm, 1920x1080, pose P_space: -11 %) and costs
// 5-40 % elsewhere (profiles/round2/ab_lane_split.txt), so "auto" (0) is one lane per ray.
int choose_split(const AtmoContext *ctx, const AtmoFrame *f) {
    (void)f;
    if (const char *e = std::getenv("ATMO_LANE_SPLIT")) {
        if (e[0] == '1') return 1;
        if (e[0] == '2') return 2;
    }
    return ctx->lane_split == 2 ? 2 : 1;
}

void drain_timing(AtmoContext *ctx) {
    for (auto &pr : ctx->pending) {
        if (hipEventSynchronize(pr.second) == hipSuccess) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                ctx->timed_ms += ms;
                ctx->timed_launches += 1;
            }
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    ctx->pending.clear();
}

}  // namespace

extern "C" {

int atmo_abi_version(void) { return ATMO_ABI_VERSION; }

int atmo_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -ATMO_E_NO_DEVICE;
    return n;
}

int atmo_create(int device, int variant, int view_steps, int cloud_steps, int light_mode, int light_steps,
                AtmoContext **out) {
    if (!out) return fail(nullptr, ATMO_E_ARG, "atmo_create: out is null");
    *out = nullptr;
    if (variant < ATMO_VARIANT_NO_CLOUDS || variant > ATMO_VARIANT_V1_CLOUDS_HIGH)
        return fail(nullptr, ATMO_E_ARG, "atmo_create: unknown variant");
    if (light_mode != ATMO_LIGHT_LUT && light_mode != ATMO_LIGHT_DIRECT)
        return fail(nullptr, ATMO_E_ARG, "atmo_create: unknown light mode");
    if (view_steps < 0 || view_steps > 4096 || cloud_steps < 0 || cloud_steps > 4096 || light_steps < 0 || light_steps > 4096)
        return fail(nullptr, ATMO_E_ARG, "atmo_create: step count out of range");
    if (light_mode == ATMO_LIGHT_DIRECT && light_steps < 1)
        return fail(nullptr, ATMO_E_ARG, "atmo_create: direct light mode needs light_steps >= 1");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, ATMO_E_NO_DEVICE, std::string("atmo_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= n) return fail(nullptr, ATMO_E_NO_DEVICE, "atmo_create: device index out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, ATMO_E_NO_DEVICE, std::string("atmo_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");

    AtmoContext *ctx = new (std::nothrow) AtmoContext();
    if (!ctx) return fail(nullptr, ATMO_E_ARG, "atmo_create: out of host memory");
    ctx->device = device;
    ctx->variant = variant;
    const bool lite = variant >= ATMO_VARIANT_V1_NO_CLOUDS;
    static const int shipped_cloud_steps[7] = {0, 32, 64, 64, 0, 32, 64};  // shaders/planet_atmosphere_*.gdshader:4-7
    ctx->view_steps = view_steps > 0 ? view_steps : (lite ? 16 : 8);
    ctx->cloud_steps = shipped_cloud_steps[variant] == 0 ? 0 : (cloud_steps > 0 ? cloud_steps : shipped_cloud_steps[variant]);
    ctx->light_steps = (light_mode == ATMO_LIGHT_DIRECT && !lite) ? light_steps : 0;
    ctx->flags = 0;
    if (ctx->cloud_steps > 0) ctx->flags |= atmo::KF_CLOUDS | atmo::KF_PRECISE;  // precise cloud density is the default
    if (variant == ATMO_VARIANT_CLOUDS_HIGH_RM) ctx->flags |= atmo::KF_CLOUD_LIGHT_RM;
    if (light_mode == ATMO_LIGHT_DIRECT && !lite) ctx->flags |= atmo::KF_LIGHT_DIRECT;
    if (lite) ctx->flags |= atmo::KF_LITE;  // the v1 atmosphere reads no optical-depth LUT and has no light march

    // u_blue_noise_texture starts all-zero (jitter 0), like an unset sampler
    int rc = dev_alloc(ctx, ctx->blue, 256 * 256);
    if (rc == ATMO_OK) {
        e = hipMemset(ctx->blue.ptr, 0, 256 * 256);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "hipMemset");
    }
    if (rc != ATMO_OK) {
        g_create_error = ctx->err;
        dev_free(ctx->blue);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return ATMO_OK;
}

int atmo_destroy(AtmoContext *ctx) {
    if (!ctx) return ATMO_OK;
    (void)hipSetDevice(ctx->device);
    drain_timing(ctx);
    dev_free(ctx->lut);
    dev_free(ctx->blue);
    dev_free(ctx->shape);
    dev_free(ctx->cube);
    if (ctx->fb_stream) {
        (void)hipStreamSynchronize(ctx->fb_stream);
        (void)hipStreamDestroy(ctx->fb_stream);
        for (int k = 0; k < 2; ++k) {
            if (ctx->fb_ev_draw[k]) (void)hipEventDestroy(ctx->fb_ev_draw[k]);
            if (ctx->fb_ev_order[k]) (void)hipEventDestroy(ctx->fb_ev_order[k]);
        }
    }
    for (int k = 0; k < 2; ++k) {
        dev_free(ctx->tile_cost[k]);
        dev_free(ctx->tile_order[k]);
    }
    delete ctx;
    return ATMO_OK;
}

int atmo_set_param_f32(AtmoContext *ctx, const char *name, const float *v, int n) {
    if (!ctx) return ATMO_E_ARG;
    if (!v) return fail(ctx, ATMO_E_ARG, "atmo_set_param_f32: value is null");
    const ParamDesc *d = find_param(name);
    if (!d) return fail(ctx, ATMO_E_NAME, std::string("atmo_set_param_f32: unknown uniform '") + (name ? name : "(null)") + "'");
    if (n != d->count)
        return fail(ctx, ATMO_E_ARG, std::string("atmo_set_param_f32: '") + name + "' takes " + std::to_string(d->count) + " floats");
    std::memcpy(reinterpret_cast<char *>(&ctx->p) + d->offset, v, sizeof(float) * n);
    return ATMO_OK;
}

int atmo_get_param_f32(AtmoContext *ctx, const char *name, float *v, int n) {
    if (!ctx) return ATMO_E_ARG;
    if (!v) return fail(ctx, ATMO_E_ARG, "atmo_get_param_f32: value is null");
    const ParamDesc *d = find_param(name);
    if (!d) return fail(ctx, ATMO_E_NAME, std::string("atmo_get_param_f32: unknown uniform '") + (name ? name : "(null)") + "'");
    if (n != d->count)
        return fail(ctx, ATMO_E_ARG, std::string("atmo_get_param_f32: '") + name + "' holds " + std::to_string(d->count) + " floats");
    std::memcpy(v, reinterpret_cast<const char *>(&ctx->p) + d->offset, sizeof(float) * n);
    return ATMO_OK;
}

int atmo_set_texture(AtmoContext *ctx, const char *name, int kind, int w, int h, int d, const void *data, int memory) {
    if (!ctx) return ATMO_E_ARG;
    if (!name) return fail(ctx, ATMO_E_NAME, "atmo_set_texture: name is null");
    if (memory != ATMO_MEM_HOST && memory != ATMO_MEM_DEVICE) return fail(ctx, ATMO_E_ARG, "atmo_set_texture: bad memory kind");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const hipMemcpyKind ck = memory == ATMO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    // Texture updates are rare (bake, scene load): serialise with any in-flight render that reads the old copy.
    HIP_TRY(ctx, hipDeviceSynchronize());

    if (std::strcmp(name, "u_optical_depth_texture") == 0) {
        if (!data) { dev_free(ctx->lut); ctx->lut_w = ctx->lut_h = 0; return ATMO_OK; }
        if (kind != ATMO_TEX_2D_R32F) return fail(ctx, ATMO_E_ARG, "u_optical_depth_texture must be ATMO_TEX_2D_R32F");
        if (w < 1 || h < 1 || w > 8192 || h > 8192) return fail(ctx, ATMO_E_ARG, "u_optical_depth_texture: bad size");
        std::vector<float> host((size_t)w * h), padded;
        if (memory == ATMO_MEM_HOST) std::memcpy(host.data(), data, host.size() * sizeof(float));
        else HIP_TRY(ctx, hipMemcpy(host.data(), data, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        build_lut_apron(host.data(), w, h, padded);
        int rc = dev_alloc(ctx, ctx->lut, padded.size() * sizeof(float));
        if (rc != ATMO_OK) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->lut.ptr, padded.data(), ctx->lut.bytes, hipMemcpyHostToDevice));
        ctx->lut_w = w; ctx->lut_h = h;
        return ATMO_OK;
    }
    if (std::strcmp(name, "u_blue_noise_texture") == 0) {
        if (!data) { HIP_TRY(ctx, hipMemset(ctx->blue.ptr, 0, 256 * 256)); return ATMO_OK; }
        if (kind != ATMO_TEX_2D_R8) return fail(ctx, ATMO_E_ARG, "u_blue_noise_texture must be ATMO_TEX_2D_R8");
        if (w != 256 || h != 256) return fail(ctx, ATMO_E_ARG, "u_blue_noise_texture must be 256x256 (indexed & 0xff, main:169)");
        HIP_TRY(ctx, hipMemcpy(ctx->blue.ptr, data, 256 * 256, ck));
        return ATMO_OK;
    }
    if (std::strcmp(name, "u_cloud_shape_texture") == 0) {
        if (!data) { dev_free(ctx->shape); ctx->shape_n = 0; return ATMO_OK; }
        if (kind != ATMO_TEX_3D_R8) return fail(ctx, ATMO_E_ARG, "u_cloud_shape_texture must be ATMO_TEX_3D_R8");
        if (w < 1 || w > 512 || h != w || d != w) return fail(ctx, ATMO_E_ARG, "u_cloud_shape_texture must be n x n x n, n <= 512");
        std::vector<uint8_t> host((size_t)w * w * w);
        if (memory == ATMO_MEM_HOST) std::memcpy(host.data(), data, host.size());
        else HIP_TRY(ctx, hipMemcpy(host.data(), data, host.size(), hipMemcpyDeviceToHost));
        std::vector<uint32_t> fp;
        build_shape_footprints(host.data(), w, fp);
        int rc = dev_alloc(ctx, ctx->shape, fp.size() * sizeof(uint32_t));
        if (rc != ATMO_OK) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->shape.ptr, fp.data(), ctx->shape.bytes, hipMemcpyHostToDevice));
        ctx->shape_n = w;
        return ATMO_OK;
    }
    if (std::strcmp(name, "u_cloud_coverage_cubemap") == 0) {
        if (!data) { dev_free(ctx->cube); ctx->cube_n = 0; return ATMO_OK; }
        if (kind != ATMO_TEX_CUBE_R8) return fail(ctx, ATMO_E_ARG, "u_cloud_coverage_cubemap must be ATMO_TEX_CUBE_R8");
        if (w < 1 || w > 4096 || h != w || d != 6) return fail(ctx, ATMO_E_ARG, "u_cloud_coverage_cubemap must be n x n x 6 faces");
        const size_t face_bytes = (size_t)w * w;
        std::vector<uint8_t> host(6 * face_bytes);
        if (memory == ATMO_MEM_HOST) std::memcpy(host.data(), data, host.size());
        else HIP_TRY(ctx, hipMemcpy(host.data(), data, host.size(), hipMemcpyDeviceToHost));
        std::vector<uint8_t> padded;
        build_cube_apron(host.data(), w, padded);
        std::vector<uint32_t> fp;
        build_cube_footprints(padded, w, fp);
        int rc = dev_alloc(ctx, ctx->cube, fp.size() * sizeof(uint32_t));
        if (rc != ATMO_OK) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->cube.ptr, fp.data(), ctx->cube.bytes, hipMemcpyHostToDevice));
        ctx->cube_n = w;
        return ATMO_OK;
    }
    return fail(ctx, ATMO_E_NAME, std::string("atmo_set_texture: unknown texture uniform '") + name + "'");
}

// ---- host-only layout helpers (no device needed): what atmo_set_texture uploads -----------------------------------
int atmo_host_layout_cubemap(const uint8_t *faces, int n, uint32_t *footprints_out) {
    if (!faces || !footprints_out || n < 1 || n > 4096) return ATMO_E_ARG;
    std::vector<uint8_t> padded;
    build_cube_apron(faces, n, padded);
    std::vector<uint32_t> fp;
    build_cube_footprints(padded, n, fp);
    std::memcpy(footprints_out, fp.data(), fp.size() * sizeof(uint32_t));
    return ATMO_OK;
}

int atmo_host_layout_shape(const uint8_t *texels, int n, uint32_t *footprints_out) {
    if (!texels || !footprints_out || n < 1 || n > 512) return ATMO_E_ARG;
    std::vector<uint32_t> fp;
    build_shape_footprints(texels, n, fp);
    std::memcpy(footprints_out, fp.data(), fp.size() * sizeof(uint32_t));
    return ATMO_OK;
}

int atmo_host_layout_lut(const float *lut, int w, int h, float *apron_out) {
    if (!lut || !apron_out || w < 1 || h < 1 || w > 8192 || h > 8192) return ATMO_E_ARG;
    std::vector<float> padded;
    build_lut_apron(lut, w, h, padded);
    std::memcpy(apron_out, padded.data(), padded.size() * sizeof(float));
    return ATMO_OK;
}

int atmo_generate_noise_cubemap(AtmoContext *ctx, int resolution, uint32_t seed, float frequency, int octaves, float gain,
                                const float *scale3, int bind, uint8_t *faces_host, double *kernel_ms) {
    if (!ctx) return ATMO_E_ARG;
    if (resolution < 1 || resolution > 4096) return fail(ctx, ATMO_E_ARG, "atmo_generate_noise_cubemap: resolution must be 1..4096 (noise_cubemap.gd:29)");
    if (octaves < 1 || octaves > 16 || !scale3) return fail(ctx, ATMO_E_ARG, "atmo_generate_noise_cubemap: bad octaves / scale");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)6 * resolution * resolution;
    uint8_t *dev = nullptr;
    HIP_TRY(ctx, hipMalloc(&dev, bytes));
    atmo::NoiseCubemapConsts nc;
    nc.resolution = resolution; nc.seed = seed; nc.frequency = frequency; nc.gain = gain; nc.octaves = octaves;
    nc.scale[0] = scale3[0]; nc.scale[1] = scale3[1]; nc.scale[2] = scale3[2];
    nc.out = dev;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    if (e == hipSuccess) e = atmo::launch_noise_cubemap(nc, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    int rc = ATMO_OK;
    if (e != hipSuccess) rc = hip_fail(ctx, e, "atmo_generate_noise_cubemap");
    if (rc == ATMO_OK && kernel_ms) *kernel_ms = ms;
    if (rc == ATMO_OK && faces_host) {
        e = hipMemcpy(faces_host, dev, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "hipMemcpy");
    }
    if (rc == ATMO_OK && bind)
        rc = atmo_set_texture(ctx, "u_cloud_coverage_cubemap", ATMO_TEX_CUBE_R8, resolution, resolution, 6, dev, ATMO_MEM_DEVICE);
    (void)hipFree(dev);
    return rc;
}

int atmo_bake_optical_depth(AtmoContext *ctx, void *stream) {
    if (!ctx) return ATMO_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int w = 256, h = 256;  // optical_depth_baker.gd:24
    if (ctx->lut_w != w || ctx->lut_h != h) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        int rc = dev_alloc(ctx, ctx->lut, (size_t)(w + 2) * (h + 2) * sizeof(float));
        if (rc != ATMO_OK) return rc;
        ctx->lut_w = w; ctx->lut_h = h;
    }
    atmo::BakeConsts bc;
    bc.planet_radius = ctx->p.u_planet_radius;
    bc.atmosphere_height = ctx->p.u_atmosphere_height;
    bc.density = ctx->p.u_density;
    bc.w = w; bc.h = h;
    bc.steps = 64;  // optical_depth.gdshader:18
    bc.out = (float *)ctx->lut.ptr;
    HIP_TRY(ctx, atmo::launch_bake(bc, (hipStream_t)stream));
    return ATMO_OK;
}

int atmo_read_optical_depth(AtmoContext *ctx, float *lut_host, uint8_t *rgba8_host, int capacity_texels, void *stream) {
    if (!ctx) return ATMO_E_ARG;
    if (!ctx->lut.ptr) return fail(ctx, ATMO_E_STATE, "atmo_read_optical_depth: no LUT bound");
    const int n = ctx->lut_w * ctx->lut_h;
    if (capacity_texels < n) return fail(ctx, ATMO_E_ARG, "atmo_read_optical_depth: buffer too small");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    std::vector<float> tmp, padded((size_t)(ctx->lut_w + 2) * (ctx->lut_h + 2));
    float *dst = lut_host;
    if (!dst) { tmp.resize(n); dst = tmp.data(); }
    HIP_TRY(ctx, hipMemcpy(padded.data(), ctx->lut.ptr, padded.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int j = 0; j < ctx->lut_h; ++j)  // strip the apron
        std::memcpy(dst + (size_t)j * ctx->lut_w, &padded[(size_t)(j + 1) * (ctx->lut_w + 2) + 1], (size_t)ctx->lut_w * sizeof(float));
    if (rgba8_host) {
        // encode_float_to_viewport (optical_depth.gdshader:33-43): byte k of the bit pattern, little-endian;
        // value/255 stored to UNORM8 gives the byte back, so the packing is the raw IEEE bytes.
        for (int i = 0; i < n; ++i) {
            uint32_t u;
            std::memcpy(&u, &dst[i], 4);
            for (int k = 0; k < 4; ++k) rgba8_host[4 * i + k] = (uint8_t)((u >> (8 * k)) & 255u);
        }
    }
    return ATMO_OK;
}

static int render_impl(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream, bool composite);

int atmo_render(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream) {
    return render_impl(ctx, frame, depth_dev, rgba_dev, stream, false);
}

int atmo_render_composite(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *scene_rgba_dev, void *stream) {
    return render_impl(ctx, frame, depth_dev, scene_rgba_dev, stream, true);
}

static int render_impl(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream, bool composite) {
    if (!ctx) return ATMO_E_ARG;
    if (!frame) return fail(ctx, ATMO_E_ARG, "atmo_render: null frame");
    if (frame->viewport_w < 1 || frame->viewport_h < 1 || frame->viewport_w > 65536 || frame->viewport_h > 65536)
        return fail(ctx, ATMO_E_ARG, "atmo_render: bad viewport size");
    if (frame->x0 < 0 || frame->y0 < 0 || frame->x1 > frame->viewport_w || frame->y1 > frame->viewport_h ||
        frame->x0 > frame->x1 || frame->y0 > frame->y1)
        return fail(ctx, ATMO_E_ARG, "atmo_render: rect outside the viewport");
    if (frame->x0 == frame->x1 || frame->y0 == frame->y1) return ATMO_OK;  // empty rect: nothing to shade
    if (!depth_dev || !rgba_dev) return fail(ctx, ATMO_E_ARG, "atmo_render: null device pointer");
    if ((reinterpret_cast<uintptr_t>(rgba_dev) & 15u) != 0) return fail(ctx, ATMO_E_ARG, "atmo_render: rgba_dev must be 16-byte aligned");
    if (!(ctx->flags & (atmo::KF_LIGHT_DIRECT | atmo::KF_LITE)) && !ctx->lut.ptr)
        return fail(ctx, ATMO_E_STATE, "atmo_render: u_optical_depth_texture not set (call atmo_bake_optical_depth or atmo_set_texture)");
    if ((ctx->flags & atmo::KF_CLOUDS) && !ctx->shape.ptr)
        return fail(ctx, ATMO_E_STATE, "atmo_render: u_cloud_shape_texture not set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    atmo::RenderConsts rc;
    AtmoFrame fixed;
    if (ctx->host_double_precision) {  // main:118-125: undo the engine's negated INV_VIEW_MATRIX origin
        fixed = *frame;
        fixed.inv_view_matrix[12] *= -1.0f;
        fixed.inv_view_matrix[13] *= -1.0f;
        fixed.inv_view_matrix[14] *= -1.0f;
        frame = &fixed;
    }
    fill_consts(ctx, frame, depth_dev, rgba_dev, rc);
    if (composite) {  // the target is the whole scene colour buffer, addressed by absolute pixel
        rc.out_pitch = frame->viewport_w;
        rc.out_x0 = 0;
        rc.out_y0 = 0;
        rc.composite = 1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int split = choose_split(ctx, frame);
    int gx = 0, gy = 0;
    atmo::render_grid(rc, split, &gx, &gy);
    rc.tiles_x = gx;
    // default (-1): on for the raymarched-light variant, whose wave costs are the most skewed (profiles/round2/ab_tile_feedback.txt)
    bool feedback = ctx->tile_feedback == 1 || (ctx->tile_feedback < 0 && (ctx->flags & atmo::KF_CLOUD_LIGHT_RM));
    if (const char *e = std::getenv("ATMO_TILE_FEEDBACK")) feedback = e[0] == '1';  // A/B runs
    if (feedback) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) feedback = false;  // no side-stream work inside a graph
    }
    int fbk = -1;
    if (feedback && (long long)gx * gy >= 512) {  // tiny launches: nothing to schedule
        const size_t bytes = (size_t)gx * gy * sizeof(uint32_t);
        if (ctx->fb_tiles_x != gx || ctx->fb_tiles_y != gy || ctx->fb_split != split || !ctx->tile_cost[0].ptr) {
            // first launch of this grid: allocate (synchronous, once) and start recording
            if (!ctx->fb_stream) {
                HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->fb_stream, hipStreamNonBlocking));
                for (int k = 0; k < 2; ++k) {
                    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fb_ev_draw[k], hipEventDisableTiming));
                    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fb_ev_order[k], hipEventDisableTiming));
                }
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->fb_stream));
            for (int k = 0; k < 2; ++k) {
                int rc1 = dev_alloc(ctx, ctx->tile_cost[k], bytes);
                if (rc1 == ATMO_OK) rc1 = dev_alloc(ctx, ctx->tile_order[k], bytes);
                if (rc1 != ATMO_OK) return rc1;
                HIP_TRY(ctx, hipMemsetAsync(ctx->tile_cost[k].ptr, 0, bytes, s));
                ctx->fb_cost_valid[k] = ctx->fb_order_valid[k] = false;
            }
            ctx->fb_tiles_x = gx; ctx->fb_tiles_y = gy; ctx->fb_split = split;
            ctx->fb_n = 0;
        }
        fbk = (int)(ctx->fb_n & 1u);
        if (ctx->fb_order_valid[fbk]) {  // sorted during the previous draw from the costs of the draw before it
            HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->fb_ev_order[fbk], 0));
            rc.tile_order = (const uint32_t *)ctx->tile_order[fbk].ptr;
        }
        rc.tile_cost = (uint32_t *)ctx->tile_cost[fbk].ptr;
    }
    // kernel timing brackets the draw kernel alone (the tile-order kernel in front of it shows in the step time)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = ctx->timing > 0 && (ctx->launch_counter++ % ctx->timing) == 0;
    if (timed) {
        HIP_TRY(ctx, hipEventCreate(&e0));
        HIP_TRY(ctx, hipEventCreate(&e1));
        HIP_TRY(ctx, hipEventRecord(e0, s));
    }
    HIP_TRY(ctx, atmo::launch_render(ctx->flags, split, rc, s));
    if (fbk >= 0) {
        // While this draw runs, sort the tiles for the NEXT draw on the side stream, from the costs the PREVIOUS draw
        // left in the other buffer (the sort also clears them for the next draw to write).
        HIP_TRY(ctx, hipEventRecord(ctx->fb_ev_draw[fbk], s));
        ctx->fb_cost_valid[fbk] = true;
        const int o = fbk ^ 1;
        if (ctx->fb_cost_valid[o]) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->fb_stream, ctx->fb_ev_draw[o], 0));
            HIP_TRY(ctx, atmo::launch_tile_order((uint32_t *)ctx->tile_cost[o].ptr, (uint32_t *)ctx->tile_order[o].ptr, gx * gy, ctx->fb_stream));
            HIP_TRY(ctx, hipEventRecord(ctx->fb_ev_order[o], ctx->fb_stream));
            ctx->fb_order_valid[o] = true;
            ctx->fb_cost_valid[o] = false;
        }
        ctx->fb_n += 1;
    }
    ctx->last_split = split;
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(e1, s));
        ctx->pending.emplace_back(e0, e1);
    }
    return ATMO_OK;
}

int atmo_set_precision(AtmoContext *ctx, int mode) {
    if (!ctx) return ATMO_E_ARG;
    if (mode != 0 && mode != 1) return fail(ctx, ATMO_E_ARG, "atmo_set_precision: mode must be 0 (fast) or 1 (precise cloud density)");
    if (mode == 1 && (ctx->flags & atmo::KF_CLOUDS)) ctx->flags |= atmo::KF_PRECISE;  // only the cloud kernels have a precise form
    else ctx->flags &= ~atmo::KF_PRECISE;
    return ATMO_OK;
}

int atmo_set_host_double_precision(AtmoContext *ctx, int enable) {
    if (!ctx) return ATMO_E_ARG;
    ctx->host_double_precision = enable ? 1 : 0;
    return ATMO_OK;
}

int atmo_set_tile_feedback(AtmoContext *ctx, int mode) {
    if (!ctx) return ATMO_E_ARG;
    if (mode < -1 || mode > 1) return fail(ctx, ATMO_E_ARG, "atmo_set_tile_feedback: -1 (by variant), 0 (off) or 1 (on)");
    ctx->tile_feedback = mode;
    ctx->fb_tiles_x = ctx->fb_tiles_y = 0;  // restart the feedback state at the next launch
    return ATMO_OK;
}

int atmo_set_lane_split(AtmoContext *ctx, int lanes_per_ray) {
    if (!ctx) return ATMO_E_ARG;
    if (lanes_per_ray < 0 || lanes_per_ray > 2) return fail(ctx, ATMO_E_ARG, "atmo_set_lane_split: 0 (auto), 1 or 2 lanes per ray");
    ctx->lane_split = lanes_per_ray;
    return ATMO_OK;
}

int atmo_set_timing(AtmoContext *ctx, int enable) {
    if (!ctx) return ATMO_E_ARG;
    (void)hipSetDevice(ctx->device);
    drain_timing(ctx);
    ctx->timing = enable > 0 ? enable : 0;
    ctx->launch_counter = 0;
    ctx->timed_launches = 0;
    ctx->timed_ms = 0.0;
    return ATMO_OK;
}

int atmo_get_timing(AtmoContext *ctx, int *launches, double *total_ms) {
    if (!ctx) return ATMO_E_ARG;
    (void)hipSetDevice(ctx->device);
    drain_timing(ctx);
    if (launches) *launches = ctx->timed_launches;
    if (total_ms) *total_ms = ctx->timed_ms;
    return ATMO_OK;
}

int atmo_selftest_exact_math(AtmoContext *ctx, uint32_t first_bits, uint32_t count, float divisor,
                             uint32_t *sqrt_mismatches, uint32_t *div_mismatches) {
    if (!ctx) return ATMO_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned int *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, 2 * sizeof(unsigned int)));
    hipError_t e = hipMemset(d, 0, 2 * sizeof(unsigned int));
    if (e == hipSuccess) e = atmo::launch_selftest(first_bits, count, divisor, 1.0f / divisor, d, nullptr);
    unsigned int h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "atmo_selftest_exact_math");
    if (sqrt_mismatches) *sqrt_mismatches = h[0];
    if (div_mismatches) *div_mismatches = h[1];
    return ATMO_OK;
}

const char *atmo_kernel_name(AtmoContext *ctx) {
    if (!ctx) return "";
    return atmo::render_kernel_name(ctx->flags, ctx->light_steps, ctx->last_split);
}

const char *atmo_last_error_string(AtmoContext *ctx) {
    if (!ctx) return g_create_error.c_str();
    return ctx->err.c_str();
}

}  // extern "C"
