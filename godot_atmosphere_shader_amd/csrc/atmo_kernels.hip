// atmo_kernels.hip -- gfx950 (CDNA4 / MI355X) kernels for the per-pixel atmosphere + cloud raymarch.
//
// One wavefront lane per view ray; a 256-thread workgroup shades a 16x16 pixel tile, each 64-lane
// wave a 16x4 strip, so a wave's RGBA float4 stores are four 256-byte contiguous runs.
//
// Numerical contract (tests/test_gpu_parity.py): max |channel - fp32 oracle| <= 1e-4.
//   * This file is compiled with -ffp-contract=off.  Everything that decides control flow (ray
//     setup, ray/sphere tests, cloud gates) and the ill-conditioned cloud chain
//     (position -> |pos| -> height ratio; a 3.2-unit shell at radius 100 amplified x50 by the
//     density ramp) is evaluated in the reference's operation order with IEEE sqrt and divide,
//     i.e. bit-identically to a scalar fp32 evaluation of the GDShader.
//   * The well-conditioned inner-loop arithmetic opts into FMA contraction and the hardware
//     transcendental units (v_exp_f32, v_rsq_f32, v_rcp_f32, v_sqrt_f32) inside
//     `#pragma clang fp contract(fast)` blocks.
//
// Reference functions restated (paths under /root/reference/addons/zylann.atmosphere/shaders/):
//   include/planet_atmosphere_main.gdshaderinc:106-197  atmosphere_fragment      -> atmo_render_kernel
//   include/util.gdshaderinc:20-40                      ray_sphere               -> SphereHit / hit_radius
//   include/atmosphere_funcs_v2.gdshaderinc:14-29       get_baked_optical_depth  -> lut_sample
//   include/atmosphere_funcs_v2.gdshaderinc:32-101      compute_atmosphere_v2    -> march_atmosphere
//   include/cloud_funcs.gdshaderinc:31-68               get_density_full         -> cloud_density
//   include/cloud_funcs.gdshaderinc:78-167              get_light*               -> inside march_clouds
//   include/cloud_funcs.gdshaderinc:175-247             raymarch_cloud           -> march_clouds
//   include/cloud_funcs.gdshaderinc:249-324             render_clouds            -> atmo_render_kernel tail
//   optical_depth.gdshader:17-31,45-68                  LUT bake                 -> atmo_bake_kernel
#include "atmo_device.h"

namespace atmo {

constexpr int TILE_W = 16;
constexpr int TILE_H = 16;
constexpr float LOG2E = 1.44269504088896340736f;

// ---- hardware transcendental units (approximate, ~1 ulp) ---------------------------------------
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// GLSL mix(a,b,t) = a*(1-t) + b*t
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
// IEEE-754 correctly rounded sqrt and divide: hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt)
// expands these to the fix-up sequences; __fsqrt_rn/__fdiv_rn are NOT used because the HIP headers map
// __fsqrt_rn to the native (1 ulp) square root.  tests/test_gpu_parity.py checks bit-exactness via the LUT bake.
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }

// ---- exact (IEEE, unfused) helpers: must match a scalar fp32 evaluation bit for bit -------------
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ float dot_lr(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ray_sphere (util.gdshaderinc:20-40) for a ray from the view-space origin: the part shared by all radii.
struct SphereHit {
    float b;    // dot(oc, dir)
    float qc2;  // dot(qc, qc)
};
__device__ __forceinline__ SphereHit sphere_setup(V3 center, V3 dir) {
    V3 oc = {0.0f - center.x, 0.0f - center.y, 0.0f - center.z};
    float b = dot_lr(oc, dir);
    V3 qc = {oc.x - b * dir.x, oc.y - b * dir.y, oc.z - b * dir.z};
    SphereHit s;
    s.b = b;
    s.qc2 = dot_lr(qc, qc);
    return s;
}
// returns (x, y); equal (1e6, 1e6) when missed
__device__ __forceinline__ float2 hit_radius(SphereHit s, float radius) {
    float h = radius * radius - s.qc2;
    if (h < 0.0f) return make_float2(1000000.0f, 1000000.0f);
    h = ieee_sqrt(h);
    return make_float2(-s.b - h, -s.b + h);
}

// ---- samplers ----------------------------------------------------------------------------------

// texture(u_optical_depth_texture, (u, v)).r : bilinear, clamp-to-edge, R32F
__device__ __forceinline__ float lut_sample(const float *__restrict__ lut, int w, int h, float u, float v) {
#pragma clang fp contract(fast)
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    int i = (int)xf, j = (int)yf;
    int i0 = max(i, 0), i1 = min(i + 1, w - 1);
    int j0 = max(j, 0), j1 = min(j + 1, h - 1);
    const float *r0 = lut + j0 * w;
    const float *r1 = lut + j1 * w;
    float t00 = r0[i0], t10 = r0[i1], t01 = r1[i0], t11 = r1[i1];
    float a = t00 + (t10 - t00) * fx;
    float b = t01 + (t11 - t01) * fx;
    return a + (b - a) * fy;
}

// texture(u_cloud_shape_texture, p).r : trilinear, repeat, R8 (n^3, x fastest)
__device__ __forceinline__ float shape_sample(const uint8_t *__restrict__ tex, int n, float px, float py, float pz) {
#pragma clang fp contract(fast)
    float nf = (float)n;
    float x = px * nf - 0.5f, y = py * nf - 0.5f, z = pz * nf - 0.5f;
    float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    float fx = x - xf, fy = y - yf, fz = z - zf;
    int i = (int)xf, j = (int)yf, k = (int)zf;
    int i0, i1, j0, j1, k0, k1;
    if ((n & (n - 1)) == 0) {
        int m = n - 1;
        i0 = i & m; i1 = (i + 1) & m;
        j0 = j & m; j1 = (j + 1) & m;
        k0 = k & m; k1 = (k + 1) & m;
    } else {
        i0 = ((i % n) + n) % n; i1 = (i0 + 1) % n;
        j0 = ((j % n) + n) % n; j1 = (j0 + 1) % n;
        k0 = ((k % n) + n) % n; k1 = (k0 + 1) % n;
    }
    const uint8_t *p00 = tex + (k0 * n + j0) * n;
    const uint8_t *p10 = tex + (k0 * n + j1) * n;
    const uint8_t *p01 = tex + (k1 * n + j0) * n;
    const uint8_t *p11 = tex + (k1 * n + j1) * n;
    float a000 = (float)p00[i0], a100 = (float)p00[i1];
    float a010 = (float)p10[i0], a110 = (float)p10[i1];
    float a001 = (float)p01[i0], a101 = (float)p01[i1];
    float a011 = (float)p11[i0], a111 = (float)p11[i1];
    float c00 = a000 + (a100 - a000) * fx;
    float c10 = a010 + (a110 - a010) * fx;
    float c01 = a001 + (a101 - a001) * fx;
    float c11 = a011 + (a111 - a011) * fx;
    float c0 = c00 + (c10 - c00) * fy;
    float c1 = c01 + (c11 - c01) * fy;
    return (c0 + (c1 - c0) * fz) * (1.0f / 255.0f);
}

// texture(u_cloud_coverage_cubemap, d).r : LOD 0, bilinear, seamless.  `cube` holds six faces of
// (n+2)^2 bytes: the face plus a one-texel apron copied from the neighbouring faces (corners: mean of 3).
__device__ __forceinline__ float cube_sample(const uint8_t *__restrict__ cube, int n, float dx, float dy, float dz) {
#pragma clang fp contract(fast)
    float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    float sc, tc, ma;
    int f;
    if (az >= ax && az >= ay) {
        ma = az;
        bool pos = dz >= 0.0f;
        f = pos ? 4 : 5;
        sc = pos ? dx : -dx;
        tc = -dy;
    } else if (ay >= ax) {
        ma = ay;
        bool pos = dy >= 0.0f;
        f = pos ? 2 : 3;
        sc = dx;
        tc = pos ? dz : -dz;
    } else {
        ma = ax;
        bool pos = dx >= 0.0f;
        f = pos ? 0 : 1;
        sc = pos ? -dz : dz;
        tc = -dy;
    }
    // s = 0.5*(sc/ma + 1); one Newton step on the hardware reciprocal keeps the quotient within 1 ulp of IEEE
    float r = hw_rcp(ma);
    float qs = sc * r, qt = tc * r;
    qs = fmaf(fmaf(-qs, ma, sc), r, qs);
    qt = fmaf(fmaf(-qt, ma, tc), r, qt);
    float nf = (float)n;
    float x = (0.5f * (qs + 1.0f)) * nf - 0.5f;
    float y = (0.5f * (qt + 1.0f)) * nf - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float fx = x - xf, fy = y - yf;
    int i = min(max((int)xf, -1), n - 1), j = min(max((int)yf, -1), n - 1);
    int stride = n + 2;
    const uint8_t *p = cube + (f * stride + (j + 1)) * stride + (i + 1);
    float t00 = (float)p[0], t10 = (float)p[1], t01 = (float)p[stride], t11 = (float)p[stride + 1];
    float a = t00 + (t10 - t00) * fx;
    float b = t01 + (t11 - t01) * fx;
    return (a + (b - a) * fy) * (1.0f / 255.0f);
}

// ---- compute_atmosphere_v2 -----------------------------------------------------------------------
// Returns RGBA.  Well-conditioned: fused arithmetic + hardware transcendentals throughout.
//   * alpha: the reference's recurrence alpha += (1-exp(-d))*(1-alpha) is 1 - prod(exp(-d_i))
//     = 1 - exp(-view_optical_depth); one exp after the loop replaces one per step.
//   * light: sum(d_i * T_i * coeff) = coeff * sum(d_i * T_i).
template <bool DIRECT>
__device__ __forceinline__ float4 march_atmosphere(const RenderConsts &rc, V3 dir, float t_begin, float step_len, float jitter) {
#pragma clang fp contract(fast)
    const int steps = rc.view_steps;
    const float inv_h = hw_rcp(rc.atmosphere_height);
    const float dens2 = rc.density * rc.density;
    const float kr = -rc.coeff[0] * LOG2E, kg = -rc.coeff[1] * LOG2E, kb = -rc.coeff[2] * LOG2E;
    const float cx = rc.center[0], cy = rc.center[1], cz = rc.center[2];
    const float sx = rc.sun_dir[0], sy = rc.sun_dir[1], sz = rc.sun_dir[2];
    const float ratm2 = rc.atmosphere_radius * rc.atmosphere_radius;
    const int light_steps = rc.light_steps;
    const float inv_light_steps = hw_rcp((float)light_steps);

    float px = dir.x * t_begin, py = dir.y * t_begin, pz = dir.z * t_begin;
    const float sdx = dir.x * step_len, sdy = dir.y * step_len, sdz = dir.z * step_len;
    float lr = 0.0f, lg = 0.0f, lb = 0.0f, view_od = 0.0f;

    for (int i = 0; i < steps; ++i) {
        float ox = px - cx, oy = py - cy, oz = pz - cz;
        float r2 = ox * ox + oy * oy + oz * oz;
        float inv_r = hw_rsq(r2);
        float r = r2 * inv_r;
        float hr = sat((r - rc.planet_radius) * inv_h);
        float y = 1.0f - hr;
        float y3 = y * y * y;
        float bdot = ox * sx + oy * sy + oz * sz;

        float sun_od;
        if (DIRECT) {
            // chord from the sample to the outer sphere along the sun direction, then a left Riemann sum
            float hh = ratm2 - (r2 - bdot * bdot);
            float sq = hw_sqrt(fmaxf(hh, 0.0f));
            float x0 = -bdot - sq, x1 = -bdot + sq;
            float ray_len = (hh < 0.0f) ? 0.0f : (x1 - fmaxf(x0, 0.0f));
            float lstep = ray_len * inv_light_steps;
            float acc = y3;  // sample 0 sits on the view sample itself
            float b2 = bdot + bdot;
            for (int j = 1; j < light_steps; ++j) {
                float s = lstep * (float)j;
                float rr = hw_sqrt(fmaf(s, s + b2, r2));
                float yy = 1.0f - sat((rr - rc.planet_radius) * inv_h);
                acc = fmaf(yy * yy, yy, acc);
            }
            sun_od = acc * lstep * dens2;
        } else {
            float uvx = 0.5f + 0.5f * (bdot * inv_r);
            sun_od = lut_sample(rc.lut, rc.lut_w, rc.lut_h, uvx, hr);
        }

        float d = y3 * dens2 * step_len;
        view_od += d;
        float od = sun_od + view_od;
        lr = fmaf(d, hw_exp2(od * kr), lr);
        lg = fmaf(d, hw_exp2(od * kg), lg);
        lb = fmaf(d, hw_exp2(od * kb), lb);

        px += sdx; py += sdy; pz += sdz;
    }

    float alpha = 1.0f - hw_exp2(-view_od * LOG2E);
    float4 o;
    o.x = sat(fmaf(lr, rc.coeff[0], rc.ambient[0])) * rc.modulate[0];
    o.y = sat(fmaf(lg, rc.coeff[1], rc.ambient[1])) * rc.modulate[1];
    o.z = sat(fmaf(lb, rc.coeff[2], rc.ambient[2])) * rc.modulate[2];
    o.w = clampf(fmaf(jitter, 0.02f, alpha), 0.0f, 0.99f);
    return o;
}

// ---- clouds ----------------------------------------------------------------------------------------

// get_density_full with CLOUDS_ALWAYS_LOW_QUALITY (detail = 0.5).  `r` = |pos| and `hr` = height ratio
// come from the exact chain in the caller.
__device__ __forceinline__ float cloud_density(const RenderConsts &rc, float px, float py, float pz, float hr) {
#pragma clang fp contract(fast)
    float t = 2.0f * hr - 1.0f;
    float hc = fmaxf(1.0f - t * t, 0.0f);
    if (!(hc > 0.0f)) return 0.0f;  // outside the layer: (..)*0*50-20 clamps to 0, skip the fetches
    float coverage = 1.0f;
    if (rc.cube != nullptr) {
        float qx = rc.cov_rot[0] * px + rc.cov_rot[2] * pz;
        float qz = rc.cov_rot[1] * px + rc.cov_rot[3] * pz;
        coverage = cube_sample(rc.cube, rc.cube_n, qx, py, qz);
    }
    coverage = coverage - 0.25f * hr + rc.coverage_bias;
    float s = rc.shape_scale;
    float shape = mixf(0.5f, shape_sample(rc.shape, rc.shape_n, px * s, py * s, pz * s), rc.shape_factor);
    if (rc.shape_invert) shape = 1.0f - shape;
    float density = (shape - 0.1f + mixf(-1.2f, 1.5f, coverage)) * hc;
    return sat(density * 50.0f - 20.0f);
}

// exact |p| and (|p| - bottom) / thickness, as a scalar fp32 evaluation would produce them
__device__ __forceinline__ void cloud_height(const RenderConsts &rc, float px, float py, float pz, float &r, float &hr) {
    r = ieee_sqrt(px * px + py * py + pz * pz);
    hr = ieee_div(r - rc.clouds_bottom, rc.cloud_thickness);
}

// get_light_raymarched (cloud_funcs.gdshaderinc:104-151): 6 density taps towards the sun.
// 1 - prod(exp(-d_i)) = 1 - exp(-sum d_i): one exp instead of six.
__device__ __forceinline__ float light_raymarched(const RenderConsts &rc, float px, float py, float pz, float hr0,
                                                  float sx, float sy, float sz) {
    float step_len = rc.rm_step0;
    float sum = 0.0f;
    for (int i = 0; i < 6; ++i) {
        float k = (float)i * step_len;
        // exact: pos0 + (i*step)*dir, unfused
        float qx = px + k * sx, qy = py + k * sy, qz = pz + k * sz;
        float r, hr;
        cloud_height(rc, qx, qy, qz, r, hr);
        float d = cloud_density(rc, qx, qy, qz, hr);
        sum += d * (step_len * rc.cloud_density_scale);
        step_len *= 1.2f;
    }
    float alpha = 1.0f - hw_exp2(-sum * LOG2E);
    return mixf(1.0f, hr0 * 0.2f, alpha);
}

// raymarch_cloud (cloud_funcs.gdshaderinc:175-247).  Returns (total_light, alpha).
template <bool RM>
__device__ __forceinline__ float2 march_clouds(const RenderConsts &rc, V3 dir_m, float t_begin, float t_end, float jitter) {
    const int steps = rc.cloud_steps;
    // exact: positions
    t_end = t_begin + fminf(t_end - t_begin, rc.max_d);
    const float step_len = (t_end - t_begin) * rc.inv_cloud_steps;
    const float js = jitter * step_len;
    float px = (rc.origin_model[0] + dir_m.x * js) + dir_m.x * t_begin;
    float py = (rc.origin_model[1] + dir_m.y * js) + dir_m.y * t_begin;
    float pz = (rc.origin_model[2] + dir_m.z * js) + dir_m.z * t_begin;
    const float sx = rc.sun_dir_model[0], sy = rc.sun_dir_model[1], sz = rc.sun_dir_model[2];

    // pow(dot(ray_dir, sun_dir), 16) is constant along the ray; dp <= 0 => 0
    float dp = dir_m.x * sx + dir_m.y * sy + dir_m.z * sz;
    float p16 = 0.0f;
    if (dp > 0.0f) {
        float p2 = dp * dp, p4 = p2 * p2, p8 = p4 * p4;
        p16 = p8 * p8;
    }

    float total_transmittance = 1.0f, total_light = 0.0f, one_minus_alpha = 1.0f;
    const float neg_scale_step_log2e = -(rc.cloud_density_scale * step_len) * LOG2E;

    for (int i = 0; i < steps; ++i) {
        float r, hr;
        cloud_height(rc, px, py, pz, r, hr);
        {
#pragma clang fp contract(fast)
            float density = cloud_density(rc, px, py, pz, hr);
            float light;
            if (RM) {
                light = light_raymarched(rc, px, py, pz, hr, sx, sy, sz);
            } else {
                light = fmaf(p16, one_minus_alpha, hr);
            }
            // get_planet_shadow: smoothstep(-0.3, 0.3, dot(normalize(pos), -sun_dir))
            float sd = -(px * sx + py * sy + pz * sz) * hw_rcp(r);
            float st = sat((sd + 0.3f) * (1.0f / 0.6f));
            float shadow = st * st * (3.0f - 2.0f * st);
            light *= fmaf(shadow, 0.002f - 1.0f, 1.0f);

            float transmittance = hw_exp2(density * neg_scale_step_log2e);
            total_transmittance = fmaxf(total_transmittance * transmittance, 0.005f);
            total_light = fmaf(light * (density * rc.cloud_density_scale) * step_len, total_transmittance, total_light);
            one_minus_alpha *= transmittance;
        }
        // exact: pos += ray_dir * step_len
        px = px + dir_m.x * step_len;
        py = py + dir_m.y * step_len;
        pz = pz + dir_m.z * step_len;
    }
    return make_float2(total_light, 1.0f - one_minus_alpha);
}

// ---- atmosphere_fragment ---------------------------------------------------------------------------
template <int FLAGS>
__global__ __launch_bounds__(TILE_W *TILE_H) void atmo_render_kernel(const RenderConsts rc) {
    constexpr bool CLOUDS = (FLAGS & KF_CLOUDS) != 0;
    constexpr bool RM = (FLAGS & KF_CLOUD_LIGHT_RM) != 0;
    constexpr bool DIRECT = (FLAGS & KF_LIGHT_DIRECT) != 0;

    const int lx = threadIdx.x % TILE_W, ly = threadIdx.x / TILE_W;
    const int px = rc.x0 + blockIdx.x * TILE_W + lx;
    const int py = rc.y0 + blockIdx.y * TILE_H + ly;
    if (px >= rc.x1 || py >= rc.y1) return;
    float4 *out = rc.out + (size_t)(py - rc.y0) * (size_t)(rc.x1 - rc.x0) + (px - rc.x0);

    // --- exact prologue (main:128-169) -----------------------------------------------------------
    const float nonlinear_depth = rc.depth[(size_t)py * rc.w + px];
    const float uvx = ieee_div((float)px + 0.5f, rc.vw);
    const float uvy = ieee_div((float)py + 0.5f, rc.vh);
    const float nx = uvx * 2.0f - 1.0f, ny = uvy * 2.0f - 1.0f, nz = nonlinear_depth;
    const float *P = rc.inv_p;
    const float vx = P[0] * nx + P[4] * ny + P[8] * nz + P[12] * 1.0f;
    const float vy = P[1] * nx + P[5] * ny + P[9] * nz + P[13] * 1.0f;
    const float vz = P[2] * nx + P[6] * ny + P[10] * nz + P[14] * 1.0f;
    const float vw = P[3] * nx + P[7] * ny + P[11] * nz + P[15] * 1.0f;
    const float *Vm = rc.inv_v;
    const float wx = Vm[0] * vx + Vm[4] * vy + Vm[8] * vz + Vm[12] * vw;
    const float wy = Vm[1] * vx + Vm[5] * vy + Vm[9] * vz + Vm[13] * vw;
    const float wz = Vm[2] * vx + Vm[6] * vy + Vm[10] * vz + Vm[14] * vw;
    const float ww = Vm[3] * vx + Vm[7] * vy + Vm[11] * vz + Vm[15] * vw;
    const float pwx = ieee_div(wx, ww), pwy = ieee_div(wy, ww), pwz = ieee_div(wz, ww);
    const float ddx = rc.cam_pos_world[0] - pwx, ddy = rc.cam_pos_world[1] - pwy, ddz = rc.cam_pos_world[2] - pwz;
    float linear_depth = ieee_sqrt(ddx * ddx + ddy * ddy + ddz * ddz);

    // ray_dir = normalize(view_coords.xyz - 0) = v * (1/sqrt(dot(v,v)))
    const float vvx = vx - 0.0f, vvy = vy - 0.0f, vvz = vz - 0.0f;
    const float inv_len = ieee_div(1.0f, ieee_sqrt(vvx * vvx + vvy * vvy + vvz * vvz));
    const V3 dir = {vvx * inv_len, vvy * inv_len, vvz * inv_len};
    const V3 center = {rc.center[0], rc.center[1], rc.center[2]};

    const SphereHit sh = sphere_setup(center, dir);
    const float2 rs_atmo = hit_radius(sh, rc.atmosphere_radius);

    if (rs_atmo.x == rs_atmo.y) {  // discard
        *out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const float t_begin = fmaxf(rs_atmo.x, 0.0f);
    float t_end = fmaxf(rs_atmo.y, 0.0f);
    const float2 rs_ground = hit_radius(sh, rc.planet_radius);
    float gd = 10000000.0f;
    if (rs_ground.x != rs_ground.y) gd = rs_ground.x;
    linear_depth = linear_depth * (1.0f - rc.sphere_depth_factor) + gd * rc.sphere_depth_factor;
    t_end = fminf(t_end, linear_depth);

    const float jx = rc.vw * uvx, jy = rc.vh * uvy;
    const int ji = ((int)jx) & 0xff, jj = ((int)jy) & 0xff;
    const float jitter = ieee_div((float)rc.blue[jj * 256 + ji], 255.0f);

    const float view_step_len = ieee_div(t_end - t_begin, (float)rc.view_steps);
    float4 rgba = march_atmosphere<DIRECT>(rc, dir, t_begin, view_step_len, jitter);

    if (CLOUDS) {
        // --- render_clouds (cloud_funcs.gdshaderinc:249-324), gates evaluated exactly -----------------
        const float2 rs_top = hit_radius(sh, rc.clouds_top);
        if (rs_top.x != rs_top.y) {
            const float2 rs_bottom = hit_radius(sh, rc.clouds_bottom);
            const float c0 = fmaxf(rs_top.x, 0.0f);
            const float c1 = fminf(rs_top.y, linear_depth);
            if (c0 < linear_depth && (linear_depth > rs_bottom.y || rs_bottom.x > 0.0f)) {
                const float *M = rc.view_to_model;
                V3 dir_m;
                dir_m.x = M[0] * dir.x + M[4] * dir.y + M[8] * dir.z;
                dir_m.y = M[1] * dir.x + M[5] * dir.y + M[9] * dir.z;
                dir_m.z = M[2] * dir.x + M[6] * dir.y + M[10] * dir.z;
                const float2 rr = march_clouds<RM>(rc, dir_m, c0, c1, jitter);
                {
#pragma clang fp contract(fast)
                    const float cl = rr.x, ca = rr.y;
                    // blend_colors(self = atmosphere, over = cloud)  (util.gdshaderinc:61-69)
                    const float sa = 1.0f - ca;
                    const float a = rgba.w * sa + ca;
                    float abx = 0.0f, aby = 0.0f, abz = 0.0f, abw = 0.0f;
                    if (a != 0.0f) {
                        const float inv_a = ieee_div(1.0f, a);
                        const float ws = rgba.w * sa, wo = cl * ca;
                        abx = (rgba.x * ws + wo) * inv_a;
                        aby = (rgba.y * ws + wo) * inv_a;
                        abz = (rgba.z * ws + wo) * inv_a;
                        abw = a;
                    }
                    const float addx = rgba.x + cl * ca, addy = rgba.y + cl * ca, addz = rgba.z + cl * ca;
                    const float addw = fmaxf(rgba.w, ca);
                    rgba.x = mixf(abx, addx, rc.cloud_blend);
                    rgba.y = mixf(aby, addy, rc.cloud_blend);
                    rgba.z = mixf(abz, addz, rc.cloud_blend);
                    rgba.w = mixf(abw, addw, rc.cloud_blend);
                }
            }
        }
    }
    *out = rgba;
}

// ---- LUT bake (optical_depth.gdshader:17-31,45-68): exact evaluation, one texel per lane ---------------
__global__ __launch_bounds__(256) void atmo_bake_kernel(const BakeConsts bc) {
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const int j = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= bc.w || j >= bc.h) return;
    const float u = ieee_div((float)i + 0.5f, (float)bc.w);
    const float v = ieee_div((float)j + 0.5f, (float)bc.h);
    const float diry = 2.0f * u - 1.0f;
    const float dirx = ieee_sqrt(1.0f - diry * diry);
    const float posx = 0.0f, posy = bc.planet_radius + bc.atmosphere_height * v;
    // ray_sphere(vec3(0), R+H, vec3(pos,0), vec3(dir,0))
    const float radius = bc.planet_radius + bc.atmosphere_height;
    const float ocx = posx - 0.0f, ocy = posy - 0.0f, ocz = 0.0f - 0.0f;
    const float b = ocx * dirx + ocy * diry + ocz * 0.0f;
    const float qx = ocx - b * dirx, qy = ocy - b * diry, qz = ocz - b * 0.0f;
    float hh = radius * radius - (qx * qx + qy * qy + qz * qz);
    float rsx = 1000000.0f, rsy = 1000000.0f;
    if (!(hh < 0.0f)) {
        hh = ieee_sqrt(hh);
        rsx = -b - hh;
        rsy = -b + hh;
    }
    const float ray_len = rsy - fmaxf(rsx, 0.0f);
    const float step_len = ieee_div(ray_len, (float)bc.steps);
    float od = 0.0f;
    for (int s = 0; s < bc.steps; ++s) {
        const float x = posx + dirx * step_len * (float)s;
        const float y = posy + diry * step_len * (float)s;
        const float d = ieee_sqrt(x * x + y * y);
        const float sd = d - bc.planet_radius;
        const float hgt = fminf(fmaxf(ieee_div(sd, bc.atmosphere_height), 0.0f), 1.0f);
        const float yy = 1.0f - hgt;
        const float density = yy * yy * yy * bc.density;
        od += density * step_len * bc.density;
    }
    bc.out[j * bc.w + i] = od;
}

// ---- launchers -----------------------------------------------------------------------------------------
template <int FLAGS>
static hipError_t launch_t(const RenderConsts &rc, hipStream_t stream) {
    dim3 grid((rc.x1 - rc.x0 + TILE_W - 1) / TILE_W, (rc.y1 - rc.y0 + TILE_H - 1) / TILE_H);
    hipLaunchKernelGGL(atmo_render_kernel<FLAGS>, grid, dim3(TILE_W * TILE_H), 0, stream, rc);
    return hipGetLastError();
}

hipError_t launch_render(int flags, const RenderConsts &rc, hipStream_t stream) {
    switch (flags) {
    case 0: return launch_t<0>(rc, stream);
    case KF_LIGHT_DIRECT: return launch_t<KF_LIGHT_DIRECT>(rc, stream);
    case KF_CLOUDS: return launch_t<KF_CLOUDS>(rc, stream);
    case KF_CLOUDS | KF_LIGHT_DIRECT: return launch_t<KF_CLOUDS | KF_LIGHT_DIRECT>(rc, stream);
    case KF_CLOUDS | KF_CLOUD_LIGHT_RM: return launch_t<KF_CLOUDS | KF_CLOUD_LIGHT_RM>(rc, stream);
    case KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT: return launch_t<KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT>(rc, stream);
    default: return hipErrorInvalidValue;
    }
}

const char *render_kernel_name(int flags) {
    switch (flags) {
    case 0: return "atmo_render_kernel<0>";
    case KF_LIGHT_DIRECT: return "atmo_render_kernel<4>";
    case KF_CLOUDS: return "atmo_render_kernel<1>";
    case KF_CLOUDS | KF_LIGHT_DIRECT: return "atmo_render_kernel<5>";
    case KF_CLOUDS | KF_CLOUD_LIGHT_RM: return "atmo_render_kernel<3>";
    case KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT: return "atmo_render_kernel<7>";
    default: return "?";
    }
}

hipError_t launch_bake(const BakeConsts &bc, hipStream_t stream) {
    dim3 grid((bc.w + 15) / 16, (bc.h + 15) / 16);
    hipLaunchKernelGGL(atmo_bake_kernel, grid, dim3(256), 0, stream, bc);
    return hipGetLastError();
}

}  // namespace atmo
