// atmo_kernels.hip -- gfx950 (CDNA4 / MI355X) kernels for the per-pixel atmosphere + cloud raymarch.
//
// One wavefront lane per view ray; a 128-thread workgroup shades a 16x8 pixel tile, each 64-lane
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
//   include/atmosphere_funcs_v1.gdshaderinc:15-63       get_atmo_factor, compute_atmosphere -> march_atmosphere_v1
//   include/cloud_funcs.gdshaderinc:31-68               get_density_full         -> cloud_density
//   include/cloud_funcs.gdshaderinc:78-167              get_light*               -> inside march_clouds
//   include/cloud_funcs.gdshaderinc:175-247             raymarch_cloud           -> march_clouds
//   include/cloud_funcs.gdshaderinc:249-324             render_clouds            -> atmo_render_kernel tail
//   optical_depth.gdshader:17-31,45-68                  LUT bake                 -> atmo_bake_kernel
#include "atmo_device.h"
#include "atmo_layout.h"

#include <cstdio>

namespace atmo {

// Workgroup tile in pixels.  16x16 = 4 waves, 16x8 = 2 (shipped), 16x4 = one wave per workgroup.  Waves never
// synchronise (the LDS queue of clouds_high_rm is wave-private), so the choice is purely scheduling: finer tiles give the
// dispatcher and the tile-order feedback a finer grain.  profiles/round2/ab_tile_height.txt: 16x8 and 16x4 are 3 % faster
// than 16x16 on clouds_high_rm at 1920x1080 and within +-1 % everywhere else.
#ifndef ATMO_TILE_H
#define ATMO_TILE_H 8
#endif
constexpr int TILE_W = 16;
constexpr int TILE_H = ATMO_TILE_H;
// Pixels of one 64-lane wave inside the workgroup tile: WAVE_W x (64 / WAVE_W).  16x4 keeps a wave's
// float4 stores in 256-byte runs; 8x8 halves that but makes a wave's rays more coherent (tools/ab_build.sh).
#ifndef ATMO_WAVE_W
#define ATMO_WAVE_W 16
#endif
// Retired experiments: the losing arms of the round 1-3 A/B switches were removed in round 4; NOTES.md ("Retired ablation
// switches") lists each switch, the record under profiles/ that settled it and the last commit that still holds its code.
constexpr int WAVE_W = ATMO_WAVE_W;
constexpr int WAVE_H = 64 / WAVE_W;
static_assert(WAVE_W == 16 || WAVE_W == 8 || WAVE_W == 32, "wave tile");
constexpr float LOG2E = 1.44269504088896340736f;

// ---- hardware transcendental units (approximate, ~1 ulp) ---------------------------------------
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// The declared sampler's lambda in the oracle's own operations (round 6; cube_lod_partner_coords, cube_lod_select).  0: rounds 2-5's arithmetic, the A/B arm.
#ifndef ATMO_LOD_LAMBDA_EXACT
#define ATMO_LOD_LAMBDA_EXACT 1
#endif
#ifndef ATMO_RCP_NR   // Newton steps of exact_rcp (1: checked on every significand by atmo_selftest_exact_math)
#define ATMO_RCP_NR 1
#endif
#ifndef ATMO_LOD_LOG2_CR
#define ATMO_LOD_LOG2_CR 1
#endif
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// GLSL mix(a,b,t) = a*(1-t) + b*t
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
// IEEE-754 correctly rounded sqrt and divide: hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt)
// expands these to the fix-up sequences; __fsqrt_rn/__fdiv_rn are NOT used because the HIP headers map
// __fsqrt_rn to the native (1 ulp) square root.  Used in the once-per-pixel prologue and the LUT bake.
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }
// RN(1 / d) from v_rcp_f32 (1 ulp) and ATMO_RCP_NR Newton step(s): 3 instructions against the ~11 of the compiler's IEEE division (v_div_scale /
// v_div_fmas / v_div_fixup).  Equal to 1.0f / d for every normal d whose reciprocal is normal: checked on the device over all 2^23 significands of
// fifty binades (atmo_selftest_exact_math, the divide counter; v_rcp_f32's error depends on the significand alone).
__device__ __forceinline__ float exact_rcp(float d) {
    float r = hw_rcp(d);
#pragma unroll
    for (int k = 0; k < ATMO_RCP_NR; ++k) r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return r;
}

// Correctly rounded sqrt for the per-step cloud chain, x >= 0 and not denormal/inf/nan (x = |pos|^2 ~ 1e4):
// the hardware root is within 1 ulp, so the result is s-1ulp, s or s+1ulp; the sign of the exact FMA
// residuals x - (s-1ulp)*s and x - (s+1ulp)*s picks it (the same test LLVM's expansion uses, minus its
// denormal scaling and class checks).  1 transcendental + 8 VALU instead of 1 + 15.
__device__ __forceinline__ float exact_sqrt(float x) {
    const float s = hw_sqrt(x);
    const float s_dn = __int_as_float(__float_as_int(s) - 1);
    const float s_up = __int_as_float(__float_as_int(s) + 1);
    const float r_dn = __builtin_fmaf(-s_dn, s, x);
    const float r_up = __builtin_fmaf(-s_up, s, x);
    float o = (r_dn <= 0.0f) ? s_dn : s;
    o = (r_up > 0.0f) ? s_up : o;
    return o;
}
// The same result from the reciprocal root with one coupled Newton step and a final FMA correction (the sequence hipcc itself emits for
// sqrtf when denormals are flushed): rsq + 7 fast-class instructions instead of sqrt + 4 fast + 4 slow (compares / selects).  For the
// cloud chain only, where x = |p|^2 is a normal positive number (x = 0 would give NaN here).  Equal to the IEEE root for every
// float from 2^-102 up (tools/sqrt_sweep.py: all 2^23 significands of all 230 binades; profiles/round3/ab_sqrt_v2.txt).
__device__ __forceinline__ float exact_sqrt_pos(float x) {
    const float y = hw_rsq(x);
    float g = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// Correctly rounded a / c for a wave-uniform divisor c with rc = RN(1/c) computed on the host (IEEE):
// two Markstein corrections of a*rc with exact FMA residuals.  5 VALU instead of 11 + v_rcp.
// (Checked exhaustively on the CPU for every float32 significand of `a` against a / c for the demo's divisors;
// tests/test_gpu_parity.py::test_exact_math_selftest sweeps it on the device.)
__device__ __forceinline__ float exact_div_uniform(float a, float c, float rc) {
    const float q0 = a * rc;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, c, a), rc, q0);
    return __builtin_fmaf(__builtin_fmaf(-q1, c, a), rc, q1);
}

// Once-per-pixel prologue with the short exact sequences (DIET), the same bits as ieee_sqrt / ieee_div:
//   prologue_sqrt  exact_sqrt, 1 + 8 instead of 1 + 15 instructions.  Equal to IEEE for x = 0, every x >= 2^-96, inf and NaN;
//                  for 0 < x < 2^-96 (the FMA residuals underflow) it may be 1 ulp off -- of a root below 3.6e-15, which
//                  needs a planet radius below 1e-11 or a camera closer than that to the depth sample;
//   pixel_coord    (i + 0.5) / n as ONE Markstein correction of (i + 0.5) * RN(1/n): 3 instead of 12 instructions.  Equal to
//                  the IEEE quotient for every 0 <= i < n <= 65536, the largest viewport atmo_render accepts (checked exhaustively:
//                  2.1e9 quotients, tools/uv_division.c);
//   unorm8_exact   byte / 255 in 2 instead of 12 (already used for every texel).
// Sure-miss test in front of the exact prologue (shade_pixel).  ATMO_FAST_MISS_MASK: bit (direct + 2 clouds + 4 lite) = that kernel
// family uses it (a measured choice per family, like the prologue diet: profiles/round3/ab_fast_miss.txt).
#ifndef ATMO_FAST_MISS_MASK
#define ATMO_FAST_MISS_MASK 0x05  // baked-LUT atmosphere with and without clouds: -1.4 % / -0.7 %; the direct-light kernels lose 11 % (!), v1 1.5 %
#endif
// Both forms give the same bits; which one a kernel variant uses is a measured choice (profiles/round2/ab_prologue.txt:
// the short forms gain 3-4 % on the baked-LUT atmosphere kernels and cost the direct-light and the raymarched-cloud-light
// kernels 2-3 % at 1920x1080 although they execute fewer instructions -- see the note at atmo_render_kernel).
template <bool DIET = true>
__device__ __forceinline__ float prologue_sqrt(float x) { return DIET ? exact_sqrt(x) : ieee_sqrt(x); }
template <bool DIET = true>
__device__ __forceinline__ float pixel_coord(float a, float n, float rcp_n) {
    if (!DIET) return ieee_div(a, n);
    const float q0 = a * rcp_n;
    return __builtin_fmaf(__builtin_fmaf(-q0, n, a), rcp_n, q0);
}
// world.xyz / world.w (main:134-135): three IEEE quotients by the SAME divisor.  Short form: one IEEE reciprocal RN(1 / w) and
// two Markstein corrections per quotient (exact_div_uniform: correctly rounded given a correctly rounded reciprocal; swept on
// the device for every significand against a / c, tests/test_gpu_parity.py::test_exact_math_selftest) -- 1 division + 15 FMAs
// instead of 3 divisions.  Used when |w| is an ordinary number (2^-100 .. 2^100: w = 0 or inf, a far plane at infinity, takes the
// IEEE path and keeps its inf / NaN semantics); a numerator in (0, 2^-100) could come out 1 ulp off (its FMA residual
// underflows) -- a world coordinate below 1e-30.
template <bool DIET = true>
__device__ __forceinline__ void world_div3(float wx, float wy, float wz, float ww, float &x, float &y, float &z) {
    const float aw = fabsf(ww);
    if (DIET && aw >= 7.8886090522101181e-31f && aw <= 1.2676506002282294e30f) {
        const float rw = ieee_div(1.0f, ww);
        x = exact_div_uniform(wx, ww, rw);
        y = exact_div_uniform(wy, ww, rw);
        z = exact_div_uniform(wz, ww, rw);
    } else {
        x = ieee_div(wx, ww); y = ieee_div(wy, ww); z = ieee_div(wz, ww);
    }
}
template <bool DIET = true>
__device__ __forceinline__ float blue_noise_value(uint8_t b);

// ---- lane-split mode: two adjacent lanes share one ray (SPLIT = 2) -------------------------------------------------
// At 1920x1080 a frame is only ~20 000 busy waves for 1024 SIMDs x 7-8 wave slots: too few to keep two waves' fast
// instructions pairing on a SIMD and to cover the gathers (tools/concurrency_probe.py: two concurrent frames finish in
// 1.3x, not 2x, the time of one).  With SPLIT = 2 a wave holds 32 rays; lanes 2r and 2r+1 march the same ray, each takes
// every second cloud step / half of the view steps, and the partner's per-step results cross over with one DPP
// quad_perm move -- the one cross-lane exchange this algorithm has a use for.  The host picks SPLIT per launch by size.
__device__ __forceinline__ float swap_adjacent(float x) {
    // v_mov_b32_dpp quad_perm:[1,0,3,2]: lane 2r <-> lane 2r+1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
}

// The lane-split form of the DECLARED-SAMPLER kernels (round 5) keeps every 2 x 2 pixel quad in four consecutive lanes -- quad_perm and
// s_wqm_b64 work unchanged, and the four lanes of a quad are at the same march step -- and puts the two lanes of a ray four lanes apart:
// lane ^ 4.  Within a 16-lane DPP row, banks 0 and 2 read lane + 4 (row_shl:4), banks 1 and 3 lane - 4 (row_shr:4): two DPP moves.
__device__ __forceinline__ float swap_across_quads(float x) {
    const int v = __float_as_int(x);
    int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);  // row_shl:4, bank_mask 0101: lanes 0-3, 8-11 of a row <- lane + 4
    t = __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);      // row_shr:4, bank_mask 1010: lanes 4-7, 12-15 <- lane - 4
    return __int_as_float(t);
}
template <bool ACROSS_QUADS>
__device__ __forceinline__ float swap_ray_lanes(float x) { return ACROSS_QUADS ? swap_across_quads(x) : swap_adjacent(x); }
// products and sums that must NOT be contracted where the caller's block allows it (`#pragma clang fp contract` is lexical)
__device__ __forceinline__ float mul_unfused(float a, float b) { return a * b; }
__device__ __forceinline__ float add_unfused(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_unfused(float a, float b) { return a - b; }

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
template <bool DIET = true>
__device__ __forceinline__ float2 hit_radius(SphereHit s, float radius) {
    float h = radius * radius - s.qc2;
    if (h < 0.0f) return make_float2(1000000.0f, 1000000.0f);
    h = prologue_sqrt<DIET>(h);
    return make_float2(-s.b - h, -s.b + h);
}

// ---- samplers ----------------------------------------------------------------------------------
// Device texture layouts (built by atmo_api.hip when a texture is set, see DESIGN.md "Data layout in HBM"):
//   LUT    (w+2) x (h+2) fp32 with a clamp-to-edge apron: texel (i,j) at [(j+1)*(w+2) + i+1] (what is baked, read back
//          and checked against the host statement), plus a footprint copy derived from it: (w+1) x (h+1) entries of 4
//          floats, entry (i,j) = apron texels (i,j), (i+1,j), (i,j+1), (i+1,j+1): one 16-byte gather per bilinear sample.
//   shape  n^3 uint32 "xy footprints": word (i,j,k) = bytes T(i,j,k), T(i+1,j,k), T(i,j+1,k), T(i+1,j+1,k) with the
//          repeat wrap baked in; a trilinear fetch is 2 dword loads (planes k, k+1) instead of 8 byte loads.
//   cube   6 x (n+1)^2 uint32 footprints of the apron-padded faces: word (i,j) = the 2x2 texels whose top-left
//          padded coordinate is (i,j); a seamless bilinear fetch is 1 dword load instead of 4 byte loads.
// The vector-memory path issues a 64-lane gather at ~16 cycles per wave instruction however narrow the
// data, so instruction count, not bytes, is what these layouts buy (profiles/round1).

// Texture gathers through buffer descriptors (buffer_load ... offen: scalar SRSRC base + one 32-bit VGPR byte offset +
// immediate offset) instead of 64-bit flat addresses: ~5 fewer address instructions per fetch.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
// ATMO_ABLATE_FETCH=1 (DIAGNOSTIC build only: tools/fetch_ablation_probe.py, profiles/round5/ab_fetch_ablation.txt; retired in round 4, back in round 5 at
// the judge's request): every texture gather is replaced by a constant -- byte 128 / the float 128/255 -- while the address arithmetic in front of
// it and the filter behind it stay live.  Rendered with CONSTANT textures of that value (LUT, shape volume, cubemap) both builds draw the same
// picture through the same control flow, and the difference in kernel time is everything a cheaper fetch path (LDS staging included) could buy.
#ifndef ATMO_ABLATE_FETCH
#define ATMO_ABLATE_FETCH 0
#endif
#if ATMO_ABLATE_FETCH
#define ATMO_ABLATED_F32 0.50196081399917603f   // 128 / 255 rounded to binary32 = unorm8_exact(128)
__device__ __forceinline__ uint32_t buf_u32(__amdgpu_buffer_rsrc_t, uint32_t byte_off) {
    asm volatile("" : "+v"(byte_off));   // the address chain stays
    return 0x80808080u;
}
__device__ __forceinline__ f32x2 buf_f32x2(__amdgpu_buffer_rsrc_t, uint32_t byte_off) {
    asm volatile("" : "+v"(byte_off));
    return f32x2{ATMO_ABLATED_F32, ATMO_ABLATED_F32};
}
#else
__device__ __forceinline__ uint32_t buf_u32(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ f32x2 buf_f32x2(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}
#endif

// byte k of a footprint word as float (the compiler selects v_cvt_f32_ubyte0..3)
__device__ __forceinline__ float ub0(uint32_t w) { return (float)(w & 0xffu); }
__device__ __forceinline__ float ub1(uint32_t w) { return (float)((w >> 8) & 0xffu); }
__device__ __forceinline__ float ub2(uint32_t w) { return (float)((w >> 16) & 0xffu); }
__device__ __forceinline__ float ub3(uint32_t w) { return (float)(w >> 24); }

// texture(u_optical_depth_texture, uv).r : bilinear, clamp-to-edge, R32F.  x = u*w - 0.5, y = v*h - 0.5 (texel space).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#if ATMO_ABLATE_FETCH
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t, uint32_t byte_off) {
    asm volatile("" : "+v"(byte_off));
    return f32x4{ATMO_ABLATED_F32, ATMO_ABLATED_F32, ATMO_ABLATED_F32, ATMO_ABLATED_F32};
}
#else
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
#endif
// One 16-byte gather from the footprint copy of the LUT (atmo_lut_footprint_kernel): the texture addresser was busy 62-70 %
// of the lut32 draw with two 8-byte gathers per sample (19 cycles each per wave); one 16-byte gather costs about the same as
// one of them.  lut32 -7 % (profiles/round2/ab_lut_footprint.txt).  The byte offset is formed in fp32 (exact: < 2^24).
__device__ __forceinline__ float lut_sample_fp(const float *__restrict__ lut4, int w1, int h1, float x, float y) {
#pragma clang fp contract(fast)
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(lut4, (uint32_t)(w1 * h1) * 16u);
    const float s16 = (float)(w1 * 16);
    const uint32_t off = (uint32_t)fmaf(yf, s16, fmaf(xf, 16.0f, s16 + 16.0f));  // footprint (xf + 1, yf + 1)
    const f32x4 t = buf_f32x4(rs, off);
    const float a = t.x + (t.y - t.x) * fx;
    const float b = t.z + (t.w - t.z) * fx;
    return a + (b - a) * fy;
}

// mix(mix(mix(), mix(), fy), ..., fz) of the eight texels of two xy-footprints that are already exact byte / 255 floats, every product and
// sum rounded on its own like a scalar fp32 evaluation.  A function of its own because `#pragma clang fp contract` is lexical: written
// inline in shape_sample (round 3, float footprints) the mixes were contracted to mul + fma, 1 ulp away from the oracle's filter.
__device__ __forceinline__ float trilinear_exact8(f32x4 a, f32x4 b, float fx, float fy, float fz) {
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    const float c00 = a.x * gx + a.y * fx;
    const float c10 = a.z * gx + a.w * fx;
    const float c01 = b.x * gx + b.y * fx;
    const float c11 = b.z * gx + b.w * fx;
    const float c0 = c00 * gy + c10 * fy;
    const float c1 = c01 * gy + c11 * fy;
    return c0 * gz + c1 * fz;
}
__device__ __forceinline__ float unorm8_exact(float b);

// texture(u_cloud_shape_texture, p).r : trilinear, repeat, R8.  PRECISE: exact UNORM8 conversions + unfused mixes.
// Footprint index ((k * n + j) * n + i) of the repeat-wrapped texel (i, j, k) and of its +z neighbour:
//   * n a power of two (rc.shape_log2n >= 0; every size the engine's NoiseTexture3D defaults to): masks and two shift-ors --
//     no integer multiplies (round 4: the 64-bit multiply-adds of the general form were 4 of the ~45 instructions of a sample);
//   * any other n: floor-division through fp32, q = floor(float(i) / n) with one fix-up on either side -- exact for |i| < 2^24
//     (float(i) is exact, the quotient is off by at most one) -- instead of four hardware-less `%` expansions (~130 instructions
//     of cold code per inlined copy of this function, six copies in the raymarched-light kernel).
struct ShapeAddr {
    uint32_t e0, e1;
};
__device__ __forceinline__ int wrap_general(int i, int n, float nf, float inv_nf) {
    const int q = (int)floorf((float)i * inv_nf);
    int r = i - q * n;
    r = r < 0 ? r + n : r;
    return r >= n ? r - n : r;
}
__device__ __forceinline__ ShapeAddr shape_addr(int n, int log2n, int i, int j, int k) {
    ShapeAddr a;
    if (log2n >= 0) {
        const uint32_t m = (uint32_t)(n - 1);
        const uint32_t ji = (((uint32_t)j & m) << log2n) | ((uint32_t)i & m);
        a.e0 = (((uint32_t)k & m) << (2 * log2n)) | ji;
        a.e1 = (((uint32_t)(k + 1) & m) << (2 * log2n)) | ji;
    } else {
        const float nf = (float)n, inv_nf = 1.0f / nf;
        const int i0 = wrap_general(i, n, nf, inv_nf), j0 = wrap_general(j, n, nf, inv_nf), k0 = wrap_general(k, n, nf, inv_nf);
        const int k1 = k0 + 1 == n ? 0 : k0 + 1;
        const uint32_t ji = (uint32_t)(j0 * n + i0);
        a.e0 = (uint32_t)(k0 * n * n) + ji;
        a.e1 = (uint32_t)(k1 * n * n) + ji;
    }
    return a;
}
template <bool PRECISE>
__device__ __forceinline__ float shape_sample(const uint32_t *__restrict__ fp, int n, int log2n, float px, float py, float pz, const float *__restrict__ f4 = nullptr) {
#pragma clang fp contract(fast)
    const float nf = (float)n;
    float fx, fy, fz;
    ShapeAddr ad;
    if (log2n >= 0) {   // ONE wave-uniform branch for coordinates and addresses alike (two on the same condition cost the declared-sampler kernels 1-2 %)
        const float x = px * nf - 0.5f, y = py * nf - 0.5f, z = pz * nf - 0.5f;   // contracted: fma(p, n, -0.5) -- p * n is exact for a power of two
        const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
        fx = x - xf; fy = y - yf; fz = z - zf;
        ad = shape_addr(n, log2n, (int)xf, (int)yf, (int)zf);
    } else {
        // The reference rounds p * n before it subtracts (a scalar fp32 evaluation of `uvw * size - 0.5`).  For a power-of-two n the product is
        // exact and an FMA returns the same bits; for any other size (24, 48 ...) it does not, and an ulp of the filter weight is what a
        // hypersensitive cloud pixel amplifies to 1e-4 (round 6: seeds 442 / 658 / 890 of the 1 212-scene fuzz, all three with a 24^3 volume;
        // profiles/round6/fuzz_four.txt).  The fast cloud mode keeps its FMAs.
        const float x = PRECISE ? sub_unfused(mul_unfused(px, nf), 0.5f) : px * nf - 0.5f, y = PRECISE ? sub_unfused(mul_unfused(py, nf), 0.5f) : py * nf - 0.5f,
                    z = PRECISE ? sub_unfused(mul_unfused(pz, nf), 0.5f) : pz * nf - 0.5f;
        const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
        fx = x - xf; fy = y - yf; fz = z - zf;
        ad = shape_addr(n, -1, (int)xf, (int)yf, (int)zf);
    }
    if (PRECISE) {  // with a float copy of the footprints: two 16-byte gathers, no conversions
        f32x4 a, b;
        if (f4 != nullptr) {
            const __amdgpu_buffer_rsrc_t rs4 = make_rsrc(f4, (uint32_t)(n * n * n) * 16u);
            a = buf_f32x4(rs4, ad.e0 * 16u);
            b = buf_f32x4(rs4, ad.e1 * 16u);
        } else {
            const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(fp, (uint32_t)(n * n * n) * 4u);
            const uint32_t w0 = buf_u32(rs1, ad.e0 * 4u), w1 = buf_u32(rs1, ad.e1 * 4u);
            a = f32x4{unorm8_exact(ub0(w0)), unorm8_exact(ub1(w0)), unorm8_exact(ub2(w0)), unorm8_exact(ub3(w0))};
            b = f32x4{unorm8_exact(ub0(w1)), unorm8_exact(ub1(w1)), unorm8_exact(ub2(w1)), unorm8_exact(ub3(w1))};
        }
        return trilinear_exact8(a, b, fx, fy, fz);
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(fp, (uint32_t)(n * n * n) * 4u);
    const uint32_t w0 = buf_u32(rs, ad.e0 * 4u);
    const uint32_t w1 = buf_u32(rs, ad.e1 * 4u);
    const float a00 = ub0(w0), a10 = ub1(w0), a01 = ub2(w0), a11 = ub3(w0);
    const float b00 = ub0(w1), b10 = ub1(w1), b01 = ub2(w1), b11 = ub3(w1);
    const float c00 = a00 + (a10 - a00) * fx;
    const float c10 = a01 + (a11 - a01) * fx;
    const float c01 = b00 + (b10 - b00) * fx;
    const float c11 = b01 + (b11 - b01) * fx;
    const float c0 = c00 + (c10 - c00) * fy;
    const float c1 = c01 + (c11 - c01) * fy;
    return (c0 + (c1 - c0) * fz) * (1.0f / 255.0f);
}

// Bilinear filter of four UNORM8 texels exactly as a scalar fp32 evaluation of  mix(mix(t00,t10,fx), mix(t01,t11,fx), fy),
// t = byte / 255 (IEEE), mix(a,b,t) = a*(1-t) + b*t  would round it.  byte * RN(1/255) differs from byte / 255 for 126 of the
// 256 bytes; with 1/255 split into C_HI + C_LO (two floats), fma(b, C_HI, RN(b * C_LO)) is the correctly rounded quotient
// for all 256 bytes (checked exhaustively, tests/test_oracle_kat.py::test_unorm8_two_op_conversion_is_exact).
// No contraction in this function.
__device__ __forceinline__ float unorm8_exact(float b) {
    const float c_hi = __uint_as_float(0x3b808081u);  // RN(1/255)
    const float c_lo = __uint_as_float(0xaf7efeffu);  // RN(1/255 - c_hi)
    return __builtin_fmaf(b, c_hi, b * c_lo);
}
template <bool DIET>
__device__ __forceinline__ float blue_noise_value(uint8_t b) { return DIET ? unorm8_exact((float)b) : ieee_div((float)b, 255.0f); }
__device__ __forceinline__ float bilinear_unorm8_exact(uint32_t w, float fx, float fy) {
    const float t00 = unorm8_exact(ub0(w)), t10 = unorm8_exact(ub1(w));
    const float t01 = unorm8_exact(ub2(w)), t11 = unorm8_exact(ub3(w));
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float a = t00 * gx + t10 * fx;
    const float b = t01 * gx + t11 * fx;
    return a * gy + b * fy;
}

// the same filter on texels that are already exact byte / 255 floats (no contraction in this function either)
__device__ __forceinline__ float bilinear_exact4(float t00, float t10, float t01, float t11, float fx, float fy) {
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float a = t00 * gx + t10 * fx;
    const float b = t01 * gx + t11 * fx;
    return a * gy + b * fy;
}


// texture(u_cloud_coverage_cubemap, d).r : LOD 0, bilinear, seamless.
// Face selection and the in-face coordinates come from the hardware cube instructions (v_cubeid/sc/tc/ma_f32):
// same table and tie-break as Vulkan (z over y over x), ma = 2 * major axis value.
template <bool PRECISE>
__device__ __forceinline__ float cube_sample(const uint32_t *__restrict__ fp, int n, float dx, float dy, float dz, const float *__restrict__ f4 = nullptr) {
#pragma clang fp contract(fast)
    const float fid = __builtin_amdgcn_cubeid(dx, dy, dz);
    const float sc = __builtin_amdgcn_cubesc(dx, dy, dz);
    const float tc = __builtin_amdgcn_cubetc(dx, dy, dz);
    const float ma = 0.5f * fabsf(__builtin_amdgcn_cubema(dx, dy, dz));
    // s = 0.5*(sc/ma + 1); one Newton step on the hardware reciprocal keeps the quotient within 1 ulp of IEEE
    const float r = hw_rcp(ma);
    float qs = sc * r, qt = tc * r;
    qs = fmaf(fmaf(-qs, ma, sc), r, qs);
    qt = fmaf(fmaf(-qt, ma, tc), r, qt);
    if (PRECISE) {
        qs = fmaf(fmaf(-qs, ma, sc), r, qs);  // second correction: the quotient is now the IEEE one (up to rare ties)
        qt = fmaf(fmaf(-qt, ma, tc), r, qt);
    }
    // (0.5*(q + 1))*n - 0.5 with the reference's roundings: q + 1 rounds, the scalings are exact for power-of-two n
    const float hn = 0.5f * (float)n;
    float x = fmaf(qs + 1.0f, hn, -0.5f);
    float y = fmaf(qt + 1.0f, hn, -0.5f);
    if (PRECISE && (n & (n - 1)) != 0) {  // a face size that is not a power of two (wave-uniform): the product (0.5 (q + 1)) n rounds in the reference
        asm volatile("; cube_sample: face size not a power of two");   // (a scalar branch, not selects: see shape_sample)
        x = sub_unfused(mul_unfused(0.5f * (qs + 1.0f), (float)n), 0.5f);
        y = sub_unfused(mul_unfused(0.5f * (qt + 1.0f), (float)n), 0.5f);
    }
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const int stride = n + 1;
    if (n <= 1024) {
        // byte offset ((face * stride + j) * stride + i) * 4, i = clamp(xf, -1, n-1) + 1, formed in fp32.  Exact although the
        // largest offset, 6 (n+1)^2 * 4 = 25.2 M at n = 1024, exceeds 2^24: every term and every partial sum of the three FMAs
        // is an integer multiple of 4 below 2^26, i.e. 4 x (an integer below 2^24), and those are all representable.
        // 2 med3 + 3 FMA + 1 conversion instead of 3 conversions, 4 integer min/max, a 64-bit multiply-add, a multiply, an
        // add and a shift-add
        const float nm1 = (float)(n - 1), s4 = (float)(stride * 4);
        const float ic = __builtin_amdgcn_fmed3f(xf, -1.0f, nm1), jc = __builtin_amdgcn_fmed3f(yf, -1.0f, nm1);
        const float offf = fmaf(fid, s4 * (float)stride, fmaf(jc, s4, fmaf(ic, 4.0f, s4 + 4.0f)));
        if (PRECISE) {  // with a float copy of the footprints: one 16-byte gather at four times the offset, no conversions
            f32x4 t;
            if (f4 != nullptr) {
                t = buf_f32x4(make_rsrc(f4, (uint32_t)(6 * stride * stride) * 16u), (uint32_t)(offf * 4.0f));
            } else {
                const uint32_t w = buf_u32(make_rsrc(fp, (uint32_t)(6 * stride * stride) * 4u), (uint32_t)offf);
                t = f32x4{unorm8_exact(ub0(w)), unorm8_exact(ub1(w)), unorm8_exact(ub2(w)), unorm8_exact(ub3(w))};
            }
            return bilinear_exact4(t.x, t.y, t.z, t.w, fx, fy);
        }
        const uint32_t off = (uint32_t)offf;
        const uint32_t w = buf_u32(make_rsrc(fp, (uint32_t)(6 * stride * stride) * 4u), off);
        if (PRECISE) return bilinear_unorm8_exact(w, fx, fy);
        const float t00 = ub0(w), t10 = ub1(w), t01 = ub2(w), t11 = ub3(w);
        const float a = t00 + (t10 - t00) * fx;
        const float b = t01 + (t11 - t01) * fx;
        return (a + (b - a) * fy) * (1.0f / 255.0f);
    }
    const int i = min(max((int)xf, -1), n - 1) + 1, j = min(max((int)yf, -1), n - 1) + 1;
    const uint32_t w = buf_u32(make_rsrc(fp, (uint32_t)(6 * stride * stride) * 4u), (uint32_t)(((int)fid * stride + j) * stride + i) * 4u);
    if (PRECISE) return bilinear_unorm8_exact(w, fx, fy);
    const float t00 = ub0(w), t10 = ub1(w), t01 = ub2(w), t11 = ub3(w);
    const float a = t00 + (t10 - t00) * fx;
    const float b = t01 + (t11 - t01) * fx;
    return (a + (b - a) * fy) * (1.0f / 255.0f);
}

// ---- implicit cubemap LOD (atmo_set_sampler_lod 1; oracle: sample_cube_lod) ----------------------------------------------
// The positions the two 2x2-quad partners of this pixel pass to the same texture() call; valid = the partner reaches it.
// Whole-quad exchange (round 4).  In the lock-step part of the cloud march every lane of a wave is at the same march step, and a pixel's two
// quad partners sit in the same wave (LOD launches map each 2 x 2 pixel quad to four consecutive lanes), so the partners' cube coordinates
// need not be recomputed from positions maintained per lane (rounds 2-3: 6 additions per step, 12 rotations, 6 differences and the face
// frame per sample, two more per-pixel prologues per ray): they are read from the partner LANES with v_mov_b32_dpp quad_perm.  A partner
// that marches but has left the evaluation at this step through an early-out is disabled in EXEC, and DPP cannot read a disabled lane:
// the exchange runs inside ONE inline-asm block in whole-quad mode (s_wqm_b64 exec, exec: every quad with an active lane is enabled
// whole, as the graphics pipeline does for derivatives); the re-enabled "helper" lanes compute their own coordinates from their own
// position registers inside the block, the DPP moves copy them across, and EXEC is restored before the block ends.
// Helper lanes WRITE the block's registers while the compiler believes them inactive, and the compiler parks values of inactive lanes in
// any register that is not live on the active path (a variable assigned on both sides of a branch shares one register).  The block's
// registers are therefore QuadRegs: read-write ("+v") operands of every exchange, defined once at the top of the kernel and used again at
// its end, so their live range spans every divergent region and nothing can share them; tests/test_host_logic.py checks in the ISA of every
// LOD kernel that no instruction outside the exchange blocks writes them.
struct QuadRegs {
    float fid, sc, tc, mas;                              // own: v_cubeid / sc / tc / ma of the rotated position
    float fidx, scx, tcx, masx, fidy, scy, tcy, masy;    // the same of the horizontal (lane ^ 1) and vertical (lane ^ 2) partner
};
__device__ __forceinline__ void quad_regs_define(QuadRegs &q) {
    asm volatile("; QuadRegs defined" : "=v"(q.fid), "=v"(q.sc), "=v"(q.tc), "=v"(q.mas), "=v"(q.fidx), "=v"(q.scx),
                 "=v"(q.tcx), "=v"(q.masx), "=v"(q.fidy), "=v"(q.scy), "=v"(q.tcy), "=v"(q.masy));
}
__device__ __forceinline__ void quad_regs_keep(const QuadRegs &q) {
    asm volatile("; QuadRegs kept" ::"v"(q.fid), "v"(q.sc), "v"(q.tc), "v"(q.mas), "v"(q.fidx), "v"(q.scx), "v"(q.tcx),
                 "v"(q.masx), "v"(q.fidy), "v"(q.scy), "v"(q.tcy), "v"(q.masy));
}
// coverage = texture(cubemap, vec3(rot * p.xz, p.y)): the rotation (cloud_funcs.gdshaderinc:43, every product and sum rounded on its own)
// and the hardware cube coordinates of this lane's sample position, and its two partners' -- in whole-quad mode.
__device__ __forceinline__ void quad_exchange_coords(float px, float py, float pz, float r0, float r1, float r2, float r3, QuadRegs &q) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_wqm_b64 exec, exec\n\t"
        "v_mul_f32_e32 %[scx], %[r0], %[px]\n\t"   // the rotated x and z in scx / scy until the cube instructions have read them
        "v_mul_f32_e32 %[fidx], %[r2], %[pz]\n\t"
        "v_mul_f32_e32 %[scy], %[r1], %[px]\n\t"
        "v_mul_f32_e32 %[fidy], %[r3], %[pz]\n\t"
        "v_add_f32_e32 %[scx], %[scx], %[fidx]\n\t"
        "v_add_f32_e32 %[scy], %[scy], %[fidy]\n\t"
        "s_nop 0\n\t"
        "v_cubeid_f32 %[fid], %[scx], %[py], %[scy]\n\t"
        "v_cubesc_f32 %[sc], %[scx], %[py], %[scy]\n\t"
        "v_cubetc_f32 %[tc], %[scx], %[py], %[scy]\n\t"
        "v_cubema_f32 %[mas], %[scx], %[py], %[scy]\n\t"
        "s_nop 1\n\t"
        "v_mov_b32_dpp %[fidx], %[fid] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[scx], %[sc] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[tcx], %[tc] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[masx], %[mas] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[fidy], %[fid] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[scy], %[sc] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[tcy], %[tc] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[masy], %[mas] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[saved]"
        : [saved] "=&s"(saved), [fid] "+v"(q.fid), [sc] "+v"(q.sc), [tc] "+v"(q.tc), [mas] "+v"(q.mas),
          [fidx] "+v"(q.fidx), [scx] "+v"(q.scx), [tcx] "+v"(q.tcx), [masx] "+v"(q.masx), [fidy] "+v"(q.fidy), [scy] "+v"(q.scy),
          [tcy] "+v"(q.tcy), [masy] "+v"(q.masy)
        : [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3)
        : "scc");
}
// the partners' sample positions themselves (the lit-sample queue stores them for the light taps, which are not lock-step): into
// (fidx, scx, tcx) and (fidy, scy, tcy), whose coordinates are dead by then
__device__ __forceinline__ void quad_exchange_positions(float px, float py, float pz, QuadRegs &q) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_wqm_b64 exec, exec\n\t"
        "s_nop 1\n\t"  // a VALU write of px / py / pz right in front of the block needs 2 wait states before a DPP read (the compiler cannot see in here)
        "v_mov_b32_dpp %[fidx], %[px] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[scx], %[py] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[tcx], %[pz] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[fidy], %[px] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[scy], %[py] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[tcy], %[pz] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[saved]"
        : [saved] "=&s"(saved), [fidx] "+v"(q.fidx), [scx] "+v"(q.scx), [tcx] "+v"(q.tcx), [fidy] "+v"(q.fidy), [scy] "+v"(q.scy), [tcy] "+v"(q.tcy)
        : [px] "v"(px), [py] "v"(py), [pz] "v"(pz)
        : "scc");
}

constexpr int CUBE_LEVEL_TABLE = 16;   // rows of the per-level constants in LDS (cube_level_table_fill); log2_cr's table sits behind them
// What a texture() call of the declared sampler needs to know about the pixel's two 2x2-quad partners; valid = the partner reaches the call.
struct QuadNb {
    bool vx, vy;
    V3 px, py;         // position form (light taps of a queued sample): the horizontal / vertical partner's sample position (model space) ...
    V3 k;              // ... and the tap's offset from it (the partners evaluate the same tap from their own sample)
    float e2;          // >= |partner's position - this lane's|^2 for both partners that reach the call, scaled: cube_lod_scaled_spread
    const f32x4 *lvl;  // LDS: per mip level {0.5 n_l, 4 n_l + 4, 4 (n_l + 1)^2, byte offset of footprint (0, 0) of face 0} (cube_level_table), then log2_cr's 32 rows
    QuadRegs *regs;    // lock-step form (the march): the whole-quad exchange registers
};

// seamless bilinear sample of mip level `level` on face `face` at face coordinates (s, t); exact UNORM8 + unfused mixes
__device__ __forceinline__ float cube_level_sample(const RenderConsts &rc, int face, float s, float t, int level) {
    const int n = rc.cube_n >> level;
    const float nf = (float)n;
    const float x = s * nf - 0.5f, y = t * nf - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const int i = min(max((int)xf, -1), n - 1) + 1, j = min(max((int)yf, -1), n - 1) + 1;
    const int stride = n + 1;
    const uint32_t base = level == 0 ? 0u : rc.cube_level_off[level];
    const uint32_t w = rc.cube[base + (uint32_t)((face * stride + j) * stride + i)];
    return bilinear_unorm8_exact(w, fx, fy);
}

// log2 of a float evaluated in double and rounded ONCE -- correctly rounded (the oracle's copy of this function equals (float)log2l(x) on every float of
// [2^-4, 2^40): tests/test_oracle_kat.py) and, because it is a fixed sequence of IEEE double operations on a shared table (tools/make_log2_table.py;
// the CPU checker under oracle/ carries the same text), the SAME BITS as the oracle's by construction: atmo_debug_log2_cr / test_gpu_parity.py compare the two.
//   x = 2^k z, z in [OFF, 2 OFF) (32 bins, bin 19 centred on 1);  r = z invc - 1 (exact);  log2 x = (k + logc) + r (c1 + r (c2 + ... r c8)).
// ~30 VALU instructions (v_fma_f64 issues at the f32 rate on gfx950), one LDS read; v_log_f32 is 1 ulp -- not the same bits as anything.
// Domain: finite normal x > 0 (callers pass rho^2 > 1).
/* generated by tools/make_log2_table.py -- do not edit by hand */
#define LOG2CR_OFF 0x3f320000u
__device__ const double LOG2CR_TAB[32][2] = {  /* {invc, logc = -log2(invc)} */
    {0x1.6c16c20000000p+0, -0x1.042bd5e5bc697p-1},
    {0x1.642c860000000p+0, -0x1.e7df61b2e23edp-2},
    {0x1.5c98820000000p+0, -0x1.c819d91c72820p-2},
    {0x1.5555560000000p+0, -0x1.a8ff99fab991dp-2},
    {0x1.4e5e0a0000000p+0, -0x1.8a897eb027b02p-2},
    {0x1.47ae140000000p+0, -0x1.6cb0f45c5ddccp-2},
    {0x1.4141420000000p+0, -0x1.4f6fbe9a14f18p-2},
    {0x1.3b13b20000000p+0, -0x1.32bff1d2620d3p-2},
    {0x1.3521d00000000p+0, -0x1.169c06a7938bbp-2},
    {0x1.2f684c0000000p+0, -0x1.f5fd8c01b8598p-3},
    {0x1.29e4120000000p+0, -0x1.bfc6745e58544p-3},
    {0x1.24924a0000000p+0, -0x1.8a898953f695dp-3},
    {0x1.1f70480000000p+0, -0x1.563dc4114f416p-3},
    {0x1.1a7b960000000p+0, -0x1.22dadb72090e4p-3},
    {0x1.15b1e60000000p+0, -0x1.e0b1af47da109p-4},
    {0x1.1111120000000p+0, -0x1.7d605d9f9a247p-4},
    {0x1.0c97140000000p+0, -0x1.1bb314bc1250dp-4},
    {0x1.0842100000000p+0, -0x1.773935884e226p-5},
    {0x1.0410420000000p+0, -0x1.743f41d467d22p-6},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.f07c200000000p-1, 0x1.6bad2043a8791p-5},
    {0x1.e1e1e20000000p-1, 0x1.663f6e3b3cbb2p-4},
    {0x1.d41d420000000p-1, 0x1.08c587b8a8459p-3},
    {0x1.c71c720000000p-1, 0x1.5c01a22e68f24p-3},
    {0x1.bacf920000000p-1, 0x1.acf5de2afc49ap-3},
    {0x1.af286c0000000p-1, 0x1.fbc16a1ed20a6p-3},
    {0x1.a41a420000000p-1, 0x1.2440796db68c3p-2},
    {0x1.99999a0000000p-1, 0x1.49a7834b7d429p-2},
    {0x1.8f9c180000000p-1, 0x1.6e22207523f6dp-2},
    {0x1.8618620000000p-1, 0x1.91bba6c447dcfp-2},
    {0x1.7d05f40000000p-1, 0x1.b47ebfcfdd47ap-2},
    {0x1.745d180000000p-1, 0x1.d6753b2085b50p-2},
};
constexpr double LOG2CR_C[8] = {  /* (-1)^(n+1) / (n ln 2), n = 1 .. 8 */
    0x1.71547652b82fep+0, -0x1.71547652b82fep-1, 0x1.ec709dc3a03fdp-2, -0x1.71547652b82fep-2, 0x1.2776c50ef9bfep-2, -0x1.ec709dc3a03fdp-3, 0x1.a61762a7aded9p-3, -0x1.71547652b82fep-3,
};

// The table in LDS (32 rows of {invc, logc}: LOG2CR_ROWS f32x4 behind the cube level table, filled by every wave for itself like that one) and the
// coefficients as s_mov_b32 immediates issued right where they are used.  A 64-bit constant needs a register pair on gfx950 (no 64-bit literals), and
// this function sits at the point of the highest register pressure of the declared-sampler kernels: as C++ literals (SGPR pairs the compiler hoists out
// of the march loop -- the kernels are at the SGPR ceiling) or as plain loads (hoisted or clustered) the constants took the raymarched-light kernel from
// 73 to 95-97 VGPRs (6 -> 4 waves per SIMD) and gave the others a stack frame; loaded one by one from global memory, each behind the Horner step before
// it, every kernel kept its registers but a frame bound by its limb waves (where lambda > 0 lives) paid nine dependent cache misses per call: clouds_high_rm
// 1920x1080 +49 % (profiles/round6/ab_lambda_exact.txt, first table).  An asm volatile is not hoisted, SALU issue is free beside VALU work, LDS answers in
// ~64 cycles.
constexpr int LOG2CR_ROWS = 32;
__device__ __forceinline__ void log2_cr_table_fill(f32x4 *tab, int lane) {   // tab: LOG2CR_ROWS f32x4 in LDS
    if (lane < LOG2CR_ROWS) tab[lane] = *reinterpret_cast<const f32x4 *>(&LOG2CR_TAB[lane][0]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}
template <uint64_t BITS>
__device__ __forceinline__ double log2_cr_const() {
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((uint32_t)BITS), "i"((uint32_t)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ float log2_cr(float x, const f32x4 *tab) {
    const uint32_t ix = __float_as_uint(x);
    const uint32_t tmp = ix - LOG2CR_OFF;
    const uint32_t i = (tmp >> 18) & 31u;
    const float z = __uint_as_float(ix - (tmp & 0xff800000u));
    const int k = (int)tmp >> 23;
    const f32x4 row = tab[i];
    const double invc = __hiloint2double(__float_as_int(row.y), __float_as_int(row.x)), logc = __hiloint2double(__float_as_int(row.w), __float_as_int(row.z));
    const double r = __builtin_fma((double)z, invc, -1.0);
    // LOG2CR_C[7] .. [0] as bit patterns (the asm takes integers); held to the generated table at compile time:
    static_assert(__builtin_bit_cast(uint64_t, LOG2CR_C[7]) == 0xbfc71547652b82feull && __builtin_bit_cast(uint64_t, LOG2CR_C[6]) == 0x3fca61762a7aded9ull &&
                  __builtin_bit_cast(uint64_t, LOG2CR_C[5]) == 0xbfcec709dc3a03fdull && __builtin_bit_cast(uint64_t, LOG2CR_C[4]) == 0x3fd2776c50ef9bfeull &&
                  __builtin_bit_cast(uint64_t, LOG2CR_C[3]) == 0xbfd71547652b82feull && __builtin_bit_cast(uint64_t, LOG2CR_C[2]) == 0x3fdec709dc3a03fdull &&
                  __builtin_bit_cast(uint64_t, LOG2CR_C[1]) == 0xbfe71547652b82feull && __builtin_bit_cast(uint64_t, LOG2CR_C[0]) == 0x3ff71547652b82feull,
                  "log2_cr: the immediates below are LOG2CR_C");
    // (each Horner step as ONE v_fma_f64 with the coefficient read straight from its SGPR pair: left to the compiler, p * r + c became v_fmac_f64 -- whose
    //  addend must sit in the destination VGPRs -- behind two v_mov_b32 per step, 14 VALU instructions of the block's 38)
    auto step = [&](double acc, double c) {
        double o;
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(acc), "v"(r), "s"(c));
        return o;
    };
    double p = step(log2_cr_const<0xbfc71547652b82feull>(), log2_cr_const<0x3fca61762a7aded9ull>());
    p = step(p, log2_cr_const<0xbfcec709dc3a03fdull>());
    p = step(p, log2_cr_const<0x3fd2776c50ef9bfeull>());
    p = step(p, log2_cr_const<0xbfd71547652b82feull>());
    p = step(p, log2_cr_const<0x3fdec709dc3a03fdull>());
    p = step(p, log2_cr_const<0xbfe71547652b82feull>());
    p = step(p, log2_cr_const<0x3ff71547652b82feull>());
    const double y0 = logc + (double)k;
    return (float)__builtin_fma(p, r, y0);
}

// (sc, tc, signed major) of v in the frame of the face selected by (isz, isy, pos): the linear maps of the Vulkan table
__device__ __forceinline__ void cube_frame(bool isz, bool isy, bool pos, V3 v, float &sc, float &tc, float &ma) {
    sc = isz ? (pos ? v.x : -v.x) : (isy ? v.x : (pos ? -v.z : v.z));
    tc = isz ? -v.y : (isy ? (pos ? v.z : -v.z) : -v.y);
    const float m = isz ? v.z : (isy ? v.y : v.x);
    ma = pos ? m : -m;
}

// texture(u_cloud_coverage_cubemap, d).r with the implicit LOD of a linear-mipmap sampler: finite differences inside the
// pixel quad, transformed to the selected face in the cancellation-free form s' - s = 0.5 (dsc ma - sc dma) / (ma ma'),
// lambda = 0.5 log2(max rho^2) clamped to the bound levels, linear mix of the two nearest levels.
__device__ __forceinline__ float cube_sample_lod(const RenderConsts &rc, V3 d, bool vx, V3 dx, bool vy, V3 dy, const f32x4 *lvl) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const bool isz = az >= ax && az >= ay, isy = !isz && ay >= ax;
    const float r = isz ? d.z : (isy ? d.y : d.x);
    const bool pos = r >= 0.0f;
    float sc, tc, ma;
    cube_frame(isz, isy, pos, d, sc, tc, ma);
    const int face = (isz ? 4 : (isy ? 2 : 0)) + (pos ? 0 : 1);
    const float s = 0.5f * (ieee_div(sc, ma) + 1.0f), t = 0.5f * (ieee_div(tc, ma) + 1.0f);
    float rho2 = 0.0f;
    const float n2 = (float)rc.cube_n * (float)rc.cube_n;
    auto axis = [&](bool valid, V3 q) {
        if (!valid) return;
        const V3 dv = {q.x - d.x, q.y - d.y, q.z - d.z};
        float dsc, dtc, dma;
        cube_frame(isz, isy, pos, dv, dsc, dtc, dma);
        const float ma2 = ma + dma;
        if (!(ma2 > 0.0f)) return;
    #if ATMO_LOD_DIV_FORM == 0
    const float inv = ieee_div(0.5f, ma * ma2);
#else
    const float inv = 0.5f * exact_rcp(ma * ma2);   // 0.5 / d = 0.5 RN(1 / d): the scaling is exact
#endif
        const float ds = (dsc * ma - sc * dma) * inv, dt = (dtc * ma - tc * dma) * inv;
        rho2 = fmaxf(rho2, (ds * ds + dt * dt) * n2);
    };
    axis(vx, dx);
    axis(vy, dy);
#if ATMO_LOD_LAMBDA_EXACT && ATMO_LOD_LOG2_CR
    float lambda = rho2 > 1.0f ? 0.5f * log2_cr(rho2, lvl + CUBE_LEVEL_TABLE) : 0.0f;   // correctly rounded, see cube_lod_select; rho2 <= 1: clamped to 0 below anyway
#else
    float lambda = rho2 > 0.0f ? 0.5f * __builtin_amdgcn_logf(rho2) : 0.0f;  // v_log_f32 = log2
#endif
    lambda = fminf(fmaxf(lambda, 0.0f), (float)(rc.cube_levels - 1));
    const float lf = floorf(lambda), fr = lambda - lf;
    const int lo = (int)lf, hi = lo + 1 < rc.cube_levels ? lo + 1 : lo;
    const float v0 = cube_level_sample(rc, face, s, t, lo);
    if (hi == lo || fr == 0.0f) return v0;
    const float v1 = cube_level_sample(rc, face, s, t, hi);
    return v0 * (1.0f - fr) + v1 * fr;
}

// ---- the same sampler on the fast path (rounds 3-4): power-of-two faces up to 1024 texels ----------------------------------------
// What the general form above spends and this one does not: five IEEE divisions per sample (here: one v_rcp + two Markstein
// steps shared by s and t -- the IEEE quotients, as in cube_sample<true> -- and a plain v_rcp per partner for the derivatives, whose
// relative error of 1e-7 moves lambda by 1e-7), face selection by compares (here: v_cubeid/sc/tc/ma), flat loads
// with integer address chains and a global load of the level offset (here: one buffer gather per level at a byte offset formed in
// fp32 from four per-level constants read from a 16-entry LDS table).  Bit-for-bit the same s, t, texels and filters as the general
// form; lambda agrees to a few ulp.
//
// Per-level constants (round 4; were ~12 VALU instructions per level and sample: ldexp, the closed-form level base, strides):
//   level l of a power-of-two chain: faces of nl = n >> l texels, (nl + 1)^2 footprints per face, packed behind the levels below it;
//   words before level l = 6 sum_{k<l} (n_k + 1)^2 = 8 n^2 - 8 nl^2 + 24 n - 24 nl + 6 l.
// Entry l = {hn = nl / 2, s4 = 4 nl + 4 (bytes per footprint row), F = s4 (nl + 1) (bytes per face), G = 4 words_before(l) + s4 + 4
// (byte offset of the footprint of texel (0, 0) on face 0)}; every entry and every partial sum below is a multiple of 4 below 2^26
// (33.6 MB for the whole chain at n = 1024), i.e. exact in fp32.  For the float copy of the chain (rc.cube_f4: 16-byte footprints)
// s4, F and G are stored times four -- multiples of 16 below 2^28, as exact.  Each wave fills the table itself before it reads it (identical
// values from both waves of a workgroup: no barrier).
__device__ __forceinline__ void cube_level_table_fill(const RenderConsts &rc, f32x4 *lvl, int lane) {
    if (lane < CUBE_LEVEL_TABLE) {
        const float nf = (float)rc.cube_n, l = (float)lane;
        const float nl = __builtin_amdgcn_ldexpf(nf, -lane);
        const float s4 = 4.0f * nl + 4.0f;
        const float c0 = __builtin_fmaf(32.0f * nf, nf, 96.0f * nf);
        const float base = __builtin_fmaf(-32.0f * nl, nl, __builtin_fmaf(-96.0f, nl, __builtin_fmaf(24.0f, l, c0)));
        const float u = rc.cube_f4 != nullptr ? 4.0f : 1.0f;  // the float copy of the chain: 16-byte footprints at four times the offsets
        lvl[lane] = f32x4{0.5f * nl, u * s4, u * (s4 * (nl + 1.0f)), u * (base + s4 + 4.0f)};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}
__device__ __forceinline__ float cube_level_sample_fast(__amdgpu_buffer_rsrc_t rs, bool f4, float fid, float q1s, float q1t, f32x4 k) {
    const float hn = k.x, s4 = k.y;
    const float x = __builtin_fmaf(q1s, hn, -0.5f), y = __builtin_fmaf(q1t, hn, -0.5f);  // ((q + 1) * 0.5) * nl - 0.5: the scalings are exact
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const float nm1 = __builtin_fmaf(hn, 2.0f, -1.0f);
    const float ic = __builtin_amdgcn_fmed3f(xf, -1.0f, nm1), jc = __builtin_amdgcn_fmed3f(yf, -1.0f, nm1);
    const float off = __builtin_fmaf(fid, k.z, __builtin_fmaf(jc, s4, __builtin_fmaf(ic, f4 ? 16.0f : 4.0f, k.w)));
    f32x4 t;
    if (f4) {  // `rs` is the chain's float copy: the table holds its strides (multiples of 16 below 2^28: exact in fp32)
        t = buf_f32x4(rs, (uint32_t)off);
    } else {
        const uint32_t w = buf_u32(rs, (uint32_t)off);
        t = f32x4{unorm8_exact(ub0(w)), unorm8_exact(ub1(w)), unorm8_exact(ub2(w)), unorm8_exact(ub3(w))};
    }
    return bilinear_exact4(t.x, t.y, t.z, t.w, fx, fy);
}

// The selected face's frame as coefficients (0, +1 or -1, so the products and sums below are exact): for any vector v
//   sc(v) = a1 v.x + a2 v.z,   tc(v) = b1 v.y + b2 v.z,   major(v) = mx v.x + my v.y + mz v.z,   ma(v) = sgn major(v)
// -- the linear maps of the Vulkan table (cube_frame) without per-use selects (a v_cndmask issues in 4 cycles, an FMA in 2).
struct CubeFaceFrame {
    float a1, a2, b1, b2, mx, my, mz, sgn;
};
__device__ __forceinline__ CubeFaceFrame cube_face_frame(bool isz, bool isy, bool pos) {
    CubeFaceFrame f;
    f.mz = isz ? 1.0f : 0.0f;
    f.my = isy ? 1.0f : 0.0f;
    f.mx = 1.0f - f.mz - f.my;
    f.sgn = pos ? 1.0f : -1.0f;
    f.a1 = f.mz * f.sgn + f.my;   // +Z: x, -Z: -x, +-Y: x
    f.a2 = -f.mx * f.sgn;         // +X: -z, -X: z
    f.b1 = -(f.mz + f.mx);        // -y on the X and Z faces
    f.b2 = f.my * f.sgn;          // +Y: z, -Y: -z
    return f;
}
// One quad partner's finite difference on the selected face (cube_lod_partner_coords below):
//   s' - s = 0.5 (dsc ma - sc dma) / (ma ma'),   ma' = ma + dma      (the cancellation-free form of the Vulkan derivative transformation);
//   rho^2 = ((s' - s)^2 + (t' - t)^2) n^2; a partner that does not reach the call, or lies beyond the face's half space (ma' <= 0), contributes 0.
// The difference q - d is rounded like the reference's; the face frame applied to it is exact (one non-zero term per sum, whether the
// compiler fuses it or not); everything behind it only moves lambda by ulps.
// The direction a set of cube coordinates came from (inverse of the Vulkan face table; sc, tc are signed copies of two components and
// v_cubema is twice the third, so this is exact).  Only for a partner whose direction lies on ANOTHER face than the lane's own.
__device__ __forceinline__ V3 cube_dir_from_coords(float fid, float sc, float tc, float mas) {
    const float m = 0.5f * mas;
    if (fid < 2.0f) return V3{m, -tc, fid < 1.0f ? -sc : sc};   // +X: sc = -z, tc = -y;  -X: sc = +z, tc = -y
    if (fid < 4.0f) return V3{sc, m, fid < 3.0f ? tc : -tc};    // +Y: sc = +x, tc = +z;  -Y: sc = +x, tc = -z
    return V3{fid < 5.0f ? sc : -sc, -tc, m};                   // +Z: sc = +x, tc = -y;  -Z: sc = -x, tc = -y
}
// max(rho2, this partner's rho^2), the partner given by its cube coordinates.  On the lane's own face -- all but the quads that straddle a cube edge --
// the differences of the coordinates ARE the face frame applied to the rounded difference of the directions, bit for bit: sc and tc are
// signed copies of one component each (so scp - sc = +-(q.c - d.c), the same rounding), and 0.5 |v_cubema| is the major component with
// the face's sign on both.
template <bool EXACT>
__device__ __forceinline__ float cube_lod_partner_coords(float fid, float sc, float tc, float ma2x, float ma, V3 d, bool valid, float fidp,
                                                         float scp, float tcp, float masp, float n2q, float rho2) {
    float dsc = scp - sc, dtc = tcp - tc, dma = 0.5f * fabsf(masp) - ma;
    if (valid && fidp != fid) {
        // (the face frame is built HERE, inside the branch only the quads across a cube edge take: computed in front of it, its dozen compares
        //  and selects ran for every sample)
        const bool isz = fid >= 4.0f, isy = !isz && fid >= 2.0f, pos = ma2x >= 0.0f;
        const CubeFaceFrame f = cube_face_frame(isz, isy, pos);
        const V3 q = cube_dir_from_coords(fidp, scp, tcp, masp);
        const V3 dv = {q.x - d.x, q.y - d.y, q.z - d.z};
        dsc = f.a1 * dv.x + f.a2 * dv.z;
        dtc = f.b1 * dv.y + f.b2 * dv.z;
        dma = f.sgn * (f.mx * dv.x + f.my * dv.y + f.mz * dv.z);
    }
    // rho^2 of this partner, in two forms of the same real function:
    //   EXACT   the operations of a scalar fp32 evaluation of the stated convention, one for one -- the IEEE quotient 0.5 / (ma ma'), every product and
    //           sum of the numerators and of rho^2 rounded on its own (the CPU checker under oracle/, sample_cube_lod) -- so that lambda is the oracle's
    //           BIT FOR BIT (with log2_cr).  One ulp of lambda moves a hypersensitive pixel (a `clouds` march of 8 long steps at a high density scale:
    //           x 135 (density ramp) x ~250 (optical thickness per unit density)) by 5e-4 -- profiles/round4/fuzz_sensitive_pixels.txt; seed 1040 of the
    //           1 212-scene fuzz sat at 2.0e-4 for three rounds.
    //           * 0.5 / d = 0.5 RN(1 / d), and a factor 0.5 commutes with every rounding behind it: ds = N RN(1 / d) is twice the oracle's, its square
    //             four times, and n2q = n^2 / 4 (exact) makes rho^2 the same bits -- the multiplication by 0.5 is never issued;
    //           * RN(1 / d) = exact_rcp: v_rcp_f32 and a Newton step, equal to the IEEE quotient on every significand (atmo_selftest_exact_math).
    //   !EXACT  rounds 2-5's arithmetic: fused numerators, the approximate reciprocal.  An ulp or two of lambda away -- since round 6 the light taps' form
    //           (cube_sample_lod_fast: "WHERE exactness is needed").
    const float ma2 = ma + dma;
    float r2;
    if (EXACT) {
        const float r = exact_rcp(ma * ma2);
        const float ds = (dsc * ma - sc * dma) * r, dt = (dtc * ma - tc * dma) * r;   // (no contraction in this function)
        r2 = (ds * ds + dt * dt) * n2q;
    } else {
        // (written with explicit FMAs: under `fp contract(fast)` hipcc chooses which product of a difference to fuse, and that choice is part of the bits)
        const float inv = hw_rcp(ma * ma2);
        const float ds = __builtin_fmaf(ma, dsc, -(sc * dma)) * inv, dt = __builtin_fmaf(ma, dtc, -(tc * dma)) * inv;
        r2 = __builtin_fmaf(dt, dt, ds * ds) * n2q;
    }
    return (valid && ma2 > 0.0f) ? fmaxf(rho2, r2) : rho2;
}

// lambda from rho^2 = max over the partners: the lower level and the weight of the one above it (0: the lower level alone)
struct CubeLod {
    int lo;
    float fr;
};
template <bool EXACT>
__device__ __forceinline__ CubeLod cube_lod_select(const RenderConsts &rc, float rho2, const f32x4 *lvl) {
    // clamp(0.5 log2(rho2), 0, levels - 1); rho2 <= 1 (or 0, or NaN) => lambda = 0
    float lambda;
    if (EXACT && ATMO_LOD_LOG2_CR) {
        // log2 correctly rounded and the oracle's bits (log2_cr); rho^2 <= 1: lambda = 0 (log2(1) = 0)
        float l2 = 0.0f;
        if (rho2 > 1.0f) l2 = log2_cr(rho2, lvl + CUBE_LEVEL_TABLE);
        lambda = fminf(0.5f * l2, (float)(rc.cube_levels - 1));
    } else {
        // v_log_f32 = log2 to 1 ulp; the lower clamp as a max in front of it (log2(1) = 0, and v_max returns the other operand for a NaN)
        lambda = fminf(0.5f * __builtin_amdgcn_logf(fmaxf(rho2, 1.0f)), (float)(rc.cube_levels - 1));
    }
#if defined(ATMO_WAVE_TRACE) && ATMO_RMQ_STATS  // diagnostic build: how many of these samples select level 0 alone (words 40..43 of the statistics block)
    if (rc.wave_trace != nullptr) {
        unsigned long long *st_ = rc.wave_trace + 16ull * gridDim.x * gridDim.y - 64 + 40;
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true), z_ = __builtin_amdgcn_ballot_w64(!(rho2 > 1.0f));
        if ((int)(threadIdx.x & 63) == __builtin_ffsll((long long)m_) - 1) {
            atomicAdd(st_, (unsigned long long)__builtin_popcountll(m_)); atomicAdd(st_ + 1, (unsigned long long)__builtin_popcountll(z_));
            atomicAdd(st_ + 2, 1ull); atomicAdd(st_ + 3, z_ == m_ ? 1ull : 0ull);
        }
    }
#endif
    const float lf = floorf(lambda);
    CubeLod l;
    l.lo = (int)lf;
    l.fr = lambda - lf;   // 0 on the last level: lambda <= levels - 1 is an integer there
    return l;
}
// level 0's tap alone: what a sample with the level-0 certificate returns ({0, 0} through cube_lod_finish, without its selects and tests)
__device__ __forceinline__ float cube_lod_level0(const RenderConsts &rc, float fid, float qs, float qt, const f32x4 *lvl) {
    const bool f4 = rc.cube_f4 != nullptr;
    const __amdgpu_buffer_rsrc_t rs = f4 ? make_rsrc(rc.cube_f4, rc.cube_bytes * 4u) : make_rsrc(rc.cube, rc.cube_bytes);
    return cube_level_sample_fast(rs, f4, fid, qs + 1.0f, qt + 1.0f, lvl[0]);
}
// the two nearest levels (cube_sample_lod_fast / _quad)
__device__ __forceinline__ float cube_lod_finish(const RenderConsts &rc, float fid, float qs, float qt, CubeLod l, const f32x4 *lvl) {
    const float q1s = qs + 1.0f, q1t = qt + 1.0f;
    const bool f4 = rc.cube_f4 != nullptr;  // wave-uniform: the float copy of the chain, or the byte footprints
    const __amdgpu_buffer_rsrc_t rs = f4 ? make_rsrc(rc.cube_f4, rc.cube_bytes * 4u) : make_rsrc(rc.cube, rc.cube_bytes);
    const float v0 = cube_level_sample_fast(rs, f4, fid, q1s, q1t, lvl[l.lo]);
    if (l.fr == 0.0f) return v0;
    const float v1 = cube_level_sample_fast(rs, f4, fid, q1s, q1t, lvl[l.lo + 1]);
    return v0 * (1.0f - l.fr) + v1 * l.fr;
}
// s = sc / ma, t = tc / ma: the IEEE quotients from one v_rcp and two Markstein steps each (as in cube_sample<true>)
__device__ __forceinline__ void cube_exact_quotients(float sc, float tc, float ma, float &qs, float &qt) {
    const float r = hw_rcp(ma);
    qs = sc * r; qt = tc * r;
    qs = __builtin_fmaf(__builtin_fmaf(-qs, ma, sc), r, qs);
    qt = __builtin_fmaf(__builtin_fmaf(-qt, ma, tc), r, qt);
    qs = __builtin_fmaf(__builtin_fmaf(-qs, ma, sc), r, qs);
    qt = __builtin_fmaf(__builtin_fmaf(-qt, ma, tc), r, qt);
}

// ---- the level-0 certificate (round 4) ---------------------------------------------------------------------------------------------
// In the scenes this path renders the coverage cubemap is MAGNIFIED almost everywhere (256^2 faces wrapped around a planet that fills a
// 1080p frame: 0.4 texels per pixel at the disc's centre; lambda = 0 for 97.2 % of the coverage samples of clouds_high(_rm) at 1920x1080,
// 99.8 % at 3840x2160, 100 % from the ground: profiles/round4/lod0_certificate.txt) -- and there the derivative machinery (two partners'
// coordinates, two rho^2, a logarithm) computes a zero.  A sample whose partners are provably within a texel takes level 0 without it.
//   With d = (dsc, dtc, dma) the partner's coordinate differences on the lane's face (cube_lod_partner_coords: a signed permutation of the
//   rounded difference of the two directions, on the lane's own face and across a cube edge alike), s = sc / ma, t = tc / ma, ma' = ma + dma:
//     s' - s = (dsc - s dma) / (2 ma'),   so   rho^2 = n^2 ((dsc - s dma)^2 + (dtc - t dma)^2) / (4 ma'^2)
//            <= n^2 (1 + s^2 + t^2) |d|^2 / (4 ma'^2)                 (triangle inequality, then Cauchy-Schwarz on |(dsc, dtc)| + |(s, t)| |dma|)
//   and ma' >= ma - |d|.  If  w E <= C ma^2  with w = 1 + s^2 + t^2, E >= |d|^2 and C = 0.97 * 4 (1 - 2/n)^2 / (n sigma)^2 (host: lod0_inv_c = 1 / C), then
//   |d| <= 2 ma / n, ma' >= (1 - 2/n) ma and rho^2 <= 0.97 for either partner: max(rho^2, 1) = 1, lambda = 0, frac = 0 -- the sample IS level 0's
//   bilinear tap, bit for bit what the full path returns.  E is taken between the UNROTATED positions (sigma bounds the rotation); the few ulp
//   the rotation's and the tap offsets' roundings add to |d|, and the 1e-6 relative error of the kernel's own rho^2, are inside the 0.97 (0.6 % of it).
// A partner that does not reach the call, or lies beyond the face's half space, contributes nothing to rho^2: leaving it out of E, or in, is
// safe.  NaN or infinite operands fail the comparison and take the full path (certificate withheld: 1 / C = +inf, 0 * inf = NaN).
// The lanes carry E / C + 1e-30 (cube_lod_scaled_spread: once per ray / per queued sample), so the test is w (E / C + 1e-30) <= ma^2: no uniform
// operand (the declared-sampler kernels spill SGPRs to VGPR lanes: every uniform read here was a v_readlane), and the floor fails it for ma^2 < 1e-30.
__device__ __forceinline__ float cube_lod_scaled_spread(const RenderConsts &rc, float e2) { return e2 * rc.lod0_inv_c + 1e-30f; }
__device__ __forceinline__ bool cube_lod_level0_certain(const RenderConsts &rc, float qs, float qt, float ma, float e2c) {
    const float w = __builtin_fmaf(qs, qs, __builtin_fmaf(qt, qt, 1.0f));
    const bool certain = w * e2c <= ma * ma;
#if defined(ATMO_WAVE_TRACE) && ATMO_RMQ_STATS  // diagnostic build: how many samples carry the certificate (words 44..47 of the statistics block)
    if (rc.wave_trace != nullptr) {
        unsigned long long *st_ = rc.wave_trace + 16ull * gridDim.x * gridDim.y - 64 + 44;
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true), z_ = __builtin_amdgcn_ballot_w64(certain);
        if ((int)(threadIdx.x & 63) == __builtin_ffsll((long long)m_) - 1) {
            atomicAdd(st_, (unsigned long long)__builtin_popcountll(m_)); atomicAdd(st_ + 1, (unsigned long long)__builtin_popcountll(z_));
            atomicAdd(st_ + 2, 1ull); atomicAdd(st_ + 3, z_ == m_ ? 1ull : 0ull);
        }
    }
#endif
    return certain;
}
template <int CTRL>
__device__ __forceinline__ float quad_read(float x) {  // v_mov_b32_dpp quad_perm: 0xB1 = [1,0,3,2] (lane ^ 1), 0x4E = [2,3,0,1] (lane ^ 2)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// E for a whole cloud march (lock-step form), once per ray where every marching lane is active: the two lanes' sample positions at step k
// are affine in k up to the drift of the rounded position chain (one rounded addition per step and component: <= k ulp(|p|) / 2 per lane), and
// the norm of an affine function is convex -- its maximum over the march is at the first or the last sample.
__device__ __forceinline__ float quad_march_spread2(const RenderConsts &rc, float px, float py, float pz, float ddx, float ddy, float ddz, bool vx, bool vy) {
    const float last = rc.lod0_last;
    const float lx = __builtin_fmaf(last, ddx, px), ly = __builtin_fmaf(last, ddy, py), lz = __builtin_fmaf(last, ddz, pz);
    auto norm2 = [](float a, float b, float c) { return __builtin_fmaf(a, a, __builtin_fmaf(b, b, c * c)); };
    const float ex = fmaxf(norm2(quad_read<0xB1>(px) - px, quad_read<0xB1>(py) - py, quad_read<0xB1>(pz) - pz),
                           norm2(quad_read<0xB1>(lx) - lx, quad_read<0xB1>(ly) - ly, quad_read<0xB1>(lz) - lz));
    const float ey = fmaxf(norm2(quad_read<0x4E>(px) - px, quad_read<0x4E>(py) - py, quad_read<0x4E>(pz) - pz),
                           norm2(quad_read<0x4E>(lx) - lx, quad_read<0x4E>(ly) - ly, quad_read<0x4E>(lz) - lz));
    const float e = hw_sqrt(fmaxf(vx ? ex : 0.0f, vy ? ey : 0.0f));                          // (v_sqrt_f32: 1 ulp, inside the 1.001 below)
    const float pmax = hw_sqrt(fmaxf(norm2(px, py, pz), norm2(lx, ly, lz))) + e;              // either lane, any step
    // drift: sqrt(3) components x (k/2 + k/2) ulp, ulp(|p|) <= 2^-23 |p|, k <= steps (one more for the rounding of `l`); 1.001: this function's own rounding
    const float r = (e + rc.lod0_drift * pmax) * 1.001f;
    return cube_lod_scaled_spread(rc, r * r);
}

// ---- lambda of a march sample: exact (round 6) -----------------------------------------------------------------------------------------------------------
// WHERE exactness is needed.  One ulp of lambda moves the filtered coverage by <= 2^-24 |level l+1 - level l|, the density ramp multiplies by 135, and a
// MARCH sample's density enters the pixel through exp(-density step_len u_cloud_density_scale) -- an optical thickness per unit density of ~250 in an
// 8-step march over a thick layer: 5e-4 of the pixel (profiles/round4/fuzz_sensitive_pixels.txt).  A light TAP's density enters through
// exp(-sum_i density_i step_i scale) with step_i = 0.15 thickness / 6 x 1.2^i -- 3 to 7.5 per unit density in the demo scene, 25 at the extreme of
// the fuzz's parameter range -- and then only as a factor of the sample's light: 1e-5 of the pixel at the very worst, unobservable against the 1e-4
// contract (the four scenes of the 1 212-scene soak that exceeded it were all march samples; none was a raymarched-light scene).  So the march's samples
// (cube_sample_lod_quad) take the oracle's operations bit for bit, and the light taps (cube_sample_lod_fast) keep rounds 2-5's arithmetic: evaluated
// exactly they cost clouds_high_rm +12.6 % at 1920x1080 and +11 % from the limb for nothing a test can see (profiles/round6/ab_lambda_exact.txt).
// WHAT the march pays for it: clouds_high +4 % at 1920x1080 pose P_space, +7.4 % from the limb, where lambda > 0 is the rule (1.4 / 1.8 % of that for
// the exact rho^2, the rest for log2_cr: ~30 instructions where v_log_f32 is one).  Tried and measured against it (same file): a fused-form screen
// (rho^2_fused <= 0.98 proves lambda = 0) in front of a from-scratch exact pass (+8 % / +27 %: the limb's waves take both), and the coverage from
// v_log_f32 as an ESTIMATE whose early-outs are tested at +- its error bound, the exact logarithm only for samples whose outcome depends on the value
// (+5 % / +38 %: along a grazing ray most samples with lambda > 0 do reach the shape fetch, and pay the level taps twice).

// position form (the light taps of a queued sample): the partners' sample positions and the tap offset are given.  Their cube coordinates come
// from the hardware cube instructions as well (4 instructions instead of the face frame applied to a difference: 13), and
// cube_lod_partner_coords takes it from there -- the same bits as the frame form on the lane's own face, the frame form itself elsewhere.
// Rounds 2-5's arithmetic (fused rho^2, v_log_f32): see "WHERE exactness is needed" above.
__device__ __forceinline__ float cube_sample_lod_fast(const RenderConsts &rc, V3 d, const QuadNb *nb) {
    const float fid = __builtin_amdgcn_cubeid(d.x, d.y, d.z);
    const float sc = __builtin_amdgcn_cubesc(d.x, d.y, d.z);
    const float tc = __builtin_amdgcn_cubetc(d.x, d.y, d.z);
    const float ma2x = __builtin_amdgcn_cubema(d.x, d.y, d.z);  // 2 * the signed major component
    const float ma = 0.5f * fabsf(ma2x);
    float qs, qt;
    cube_exact_quotients(sc, tc, ma, qs, qt);
    // (one path to the level taps here, the certain samples arriving with {0, 0}: a direct level-0 return as in the lock-step form below makes the
    //  rolled tap loop longer -- clouds_high_rm 1080p +1.2 %, P_limb +8 % against -2.5 % / +2.6 % with the direct return in the march alone)
    CubeLod lod = {0, 0.0f};
    if (!cube_lod_level0_certain(rc, qs, qt, ma, nb->e2)) {
        float rho2 = 0.0f;
        auto at = [&](V3 p) {  // the partner's tap position, rotated like the lane's own (cloud_funcs.gdshaderinc:43)
            const float x = p.x + nb->k.x, y = p.y + nb->k.y, z = p.z + nb->k.z;
            return V3{rc.cov_rot[0] * x + rc.cov_rot[2] * z, y, rc.cov_rot[1] * x + rc.cov_rot[3] * z};
        };
        const V3 dx = at(nb->px), dy = at(nb->py);
        const float nf = (float)rc.cube_n, n2q = 0.25f * (nf * nf);   // n^2 / 4: see cube_lod_partner_coords
        rho2 = cube_lod_partner_coords<false>(fid, sc, tc, ma2x, ma, d, nb->vx, __builtin_amdgcn_cubeid(dx.x, dx.y, dx.z), __builtin_amdgcn_cubesc(dx.x, dx.y, dx.z),
                                              __builtin_amdgcn_cubetc(dx.x, dx.y, dx.z), __builtin_amdgcn_cubema(dx.x, dx.y, dx.z), n2q, rho2);
        rho2 = cube_lod_partner_coords<false>(fid, sc, tc, ma2x, ma, d, nb->vy, __builtin_amdgcn_cubeid(dy.x, dy.y, dy.z), __builtin_amdgcn_cubesc(dy.x, dy.y, dy.z),
                                              __builtin_amdgcn_cubetc(dy.x, dy.y, dy.z), __builtin_amdgcn_cubema(dy.x, dy.y, dy.z), n2q, rho2);
        lod = cube_lod_select<false>(rc, rho2, nb->lvl);
    }
    return cube_lod_finish(rc, fid, qs, qt, lod, nb->lvl);
}
// lock-step form (the march): the partners' cube coordinates come out of the whole-quad exchange -- which only the lanes without a certificate
// enter (their quad mates, certain or gone through an early-out, are its helper lanes).  The oracle's operations, bit for bit (round 6).
__device__ __forceinline__ float cube_sample_lod_quad(const RenderConsts &rc, float px, float py, float pz, const QuadNb *nb) {
    const float qx = rc.cov_rot[0] * px + rc.cov_rot[2] * pz, qz = rc.cov_rot[1] * px + rc.cov_rot[3] * pz;
    const V3 d = {qx, py, qz};
    const float fid = __builtin_amdgcn_cubeid(qx, py, qz);
    const float sc = __builtin_amdgcn_cubesc(qx, py, qz);
    const float tc = __builtin_amdgcn_cubetc(qx, py, qz);
    const float ma2x = __builtin_amdgcn_cubema(qx, py, qz);
    const float ma = 0.5f * fabsf(ma2x);
    float qs, qt;
    cube_exact_quotients(sc, tc, ma, qs, qt);
    if (cube_lod_level0_certain(rc, qs, qt, ma, nb->e2)) return cube_lod_level0(rc, fid, qs, qt, nb->lvl);   // (never when !cube_lod_fast: 1 / C = inf)
    QuadRegs &q = *nb->regs;
    quad_exchange_coords(px, py, pz, rc.cov_rot[0], rc.cov_rot[1], rc.cov_rot[2], rc.cov_rot[3], q);  // q.fid, sc, tc, mas, qx, qz: the values above
    if (!rc.cube_lod_fast)  // faces that are not a power of two (or above 1024): the general sampler on the partners' directions (exact as it stands)
        return cube_sample_lod(rc, d, nb->vx, cube_dir_from_coords(q.fidx, q.scx, q.tcx, q.masx), nb->vy, cube_dir_from_coords(q.fidy, q.scy, q.tcy, q.masy), nb->lvl);
    const float nf = (float)rc.cube_n, n2q = 0.25f * (nf * nf);   // n^2 / 4: see cube_lod_partner_coords
    constexpr bool EXACT = ATMO_LOD_LAMBDA_EXACT != 0;
    float rho2 = cube_lod_partner_coords<EXACT>(fid, sc, tc, ma2x, ma, d, nb->vx, q.fidx, q.scx, q.tcx, q.masx, n2q, 0.0f);
    rho2 = cube_lod_partner_coords<EXACT>(fid, sc, tc, ma2x, ma, d, nb->vy, q.fidy, q.scy, q.tcy, q.masy, n2q, rho2);
    return cube_lod_finish(rc, fid, qs, qt, cube_lod_select<EXACT>(rc, rho2, nb->lvl), nb->lvl);
}

// The direct light march of one view sample (ATMO_LIGHT_DIRECT; SURVEY.md 8d "N view x M light steps"): the quantity the LUT
// tabulates -- get_optical_depth of optical_depth.gdshader:17-31 over the chord of :56-65 -- from the sample at squared radius
// r2 (position relative to the planet centre), bdot = dot(position, sun_dir), y3 = the sample's own (1 - height ratio)^3.
// One definition for march_atmosphere and for the probe kernel behind atmo_debug_marched_optical_depth.
struct LightMarchConsts {
    float ninv_h, c1, dens2, ratm2, inv_light_steps;
    int light_steps;
};
template <int LSTEPS>
__device__ __forceinline__ float sun_od_direct(const LightMarchConsts &k, float r2, float bdot, float y3) {
#pragma clang fp contract(fast)
    const float ninv_h = k.ninv_h, c1 = k.c1, dens2 = k.dens2, ratm2 = k.ratm2, inv_light_steps = k.inv_light_steps;
    const int light_steps = k.light_steps;
    // chord from the sample to the outer sphere along the sun direction, then a left Riemann sum.
    // x1 - max(x0, 0) with x0 = -b - sq, x1 = sq - b  ==  min(x1 - x0, x1) = min(2 sq, sq - b)
    const float hh = ratm2 - (r2 - bdot * bdot);
    const float sq = hw_sqrt(fmaxf(hh, 0.0f));
    // inside the outer sphere the forward exit distance is >= 0; hh < 0 (rounding at the shell) gives sq = 0 and
    // min(0, -b), which the max folds to the reference's 0
    const float ray_len = fmaxf(fminf(sq + sq, sq - bdot), 0.0f);
    const float lstep = ray_len * inv_light_steps;
    float acc = y3;  // sample 0 sits on the view sample itself
    if (LSTEPS == 8) {
        const float lb = lstep * (bdot + bdot), l2 = lstep * lstep;
        float q[8], rr[8];
#pragma unroll
        for (int j = 1; j < 8; ++j) q[j] = fmaf((float)(j * j), l2, fmaf((float)j, lb, r2));
        asm volatile("v_sqrt_f32 %0, %7\n\tv_sqrt_f32 %1, %8\n\tv_sqrt_f32 %2, %9\n\tv_sqrt_f32 %3, %10\n\t"
                     "v_sqrt_f32 %4, %11\n\tv_sqrt_f32 %5, %12\n\tv_sqrt_f32 %6, %13\n\ts_nop 0"
                     : "=&v"(rr[1]), "=&v"(rr[2]), "=&v"(rr[3]), "=&v"(rr[4]), "=&v"(rr[5]), "=&v"(rr[6]), "=&v"(rr[7])
                     : "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]));
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            const float yy = sat(fmaf(rr[j], ninv_h, c1));
            acc = fmaf(yy * yy, yy, acc);
        }
    } else if (LSTEPS > 0) {
        // |o + j*l*sun|^2 = r2 + j*(l*2b) + j^2*(l*l), |sun| = 1: two FMAs per sample
        const float lb = lstep * (bdot + bdot), l2 = lstep * lstep;
#pragma unroll
        for (int j = 1; j < (LSTEPS > 0 ? LSTEPS : 1); ++j) {
            const float rr = hw_sqrt(fmaf((float)(j * j), l2, fmaf((float)j, lb, r2)));
            const float yy = sat(fmaf(rr, ninv_h, c1));
            acc = fmaf(yy * yy, yy, acc);
        }
    } else {
        const float b2 = bdot + bdot;
        float sl = lstep;
        for (int j = 1; j < light_steps; ++j) {
            const float rr = hw_sqrt(fmaf(sl, sl + b2, r2));
            const float yy = sat(fmaf(rr, ninv_h, c1));
            acc = fmaf(yy * yy, yy, acc);
            sl += lstep;
        }
    }
    return acc * lstep * dens2;
}

// ---- compute_atmosphere_v2 -----------------------------------------------------------------------
// Returns RGBA.  Well-conditioned: fused arithmetic + hardware transcendentals throughout.
//   * alpha: the reference's recurrence alpha += (1-exp(-d))*(1-alpha) is 1 - prod(exp(-d_i))
//     = 1 - exp(-view_optical_depth); one exp after the loop replaces one per step.
//   * light: sum(d_i * T_i * coeff) = coeff * sum(d_i * T_i).
//   * positions are kept relative to the planet centre; 1 - clamp((r-R)/H, 0, 1) = clamp(fma(r, -1/H, 1 + R/H), 0, 1)
//     is one v_fma with the clamp modifier.
// LSTEPS > 0: light-step count known at compile time (fully unrolled); 0: rc.light_steps at run time.
// SPLIT = 2: lane `half` integrates its half of the view steps ([0, n0) / [n0, steps)) with the optical depth counted
// from its own first sample; exp(-(V_A + v) k) = exp(-V_A k) exp(-v k) folds the first half's total V_A into the
// second half's sum afterwards (same real function, one extra exp per channel per ray).
// VIEWPOS (KF_VIEW_POS; contexts with more than 32 view steps): the sample position is accumulated the way the reference does it
// (atmosphere_funcs_v2.gdshaderinc:57-58,81) -- in VIEW space from ray_origin + ray_dir * t_begin, one rounded addition of the rounded
// ray_dir * step_len per step, the planet centre subtracted at every use -- instead of one running position relative to the centre.  In real
// arithmetic the same points; in fp32 the two running sums drift apart by a few ulp of the position per step, and on a thin atmosphere
// (H / R ~ 0.05) 64 steps of that reached 1.08e-4 of alpha (2 of 252 fuzz scenes, round 3).  Three more subtractions per step, 4-7 % on
// the direct-light loop, so the kernels with up to 32 steps keep the cheaper form (and their ISA): this is a separate instantiation.
__device__ __forceinline__ V3 scale_unfused(V3 d, float s) { return V3{d.x * s, d.y * s, d.z * s}; }  // no contraction here: products rounded on their own
template <bool DIRECT, int LSTEPS, int SPLIT, bool VIEWPOS = false>
__device__ __forceinline__ float4 march_atmosphere(const RenderConsts &rc, V3 dir, float t_begin, float step_len, float jitter, int half) {
#pragma clang fp contract(fast)
    const int n0 = SPLIT == 2 ? (rc.view_steps + 1) / 2 : rc.view_steps;
    const int first = (SPLIT == 2 && half) ? n0 : 0;
    const int steps = (SPLIT == 2 && half) ? rc.view_steps - n0 : n0;
    t_begin = SPLIT == 2 ? fmaf((float)first, step_len, t_begin) : t_begin;
    const float inv_h = hw_rcp(rc.atmosphere_height);
    const float ninv_h = -inv_h;
    const float c1 = fmaf(rc.planet_radius, inv_h, 1.0f);
    const float dens2 = rc.density * rc.density;
    const float kr = -rc.coeff[0] * LOG2E, kg = -rc.coeff[1] * LOG2E, kb = -rc.coeff[2] * LOG2E;
    const float sx = rc.sun_dir[0], sy = rc.sun_dir[1], sz = rc.sun_dir[2];
    const float ratm2 = rc.atmosphere_radius * rc.atmosphere_radius;
    const int light_steps = LSTEPS > 0 ? LSTEPS : rc.light_steps;
    const float inv_light_steps = LSTEPS > 0 ? 1.0f / (float)(LSTEPS > 0 ? LSTEPS : 1) : hw_rcp((float)light_steps);
    const LightMarchConsts lmc = {ninv_h, c1, dens2, ratm2, inv_light_steps, light_steps};
    const float half_w = 0.5f * (float)rc.lut_w, x_off = half_w - 0.5f;
    const float lut_hf = (float)rc.lut_h, y_off = lut_hf - 0.5f;

    static_assert(!VIEWPOS || SPLIT == 1, "reference-form position accumulation: one lane per ray");
    const V3 p0 = scale_unfused(dir, t_begin), sd = scale_unfused(dir, step_len);  // ray_origin (0) + ray_dir * t_begin; ray_dir * step_len
    float pvx = p0.x, pvy = p0.y, pvz = p0.z;                                      // VIEWPOS: the view-space position
    float ox = fmaf(dir.x, t_begin, -rc.center[0]);
    float oy = fmaf(dir.y, t_begin, -rc.center[1]);
    float oz = fmaf(dir.z, t_begin, -rc.center[2]);
    const float sdx = VIEWPOS ? sd.x : dir.x * step_len, sdy = VIEWPOS ? sd.y : dir.y * step_len, sdz = VIEWPOS ? sd.z : dir.z * step_len;
    const float dstep = dens2 * step_len;
    float lr = 0.0f, lg = 0.0f, lb = 0.0f, view_od = 0.0f;

    // WHERE THIS LOOP SITS IN THE INSTRUCTION STREAM IS PART OF THE HEADLINE KERNEL'S SPEED (round 6, profiles/round6/ab_loop_phase.txt).  In <4, 8, 1> the loop
    // is 436 bytes, and the draw takes 0.0865 ms when its first instruction -- the target of the backward branch -- lies 12 bytes into a 32-byte block, 0.094-0.096 ms
    // (+8.5 ... +11 %) at each of the other seven 4-byte positions; shifted by 32 bytes it is fast again.  Instructions inserted INSIDE the loop behind its first one
    // cost their issue slot and nothing else, a 2 x unrolled body has no fast position at all, the LUT kernels' and the cloud kernels' loops do not care.  It is what
    // made rounds 2-5's "any scalar instruction in the preamble costs 8-10 %" and "an 80-SGPR cap (8 waves per SIMD) loses 8 %" -- both moved this loop by 4 bytes
    // (at the right position the capped build is exactly as fast as this one: the eighth wave buys nothing).  tests/test_host_logic.py holds the position
    // (tools/loop_phase.py reads it from the built library); if a change to the code in front of the loop moves it, ATMO_LOOP_PAD = the number of s_nop that puts it back.
#ifndef ATMO_LOOP_PAD
#define ATMO_LOOP_PAD 0   // (the geometric tile order's first form in the preamble needed 1; with its hint tables the loop is back on 12 mod 32 by itself)
#endif
#if ATMO_LOOP_PAD > 0
#define ATMO_STR2(x) #x
#define ATMO_STR(x) ATMO_STR2(x)
    if (DIRECT && LSTEPS == 8 && SPLIT == 1 && !VIEWPOS) asm volatile(".rept " ATMO_STR(ATMO_LOOP_PAD) "\n\ts_nop 0\n\t.endr");
#endif
    for (int i = 0; i < steps; ++i) {
        if (VIEWPOS) {
            ox = pvx - rc.center[0]; oy = pvy - rc.center[1]; oz = pvz - rc.center[2];
        }
        const float r2 = ox * ox + oy * oy + oz * oz;
        const float bdot = ox * sx + oy * sy + oz * sz;
        // LUT mode needs 1/r for the cosine; the direct light march only needs r
        const float inv_r = DIRECT ? 0.0f : hw_rsq(r2);
        const float r = DIRECT ? hw_sqrt(r2) : r2 * inv_r;
        const float y = sat(fmaf(r, ninv_h, c1));  // 1 - height_ratio
        const float y3 = y * y * y;

        float sun_od;
        if (DIRECT) {
            sun_od = sun_od_direct<LSTEPS>(lmc, r2, bdot, y3);
        } else {
            // uv = (0.5 + 0.5*cos, height_ratio) -> texel space
            const float x = fmaf(bdot * inv_r, half_w, x_off);
            const float yv = fmaf(-y, lut_hf, y_off);
            sun_od = lut_sample_fp(rc.lut4, rc.lut_w + 1, rc.lut_h + 1, x, yv);
        }

        const float d = y3 * dstep;
        view_od += d;
        const float od = sun_od + view_od;
        {
            const float ar = od * kr, ag = od * kg, ab = od * kb;
            float er, eg, eb;
            asm volatile("v_exp_f32 %0, %3\n\tv_exp_f32 %1, %4\n\tv_exp_f32 %2, %5\n\ts_nop 0" : "=&v"(er), "=&v"(eg), "=&v"(eb) : "v"(ar), "v"(ag), "v"(ab));
            lr = fmaf(d, er, lr);
            lg = fmaf(d, eg, lg);
            lb = fmaf(d, eb, lb);
        }

        if (VIEWPOS) {
            pvx += sdx; pvy += sdy; pvz += sdz;
        } else {
            ox += sdx; oy += sdy; oz += sdz;
        }
    }

    if (SPLIT == 2) {
        // lane 0 of the pair: total = A + exp2(V_A k) * B  (its own sums are A, the partner's are B); lane 1's result is unused
        const float od_b = swap_adjacent(view_od), lr_b = swap_adjacent(lr), lg_b = swap_adjacent(lg), lb_b = swap_adjacent(lb);
        lr = fmaf(lr_b, hw_exp2(view_od * kr), lr);
        lg = fmaf(lg_b, hw_exp2(view_od * kg), lg);
        lb = fmaf(lb_b, hw_exp2(view_od * kb), lb);
        view_od += od_b;
    }
    const float alpha = 1.0f - hw_exp2(-view_od * LOG2E);
    float4 o;
    o.x = sat(fmaf(lr, rc.coeff[0], rc.ambient[0])) * rc.modulate[0];
    o.y = sat(fmaf(lg, rc.coeff[1], rc.ambient[1])) * rc.modulate[1];
    o.z = sat(fmaf(lb, rc.coeff[2], rc.ambient[2])) * rc.modulate[2];
    o.w = clampf(fmaf(jitter, 0.02f, alpha), 0.0f, 0.99f);
    return o;
}

// ---- compute_atmosphere, v1 "lite" (shaders/include/atmosphere_funcs_v1.gdshaderinc:15-63) ----------------
// Faked 4-colour model: no LUT, no exp.  factor = prod(1 - density*step), light = mean(clamp(1.2*cos + 0.5)^2).
template <int SPLIT>
__device__ __forceinline__ float4 march_atmosphere_v1(const RenderConsts &rc, V3 dir, float t_begin, float t_end, int half) {
#pragma clang fp contract(fast)
    const float inv_steps = 1.0f / (float)rc.view_steps;
    const float step_len = (t_end - t_begin) * inv_steps;
    const int n0 = SPLIT == 2 ? (rc.view_steps + 1) / 2 : rc.view_steps;
    const int steps = (SPLIT == 2 && half) ? rc.view_steps - n0 : n0;
    t_begin = (SPLIT == 2 && half) ? fmaf((float)n0, step_len, t_begin) : t_begin;
    const float inv_h = hw_rcp(rc.atmosphere_height);
    const float ninv_h = -inv_h;
    const float c1 = fmaf(rc.planet_radius, inv_h, 1.0f);
    const float sx = rc.sun_dir[0], sy = rc.sun_dir[1], sz = rc.sun_dir[2];
    float ox = fmaf(dir.x, t_begin, -rc.center[0]);
    float oy = fmaf(dir.y, t_begin, -rc.center[1]);
    float oz = fmaf(dir.z, t_begin, -rc.center[2]);
    const float sdx = dir.x * step_len, sdy = dir.y * step_len, sdz = dir.z * step_len;
    const float nds = -rc.density * step_len;
    float factor = 1.0f, light_sum = 0.0f;
    for (int i = 0; i < steps; ++i) {
        const float r2 = ox * ox + oy * oy + oz * oz;
        const float inv_r = hw_rsq(r2);
        const float y = sat(fmaf(r2 * inv_r, ninv_h, c1));
        const float cosang = (ox * sx + oy * sy + oz * sz) * inv_r;
        const float l = sat(fmaf(1.2f, cosang, 0.5f));
        light_sum = fmaf(l, l, light_sum);
        factor *= fmaf(y * y * y, nds, 1.0f);
        ox += sdx; oy += sdy; oz += sdz;
    }
    if (SPLIT == 2) {  // product and sum over both halves
        factor *= swap_adjacent(factor);
        light_sum += swap_adjacent(light_sum);
    }
    const float light_factor = light_sum * inv_steps;
    const float atmo_factor = 1.0f - factor;
    const float day_factor = sat(light_factor * rc.day_night_transition_scale);
    float4 o;
    o.x = mixf(mixf(rc.night0[0], rc.night1[0], atmo_factor), mixf(rc.day0[0], rc.day1[0], atmo_factor), day_factor);
    o.y = mixf(mixf(rc.night0[1], rc.night1[1], atmo_factor), mixf(rc.day0[1], rc.day1[1], atmo_factor), day_factor);
    o.z = mixf(mixf(rc.night0[2], rc.night1[2], atmo_factor), mixf(rc.day0[2], rc.day1[2], atmo_factor), day_factor);
    o.w = sat(atmo_factor);
    return o;
}

// The same in the reference's operation order (atmo_set_precision(ctx, 1), the default of the v1 variants): unfused, IEEE
// sqrt and divisions, the position accumulated step by step.  The fast form above is within 1e-6 of it while
// density * step_len < 1, i.e. inside the model's range; outside (a thick, dense atmosphere marched in 16 steps) the
// product of (1 - density * step_len) grows like 1e5 and amplifies the fast form's rsq / fused rounding to 2e-4 relative
// (found by the executed-reference fuzz, tests/test_reference_exec.py seed 10).  ~3x the instructions of the fast form,
// still the cheapest kernels of the set.  SPLIT = 2: both lanes of a ray evaluate the whole march (no sharing).
// ---- compute_atmosphere_v2 in the REFERENCE'S operation order (atmo_set_precision 1 on the no-cloud variants) -----------------------
// The fast form above keeps one running position relative to the planet centre, sums the view optical depth and takes alpha from it in
// closed form; the reference accumulates the VIEW-SPACE position, subtracts the centre at every use, and builds alpha step by step
// (atmosphere_funcs_v2.gdshaderinc:60-82).  In fp32 the two drift apart with the number of view steps (1e-4 of alpha after 64 steps on a
// thin atmosphere: tests/checks/fuzz_debug.py).  This form follows the reference statement by statement -- unfused, IEEE sqrt / divide, expf,
// the LUT's bilinear filter as mix(mix(), mix()) on clamped texels, the direct light march as get_optical_depth writes it -- at about
// two to five times the time; it exists so that parity can be had to 1e-6 where it matters, not for the benchmarks (KF_ATMO_REF).
__device__ __forceinline__ float precise_density(const RenderConsts &rc, float dist) {  // get_atmosphere_density, atmosphere_common:12-24
    const float sd = dist - rc.planet_radius;
    const float h = fminf(fmaxf(ieee_div(sd, rc.atmosphere_height), 0.0f), 1.0f);
    const float y = 1.0f - h;
    return y * y * y * rc.density;
}
__device__ __forceinline__ float precise_lut(const RenderConsts &rc, float u, float v) {  // texture(u_optical_depth_texture, uv).r, clamp, linear
    const int w = rc.lut_w, h = rc.lut_h, st = w + 2;
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const int i0 = min(max((int)xf, 0), w - 1), i1 = min(max((int)xf + 1, 0), w - 1);
    const int j0 = min(max((int)yf, 0), h - 1), j1 = min(max((int)yf + 1, 0), h - 1);
    const float *t = rc.lut + st + 1;  // texel (0, 0) inside the apron
    const float t00 = t[j0 * st + i0], t10 = t[j0 * st + i1], t01 = t[j1 * st + i0], t11 = t[j1 * st + i1];
    return mixf(mixf(t00, t10, fx), mixf(t01, t11, fx), fy);
}
template <bool DIRECT>
__device__ __forceinline__ float4 march_atmosphere_v2_precise(const RenderConsts &rc, V3 dir, float t_begin, float t_end, float jitter) {
    const int steps = rc.view_steps;
    const V3 c = {rc.center[0], rc.center[1], rc.center[2]}, sun = {rc.sun_dir[0], rc.sun_dir[1], rc.sun_dir[2]};
    const float step_len = ieee_div(t_end - t_begin, (float)steps);
    float lr = 0.0f, lg = 0.0f, lb = 0.0f, view_od = 0.0f, alpha = 0.0f;
    V3 pos = {0.0f + dir.x * t_begin, 0.0f + dir.y * t_begin, 0.0f + dir.z * t_begin};  // ray_origin (0) + ray_dir * t_begin
    for (int i = 0; i < steps; ++i) {
        const V3 oc = {pos.x - c.x, pos.y - c.y, pos.z - c.z};
        const float dist = ieee_sqrt(oc.x * oc.x + oc.y * oc.y + oc.z * oc.z);
        float sun_od;
        if (DIRECT) {
            // get_optical_depth over the chord to the outer sphere (optical_depth.gdshader:17-31,56-65), light_steps left-Riemann samples
            const float b = oc.x * sun.x + oc.y * sun.y + oc.z * sun.z;
            const V3 qc = {oc.x - sun.x * b, oc.y - sun.y * b, oc.z - sun.z * b};
            float hh = rc.atmosphere_radius * rc.atmosphere_radius - (qc.x * qc.x + qc.y * qc.y + qc.z * qc.z);
            float x0 = 1000000.0f, x1 = 1000000.0f;
            if (!(hh < 0.0f)) {
                hh = ieee_sqrt(hh);
                x0 = -b - hh;
                x1 = -b + hh;
            }
            const float ray_len = x1 - fmaxf(x0, 0.0f);
            const float lstep = ieee_div(ray_len, (float)rc.light_steps);
            const V3 ls = {sun.x * lstep, sun.y * lstep, sun.z * lstep};
            sun_od = 0.0f;
            for (int j = 0; j < rc.light_steps; ++j) {
                const float fj = (float)j;
                const V3 p = {pos.x + ls.x * fj, pos.y + ls.y * fj, pos.z + ls.z * fj};
                const V3 d = {p.x - c.x, p.y - c.y, p.z - c.z};
                const float density = precise_density(rc, ieee_sqrt(d.x * d.x + d.y * d.y + d.z * d.z));
                sun_od += density * lstep * rc.density;
            }
        } else {
            // get_baked_optical_depth (v2:14-29)
            const float height = dist - rc.planet_radius;
            const float height_ratio = fminf(fmaxf(ieee_div(height, rc.atmosphere_height), 0.0f), 1.0f);
            const float inv = ieee_div(1.0f, dist);
            const V3 up = {oc.x * inv, oc.y * inv, oc.z * inv};
            const float uvx = 0.5f + 0.5f * (up.x * sun.x + up.y * sun.y + up.z * sun.z);
            sun_od = precise_lut(rc, uvx, height_ratio);
        }
        const float local_density = precise_density(rc, dist) * rc.density;
        view_od += local_density * step_len;
        const float od = sun_od + view_od;
        const float tr = expf(-od * rc.coeff[0]), tg = expf(-od * rc.coeff[1]), tb = expf(-od * rc.coeff[2]);
        lr += local_density * step_len * tr * rc.coeff[0];
        lg += local_density * step_len * tg * rc.coeff[1];
        lb += local_density * step_len * tb * rc.coeff[2];
        const float vt = expf(-local_density * step_len);
        alpha += (1.0f - vt) * (1.0f - alpha);
        pos = V3{pos.x + dir.x * step_len, pos.y + dir.y * step_len, pos.z + dir.z * step_len};
    }
    float4 o;
    o.x = fminf(fmaxf(lr + rc.ambient[0], 0.0f), 1.0f) * rc.modulate[0];
    o.y = fminf(fmaxf(lg + rc.ambient[1], 0.0f), 1.0f) * rc.modulate[1];
    o.z = fminf(fmaxf(lb + rc.ambient[2], 0.0f), 1.0f) * rc.modulate[2];
    o.w = fminf(fmaxf(alpha + jitter * 0.02f, 0.0f), 0.99f);
    return o;
}

__device__ __forceinline__ float4 march_atmosphere_v1_precise(const RenderConsts &rc, V3 dir, float t_begin, float t_end) {
#pragma clang fp contract(off)
    const float inv_steps = ieee_div(1.0f, (float)rc.view_steps);
    const float step_len = (t_end - t_begin) * inv_steps;
    const float svx = dir.x * step_len, svy = dir.y * step_len, svz = dir.z * step_len;
    // ray_origin + ray_dir * t_begin with ray_origin = 0
    float px = 0.0f + dir.x * t_begin, py = 0.0f + dir.y * t_begin, pz = 0.0f + dir.z * t_begin;
    float factor = 1.0f, light_sum = 0.0f;
    for (int i = 0; i < rc.view_steps; ++i) {
        const float rx = px - rc.center[0], ry = py - rc.center[1], rz = pz - rc.center[2];
        const float d = exact_sqrt(rx * rx + ry * ry + rz * rz);
        const float ux = ieee_div(rx, d), uy = ieee_div(ry, d), uz = ieee_div(rz, d);
        // get_atmosphere_density(d), atmosphere_common.gdshaderinc:12-24
        const float sd = d - rc.planet_radius;
        const float h = fminf(fmaxf(ieee_div(sd, rc.atmosphere_height), 0.0f), 1.0f);
        const float y = 1.0f - h;
        const float density = y * y * y * rc.density;
        float light = fminf(fmaxf(1.2f * (rc.sun_dir[0] * ux + rc.sun_dir[1] * uy + rc.sun_dir[2] * uz) + 0.5f, 0.0f), 1.0f);
        light = light * light;
        light_sum += light * inv_steps;
        factor *= (1.0f - density * step_len);
        px += svx; py += svy; pz += svz;
    }
    const float atmo_factor = 1.0f - factor;
    const float day_factor = fminf(fmaxf(light_sum * rc.day_night_transition_scale, 0.0f), 1.0f);
    float4 o;
    o.x = mixf(mixf(rc.night0[0], rc.night1[0], atmo_factor), mixf(rc.day0[0], rc.day1[0], atmo_factor), day_factor);
    o.y = mixf(mixf(rc.night0[1], rc.night1[1], atmo_factor), mixf(rc.day0[1], rc.day1[1], atmo_factor), day_factor);
    o.z = mixf(mixf(rc.night0[2], rc.night1[2], atmo_factor), mixf(rc.day0[2], rc.day1[2], atmo_factor), day_factor);
    o.w = fminf(fmaxf(atmo_factor, 0.0f), 1.0f);
    return o;
}

// ---- clouds ----------------------------------------------------------------------------------------

// Precise mode (atmo_set_precision(ctx, 1)): the whole density expression in the reference's operation order, unfused,
// with exact UNORM8 conversions and filters -- the density ramp (x50) then sees the same X as a scalar fp32 evaluation,
// bit for bit, and the cloud variants' error drops to the atmosphere's (profiles/round1/ab_clouds_exact.txt); -15 % speed.
#if defined(ATMO_WAVE_TRACE) && ATMO_RMQ_STATS
// diagnostic build: lanes x calls and wave x calls that reach a stage of the density evaluation (words 16.. of the statistics block;
// DENS_STAT_PHASE is 0 in the march, 1 in the light taps)
#define DENS_STAT(stage_) do { \
        const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); \
        if (rc.wave_trace != nullptr && (int)(threadIdx.x & 63) == __builtin_ffsll((long long)m_) - 1) { \
            unsigned long long *st_ = rc.wave_trace + 16ull * gridDim.x * gridDim.y - 64 + 16 + 8 * stat_phase + 2 * (stage_); \
            atomicAdd(st_, (unsigned long long)__builtin_popcountll(m_)); atomicAdd(st_ + 1, 1ull); } } while (0)
#define DENS_STAT_ARG , int stat_phase = 0
#define DENS_STAT_PASS(ph_) , ph_
#else
#define DENS_STAT(stage_) do { } while (0)
#define DENS_STAT_ARG
#define DENS_STAT_PASS(ph_)
#endif
// The density ramp clamp(x * 50 - 20, 0, 1) of get_density (clouds:60-75), unfused fp32: RN(RN(x * 50) - 20) <= 0  <=>  RN(x * 50) <= 20
// <=>  x <= 0x3ecccccd (the largest float whose product with 50 rounds to at most 20), and RN(RN(x * 50) - 20) >= 1  <=>  RN(x * 50) >= 21
// (the subtraction is exact there)  <=>  x >= 0x3ed70a3d.  Both equivalences checked for every float in [0.125, 1) and samples of the rest
// (profiles/round3/ab_density_ramp.txt); a NaN fails both tests in either form.
#define DENSITY_RAMP_ZERO __uint_as_float(0x3ecccccdu)
#define DENSITY_RAMP_ONE __uint_as_float(0x3ed70a3du)
// QUAD (with LOD): called in lock-step by the lanes of a wave (the march): the partners' coordinates come from the whole-quad exchange;
// otherwise from the partner positions in *nb (light taps).
template <bool EARLY_OUT, bool LOD = false, bool QUAD = false>
__device__ __forceinline__ float cloud_density_precise(const RenderConsts &rc, float px, float py, float pz, float hr, const QuadNb *nb = nullptr DENS_STAT_ARG) {
    DENS_STAT(0);
    const float t = 2.0f * hr - 1.0f;
    const float hc = fmaxf(1.0f - t * t, 0.0f);
    if (EARLY_OUT && !(hc > 0.0f)) return 0.0f;
    DENS_STAT(1);
    if (EARLY_OUT && LOD) {
        // Before the (expensive) implicit-LOD coverage sample: near the top and the bottom of the layer the height curve alone
        // decides.  A filtered UNORM8 texel (any level, any mix of two levels) is at most 1 + 2^-20, and every fp32 step from the
        // texel to the density is monotone non-decreasing in it (cloud_density_precise below), so if the expression is <= 0 at
        // that bound and at the largest shape value the exact result is 0.  ~10 % of the samples inside the layer.
        const float cov_hi = (1.0f + 9.5367431640625e-07f) - 0.25f * hr + rc.coverage_bias;
        const float m_hi = -1.2f * (1.0f - cov_hi) + 1.5f * cov_hi;
        if ((rc.shape_hi01 + m_hi) * hc <= DENSITY_RAMP_ZERO) return 0.0f;
    }
    float coverage = 1.0f;
    if (LOD && QUAD) {
        if (rc.cube != nullptr) coverage = cube_sample_lod_quad(rc, px, py, pz, nb);
    } else if (rc.cube != nullptr) {
        const float qx = rc.cov_rot[0] * px + rc.cov_rot[2] * pz;
        const float qz = rc.cov_rot[1] * px + rc.cov_rot[3] * pz;
        if (LOD) {
            auto at = [&](V3 p) {
                const float x = p.x + nb->k.x, y = p.y + nb->k.y, z = p.z + nb->k.z;
                return V3{rc.cov_rot[0] * x + rc.cov_rot[2] * z, y, rc.cov_rot[1] * x + rc.cov_rot[3] * z};
            };
            if (rc.cube_lod_fast) coverage = cube_sample_lod_fast(rc, V3{qx, py, qz}, nb);
            else coverage = cube_sample_lod(rc, V3{qx, py, qz}, nb->vx, at(nb->px), nb->vy, at(nb->py), nb->lvl);
        } else {
            coverage = cube_sample<true>(rc.cube, rc.cube_n, qx, py, qz, rc.cube_f4);
        }
    }
    coverage = coverage - 0.25f * hr + rc.coverage_bias;
    const float m = -1.2f * (1.0f - coverage) + 1.5f * coverage;
    if (EARLY_OUT) {
        // Coverage decides most samples before the shape texture is touched.  Every step of
        //   density(shape) = clamp((((shape - 0.1) + m) * hc) * 50 - 20, 0, 1)      (hc > 0 here)
        // is a monotone non-decreasing fp32 operation of `shape`, and shape lies in [shape_lo, shape_hi] (host: the
        // mix/invert of a filtered UNORM8 value in [0, 1 + 2^-20]).  So if the expression is <= 0 at shape_hi the exact
        // result is 0, and if it is >= 1 at shape_lo the exact result is 1: bit-identical, no trilinear fetch (the
        // 8-texel exact filter is ~45 % of a density evaluation).
        // (x * 50 - 20 <= 0 and >= 1 decided on x itself: DENSITY_RAMP_ZERO / _ONE, two instructions fewer per test)
        if ((rc.shape_hi01 + m) * hc <= DENSITY_RAMP_ZERO) return 0.0f;
        if ((rc.shape_lo01 + m) * hc >= DENSITY_RAMP_ONE) return 1.0f;
    }
    DENS_STAT(2);
    const float s = rc.shape_scale;
    const float tex = shape_sample<true>(rc.shape, rc.shape_n, rc.shape_log2n, px * s, py * s, pz * s, rc.shape_f4);
    float shape = 0.5f * (1.0f - rc.shape_factor) + tex * rc.shape_factor;
    if (rc.shape_invert) shape = 1.0f - shape;
    float density = (shape - 0.1f + m) * hc;
    density = density * 50.0f - 20.0f;
    return fminf(fmaxf(density, 0.0f), 1.0f);
}

// Fast mode (default): fused arithmetic after the exact height chain.
template <bool EARLY_OUT>
__device__ __forceinline__ float cloud_density_fast(const RenderConsts &rc, float px, float py, float pz, float hr) {
#pragma clang fp contract(fast)
    const float t = 2.0f * hr - 1.0f;
    const float hc = fmaxf(1.0f - t * t, 0.0f);
    if (EARLY_OUT && !(hc > 0.0f)) return 0.0f;  // outside the layer: (..)*0*50-20 clamps to 0, skip the fetches
    float coverage = 1.0f;
    if (rc.cube != nullptr) {
        const float qx = rc.cov_rot[0] * px + rc.cov_rot[2] * pz;
        const float qz = rc.cov_rot[1] * px + rc.cov_rot[3] * pz;
        coverage = cube_sample<false>(rc.cube, rc.cube_n, qx, py, qz);
    }
    coverage = coverage - 0.25f * hr + rc.coverage_bias;
    const float m = mixf(-1.2f, 1.5f, coverage);
    if (EARLY_OUT) {  // coverage-first early outs, see cloud_density_precise (here within the fast mode's tolerance)
        if (((rc.shape_hi01 + m) * hc) * 50.0f - 20.0f <= 0.0f) return 0.0f;
        if (((rc.shape_lo01 + m) * hc) * 50.0f - 20.0f >= 1.0f) return 1.0f;
    }
    const float s = rc.shape_scale;
    float shape = mixf(0.5f, shape_sample<false>(rc.shape, rc.shape_n, rc.shape_log2n, px * s, py * s, pz * s), rc.shape_factor);
    if (rc.shape_invert) shape = 1.0f - shape;
    const float density = (shape - 0.1f + m) * hc;
    return sat(density * 50.0f - 20.0f);
}

// get_density_full with CLOUDS_ALWAYS_LOW_QUALITY (detail = 0.5).  `hr` = height ratio from the exact chain.
// EARLY_OUT = false evaluates the fetches unconditionally (result is the same: hc = 0 forces the clamp to 0), which
// removes the divergent branch so that several independent taps can be interleaved by the scheduler.
template <bool EARLY_OUT, bool PRECISE, bool LOD = false, bool QUAD = false>
__device__ __forceinline__ float cloud_density(const RenderConsts &rc, float px, float py, float pz, float hr, const QuadNb *nb = nullptr DENS_STAT_ARG) {
    return PRECISE ? cloud_density_precise<EARLY_OUT, LOD, QUAD>(rc, px, py, pz, hr, nb DENS_STAT_PASS(stat_phase)) : cloud_density_fast<EARLY_OUT>(rc, px, py, pz, hr);
}

// exact |p| and (|p| - bottom) / thickness, as a scalar fp32 evaluation would produce them
__device__ __forceinline__ void cloud_height_r2(const RenderConsts &rc, float r2, float &r, float &hr) {
    r = exact_sqrt_pos(r2);
    hr = exact_div_uniform(r - rc.clouds_bottom, rc.cloud_thickness, rc.inv_cloud_thickness);
}
__device__ __forceinline__ void cloud_height(const RenderConsts &rc, float px, float py, float pz, float &r, float &hr) {
    cloud_height_r2(rc, px * px + py * py + pz * pz, r, hr);
}
// A march step at which NO lane of the wave can be inside the cloud layer skips the exact height chain and the density evaluation (28 % of the
// steps of a 1920x1080 frame at pose P_space: the stretch of a ray between the bottom shell and the ground).  |p|^2 below rc.layer_r2_lo means
// r < bottom, hr < 0, above rc.layer_r2_hi r > top, hr >= 1 (margins in fill_consts): hc = 0 and the density is 0 in the exact chain as well.
__device__ __forceinline__ bool wave_may_be_in_layer(const RenderConsts &rc, float r2) {
    return __builtin_amdgcn_ballot_w64(r2 >= rc.layer_r2_lo && r2 <= rc.layer_r2_hi) != 0ull;
}

// get_light_raymarched (cloud_funcs.gdshaderinc:104-151): 6 density taps towards the sun.
// 1 - prod(exp(-d_i)) = 1 - exp(-sum d_i): one exp instead of six.
// Tap 0 is the sample itself: pos0 + float(0) * step_len * dir = pos0 (clouds:129 with i = 0), and get_density there is the value
// raymarch_cloud computes for the same position one line later (clouds:217) -- the same function of the same arguments (both
// alpha0 branches are, with CLOUDS_ALWAYS_LOW_QUALITY).  The caller passes that density in as d0: five taps
// are evaluated instead of six, and the one saved is the expensive one -- a lit sample has density > 0, so its tap 0 never takes
// an early-out and always pays the full exact shape + coverage filters.  Bit-identical.
template <bool PRECISE, bool LOD = false>
__device__ __forceinline__ float light_raymarched(const RenderConsts &rc, float px, float py, float pz, float hr0, float d0,
                                                  float sx, float sy, float sz, const QuadNb *nb = nullptr) {
    float sum = __builtin_fmaf(d0, rc.rm_weight[0], 0.0f);
    auto tap_i = [&](int i) {
        // exact: pos0 + (i*step)*dir, unfused; the product (float(i) * step_len_i) * dir is uniform and comes rounded from the host
        // (two SGPR factors would cost a move and a multiply per component and tap)
        const float kx = rc.rm_tap[i][0], ky = rc.rm_tap[i][1], kz = rc.rm_tap[i][2];
        const float qx = px + kx, qy = py + ky, qz = pz + kz;
        float r, hr;
        cloud_height(rc, qx, qy, qz, r, hr);
        QuadNb tap;
        if (LOD) {  // the quad partners evaluate the same tap from their own sample position
            tap.vx = nb->vx; tap.vy = nb->vy; tap.lvl = nb->lvl; tap.regs = nullptr;
            tap.px = nb->px; tap.py = nb->py; tap.k = V3{kx, ky, kz}; tap.e2 = nb->e2;
        }
        const float d = cloud_density<true, PRECISE, LOD>(rc, qx, qy, qz, hr, LOD ? &tap : nullptr DENS_STAT_PASS(1));
        sum = __builtin_fmaf(d, rc.rm_weight[i], sum);  // step_len_i * density_scale  [host]
    };
    if (LOD) {  // the LOD sampler inlined five times doubles the kernel's code (8 300 -> 4 200 lines of ISA): rolled, the draw is 6 % faster
        // (instruction cache); the LOD-0 kernel, 3 800 lines either way, is 1-2 % faster unrolled (profiles/round3/ab_lod_taps_rolled.txt)
#pragma unroll 1
        for (int i = 1; i < 6; ++i) tap_i(i);
    } else {
#pragma unroll
        for (int i = 1; i < 6; ++i) tap_i(i);
    }
    const float alpha = 1.0f - hw_exp2(-sum * LOG2E);
    return mixf(1.0f, hr0 * 0.2f, alpha);
}

// Where a pixel's cloud march starts and how it advances (model space): the first lines of raymarch_cloud
// (clouds:186-213), exact arithmetic.  Used for the pixel itself and, in the LOD mode, for its two quad partners.
struct MarchRay {
    bool valid;
    float px, py, pz, ddx, ddy, ddz, step_len;
};
__device__ __forceinline__ MarchRay cloud_march_ray(const RenderConsts &rc, V3 dir_m, float t_begin, float t_end, float jitter) {
    MarchRay m;
    t_end = t_begin + fminf(t_end - t_begin, rc.max_d);
    m.step_len = (t_end - t_begin) * rc.inv_cloud_steps;
    const float js = jitter * m.step_len;
    m.px = (rc.origin_model[0] + dir_m.x * js) + dir_m.x * t_begin;
    m.py = (rc.origin_model[1] + dir_m.y * js) + dir_m.y * t_begin;
    m.pz = (rc.origin_model[2] + dir_m.z * js) + dir_m.z * t_begin;
    m.ddx = dir_m.x * m.step_len; m.ddy = dir_m.y * m.step_len; m.ddz = dir_m.z * m.step_len;
    m.valid = true;
    return m;
}

// The uniforms the density evaluation reads at every sample, in VGPRs (round 4, late).  The declared-sampler kernels run at the SGPR ceiling and
// spill uniforms to VGPR lanes: every use in the march loop was a v_readlane (18 static in <49, 0, 1>, 0 with this), on top of the slow-class issue
// of a VALU instruction with an SGPR operand.  `clouds_high` 0.1907 -> 0.1745 ms, `clouds` 0.1141 -> 0.1057, P_limb -9 %, v1 -3.4 %;
// the level-0 kernels, which spill nothing, are unchanged (+-0.5 %): profiles/round4/ab_vgpr_uniforms.txt.  NOT a general rule: the sun direction
// of the hand-scheduled atmosphere march in VGPRs costs the headline kernel 10 %.
// A copy of the constants whose hot fields pass through an empty asm with "+v" operands; everything else of the copy stays what it was.
#ifndef ATMO_VGPR_UNIFORMS
#define ATMO_VGPR_UNIFORMS 1
#endif
__device__ __forceinline__ void cloud_uniforms_to_vgprs(RenderConsts &v) {
#if ATMO_VGPR_UNIFORMS
    asm volatile("" : "+v"(v.cov_rot[0]), "+v"(v.cov_rot[1]), "+v"(v.cov_rot[2]), "+v"(v.cov_rot[3]), "+v"(v.clouds_bottom), "+v"(v.cloud_thickness),
                 "+v"(v.inv_cloud_thickness), "+v"(v.coverage_bias));
#endif
}

// get_planet_shadow (clouds:153-167) of a lit sample as a light factor: smoothstep(-0.3, 0.3, dot(normalize(pos), -sun_dir)) mixed into
// [0.002, 1].  ONE definition for the lit-sample queue and the in-place forms: the sum of three products leaves the compiler a choice of
// what to fuse, and that choice is part of the bits.
__device__ __forceinline__ float cloud_shadow_light(float px, float py, float pz, float sx, float sy, float sz, float r) {
#pragma clang fp contract(fast)
    const float sd = -(px * sx + py * sy + pz * sz) * hw_rcp(r);
    const float st = sat((sd + 0.3f) * (1.0f / 0.6f));
    const float shadow = st * st * (3.0f - 2.0f * st);
    return fmaf(shadow, 0.002f - 1.0f, 1.0f);
}
// One lit sample of the raymarched-light recurrence as the lit-sample queue evaluates it (march_clouds_rm_queue, phases A and C): the
// transmittance chain, the sample's weight w = shadow light * density * step * total transmittance, and -- given the sample's raymarched
// light -- total_light += light * w with the product and the sum rounded on their own (they pass through LDS there).
struct RmRecurrence {
    float total_transmittance, one_minus_alpha, total_light;
};
__device__ __forceinline__ float rm_sample_weight(RmRecurrence &s, float density, float lb, float scale_step, float neg_scale_step_log2e) {
#pragma clang fp contract(fast)
    const float transmittance = hw_exp2(density * neg_scale_step_log2e);
    s.total_transmittance = fmaxf(s.total_transmittance * transmittance, 0.005f);
    s.one_minus_alpha *= transmittance;
    return (lb * (density * scale_step)) * s.total_transmittance;
}

// raymarch_cloud (cloud_funcs.gdshaderinc:175-247).  Returns (total_light, alpha).
// SPLIT = 2: lane `half` of a pair evaluates the samples with step index = half (mod 2) -- position chain, density,
// light -- while the recurrence over the samples (transmittance floor, light sum, alpha) runs in step order on the
// pair's values exchanged by DPP.  Every sample is evaluated with the same arithmetic as in the one-lane form (the
// position is still advanced one rounded addition per step), so the result is bit-identical; only lane 0's is used.
template <bool RM, bool PRECISE, int SPLIT, bool LOD = false>
__device__ __forceinline__ float2 march_clouds(const RenderConsts &rc_in, V3 dir_m, float t_begin, float t_end, float jitter, int half,
                                               QuadRegs *qregs = nullptr, const f32x4 *lvl = nullptr) {
    // Under the declared sampler the raymarched light is evaluated IN PLACE with one lane per ray as well (round 5) -- this function instead of the
    // lit-sample queue, whose arithmetic it keeps (same bits): 73 VGPRs and no queue in LDS instead of 86 / 15 KB, six waves per SIMD instead of five, no
    // twelve-word entries through LDS, no second whole-quad block per lit sample.  Measured then (profiles/round5/ab_rm_inplace.txt): 3840x2160 -11.4 %
    // (BASELINE configs[3]), from the ground -10.4 %, 1920x1080 pose P_space +1.6 %; on round 6's kernels the queue form loses everywhere, by 2-22 %, also
    // when pushed to six waves (profiles/round6/ab_rm_tile_choice.txt, 3) -- its declared-sampler branches, kept through round 5 as the A/B arm
    // ATMO_RM_INPLACE=0, are gone (the last commit that holds them: 9e43032).  NOT for the level-0 sampler: its queue kernel (six-word entries, 10 KB, taps
    // unrolled) is 11-30 % faster than its in-place form.
    // LOD && SPLIT == 2 (round 5, the heavy tiles of a frame): the two lanes of a ray are lane and lane ^ 4, a pixel quad keeps four consecutive
    // lanes (all at the same step); with RM the light taps are evaluated in place -- the partners' sample positions come from the quad
    // mates by the same whole-quad block the queue uses at enqueue -- and the recurrence runs in the lit-sample queue's arithmetic
    // (rm_sample_weight, product and sum of the light term rounded on their own), so a ray's result is the queue kernel's, bit for bit.
    constexpr bool RMQ_FORM = RM && LOD;
    RenderConsts rc_v;   // (not for RMQ_FORM: a private copy whose rm_tap[] is indexed in the rolled tap loop becomes a 1.2 KB stack frame)
    if constexpr (PRECISE && !RMQ_FORM) {
        rc_v = rc_in;
        cloud_uniforms_to_vgprs(rc_v);
    }
    const RenderConsts &rc = (PRECISE && !RMQ_FORM) ? rc_v : rc_in;
    const int steps = rc.cloud_steps;
    // exact: positions
    const MarchRay self = cloud_march_ray(rc, dir_m, t_begin, t_end, jitter);
    const float step_len = self.step_len;
    float px = self.px, py = self.py, pz = self.pz;
    const float ddx = self.ddx, ddy = self.ddy, ddz = self.ddz;
    float sx = rc.sun_dir_model[0], sy = rc.sun_dir_model[1], sz = rc.sun_dir_model[2];
    if (SPLIT == 2 && half) {  // lane 1 starts on sample 1
        px = px + ddx; py = py + ddy; pz = pz + ddz;
    }
    QuadNb nb;
    if (LOD) {  // a quad partner "reaches the call" iff it marches: the lanes that are active here (2x2 quads = 4 consecutive lanes)
        const unsigned long long marching = __builtin_amdgcn_ballot_w64(true);
        const int lane = threadIdx.x & 63;
        nb.vx = (marching >> (lane ^ 1)) & 1ull; nb.vy = (marching >> (lane ^ 2)) & 1ull;
        nb.lvl = lvl; nb.regs = qregs;
        // (SPLIT == 2: the second lane of a ray bounds the spread over samples 1 .. steps, one beyond its last -- a superset of what it
        //  evaluates, and the distance of two affine motions is convex: still an upper bound)
        nb.e2 = quad_march_spread2(rc, px, py, pz, ddx, ddy, ddz, nb.vx, nb.vy);
        if (RMQ_FORM) asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));   // as in the queue form: the sun direction in VGPRs, no stack frame
    }

    // pow(dot(ray_dir, sun_dir), 16) is constant along the ray; dp <= 0 => 0
    const float dp = dir_m.x * sx + dir_m.y * sy + dir_m.z * sz;
    float p16 = 0.0f;
    if (dp > 0.0f) {
        const float p2 = dp * dp, p4 = p2 * p2, p8 = p4 * p4;
        p16 = p8 * p8;
    }

    float total_transmittance = 1.0f, total_light = 0.0f, one_minus_alpha = 1.0f;
    RmRecurrence rec = {1.0f, 1.0f, 0.0f};   // RMQ_FORM
    const float scale_step = rc.cloud_density_scale * step_len;
    const float neg_scale_step_log2e = -scale_step * LOG2E;

    // one sample of the recurrence (clouds:216-236); la = raymarched light or the height ratio, lb = planet-shadow factor
    auto integrate = [&](float density, float la, float lb) {
        if (RMQ_FORM) {  // the lit-sample queue's arithmetic
            if (density > 0.0f) {
                const float w = rm_sample_weight(rec, density, lb, scale_step, neg_scale_step_log2e);
                rec.total_light = add_unfused(rec.total_light, mul_unfused(la, w));
            }
            return;
        }
        // A zero-density sample contributes nothing: transmittance 1, light term 0.
        if (density > 0.0f) {
#pragma clang fp contract(fast)
            float light = RM ? la : fmaf(p16, one_minus_alpha, la);
            light *= lb;
            const float transmittance = hw_exp2(density * neg_scale_step_log2e);
            total_transmittance = fmaxf(total_transmittance * transmittance, 0.005f);
            total_light = fmaf(light * (density * scale_step), total_transmittance, total_light);
            one_minus_alpha *= transmittance;
        }
    };

    const int iters = (steps + SPLIT - 1) / SPLIT;
    for (int it = 0; it < iters; ++it) {
        float density = 0.0f, la = 0.0f, lb = 0.0f;
        if (SPLIT == 1 || it * SPLIT + half < steps) {
            float r = 0.0f, hr = 0.0f;
            const float r2 = px * px + py * py + pz * pz;
            if (wave_may_be_in_layer(rc, r2)) {
                cloud_height_r2(rc, r2, r, hr);
                density = cloud_density<true, PRECISE, LOD, LOD>(rc, px, py, pz, hr, LOD ? &nb : nullptr);
            }
            // the light value of a zero-density sample (6 more density taps in the raymarched variant) is never observed
            if (RMQ_FORM) {
                if (density > 0.0f) {
                    lb = cloud_shadow_light(px, py, pz, sx, sy, sz, r);
                    // the light taps of this sample difference the partners' tap positions: their sample positions, from the quad mates (which
                    // march in lock-step but may be unlit, i.e. disabled here: whole-quad mode), as the queue form fetches them at enqueue
                    quad_exchange_positions(px, py, pz, *qregs);
                    QuadNb enb;
                    enb.vx = nb.vx; enb.vy = nb.vy; enb.lvl = lvl; enb.regs = nullptr;
                    enb.px = V3{qregs->fidx, qregs->scx, qregs->tcx};
                    enb.py = V3{qregs->fidy, qregs->scy, qregs->tcy};
                    enb.k = V3{0.0f, 0.0f, 0.0f};
                    auto dist2 = [&](V3 p) { const float a = p.x - px, b = p.y - py, c = p.z - pz; return __builtin_fmaf(a, a, __builtin_fmaf(b, b, c * c)); };
                    enb.e2 = cube_lod_scaled_spread(rc, fmaxf(enb.vx ? dist2(enb.px) : 0.0f, enb.vy ? dist2(enb.py) : 0.0f) * 1.001f);
                    la = light_raymarched<PRECISE, true>(rc, px, py, pz, hr, density, sx, sy, sz, &enb);
                }
            } else if (density > 0.0f) {
#pragma clang fp contract(fast)
                la = RM ? light_raymarched<PRECISE, false>(rc, px, py, pz, hr, density, sx, sy, sz, nullptr) : hr;
                // get_planet_shadow: smoothstep(-0.3, 0.3, dot(normalize(pos), -sun_dir))
                const float sd = -(px * sx + py * sy + pz * sz) * hw_rcp(r);
                const float st = sat((sd + 0.3f) * (1.0f / 0.6f));
                const float shadow = st * st * (3.0f - 2.0f * st);
                lb = fmaf(shadow, 0.002f - 1.0f, 1.0f);
            }
        }
        // exact: pos += ray_dir * step_len, once per step of the ray
#pragma unroll
        for (int k = 0; k < SPLIT; ++k) {
            px = px + ddx; py = py + ddy; pz = pz + ddz;
        }
        if (SPLIT == 1) {
            integrate(density, la, lb);
        } else {
            // step order for lane 0 of the pair: its own sample, then the partner's (lane 1's order is irrelevant)
            const float d1 = swap_ray_lanes<LOD>(density), a1 = swap_ray_lanes<LOD>(la), b1 = swap_ray_lanes<LOD>(lb);
            integrate(density, la, lb);
            integrate(d1, a1, b1);
        }
    }
    if (RMQ_FORM) return make_float2(rec.total_light, 1.0f - rec.one_minus_alpha);
    return make_float2(total_light, 1.0f - one_minus_alpha);
}

// ---- raymarch_cloud with raymarched light, lit samples regrouped through LDS -------------------------------
// (The kernels of the LEVEL-0 sampler and of the fast cloud mode: under the declared sampler the raymarched light is evaluated in place, march_clouds<RM, .., LOD>.)
// In clouds_high_rm only the samples with density > 0 need get_light_raymarched (6 more density evaluations each), and in a
// lock-step march they are a changing subset of the wave: 21 % of the issued lanes idle (VALUUtilization 78.7 %,
// profiles/round2/pmc_clouds_high_rm_1920x1080.json).  The light value does not feed back into the march -- it only scales
// the sample's contribution to total_light -- so a wave can queue its lit samples and evaluate them later, a full wave at a time:
//   phase A, every step: position chain, density, transmittance recurrence; a lit lane stores w = shadow * density * step *
//     total_transmittance in its slot[step][lane] and appends (position, height ratio, slot) to the wave's ring queue in LDS
//     (ballot + mbcnt give the lane its place; the fill count lives in an SGPR);
//   phase B, whenever the queue holds one entry per marching lane (and at the end of a 16-step chunk): lane r takes entry r,
//     evaluates the 6-tap light for THAT sample and multiplies it into the slot;
//   phase C, end of the chunk: every lane adds its slots in step order.
// Per ray the arithmetic is fixed (own slots, step order), so the picture does not depend on which rays share a wave.
// LDS per wave: 6 x 128 queue words + 8 x 64 slots = 5 KB (10 KB per 2-wave workgroup).
// With the implicit cubemap LOD an entry also carries the sample positions of the two quad partners (the light taps of a queued
// sample difference THEIR tap positions: 6 more words) and, in bits 10-11 of the slot word, whether each partner marches.
// Steps per chunk of the lit-sample queue = rows of the per-lane slot array in LDS.  8 (round 3; was 16): 5 KB instead of 7 KB per wave,
// i.e. LDS no longer caps the raymarched-light kernel at 5.5 waves per SIMD -- together with __launch_bounds__(128, 6) (80 instead of
// 84 VGPRs, one spilled dword) it runs 6 waves per SIMD: -6.9 % at 1920x1080, -5.2 % at 3840x2160 (profiles/round3/ab_occupancy.txt;
// either change alone: -1.5 % / +-0.4 %).
#ifndef ATMO_RMQ_CHUNK
#define ATMO_RMQ_CHUNK 8
#endif
// (Round 3, measured and dropped: summing a half-chunk one half later so that no partial batch is lit at the chunk ends fills the
// batches to 97 % instead of 93 % -- tools/rmq_stats.py -- and is 0.7-2 % slower: profiles/round3/rmq_stage_stats.txt.)
constexpr int RMQ_CAP = 128;
constexpr int RMQ_CHUNK = ATMO_RMQ_CHUNK;
constexpr int RMQ_WORDS_PER_WAVE = 6 * RMQ_CAP + RMQ_CHUNK * 64;

template <bool PRECISE>
__device__ __forceinline__ float2 march_clouds_rm_queue(const RenderConsts &rc, V3 dir_m, float t_begin, float t_end, float jitter,
                                                        float *__restrict__ lds) {
    // (no cloud_uniforms_to_vgprs here: eight more VGPRs take the declared-sampler kernel from 88 to 103 -- 4 waves per SIMD, +9 % -- and under a
    //  96-VGPR bound it spills for a gain of 1 %; three or four of them (93-95 VGPRs, one short of the step) bought 0.5-1.5 %; the level-0 kernel,
    //  bound to 80 VGPRs, is 0.5-1.3 % slower with them: profiles/round4/ab_vgpr_uniforms.txt)
    float *qx = lds, *qy = lds + RMQ_CAP, *qz = lds + 2 * RMQ_CAP, *qh = lds + 3 * RMQ_CAP;
    uint32_t *qs = reinterpret_cast<uint32_t *>(lds + 4 * RMQ_CAP);
    float *qd = lds + 5 * RMQ_CAP;  // the sample's own density = light tap 0
    float *slot = lds + 6 * RMQ_CAP;
    const int lane = threadIdx.x & 63;
    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);  // the lanes of this wave that march
    auto rank_in = [&](unsigned long long m) {
        return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    };
    const int rank = rank_in(active);
    const int batch = __builtin_popcountll(active);
    int qhead = 0, qcount = 0;  // wave-uniform (SGPRs): entries [qhead, qcount) are waiting, indices modulo RMQ_CAP
#if defined(ATMO_WAVE_TRACE) && ATMO_RMQ_STATS  // diagnostic build: how full the light batches run (tools/rmq_stats.py)
    unsigned long long st_calls = 0, st_avail = 0, st_ticks_b = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define RMQ_STAT_BATCH(avail_) do { ++st_calls; st_avail += (unsigned long long)(avail_); } while (0)
#define RMQ_STAT_B_BEGIN() const unsigned long long st_b0 = __builtin_amdgcn_s_memtime()
#define RMQ_STAT_B_END() st_ticks_b += __builtin_amdgcn_s_memtime() - st_b0
#else
#define RMQ_STAT_BATCH(avail_) do { } while (0)
#define RMQ_STAT_B_BEGIN() do { } while (0)
#define RMQ_STAT_B_END() do { } while (0)
#endif

    const int steps = rc.cloud_steps;
    const MarchRay self = cloud_march_ray(rc, dir_m, t_begin, t_end, jitter);
    const float step_len = self.step_len;
    float px = self.px, py = self.py, pz = self.pz;
    const float ddx = self.ddx, ddy = self.ddy, ddz = self.ddz;
    float sx = rc.sun_dir_model[0], sy = rc.sun_dir_model[1], sz = rc.sun_dir_model[2];
    RmRecurrence rec = {1.0f, 1.0f, 0.0f};
    const float scale_step = rc.cloud_density_scale * step_len;
    const float neg_scale_step_log2e = -scale_step * LOG2E;
    auto light_batch = [&](int avail) {  // phase B: lane `rank` lights queue entry qhead + rank
        RMQ_STAT_BATCH(avail);
        RMQ_STAT_B_BEGIN();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (rank < avail) {
            const int e = (qhead + rank) & (RMQ_CAP - 1);
            const float ex = qx[e], ey = qy[e], ez = qz[e], eh = qh[e], ed = qd[e];
            const uint32_t sl = qs[e];
            const float light = light_raymarched<PRECISE, false>(rc, ex, ey, ez, eh, ed, sx, sy, sz, nullptr);
            slot[sl] = light * slot[sl];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        RMQ_STAT_B_END();
    };

    for (int c0 = 0; c0 < steps; c0 += RMQ_CHUNK) {
        const int cn = steps - c0 < RMQ_CHUNK ? steps - c0 : RMQ_CHUNK;
        uint32_t lit_bits = 0;
        for (int k = 0; k < cn; ++k) {
            float r = 0.0f, hr = 0.0f, density = 0.0f;
            const float r2 = px * px + py * py + pz * pz;
            if (wave_may_be_in_layer(rc, r2)) {
                cloud_height_r2(rc, r2, r, hr);
                density = cloud_density<true, PRECISE>(rc, px, py, pz, hr);
            }
            const bool lit = density > 0.0f;
            float w = 0.0f;
            if (lit) w = rm_sample_weight(rec, density, cloud_shadow_light(px, py, pz, sx, sy, sz, r), scale_step, neg_scale_step_log2e);
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(lit);
            if (lit) {
                const int e = (qcount + rank_in(lm)) & (RMQ_CAP - 1);
                const uint32_t sl = (uint32_t)(k * 64 + lane);
                qx[e] = px; qy[e] = py; qz[e] = pz; qh[e] = hr; qd[e] = density;
                qs[e] = sl;
                slot[sl] = w;
                lit_bits |= 1u << k;
            }
            qcount += __builtin_popcountll(lm);
            // exact: pos += ray_dir * step_len
            px = px + ddx; py = py + ddy; pz = pz + ddz;
            // a full batch is waiting -- or the chunk ends and its slots are read next: drain (one partial batch at most)
            const bool last = k == cn - 1;
            while (qcount - qhead >= batch || (last && qcount > qhead)) {  // single call site: one copy of the 6-tap block
                const int avail = qcount - qhead < batch ? qcount - qhead : batch;
                light_batch(avail);
                qhead += avail;
            }
        }
        for (int k = 0; k < cn; ++k)  // phase C, step order
            if ((lit_bits >> k) & 1u) rec.total_light += slot[k * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#if defined(ATMO_WAVE_TRACE) && ATMO_RMQ_STATS
    if (rc.wave_trace != nullptr && rank == 0) {  // the last 64 words of the trace buffer (its upper half is unused: 2 waves per tile)
        unsigned long long *st = rc.wave_trace + 16ull * gridDim.x * gridDim.y - 64;
        atomicAdd(st + 0, 1ull);                                   // waves that march
        atomicAdd(st + 1, (unsigned long long)batch);              // marching lanes
        atomicAdd(st + 2, st_calls);                               // light batches
        atomicAdd(st + 3, st_avail);                               // entries lit (= lit samples)
        atomicAdd(st + 4, st_calls * (unsigned long long)batch);   // lanes offered to the batches
        atomicAdd(st + 5, (unsigned long long)batch * (unsigned long long)steps);  // density evaluations of the march (lanes x steps)
        atomicAdd(st + 6, st_ticks_b);                             // s_memtime ticks inside phase B
        atomicAdd(st + 7, __builtin_amdgcn_s_memtime() - st_t0);   // ... inside the whole cloud march
        atomicAdd(st + 8, st_calls * 64ull);                       // lanes of the wave x batches
    }
#endif
    return make_float2(rec.total_light, 1.0f - rec.one_minus_alpha);
}

template <int FLAGS, int LSTEPS, int SPLIT>
__device__ __forceinline__ void shade_pixel(const RenderConsts &rc, const int tile_x, const int tile_y) {
    constexpr bool CLOUDS = (FLAGS & KF_CLOUDS) != 0;
    constexpr bool RM = (FLAGS & KF_CLOUD_LIGHT_RM) != 0;
    constexpr bool DIRECT = (FLAGS & KF_LIGHT_DIRECT) != 0;
    constexpr bool LITE = (FLAGS & KF_LITE) != 0;
    constexpr bool PRECISE = (FLAGS & KF_PRECISE) != 0;
    constexpr bool LOD = (FLAGS & KF_CUBE_LOD) != 0;
    constexpr bool ATMO_REF = (FLAGS & KF_ATMO_REF) != 0;
    constexpr bool VIEWPOS = (FLAGS & KF_VIEW_POS) != 0;
    static_assert(!VIEWPOS || (!LITE && !ATMO_REF && SPLIT == 1), "KF_VIEW_POS: the fast v2 march, one lane per ray");
    constexpr bool DIET = !DIRECT && !((FLAGS & KF_CLOUDS) && (FLAGS & KF_CLOUD_LIGHT_RM));
    constexpr bool FASTMISS = (ATMO_FAST_MISS_MASK >> ((DIRECT ? 1 : 0) + (CLOUDS ? 2 : 0) + (LITE ? 4 : 0))) & 1;
    static_assert(!LOD || (CLOUDS && PRECISE), "implicit cubemap LOD: precise cloud kernels");
    static_assert(!(LOD && SPLIT == 2) || (!ATMO_REF && !VIEWPOS && !LITE && !DIRECT), "two lanes per ray under the declared sampler: the two BASELINE cloud kernels");
    static_assert(!LOD || (WAVE_W == 16 && WAVE_H == 4), "the quad-major lane order of the LOD kernels is written for 16 x 4 pixel waves");

    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const f32x4 *lvl_table = nullptr;
    if constexpr (LOD) {  // the declared sampler's per-level constants (cube_level_table_fill), written by every wave before any lane leaves
        __shared__ f32x4 lvl_lds[CUBE_LEVEL_TABLE + LOG2CR_ROWS];
        cube_level_table_fill(rc, lvl_lds, lane);
        log2_cr_table_fill(lvl_lds + CUBE_LEVEL_TABLE, lane);   // (round 6) the table of the declared sampler's logarithm (not what the kernels' +1 % is: ab_lambda_exact.txt, 8)
        lvl_table = lvl_lds;
    }
    // SPLIT = 2: lanes 2r, 2r+1 share ray r; a wave covers WAVE_W x (32 / WAVE_W) pixels, the workgroup TILE_W x TILE_H / 2
    // (LOD && SPLIT == 2: lane bits x0, y0, half, quad index: a pixel quad keeps its four consecutive lanes, the two lanes of a ray are lane ^ 4)
    const int ray = SPLIT == 2 ? (LOD ? ((lane >> 3) << 2) | (lane & 3) : lane >> 1) : lane;
    const int half = SPLIT == 2 ? (LOD ? (lane >> 2) & 1 : lane & 1) : 0;
    // Under the declared sampler only the CLOUD march is split: both lanes of a ray run the whole atmosphere march (a few per cent of a
    // heavy cloud ray), whose regrouped sums would differ from the one-lane kernel's in the last bits -- a frame drawn partly with this form
    // (the heavy tiles) must be the same picture.
    constexpr int ASPLIT = LOD ? 1 : SPLIT;
    const int ahalf = LOD ? 0 : half;
    constexpr int WAVES_X = TILE_W / (WAVE_W > TILE_W ? TILE_W : WAVE_W);
    constexpr int WAVE_ROWS = WAVE_H / SPLIT;
    // LOD: every 2 x 2 pixel quad occupies four consecutive lanes (lane bits: x0, y0, x1..x3, y1), so that the quad partners are
    // lane ^ 1 and lane ^ 2 (v_mov_b32_dpp quad_perm) -- within a wave's 16 x 4 pixels all the same
    const int lx = (wave % WAVES_X) * WAVE_W + (LOD ? ((ray >> 2) & 7) * 2 + (ray & 1) : ray % WAVE_W);
    const int ly = (wave / WAVES_X) * WAVE_ROWS + (LOD ? ((ray >> 5) & 1) * 2 + ((ray >> 1) & 1) : ray / WAVE_W);
    // LOD: the launch grid starts on an even pixel in x and y (rc.gx0, rc.gy0: the quads are those of the VIEWPORT, whatever the rect), and
    // a pixel outside the rect but inside the viewport is shaded as a HELPER, like the rasteriser's helper invocations: it marches, its
    // quad mates difference against it, it stores nothing.  The picture of a rect is the crop of the full frame's, bit for bit.
    const int px = (LOD ? rc.gx0 : rc.x0) + tile_x * TILE_W + lx;
    const int py = (LOD ? rc.gy0 : rc.y0) + tile_y * (TILE_H / SPLIT) + ly;
    QuadRegs qregs;
    if constexpr (LOD) quad_regs_define(qregs);  // all 64 lanes are still here: nothing else may ever live in these registers
    if (LOD ? (px >= rc.w || py >= rc.h) : (px >= rc.x1 || py >= rc.y1)) return;
    const bool helper = LOD && (px < rc.x0 || py < rc.y0 || px >= rc.x1 || py >= rc.y1);
    float4 *out = rc.out + (size_t)(py - rc.out_y0) * (size_t)rc.out_pitch + (px - rc.out_x0);

    // --- sure-miss test in front of the exact prologue (round 3) ----------------------------------------
    // A ray from the view-space origin along v misses the shell iff (c.v)^2 < (|c|^2 - R^2) |v|^2 (that is h < 0 in ray_sphere,
    // util.gdshaderinc:27-31).  With a projection whose ray direction does not depend on the depth sample (rc.miss_k > 0: the host
    // checked inv_p[8..10] == 0 and that the camera is well outside the shell) this lane evaluates the inequality on the
    // UNNORMALISED direction in plain fused fp32 -- no depth load, no divisions, no roots -- against a right-hand side shrunk
    // by 0.2 % (rc.miss_k = (|c|^2 - R^2) (1 - 1e-3)^2), three orders of magnitude more than the rounding of either evaluation.
    // "Surely misses" therefore implies the exact prologue below would discard: the lane stores the discard value and is done;
    // every other lane runs the exact path as before, so the discard set is bit-identical.  A frame whose rays all miss cost
    // 10.2 us at 1920x1080 against 6.2 us for the bare 33 MB store stream (profiles/round2/floor_probe.txt).
    if (FASTMISS && rc.miss_k > 0.0f) {
#pragma clang fp contract(fast)
        const float fnx = fmaf((float)px + 0.5f, rc.rcp_vw + rc.rcp_vw, -1.0f), fny = fmaf((float)py + 0.5f, rc.rcp_vh + rc.rcp_vh, -1.0f);
        const float *Q = rc.inv_p;
        const float ax = fmaf(Q[0], fnx, fmaf(Q[4], fny, Q[12]));
        const float ay = fmaf(Q[1], fnx, fmaf(Q[5], fny, Q[13]));
        const float az = fmaf(Q[2], fnx, fmaf(Q[6], fny, Q[14]));
        const float cv = rc.center[0] * ax + rc.center[1] * ay + rc.center[2] * az;
        const float vv = ax * ax + ay * ay + az * az;
        if (cv * cv < rc.miss_k * vv) {
            if (rc.store_discards && half == 0 && !helper) *out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            return;
        }
    }

    // --- exact prologue (main:128-169) -----------------------------------------------------------
    const float nonlinear_depth = rc.depth[(size_t)py * rc.w + px];
    const float uvx = pixel_coord<DIET>((float)px + 0.5f, rc.vw, rc.rcp_vw);
    const float uvy = pixel_coord<DIET>((float)py + 0.5f, rc.vh, rc.rcp_vh);
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
    float pwx, pwy, pwz;
    world_div3<DIET>(wx, wy, wz, ww, pwx, pwy, pwz);
    const float ddx = rc.cam_pos_world[0] - pwx, ddy = rc.cam_pos_world[1] - pwy, ddz = rc.cam_pos_world[2] - pwz;
    float linear_depth = prologue_sqrt<DIET>(ddx * ddx + ddy * ddy + ddz * ddz);

    // ray_dir = normalize(view_coords.xyz - 0) = v * (1/sqrt(dot(v,v)))
    const float vvx = vx - 0.0f, vvy = vy - 0.0f, vvz = vz - 0.0f;
    const float inv_len = ieee_div(1.0f, prologue_sqrt<DIET>(vvx * vvx + vvy * vvy + vvz * vvz));
    const V3 dir = {vvx * inv_len, vvy * inv_len, vvz * inv_len};
    const V3 center = {rc.center[0], rc.center[1], rc.center[2]};

    const SphereHit sh = sphere_setup(center, dir);
    const float2 rs_atmo = hit_radius<DIET>(sh, rc.atmosphere_radius);

    if (rs_atmo.x == rs_atmo.y) {  // discard: nothing reaches the blend stage
        if (rc.store_discards && half == 0 && !helper) *out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const float t_begin = fmaxf(rs_atmo.x, 0.0f);
    float t_end = fmaxf(rs_atmo.y, 0.0f);
    const float2 rs_ground = hit_radius<DIET>(sh, rc.planet_radius);
    float gd = 10000000.0f;
    if (rs_ground.x != rs_ground.y) gd = rs_ground.x;
    linear_depth = linear_depth * (1.0f - rc.sphere_depth_factor) + gd * rc.sphere_depth_factor;
    t_end = fminf(t_end, linear_depth);

    const float jx = rc.vw * uvx, jy = rc.vh * uvy;
    const int ji = ((int)jx) & 0xff, jj = ((int)jy) & 0xff;
    const float jitter = blue_noise_value<DIET>(rc.blue[jj * 256 + ji]);

    float4 rgba;
    if (LITE) {
        if constexpr (PRECISE) rgba = march_atmosphere_v1_precise(rc, dir, t_begin, t_end);
        else rgba = march_atmosphere_v1<ASPLIT>(rc, dir, t_begin, t_end, ahalf);  // main:172-175
    } else {
        if constexpr (ATMO_REF && SPLIT == 1) {
            rgba = march_atmosphere_v2_precise<DIRECT>(rc, dir, t_begin, t_end, jitter);  // reference order (atmo_set_precision 2)
        } else {
            const float view_step_len = ieee_div(t_end - t_begin, (float)rc.view_steps);
            rgba = march_atmosphere<DIRECT, LSTEPS, ASPLIT, VIEWPOS>(rc, dir, t_begin, view_step_len, jitter, ahalf);
        }
    }

    if (CLOUDS) {
        // --- render_clouds (cloud_funcs.gdshaderinc:249-324), gates evaluated exactly -----------------
        const float2 rs_top = hit_radius<DIET>(sh, rc.clouds_top);
        if (rs_top.x != rs_top.y) {
            const float2 rs_bottom = hit_radius<DIET>(sh, rc.clouds_bottom);
            const float c0 = fmaxf(rs_top.x, 0.0f);
            const float c1 = fminf(rs_top.y, linear_depth);
            if (c0 < linear_depth && (linear_depth > rs_bottom.y || rs_bottom.x > 0.0f)) {
                const float *M = rc.view_to_model;
                V3 dir_m;
                dir_m.x = M[0] * dir.x + M[4] * dir.y + M[8] * dir.z;
                dir_m.y = M[1] * dir.x + M[5] * dir.y + M[9] * dir.z;
                dir_m.z = M[2] * dir.x + M[6] * dir.y + M[10] * dir.z;
                float2 rr;
                if constexpr (RM && SPLIT == 1 && !LOD) {
                    __shared__ float rmq[(TILE_W * TILE_H / 64) * RMQ_WORDS_PER_WAVE];
                    rr = march_clouds_rm_queue<PRECISE>(rc, dir_m, c0, c1, jitter, rmq + wave * RMQ_WORDS_PER_WAVE);
                } else {
                    rr = march_clouds<RM, PRECISE, SPLIT, LOD>(rc, dir_m, c0, c1, jitter, half, &qregs, lvl_table);
                }
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
    if constexpr (LOD) quad_regs_keep(qregs);
    if (SPLIT == 2 && half) return;  // lane 0 of the pair holds the ray's result
    if (helper) return;              // outside the rect: shaded for its quad mates only
    if (rc.composite) {
        // What the engine's blend stage does with ALBEDO/ALPHA of an unshaded, blend_mix spatial material:
        // colour: SRC_ALPHA, ONE_MINUS_SRC_ALPHA; alpha: ONE, ONE_MINUS_SRC_ALPHA.
#pragma clang fp contract(off)
        const float4 dst = *out;
        const float ia = 1.0f - rgba.w;
        float4 o;
        o.x = rgba.x * rgba.w + dst.x * ia;
        o.y = rgba.y * rgba.w + dst.y * ia;
        o.z = rgba.z * rgba.w + dst.z * ia;
        o.w = rgba.w + dst.w * ia;
        *out = o;
    } else {
        *out = rgba;
    }
}

// The launch: one workgroup per TILE_W x (TILE_H / SPLIT) pixel tile.
//
// Tile order with cost feedback (rc.tile_order / rc.tile_cost, atmo_set_tile_feedback): the hardware dispatches
// workgroups in blockIdx order, and at 1920x1080 the kernel time of the cloud variants is set by the critical path of
// the few heaviest waves (all lanes in dense cloud: ~100 k VALU instructions against 11 k on average) when they happen
// to start late; and every variant drains for the last ~12 % of a 1920x1080 draw (tools/wave_timeline.py), which cheap
// tiles dispatched last fill.  On a recording draw (rc.tile_cost set: every 8th) every wave records its duration
// (s_memtime); atmo_tile_order_kernel then sorts the tiles by that cost on a side stream, heaviest first
// (longest-processing-time-first list scheduling), and blockIdx indexes the sorted list in the draws that follow.  Frames
// of an animation are coherent, so earlier costs predict this frame's; the picture does not depend on the order.
// SGPR budget: a SIMD holds 800 SGPRs and a wave is charged its allocation (16-granular) + 16, so waves per SIMD =
// floor(800 / (ceil(.sgpr_count / 16) * 16 + 16)): .sgpr_count <= 80 => 8, 81-96 => 7, 97-112 => 6, although the occupancy API
// and the compiler's "Occupancy" line still say 8 (MI355X_MICROARCH.md "Residency").  What the build does: NOTHING -- no kernel is
// compiled under an SGPR cap.  The direct-light no-cloud kernels sit at .sgpr_count 82 (7 waves per SIMD; their 40 VGPRs would allow
// 8); the twin kernel atmo_render_kernel_s80 below compiles them to 78 with the same loop ISA.  Round 4 measured that build 8 % SLOWER at
// 1920x1080 and 4 % at 3840x2160 (profiles/round4/ab_sgpr_cap.txt) and blamed the cap; round 6 found the cause -- the capped build moves the
// view loop by four bytes, off its fast position (march_atmosphere, profiles/round6/ab_loop_phase.txt 3) -- and that at the right position
// 8 waves per SIMD are exactly as fast as 7 (0.0867 against 0.0865 ms): the kernel is bound by VALU issue, residency buys nothing.
// The cloud kernels (86-106 SGPRs, 53-89 VGPRs: 5-7 waves either way) lost
// 5-7 % under a cap in round 2.  The mask below selects families for the next A/B (amdgpu_num_sgpr takes a literal, hence the twin).
#ifndef ATMO_MIN_WAVES  // __launch_bounds__ second argument: minimum waves per SIMD the register allocation must allow (0 = none).
#define ATMO_MIN_WAVES 6  // 6: only atmo_render_kernel<19 / 23, ..> change (84 -> 80 VGPRs); see ATMO_RMQ_CHUNK
#endif
#ifndef ATMO_MIN_WAVES_LOD_RM  // the declared-sampler raymarched-light kernels: bound 4 = no constraint in practice.  The in-place form (rounds 5-6) needs 74-75
#define ATMO_MIN_WAVES_LOD_RM 4  // VGPRs by itself = six waves; the queue form it replaced needed 111 (round 3) / 88 (round 4) and a bound of 5 cost it +3 %
#endif                           // (profiles/round3/ab_occupancy.txt).
#ifndef ATMO_MIN_WAVES_LOD
#define ATMO_MIN_WAVES_LOD ATMO_MIN_WAVES
#endif
constexpr int render_min_waves(int flags) {
    return (flags & KF_CUBE_LOD) ? ((flags & KF_CLOUD_LIGHT_RM) ? ATMO_MIN_WAVES_LOD_RM : ATMO_MIN_WAVES_LOD) : ATMO_MIN_WAVES;
}
#if ATMO_MIN_WAVES > 0
#define ATMO_MIN_WAVES_ARG , render_min_waves(FLAGS)
#else
#define ATMO_MIN_WAVES_ARG
#endif
// ATMO_SGPR_CAP_MASK: bit number (direct + 2 clouds + 4 lite) set = that kernel family runs under the cap (tools/ab_build.sh x -DATMO_SGPR_CAP_MASK=0x02)
#ifndef ATMO_SGPR_CAP_VALUE
#define ATMO_SGPR_CAP_VALUE 80
#endif
#ifndef ATMO_SGPR_CAP_MASK
#define ATMO_SGPR_CAP_MASK 0x00
#endif
constexpr bool render_sgpr_cap80(int flags) {
    return (ATMO_SGPR_CAP_MASK >> (((flags & KF_LIGHT_DIRECT) ? 1 : 0) + ((flags & KF_CLOUDS) ? 2 : 0) + ((flags & KF_LITE) ? 4 : 0))) & 1;
}

// The kernel body, textually the same in both kernels (a shared __forceinline__ function changes hipcc's scheduling of the
// preamble, and these kernels are sensitive to exactly that: see the note inside).
#ifdef ATMO_WAVE_TRACE
#define ATMO_TRACE_ENTRY const uint64_t trace_entry = __builtin_amdgcn_s_memrealtime();
#define ATMO_SHADE_TRACED                                                                                                          \
    const uint64_t trace_t0 = __builtin_amdgcn_s_memrealtime();                                                                    \
    shade_pixel<FLAGS, LSTEPS, SPLIT>(rc, (int)tile_x, (int)tile_y);                                                               \
    if (rc.wave_trace != nullptr && (threadIdx.x & 63) == 0) {                                                                     \
        const uint32_t slot = (blockIdx.y * gridDim.x + blockIdx.x) * (TILE_W * TILE_H / 64) + (threadIdx.x >> 6);                 \
        unsigned long long *w = rc.wave_trace + 4ull * slot;                                                                       \
        w[0] = trace_t0;                                                                                                           \
        w[1] = __builtin_amdgcn_s_memrealtime();                                                                                   \
        w[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); /* HW_REG_HW_ID, 32 bits */                                              \
        w[3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) | ((trace_t0 - trace_entry) << 8); /* XCC_ID, preamble ticks */ \
    }
#else
#define ATMO_TRACE_ENTRY
#define ATMO_SHADE_TRACED shade_pixel<FLAGS, LSTEPS, SPLIT>(rc, (int)tile_x, (int)tile_y);
#endif
// The geometric tile order (RenderConsts::geo_rows; cloudless kernels): block b shades the b-th tile of "the tiles that can shade, row-major, then the others,
// row-major".  Uniform: scalar ALU and scalar loads from the kernel-argument segment only (a hint per 256 blocks, then one or two steps of a binary search).
__device__ __forceinline__ uint32_t geo_tile(const RenderConsts &rc, uint32_t b) {
    const uint32_t rows = (uint32_t)rc.geo_rows, tx = (uint32_t)rc.tiles_x, total = rc.geo_prefix[rows];
    if (b < total) {                       // largest row r with geo_prefix[r] <= b, between the two hints around b
        uint32_t lo = rc.geo_hint[0][b >> 8], hi = min((uint32_t)rc.geo_hint[0][(b >> 8) + 1] + 1u, rows);
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rc.geo_prefix[mid] <= b) lo = mid; else hi = mid;
        }
        return lo * tx + rc.geo_first[lo] + (b - rc.geo_prefix[lo]);
    }
    const uint32_t m = b - total;           // index among the tiles that cannot shade: row r holds r tx - geo_prefix[r] of them above it
    uint32_t lo = rc.geo_hint[1][m >> 8], hi = min((uint32_t)rc.geo_hint[1][(m >> 8) + 1] + 1u, rows);
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (mid * tx - rc.geo_prefix[mid] <= m) lo = mid; else hi = mid;
    }
    const uint32_t o = m - (lo * tx - rc.geo_prefix[lo]), first = rc.geo_first[lo], len = rc.geo_prefix[lo + 1] - rc.geo_prefix[lo];
    return lo * tx + (o < first ? o : o + len);
}
#ifndef ATMO_LOOP_PAD_GEO   // s_nop in front of shade_pixel in the twin kernels <4 | KF_GEO, 8, 1>: puts their view loop 12 bytes into a 32-byte block
#define ATMO_LOOP_PAD_GEO 6
#endif
#define ATMO_GEO_STR2(x) #x
#define ATMO_GEO_STR(x) ATMO_GEO_STR2(x)
#define ATMO_GEO_PAD_STR ATMO_GEO_STR(ATMO_LOOP_PAD_GEO)
// Keep this preamble exactly as it is for every variant.  Measured on the direct-light kernel (same loop ISA in all
// three builds, profiles/round2/ab_direct_kernel.txt): this form 0.108-0.109 ms; a branch on tile_order in front of
// the division 0.115 ms; NO preamble at all (blockIdx used directly) 0.115 ms as well.  Round 2 read that as "the scalar work in
// front of the depth load helps"; round 6 found what it was: every one of those edits moved the direct-light kernel's view loop by a few bytes, and
// that loop is 8.5-11 % slower at seven of its eight possible 4-byte positions (march_atmosphere; profiles/round6/ab_loop_phase.txt).  The preamble
// itself is neutral; the POSITION is what must be kept (tests/test_host_logic.py::test_headline_view_loop_sits_at_its_fast_position).
#define ATMO_RENDER_KERNEL_BODY                                                                                  \
    ATMO_TRACE_ENTRY                                                                                             \
    uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;                                                         \
    if constexpr ((FLAGS & KF_GEO) != 0) {   /* the twin kernels of the geometric order only: render_impl launches them with a table */ \
        tile = geo_tile(rc, tile);                                                                               \
        /* ... and their view loop on ITS fast position (march_atmosphere; tools/loop_phase.py reads both kernels) */ \
        asm volatile(".rept " ATMO_GEO_PAD_STR "\n\ts_nop 0\n\t.endr");                                          \
    }                                                                                                            \
    if (rc.tile_order != nullptr) tile = rc.tile_order[tile];                                                    \
    const uint32_t tile_y = tile / (uint32_t)rc.tiles_x, tile_x = tile - tile_y * (uint32_t)rc.tiles_x;          \
    uint64_t t0 = 0;                                                                                             \
    if (rc.tile_cost != nullptr) t0 = __builtin_amdgcn_s_memtime();                                              \
    ATMO_SHADE_TRACED                                                                                            \
    if (rc.tile_cost != nullptr && (threadIdx.x & 63) == 0) {                                                    \
        uint64_t dt = __builtin_amdgcn_s_memtime() - t0;                                                         \
        uint32_t cost_tile = tile;                                                                               \
        if constexpr (SPLIT == 2 && (FLAGS & KF_CUBE_LOD) != 0) {                                                \
            /* a heavy tile of a one-lane frame drawn split: the cost map stays the one-lane grid's.  Only there: a WHOLE frame   */ \
            /* on two lanes per ray (atmo_set_lane_split 2, --shard tiles --lanes 2) has a cost map of its own, half-height grid.          */ \
            if (rc.cost_rows_halved) {                                                                           \
                cost_tile = (tile_y >> 1) * (uint32_t)rc.tiles_x + tile_x;                                       \
                dt *= 2;                                                                                         \
            }                                                                                                    \
        }                                                                                                        \
        atomicMax(&rc.tile_cost[cost_tile], (uint32_t)(dt > 0xffffffffull ? 0xffffffffull : dt));                \
    }

template <int FLAGS, int LSTEPS, int SPLIT = 1>
__global__ __launch_bounds__(TILE_W *TILE_H ATMO_MIN_WAVES_ARG) void atmo_render_kernel(const RenderConsts rc) {
    ATMO_RENDER_KERNEL_BODY
}
// the same kernel under an 80-SGPR cap (8 waves per SIMD), for the families render_sgpr_cap80 names (none in the shipped build)
template <int FLAGS, int LSTEPS, int SPLIT = 1>
__global__ __launch_bounds__(TILE_W *TILE_H ATMO_MIN_WAVES_ARG) __attribute__((amdgpu_num_sgpr(ATMO_SGPR_CAP_VALUE))) void atmo_render_kernel_s80(const RenderConsts rc) {
    ATMO_RENDER_KERNEL_BODY
}

// Stable counting sort of the tiles by the cost a recording draw measured, heaviest class first; clears the costs for
// the next recording.  64 classes = quarter octaves of the wave duration (2^8 .. 2^24 cycles; 32 half octaves until round 6); tiles of one class keep their
// row-major order, so neighbouring tiles -- which share texture footprints in L1/L2 -- still run together.  How fine: measured with tile-list draws in the
// library's class order at 32 / 64 / 128 classes and in exact cost order (profiles/round6/ab_order_classes.txt) -- clouds_high at 1920x1080 gains 2.2 / 3.0 / 3.0 %
// over 32, the headline kernel loses 0.6 / 0.7 / 2.7 % (an exact sort leaves no row-major runs), the rest is within +-0.5 %: 64, which also still fits the sort's
// one-lane-per-class state.
// Three small kernels on the context's high-priority side stream (round 3; the round-2 form was ONE 512-thread workgroup
// with 64 KB of LDS that had to find a free CU beside the draw it runs next to: 33 us at best, 52-713 us on average --
// that latency is feedback lag when the camera moves):
//   histogram  256 single-wave workgroups, each owns a contiguous chunk of tiles and counts its classes (wave ballots);
//   scan       one workgroup turns the 256 x ORDER_CLASSES counts into the first output index of every (chunk, class);
//   scatter    the 256 waves write their tiles, in order, behind those indices and clear the costs.
constexpr int ORDER_BLOCKS = 256, ORDER_CLASSES = TILE_ORDER_CLASSES;
constexpr int ORDER_SUB_BITS = TILE_ORDER_PER_OCTAVE == 4 ? 2 : 1, ORDER_KEY_BITS = 4 + ORDER_SUB_BITS;   // mantissa bits that split an octave; bits of a class index
__device__ __forceinline__ uint32_t tile_cost_class(uint32_t c) {
    if (c == 0) return ORDER_CLASSES - 1;
    const int msb = 31 - __builtin_clz(c);
    const int q = msb * TILE_ORDER_PER_OCTAVE + (msb >= ORDER_SUB_BITS ? (int)((c >> (msb - ORDER_SUB_BITS)) & (uint32_t)(TILE_ORDER_PER_OCTAVE - 1)) : 0) - 8 * TILE_ORDER_PER_OCTAVE;
    return (uint32_t)(ORDER_CLASSES - 1 - (q < 0 ? 0 : (q > ORDER_CLASSES - 1 ? ORDER_CLASSES - 1 : q)));
}
// lanes of this wave whose class index equals mine (inactive lanes excluded by `valid`)
__device__ __forceinline__ unsigned long long match_class(uint32_t key, bool valid) {
    unsigned long long m = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
    for (int bit = 0; bit < ORDER_KEY_BITS; ++bit) {
        const bool one = (key >> bit) & 1u;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(one);
        m &= one ? b : ~b;
    }
    return m;
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__global__ __launch_bounds__(64) void atmo_tile_hist_kernel(const uint32_t *__restrict__ cost, uint32_t *__restrict__ hist, int n) {
    __shared__ uint32_t cnt[ORDER_CLASSES];
    const int lane = threadIdx.x, chunk = (n + ORDER_BLOCKS - 1) / ORDER_BLOCKS;
    const int i0 = min((int)blockIdx.x * chunk, n), i1 = min(i0 + chunk, n);
    if (lane < ORDER_CLASSES) cnt[lane] = 0;
    __syncthreads();
    for (int i = i0; i < i1; i += 64) {
        const bool valid = i + lane < i1;
        const uint32_t c = valid ? tile_cost_class(cost[i + lane]) : 0u;
        const unsigned long long m = match_class(c, valid);
        if (valid && lanes_below(m) == 0) cnt[c] += (uint32_t)__builtin_popcountll(m);  // one lane per class present
    }
    __syncthreads();
    if (lane < ORDER_CLASSES) hist[lane * ORDER_BLOCKS + blockIdx.x] = cnt[lane];   // [class][block]: the scan reads a class's row contiguously
}

// hist[class][block] -> first output index of (block, class): classes in order 0 (heaviest) .. ORDER_CLASSES - 1, blocks in order inside a class
// class_totals (may be null): the number of tiles in every cost class, for the host (pinned memory: it picks the frame's heavy tiles from them)
__global__ __launch_bounds__(256) void atmo_tile_scan_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ class_totals) {
    __shared__ uint32_t total[ORDER_CLASSES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER_LANE = ORDER_BLOCKS / 64, PER_WAVE = ORDER_CLASSES / 4;
    uint32_t excl_keep[PER_WAVE][PER_LANE];
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const int c = wave * PER_WAVE + q;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            excl_keep[q][k] = sum;
            sum += hist[c * ORDER_BLOCKS + lane * PER_LANE + k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(incl, d);
            if (lane >= d) incl += x;
        }
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) excl_keep[q][k] += incl - sum;
        if (lane == 63) total[c] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the class totals
        const uint32_t c = lane < ORDER_CLASSES ? total[lane] : 0u;
        if (class_totals != nullptr && lane < ORDER_CLASSES) class_totals[lane] = c;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(incl, d);
            if (lane >= d) incl += x;
        }
        if (lane < ORDER_CLASSES) total[lane] = incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const int c = wave * PER_WAVE + q;
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) hist[c * ORDER_BLOCKS + lane * PER_LANE + k] = total[c] + excl_keep[q][k];
    }
}

// order2 (may be null): the same order for the launch grid of the two-lanes-per-ray kernels, whose tiles are half as high -- entries 2 p and
// 2 p + 1 are the upper and the lower half of the tile at position p (the host draws the first few, the heavy tiles, with those kernels)
__global__ __launch_bounds__(64) void atmo_tile_scatter_kernel(const uint32_t *key, uint32_t *cost,  /* key == cost without dilation */
                                                               const uint32_t *__restrict__ base, uint32_t *__restrict__ order, int n,
                                                               uint32_t *__restrict__ order2, int tiles_x) {
    __shared__ uint32_t off[ORDER_CLASSES];
    const int lane = threadIdx.x, chunk = (n + ORDER_BLOCKS - 1) / ORDER_BLOCKS;
    const int i0 = min((int)blockIdx.x * chunk, n), i1 = min(i0 + chunk, n);
    if (lane < ORDER_CLASSES) off[lane] = base[lane * ORDER_BLOCKS + blockIdx.x];
    __syncthreads();
    for (int i = i0; i < i1; i += 64) {
        const bool valid = i + lane < i1;
        const uint32_t c = valid ? tile_cost_class(key[i + lane]) : 0u;
        const unsigned long long m = match_class(c, valid);
        const int r = lanes_below(m);
        uint32_t pos = 0;
        if (valid) pos = off[c] + (uint32_t)r;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && r == 0) off[c] += (uint32_t)__builtin_popcountll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid) {
            const uint32_t t = (uint32_t)(i + lane);
            order[pos] = t;
            if (order2 != nullptr) {
                const uint32_t ty = t / (uint32_t)tiles_x, tx = t - ty * (uint32_t)tiles_x;
                order2[2u * pos] = (2u * ty) * (uint32_t)tiles_x + tx;
                order2[2u * pos + 1u] = (2u * ty + 1u) * (uint32_t)tiles_x + tx;
            }
            cost[i + lane] = 0;   // ready for the next recording draw
        }
    }
}

// Dilation of the cost map for a MOVING camera (separable max filter, radius rx / ry tiles): the order a sort produces is
// used a few frames later, when every feature of the cost map -- the planet's limb, the terminator, a cloud bank -- has
// moved on by up to the distance the host predicts from the camera matrices (atmo_api.hip, feedback_motion_px).  A tile that
// was cheap but lies within that distance of an expensive one may be expensive by then, and one such tile dispatched at
// the very end of a draw costs its whole duration as tail (measured: panning at 1 degree per frame turned the +48 % of
// clouds_high_rm into -6.5 %, profiles/round3/ab_tile_feedback_motion.txt).  The max filter makes the cost the sort sees an
// upper bound of what the tile can cost after the move: only tiles that are SURELY still cheap go last.
__global__ __launch_bounds__(256) void atmo_tile_dilate_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int tiles_x, int tiles_y,
                                                               int rx, int ry) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= tiles_x * tiles_y) return;
    const int y = i / tiles_x, x = i - y * tiles_x;
    uint32_t m = 0;
    for (int d = -ry; d <= ry; ++d) {
        const int yy = y + d;
        if (yy < 0 || yy >= tiles_y) continue;
        for (int e = -rx; e <= rx; ++e) {
            const int xx = x + e;
            if (xx >= 0 && xx < tiles_x) m = max(m, in[yy * tiles_x + xx]);
        }
    }
    out[i] = m;
}

// The class totals of the UNDILATED costs (one workgroup; a moving camera's sort): the host's estimate of the draw's duration and of its heaviest
// wavefront (heavy_tile_count) must not see the max-filtered map the order is sorted from -- every tile within reach of a heavy one counts as heavy
// there, the sum of lifetimes is inflated severalfold and no draw ever looks bound by its tail.
__global__ __launch_bounds__(1024) void atmo_tile_class_totals_kernel(const uint32_t *__restrict__ cost, int n, uint32_t *__restrict__ class_totals) {
    __shared__ uint32_t cnt[ORDER_CLASSES];
    if (threadIdx.x < ORDER_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = 0; i < n; i += 1024) {
        const bool valid = i + (int)threadIdx.x < n;
        const uint32_t c = valid ? tile_cost_class(cost[i + threadIdx.x]) : 0u;
        const unsigned long long m = match_class(c, valid);
        if (valid && lanes_below(m) == 0) atomicAdd(&cnt[c], (uint32_t)__builtin_popcountll(m));   // one lane per class and wave
    }
    __syncthreads();
    if (threadIdx.x < ORDER_CLASSES) class_totals[threadIdx.x] = cnt[threadIdx.x];
}

// scratch: ORDER_BLOCKS * ORDER_CLASSES uint32 (tile_order_scratch_bytes); tmp1 / tmp2: n uint32 each, used when rx | ry > 0
size_t tile_order_scratch_bytes() { return (size_t)ORDER_BLOCKS * ORDER_CLASSES * sizeof(uint32_t); }
hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, int tiles_x, int tiles_y, int rx, int ry, uint32_t *tmp1, uint32_t *tmp2,
                             uint32_t *scratch, hipStream_t stream, uint32_t *order2, uint32_t *class_totals) {
    const int n = tiles_x * tiles_y;
    const uint32_t *key = cost;
    if ((rx > 0 || ry > 0) && class_totals != nullptr) {   // the histogram the host reads: of the measured costs, not of the dilated key
        hipLaunchKernelGGL(atmo_tile_class_totals_kernel, dim3(1), dim3(1024), 0, stream, cost, n, class_totals);
        class_totals = nullptr;
    }
    if ((rx > 0 || ry > 0) && (2 * rx + 1) * (2 * ry + 1) <= 81) {  // a small window (the in-stream sort's: a tile or two): one pass
        hipLaunchKernelGGL(atmo_tile_dilate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cost, tmp2, tiles_x, tiles_y, rx, ry);
        key = tmp2;
    } else if (rx > 0 || ry > 0) {  // separable: rows, then columns
        hipLaunchKernelGGL(atmo_tile_dilate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cost, tmp1, tiles_x, tiles_y, rx, 0);
        hipLaunchKernelGGL(atmo_tile_dilate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tmp1, tmp2, tiles_x, tiles_y, 0, ry);
        key = tmp2;
    }
    hipLaunchKernelGGL(atmo_tile_hist_kernel, dim3(ORDER_BLOCKS), dim3(64), 0, stream, key, scratch, n);
    hipLaunchKernelGGL(atmo_tile_scan_kernel, dim3(1), dim3(256), 0, stream, scratch, class_totals);
    hipLaunchKernelGGL(atmo_tile_scatter_kernel, dim3(ORDER_BLOCKS), dim3(64), 0, stream, key, cost, scratch, order, n, order2, tiles_x);
    return hipGetLastError();
}

// atmo_render_tiles: the caller's tile list with every index beyond the launch grid replaced by `sentinel`, a tile that lies wholly below the
// viewport (its lanes leave at shade_pixel's bounds test).  In front of the draw, on its stream; the render kernels stay untouched.
// n_heavy > 0 (atmo_render_tiles_split): the first n_heavy tiles of the list are drawn with two lanes per ray -- their two half-height tiles in the
// lane-split kernels' launch grid go to out2[2 i], out2[2 i + 1] (sentinel2: a tile of THAT grid below the viewport).
__global__ __launch_bounds__(256) void atmo_tile_list_bound_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int n, uint32_t tiles_n,
                                                                   uint32_t sentinel, uint32_t *__restrict__ out2, int n_heavy, int tiles_x,
                                                                   uint32_t sentinel2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t t = in[i];
        const bool ok = t < tiles_n;
        out[i] = ok ? t : sentinel;
        if (i < n_heavy) {
            const uint32_t ty = t / (uint32_t)tiles_x, tx = t - ty * (uint32_t)tiles_x;
            out2[2 * i] = ok ? (2u * ty) * (uint32_t)tiles_x + tx : sentinel2;
            out2[2 * i + 1] = ok ? (2u * ty + 1u) * (uint32_t)tiles_x + tx : sentinel2;
        }
    }
}
hipError_t launch_tile_list_bound(const uint32_t *in, uint32_t *out, int n, uint32_t tiles_n, uint32_t sentinel, hipStream_t stream, uint32_t *out2,
                                  int n_heavy, int tiles_x, uint32_t sentinel2) {
    hipLaunchKernelGGL(atmo_tile_list_bound_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, out, n, tiles_n, sentinel, out2, n_heavy, tiles_x,
                       sentinel2);
    return hipGetLastError();
}

// ---- texture re-layout on the device (atmo_set_texture), stream-ordered; element definitions in atmo_layout.h ------------
__global__ __launch_bounds__(256) void atmo_layout_lut_kernel(const float *__restrict__ lut, int w, int h, float *__restrict__ out) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < w + 2 && j < h + 2) out[(size_t)j * (w + 2) + i] = lut_apron_value(lut, w, h, i, j);
}

__global__ __launch_bounds__(256) void atmo_layout_shape_kernel(const uint8_t *__restrict__ t, int n, uint32_t *__restrict__ out) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6), k = blockIdx.z;
    if (i < n && j < n) out[((size_t)k * n + j) * n + i] = shape_footprint_word(t, n, i, j, k);
}

// one mip level: faces = 6 x n^2 texels of that level, out = its 6 x (n+1)^2 footprint words
__global__ __launch_bounds__(256) void atmo_layout_cube_kernel(const uint8_t *__restrict__ faces, int n, uint32_t *__restrict__ out) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (i <= n && j <= n) out[((size_t)f * (n + 1) + j) * (n + 1) + i] = cube_footprint_word(faces, n, f, i, j);
}

// next = 2x2 box of level (n per side) -> n/2 per side; Image.generate_mipmaps on L8 (noise_cubemap.gd:135)
__global__ __launch_bounds__(256) void atmo_cube_mip_kernel(const uint8_t *__restrict__ level, int n, uint8_t *__restrict__ next) {
    const int m = n >> 1;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (i < m && j < m) next[((size_t)f * m + j) * m + i] = cube_mip_texel(level, n, f, i, j);
}

hipError_t launch_layout_lut(const float *lut, int w, int h, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_layout_lut_kernel, dim3((w + 2 + 63) / 64, (h + 2 + 3) / 4), dim3(256), 0, stream, lut, w, h, out);
    return hipGetLastError();
}
hipError_t launch_layout_shape(const uint8_t *t, int n, uint32_t *out, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_layout_shape_kernel, dim3((n + 63) / 64, (n + 3) / 4, n), dim3(256), 0, stream, t, n, out);
    return hipGetLastError();
}
// Footprint words -> four exact byte / 255 floats each: the samplers then skip the 12 conversion instructions.
__global__ __launch_bounds__(256) void atmo_footprints_f4_kernel(const uint32_t *__restrict__ words, size_t n, float4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = words[i];
    out[i] = make_float4(unorm8_exact(ub0(w)), unorm8_exact(ub1(w)), unorm8_exact(ub2(w)), unorm8_exact(ub3(w)));
}
hipError_t launch_footprints_f4(const uint32_t *words, size_t n_words, float *out4, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_footprints_f4_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, words, n_words, (float4 *)out4);
    return hipGetLastError();
}

hipError_t launch_layout_cube(const uint8_t *faces, int n, uint32_t *out, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_layout_cube_kernel, dim3((n + 1 + 63) / 64, (n + 1 + 3) / 4, 6), dim3(256), 0, stream, faces, n, out);
    return hipGetLastError();
}
hipError_t launch_cube_mip(const uint8_t *level, int n, uint8_t *next, hipStream_t stream) {
    const int m = n >> 1;
    if (m < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(atmo_cube_mip_kernel, dim3((m + 63) / 64, (m + 3) / 4, 6), dim3(256), 0, stream, level, n, next);
    return hipGetLastError();
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
    // apron layout: texel (i,j) at [(j+1)*(w+2) + i+1]; edge lanes also fill the clamp-to-edge border
    const int st = bc.w + 2;
    float *o = bc.out + (j + 1) * st + (i + 1);
    o[0] = od;
    const bool el = (i == 0), er = (i == bc.w - 1), et = (j == 0), eb = (j == bc.h - 1);
    if (el) o[-1] = od;
    if (er) o[1] = od;
    if (et) o[-st] = od;
    if (eb) o[st] = od;
    if (el && et) o[-st - 1] = od;
    if (er && et) o[-st + 1] = od;
    if (el && eb) o[st - 1] = od;
    if (er && eb) o[st + 1] = od;
}

// ---- NoiseCubemap generator (noise_cubemap.gd:101-140) -------------------------------------------------------
// One lane per texel, exact fp32 arithmetic (unfused, IEEE sqrt/divide) so the bytes equal a scalar evaluation.
// The noise itself is this build's seeded value noise (the reference calls Godot's FastNoiseLite, engine code).
__device__ __forceinline__ uint32_t nz_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float nz_lattice(int ix, int iy, int iz, uint32_t seed) {
    const uint32_t h = ((uint32_t)ix * 0x9E3779B1u) ^ ((uint32_t)iy * 0x85EBCA77u) ^ ((uint32_t)iz * 0xC2B2AE3Du) ^ seed;
    return (float)(nz_hash(h) >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float nz_value(float px, float py, float pz, uint32_t seed) {
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const float tx = px - fx, ty = py - fy, tz = pz - fz;
    const float wx = tx * tx * (3.0f - 2.0f * tx), wy = ty * ty * (3.0f - 2.0f * ty), wz = tz * tz * (3.0f - 2.0f * tz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const float c00 = nz_lattice(x0, y0, z0, seed) * (1.0f - wx) + nz_lattice(x1, y0, z0, seed) * wx;
    const float c10 = nz_lattice(x0, y1, z0, seed) * (1.0f - wx) + nz_lattice(x1, y1, z0, seed) * wx;
    const float c01 = nz_lattice(x0, y0, z1, seed) * (1.0f - wx) + nz_lattice(x1, y0, z1, seed) * wx;
    const float c11 = nz_lattice(x0, y1, z1, seed) * (1.0f - wx) + nz_lattice(x1, y1, z1, seed) * wx;
    const float c0 = c00 * (1.0f - wy) + c10 * wy;
    const float c1 = c01 * (1.0f - wy) + c11 * wy;
    return c0 * (1.0f - wz) + c1 * wz;
}

__global__ __launch_bounds__(256) void atmo_noise_cubemap_kernel(const NoiseCubemapConsts nc) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int side = blockIdx.z;
    const int res = nc.resolution;
    if (x >= res || y >= res) return;
    // noise_cubemap.gd:110-113: pos2d, +X direction
    const float half = 0.5f * (float)res;
    const float p2x = ieee_div((float)x + 0.5f, half) - 1.0f;
    const float p2y = ieee_div((float)(res - y - 1) + 0.5f, half) - 1.0f;
    float vx = 1.0f, vy = p2y, vz = -p2x;
    const float len = ieee_sqrt(vx * vx + vy * vy + vz * vz);
    vx = ieee_div(vx, len); vy = ieee_div(vy, len); vz = ieee_div(vz, len);
    float ox, oy, oz;  // :116-128
    switch (side) {
    case 0: ox = vx;  oy = vy;  oz = vz;  break;
    case 1: ox = -vx; oy = vy;  oz = -vz; break;
    case 2: ox = -vz; oy = vx;  oz = -vy; break;
    case 3: ox = -vz; oy = -vx; oz = vy;  break;
    case 4: ox = -vz; oy = vy;  oz = vx;  break;
    default: ox = vz; oy = vy;  oz = -vx; break;
    }
    const float px = ox * nc.scale[0], py = oy * nc.scale[1], pz = oz * nc.scale[2];
    float total = 0.0f, amp = 1.0f, norm = 0.0f, freq = nc.frequency;
    for (int o = 0; o < nc.octaves; ++o) {
        total += amp * nz_value(px * freq, py * freq, pz * freq, nc.seed + 1013u * (uint32_t)o);
        norm += amp;
        amp *= nc.gain;
        freq *= 2.0f;
    }
    const float n = 2.0f * ieee_div(total, norm) - 1.0f;
    const float density = 0.5f + 0.5f * n;  // :130
    float v = density * 255.0f;
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    nc.out[((size_t)side * res + y) * res + x] = (uint8_t)v;  // L8 store, truncating
}

hipError_t launch_noise_cubemap(const NoiseCubemapConsts &nc, hipStream_t stream) {
    dim3 grid((nc.resolution + 63) / 64, (nc.resolution + 3) / 4, 6);
    hipLaunchKernelGGL(atmo_noise_cubemap_kernel, grid, dim3(256), 0, stream, nc);
    return hipGetLastError();
}

// ---- probe of the direct light march (atmo_debug_marched_optical_depth) ----------------------------------------------------
// Evaluates sun_od_direct -- the very function march_atmosphere<DIRECT> inlines -- for n given sample positions (relative to the
// planet centre) and sun directions, so the kernel's light march can be held against the reference's LUT texels, which
// tabulate the same integral at its texel-centre geometry (optical_depth.gdshader:45-65).  Same set-up arithmetic as the
// head of march_atmosphere's loop body.
__global__ __launch_bounds__(256) void atmo_light_probe_kernel(const float *__restrict__ pos, const float *__restrict__ dir, int n,
                                                               float planet_radius, float atmosphere_height, float density, int light_steps,
                                                               float *__restrict__ out) {
#pragma clang fp contract(fast)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float inv_h = hw_rcp(atmosphere_height);
    LightMarchConsts k;
    k.ninv_h = -inv_h;
    k.c1 = fmaf(planet_radius, inv_h, 1.0f);
    k.dens2 = density * density;
    const float ratm = planet_radius + atmosphere_height;
    k.ratm2 = ratm * ratm;
    k.light_steps = light_steps;
    k.inv_light_steps = hw_rcp((float)light_steps);
    const float ox = pos[3 * i], oy = pos[3 * i + 1], oz = pos[3 * i + 2];
    const float sx = dir[3 * i], sy = dir[3 * i + 1], sz = dir[3 * i + 2];
    const float r2 = ox * ox + oy * oy + oz * oz;
    const float bdot = ox * sx + oy * sy + oz * sz;
    const float r = hw_sqrt(r2);
    const float y = sat(fmaf(r, k.ninv_h, k.c1));
    out[i] = light_steps == 8 ? sun_od_direct<8>(k, r2, bdot, y * y * y) : sun_od_direct<0>(k, r2, bdot, y * y * y);
}

hipError_t launch_light_probe(const float *pos, const float *dir, int n, float planet_radius, float atmosphere_height, float density,
                              int light_steps, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_light_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pos, dir, n, planet_radius, atmosphere_height,
                       density, light_steps, out);
    return hipGetLastError();
}

// ---- self-test of the exact helpers against the compiler's IEEE expansions ---------------------------------
// For `count` consecutive float bit patterns starting at first_bits: counts exact_sqrt(x) != sqrtf(x) and
// exact_div_uniform(x, c, RN(1/c)) != x / c.
__global__ __launch_bounds__(256) void atmo_selftest_kernel(uint32_t first_bits, uint32_t count, float c, float rc,
                                                            unsigned int *mismatch /* [2] */) {
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned int bad_sqrt = 0, bad_div = 0;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const float x = __int_as_float((int)(first_bits + k));
        const int want = __float_as_int(ieee_sqrt(x));  // both short forms: the prologue's and the cloud chain's
        if (x >= 0.0f && x < 1.9721523e-31f) {
            // 0, denormals and the binades below 2^-102: exact_sqrt_pos is not the IEEE root there (x = 0 gives NaN) and does not have to be --
            // what the cloud chain needs from such an |p|^2 (a sample at the planet's centre) is the OUTCOME: the height curve of a layer
            // whose bottom shell has a positive radius is 0, i.e. the density evaluation leaves through its first early-out.  A NaN root gets
            // there through fmaxf(NaN, 0) = 0 (IEEE maxNum, which v_max_f32 implements): pinned here instead of assumed.
            const float bottom = c, thick = c;  // any positive shell radius and thickness
            auto height_curve = [&](float r) {
                const float hr = exact_div_uniform(r - bottom, thick, rc);
                const float t = 2.0f * hr - 1.0f;
                return fmaxf(1.0f - t * t, 0.0f);
            };
            const float hc_fast = height_curve(exact_sqrt_pos(x)), hc_ieee = height_curve(ieee_sqrt(x));
            // (the prologue's exact_sqrt equals IEEE for x = 0 and from 2^-96 up; in between it may be 1 ulp off, as its comment says)
            if ((hc_fast > 0.0f) || (hc_ieee > 0.0f) || (x == 0.0f && __float_as_int(exact_sqrt(x)) != want)) ++bad_sqrt;
        } else if (__float_as_int(exact_sqrt(x)) != want || __float_as_int(exact_sqrt_pos(x)) != want) ++bad_sqrt;
        if (__float_as_int(exact_div_uniform(x, c, rc)) != __float_as_int(ieee_div(x, c))) ++bad_div;
        // round 6: exact_rcp (the declared sampler's lambda: 0.5 / (ma ma')) against the IEEE reciprocal, wherever both are normal
        if (fabsf(x) >= 7.8886091e-31f && fabsf(x) <= 1.2676506e30f && __float_as_int(exact_rcp(x)) != __float_as_int(ieee_div(1.0f, x))) ++bad_div;
    }
    if (bad_sqrt) atomicAdd(&mismatch[0], bad_sqrt);
    if (bad_div) atomicAdd(&mismatch[1], bad_div);
}

// log2_cr over an array (atmo_debug_log2_cr: compared with the oracle's copy of the function bit for bit)
__global__ __launch_bounds__(256) void atmo_log2_cr_kernel(const float *__restrict__ x, float *__restrict__ out, int n) {
    __shared__ f32x4 tab[LOG2CR_ROWS];
    log2_cr_table_fill(tab, threadIdx.x);   // (every wave of the block writes the same 32 rows)
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log2_cr(x[i], tab);
}
hipError_t launch_log2_cr(const float *x_dev, float *out_dev, int n, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_log2_cr_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x_dev, out_dev, n);
    return hipGetLastError();
}

hipError_t launch_selftest(uint32_t first_bits, uint32_t count, float c, float rc, unsigned int *mismatch_dev, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_selftest_kernel, dim3(2048), dim3(256), 0, stream, first_bits, count, c, rc, mismatch_dev);
    return hipGetLastError();
}

// ---- launchers -----------------------------------------------------------------------------------------
void render_tile_size(int split, int *tile_w, int *tile_h) {
    *tile_w = TILE_W;
    *tile_h = TILE_H / (split == 2 ? 2 : 1);
}
void render_grid(const RenderConsts &rc, int split, int *tiles_x, int *tiles_y) {
    const int th = TILE_H / (split == 2 ? 2 : 1);  // pixel rows per workgroup
    *tiles_x = (rc.x1 - rc.gx0 + TILE_W - 1) / TILE_W;  // (rc.gx0, rc.gy0): the grid's origin, set by the host before this call
    *tiles_y = (rc.y1 - rc.gy0 + th - 1) / th;
}

// Blocks of the launch in flight: 0 = the whole grid of the rect; n > 0 = a tile-list draw (atmo_render_tiles): n blocks, block b shades
// tile rc.tile_order[b] of that grid.  (A plain variable instead of a parameter threaded through the 80-case dispatch below; the host API
// is single-threaded per context and sets it around one launch_render call.)
static thread_local int g_launch_blocks = 0;

template <int FLAGS, int LSTEPS, int SPLIT>
static hipError_t launch_s(const RenderConsts &rc, hipStream_t stream) {
    int gx, gy;
    render_grid(rc, SPLIT, &gx, &gy);
    if (gx != rc.tiles_x) return hipErrorInvalidValue;
    dim3 grid(gx, gy);
    if (g_launch_blocks > 0) {
        if (rc.tile_order == nullptr) return hipErrorInvalidValue;
        grid = dim3(g_launch_blocks, 1);
    }
    if constexpr (render_sgpr_cap80(FLAGS))
        hipLaunchKernelGGL((atmo_render_kernel_s80<FLAGS, LSTEPS, SPLIT>), grid, dim3(TILE_W * TILE_H), 0, stream, rc);
    else
        hipLaunchKernelGGL((atmo_render_kernel<FLAGS, LSTEPS, SPLIT>), grid, dim3(TILE_W * TILE_H), 0, stream, rc);
    return hipGetLastError();
}

template <int FLAGS, int LSTEPS>
static hipError_t launch_t(const RenderConsts &rc, int split, hipStream_t stream) {
    return split == 2 ? launch_s<FLAGS, LSTEPS, 2>(rc, stream) : launch_s<FLAGS, LSTEPS, 1>(rc, stream);
}

// direct light mode: 8 light steps (BASELINE's "32 view x 8 light") has an unrolled instantiation
template <int FLAGS>
static hipError_t launch_direct(const RenderConsts &rc, int split, hipStream_t stream) {
    return rc.light_steps == 8 ? launch_t<FLAGS, 8>(rc, split, stream) : launch_t<FLAGS, 0>(rc, split, stream);
}

static hipError_t launch_render_grid(int flags, int split, const RenderConsts &rc, hipStream_t stream);
hipError_t launch_render(int flags, int split, const RenderConsts &rc, hipStream_t stream, int tile_list_blocks) {
    g_launch_blocks = tile_list_blocks;
    const hipError_t e = launch_render_grid(flags, split, rc, stream);
    g_launch_blocks = 0;
    return e;
}
static hipError_t launch_render_grid(int flags, int split, const RenderConsts &rc, hipStream_t stream) {
    switch (flags) {
    case 0: return launch_t<0, 0>(rc, split, stream);
    case KF_LIGHT_DIRECT: return launch_direct<KF_LIGHT_DIRECT>(rc, split, stream);
    case KF_LIGHT_DIRECT | KF_GEO:   // (one lane per ray; render_impl has filled rc.geo_rows)
        return rc.light_steps == 8 ? launch_s<KF_LIGHT_DIRECT | KF_GEO, 8, 1>(rc, stream) : launch_s<KF_LIGHT_DIRECT | KF_GEO, 0, 1>(rc, stream);
    case KF_CLOUDS: return launch_t<KF_CLOUDS, 0>(rc, split, stream);
    case KF_CLOUDS | KF_LIGHT_DIRECT: return launch_direct<KF_CLOUDS | KF_LIGHT_DIRECT>(rc, split, stream);
    case KF_CLOUDS | KF_CLOUD_LIGHT_RM: return launch_t<KF_CLOUDS | KF_CLOUD_LIGHT_RM, 0>(rc, split, stream);
    case KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT: return launch_direct<KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT>(rc, split, stream);
    case KF_LITE: return launch_t<KF_LITE, 0>(rc, split, stream);
    case KF_LITE | KF_CLOUDS: return launch_t<KF_LITE | KF_CLOUDS, 0>(rc, split, stream);
    // the v2 atmosphere in the reference's operation order (atmo_set_precision 2), one lane per ray, run-time light-step loop
    case KF_ATMO_REF: return launch_s<KF_ATMO_REF, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_LIGHT_DIRECT: return launch_s<KF_ATMO_REF | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_PRECISE | KF_CLOUDS: return launch_s<KF_ATMO_REF | KF_PRECISE | KF_CLOUDS, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT: return launch_s<KF_ATMO_REF | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM: return launch_s<KF_ATMO_REF | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT:
        return launch_s<KF_ATMO_REF | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    // ... under the declared cubemap sampler
    case KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS: return launch_s<KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM:
        return launch_s<KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT:
        return launch_s<KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    case KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT:
        return launch_s<KF_ATMO_REF | KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    // precise cloud density (atmo_set_precision 1, the default of the cloud variants)
    case KF_PRECISE | KF_CLOUDS: return launch_t<KF_PRECISE | KF_CLOUDS, 0>(rc, split, stream);
    case KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT: return launch_direct<KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT>(rc, split, stream);
    case KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM: return launch_t<KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM, 0>(rc, split, stream);
    case KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT:
        return launch_direct<KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT>(rc, split, stream);
    case KF_PRECISE | KF_LITE: return launch_t<KF_PRECISE | KF_LITE, 0>(rc, split, stream);
    case KF_PRECISE | KF_LITE | KF_CLOUDS: return launch_t<KF_PRECISE | KF_LITE | KF_CLOUDS, 0>(rc, split, stream);
    // implicit cubemap LOD (atmo_set_sampler_lod 1): precise cloud kernels, one lane per ray
    // (split == 2: the cloud march on two lanes per ray -- what the host draws a frame's heavy tiles with, bit-identical to the one-lane form)
    case KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS: return launch_t<KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS, 0>(rc, split, stream);
    case KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM: return launch_t<KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM, 0>(rc, split, stream);
    case KF_CUBE_LOD | KF_PRECISE | KF_LITE | KF_CLOUDS: return launch_s<KF_CUBE_LOD | KF_PRECISE | KF_LITE | KF_CLOUDS, 0, 1>(rc, stream);
    // ... and with the direct light march of the atmosphere (one lane per ray; the two-lanes-per-ray launch shape has no LOD form)
    case KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT:
        return rc.light_steps == 8 ? launch_s<KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT, 8, 1>(rc, stream)
                                   : launch_s<KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    case KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT:
        return rc.light_steps == 8 ? launch_s<KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT, 8, 1>(rc, stream)
                                   : launch_s<KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT, 0, 1>(rc, stream);
    default: break;
    }
    // more than 32 view steps (KF_VIEW_POS, set by the host): the fast v2 march with the reference's position accumulation; one lane per ray,
    // run-time light-step loop
    if (flags & KF_VIEW_POS) {
        switch (flags & ~KF_VIEW_POS) {
#define ATMO_VP_CASE(F) case (F): return launch_s<(F) | KF_VIEW_POS, 0, 1>(rc, stream);
            ATMO_VP_CASE(0)
            ATMO_VP_CASE(KF_LIGHT_DIRECT)
            ATMO_VP_CASE(KF_CLOUDS)
            ATMO_VP_CASE(KF_CLOUDS | KF_LIGHT_DIRECT)
            ATMO_VP_CASE(KF_CLOUDS | KF_CLOUD_LIGHT_RM)
            ATMO_VP_CASE(KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT)
            ATMO_VP_CASE(KF_PRECISE | KF_CLOUDS)
            ATMO_VP_CASE(KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT)
            ATMO_VP_CASE(KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM)
            ATMO_VP_CASE(KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT)
            ATMO_VP_CASE(KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS)
            ATMO_VP_CASE(KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_LIGHT_DIRECT)
            ATMO_VP_CASE(KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM)
            ATMO_VP_CASE(KF_CUBE_LOD | KF_PRECISE | KF_CLOUDS | KF_CLOUD_LIGHT_RM | KF_LIGHT_DIRECT)
#undef ATMO_VP_CASE
        default: break;
        }
    }
    return hipErrorInvalidValue;
}

const char *render_kernel_name(int flags, int light_steps, int split) {
    // demangled template name as rocprofv3 prints it: atmo_render_kernel<FLAGS, LSTEPS, SPLIT>
    static thread_local char name[64];
    const bool v2_precise = (flags & KF_ATMO_REF) != 0;  // its light march is a run-time loop
    const int lsteps = ((flags & KF_LIGHT_DIRECT) && light_steps == 8 && !v2_precise && !(flags & KF_VIEW_POS)) ? 8 : 0;
    snprintf(name, sizeof(name), "atmo_render_kernel%s<%d, %d, %d>", render_sgpr_cap80(flags) ? "_s80" : "", flags, lsteps, split == 2 ? 2 : 1);
    return name;
}

__global__ void atmo_lut_footprint_kernel(const float *__restrict__ apron, int w, int h, float *__restrict__ out4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i > w || j > h) return;
    const int st = w + 2;
    const float *p = apron + (size_t)j * st + i;
    float *o = out4 + ((size_t)j * (w + 1) + i) * 4;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[st]; o[3] = p[st + 1];
}
hipError_t launch_lut_footprints(const float *apron, int w, int h, float *out4, hipStream_t stream) {
    hipLaunchKernelGGL(atmo_lut_footprint_kernel, dim3((w + 1 + 255) / 256, h + 1), dim3(256), 0, stream, apron, w, h, out4);
    return hipGetLastError();
}

hipError_t launch_bake(const BakeConsts &bc, hipStream_t stream) {
    dim3 grid((bc.w + 15) / 16, (bc.h + 15) / 16);
    hipLaunchKernelGGL(atmo_bake_kernel, grid, dim3(256), 0, stream, bc);
    return hipGetLastError();
}

}  // namespace atmo
