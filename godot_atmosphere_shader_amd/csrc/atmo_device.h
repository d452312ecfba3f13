// atmo_device.h -- structures shared by the host API (atmo_api.hip) and the gfx950 kernels
// (atmo_kernels.hip).  Not part of the public C ABI (that is include/atmo.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atmo {

// Everything a render launch needs that is constant over the draw.  Passed BY VALUE as the kernel
// argument, so every field lands in SGPRs through scalar loads of the kernarg segment.
// Values marked [host] are per-frame expressions of the reference shader that do not depend on the
// pixel; the API evaluates them once in fp32, in the reference's operation order.
constexpr int GEO_MAX_ROWS = 272;   // tile rows the geometric order covers: 3840 x 2160 in 8-pixel rows (taller grids: the learnt order)
constexpr int GEO_MAX_HINTS = 256;  // ... and 65 536 tiles, one hint per 256 blocks
struct RenderConsts {
    // --- atmosphere_fragment prologue (shaders/include/planet_atmosphere_main.gdshaderinc:128-169)
    float inv_p[16];
    float inv_v[16];
    float cam_pos_world[3];  // [host] inv_v * (0,0,0,1)                      main:136
    float vw, vh;            // VIEWPORT_SIZE as floats
    int32_t w, h;            // VIEWPORT_SIZE
    int32_t x0, y0, x1, y1;  // rect shaded by this launch
    float center[3];         // v_planet_center_viewspace
    float sun_dir[3];        // [host] normalize(sun_center_vs - planet_center_vs)   main:164
    float planet_radius, atmosphere_height;
    float atmosphere_radius; // [host] R + H                                   main:144
    float density;           // u_density
    float sphere_depth_factor;
    // --- compute_atmosphere_v2 (shaders/include/atmosphere_funcs_v2.gdshaderinc:32-101)
    float coeff[3];          // [host] pow4(400/lambda) * strength             v2:47-51
    float ambient[3], modulate[3];
    int32_t view_steps;
    int32_t light_steps;     // direct light mode only
    // --- compute_atmosphere, v1 "lite" (shaders/include/atmosphere_funcs_v1.gdshaderinc:8-12,48-63)
    float day0[3], day1[3], night0[3], night1[3];
    float day_night_transition_scale;
    // --- render_clouds / raymarch_cloud (shaders/include/cloud_funcs.gdshaderinc:175-324)
    float clouds_bottom, clouds_top;  // [host] R + u_cloud_{bottom,top} * H   clouds:260-261
    float cloud_thickness;            // [host] top - bottom
    float inv_cloud_thickness;        // [host] RN(1 / thickness), for exact_div_uniform
    float layer_r2_lo, layer_r2_hi;   // [host] (bottom (1 - 1e-6))^2, (top (1 + 1e-6))^2: |p|^2 outside [lo, hi] => surely outside the cloud layer
    float cloud_density_scale, cloud_blend, coverage_bias, shape_factor, shape_scale;
    int32_t shape_invert;             // u_cloud_shape_invert == 1.0           clouds:57
    float shape_lo01, shape_hi01;     // [host] bounds of (shape - 0.1) over every filtered texel value: coverage-first early outs
    float cov_rot[4];                 // mat2 column-major
    float view_to_model[16];          // [host] u_world_to_model_matrix * inv_view   clouds:285
    float origin_model[3];            // [host] (view_to_model * (0,0,0,1)).xyz      clouds:286
    float sun_dir_model[3];           // [host] (view_to_model * (sun_dir,0)).xyz    clouds:288
    float max_d;                      // [host] march-distance cap             clouds:186-202
    float inv_cloud_steps;            // [host] 1/float(steps)                 clouds:206
    int32_t cloud_steps;
    float rm_offset[6];               // [host] float(i) * step_len_i, step_len_i = reach/6 * 1.2^i   clouds:108,114-115,129,143
    float rm_weight[6];               // [host] step_len_i * density_scale     clouds:138
    float rm_tap[6][3];               // [host] rm_offset[i] * sun_dir_model: the uniform product of clouds:129, rounded once on the host
    // --- textures (device memory owned by the context)
    const float *lut;        // u_optical_depth_texture: (lut_h+2) rows of (lut_w+2), clamp-to-edge apron
    int32_t lut_w, lut_h;
    const float *lut4;       // what the kernels sample: (lut_h+1) rows of (lut_w+1) footprints, 4 floats = the 2x2 apron texels at (i,j),(i+1,j),(i,j+1),(i+1,j+1)
    const uint8_t *blue;     // u_blue_noise_texture 256x256
    const uint32_t *shape;   // u_cloud_shape_texture: n^3 xy-footprint words (repeat wrap baked in)
    int32_t shape_n;
    int32_t shape_log2n;     // [host] log2(shape_n) when it is a power of two, else -1 (shape_addr)
    const uint32_t *cube;    // u_cloud_coverage_cubemap: 6 x (n+1)^2 footprint words of the seamless-apron faces; null => 1.0
    int32_t cube_n;
    int32_t cube_levels;             // mip levels bound; level l = (cube_n >> l)-sided faces, footprints at cube + cube_level_off[l]
    const uint32_t *cube_level_off;  // device array of 16 element offsets
    uint32_t cube_bytes;             // bytes of the packed footprint chain (buffer-load bound)
    int32_t cube_lod_fast;           // 1: cube_n is a power of two <= 1024 (closed-form level offsets, fp32 addressing)
    float lod0_inv_c;                // [host] 1 / the level-0 certificate's constant C (cube_lod_level0_certain); +inf: never certain
    float lod0_last, lod0_drift;     // [host] cloud_steps - 1 and (cloud_steps + 1) * sqrt(3) 2^-23 (quad_march_spread2)
    // --- per-pixel streams
    const float *depth;      // h rows of w
    float4 *out;             // plain: (y1-y0) rows of (x1-x0); composite: the h x w scene colour buffer, blended in place
    int32_t out_pitch;       // pixels per output row
    int32_t out_x0, out_y0;  // viewport pixel that maps to out[0]
    int32_t composite;       // 1 => straight-alpha "mix" blend over the existing contents, discarded pixels untouched
    // --- launch order (atmo_set_tile_feedback)
    int32_t tiles_x;                // tiles per row of the launch grid
    const uint32_t *tile_order;     // null => tile = linear block index; else the tile each block shades (heaviest first)
    uint32_t *tile_cost;            // null => no feedback; else per-tile max wave duration in shader cycles (atomicMax)
#ifdef ATMO_WAVE_TRACE  // diagnostic build (tools/wave_timeline.py): 4 x uint64 per wave = start, end (100 MHz), HW_ID, XCC_ID
    unsigned long long *wave_trace;
#endif
    // --- exact short division of the pixel coordinates (pixel_coord in atmo_kernels.hip)
    float rcp_vw, rcp_vh;           // [host] RN(1 / vw), RN(1 / vh)
    // --- sure-miss test in front of the exact prologue (shade_pixel)
    const float *cube_f4;           // level 0 of `cube` with every footprint's four texels as exact byte / 255 floats (16 B), or null
    const float *shape_f4;          // the same for `shape`
    float miss_k;                   // [host] (|c|^2 - R_atm^2) (1 - 1e-3)^2 when the test is usable, else 0
    int32_t gx0, gy0;               // [host] viewport pixel of the launch grid's first tile: (x0, y0), rounded down to even for the declared-sampler kernels
    int32_t store_discards;         // 1: a discarded fragment stores (0,0,0,0); 0: it stores nothing (composite, or atmo_set_target_cleared)
    int32_t cost_rows_halved;       // 1: this launch draws HEAVY tiles of a one-lane grid as pairs of half-height tiles (render_impl's split path): tile_cost is indexed by the one-lane tile
    // --- the GEOMETRIC tile order of the cloudless variants (round 6; geo_tile in atmo_kernels.hip, geo_order_fill in atmo_api.hip).  Seen from outside the
    // atmosphere shell the tiles whose rays can hit it form ONE run of columns per tile row -- the shell's silhouette is a conic --, so "the tiles that shade
    // first, row-major, the all-miss tiles behind them" is a closed-form map from the block index to the tile: no order buffer, no recording draw, no sort,
    // nothing learnt from earlier frames and therefore nothing that lags behind a moving camera.  geo_rows = 0: off (tile = block index, or tile_order).
    int32_t geo_rows;                           // [host] tile rows of the launch grid, or 0
    uint16_t geo_prefix[GEO_MAX_ROWS + 1];      // [host] geo_prefix[r] = tiles that can shade in the rows above r; [geo_rows] = all of them
    uint8_t geo_first[GEO_MAX_ROWS];            // [host] first column of row r's run (its length: geo_prefix[r + 1] - geo_prefix[r])
    // where to start looking: geo_hint[0][v] = the last row r with geo_prefix[r] <= 256 v (blocks of the first part), geo_hint[1][v] = the last row r with
    // r tiles_x - geo_prefix[r] <= 256 v (blocks of the second part) -- the row of block b then lies in [hint[b >> 8], hint[(b >> 8) + 1]]: one or two steps of a
    // binary search instead of nine (every step is a dependent scalar load in every wave's preamble)
    uint16_t geo_hint[2][GEO_MAX_HINTS + 2];
};

struct BakeConsts {
    float planet_radius, atmosphere_height, density;
    int32_t w, h, steps;
    float *out;  // (h+2) x (w+2), apron written by the edge texels' lanes
};

struct NoiseCubemapConsts {
    int32_t resolution;
    uint32_t seed;
    float frequency, gain;
    int32_t octaves;
    float scale[3];
    uint8_t *out;  // 6 faces of resolution^2 bytes
};

// kernel launchers (atmo_kernels.hip)
enum KernelFlags : int { KF_CLOUDS = 1, KF_CLOUD_LIGHT_RM = 2, KF_LIGHT_DIRECT = 4, KF_LITE = 8, KF_PRECISE = 16, KF_CUBE_LOD = 32,
                          KF_ATMO_REF = 64 /* the v2 atmosphere march in the reference's operation order (atmo_set_precision 2) */,
                          KF_VIEW_POS = 128 /* view_steps > 32: the fast v2 march accumulates the view-space position like the reference (march_atmosphere<VIEWPOS>) */,
                          KF_GEO = 256 /* the block -> tile map is the geometric order's closed form (RenderConsts::geo_rows): a twin of the plain direct-light kernel, so that
                                          the draws that do not use it keep their preamble to the byte (the lookup compiled in cost a still camera 0.6 %) */ };

hipError_t launch_render(int flags, int split, const RenderConsts &rc, hipStream_t stream, int tile_list_blocks = 0);  // > 0: rc.tile_order lists that many tiles of the rect's grid
hipError_t launch_bake(const BakeConsts &bc, hipStream_t stream);
hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, int tiles_x, int tiles_y, int rx, int ry, uint32_t *tmp1, uint32_t *tmp2,
                             uint32_t *scratch, hipStream_t stream, uint32_t *order2 = nullptr, uint32_t *class_totals = nullptr);
#ifndef ATMO_ORDER_CLASSES   // 64 since round 6 (was 32, half octaves): clouds_high 1920x1080 -3.7 %, its level-0 form -5.1 %, nothing else beyond +-1 % (profiles/round6/ab_order_classes.txt)
#define ATMO_ORDER_CLASSES 64
#endif
constexpr int TILE_ORDER_CLASSES = ATMO_ORDER_CLASSES;   // cost classes of the sort: 16 octaves of the wave duration (2^8 .. 2^24 cycles) in TILE_ORDER_PER_OCTAVE equal
constexpr int TILE_ORDER_PER_OCTAVE = TILE_ORDER_CLASSES / 16;   // parts each (by the duration's leading mantissa bits), class 0 the heaviest (tile_cost_class)
static_assert(TILE_ORDER_CLASSES == 32 || TILE_ORDER_CLASSES == 64, "32 (half octaves) or 64 (quarter octaves): the sort's per-class state is one lane of a wave");
size_t tile_order_scratch_bytes();
hipError_t launch_tile_list_bound(const uint32_t *in, uint32_t *out, int n, uint32_t tiles_n, uint32_t sentinel, hipStream_t stream,
                                  uint32_t *out2 = nullptr, int n_heavy = 0, int tiles_x = 1, uint32_t sentinel2 = 0);  // atmo_render_tiles[_split]
hipError_t launch_layout_lut(const float *lut, int w, int h, float *out, hipStream_t stream);
hipError_t launch_lut_footprints(const float *apron, int w, int h, float *out4, hipStream_t stream);
hipError_t launch_layout_shape(const uint8_t *t, int n, uint32_t *out, hipStream_t stream);
hipError_t launch_layout_cube(const uint8_t *faces, int n, uint32_t *out, hipStream_t stream);
hipError_t launch_footprints_f4(const uint32_t *words, size_t n_words, float *out4, hipStream_t stream);
hipError_t launch_cube_mip(const uint8_t *level, int n, uint8_t *next, hipStream_t stream);
void render_grid(const RenderConsts &rc, int split, int *tiles_x, int *tiles_y);
void render_tile_size(int split, int *tile_w, int *tile_h);  // pixels per workgroup tile of a launch with `split` lanes per ray
hipError_t launch_noise_cubemap(const NoiseCubemapConsts &nc, hipStream_t stream);
const char *render_kernel_name(int flags, int light_steps, int split);
hipError_t launch_log2_cr(const float *x_dev, float *out_dev, int n, hipStream_t stream);
hipError_t launch_light_probe(const float *pos, const float *dir, int n, float planet_radius, float atmosphere_height, float density,
                              int light_steps, float *out, hipStream_t stream);
hipError_t launch_selftest(uint32_t first_bits, uint32_t count, float c, float rc, unsigned int *mismatch_dev, hipStream_t stream);

}  // namespace atmo
