// atmo_api.hip -- host side of the C ABI declared in include/atmo.h.
//
// Holds the uniform table (the reference's `shader_params` names), owns the device copies of the four
// textures, evaluates the per-frame (pixel-independent) expressions of the shader once per launch in
// fp32 in the reference's operation order (this file is built with -ffp-contract=off, so nothing is
// fused), and enqueues the gfx950 kernels of atmo_kernels.hip.  There is no CPU fallback: without a
// HIP device every compute entry point fails with ATMO_E_NO_DEVICE / ATMO_E_HIP.
#include "../../include/atmo.h"
#include "../../include/atmo_debug.h"
#include "atmo_device.h"
#include "atmo_layout.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
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

// Largest textures that get a float copy of their footprints (tools/texsize_probe.py, profiles/round3/ab_float_footprints.txt).  Of the cubemap
// a frame touches only the visible part: the copy wins up to the 1024^2 faces the fp32-addressed sampler handles.  The shape volume repeats, so
// every XCD touches all of it in every frame, and its copy has to fit into an XCD's 4 MB L2 beside everything else: at 32^3 (0.5 MB) it is worth
// 4 %, at 64^3 (4 MB) it was worth 1 % of round 3's kernel time for 27 MB more fabric traffic per 1920x1080 frame -- and is worth 2-5 % of round 6's
// (the kernels issue 1.4x fewer VALU instructions since: the texture path is what they wait for; profiles/round6/ab_gathers.txt) --, at 128^3 (34 MB)
// it COSTS 50 %.
constexpr int F4_MAX_CUBE_N = 1024, F4_MAX_SHAPE_N = 64;   // (64 since round 6: profiles/round6/ab_gathers.txt; ATMO_F4_SHAPE_MAX overrides for A/B)

thread_local std::string g_create_error = "";

}  // namespace

struct AtmoContext {
    int device = 0;
    int variant = 0;
    int flags = 0;
    int view_steps = 8, cloud_steps = 0, light_steps = 0;
    Params p;
    DeviceBuffer lut, blue, shape, cube;
    DeviceBuffer cube_f4, shape_f4;  // float copies of the level-0 cubemap / shape footprints (16 B each): what the precise samplers read
    int f4_max_shape_n = F4_MAX_SHAPE_N;   // ATMO_F4_SHAPE_MAX (A/B): largest volume that gets the float copy
    DeviceBuffer lut4;  // footprint copy of `lut` (derived on the device whenever `lut` is written): what the kernels sample
    int lut_w = 0, lut_h = 0, shape_n = 0, cube_n = 0;
    int cube_levels = 0;                 // mip levels bound (footprint arrays packed level after level in `cube`)
    DeviceBuffer cube_level_off;         // element offset of every level's footprint array (uint32 x 16, device)
    uint32_t cube_level_off_host[16] = {0};
    int sampler_lod = -1;                // atmo_set_sampler_lod: -1 = as the reference declares it (implicit LOD when a mip chain is bound), 0 = LOD 0, 1 = implicit LOD required
    DeviceBuffer staging;                // raw texels on their way into a re-layout kernel (grow-only)
    hipStream_t staging_stream = nullptr;
    bool staging_used = false;
    hipEvent_t tex_event = nullptr;      // recorded after the last texture update on tex_stream
    hipStream_t tex_stream = nullptr;
    bool tex_pending = false;
    unsigned tex_version = 0;            // bumped by every texture update
    hipStream_t tex_waited_stream = nullptr;  // the last other stream that was ordered behind tex_event ...
    unsigned tex_waited_version = 0;          // ... and for which update (one wait per stream and update, not per launch)
    int host_double_precision = 0;  // DOUBLE_PRECISION (main:25,118-125)
    int target_cleared = 0;         // atmo_set_target_cleared: discarded fragments write nothing
    int lane_split = 0;             // 0 = choose per launch by size, 1 = one lane per ray, 2 = two lanes per ray
    int last_split = 1;             // what the most recent launch used (atmo_kernel_name)
    int last_flags = -1;
    // tile order with cost feedback (atmo_set_tile_feedback): -1 = default (on), 0 off, 1 on
    int tile_feedback = -1;
    // One feedback state per (launch grid, lanes per ray, draw stream): a context that alternates between a few rects or
    // streams (split screen, stereo eyes, uneven row bands) keeps one state for each instead of starting over -- and
    // synchronising -- at every change.  Buffers are grow-only; a slot is recycled (least recently used) only when a
    // fifth key appears, and a context that keeps producing new keys runs out of recycling budget (8, one regained
    // every 256 draws) and draws those keys in row-major order: nothing waits, the resident states keep working.
    struct FeedbackState {
        bool used = false;
        int tiles_x = 0, tiles_y = 0, split = 0;
        hipStream_t draw_stream = nullptr;
        DeviceBuffer cost, order[2];   // one cost buffer; two orders: the one in use and the one being sorted
        DeviceBuffer dil[2];           // scratch of the cost-map dilation (moving camera)
        AtmoFrame prev_frame;          // the previous draw's camera: screen-space motion estimate
        bool have_prev = false;
        float motion_px = 0.0f;        // pixels per frame the picture's features move (peak-held estimate)
        float sil_px[2] = {0.0f, 0.0f};  // ... and the planet's silhouette alone, per screen axis (the in-stream sort's dilation window)
        float order_reach_px[2] = {0.0f, 0.0f};  // how far features may have moved for order[k] to stay conservative
        unsigned order_born[2] = {0, 0};         // n of the recording draw order[k] was sorted from
        // in-stream mode (moving camera, long frames): the sort runs on the draw stream right behind every draw, so the
        // next draw is ordered by THIS frame's costs; no host queries, one frame of lag
        DeviceBuffer is_order, is_scratch;
        unsigned is_last_n = ~0u;                // n of the draw behind which the last in-stream sort was enqueued
        // heavy tiles on two lanes per ray (round 5): every order also exists for the launch grid of the lane-split kernels (tiles half as
        // high: two entries per tile), and the sort leaves the number of tiles per cost class in pinned host memory
        DeviceBuffer order2[2], is_order2;
        uint32_t *class_totals = nullptr;        // pinned: 3 x 32 counters -- order[0], order[1], is_order
        unsigned n = 0;                // draws of this key so far
        unsigned last_record = 0;      // n of the last draw that recorded costs
        int active = -1;               // order[active] is complete and in use; -1: row-major order
        int write = 0;                 // order[write] is what the next / pending sort writes
        bool pending = false;          // a sort is in flight on fb_stream
        hipEvent_t ev_draw = nullptr, ev_order[2] = {nullptr, nullptr};
        unsigned long long last_use = 0;
        bool dirty = false;            // released by atmo_set_tile_feedback with work possibly in flight: quiesced when the slot is taken again
    };
    // the geometric tile order of the cloudless variants (RenderConsts::geo_rows; geo_order_fill): the last frame's table, reused while the camera stands still
    struct GeoCache {
        float key[26] = {0};
        int rows = -1;                      // -1: nothing cached; 0: this camera has no usable table
        uint16_t prefix[atmo::GEO_MAX_ROWS + 1];
        uint8_t first[atmo::GEO_MAX_ROWS];
        uint16_t hint[2][atmo::GEO_MAX_HINTS + 2];
    } geo;
    int geo_order = 1;                                 // ATMO_GEO_ORDER=0 (A/B): the learnt order for these variants, as in rounds 2-5
    unsigned geo_draws = 0;
    static constexpr int FB_SLOTS = 4;
    FeedbackState fb[FB_SLOTS];
    unsigned long long fb_clock = 0;
    int fb_budget = 8;                                 // slot recyclings allowed right now; one comes back every 256 draws
    unsigned fb_ordered_draws = 0, fb_recycled = 0, fb_sorts = 0;  // atmo_get_feedback_stats
    hipStream_t fb_stream = nullptr;                   // the sort kernels run here, beside the draws (high priority)
    hipStream_t split_stream = nullptr;                // the heavy tiles of a frame, on two lanes per ray, run here beside the rest of the draw
    int heavy_split = 1;                               // ATMO_HEAVY_SPLIT=0 (A/B): every tile with one lane per ray; 2 (tests): also the kernel without raymarched light, no trigger
    float heavy_split_trigger = 2.0f;                  // ATMO_HEAVY_SPLIT_TRIGGER: only when the heaviest class lives longer than this x the draw's estimated duration ...
    float heavy_split_trigger_moving = 2.0f;           // ATMO_HEAVY_SPLIT_TRIGGER_MOVING: the trigger while the order comes from the in-stream sort (a moving camera)
    float heavy_split_ratio = 0.3f;                    // ATMO_HEAVY_SPLIT_RATIO: ... the tiles whose longest wave lives longer than this x that estimate are heavy
    unsigned split_draws = 0, split_tiles_last = 0;    // atmo_get_split_stats
    DeviceBuffer fb_scratch;                           // the sort's block histograms (sorts are serialised on fb_stream)
    unsigned fb_period = 8;                            // every fb_period-th draw of a key records costs
    // A/B overrides read ONCE, in atmo_create (tools/ab_feedback.sh, tools/ab_bench.sh): ATMO_TILE_FEEDBACK=0/1,
    // ATMO_TILE_FEEDBACK_PERIOD=n, ATMO_LANE_SPLIT=1/2
    int env_feedback = -1, env_split = 0;
    float env_reach_scale = 1.0f;                      // ATMO_FB_REACH_SCALE: multiplies the predicted reach (A/B)
    unsigned moving_period = 2;                        // ATMO_FB_MOVING_PERIOD: recording period while the camera moves
    int f4_footprints = 3;                             // ATMO_F4=0..3 (A/B): bit 0 = float copy of the cubemap footprints, bit 1 = of the shape volume's
    int env_lod0_cert = 1;                             // ATMO_LOD0_CERT=0 (A/B): the declared sampler's level-0 certificate off
    int instream = 1;                                  // ATMO_FB_INSTREAM=0: never sort on the draw stream (A/B)
    int fb_axis_windows = 1;                           // ATMO_FB_AXIS_WINDOWS=0: the in-stream sort dilates by motion_px in both axes, as in round 3 (A/B)
    uint32_t *measure_cost = nullptr;                  // atmo_measure_tile_costs: the next draw records here
    // the streams draws of this context have been enqueued on since the last texture update waited for them: an update arriving on
    // stream s is stream-ordered behind the draws on s and has to wait, on the host, for those on every OTHER stream of this set
    // A remembered stream OTHER than the context's home stream (= the stream of its most recent texture update: where a single-stream host
    // does everything, and stream order is all it needs) carries a MARKER EVENT recorded behind its most recent draw (round 5): whoever
    // must be ordered behind those draws waits for the event on the device -- the stream's handle is never touched again, so a stream the
    // caller has destroyed since is harmless (HIP dereferences stale stream handles: hipEventRecord on one crashes).  Not on the home
    // stream: a marker costs 3-4 us per draw (profiles/round5/ab_draw_events.txt: headline -4 %, shipped8 -17 %).
    struct DrawStream { hipStream_t stream = nullptr; hipEvent_t last_draw = nullptr; bool recorded = false; unsigned long long last_use = 0; };
    unsigned long long draw_stream_clock = 0;
    std::vector<DrawStream> draw_streams;              // (at most 8 remembered; beyond that `draw_streams_many` stands for "some other stream")
    bool draw_streams_many = false;
    int draw_events = 1;                               // ATMO_DRAW_EVENTS=0 (A/B): no markers, the waits fall back to hipDeviceSynchronize; 2: markers on the home stream too
    hipEvent_t xs_event[8] = {nullptr};                // order_after_stream: markers on the context's OWN streams (sort stream, staging stream)
    unsigned xs_next = 0;
    unsigned device_syncs = 0;                         // how often a call of this context fell back to hipDeviceSynchronize (atmo_get_host_wait_stats)
    DeviceBuffer measure_buf;                          // atmo_measure_tile_costs: the tile costs on their way to the host (grow-only)
    // atmo_render_tiles: the caller's tile list, bounded on the device before the draw reads it (one grow-only copy per draw stream, up to 4)
    struct TileListBuf { hipStream_t stream = nullptr; DeviceBuffer buf; bool used = false; unsigned long long last_use = 0; } tile_lists[4];
    unsigned long long tile_list_clock = 0;
#ifdef ATMO_WAVE_TRACE
    DeviceBuffer wave_trace;                           // diagnostic build only
    size_t wave_trace_waves = 0;
#endif
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

// ---- host-side statement of the device texture layouts (atmo_host_layout_*: the CPU checker) ----------------------
// The same per-element functions (atmo_layout.h) the re-layout kernels evaluate on the GPU.
void build_cube_footprints(const uint8_t *faces, int n, std::vector<uint32_t> &out) {
    const int fs = n + 1;
    out.assign((size_t)6 * fs * fs, 0u);
    for (int f = 0; f < 6; ++f)
        for (int j = 0; j < fs; ++j)
            for (int i = 0; i < fs; ++i) out[((size_t)f * fs + j) * fs + i] = atmo::cube_footprint_word(faces, n, f, i, j);
}

void build_shape_footprints(const uint8_t *t, int n, std::vector<uint32_t> &out) {
    out.assign((size_t)n * n * n, 0u);
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) out[((size_t)k * n + j) * n + i] = atmo::shape_footprint_word(t, n, i, j, k);
}

void build_lut_apron(const float *lut, int w, int h, std::vector<float> &out) {
    const int st = w + 2;
    out.assign((size_t)st * (h + 2), 0.0f);
    for (int j = 0; j < h + 2; ++j)
        for (int i = 0; i < st; ++i) out[(size_t)j * st + i] = atmo::lut_apron_value(lut, w, h, i, j);
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
    rc.rcp_vw = 1.0f / rc.vw;  // pixel_coord(): exact for every viewport size atmo_render accepts (tools/uv_division.c)
    rc.rcp_vh = 1.0f / rc.vh;
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
    {   // sure-outside bounds on |p|^2 for the march (atmo_kernels.hip, ATMO_SURE_OUTSIDE): 2e-6 relative is ten times the rounding of the
        // unfused |p|^2 sum and of the root; below lo the exact chain gives r < bottom, hr < 0, above hi r > top, hr >= 1: hc = 0 either way
        const double lo = (double)rc.clouds_bottom * (double)rc.clouds_bottom * (1.0 - 2e-6);
        const double hi = (double)rc.clouds_top * (double)rc.clouds_top * (1.0 + 2e-6);
        rc.layer_r2_lo = (float)lo;
        rc.layer_r2_hi = (float)hi;
        if (!(rc.clouds_bottom > 0.0f)) rc.layer_r2_lo = 0.0f;  // a bottom shell of radius <= 0: |p| < bottom cannot be told from |p|^2
        if (!(rc.cloud_thickness > 0.0f) || !(rc.clouds_top > 0.0f) || !std::isfinite(rc.layer_r2_hi)) { rc.layer_r2_lo = 0.0f; rc.layer_r2_hi = INFINITY; }  // odd layers: never skip
    }
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
            for (int c = 0; c < 3; ++c) rc.rm_tap[i][c] = rc.rm_offset[i] * rc.sun_dir_model[c];
            step_len *= 1.2f;
        }
    }
    rc.lut = (const float *)ctx->lut.ptr; rc.lut_w = ctx->lut_w; rc.lut_h = ctx->lut_h;
    rc.lut4 = (const float *)ctx->lut4.ptr;
    rc.blue = (const uint8_t *)ctx->blue.ptr;
    rc.shape = (const uint32_t *)ctx->shape.ptr; rc.shape_n = ctx->shape_n;
    rc.shape_log2n = -1;
    for (int b = 0; b < 10; ++b) if (ctx->shape_n == (1 << b)) rc.shape_log2n = b;
    rc.cube = (const uint32_t *)ctx->cube.ptr; rc.cube_n = ctx->cube_n;
    rc.cube_levels = ctx->cube_levels;
    rc.cube_level_off = (const uint32_t *)ctx->cube_level_off.ptr;
    rc.cube_bytes = (uint32_t)ctx->cube.bytes;
    rc.cube_f4 = (ctx->f4_footprints & 1) ? (const float *)ctx->cube_f4.ptr : nullptr;
    rc.shape_f4 = (ctx->f4_footprints & 2) ? (const float *)ctx->shape_f4.ptr : nullptr;
    rc.cube_lod_fast = (ctx->cube_n >= 1 && ctx->cube_n <= 1024 && (ctx->cube_n & (ctx->cube_n - 1)) == 0) ? 1 : 0;
    {   // the level-0 certificate of the declared sampler (atmo_kernels.hip: cube_lod_level0_certain):  w E <= C ma^2  =>  lambda = 0.
        //   C = 0.97 * 4 (1 - 2/n)^2 / (n^2 max(sigma, 1)^2),  sigma = the largest singular value of u_cloud_coverage_rotation (1 for the rotation the node
        // builds from one angle; the bound holds for any linear map, the certificate is offered while both singular values lie in [0.5, 2] -- the
        // rounding budget below is a rotation's times sigma_max / sigma_min <= 4 then).  The 0.97 (rho^2 <= 0.97 proves lambda = 0) pays for what the kernel's E leaves out: the rounding of the tap
        // offsets and of the rotation, <= 8 ulp(|p|) = 1.6e-6 ma against a distance threshold >= 1.1e-3 ma (n = 1024, w = 3): 0.3 % of rho^2; the
        // 1.001 factors on E and the fp32 evaluation of the test and of the kernel's own rho^2: < 0.3 % more (the march's drift is in E itself).
        // (How much a tighter bound would buy, measured with the factor set beyond 1, i.e. NOT rigorous: 1.5 -> clouds_high -6 %, the limb frame
        //  -10 %; 0.9 -> 0.97 itself: -1 %.)
        rc.lod0_inv_c = std::numeric_limits<float>::infinity();
        rc.lod0_last = (float)(rc.cloud_steps - 1);
        rc.lod0_drift = (float)((rc.cloud_steps + 1) * 2.07e-7);
        const double n = (double)ctx->cube_n;
        const double a = rc.cov_rot[0], b = rc.cov_rot[2], c = rc.cov_rot[1], d = rc.cov_rot[3];  // column-major: rows (a b), (c d)
        const double fro = a * a + b * b + c * c + d * d, det = a * d - b * c;
        const double disc = fro * fro - 4.0 * det * det;
        const double sigma = std::sqrt(0.5 * (fro + std::sqrt(disc > 0.0 ? disc : 0.0))) * (1.0 + 1e-6);  // largest singular value of a 2 x 2 matrix
        const double sigma_min = sigma > 0.0 ? std::fabs(det) / sigma : 0.0;   // ... and the smallest: the products of a nearly singular map cancel
        // The matrix acts on (x, z) only (clouds:43): the y component of a partner difference passes through UNSCALED, so the map that takes
        // the unrotated difference to the cube direction's has norm max(sigma, 1), not sigma (a contracting matrix such as 0.6 I
        // had certified samples whose true rho^2 was 2.25).
        const double sigma_eff = sigma > 1.0 ? sigma : 1.0;
        if (rc.cube_lod_fast && ctx->cube_n >= 4 && sigma <= 2.0 && sigma_min >= 0.5) {
            rc.lod0_inv_c = (float)(1.0 / (0.97 * 4.0 * (1.0 - 2.0 / n) * (1.0 - 2.0 / n) / (n * n * sigma_eff * sigma_eff)));  // the kernels carry E / C
        }
        if (ctx->env_lod0_cert == 0) rc.lod0_inv_c = std::numeric_limits<float>::infinity();  // ATMO_LOD0_CERT=0 (A/B): every sample takes the derivative path
    }
    {   // sure-miss test (shade_pixel): usable when the view-ray direction does not depend on the depth sample (x, y, z rows of
        // inv_projection have no depth column: every perspective and orthographic-free Godot camera) and the camera is well
        // outside the shell, so that the 0.2 % margin dwarfs fp32 rounding of h = R^2 - |c|^2 + (c.d)^2 (a few 1e-7 |c|^2)
        const float *Pm = f->inv_projection_matrix;
        const double cc = (double)rc.center[0] * rc.center[0] + (double)rc.center[1] * rc.center[1] + (double)rc.center[2] * rc.center[2];
        const double k = cc - (double)rc.atmosphere_radius * rc.atmosphere_radius;
        const bool depth_free = Pm[8] == 0.0f && Pm[9] == 0.0f && Pm[10] == 0.0f;
        rc.miss_k = (depth_free && k > 0.01 * cc && std::isfinite(k)) ? (float)(k * (1.0 - 1e-3) * (1.0 - 1e-3)) : 0.0f;
    }
    rc.depth = depth;
    rc.out = (float4 *)rgba;
    rc.out_pitch = f->x1 - f->x0;
    rc.out_x0 = f->x0;
    rc.out_y0 = f->y0;
    rc.composite = 0;
    rc.store_discards = ctx->target_cleared ? 0 : 1;
    rc.gx0 = f->x0;
    rc.gy0 = f->y0;
}

// Which coverage-cubemap sampler a draw of this context uses (atmo_set_sampler_lod): 1 = the implicit LOD of the linear-mipmap sampler the
// reference declares (cloud_funcs.gdshaderinc:15,45; needs a bound mip chain and the precise cloud kernels), 0 = level 0 only.
// Mode -1 (default) picks 1 whenever it is available; mode 1 demands it: *why_not says what is missing.
int resolve_sampler_lod(const AtmoContext *ctx, const char **why_not) {
    if (why_not) *why_not = nullptr;
    if (ctx->sampler_lod == 0 || !(ctx->flags & atmo::KF_CLOUDS) || !ctx->cube.ptr || ctx->cube_levels <= 1) return 0;  // nothing to choose a level from
    if (!(ctx->flags & atmo::KF_PRECISE)) {
        if (why_not) *why_not = "atmo_render: the implicit cubemap LOD (atmo_set_sampler_lod 1) needs the precise cloud mode (atmo_set_precision 1)";
        return 0;
    }
    return 1;
}

// How many tiles at the head of a cost-sorted order are HEAVY: tiles whose longest wavefront lives longer than `ratio` x the draw itself.
// A cloud frame at 1920x1080 is as long as its heaviest wavefront (all 64 rays in dense cloud: 0.36 of the 0.42 ms of clouds_high_rm; a
// wave issues one instruction every ~5 cycles however empty the SIMD is) -- those tiles are drawn with two lanes per ray, which halves
// exactly that, for 13-29 % more work on them; at 3840x2160 the same waves are a quarter of the draw and nothing is split.
//   totals[k]: tiles in cost class k (half octaves of the wave duration in shader cycles, 0 = heaviest: tile_cost_class);
//   the draw's duration is estimated from the same numbers: sum of wave lifetimes / resident waves (2 waves per tile; SIMDs x waves per SIMD).
int heavy_tile_count(const uint32_t *totals, int n_tiles, float ratio, float trigger, int resident_waves) {
    double life[atmo::TILE_ORDER_CLASSES], sum = 0.0;
    long long counted = 0;
    for (int k = 0; k < atmo::TILE_ORDER_CLASSES; ++k) {
        constexpr int P = atmo::TILE_ORDER_PER_OCTAVE;
        const int q = atmo::TILE_ORDER_CLASSES - 1 - k + 8 * P;               // the class holds durations from 2^(q div P) (1 + (q mod P) / P) up to the next 1 / P of the octave
        life[k] = P == 2 ? std::ldexp(1.0, q >> 1) * ((q & 1) ? 1.5 : 1.0) * 1.2      // ... its middle (half octaves: the constants the split's trigger was tuned with)
                         : std::ldexp(1.0, q / P) * (1.0 + ((q % P) + 0.5) / P);
        if (k == atmo::TILE_ORDER_CLASSES - 1) life[k] = 0.0;                 // the last class also holds the tiles without a measurement
        sum += life[k] * (double)totals[k];
        counted += totals[k];
    }
    if (counted != n_tiles || sum <= 0.0) return 0;                           // not (yet) the histogram of this grid
    const double draw = sum * 2.0 / (double)resident_waves;
    int first = 0;
    while (first < atmo::TILE_ORDER_CLASSES - 1 && totals[first] == 0) ++first;
    if (!(life[first] > trigger * draw)) return 0;   // no wavefront outlives the draw's throughput estimate: the frame is not bound by its tail
    int heavy = 0;
    for (int k = 0; k < atmo::TILE_ORDER_CLASSES && life[k] > ratio * draw; ++k) heavy += (int)totals[k];
    return heavy > n_tiles / 3 ? n_tiles / 3 : heavy;
}

// Lanes per ray for one launch (atmo_set_lane_split; ATMO_LANE_SPLIT, read in atmo_create, overrides for A/B runs).  Two lanes per ray double
// the wave count at the price of a duplicated per-pixel prologue and regrouped view sums; measured on MI355X it pays
// only where a few very long waves set the kernel time (clouds_high_rm, 1920x1080, pose P_space: -11 %) and costs
// 5-40 % elsewhere (profiles/round2/ab_lane_split.txt), so "auto" (0) is one lane per ray.
int choose_split(const AtmoContext *ctx, const AtmoFrame *f) {
    (void)f;
    if (ctx->flags & atmo::KF_ATMO_REF) return 1;  // the reference-order v2 march has one launch shape
    if (ctx->env_split) return ctx->env_split;
    return ctx->lane_split == 2 ? 2 : 1;
}

// The kernel family and launch shape a draw of this context uses as it stands: the context's flags plus what is decided per draw -- long view
// marches in the reference's position form (KF_VIEW_POS), the cubemap's sampler (KF_CUBE_LOD) -- and the lanes per ray those forms exist in.
void launch_shape(const AtmoContext *ctx, const AtmoFrame *frame, int *flags_out, int *split_out, bool *lod_out) {
    int flags = ctx->flags, split = choose_split(ctx, frame);
    if (ctx->view_steps > 32 && !(flags & (atmo::KF_LITE | atmo::KF_ATMO_REF))) {
        // long view marches accumulate the position in the reference's form (march_atmosphere<VIEWPOS>): the default form's running sum
        // drifted to 1.08e-4 of alpha at 64 steps on thin atmospheres; up to 32 steps the kernels are the round-3 ones, byte for byte
        flags |= atmo::KF_VIEW_POS;
        split = 1;
    }
    const bool lod = resolve_sampler_lod(ctx, nullptr) != 0;
    if (lod) {
        flags |= atmo::KF_CUBE_LOD;
        // two lanes per ray under the declared sampler: only the two BASELINE cloud kernels have that form (atmo_set_lane_split 2 draws a
        // whole frame with it -- a test / A-B knob; by itself the library draws a frame's HEAVY tiles that way, render_impl)
        if ((flags & ~atmo::KF_CLOUD_LIGHT_RM) != (atmo::KF_CUBE_LOD | atmo::KF_PRECISE | atmo::KF_CLOUDS)) split = 1;
    }
    *flags_out = flags;
    *split_out = split;
    if (lod_out) *lod_out = lod;
}

// Reads back finished timing pairs.  only_completed: keep the pairs whose second event has not happened yet
// (used to bound `pending` without blocking); otherwise wait for every pair.
void drain_timing(AtmoContext *ctx, bool only_completed = false) {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> keep;
    for (auto &pr : ctx->pending) {
        if (only_completed && hipEventQuery(pr.second) == hipErrorNotReady) {
            keep.push_back(pr);
            continue;
        }
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
    ctx->pending.swap(keep);
}

// Grow-only device buffer: reallocates (hipFree = device-wide wait) only when the capacity is too small.
int dev_reserve(AtmoContext *ctx, DeviceBuffer &b, size_t bytes) {
    if (b.ptr && b.bytes >= bytes) return ATMO_OK;
    return dev_alloc(ctx, b, bytes);
}

// How far, in pixels, the picture's cost features move between two consecutive frames.  Scheduling only -- the pixels never
// depend on it -- so a plain symmetric perspective is assumed (inv_p[0], inv_p[5] = the tangents of the half field of view).
//   silhouette: the planet's centre and the four points of its limb along the camera's own right / up axes, each frame's
//               taken with that frame's camera: where the disc sits on the screen and how large it is (all a kernel without
//               clouds can see of the camera: a march costs the same wherever the sun and the ground features are);
//   surface:    six points fixed on the planet (world axes): where the cloud pattern is (the cloud kernels' cost map).
float feedback_motion_px(const AtmoFrame &a, const AtmoFrame &b, float radius, bool surface, float *silhouette_xy = nullptr) {
    auto to_world = [](const AtmoFrame &f, const float *v, float *w) {
        const float *M = f.inv_view_matrix;
        for (int r = 0; r < 3; ++r) w[r] = M[r] * v[0] + M[4 + r] * v[1] + M[8 + r] * v[2] + M[12 + r];
    };
    auto view_to_pixel = [](const AtmoFrame &f, const float *v, float *px) {
        if (!(v[2] < -1e-6f)) return false;
        const float tx = f.inv_projection_matrix[0], ty = f.inv_projection_matrix[5];
        if (tx == 0.0f || ty == 0.0f) return false;
        px[0] = (0.5f + 0.5f * (v[0] / -v[2]) / tx) * (float)f.viewport_w;
        px[1] = (0.5f + 0.5f * (v[1] / -v[2]) / ty) * (float)f.viewport_h;
        return px[0] == px[0] && px[1] == px[1];
    };
    auto world_to_pixel = [&](const AtmoFrame &f, const float *w, float *px) {
        const float *M = f.inv_view_matrix;  // rigid: view = R^T (w - t)
        const float d[3] = {w[0] - M[12], w[1] - M[13], w[2] - M[14]};
        const float v[3] = {M[0] * d[0] + M[1] * d[1] + M[2] * d[2], M[4] * d[0] + M[5] * d[1] + M[6] * d[2],
                            M[8] * d[0] + M[9] * d[1] + M[10] * d[2]};
        return view_to_pixel(f, v, px);
    };
    float worst = 0.0f, sil_x = 0.0f, sil_y = 0.0f;
    bool any = false, silhouette = true;
    auto take = [&](bool oka, const float *pa, bool okb, const float *pb) {
        if (!oka || !okb) return;
        const float dx = pa[0] - pb[0], dy = pa[1] - pb[1];
        worst = std::fmax(worst, std::sqrt(dx * dx + dy * dy));
        if (silhouette) { sil_x = std::fmax(sil_x, std::fabs(dx)); sil_y = std::fmax(sil_y, std::fabs(dy)); }
        any = true;
    };
    for (int k = 0; k < 5; ++k) {  // silhouette: view-space centre and centre +- R along view x / y, per frame
        float va[3] = {a.planet_center_viewspace[0], a.planet_center_viewspace[1], a.planet_center_viewspace[2]};
        float vb[3] = {b.planet_center_viewspace[0], b.planet_center_viewspace[1], b.planet_center_viewspace[2]};
        if (k > 0) {
            const float o = ((k - 1) & 1) ? -radius : radius;
            va[(k - 1) >> 1] += o;
            vb[(k - 1) >> 1] += o;
        }
        float pa[2], pb[2];
        const bool oka = view_to_pixel(a, va, pa), okb = view_to_pixel(b, vb, pb);
        take(oka, pa, okb, pb);
    }
    silhouette = false;
    if (surface) {
        float c[3];
        to_world(a, a.planet_center_viewspace, c);
        for (int k = 0; k < 6; ++k) {
            float w[3] = {c[0], c[1], c[2]};
            w[k >> 1] += (k & 1) ? -radius : radius;
            float pa[2], pb[2];
            const bool oka = world_to_pixel(a, w, pa), okb = world_to_pixel(b, w, pb);
            take(oka, pa, okb, pb);
        }
    }
    if (!any) {  // nothing of the planet in front of the camera: the camera's own turn, in pixels
        const float *A = a.inv_view_matrix, *B = b.inv_view_matrix;
        float tr = 0.0f;
        for (int cidx = 0; cidx < 3; ++cidx)
            for (int r = 0; r < 3; ++r) tr += A[cidx * 4 + r] * B[cidx * 4 + r];
        const float cosang = std::fmin(std::fmax(0.5f * (tr - 1.0f), -1.0f), 1.0f);
        const float ty = std::fabs(a.inv_projection_matrix[5]);
        worst = std::acos(cosang) * (ty > 0.0f ? 0.5f * (float)a.viewport_h / ty : (float)a.viewport_h);
        sil_x = sil_y = worst;
    }
    if (silhouette_xy) { silhouette_xy[0] = sil_x; silhouette_xy[1] = sil_y; }
    return worst;
}

// Orders everything enqueued on `waiter` from now on behind everything enqueued on `other` so far -- on the DEVICE: a marker event recorded
// on `other`, a stream-side wait on `waiter`; the host does not block and no other queue of the process is involved (round 4 called
// hipDeviceSynchronize in these places, which stalls every stream of the engine: VERDICT r4 weak #11).
// ONLY for an `other` that is known to be alive: a stream the context owns (its sort stream) or one the caller handed to the call in
// progress.  A stream the context merely REMEMBERS may have been destroyed by the caller since, and HIP dereferences stale stream handles
// (hipEventRecord on one crashes the process: round 5, first attempt) -- draws on remembered streams are waited for through the marker
// event recorded behind them at draw time instead (order_after_draws).
int order_after_stream(AtmoContext *ctx, hipStream_t waiter, hipStream_t other) {
    if (waiter == other) return ATMO_OK;
    hipEvent_t &ev = ctx->xs_event[ctx->xs_next++ % 8u];
    if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, other);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->device_syncs += 1;
        HIP_TRY(ctx, hipDeviceSynchronize());
    }
    return ATMO_OK;
}

// Orders `waiter` behind the draws this context has enqueued on the remembered stream `d` (its marker event; no use of d's handle).
int order_after_draws(AtmoContext *ctx, hipStream_t waiter, const AtmoContext::DrawStream &d) {
    if (d.stream == waiter) return ATMO_OK;   // stream order
    if (d.recorded && d.last_draw) {
        HIP_TRY(ctx, hipStreamWaitEvent(waiter, d.last_draw, 0));
        return ATMO_OK;
    }
    if (d.stream == nullptr) return order_after_stream(ctx, waiter, nullptr);   // the null stream cannot have been destroyed
    // draws without a marker on a stream that may be gone: the home stream of a host that has just moved its texture updates to another
    // stream (or ATMO_DRAW_EVENTS=0) -- the device-wide wait of rounds 3-4, once per such move
    ctx->device_syncs += 1;
    HIP_TRY(ctx, hipDeviceSynchronize());
    return ATMO_OK;
}
int order_after_draws_on(AtmoContext *ctx, hipStream_t waiter, hipStream_t drawn_on) {
    if (drawn_on == waiter) return ATMO_OK;
    for (const AtmoContext::DrawStream &d : ctx->draw_streams)
        if (d.stream == drawn_on) return order_after_draws(ctx, waiter, d);
    ctx->device_syncs += 1;   // a stream this context no longer remembers (more than 8 in use)
    HIP_TRY(ctx, hipDeviceSynchronize());
    return ATMO_OK;
}

// Makes sure nothing enqueued earlier can still touch the buffers of feedback state `f` (its pending sort; draws on its stream reading an
// order) once work enqueued on `new_stream` from now on runs: stream-side waits, the host does not block.  Only needed when the slot is
// recycled for another key.  (A buffer that has to GROW is freed, and hipFree waits for the device by itself.)
int feedback_quiesce(AtmoContext *ctx, AtmoContext::FeedbackState &f, hipStream_t new_stream) {
    if (f.pending && ctx->fb_stream) {
        const int rc0 = order_after_stream(ctx, new_stream, ctx->fb_stream);
        if (rc0 != ATMO_OK) return rc0;
    }
    if ((f.n > 0 || f.dirty) && f.draw_stream != new_stream) {
        const int rc0 = order_after_draws_on(ctx, new_stream, f.draw_stream);
        if (rc0 != ATMO_OK) return rc0;
    }
    f.pending = false;
    f.dirty = false;
    return ATMO_OK;
}

// The feedback state for draws of a (gx, gy, split) grid on stream s: the cached one, a free slot, or -- at most a few
// times in a row -- the least recently used slot recycled.  *out stays null when the context is producing new keys faster
// than its recycling budget allows: those draws run in row-major order and nothing waits.
int feedback_state(AtmoContext *ctx, int gx, int gy, int split, hipStream_t s, AtmoContext::FeedbackState **out) {
    *out = nullptr;
    ctx->fb_clock += 1;
    if ((ctx->fb_clock & 255u) == 0 && ctx->fb_budget < 8) ctx->fb_budget += 1;
    AtmoContext::FeedbackState *slot = nullptr, *lru = nullptr;
    for (AtmoContext::FeedbackState &f : ctx->fb) {
        if (f.used && f.tiles_x == gx && f.tiles_y == gy && f.split == split && f.draw_stream == s) {
            f.last_use = ctx->fb_clock;
            *out = &f;
            return ATMO_OK;
        }
        if (!f.used) { if (!slot) slot = &f; }
        else if (!lru || f.last_use < lru->last_use) lru = &f;
    }
    if (!ctx->fb_stream) {
        int lo = 0, hi = 0;  // numerically lower = higher priority
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->fb_stream, hipStreamNonBlocking, hi));
        const int rc0 = dev_reserve(ctx, ctx->fb_scratch, atmo::tile_order_scratch_bytes());
        if (rc0 != ATMO_OK) return rc0;
    }
    if (!slot) {
        if (ctx->fb_budget <= 0) return ATMO_OK;  // no feedback for this draw
        ctx->fb_budget -= 1;
        ctx->fb_recycled += 1;
        slot = lru;
        const int rc0 = feedback_quiesce(ctx, *slot, s);
        if (rc0 != ATMO_OK) return rc0;
    }
    if (slot->dirty) {  // released by atmo_set_tile_feedback while its draws / sort may still have been in flight
        const int rc0 = feedback_quiesce(ctx, *slot, s);
        if (rc0 != ATMO_OK) return rc0;
    }
    AtmoContext::FeedbackState &f = *slot;
    const size_t bytes = (size_t)gx * gy * sizeof(uint32_t);
    // (a slot taken here is ordered behind whatever used it before; growing a buffer frees it, which waits for the device)
    int rc1 = dev_reserve(ctx, f.cost, bytes);
    for (int k = 0; k < 2 && rc1 == ATMO_OK; ++k) rc1 = dev_reserve(ctx, f.order[k], bytes);
    for (int k = 0; k < 2 && rc1 == ATMO_OK; ++k) rc1 = dev_reserve(ctx, f.dil[k], bytes);
    if (rc1 == ATMO_OK) rc1 = dev_reserve(ctx, f.is_order, bytes);
    if (rc1 == ATMO_OK) rc1 = dev_reserve(ctx, f.is_scratch, atmo::tile_order_scratch_bytes());
    for (int k = 0; k < 2 && rc1 == ATMO_OK; ++k) rc1 = dev_reserve(ctx, f.order2[k], 2 * bytes);
    if (rc1 == ATMO_OK) rc1 = dev_reserve(ctx, f.is_order2, 2 * bytes);
    if (rc1 == ATMO_OK && !f.class_totals) {
        void *p = nullptr;
        HIP_TRY(ctx, hipHostMalloc(&p, 3 * atmo::TILE_ORDER_CLASSES * sizeof(uint32_t), hipHostMallocDefault));
        f.class_totals = (uint32_t *)p;
    }
    if (rc1 == ATMO_OK) std::memset(f.class_totals, 0, 3 * atmo::TILE_ORDER_CLASSES * sizeof(uint32_t));
    if (rc1 != ATMO_OK) { f.used = false; return rc1; }
    if (!f.ev_draw) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&f.ev_draw, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) HIP_TRY(ctx, hipEventCreateWithFlags(&f.ev_order[k], hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipMemsetAsync(f.cost.ptr, 0, bytes, s));
    f.used = true;
    f.tiles_x = gx; f.tiles_y = gy; f.split = split;
    f.draw_stream = s;
    f.n = 0;
    f.last_record = 0;
    f.active = -1;
    f.write = 0;
    f.pending = false;
    f.have_prev = false;
    f.motion_px = 0.0f;
    f.is_last_n = ~0u;
    f.last_use = ctx->fb_clock;
    *out = &f;
    return ATMO_OK;
}

}  // namespace

extern "C" {

int atmo_abi_version(void) { return ATMO_ABI_VERSION; }
#ifndef ATMO_BUILD_ID
#define ATMO_BUILD_ID "unstamped"   /* a library not built through godot_atmosphere_shader_amd/build.py (tools/ab_build.sh ...) */
#endif
const char *atmo_build_id(void) { return ATMO_BUILD_ID; }

int atmo_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -ATMO_E_NO_DEVICE;
    return n;
}

// the argument checks and the variant -> kernel-family mapping of atmo_create, shared with the host-only form of atmo_debug.h
static int check_create_args(int variant, int view_steps, int cloud_steps, int light_mode, int light_steps, AtmoContext **out) {
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
    return ATMO_OK;
}
static void init_variant(AtmoContext *ctx, int device, int variant, int view_steps, int cloud_steps, int light_mode, int light_steps) {
    ctx->device = device;
    ctx->variant = variant;
    const bool lite = variant >= ATMO_VARIANT_V1_NO_CLOUDS;
    static const int shipped_cloud_steps[7] = {0, 32, 64, 64, 0, 32, 64};  // shaders/planet_atmosphere_*.gdshader:4-7
    ctx->view_steps = view_steps > 0 ? view_steps : (lite ? 16 : 8);
    ctx->cloud_steps = shipped_cloud_steps[variant] == 0 ? 0 : (cloud_steps > 0 ? cloud_steps : shipped_cloud_steps[variant]);
    ctx->light_steps = (light_mode == ATMO_LIGHT_DIRECT && !lite) ? light_steps : 0;
    ctx->flags = 0;
    if (ctx->cloud_steps > 0) ctx->flags |= atmo::KF_CLOUDS | atmo::KF_PRECISE;  // precise cloud density is the default
    if (lite) ctx->flags |= atmo::KF_PRECISE;                                    // and so is the v1 march in reference order
    if (variant == ATMO_VARIANT_CLOUDS_HIGH_RM) ctx->flags |= atmo::KF_CLOUD_LIGHT_RM;
    if (light_mode == ATMO_LIGHT_DIRECT && !lite) ctx->flags |= atmo::KF_LIGHT_DIRECT;
    if (lite) ctx->flags |= atmo::KF_LITE;  // the v1 atmosphere reads no optical-depth LUT and has no light march
}

int atmo_create(int device, int variant, int view_steps, int cloud_steps, int light_mode, int light_steps,
                AtmoContext **out) {
    { const int rc0 = check_create_args(variant, view_steps, cloud_steps, light_mode, light_steps, out); if (rc0 != ATMO_OK) return rc0; }
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
    init_variant(ctx, device, variant, view_steps, cloud_steps, light_mode, light_steps);

    // A/B overrides for the tools (tools/ab_feedback.sh, tools/ab_bench.sh): read here, once -- never in the launch path
    if (const char *ev = std::getenv("ATMO_LANE_SPLIT")) ctx->env_split = ev[0] == '1' ? 1 : (ev[0] == '2' ? 2 : 0);
    if (const char *ev = std::getenv("ATMO_TILE_FEEDBACK")) ctx->env_feedback = ev[0] == '1' ? 1 : 0;
    if (const char *ev = std::getenv("ATMO_F4")) ctx->f4_footprints = std::atoi(ev) & 3;
    if (const char *ev = std::getenv("ATMO_F4_SHAPE_MAX")) ctx->f4_max_shape_n = std::atoi(ev);
    if (const char *ev = std::getenv("ATMO_LOD0_CERT")) ctx->env_lod0_cert = ev[0] == '0' ? 0 : 1;
    if (const char *ev = std::getenv("ATMO_HEAVY_SPLIT")) ctx->heavy_split = ev[0] == '0' ? 0 : (ev[0] == '2' ? 2 : 1);
    if (const char *ev = std::getenv("ATMO_HEAVY_SPLIT_RATIO")) ctx->heavy_split_ratio = (float)std::atof(ev);
    if (const char *ev = std::getenv("ATMO_HEAVY_SPLIT_TRIGGER")) ctx->heavy_split_trigger = (float)std::atof(ev);
    if (const char *ev = std::getenv("ATMO_HEAVY_SPLIT_TRIGGER_MOVING")) ctx->heavy_split_trigger_moving = (float)std::atof(ev);
    if (const char *ev = std::getenv("ATMO_DRAW_EVENTS")) ctx->draw_events = ev[0] == '0' ? 0 : (ev[0] == '2' ? 2 : 1);
    if (const char *ev = std::getenv("ATMO_TARGET_CLEARED")) ctx->target_cleared = ev[0] == '1' ? 1 : 0;  // tools/ab_env.sh: atmo_set_target_cleared
    if (const char *ev = std::getenv("ATMO_GEO_ORDER")) ctx->geo_order = ev[0] == '0' ? 0 : 1;
    if (const char *ev = std::getenv("ATMO_FB_INSTREAM")) ctx->instream = ev[0] >= '1' && ev[0] <= '2' ? ev[0] - '0' : 0;  // 2 (A/B): every cloud and direct-light kernel
    if (const char *ev = std::getenv("ATMO_FB_AXIS_WINDOWS")) ctx->fb_axis_windows = ev[0] == '1' ? 1 : 0;
    if (const char *ev = std::getenv("ATMO_FB_REACH_SCALE")) ctx->env_reach_scale = (float)std::atof(ev);
    if (const char *ev = std::getenv("ATMO_FB_MOVING_PERIOD")) { const int v = std::atoi(ev); ctx->moving_period = (unsigned)(v < 1 ? 1 : v); }
    if (const char *ev = std::getenv("ATMO_TILE_FEEDBACK_PERIOD")) { const int v = std::atoi(ev); ctx->fb_period = (unsigned)(v < 1 ? 1 : v); }

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
    if (ctx->device < 0) { delete ctx; return ATMO_OK; }  // a host-only context (atmo_debug_create_host_only) owns nothing on a device
    (void)hipSetDevice(ctx->device);
    drain_timing(ctx);
    dev_free(ctx->lut);
    dev_free(ctx->lut4);
    dev_free(ctx->blue);
    dev_free(ctx->shape);
    dev_free(ctx->cube);
    dev_free(ctx->cube_level_off);
    dev_free(ctx->cube_f4);
    dev_free(ctx->shape_f4);
    dev_free(ctx->staging);
    dev_free(ctx->measure_buf);
    for (AtmoContext::TileListBuf &t : ctx->tile_lists) dev_free(t.buf);
    if (ctx->tex_event) (void)hipEventDestroy(ctx->tex_event);
    for (hipEvent_t &ev : ctx->xs_event) if (ev) (void)hipEventDestroy(ev);
    for (AtmoContext::DrawStream &d : ctx->draw_streams) if (d.last_draw) (void)hipEventDestroy(d.last_draw);

    if (ctx->fb_stream) {
        (void)hipStreamSynchronize(ctx->fb_stream);
        (void)hipStreamDestroy(ctx->fb_stream);
    }
    if (ctx->split_stream) {
        (void)hipStreamSynchronize(ctx->split_stream);
        (void)hipStreamDestroy(ctx->split_stream);
    }
    for (AtmoContext::FeedbackState &f : ctx->fb) {
        if (f.ev_draw) (void)hipEventDestroy(f.ev_draw);
        for (int k = 0; k < 2; ++k) {
            if (f.ev_order[k]) (void)hipEventDestroy(f.ev_order[k]);
            dev_free(f.order[k]);
            dev_free(f.dil[k]);
        }
        dev_free(f.is_order);
        dev_free(f.is_scratch);
        dev_free(f.is_order2);
        for (int k = 0; k < 2; ++k) dev_free(f.order2[k]);
        if (f.class_totals) (void)hipHostFree(f.class_totals);
        dev_free(f.cost);
    }
    dev_free(ctx->fb_scratch);
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

// Raw texels of a texture update, on the device: the caller's buffer itself (ATMO_MEM_DEVICE) or the context's staging
// buffer filled by an asynchronous copy on `s` (ATMO_MEM_HOST).  The staging buffer is reused by the next update, which
// is safe on one stream; an update arriving on another stream first waits for the previous one.
static int stage_texels(AtmoContext *ctx, const void *data, size_t bytes, int memory, hipStream_t s, size_t extra_bytes, uint8_t **out) {
    if (memory == ATMO_MEM_DEVICE && extra_bytes == 0) { *out = (uint8_t *)const_cast<void *>(data); return ATMO_OK; }
    // the previous update's kernels may still read the staging buffer on their stream: every update ends with tex_event recorded behind its
    // last kernel (tex_updated), and tex_begin_update has already ordered `s` behind it -- nothing more to wait for here
    if (ctx->staging.bytes < bytes + extra_bytes) {  // (re-allocation: hipFree waits for the device by itself)
        const int rc = dev_alloc(ctx, ctx->staging, bytes + extra_bytes);
        if (rc != ATMO_OK) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->staging.ptr, data, bytes, memory == ATMO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    ctx->staging_stream = s;
    ctx->staging_used = true;
    *out = (uint8_t *)ctx->staging.ptr;
    return ATMO_OK;
}

// same-size updates overwrite the bound copy in place (stream-ordered); a size change frees it, which waits for the device
static int tex_alloc(AtmoContext *ctx, DeviceBuffer &b, size_t bytes) { return dev_alloc(ctx, b, bytes); }

// The two device copies of a w x h optical-depth LUT (apron layout + footprints), allocated together: on failure
// neither stays bound, so no launch ever pairs one copy with the other's stale size.
static int lut_alloc(AtmoContext *ctx, int w, int h) {
    int rc = tex_alloc(ctx, ctx->lut, (size_t)(w + 2) * (h + 2) * sizeof(float));
    if (rc == ATMO_OK) rc = tex_alloc(ctx, ctx->lut4, (size_t)(w + 1) * (h + 1) * 16);
    if (rc != ATMO_OK) {
        dev_free(ctx->lut);
        dev_free(ctx->lut4);
        ctx->lut_w = ctx->lut_h = 0;
        return rc;
    }
    ctx->lut_w = w; ctx->lut_h = h;
    return ATMO_OK;
}

static int tex_updated(AtmoContext *ctx, hipStream_t s) {
    if (!ctx->tex_event) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->tex_event, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->tex_event, s));
    ctx->tex_stream = s;
    ctx->tex_pending = true;
    ctx->tex_version += 1;
    return ATMO_OK;
}

// In front of every texture update on stream `s`:
//   * an earlier update on ANOTHER stream is chained in front (hipStreamWaitEvent on its event), so the updates of a
//     context happen in call order whatever streams they come in on, and a draw on `s` that finds tex_stream == s is
//     behind all of them;
//   * draws still reading the bound copy on ANY other stream are ordered in front of the update ON THE DEVICE (order_after_draws: the
//     marker event recorded behind each stream's last draw, a stream-side wait on `s`; round 5 -- rounds 3-4 waited device-wide on the
//     host, which stalls every queue of the engine).  Every stream that has carried a draw since the last update counts, not only the most recent one (round 3
//     remembered one stream: with draws in flight on A and B, an update on B overwrote what A was still reading).  A context that has
//     drawn on more streams than it tracks (8) falls back to the device-wide wait.  Updates are rare and normally arrive on the one
//     draw stream, where stream order is enough and nothing is enqueued at all.
static int tex_begin_update(AtmoContext *ctx, hipStream_t s) {
    if (ctx->tex_pending && ctx->tex_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->tex_event, 0));
    if (ctx->draw_streams_many) {
        ctx->device_syncs += 1;
        HIP_TRY(ctx, hipDeviceSynchronize());
    } else {
        for (const AtmoContext::DrawStream &d : ctx->draw_streams) {
            const int rc0 = order_after_draws(ctx, s, d);
            if (rc0 != ATMO_OK) return rc0;
        }
    }
    // every earlier draw of this context is in front of the update now; later ones wait for tex_event (tex_order).  The streams stay
    // remembered with their markers: the feedback states and tile-list copies that drew on them are handed on behind those markers
    ctx->draw_streams_many = false;
    return ATMO_OK;
}

// Orders stream `s` behind the last texture update when that happened on another stream.  Once per (stream, update):
// a wait in front of EVERY launch puts a barrier packet between back-to-back draws (+6 us per draw measured).
static int tex_order(AtmoContext *ctx, hipStream_t s) {
    if (!ctx->tex_pending || ctx->tex_stream == s) return ATMO_OK;
    if (ctx->tex_waited_stream == s && ctx->tex_waited_version == ctx->tex_version) return ATMO_OK;
    HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->tex_event, 0));
    ctx->tex_waited_stream = s;
    ctx->tex_waited_version = ctx->tex_version;
    return ATMO_OK;
}

int atmo_set_texture(AtmoContext *ctx, const char *name, int kind, int w, int h, int d, int mips, const void *data, int memory,
                     void *stream) {
    if (!ctx) return ATMO_E_ARG;
    if (!name) return fail(ctx, ATMO_E_NAME, "atmo_set_texture: name is null");
    if (memory != ATMO_MEM_HOST && memory != ATMO_MEM_DEVICE) return fail(ctx, ATMO_E_ARG, "atmo_set_texture: bad memory kind");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const hipMemcpyKind ck = memory == ATMO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const bool is_cube = std::strcmp(name, "u_cloud_coverage_cubemap") == 0;
    if (!is_cube && data && mips != 0 && mips != 1) return fail(ctx, ATMO_E_ARG, "atmo_set_texture: only the cubemap takes mip levels");

    if (std::strcmp(name, "u_optical_depth_texture") == 0) {
        if (!data) { dev_free(ctx->lut); dev_free(ctx->lut4); ctx->lut_w = ctx->lut_h = 0; return ATMO_OK; }
        if (kind != ATMO_TEX_2D_R32F) return fail(ctx, ATMO_E_ARG, "u_optical_depth_texture must be ATMO_TEX_2D_R32F");
        // the kernels form the footprint byte offset in fp32: (w + 1) (h + 1) 16 must stay below 2^24
        if (w < 1 || h < 1 || w > 1023 || h > 1023) return fail(ctx, ATMO_E_ARG, "u_optical_depth_texture: each side must be 1..1023 texels (the reference bakes 256 x 256)");
        uint8_t *raw = nullptr;
        int rc = tex_begin_update(ctx, s);
        if (rc == ATMO_OK) rc = stage_texels(ctx, data, (size_t)w * h * sizeof(float), memory, s, 0, &raw);
        if (rc == ATMO_OK) rc = lut_alloc(ctx, w, h);  // both copies, before anything is enqueued
        if (rc != ATMO_OK) return rc;
        HIP_TRY(ctx, atmo::launch_layout_lut((const float *)raw, w, h, (float *)ctx->lut.ptr, s));
        HIP_TRY(ctx, atmo::launch_lut_footprints((const float *)ctx->lut.ptr, w, h, (float *)ctx->lut4.ptr, s));
        return tex_updated(ctx, s);
    }
    if (std::strcmp(name, "u_blue_noise_texture") == 0) {
        if (!data) {
            const int rc0 = tex_begin_update(ctx, s);
            if (rc0 != ATMO_OK) return rc0;
            HIP_TRY(ctx, hipMemsetAsync(ctx->blue.ptr, 0, 256 * 256, s));
            return tex_updated(ctx, s);
        }
        if (kind != ATMO_TEX_2D_R8) return fail(ctx, ATMO_E_ARG, "u_blue_noise_texture must be ATMO_TEX_2D_R8");
        if (w != 256 || h != 256) return fail(ctx, ATMO_E_ARG, "u_blue_noise_texture must be 256x256 (indexed & 0xff, main:169)");
        { const int rc0 = tex_begin_update(ctx, s); if (rc0 != ATMO_OK) return rc0; }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->blue.ptr, data, 256 * 256, ck, s));
        return tex_updated(ctx, s);
    }
    if (std::strcmp(name, "u_cloud_shape_texture") == 0) {
        if (!data) { dev_free(ctx->shape); dev_free(ctx->shape_f4); ctx->shape_n = 0; return ATMO_OK; }
        if (kind != ATMO_TEX_3D_R8) return fail(ctx, ATMO_E_ARG, "u_cloud_shape_texture must be ATMO_TEX_3D_R8");
        if (w < 1 || w > 512 || h != w || d != w) return fail(ctx, ATMO_E_ARG, "u_cloud_shape_texture must be n x n x n, n <= 512");
        uint8_t *raw = nullptr;
        int rc = tex_begin_update(ctx, s);
        if (rc == ATMO_OK) rc = stage_texels(ctx, data, (size_t)w * w * w, memory, s, 0, &raw);
        if (rc == ATMO_OK) rc = tex_alloc(ctx, ctx->shape, (size_t)w * w * w * sizeof(uint32_t));
        // the float copy (16 B per footprint) only for volumes small enough to stay L2-resident; larger ones are sampled from the byte footprints
        const bool f4 = w <= ctx->f4_max_shape_n;
        if (rc == ATMO_OK) { if (f4) rc = tex_alloc(ctx, ctx->shape_f4, (size_t)w * w * w * 16); else dev_free(ctx->shape_f4); }
        if (rc != ATMO_OK) {
            dev_free(ctx->shape);
            dev_free(ctx->shape_f4);
            ctx->shape_n = 0;
            return rc;
        }
        HIP_TRY(ctx, atmo::launch_layout_shape(raw, w, (uint32_t *)ctx->shape.ptr, s));
        if (f4) HIP_TRY(ctx, atmo::launch_footprints_f4((const uint32_t *)ctx->shape.ptr, (size_t)w * w * w, (float *)ctx->shape_f4.ptr, s));
        ctx->shape_n = w;
        return tex_updated(ctx, s);
    }
    if (is_cube) {
        if (!data) { dev_free(ctx->cube); dev_free(ctx->cube_f4); ctx->cube_n = 0; ctx->cube_levels = 0; return ATMO_OK; }
        if (kind != ATMO_TEX_CUBE_R8) return fail(ctx, ATMO_E_ARG, "u_cloud_coverage_cubemap must be ATMO_TEX_CUBE_R8");
        if (w < 1 || w > 4096 || h != w || d != 6) return fail(ctx, ATMO_E_ARG, "u_cloud_coverage_cubemap must be n x n x 6 faces");
        const int full = atmo::cube_full_mip_count(w);
        if (mips < 0 || mips > full) return fail(ctx, ATMO_E_ARG, "u_cloud_coverage_cubemap: mips must be 0 (generate the chain), 1 (level 0 only) .. log2(n)+1");
        const int given = mips == 0 ? 1 : mips;      // levels present in `data`, packed level after level
        const int levels = mips == 0 ? full : mips;  // levels bound afterwards
        size_t given_bytes = 0, chain_bytes = 0, fp_words = 0;
        uint32_t level_off[16] = {0};
        for (int l = 0; l < levels; ++l) {
            if (l < given) given_bytes += atmo::cube_level_texels(w, l);
            chain_bytes += atmo::cube_level_texels(w, l);
            level_off[l] = (uint32_t)fp_words;
            fp_words += atmo::cube_level_footprints(w, l);
        }
        // the texel chain lives in the staging buffer: given levels copied in, missing ones generated behind them
        uint8_t *raw = nullptr;
        int rc = tex_begin_update(ctx, s);
        if (rc == ATMO_OK) rc = stage_texels(ctx, data, given_bytes, memory, s, chain_bytes - given_bytes + 1, &raw);
        if (rc == ATMO_OK) rc = tex_alloc(ctx, ctx->cube, fp_words * sizeof(uint32_t));
        if (rc == ATMO_OK) rc = dev_alloc(ctx, ctx->cube_level_off, sizeof(ctx->cube_level_off_host));
        // the float copy of the chain (16 B per footprint) up to 1024^2 faces = 134 MB; larger faces are sampled from the byte footprints
        const bool f4 = w <= F4_MAX_CUBE_N;
        if (rc == ATMO_OK) { if (f4) rc = tex_alloc(ctx, ctx->cube_f4, fp_words * 16); else dev_free(ctx->cube_f4); }
        if (rc != ATMO_OK) {  // nothing half-bound: the cubemap is unset (= 1.0) until a later update succeeds
            dev_free(ctx->cube);
            dev_free(ctx->cube_f4);
            ctx->cube_n = 0; ctx->cube_levels = 0;
            return rc;
        }
        std::memcpy(ctx->cube_level_off_host, level_off, sizeof(level_off));
        size_t off = 0;
        for (int l = 0; l < levels; ++l) {
            const int nl = w >> l;
            if (l + 1 < levels && l + 1 >= given)
                HIP_TRY(ctx, atmo::launch_cube_mip(raw + off, nl, raw + off + atmo::cube_level_texels(w, l), s));
            HIP_TRY(ctx, atmo::launch_layout_cube(raw + off, nl, (uint32_t *)ctx->cube.ptr + ctx->cube_level_off_host[l], s));
            off += atmo::cube_level_texels(w, l);
        }
        if (f4) HIP_TRY(ctx, atmo::launch_footprints_f4((const uint32_t *)ctx->cube.ptr, fp_words, (float *)ctx->cube_f4.ptr, s));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->cube_level_off.ptr, ctx->cube_level_off_host, sizeof(ctx->cube_level_off_host), hipMemcpyHostToDevice, s));
        ctx->cube_n = w;
        ctx->cube_levels = levels;
        return tex_updated(ctx, s);
    }
    return fail(ctx, ATMO_E_NAME, std::string("atmo_set_texture: unknown texture uniform '") + name + "'");
}

int atmo_get_texture_size(AtmoContext *ctx, const char *name, int *w, int *h, int *d, int *mips) {
    if (!ctx) return ATMO_E_ARG;
    if (!name) return fail(ctx, ATMO_E_NAME, "atmo_get_texture_size: name is null");
    int W = 0, H = 0, D = 0, M = 0;
    if (std::strcmp(name, "u_optical_depth_texture") == 0) { W = ctx->lut_w; H = ctx->lut_h; D = W ? 1 : 0; M = W ? 1 : 0; }
    else if (std::strcmp(name, "u_blue_noise_texture") == 0) { W = H = 256; D = 1; M = 1; }
    else if (std::strcmp(name, "u_cloud_shape_texture") == 0) { W = H = D = ctx->shape_n; M = W ? 1 : 0; }
    else if (std::strcmp(name, "u_cloud_coverage_cubemap") == 0) { W = H = ctx->cube_n; D = W ? 6 : 0; M = ctx->cube_levels; }
    else return fail(ctx, ATMO_E_NAME, std::string("atmo_get_texture_size: unknown texture uniform '") + name + "'");
    if (w) *w = W;
    if (h) *h = H;
    if (d) *d = D;
    if (mips) *mips = M;
    return ATMO_OK;
}

int atmo_read_texture_layout(AtmoContext *ctx, const char *name, void *out_host, size_t capacity_bytes, size_t *bytes_out, void *stream) {
    if (!ctx) return ATMO_E_ARG;
    if (!name) return fail(ctx, ATMO_E_NAME, "atmo_read_texture_layout: name is null");
    const DeviceBuffer *b = nullptr;
    if (std::strcmp(name, "u_optical_depth_texture") == 0) b = &ctx->lut;
    else if (std::strcmp(name, "u_blue_noise_texture") == 0) b = &ctx->blue;
    else if (std::strcmp(name, "u_cloud_shape_texture") == 0) b = &ctx->shape;
    else if (std::strcmp(name, "u_cloud_coverage_cubemap") == 0) b = &ctx->cube;
    else return fail(ctx, ATMO_E_NAME, std::string("atmo_read_texture_layout: unknown texture uniform '") + name + "'");
    if (bytes_out) *bytes_out = b->bytes;
    if (!out_host) return ATMO_OK;
    if (capacity_bytes < b->bytes) return fail(ctx, ATMO_E_ARG, "atmo_read_texture_layout: buffer too small");
    if (!b->ptr) return ATMO_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    { const int rc0 = tex_order(ctx, s); if (rc0 != ATMO_OK) return rc0; }
    HIP_TRY(ctx, hipMemcpyAsync(out_host, b->ptr, b->bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return ATMO_OK;
}

int atmo_set_sampler_lod(AtmoContext *ctx, int mode) {
    if (!ctx) return ATMO_E_ARG;
    if (mode < -1 || mode > 1) return fail(ctx, ATMO_E_ARG, "atmo_set_sampler_lod: -1 (as declared: implicit LOD when a mip chain is bound), 0 (LOD 0) or 1 (implicit LOD required)");
    ctx->sampler_lod = mode;
    return ATMO_OK;
}

// ---- host-only layout helpers (no device needed): what atmo_set_texture uploads -----------------------------------
int atmo_host_layout_cubemap(const uint8_t *faces, int n, uint32_t *footprints_out) {
    if (!faces || !footprints_out || n < 1 || n > 4096) return ATMO_E_ARG;
    std::vector<uint32_t> fp;
    build_cube_footprints(faces, n, fp);
    std::memcpy(footprints_out, fp.data(), fp.size() * sizeof(uint32_t));
    return ATMO_OK;
}

int atmo_host_cubemap_mip(const uint8_t *level, int n, uint8_t *next_out) {
    if (!level || !next_out || n < 2 || n > 4096) return ATMO_E_ARG;
    const int m = n >> 1;
    for (int f = 0; f < 6; ++f)
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) next_out[((size_t)f * m + j) * m + i] = atmo::cube_mip_texel(level, n, f, i, j);
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
        rc = atmo_set_texture(ctx, "u_cloud_coverage_cubemap", ATMO_TEX_CUBE_R8, resolution, resolution, 6, /*mips: generate the chain*/ 0,
                              dev, ATMO_MEM_DEVICE, nullptr);  // stays on the device: mip + re-layout kernels, no host round trip
    (void)hipFree(dev);
    return rc;
}

int atmo_bake_optical_depth(AtmoContext *ctx, void *stream) {
    if (!ctx) return ATMO_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int w = 256, h = 256;  // optical_depth_baker.gd:24
    { const int rc0 = tex_begin_update(ctx, (hipStream_t)stream); if (rc0 != ATMO_OK) return rc0; }
    if (ctx->lut_w != w || ctx->lut_h != h || !ctx->lut.ptr || !ctx->lut4.ptr) {
        const int rc = lut_alloc(ctx, w, h);  // a size change waits for the device (hipFree)
        if (rc != ATMO_OK) return rc;
    }
    atmo::BakeConsts bc;
    bc.planet_radius = ctx->p.u_planet_radius;
    bc.atmosphere_height = ctx->p.u_atmosphere_height;
    bc.density = ctx->p.u_density;
    bc.w = w; bc.h = h;
    bc.steps = 64;  // optical_depth.gdshader:18
    bc.out = (float *)ctx->lut.ptr;
    HIP_TRY(ctx, atmo::launch_bake(bc, (hipStream_t)stream));
    HIP_TRY(ctx, atmo::launch_lut_footprints((const float *)ctx->lut.ptr, w, h, (float *)ctx->lut4.ptr, (hipStream_t)stream));
    return tex_updated(ctx, (hipStream_t)stream);  // draws and read-backs on other streams wait for this bake
}

int atmo_read_optical_depth(AtmoContext *ctx, float *lut_host, uint8_t *rgba8_host, int capacity_texels, void *stream) {
    if (!ctx) return ATMO_E_ARG;
    if (!ctx->lut.ptr) return fail(ctx, ATMO_E_STATE, "atmo_read_optical_depth: no LUT bound");
    const int n = ctx->lut_w * ctx->lut_h;
    if (capacity_texels < n) return fail(ctx, ATMO_E_ARG, "atmo_read_optical_depth: buffer too small (atmo_get_texture_size gives w x h)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    { const int rc0 = tex_order(ctx, s); if (rc0 != ATMO_OK) return rc0; }  // a bake on another stream
    std::vector<float> tmp, padded((size_t)(ctx->lut_w + 2) * (ctx->lut_h + 2));
    float *dst = lut_host;
    if (!dst) { tmp.resize(n); dst = tmp.data(); }
    HIP_TRY(ctx, hipMemcpyAsync(padded.data(), ctx->lut.ptr, padded.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
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

static int render_impl(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream, bool composite,
                       const uint32_t *tiles_dev = nullptr, int n_tiles = 0, int n_heavy = 0);

int atmo_render(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream) {
    return render_impl(ctx, frame, depth_dev, rgba_dev, stream, false);
}

int atmo_render_composite(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *scene_rgba_dev, void *stream) {
    return render_impl(ctx, frame, depth_dev, scene_rgba_dev, stream, true);
}

// The geometric tile order of a cloudless frame seen from outside the atmosphere shell (rc.miss_k > 0: the projection's ray directions do not depend on depth and
// the camera is well outside): fills rc.geo_rows / geo_prefix / geo_first, or leaves geo_rows = 0 (taller or wider grids than the table holds, a silhouette that is
// not one run of columns per row, nothing or everything hit).  The view ray through pixel (x, y) is a(x) = A x + B(y), affine; shade_pixel's sure-miss test
// (c.a)^2 < k |a|^2 is a quadratic in x on every pixel row: the rays that can hit lie between its roots.  A tile row takes the union over its first, middle and
// last pixel row.  The ORDER depends on this arithmetic, the picture does not.
static void geo_order_fill(AtmoContext *ctx, atmo::RenderConsts &rc, int gx, int gy, int tile_h) {
    if (gy > atmo::GEO_MAX_ROWS || gx > 255 || (long long)gx * gy > 65535) return;
    AtmoContext::GeoCache &g = ctx->geo;
    float key[26];
    for (int i = 0; i < 16; ++i) key[i] = rc.inv_p[i];
    key[16] = rc.center[0]; key[17] = rc.center[1]; key[18] = rc.center[2]; key[19] = rc.miss_k;
    key[20] = (float)rc.x0; key[21] = (float)rc.y0; key[22] = (float)rc.x1; key[23] = (float)rc.y1; key[24] = (float)(gx * 1024 + tile_h); key[25] = rc.rcp_vw + 3.0f * rc.rcp_vh;
    if (g.rows < 0 || std::memcmp(key, g.key, sizeof key) != 0) {
        std::memcpy(g.key, key, sizeof key);
        g.rows = 0;
        const float *Q = rc.inv_p;
        const double sx = 2.0 * rc.rcp_vw, ox = 0.5 * sx - 1.0, sy = 2.0 * rc.rcp_vh, oy = 0.5 * sy - 1.0, k = rc.miss_k;
        const double c[3] = {rc.center[0], rc.center[1], rc.center[2]};
        const double A[3] = {Q[0] * sx, Q[1] * sx, Q[2] * sx};
        const double cA = c[0] * A[0] + c[1] * A[1] + c[2] * A[2], AA = A[0] * A[0] + A[1] * A[1] + A[2] * A[2];
        const double alpha = cA * cA - k * AA;
        bool ok = alpha < 0.0;   // an ellipse-like silhouette along the rows; anything else: no table
        unsigned total = 0;
        for (int r = 0; ok && r < gy; ++r) {
            const int py0 = rc.y0 + r * tile_h, py1 = std::min(py0 + tile_h, (int)rc.y1) - 1;
            const int ys[3] = {py0, (py0 + py1) >> 1, py1};
            double lo = 1e30, hi = -1e30;
            for (int q = 0; q < 3; ++q) {
                const double fny = ys[q] * sy + oy;
                const double B[3] = {Q[0] * ox + Q[4] * fny + Q[12], Q[1] * ox + Q[5] * fny + Q[13], Q[2] * ox + Q[6] * fny + Q[14]};
                const double cB = c[0] * B[0] + c[1] * B[1] + c[2] * B[2];
                const double AB = A[0] * B[0] + A[1] * B[1] + A[2] * B[2], BB = B[0] * B[0] + B[1] * B[1] + B[2] * B[2];
                const double beta = 2.0 * (cA * cB - k * AB), gamma = cB * cB - k * BB;
                const double disc = beta * beta - 4.0 * alpha * gamma;
                if (!(disc >= 0.0)) continue;                       // this pixel row misses everywhere
                const double sq = std::sqrt(disc), x1 = (-beta + sq) / (2.0 * alpha), x2 = (-beta - sq) / (2.0 * alpha);   // alpha < 0: x1 <= x2
                if (!(cA * 0.5 * (x1 + x2) + cB > 0.0)) continue;   // the cone BEHIND the camera
                lo = std::min(lo, x1);
                hi = std::max(hi, x2);
            }
            int first = 0, len = 0;
            if (lo <= hi && hi >= (double)rc.x0 && lo <= (double)(rc.x1 - 1)) {
                const int c0 = (int)std::floor((std::max(lo, (double)rc.x0) - rc.x0) / 16.0), c1 = (int)std::floor((std::min(hi, (double)(rc.x1 - 1)) - rc.x0) / 16.0);
                first = std::min(std::max(c0, 0), gx - 1);
                len = std::min(std::max(c1, first), gx - 1) - first + 1;
            }
            g.prefix[r] = (uint16_t)total;
            g.first[r] = (uint8_t)first;
            total += (unsigned)len;
        }
        if (ok && total > 0 && total < (unsigned)(gx * gy)) {
            g.prefix[gy] = (uint16_t)total;
            g.rows = gy;
            // hint[0][v]: the last row whose prefix is <= 256 v; hint[1][v]: the last row with r gx - prefix[r] (all-miss tiles above it) <= 256 v.  Both sequences
            // are non-decreasing in r: one merge pass each.  Entries beyond the last block repeat the last row.
            for (int part = 0; part < 2; ++part) {
                int r = 0;
                for (int v = 0; v < atmo::GEO_MAX_HINTS + 2; ++v) {
                    const unsigned lim = 256u * (unsigned)v;
                    while (r + 1 <= gy && (part == 0 ? (unsigned)g.prefix[r + 1] : (unsigned)((r + 1) * gx) - g.prefix[r + 1]) <= lim) ++r;
                    g.hint[part][v] = (uint16_t)r;
                }
            }
        }
    }
    if (g.rows > 0) {
        rc.geo_rows = g.rows;
        std::memcpy(rc.geo_prefix, g.prefix, (size_t)(g.rows + 1) * sizeof(uint16_t));
        std::memcpy(rc.geo_first, g.first, (size_t)g.rows * sizeof(uint8_t));
        std::memcpy(rc.geo_hint, g.hint, sizeof g.hint);
    }
}


int atmo_render_tiles(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, const uint32_t *tiles_dev, int n_tiles,
                      void *stream) {
    if (!ctx) return ATMO_E_ARG;
    if (n_tiles < 0 || (n_tiles > 0 && !tiles_dev)) return fail(ctx, ATMO_E_ARG, "atmo_render_tiles: bad tile list");
    if (n_tiles == 0) return ATMO_OK;  // an empty share of the frame: nothing to shade
    return render_impl(ctx, frame, depth_dev, rgba_dev, stream, false, tiles_dev, n_tiles);
}

int atmo_render_tiles_split(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, const uint32_t *tiles_dev, int n_tiles,
                            int n_heavy, void *stream) {
    if (!ctx) return ATMO_E_ARG;
    if (n_tiles < 0 || (n_tiles > 0 && !tiles_dev) || n_heavy < 0 || n_heavy > n_tiles) return fail(ctx, ATMO_E_ARG, "atmo_render_tiles_split: bad tile list");
    if (n_tiles == 0) return ATMO_OK;
    return render_impl(ctx, frame, depth_dev, rgba_dev, stream, false, tiles_dev, n_tiles, n_heavy);
}

int atmo_measure_tile_costs(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream,
                            uint32_t *cost_host, int capacity_tiles, int *tiles_x, int *tiles_y, int *tile_w, int *tile_h) {
    if (!ctx) return ATMO_E_ARG;
    if (!frame) return fail(ctx, ATMO_E_ARG, "atmo_measure_tile_costs: null frame");
    if (frame->x1 <= frame->x0 || frame->y1 <= frame->y0) return fail(ctx, ATMO_E_ARG, "atmo_measure_tile_costs: empty rect");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the grid this rect is drawn with (same choice as render_impl)
    atmo::RenderConsts probe;
    std::memset(&probe, 0, sizeof(probe));
    probe.x0 = frame->x0; probe.y0 = frame->y0; probe.x1 = frame->x1; probe.y1 = frame->y1;
    int probe_flags = 0, split = 1;
    bool lod_grid = false;
    launch_shape(ctx, frame, &probe_flags, &split, &lod_grid);
    probe.gx0 = lod_grid ? (frame->x0 & ~1) : frame->x0;
    probe.gy0 = lod_grid ? (frame->y0 & ~1) : frame->y0;
    int gx = 0, gy = 0;
    atmo::render_grid(probe, split, &gx, &gy);
    if (tiles_x) *tiles_x = gx;
    if (tiles_y) *tiles_y = gy;
    {
        int tw = 0, th = 0;
        atmo::render_tile_size(split, &tw, &th);  // what render_grid cuts the rect into (a build knob: ATMO_TILE_H)
        if (tile_w) *tile_w = tw;
        if (tile_h) *tile_h = th;
    }
    if (!cost_host) return ATMO_OK;
    if (capacity_tiles < gx * gy) return fail(ctx, ATMO_E_ARG, "atmo_measure_tile_costs: cost buffer too small (call with cost_host = NULL for the grid)");
    const size_t bytes = (size_t)gx * gy * sizeof(uint32_t);
    hipStream_t s = (hipStream_t)stream;
    if (ctx->measure_buf.bytes < bytes) {  // grow-only; every earlier measurement waited for its own copy-out before it returned (below)
        const int rc0 = dev_alloc(ctx, ctx->measure_buf, bytes);
        if (rc0 != ATMO_OK) return rc0;
    }
    uint32_t *d = (uint32_t *)ctx->measure_buf.ptr;
    hipError_t e = hipMemsetAsync(d, 0, bytes, s);
    int rc = ATMO_OK;
    if (e != hipSuccess) rc = hip_fail(ctx, e, "hipMemsetAsync");
    if (rc == ATMO_OK) {
        ctx->measure_cost = d;  // render_impl: record into this buffer, row-major order, no feedback bookkeeping
        rc = render_impl(ctx, frame, depth_dev, rgba_dev, stream, false);
        ctx->measure_cost = nullptr;
    }
    if (rc == ATMO_OK) {
        e = hipMemcpyAsync(cost_host, d, bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "atmo_measure_tile_costs");
    } else {
        (void)hipStreamSynchronize(s);
    }
    return rc;
}

static int render_impl(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream, bool composite,
                       const uint32_t *tiles_dev, int n_tiles, int n_heavy) {
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
    if (!(ctx->flags & (atmo::KF_LIGHT_DIRECT | atmo::KF_LITE)) && !(ctx->lut.ptr && ctx->lut4.ptr))
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
        rc.store_discards = 0;
    }
    hipStream_t s = (hipStream_t)stream;
    { const int rc0 = tex_order(ctx, s); if (rc0 != ATMO_OK) return rc0; }  // texture updated on another stream
    int split = 1, flags = 0;
    {   // the coverage cubemap's sampler: as declared (implicit LOD) when a mip chain is bound
        const char *why_not = nullptr;
        (void)resolve_sampler_lod(ctx, &why_not);
        if (ctx->sampler_lod == 1 && why_not) return fail(ctx, ATMO_E_STATE, why_not);
        bool lod = false;
        launch_shape(ctx, frame, &flags, &split, &lod);
        if (lod) {
            // the 2 x 2 quads are the viewport's: the grid starts on an even pixel, pixels in front of the rect are helper lanes
            rc.gx0 = frame->x0 & ~1;
            rc.gy0 = frame->y0 & ~1;
        }
    }
    int gx = 0, gy = 0;
    atmo::render_grid(rc, split, &gx, &gy);
    rc.tiles_x = gx;
    // default (-1): on for every variant since the sort no longer costs the draws anything (profiles/round2/ab_tile_feedback.txt)
    bool feedback = ctx->env_feedback >= 0 ? ctx->env_feedback != 0 : ctx->tile_feedback != 0;
    if (ctx->measure_cost) feedback = false;  // a measuring draw: plain row-major launch that records into the caller's buffer
    if (tiles_dev) feedback = false;          // a tile-list draw: the caller's order
    if (feedback && (long long)gx * gy < 512) feedback = false;  // tiny launches: nothing to schedule
    const bool geo_policy = feedback;   // (also inside a graph capture: the geometric order enqueues nothing of its own)
    if (feedback) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) feedback = false;  // no side-stream work inside a graph
    }
    // Every fb_period-th draw records the wave durations per tile; a sort on the side stream turns them into the next
    // order, which later draws pick up once a host-side event query says it is complete: no draw ever waits for a sort.
    bool fb_record = false, fb_instream = false;
    AtmoContext::FeedbackState *fb = nullptr;
    const uint32_t *order2 = nullptr, *class_totals = nullptr;   // of the order in use (heavy tiles on two lanes per ray)
    if (feedback) {
        const int rc1 = feedback_state(ctx, gx, gy, split, s, &fb);
        if (rc1 != ATMO_OK) return rc1;
    }
    // A moving camera (planet_atmosphere.gd:285-341 writes new matrices every frame; demo/avatar.gd, demo/mouse_look.gd): an order
    // is used fb_lag frames after the costs it was sorted from were measured.  The host predicts how far the picture's features
    // move in that time; the sort dilates the cost map by that distance, so a tile counts as cheap only if everything within
    // reach of it was cheap (atmo_tile_dilate_kernel); costs are recorded every 2nd draw instead of every fb_period-th while
    // the camera moves; and an order whose reach the motion has outrun is not used (row-major instead).
    // reach beyond which an order says nothing about the frame it would be used on: 160 px for the in-stream sort (a frame of lag),
    // 48 px for the side-stream sort (four to six frames of lag: measured, recording every 2nd frame without a usable order costs 1-2 %)
    constexpr float FB_STILL_PX = 0.5f, FB_MAX_REACH_PX = 160.0f, FB_MAX_REACH_SIDE_PX = 48.0f, FB_INSTREAM_PX = 3.0f, FB_INSTREAM_LONG_PX = 8.0f;
    int dil_rx = 0, dil_ry = 0;
    float reach_px = 0.0f;
    if (fb) {
        if (fb->have_prev) {
            float sil[2] = {0.0f, 0.0f};
            const float m = feedback_motion_px(*frame, fb->prev_frame, ctx->p.u_planet_radius + ctx->p.u_atmosphere_height,
                                               (ctx->flags & atmo::KF_CLOUDS) != 0, sil);
            fb->motion_px = std::fmax(m, 0.75f * fb->motion_px);  // peak hold: one still frame does not end a camera move
            if (fb->motion_px < 0.01f) fb->motion_px = 0.0f;
            for (int k = 0; k < 2; ++k) fb->sil_px[k] = std::fmax(sil[k], 0.75f * fb->sil_px[k]);
        }
        fb->prev_frame = *frame;
        fb->have_prev = true;
        const bool moving = fb->motion_px > FB_STILL_PX;
        const unsigned period = moving ? (ctx->fb_period < ctx->moving_period ? ctx->fb_period : ctx->moving_period) : ctx->fb_period;
        // what an order sorted now would have to cover: it is in use from ~2 frames after its recording draw until the next takes over
        const float want_reach = moving ? fb->motion_px * (float)(period + 4u) * ctx->env_reach_scale : 0.0f;
        // nothing measured now says anything about the frame it would order -- or the frames are so short (the baked-LUT atmosphere
        // without clouds: 20-50 us, +3 % from the order at best) that recording and sorting every other frame costs more than it brings
        const bool short_frames = !(flags & (atmo::KF_CLOUDS | atmo::KF_LIGHT_DIRECT));
        const bool too_fast = want_reach > FB_MAX_REACH_SIDE_PX || (moving && short_frames);
        if (fb->pending) {
            if (hipEventQuery(fb->ev_order[fb->write]) == hipSuccess) {
                fb->active = fb->write;  // complete: no stream-side wait needed
                fb->write ^= 1;
                fb->pending = false;
            } else {
                (void)hipGetLastError();  // hipErrorNotReady is an answer, not an error: keep it out of the launch checks below
            }
        }
        // In-stream mode: while the camera moves by more than a few pixels per frame, the kernels whose cost map is worth it
        // (raymarched cloud light: the heaviest tiles cost 10x the mean, frames of 0.4-1.3 ms, +48 % from the order on a still
        // camera) sort on the DRAW stream, right behind every draw.  The next draw is then ordered by this frame's costs -- one
        // frame of lag instead of four to six, so the dilation stays at a tile or two and the order keeps its meaning -- at the
        // price of ~10 us of sort kernels on the critical path per frame.  Measured (profiles/round3/ab_tile_feedback_motion.txt):
        // clouds_high_rm panning 1 degree per frame +20 % in-stream against +7 % with the side-stream sort, but 35 % against 37 %
        // at 0.1 degree per frame; clouds_high (0.18 ms frames, +7 % at best) loses 7 % in-stream: side stream only.
        // one frame of lag and one of margin -- of the silhouette's motion when the window is taken from it (below)
        const float is_reach = (ctx->fb_axis_windows ? std::fmax(fb->sil_px[0], fb->sil_px[1]) : fb->motion_px) * 2.0f * ctx->env_reach_scale;
        // Which kernels: raymarched cloud light from 3 px per frame; since round 4 (per-axis windows) also the other 64-step cloud kernels in
        // the precise mode (0.2 ms frames: +10..13 % where the side-stream order had nothing left, measured from 8 px per frame; at 1.4 px per
        // frame the side stream is 2-3 points better).  Shorter frames (`clouds`, the fast cloud mode: 0.12-0.16 ms) are neutral in-stream
        // (-0.4..+3 %) and stay on the side stream; the cloudless direct-light kernel LOSES 9-12 % in-stream under a pan (ATMO_FB_INSTREAM=2).
        const bool is_rm = (flags & atmo::KF_CLOUD_LIGHT_RM) != 0;
        const bool is_long = (flags & atmo::KF_CLOUDS) && (flags & atmo::KF_PRECISE) && ctx->cloud_steps >= 64;
        const bool is_kernel = ctx->instream == 2 ? (flags & (atmo::KF_CLOUDS | atmo::KF_LIGHT_DIRECT)) != 0 : (is_rm || (is_long && ctx->fb_axis_windows));
        const float is_px = is_rm ? FB_INSTREAM_PX : FB_INSTREAM_LONG_PX;
        fb_instream = ctx->instream && fb->motion_px >= is_px && is_kernel && !fb->pending && is_reach <= FB_MAX_REACH_PX;
        const int tile_h = (rc.y1 - rc.y0 + gy - 1) / gy;  // pixel rows per tile of this launch (8, or 4 with two lanes per ray)
        if (fb_instream) {
            if (fb->is_last_n + 1u == fb->n) {  // the sort behind the previous draw of this key wrote is_order
                rc.tile_order = (const uint32_t *)fb->is_order.ptr;
                order2 = (const uint32_t *)fb->is_order2.ptr;
                class_totals = fb->class_totals + 2 * atmo::TILE_ORDER_CLASSES;   // (read without waiting: at worst last frame's, or none yet)
                ctx->fb_ordered_draws += 1;
            }
            rc.tile_cost = (uint32_t *)fb->cost.ptr;
            reach_px = is_reach;
            dil_rx = (int)std::ceil(reach_px / 16.0f);
            dil_ry = (int)std::ceil(reach_px / (float)(tile_h > 0 ? tile_h : 8));
            if (ctx->fb_axis_windows) {
                // Round 4: the window per screen axis, from the motion of the planet's SILHOUETTE alone (two frames of it, one tile at least).  The
                // expensive tiles of these kernels sit on the limb, which an orbit leaves where it is while the surface points behind motion_px sweep
                // across the disc: the isotropic window (7 x 13 tiles at 1 degree of orbit per frame) buried the ranking of exactly those tiles, and
                // a pan needs nothing vertically (profiles/round4/ab_tile_feedback_motion.txt).
                dil_rx = (int)std::ceil(2.0f * fb->sil_px[0] * ctx->env_reach_scale / 16.0f);
                dil_ry = (int)std::ceil(2.0f * fb->sil_px[1] * ctx->env_reach_scale / (float)(tile_h > 0 ? tile_h : 8));
                dil_rx = dil_rx < 1 ? 1 : (dil_rx > 10 ? 10 : dil_rx);
                dil_ry = dil_ry < 1 ? 1 : (dil_ry > 10 ? 10 : dil_ry);
            }
            fb->active = -1;  // whatever the side stream sorted last belongs to an older picture
        } else {
            if (fb->active >= 0) {
                // still conservative?  features have moved about motion_px * (frames since the costs were measured)
                const float moved = fb->motion_px * (float)(fb->n - fb->order_born[fb->active]);
                if (moved <= fb->order_reach_px[fb->active] + 8.0f) {
                    rc.tile_order = (const uint32_t *)fb->order[fb->active].ptr;
                    order2 = (const uint32_t *)fb->order2[fb->active].ptr;
                    class_totals = fb->class_totals + fb->active * atmo::TILE_ORDER_CLASSES;   // complete: the host has seen this sort's event
                    ctx->fb_ordered_draws += 1;
                }
            }
            // the first two draws of a key are not measured (cold clocks and caches rank the tiles poorly); the next four
            // record back to back (the order settles in a few frames), then every period-th
            fb_record = !too_fast && !fb->pending && fb->n >= 2 && (fb->n < 6 || fb->n - fb->last_record >= period);
            if (fb_record) {
                rc.tile_cost = (uint32_t *)fb->cost.ptr;
                reach_px = want_reach;
                dil_rx = reach_px > 0.0f ? (int)std::ceil(reach_px / 16.0f) : 0;
                dil_ry = reach_px > 0.0f ? (int)std::ceil(reach_px / (float)(tile_h > 0 ? tile_h : 8)) : 0;
            }
        }
    }
    if (ctx->measure_cost) rc.tile_cost = ctx->measure_cost;
    // The direct-light cloudless kernels seen from outside the shell, when the learnt order has nothing for this draw -- a camera that moves too fast for it (a pan:
    // the silhouette slides a tile per frame), the first draws of a key, a draw inside a graph capture: the order is a closed form of the camera
    // (RenderConsts::geo_rows), looked up by the kernel's preamble.  That lookup costs every wave ~9 dependent scalar loads (+5 us on a 1920x1080 draw), which the
    // learnt order does not: so only where there is no learnt order (profiles/round6/geo_order.txt: under a pan 0.1046 -> 0.0979 ms; with it always on a still
    // camera loses 6 %, and the 20-50 us LUT kernels lose under every motion).
    int launch_flags = flags;
    if (geo_policy && ctx->geo_order && rc.tile_order == nullptr && flags == atmo::KF_LIGHT_DIRECT && rc.miss_k > 0.0f && split == 1) {
        geo_order_fill(ctx, rc, gx, gy, (rc.y1 - rc.y0 + gy - 1) / gy);
        if (rc.geo_rows > 0) {
            launch_flags |= atmo::KF_GEO;   // the twin kernel whose preamble looks the tile up (the plain kernel's preamble stays what it was)
            ctx->geo_draws += 1;
            ctx->fb_ordered_draws += 1;
        }
    }
    // kernel timing brackets the draw kernel alone (the tile-order kernel runs beside the previous draw).  The event pair
    // is owned by a guard until it is handed to ctx->pending, so no error path leaks it.
    struct EventPair {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
#ifdef ATMO_WAVE_TRACE
    {
        const size_t waves = (size_t)gx * gy * 4;  // upper bound for every tile height
        if (ctx->wave_trace_waves < waves) {
            int rc1 = dev_alloc(ctx, ctx->wave_trace, waves * 4 * sizeof(unsigned long long));
            if (rc1 != ATMO_OK) return rc1;
            ctx->wave_trace_waves = waves;
        }
        HIP_TRY(ctx, hipMemsetAsync(ctx->wave_trace.ptr, 0, waves * 4 * sizeof(unsigned long long), s));
        rc.wave_trace = (unsigned long long *)ctx->wave_trace.ptr;
    }
#endif
    if (tiles_dev) {   // (before anything is created or enqueued for this draw)
        // the bounded copy of the list lives in a context-owned buffer that a later, longer list frees and reallocates -- a graph that had
        // recorded this draw would replay on freed memory (and an allocation fails inside a global-mode capture anyway): refused, stated in atmo.h
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
            return fail(ctx, ATMO_E_STATE, "tile-list draws cannot be captured into a HIP graph (context-owned list buffer); capture atmo_render instead");
    }
    const bool timed = ctx->timing > 0 && (ctx->launch_counter % ctx->timing) == 0;
    if (timed) {
        if (ctx->pending.size() >= 64) drain_timing(ctx, /*only_completed=*/true);  // a long loop with timing left on stays bounded
        HIP_TRY(ctx, hipEventCreate(&ev.e0));
        HIP_TRY(ctx, hipEventCreate(&ev.e1));
        HIP_TRY(ctx, hipEventRecord(ev.e0, s));
    }
    if (tiles_dev) {
        // The list is the caller's: an index beyond the grid must shade nothing (include/atmo.h) -- and must not be turned into addresses.  The
        // render kernels' preamble is not the place (one scalar compare-and-branch there costs the headline kernel 10 %:
        // profiles/round5/ab_tile_bound.txt), so a copy of the list is bounded on the device in front of the draw: out-of-range entries
        // become the first tile BELOW the viewport, whose lanes all leave at the kernel's own bounds test.  One copy per draw stream
        // (stream order protects it from the next tile-list draw on that stream); a fifth stream takes over the least recently used copy,
        // ordered behind its last reader on the device.
        AtmoContext::TileListBuf *tl = nullptr;
        ctx->tile_list_clock += 1;
        for (AtmoContext::TileListBuf &t : ctx->tile_lists) if (!tl && t.used && t.stream == s) tl = &t;
        for (AtmoContext::TileListBuf &t : ctx->tile_lists) if (!tl && !t.used) tl = &t;
        if (!tl) {
            tl = &ctx->tile_lists[0];
            for (AtmoContext::TileListBuf &t : ctx->tile_lists) if (t.last_use < tl->last_use) tl = &t;
            const int rc0 = order_after_draws_on(ctx, s, tl->stream);  // its last reader: a draw on that stream
            if (rc0 != ATMO_OK) return rc0;
        }
        tl->used = true;
        tl->stream = s;
        tl->last_use = ctx->tile_list_clock;
        // the leading n_heavy tiles on two lanes per ray (atmo_render_tiles_split): only where that form exists and is the same bits
        // (as for whole frames: the raymarched-light kernel; the kernel without it gains only from eight GPUs on and loses 21 % at two --
        //  profiles/round5/band_balance.txt -- so it takes part only when forced, ATMO_HEAVY_SPLIT=2)
        const int lod_cloud0 = atmo::KF_CUBE_LOD | atmo::KF_PRECISE | atmo::KF_CLOUDS;
        const bool form = ctx->heavy_split == 2 ? (flags & ~atmo::KF_CLOUD_LIGHT_RM) == lod_cloud0 : flags == (lod_cloud0 | atmo::KF_CLOUD_LIGHT_RM);
        if (split != 1 || !form || !ctx->heavy_split) n_heavy = 0;
        { const int rc0 = dev_reserve(ctx, tl->buf, ((size_t)n_tiles + 2 * (size_t)n_heavy) * sizeof(uint32_t)); if (rc0 != ATMO_OK) return rc0; }
        int tw = 0, th = 0;
        atmo::render_tile_size(split, &tw, &th);                                         // pixels per tile of this launch
        const uint32_t below = (uint32_t)((frame->viewport_h - rc.gy0 + th - 1) / th);   // the first tile row that starts below the viewport's last row
        const int th2 = th / 2 > 0 ? th / 2 : 1;
        const uint32_t below2 = (uint32_t)((frame->viewport_h - rc.gy0 + th2 - 1) / th2);
        uint32_t *list2 = (uint32_t *)tl->buf.ptr + n_tiles;
        HIP_TRY(ctx, atmo::launch_tile_list_bound(tiles_dev, (uint32_t *)tl->buf.ptr, n_tiles, (uint32_t)gx * (uint32_t)gy, below * (uint32_t)gx, s,
                                                  list2, n_heavy, gx, below2 * (uint32_t)gx));
        rc.tile_order = (const uint32_t *)tl->buf.ptr;
        if (n_heavy > 0) order2 = list2;
    }
    // Heavy tiles on two lanes per ray, beside the rest of the draw (round 5).  Only the two BASELINE cloud kernels under the declared sampler have
    // the lane-split form whose frames are bit-identical to the one-lane kernel's (atmo_kernels.hip: march_clouds<.., SPLIT = 2, LOD>).
    // Measured (profiles/round5/ab_heavy_tile_split.txt): it pays where a draw is as long as its heaviest wavefront -- clouds_high_rm at
    // 1280x720 -28 %, from the limb -42..-44 % at 1280x720 and 1920x1080, the night side -33 % -- and costs 9 % where the draw is bound by
    // throughput (1920x1080 pose P_space, where the lane-split kernel on EVERY tile is 43 % slower; 3840x2160), which the trigger keeps out;
    // the kernel without raymarched light never gains (its heaviest wave is a fifth as long): not split unless forced (ATMO_HEAVY_SPLIT=2).
    int heavy = 0;
    const int lod_cloud = atmo::KF_CUBE_LOD | atmo::KF_PRECISE | atmo::KF_CLOUDS;
    const bool split_form = ctx->heavy_split == 2 ? (flags & ~atmo::KF_CLOUD_LIGHT_RM) == lod_cloud : flags == (lod_cloud | atmo::KF_CLOUD_LIGHT_RM);
    if (ctx->heavy_split && order2 && class_totals && rc.tile_order && !tiles_dev && split == 1 && split_form)
        heavy = heavy_tile_count(class_totals, gx * gy, ctx->heavy_split_ratio,
                                 ctx->heavy_split == 2 ? 0.0f : (fb_instream ? ctx->heavy_split_trigger_moving : ctx->heavy_split_trigger),
                                 1024 * 6);   // both declared-sampler cloud kernels hold six waves per SIMD
    if (tiles_dev) heavy = n_heavy;   // a tile-list draw: the caller says how many of its leading tiles are heavy (atmo_render_tiles_split)
    const int total_tiles = tiles_dev ? n_tiles : gx * gy;
    if (heavy > 0) {
        if (!ctx->split_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->split_stream, hipStreamNonBlocking));
        { const int rc0 = order_after_stream(ctx, ctx->split_stream, s); if (rc0 != ATMO_OK) return rc0; }   // fork: behind everything the draw is behind
        atmo::RenderConsts rc2 = rc;
        rc2.tile_order = order2;                       // two half-height tiles per heavy tile, heaviest first
        rc2.cost_rows_halved = 1;                      // ... whose costs belong to the one-lane grid's tile (only this launch: ADVICE r5)
        HIP_TRY(ctx, atmo::launch_render(flags, 2, rc2, ctx->split_stream, 2 * heavy));
        if (total_tiles - heavy > 0) {
            rc.tile_order += heavy;                    // the rest of the order, one lane per ray
            HIP_TRY(ctx, atmo::launch_render(flags, split, rc, s, total_tiles - heavy));
        }
        { const int rc0 = order_after_stream(ctx, s, ctx->split_stream); if (rc0 != ATMO_OK) return rc0; }   // join
        ctx->split_draws += 1;
        ctx->split_tiles_last = (unsigned)heavy;
    } else {
        HIP_TRY(ctx, atmo::launch_render(launch_flags, split, rc, s, tiles_dev ? n_tiles : 0));
    }
    ctx->last_flags = flags;
    if (fb_record) {
        // Sort on the side stream as soon as this draw is done (the sort also clears the costs for the next recording).
        // order[write] was last read by draws enqueued on `s` before this one, so the event orders the write too.
        HIP_TRY(ctx, hipEventRecord(fb->ev_draw, s));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->fb_stream, fb->ev_draw, 0));
        HIP_TRY(ctx, atmo::launch_tile_order((uint32_t *)fb->cost.ptr, (uint32_t *)fb->order[fb->write].ptr, gx, gy, dil_rx, dil_ry,
                                             (uint32_t *)fb->dil[0].ptr, (uint32_t *)fb->dil[1].ptr, (uint32_t *)ctx->fb_scratch.ptr, ctx->fb_stream,
                                             (uint32_t *)fb->order2[fb->write].ptr, fb->class_totals + fb->write * atmo::TILE_ORDER_CLASSES));
        fb->order_reach_px[fb->write] = reach_px;
        fb->order_born[fb->write] = fb->n;
        HIP_TRY(ctx, hipEventRecord(fb->ev_order[fb->write], ctx->fb_stream));
        fb->pending = true;
        fb->last_record = fb->n;
        ctx->fb_sorts += 1;
    }
    if (fb) fb->n += 1;
    hipEvent_t marker = nullptr;   // the draw stream's marker, when this draw recorded one (re-recorded behind the in-stream sort below)
    {   // remember the stream and put its marker behind this draw
        AtmoContext::DrawStream *ds = nullptr;
        for (AtmoContext::DrawStream &d : ctx->draw_streams) if (d.stream == s) ds = &d;
        if (!ds) {
            if (ctx->draw_streams.size() < 8) {
                ctx->draw_streams.emplace_back();
                ds = &ctx->draw_streams.back();
                ds->stream = s;
            } else {
                // a host that draws on a new stream every frame: bounded memory.  The least recently used entry makes room (its marker is
                // re-recorded for the new stream); what it stood for can only be waited for device-wide from now on: the next update does
                ds = &ctx->draw_streams[0];
                for (AtmoContext::DrawStream &d : ctx->draw_streams) if (d.last_use < ds->last_use) ds = &d;
                ds->stream = s;
                ds->recorded = false;
                ctx->draw_streams_many = true;
            }
        }
        ds->recorded = false;   // whatever marker it carries is not behind THIS draw
        ds->last_use = ++ctx->draw_stream_clock;
        const bool home = s == ctx->tex_stream;   // the stream of the last texture update (the null stream before the first)
        if (ctx->draw_events && (!home || ctx->draw_events == 2)) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
            if (!capturing) {   // (a draw inside a graph capture gets no marker: replays are the caller's to order)
                if (!ds->last_draw) HIP_TRY(ctx, hipEventCreateWithFlags(&ds->last_draw, hipEventDisableTiming));
                HIP_TRY(ctx, hipEventRecord(ds->last_draw, s));
                ds->recorded = true;
                marker = ds->last_draw;
            }
        }
    }
    ctx->last_split = split;
    ctx->launch_counter += 1;  // counted only once the launch was accepted
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(ev.e1, s));
        ctx->pending.emplace_back(ev.e0, ev.e1);
        ev.e0 = ev.e1 = nullptr;  // ownership moved
    }
    if (fb_instream) {  // behind the draw (and behind the timing bracket, which is the draw kernel's alone)
        HIP_TRY(ctx, atmo::launch_tile_order((uint32_t *)fb->cost.ptr, (uint32_t *)fb->is_order.ptr, gx, gy, dil_rx, dil_ry,
                                             (uint32_t *)fb->dil[0].ptr, (uint32_t *)fb->dil[1].ptr, (uint32_t *)fb->is_scratch.ptr, s,
                                             (uint32_t *)fb->is_order2.ptr, fb->class_totals + 2 * atmo::TILE_ORDER_CLASSES));
        fb->is_last_n = fb->n - 1;
        fb->last_record = fb->n - 1;
        ctx->fb_sorts += 1;
        // that sort reads and writes the feedback state (cost, is_order, dil[], class_totals) on this stream BEHIND the marker recorded above,
        // and whoever takes the state over (feedback_quiesce, order_after_draws_on: a recycled slot, a key handed to another stream) waits for the
        // marker only -- so the marker goes behind the sort (the stream handle is alive: we are inside the caller's draw call)
        if (marker) HIP_TRY(ctx, hipEventRecord(marker, s));
    }
    return ATMO_OK;
}

int atmo_set_precision(AtmoContext *ctx, int mode) {
    if (!ctx) return ATMO_E_ARG;
    if (mode < 0 || mode > 2) return fail(ctx, ATMO_E_ARG, "atmo_set_precision: mode must be 0 (fast), 1 (precise) or 2 (precise, and the v2 atmosphere march of the no-cloud variants in reference order)");
    // 1: reference operation order for the cloud density and the v1 march (their default); 2: also for the v2 atmosphere march
    // (march_atmosphere_v2_precise; the v1 variants have no other form), which modes 0 and 1 leave in its fast form
    if (mode >= 1 && (ctx->flags & (atmo::KF_CLOUDS | atmo::KF_LITE))) ctx->flags |= atmo::KF_PRECISE;
    else ctx->flags &= ~atmo::KF_PRECISE;
    if (mode == 2 && !(ctx->flags & atmo::KF_LITE)) ctx->flags |= atmo::KF_ATMO_REF;
    else ctx->flags &= ~atmo::KF_ATMO_REF;
    return ATMO_OK;
}

int atmo_set_host_double_precision(AtmoContext *ctx, int enable) {
    if (!ctx) return ATMO_E_ARG;
    ctx->host_double_precision = enable ? 1 : 0;
    return ATMO_OK;
}

int atmo_set_target_cleared(AtmoContext *ctx, int cleared) {
    if (!ctx) return ATMO_E_ARG;
    ctx->target_cleared = cleared ? 1 : 0;
    return ATMO_OK;
}

int atmo_set_tile_feedback(AtmoContext *ctx, int mode) {
    if (!ctx) return ATMO_E_ARG;
    if (mode < -1 || mode > 1) return fail(ctx, ATMO_E_ARG, "atmo_set_tile_feedback: -1 (by variant), 0 (off) or 1 (on)");
    ctx->tile_feedback = mode;
    // restart every feedback state at its next launch.  Buffers are kept and nothing waits here: a slot whose draws or sort may still be
    // in flight is marked dirty, and whoever takes it next is ordered behind that work on the device (feedback_state -> feedback_quiesce)
    for (AtmoContext::FeedbackState &f : ctx->fb) {
        if (f.used && (f.n > 0 || f.pending)) f.dirty = true;   // (f.pending stays: the quiesce orders the new owner behind the sort stream)
        f.used = false;
        f.n = 0;
    }
    ctx->fb_budget = 8;
    return ATMO_OK;
}

float atmo_debug_motion_px(const AtmoFrame *a, const AtmoFrame *b, float radius, int surface_points) {
    if (!a || !b) return -1.0f;
    return feedback_motion_px(*a, *b, radius, surface_points != 0);
}

int atmo_get_feedback_stats(AtmoContext *ctx, int *states, unsigned *ordered_draws, unsigned *sorts, unsigned *recycled) {
    if (!ctx) return ATMO_E_ARG;
    int n = 0;
    for (const AtmoContext::FeedbackState &f : ctx->fb) n += f.used ? 1 : 0;
    if (states) *states = n;
    if (ordered_draws) *ordered_draws = ctx->fb_ordered_draws;
    if (sorts) *sorts = ctx->fb_sorts;
    if (recycled) *recycled = ctx->fb_recycled;
    return ATMO_OK;
}

int atmo_set_lane_split(AtmoContext *ctx, int lanes_per_ray) {
    if (!ctx) return ATMO_E_ARG;
    if (lanes_per_ray < 0 || lanes_per_ray > 2) return fail(ctx, ATMO_E_ARG, "atmo_set_lane_split: 0 (auto), 1 or 2 lanes per ray");
    ctx->lane_split = lanes_per_ray;
    return ATMO_OK;
}

int atmo_set_timing(AtmoContext *ctx, int every_kth) {
    if (!ctx) return ATMO_E_ARG;
    (void)hipSetDevice(ctx->device);
    drain_timing(ctx);
    ctx->timing = every_kth > 0 ? every_kth : 0;
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

int atmo_debug_marched_optical_depth(AtmoContext *ctx, int n, const float *pos_xyz, const float *dir_xyz, int light_steps, float *out) {
    if (!ctx) return ATMO_E_ARG;
    if (n < 1 || !pos_xyz || !dir_xyz || !out || light_steps < 1 || light_steps > 4096)
        return fail(ctx, ATMO_E_ARG, "atmo_debug_marched_optical_depth: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n * 3 * sizeof(float);
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, 2 * bytes + (size_t)n * sizeof(float)));
    float *dpos = d, *ddir = d + (size_t)n * 3, *dout = d + (size_t)n * 6;
    hipError_t e = hipMemcpy(dpos, pos_xyz, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ddir, dir_xyz, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = atmo::launch_light_probe(dpos, ddir, n, ctx->p.u_planet_radius, ctx->p.u_atmosphere_height, ctx->p.u_density,
                                                      light_steps, dout, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "atmo_debug_marched_optical_depth");
    return ATMO_OK;
}

int atmo_debug_log2_cr(AtmoContext *ctx, int n, const float *x, float *out) {
    if (!ctx) return ATMO_E_ARG;
    if (n < 1 || !x || !out) return fail(ctx, ATMO_E_ARG, "atmo_debug_log2_cr: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, 2 * (size_t)n * sizeof(float)));
    hipError_t e = hipMemcpy(d, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = atmo::launch_log2_cr(d, d + n, n, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(ctx, e, "atmo_debug_log2_cr");
    return ATMO_OK;
}

const char *atmo_kernel_name(AtmoContext *ctx) {
    if (!ctx) return "";
    if (ctx->last_flags >= 0) return atmo::render_kernel_name(ctx->last_flags, ctx->light_steps, ctx->last_split);
    // before the first draw: what a draw would launch as the context stands (render_impl's choices: long view marches, the cubemap's sampler)
    int flags = 0, split = 1;
    launch_shape(ctx, nullptr, &flags, &split, nullptr);
    return atmo::render_kernel_name(flags, ctx->light_steps, split);
}

const char *atmo_last_error_string(AtmoContext *ctx) {
    if (!ctx) return g_create_error.c_str();
    return ctx->err.c_str();
}

int atmo_get_split_stats(AtmoContext *ctx, unsigned *split_draws, unsigned *heavy_tiles_last) {
    if (!ctx) return ATMO_E_ARG;
    if (split_draws) *split_draws = ctx->split_draws;
    if (heavy_tiles_last) *heavy_tiles_last = ctx->split_tiles_last;
    return ATMO_OK;
}

int atmo_get_host_wait_stats(AtmoContext *ctx, unsigned *device_syncs) {
    if (!ctx) return ATMO_E_ARG;
    if (device_syncs) *device_syncs = ctx->device_syncs;
    return ATMO_OK;
}

// ---- the host side without a device (atmo_debug.h): for the CPU test suite and the sanitizer build ----------------------------------
int atmo_debug_create_host_only(int variant, int view_steps, int cloud_steps, int light_mode, int light_steps, AtmoContext **out) {
    { const int rc0 = check_create_args(variant, view_steps, cloud_steps, light_mode, light_steps, out); if (rc0 != ATMO_OK) return rc0; }
    AtmoContext *ctx = new (std::nothrow) AtmoContext();
    if (!ctx) return fail(nullptr, ATMO_E_ARG, "atmo_debug_create_host_only: out of host memory");
    init_variant(ctx, /*device: none*/ -1, variant, view_steps, cloud_steps, light_mode, light_steps);
    *out = ctx;
    return ATMO_OK;
}

int atmo_debug_frame_constants(AtmoContext *ctx, const AtmoFrame *frame, int cube_n, float *out, int capacity, int *count) {
    if (!ctx) return ATMO_E_ARG;
    if (!frame) return fail(ctx, ATMO_E_ARG, "atmo_debug_frame_constants: null frame");
    if (cube_n < 0 || cube_n > 4096) return fail(ctx, ATMO_E_ARG, "atmo_debug_frame_constants: bad cubemap size");
    const int saved_n = ctx->cube_n;
    ctx->cube_n = cube_n;  // the level-0 certificate's constant depends on the face size; no texture needs to be bound for it
    AtmoFrame fixed = *frame;
    if (ctx->host_double_precision) for (int k = 12; k < 15; ++k) fixed.inv_view_matrix[k] *= -1.0f;
    atmo::RenderConsts rc;
    fill_consts(ctx, &fixed, nullptr, nullptr, rc);
    ctx->cube_n = saved_n;
    std::vector<float> v;
    auto put = [&](const float *p, int n) { v.insert(v.end(), p, p + n); };
    auto put1 = [&](float x) { v.push_back(x); };
    put(rc.cam_pos_world, 3); put(rc.sun_dir, 3); put1(rc.atmosphere_radius); put(rc.coeff, 3);                       //  0 .. 9
    put1(rc.clouds_bottom); put1(rc.clouds_top); put1(rc.cloud_thickness); put1(rc.inv_cloud_thickness);             // 10 .. 13
    put1(rc.layer_r2_lo); put1(rc.layer_r2_hi); put1(rc.shape_lo01); put1(rc.shape_hi01);                            // 14 .. 17
    put(rc.view_to_model, 16); put(rc.origin_model, 3); put(rc.sun_dir_model, 3);                                    // 18 .. 39
    put1(rc.max_d); put1(rc.inv_cloud_steps); put(rc.rm_offset, 6); put(rc.rm_weight, 6); put(&rc.rm_tap[0][0], 18); // 40 .. 71
    put1(rc.lod0_inv_c); put1(rc.lod0_last); put1(rc.lod0_drift); put1(rc.miss_k); put1(rc.rcp_vw); put1(rc.rcp_vh); // 72 .. 77
    put1((float)rc.shape_invert); put1((float)rc.cube_lod_fast); put1((float)rc.store_discards);                     // 78 .. 80
    if (count) *count = (int)v.size();
    if (!out) return ATMO_OK;
    if (capacity < (int)v.size()) return fail(ctx, ATMO_E_ARG, "atmo_debug_frame_constants: buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(float));
    return ATMO_OK;
}

}  // extern "C"

#ifdef ATMO_WAVE_TRACE
// diagnostic build only: copies the wave trace of the last draw (4 x uint64 per wave) to the host; returns the wave count
extern "C" long long atmo_debug_wave_trace(AtmoContext *ctx, unsigned long long *host, long long max_waves) {
    if (!ctx || !ctx->wave_trace.ptr) return 0;
    const long long n = (long long)ctx->wave_trace_waves < max_waves ? (long long)ctx->wave_trace_waves : max_waves;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(host, ctx->wave_trace.ptr, (size_t)n * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#endif
