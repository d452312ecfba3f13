/*
 * atmo.h -- C ABI of libatmo_hip.so: the MI355X (gfx950) implementation of the per-pixel
 * atmosphere / volumetric-cloud raymarch of Zylann/godot_atmosphere_shader.
 *
 * The reference has no FFI: the path sits behind Godot's shader-uniform interface.  Each entry
 * point below names the reference interface it replaces (paths relative to
 * /root/reference/addons/zylann.atmosphere/).  A GDExtension (or any other host) binds exactly
 * these symbols; see INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions: opaque handle, int error codes (0 = ATMO_OK), no exceptions or aborts across the
 * boundary, caller owns every buffer it passes, one context per GPU, a context is not thread-safe
 * but distinct contexts are independent.  All matrices are 16 floats, column-major (GLSL/Godot
 * memory order); mat2 is 4 floats column-major.  Colours are linear (the host applies Godot's
 * `source_color` sRGB->linear conversion).  Device pointers are plain HIP device addresses.
 */
#ifndef ATMO_H
#define ATMO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATMO_ABI_VERSION 4

typedef struct AtmoContext AtmoContext;

enum AtmoError {
    ATMO_OK = 0,
    ATMO_E_NAME = 1,     /* unknown uniform / texture name (Godot ignores these silently; the ABI reports them) */
    ATMO_E_ARG = 2,      /* bad count, size, null pointer, unsupported step count */
    ATMO_E_STATE = 3,    /* render without a LUT / required texture */
    ATMO_E_HIP = 4,      /* HIP runtime error; see atmo_last_error_string */
    ATMO_E_NO_DEVICE = 5 /* no gfx950 device / device index out of range */
};

/* Shader variants: shaders/planet_atmosphere_{no_clouds,clouds,clouds_high,clouds_high_rm}.gdshader:4-7 */
enum AtmoVariant {
    ATMO_VARIANT_NO_CLOUDS = 0,      /* ATMOSPHERE_RAYMARCH_STEPS 8 */
    ATMO_VARIANT_CLOUDS = 1,         /* + CLOUDS_ENABLED, CLOUDS_MAX_RAYMARCH_STEPS 32 */
    ATMO_VARIANT_CLOUDS_HIGH = 2,    /* + CLOUDS_MAX_RAYMARCH_STEPS 64 */
    ATMO_VARIANT_CLOUDS_HIGH_RM = 3, /* + CLOUDS_RAYMARCHED_LIGHTING (README's "clouds_high_m") */
    /* ATMOSPHERE_LITE variants, shaders/planet_atmosphere_v1_{no_clouds,clouds,clouds_high}.gdshader:4-7:
     * compute_atmosphere of shaders/include/atmosphere_funcs_v1.gdshaderinc, ATMOSPHERE_RAYMARCH_STEPS 16, no LUT */
    ATMO_VARIANT_V1_NO_CLOUDS = 4,
    ATMO_VARIANT_V1_CLOUDS = 5,      /* CLOUDS_MAX_RAYMARCH_STEPS 32 */
    ATMO_VARIANT_V1_CLOUDS_HIGH = 6  /* CLOUDS_MAX_RAYMARCH_STEPS 64 */
};

/* How the sun-ray optical depth of compute_atmosphere_v2 is obtained. */
enum AtmoLightMode {
    ATMO_LIGHT_LUT = 0,    /* reference: get_baked_optical_depth, shaders/include/atmosphere_funcs_v2.gdshaderinc:14-29 */
    ATMO_LIGHT_DIRECT = 1  /* inline light march of `light_steps` samples (BASELINE "N view x M light steps");
                              the integrand of shaders/optical_depth.gdshader:17-31 from the sample position */
};

/* Texture formats accepted by atmo_set_texture. */
enum AtmoTextureKind {
    ATMO_TEX_2D_R32F = 0,  /* u_optical_depth_texture (what optical_depth_baker.gd:75-80 uploads) */
    ATMO_TEX_2D_R8 = 1,    /* u_blue_noise_texture, 256x256 */
    ATMO_TEX_3D_R8 = 2,    /* u_cloud_shape_texture, n^3, x fastest */
    ATMO_TEX_CUBE_R8 = 3   /* u_cloud_coverage_cubemap, 6 faces +X,-X,+Y,-Y,+Z,-Z of n^2 per mip level */
};

enum AtmoMemory { ATMO_MEM_HOST = 0, ATMO_MEM_DEVICE = 1 };

/*
 * Per-frame arguments: the parameters of atmosphere_fragment
 * (shaders/include/planet_atmosphere_main.gdshaderinc:106-117) that are constant over a draw, plus
 * the rect of the viewport this call shades (row-band / tile sharding; the whole viewport when
 * x0=y0=0, x1=viewport_w, y1=viewport_h).
 */
typedef struct AtmoFrame {
    float inv_projection_matrix[16]; /* INV_PROJECTION_MATRIX: (SCREEN_UV*2-1, depth, 1) -> view space */
    float inv_view_matrix[16];       /* INV_VIEW_MATRIX */
    int32_t viewport_w, viewport_h;  /* VIEWPORT_SIZE */
    float planet_center_viewspace[3];/* varying v_planet_center_viewspace (atmosphere_vertex, main:101-102) */
    float sun_center_viewspace[3];   /* varying v_sun_center_viewspace (main:103) */
    float time;                      /* TIME (cloud_funcs.gdshaderinc:298; dead in the shipped shaders) */
    int32_t x0, y0, x1, y1;          /* rect to shade, in pixels */
} AtmoFrame;

/* ABI version of the loaded library (== ATMO_ABI_VERSION of the header it was built from). */
int atmo_abi_version(void);

/* Number of usable gfx950 devices, or a negative AtmoError. Does not create a context. */
int atmo_device_count(void);

/*
 * Replaces: assigning a shader variant to the node (`custom_shader`, planet_atmosphere.gd:118-141) and
 * its compile-time #defines (shaders/planet_atmosphere_*.gdshader:4-7).
 * view_steps = ATMOSPHERE_RAYMARCH_STEPS (0 => the variant's shipped value: 8, or 16 for the v1 variants); cloud_steps =
 * CLOUDS_MAX_RAYMARCH_STEPS (0 => shipped 32/64); light_mode/light_steps: see AtmoLightMode
 * (light_steps ignored for ATMO_LIGHT_LUT).  Uniforms start at the shader defaults
 * (SURVEY.md 8b); u_blue_noise_texture starts all-zero, u_cloud_coverage_cubemap unset (= 1.0).
 */
int atmo_create(int device, int variant, int view_steps, int cloud_steps, int light_mode, int light_steps,
                AtmoContext **out);

int atmo_destroy(AtmoContext *ctx);

/*
 * Replaces: ShaderMaterial.set_shader_parameter(name, value) for float / vecN / matN uniforms
 * (planet_atmosphere.gd:106-108,114-115,175-176,216,235,250,331,336,340).  n = number of floats
 * (1, 2, 3, 4 or 16) and must match the uniform's type.  Names are the reference's uniform names.
 */
int atmo_set_param_f32(AtmoContext *ctx, const char *name, const float *v, int n);
int atmo_get_param_f32(AtmoContext *ctx, const char *name, float *v, int n);

/*
 * Replaces: ShaderMaterial.set_shader_parameter(name, Texture) for u_optical_depth_texture
 * (planet_atmosphere.gd:156), u_blue_noise_texture (:107), u_cloud_shape_texture and
 * u_cloud_coverage_cubemap (user-set).  The data is copied; `memory` says where `data` lives.
 * data == NULL unsets the texture (cubemap => constant 1.0).  2-D: w x h (u_optical_depth_texture: 1..1023 per side, the
 * reference bakes 256 x 256); 3-D: w=h=d=n; cube: w=h=n, d=6.
 * mips (cubemap only; 0 or 1 for the others): number of mip levels in `data`, packed level after level (level l = 6 faces
 * of (n >> l)^2 texels); 1 = level 0 only; 0 = level 0 given, the rest of the chain generated on the device with the 2x2
 * box filter Image.generate_mipmaps applies to L8 (noise_cubemap.gd:107,135).  Levels above 0 are read by the declared
 * linear-mipmap sampler (atmo_set_sampler_lod, default), not in its LOD-0 mode.
 * The copy and the re-layout into the kernels' footprint layouts are enqueued on `stream` (hipStream_t, NULL = default
 * stream).  Updates of one context take effect in call order whatever streams they arrive on (a later update is chained
 * behind an earlier one's event), and draws on other streams wait for them (stream-side).  An update that arrives on a
 * stream other than the ones the context has drawn on is ordered behind those draws ON THE DEVICE (they may still be
 * reading the bound copy): the context records a marker event behind every draw it enqueues on a stream other than its
 * "home" stream -- the stream of its most recent texture update (the null stream before the first) -- and the update's
 * stream waits for those markers; no queue of the process other than the update's own is held up, and a draw stream
 * the caller has destroyed in the meantime is never touched again.  Nothing waits on the host except: a copy from
 * pageable host memory; the re-allocation when a texture changes size; and, device-wide (hipDeviceSynchronize), an
 * update that moves AWAY from a non-null home stream which has carried draws since (those have no marker, and a stream
 * that may be gone cannot be asked) or a context that has drawn on more than 8 streams since its last update.  A host
 * that sends updates down its draw stream pays nothing at all: stream order is all there is (a marker costs 3-4 us
 * per draw; a host with a separate upload stream pays that on its draws).
 * Device memory per bound texture: 4 bytes per texel position for the two cloud textures (bilinear footprints) plus, up to
 * 1024^2 faces (cubemap, whole chain) and 64^3 (shape volume; 48^3 until round 6), a float copy of the same footprints at 16 bytes each, which
 * the precise cloud kernels sample (same bits, fewer instructions); larger textures are sampled from the 4-byte footprints.
 */
int atmo_set_texture(AtmoContext *ctx, const char *name, int kind, int w, int h, int d, int mips,
                     const void *data, int memory, void *stream);

/* Size of the texture bound to a texture uniform (0 when unset): 2-D w x h (d = 1), 3-D n^3, cube n x n x 6 and its mips. */
int atmo_get_texture_size(AtmoContext *ctx, const char *name, int *w, int *h, int *d, int *mips);

/*
 * Replaces: the sampler state of `uniform samplerCube u_cloud_coverage_cubemap` (cloud_funcs.gdshaderinc:15,45: no filter hint,
 * i.e. the default linear-mipmap filter; NoiseCubemap builds the chain, noise_cubemap.gd:107,135).
 * -1 (default) = as declared: whenever the bound cubemap has a mip chain (atmo_set_texture mips != 1) and the precise cloud
 *   kernels are selected (atmo_set_precision >= 1, the default), `texture(cubemap, dir)` takes its implicit level of detail:
 *   every 2x2 pixel quad differences the cube directions of its rays at the same march step (what the fragment pipeline's
 *   derivatives are), transforms them to the selected face (Vulkan 1.3 cube-map derivative transformation),
 *   lambda = log2(max(rho_x, rho_y)) clamped to the bound levels, and the sample is the linear mix of the seamless bilinear
 *   samples of the two nearest levels; exact rule in oracle/atmo_oracle.h.  A partner pixel's ray is a function of that pixel
 *   alone, so the picture does not depend on which pixels share a wavefront or on the launch rect.  With a single level bound,
 *   or in the fast cloud mode (atmo_set_precision 0), level 0 is sampled.
 *  0 = level 0 only, whatever is bound (the stated convention of rounds 1-3; 1.5-2x faster on the cloud variants).
 *  1 = as -1, but a draw that cannot use the implicit LOD although a chain is bound (fast cloud mode) fails with ATMO_E_STATE.
 */
int atmo_set_sampler_lod(AtmoContext *ctx, int mode);

/*
 * Replaces: OpticalDepthBaker (optical_depth_baker.gd:37-85) + shaders/optical_depth.gdshader:45-68:
 * bakes the 256x256, 64-sample optical-depth LUT on the device from the current u_planet_radius,
 * u_atmosphere_height, u_density and binds it as u_optical_depth_texture (no viewport, no readback).
 */
int atmo_bake_optical_depth(AtmoContext *ctx, void *stream);

/*
 * Replaces: NoiseCubemap._generate_images (noise_cubemap.gd:101-140), the CPU triple loop the reference calls
 * "really slow" (:100): 6 x resolution^2 L8 texels, density = 0.5 + 0.5 * noise(direction * scale), same
 * texel -> direction mapping (:110-128).  The noise is this library's seeded fractal value noise (the reference calls
 * Godot's FastNoiseLite, which is engine code): `seed`, `frequency`, `octaves`, `gain` play the roles of the Noise
 * resource's properties; scale3 is NoiseCubemap.scale.  bind != 0 sets the result as u_cloud_coverage_cubemap;
 * faces_host (may be NULL) receives the 6 faces (+X,-X,+Y,-Y,+Z,-Z), rows top to bottom -- lay them out 3 x 2 for
 * generate_importable_image (:143-155).  kernel_ms (may be NULL) receives the generator kernel's device time.
 */
int atmo_generate_noise_cubemap(AtmoContext *ctx, int resolution, uint32_t seed, float frequency, int octaves, float gain,
                                const float *scale3, int bind, uint8_t *faces_host, double *kernel_ms);

/* Copy the currently bound LUT (w*h floats) to host memory; also writes the RGBA8 packing of
 * shaders/optical_depth.gdshader:33-43 when rgba8 != NULL (w*h*4 bytes). For hosts that want to hand the
 * LUT back to a Godot ImageTexture. Synchronises the stream. */
int atmo_read_optical_depth(AtmoContext *ctx, float *lut_host, uint8_t *rgba8_host, int capacity_texels, void *stream);

/*
 * Replaces: one draw of the atmosphere mesh, i.e. fragment() of shaders/planet_atmosphere_*.gdshader:20-27
 * calling atmosphere_fragment for every pixel of the rect.
 *   depth_dev: device pointer, viewport_h rows of viewport_w floats (u_depth_texture, reversed-Z).
 *   rgba_dev:  device pointer, (y1-y0) rows of (x1-x0) RGBA float4: ALBEDO.rgb, ALPHA; discarded
 *              fragments are written as (0,0,0,0).  Must be 16-byte aligned.
 *   stream:    hipStream_t (NULL = default stream).  The call only enqueues work.
 */
int atmo_render(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream);

/*
 * Replaces: the same draw *including the renderer's blend stage* (SURVEY.md 8f row 4): shades the rect and blends
 * ALBEDO/ALPHA over the scene colour buffer in place, as Godot does for an unshaded `blend_mix` spatial material
 * (colour: SRC_ALPHA, ONE_MINUS_SRC_ALPHA; alpha: ONE, ONE_MINUS_SRC_ALPHA).  Discarded fragments leave the
 * buffer untouched.  scene_rgba_dev: viewport_h rows of viewport_w RGBA float4 (the whole viewport, whatever the rect).
 */
int atmo_render_composite(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *scene_rgba_dev, void *stream);

/*
 * Sharding aid (no reference counterpart; BASELINE north_star: "independent framebuffer tiles shard across the GPUs of one node"): draws
 * only the listed pixel tiles of the rect, in list order.  tiles_dev: n_tiles indices (device memory, stream-ordered) into the launch grid
 * atmo_measure_tile_costs reports for this rect, row-major (tile t covers pixel columns x0 + (t % tiles_x) * tile_w .. and rows
 * y0 + (t / tiles_x) * tile_h ..., clipped by the rect; with the declared cubemap sampler the grid starts on the even pixel at or before
 * (x0, y0)).  rgba_dev is addressed exactly as in atmo_render -- (y1-y0) rows of (x1-x0) pixels -- and only the listed tiles' pixels are
 * written: N GPUs given a partition of the tile list into N lists produce, between them, the frame atmo_render draws, bit for bit.
 * An index beyond the grid shades nothing (its pixels lie outside the rect).
 * Tile-order feedback does not apply (the list is the order: put the heaviest tiles first).
 */
int atmo_render_tiles(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, const uint32_t *tiles_dev, int n_tiles,
                      void *stream);
/*
 * atmo_render_tiles with the list's first n_heavy tiles drawn on two lanes per ray beside the rest (what `bench.py --shard tiles` and
 * sharding.heavy_tiles use: product API of the tile-sharded path; in atmo_debug.h until round 6): a GPU's share of ONE frame is as long as its
 * heaviest wavefront from two GPUs on (profiles/round4/band_balance.txt), which is the regime where the lane-split kernel pays.  The caller orders
 * its list heaviest first and picks n_heavy from the measured costs.  Ignored (n_heavy = 0) where the kernel family has no bit-identical
 * lane-split form: the frame is the same bits either way.
 * Both tile-list draws keep a bounded copy of the caller's list in a buffer the CONTEXT owns and grows on demand, so neither may be recorded into a
 * HIP graph: on a capturing stream they return ATMO_E_STATE (a replay would read a buffer a later, longer list has freed; atmo_render allocates
 * nothing and can be captured).
 */
int atmo_render_tiles_split(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, const uint32_t *tiles_dev, int n_tiles,
                            int n_heavy, void *stream);

/*
 * Sharding aid (no reference counterpart: the reference is single-GPU): draws the rect like atmo_render and
 * returns what every pixel tile of that draw cost -- the longest of its wavefronts, in shader cycles, row-major over the
 * launch grid (tiles_x x tiles_y tiles of tile_w x tile_h pixels, the last column / row clipped by the rect).  A host that
 * cuts one viewport into row bands for several GPUs sums these per row and cuts at equal cost (sharding.balanced_row_bands)
 * instead of guessing from geometry.  cost_host == NULL only reports the grid.  Waits for the draw (a set-up call).
 */
int atmo_measure_tile_costs(AtmoContext *ctx, const AtmoFrame *frame, const float *depth_dev, float *rgba_dev, void *stream,
                            uint32_t *cost_host, int capacity_tiles, int *tiles_x, int *tiles_y, int *tile_w, int *tile_h);

/*
 * Numerical mode of the cloud kernels, of the v1 ("lite") atmosphere and -- mode 2 -- of the v2 atmosphere march (no reference
 * counterpart).
 * 1 (default, precise): the whole cloud density expression, both texture filters included, is evaluated in the
 *   reference's operation order with exact UNORM8 conversions, so the x50 density ramp sees bit-identical inputs; the
 *   deviation from a scalar fp32 evaluation of the GDShader is that of the atmosphere term (<= 2.1e-5) on every
 *   planet scale tested (tests/test_gpu_parity.py::test_parity_other_planet_scales).  The v1 atmosphere march runs in the
 *   reference's operation order as well (unfused, IEEE sqrt / divide): needed outside the model's range, where
 *   density * step_len > 1 makes the product of (1 - density * step_len) amplify rounding to > 1e-4 relative.
 * 0 (fast): the well-conditioned part of the density expression runs fused: ~15 % more cloud-kernel throughput, max
 *   deviation 5.1e-5 at 1920x1080 and 6.9e-5 at 3840x2160 on the demo scene, but it grows with u_cloud_density_scale
 *   (1.8e-4 at 10x the demo's value).  VERDICT: THIS MODE IS OUTSIDE THE 1e-4 CONTRACT -- it holds on the demo scene and is not promised elsewhere
 *   (the test suite bars it at 3e-4 on density_scale x 10); modes 1 and 2 are the ones the contract is stated for.
 * 2: as 1, and the v2 atmosphere march itself runs in the reference's operation order (view-space
 *   position accumulated and the centre subtracted at every use, alpha built step by step, IEEE sqrt / divide, expf,
 *   unfused): a fifth (direct light march) to a half (8 view steps, baked LUT) of the default form's throughput on the
 *   no-cloud variants, 7-10 % less on the cloud variants, deviation
 *   from a scalar fp32 evaluation of the GDShader below 1e-6 whatever the step count.  The default form's running sums drift with the number of view steps (up to 1.1e-4 of alpha
 *   at 64 steps on a thin atmosphere); modes 0 and 1 leave it in place.  The v1 variants have had the reference order since mode 1.
 */
int atmo_set_precision(AtmoContext *ctx, int mode);

/*
 * Replaces: the compile switch `#define DOUBLE_PRECISION` (shaders/include/planet_atmosphere_main.gdshaderinc:25,118-125).
 * A double-precision Godot build hands INV_VIEW_MATRIX with its origin negated (godotengine/godot#93108); with the
 * switch on, the library negates inv_view_matrix[12..14] of every AtmoFrame back, exactly as the shader does, so such
 * a host passes the engine's matrix unchanged.  0 (default) = single-precision engine build.
 */
int atmo_set_host_double_precision(AtmoContext *ctx, int enable);

/*
 * Replaces: what `discard` means to the engine (planet_atmosphere_main.gdshaderinc:189-196, 150-152: a fragment whose ray misses the
 * atmosphere shell writes NOTHING -- the render target keeps what it held).  atmo_render has no render target of the engine's to leave
 * alone, so by default (0) it writes (0, 0, 0, 0) into every discarded pixel of the rect: the output buffer then needs no preparation.
 * 1 = the caller states that rgba_dev already holds what discarded pixels shall show (a target cleared once and re-used: a discarded
 * pixel stays discarded while the camera stands still; or the caller clears it per frame anyway): discarded fragments retire without a
 * store, exactly like the shader's -- at pose P_space of the demo 40 % of a 1920x1080 frame's 33 MB store stream.
 * atmo_render_composite never stores discarded fragments, whatever this is set to.
 */
int atmo_set_target_cleared(AtmoContext *ctx, int cleared);

/*
 * Launch order (no reference counterpart): with feedback on, every 8th draw (the first four back to back) records how
 * long each pixel tile's waves ran; a small sort kernel on a side stream turns those costs into a tile order, heaviest
 * tiles first (longest-processing-time-first list scheduling), and later draws pick it up once the host sees the sort
 * complete (hipEventQuery): no draw waits for a sort, 7 of 8 draws carry no bookkeeping.  A draw in row-major order
 * leaves the SIMDs empty for the last ~12 % of its time at 1920x1080; heaviest-first puts the cheap tiles (rays that
 * miss the planet, clear sky) into that drain: direct light 32x8 +9.7 %, clouds_high +5 %, clouds_high_rm +49 %
 * (its heaviest tiles are ~10x the mean), baked-LUT atmosphere +4..5 %; within +-1.5 % on frames whose tiles all weigh
 * the same (profiles/round2/ab_tile_feedback.txt).  The picture does not depend on the order.
 * -1 (default) = on; 0 = off; 1 = on.  Launches inside a HIP graph capture never use the LEARNT order (see the last paragraph).
 * Host-side waits: none since round 5 in a host whose draws and texture updates use different streams or the null stream.  Changing the
 * mode never waits (a state with work in flight is handed to its next owner behind that work, on the device); a FIFTH distinct (rect
 * grid, stream) pair while four are cached recycles the least recently used state (at most 8 times in a row, then such draws simply run
 * in row-major order) behind the marker event of that state's last draw -- device-wide (hipDeviceSynchronize) only when that state drew on
 * the context's non-null home stream (see atmo_set_texture), whose draws carry no marker.  A host that cycles through many rects should
 * still turn the feedback off (0) for that context.  A context keeps one feedback
 * state per (launch grid, draw stream) it sees, up to four (split screen, stereo eyes, uneven row bands), each allocated by
 * the first launch of its key; a fifth key recycles the least recently used state (ordered behind that state's work),
 * and a context that keeps producing new keys runs out of recycling budget (8, one regained every 256 draws) and draws
 * those keys in row-major order while the resident states keep working.
 * A state follows the camera: it compares the matrices of consecutive draws of its key, dilates the measured cost map by the
 * distance the picture moves while an order is in use, records more often (the raymarched-cloud-light kernels sort on the
 * draw stream itself, one frame of lag), and falls back to the row-major launch when the picture moves faster than a cost
 * map stays meaningful -- so two different views alternating on ONE (grid, stream) key (stereo eyes) look like a fast camera
 * and get no reordering: give each eye its own stream or context.
 * Round 6: a draw of the direct-light cloudless kernels that has no learnt order (such a fast camera, the first draws of a key, a draw inside a
 * graph capture) is ordered anyway while the camera is outside the atmosphere shell -- from the camera alone: the tiles whose rays can hit the
 * shell form one run of columns per tile row, passed by value with the kernel arguments (no buffer, no extra launch, nothing to wait for; a
 * captured draw replays with the table it was captured with).  Mode 0 turns that off as well.
 */
int atmo_set_tile_feedback(AtmoContext *ctx, int mode);

/* Last error message of this context (or of the failed atmo_create when ctx == NULL). Never NULL. */
const char *atmo_last_error_string(AtmoContext *ctx);

#ifdef __cplusplus
}
#endif

#endif /* ATMO_H */
