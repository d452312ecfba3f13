/*
 * atmo_debug.h -- experiment knobs and diagnostics of libatmo_hip.so.  NOT part of the surface a Godot host binds
 * (that is include/atmo.h, whose entry points each replace a reference interface): nothing here has a counterpart in
 * the reference.  Used by this repository's tests, bench.py and tools/; may change between ABI versions.
 */
#ifndef ATMO_DEBUG_H
#define ATMO_DEBUG_H

#include "atmo.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ATMO_EXPERIMENTAL 1

/*
 * Launch shape (no reference counterpart): lanes per view ray.  1 = one wavefront lane per ray (64 rays per wave);
 * 2 = two adjacent lanes share a ray (32 rays per wave: each lane takes every second cloud sample / half of the view
 * samples, results cross by DPP) -- twice the waves for the same frame; the cloud march is the same bits, the
 * atmosphere sums agree within rounding.  It pays only for small launches dominated by a few very long waves
 * (clouds_high_rm at 1920x1080: -11 %), so 0 (default, "auto") currently means 1.
 */
int atmo_set_lane_split(AtmoContext *ctx, int lanes_per_ray);

/* Diagnostics of the tile-order feedback (atmo_set_tile_feedback): feedback states in use (one per launch grid and draw
 * stream, at most 4), draws so far that were dispatched in a sorted order, sorts enqueued, states recycled for another key. */
int atmo_get_feedback_stats(AtmoContext *ctx, int *states, unsigned *ordered_draws, unsigned *sorts, unsigned *recycled);

/* The host-side motion estimate behind the tile-order feedback (no device, no context): how many pixels the picture's cost
 * features move between two frames -- the planet's silhouette (centre and four limb points along the camera's axes), plus, with
 * surface_points != 0, six points fixed on the planet (the cloud pattern).  radius = u_planet_radius + u_atmosphere_height. */
float atmo_debug_motion_px(const AtmoFrame *a, const AtmoFrame *b, float radius, int surface_points);

/* Device time of `atmo_render` kernels measured with HIP events recorded around the launch on its own stream:
 * atmo_set_timing(ctx, k): k = 0 off, k >= 1 brackets every k-th launch (k > 1 keeps the ~5 us cost of recording two
 * events out of most steps); atmo_get_timing returns the number of bracketed launches and their total milliseconds
 * since enabling (it waits for them). */
int atmo_set_timing(AtmoContext *ctx, int every_kth);
int atmo_get_timing(AtmoContext *ctx, int *launches, double *total_ms);

/*
 * Host-only helpers (no device, no context): the device layouts atmo_set_texture builds, exposed so they can be
 * checked without a GPU.  cubemap: 6*(n+1)^2 words, word (i,j) of a face = the 2x2 texels of the seamless-apron
 * padded face starting at padded (i,j), bytes 0..3 = (i,j),(i+1,j),(i,j+1),(i+1,j+1).  shape: n^3 words, word
 * (i,j,k) = T(i,j,k),T(i+1,j,k),T(i,j+1,k),T(i+1,j+1,k) with repeat wrap.  lut: (h+2) x (w+2) floats, clamp apron.
 */
int atmo_host_layout_cubemap(const uint8_t *faces, int n, uint32_t *footprints_out);
int atmo_host_layout_shape(const uint8_t *texels, int n, uint32_t *footprints_out);
int atmo_host_layout_lut(const float *lut, int w, int h, float *apron_out);
/* next mip level (n/2 per side, 6 faces) of a 6 x n^2 level: the 2x2 box (a + b + c + d + 2) >> 2 */
int atmo_host_cubemap_mip(const uint8_t *level, int n, uint8_t *next_out);

/* Diagnostics (no reference counterpart): copies the DEVICE layout of a bound texture (what the re-layout kernels of
 * atmo_set_texture wrote: LUT apron / shape footprints / cubemap footprints of all bound levels / blue-noise bytes) to
 * host memory, so it can be compared with atmo_host_layout_*.  out_host == NULL only reports the size in *bytes_out. */
int atmo_read_texture_layout(AtmoContext *ctx, const char *name, void *out_host, size_t capacity_bytes, size_t *bytes_out, void *stream);

/* Diagnostics (no reference counterpart): on the device, compares the kernels' cheap correctly-rounded sqrt and
 * divide-by-uniform helpers with the compiler's IEEE expansions over `count` consecutive float bit patterns
 * starting at `first_bits`, and reports the number of mismatches (must be 0).  Since round 6 the divide counter also holds exact_rcp(x) != 1 / x
 * (the reciprocal inside the declared sampler's lambda) for 2^-100 <= |x| <= 2^100. */
int atmo_selftest_exact_math(AtmoContext *ctx, uint32_t first_bits, uint32_t count, float divisor,
                             uint32_t *sqrt_mismatches, uint32_t *div_mismatches);

/* Diagnostics (no reference counterpart): runs the kernels' direct light march (ATMO_LIGHT_DIRECT; the device function the render
 * kernel inlines) for n sample positions -- xyz relative to the planet centre -- and unit sun directions, host arrays in,
 * host array out, with the context's u_planet_radius / u_atmosphere_height / u_density.  With light_steps = 64 at the LUT's
 * texel-centre geometry (shaders/optical_depth.gdshader:45-65) this is the integral the reference bakes into that texel. */
int atmo_debug_marched_optical_depth(AtmoContext *ctx, int n, const float *pos_xyz, const float *dir_xyz, int light_steps, float *out);

/* Diagnostics (no reference counterpart): the kernels' log2_cr -- log2 of a float evaluated in double and rounded once, the logarithm of the declared
 * cubemap sampler's lambda since round 6 -- for n host floats (finite, normal, > 0), host array out.  The oracle carries the same operation sequence
 * on the same table (tools/make_log2_table.py); tests/test_gpu_parity.py holds the two to the same bits. */
int atmo_debug_log2_cr(AtmoContext *ctx, int n, const float *x, float *out);

/* What this library was built from: 16 hex digits of sha256 over the two .hip sources under csrc/, their headers and the compiler flags (godot_atmosphere_shader_amd/build.py,
 * source_id), "unstamped" for a build made any other way.  tools/profile.sh stamps the rocprofv3 counter files with it, bench.py compares
 * (`traffic_stale`): counters of other kernels than the ones being timed are not presented as theirs. */
const char *atmo_build_id(void);

/* Name of the kernel the most recent atmo_render of this context launched, "atmo_render_kernel<FLAGS, LSTEPS, SPLIT>"
 * (before the first launch: the one-lane-per-ray form), for matching rocprofv3 kernel traces. */
const char *atmo_kernel_name(AtmoContext *ctx);

/* Heavy tiles on two lanes per ray (round 5; the default cloud kernels under the declared sampler, tile-order feedback on): draws so far that
 * drew their heaviest tiles with the lane-split kernel beside the rest, and how many tiles the last such draw split.  ATMO_HEAVY_SPLIT=0
 * (read in atmo_create) turns it off; the picture does not depend on it, bit for bit. */
int atmo_get_split_stats(AtmoContext *ctx, unsigned *split_draws, unsigned *heavy_tiles_last);

/* How often calls of this context fell back to a device-wide host wait (hipDeviceSynchronize) where a stream-side wait on the device was
 * not possible: a remembered stream that the caller has destroyed since, or more draw streams than the context tracks (8).  0 in a host
 * that keeps its streams alive (tests/test_gpu_parity.py::test_texture_update_does_not_wait_for_unrelated_streams). */
int atmo_get_host_wait_stats(AtmoContext *ctx, unsigned *device_syncs);

/*
 * The host side WITHOUT a device (round 5; for the CPU test suite and the sanitizer build of libatmo_hip.so, `make sanitize-host`): a
 * context that owns nothing on a GPU -- the uniform table (atmo_set_param_f32 / atmo_get_param_f32), atmo_set_precision,
 * atmo_set_sampler_lod, atmo_set_host_double_precision, atmo_set_target_cleared work on it; an entry point that needs the device
 * fails -- with ATMO_E_ARG / ATMO_E_STATE where its argument and state checks come first, with ATMO_E_HIP (hipSetDevice(-1)) where the device is
 * the first thing it touches -- and never succeeds; atmo_set_timing / atmo_get_timing only keep their bookkeeping (no events exist without draws);
 * atmo_destroy frees it.  Arguments as atmo_create (minus the device index).
 * atmo_debug_frame_constants evaluates the per-frame (pixel-independent) expressions of the shader exactly as a draw would
 * (fill_consts: main:136,164; v2:47-51; clouds:104-115,186-206,260-261,285-294; the level-0 certificate's constant for faces of
 * cube_n texels) and returns them as floats, in the order documented at its definition in csrc/atmo_api.hip (81 values).
 */
int atmo_debug_create_host_only(int variant, int view_steps, int cloud_steps, int light_mode, int light_steps, AtmoContext **out);
int atmo_debug_frame_constants(AtmoContext *ctx, const AtmoFrame *frame, int cube_n, float *out, int capacity, int *count);

#ifdef ATMO_WAVE_TRACE
/* Diagnostic builds only (-DATMO_WAVE_TRACE: tools/wave_timeline.py, tools/rmq_stats.py; the shipped library does not export it): copies the wave
 * trace of the last draw -- 4 x uint64 per wave: start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID | preamble ticks << 8 -- to the host and
 * returns the wave count (-1 on a HIP error).  Waits for the device. */
long long atmo_debug_wave_trace(AtmoContext *ctx, unsigned long long *host, long long max_waves);
#endif

#ifdef __cplusplus
}
#endif

#endif /* ATMO_DEBUG_H */
