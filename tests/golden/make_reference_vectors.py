#!/usr/bin/env python3
"""Generates tests/golden/reference_exec.npz: outputs of the REFERENCE'S OWN SHADER SOURCE executed here.

The reference (GDShader) ships no tests and Godot does not exist in this image (a GLSL compiler does, found in round 5: Mesa's, see
make_mesa_vectors.py -- the second, independent executor of the same text), so this script runs
the shader text itself: tests/golden/gdshader_vm.py interprets the files under
/root/reference/addons/zylann.atmosphere/shaders/ (read at generation time only -- nothing of them is stored) with IEEE
binary32 arithmetic, the engine's texture units supplied by tests/golden/vm_textures.py under the conventions stated in
DESIGN.md section 2.  The committed .npz holds inputs (scene parameters, matrices, depth buffers; textures are
regenerated from seeds and pinned by CRC) and the expected outputs (RGBA per pixel, the baked optical-depth LUT, the
vertex-stage varyings).  tests/test_reference_exec.py checks the CPU oracle against them (-m "not gpu") and the HIP path
through the C ABI (-m gpu).

    python tests/golden/make_reference_vectors.py            # needs /root/reference; about a minute
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gdshader_vm as VM  # noqa: E402
import vm_textures as T  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
import reference_scenes as RS  # noqa: E402
from reference_scenes import CUBE_N, FULL_SIZE, H, POSES, SHAPE_N, VARIANTS, W, camera_matrices, scenes  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures  # noqa: E402

SHADERS = "/root/reference/addons/zylann.atmosphere/shaders"
F32 = np.float32

def uniforms_for(parser, params, world_to_model, sun=S.DEMO_SUN_POSITION):
    """Host values for the uniforms the shader declares (those the scene does not set keep the shader's defaults)."""
    out = {}
    for name in parser.uniforms:
        if name == "u_world_to_model_matrix":
            out[name] = S.col_major(world_to_model)
        elif name == "u_sun_position":
            out[name] = sun
        elif name in params:
            out[name] = params[name]
    return out


def run_bake(params):
    """optical_depth.gdshader fragment() for every texel of the 256 x 256 target, then what the baker does with it:
    RGBA8 viewport -> bytes reinterpreted as R32F (optical_depth_baker.gd)."""
    n = 256
    p = VM.load(os.path.join(SHADERS, "optical_depth.gdshader"))
    m = VM.Machine(p, n * n, {}, {k: params[k] for k in ("u_planet_radius", "u_atmosphere_height", "u_density")})
    ii, jj = np.meshgrid(np.arange(n), np.arange(n))
    uv = np.stack([(ii.reshape(-1).astype(F32) + F32(0.5)) / F32(n), (jj.reshape(-1).astype(F32) + F32(0.5)) / F32(n)])
    m.globals["UV"] = VM.V("vec2", uv.astype(F32))
    m.globals["COLOR"] = m.zero("vec4")
    m.run("fragment")
    color = m.globals["COLOR"].a  # (4, lanes) floats k / 255
    # the RGBA8 render target stores round(c * 255)
    b = np.rint(color.astype(np.float64) * 255.0).astype(np.uint32)
    bits = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)
    return bits.astype(np.uint32).view(F32).reshape(n, n), m.calls


def run_frame(shader, defines, params, world_to_model, model_matrix, cam, depth, tex_units, time_s=0.0, rows=None,
              sun=S.DEMO_SUN_POSITION, force_defines=None, cube_chain=None, merge_twin_calls=True, conventions=None,
              cube_kwargs=None):
    """vertex() once, fragment() for every pixel of the viewport (or of the given rows).
    force_defines: macros that win over the shader file's own #defines (gdshader_vm.preprocess).
    cube_chain: the coverage cubemap's mip chain [(6, n, n), ...] -> u_cloud_coverage_cubemap is sampled with the implicit LOD
    of a linear-mipmap sampler (vm_textures.CubeTextureLod; derivatives from the 2 x 2 pixel quads of the executed lanes)."""
    p = VM.load(os.path.join(SHADERS, shader + ".gdshader"), defines, force_defines)
    rows = list(range(cam.height)) if rows is None else list(rows)
    n = cam.width * len(rows)
    samplers = dict(tex_units, u_depth_texture=T.DepthTexture(depth))
    if cube_chain is not None:
        samplers["u_cloud_coverage_cubemap"] = T.CubeTextureLod(cube_chain, RS.quad_partners(cam.width, rows, cam.height),
                                                                **(cube_kwargs or {}))
    m = VM.Machine(p, n, samplers, uniforms_for(p, params, world_to_model, sun), source_color=S.srgb_to_linear,
                   merge_twin_calls=merge_twin_calls, conventions=conventions)
    g = m.globals
    ident = m.from_host("mat4", np.eye(4).reshape(-1))
    # vertex stage: the quad's vertices all produce the same varyings, one lane is enough
    g["VERTEX"] = m.zero("vec3")
    g["POSITION"] = m.zero("vec4")
    g["PROJECTION_MATRIX"] = ident
    g["MODELVIEW_MATRIX"] = ident
    g["MODEL_MATRIX"] = m.from_host("mat4", S.col_major(model_matrix))
    g["VIEW_MATRIX"] = m.from_host("mat4", S.col_major(cam.view))
    g["TIME"] = m.from_host("float", [time_s])
    m.run("vertex")
    varyings = (g["v_planet_center_viewspace"].a[:, 0].copy(), g["v_sun_center_viewspace"].a[:, 0].copy())
    # fragment stage, one lane per pixel
    px, py = np.meshgrid(np.arange(cam.width), np.asarray(rows))
    uv = np.stack([(px.reshape(-1).astype(F32) + F32(0.5)) / F32(cam.width),
                   (py.reshape(-1).astype(F32) + F32(0.5)) / F32(cam.height)])
    g["SCREEN_UV"] = VM.V("vec2", uv.astype(F32))
    g["VIEWPORT_SIZE"] = m.from_host("vec2", [cam.width, cam.height])
    g["INV_PROJECTION_MATRIX"] = m.from_host("mat4", S.col_major(cam.inv_projection))
    g["INV_VIEW_MATRIX"] = m.from_host("mat4", S.col_major(cam.inv_view))
    g["ALBEDO"] = m.zero("vec3")
    g["ALPHA"] = m.zero("float")
    m.run("fragment")
    rgb = np.broadcast_to(g["ALBEDO"].a, (3, n))
    a = np.broadcast_to(g["ALPHA"].a, (n,))
    out = np.concatenate([rgb, a[None, :]], axis=0).T.reshape(len(rows), cam.width, 4).astype(F32).copy()
    disc = m.discarded.reshape(len(rows), cam.width)
    out[disc] = 0.0  # a discarded fragment leaves the (cleared) target untouched
    return out, disc, varyings, m.calls


def main():
    t0 = time.time()
    blue = S.make_blue_noise()
    shape = S.make_shape_texture(SHAPE_N)
    cube = S.make_coverage_cubemap(CUBE_N)
    padded = T.pad_cubemap(cube)
    out = {
        "viewport": np.array([W, H]), "shape_n": np.int64(SHAPE_N), "cube_n": np.int64(CUBE_N),
        "crc_blue_noise": np.uint32(S.checksum(blue)), "crc_shape": np.uint32(S.checksum(shape)),
        "crc_cubemap": np.uint32(S.checksum(cube)),
        "poses": np.array(POSES), "variants": np.array(list(VARIANTS)), "scenes": np.array(list(scenes())),
    }
    calls = {}
    for sname, (params, model_matrix) in scenes().items():
        world_to_model = np.linalg.inv(model_matrix)
        lut, c = run_bake(params)
        calls.update(c)
        out[f"lut_{sname}"] = lut
        out[f"model_matrix_{sname}"] = model_matrix
        units = dict(u_optical_depth_texture=T.LutTexture(lut), u_blue_noise_texture=T.ByteTexture2D(blue),
                     u_cloud_shape_texture=T.ShapeTexture(shape), u_cloud_coverage_cubemap=T.CubeTexture(padded))
        for pose in POSES:
            cam = S.Camera.from_pose(W, H, pose)
            out[f"cam_{W}x{H}_{pose}"] = camera_matrices(cam)
            if sname == "alt":  # the planet of the second scene is not at the origin
                depth = S.depth_ground_sphere(cam, center_world=model_matrix[:3, 3], radius=params["u_planet_radius"])
            else:
                depth = S.depth_ground_sphere(cam)
            out[f"depth_{sname}_{pose}"] = depth
            for shader in VARIANTS:
                rgba, disc, vary, c = run_frame(shader, None, params, world_to_model, model_matrix, cam, depth, units)
                calls.update(c)
                key = f"{sname}_{pose}_{shader}"
                out[f"rgba_{key}"] = rgba
                out[f"discard_{key}"] = np.packbits(disc)
                out[f"planet_vs_{sname}_{pose}"], out[f"sun_vs_{sname}_{pose}"] = vary
                print(f"{time.time() - t0:6.1f}s {key}: {int((~disc).sum())} of {disc.size} fragments kept, "
                      f"max rgba {rgba.max():.4f}", flush=True)
    # the DOUBLE_PRECISION compile switch (planet_atmosphere_main.gdshaderinc:25,118-125), one frame
    params, model_matrix = scenes()["demo"]
    cam = S.Camera.from_pose(W, H, "P_limb")
    units = dict(u_optical_depth_texture=T.LutTexture(out["lut_demo"]), u_blue_noise_texture=T.ByteTexture2D(blue),
                 u_cloud_shape_texture=T.ShapeTexture(shape), u_cloud_coverage_cubemap=T.CubeTexture(padded))
    neg = S.Camera.from_pose(W, H, "P_limb")
    neg.inv_view = cam.inv_view.copy()
    neg.inv_view[:3, 3] *= -1.0  # what a double-precision engine build hands the shader
    rgba, disc, _, _ = run_frame("planet_atmosphere_clouds", {"DOUBLE_PRECISION": ""}, params, np.eye(4), model_matrix, neg,
                                 out["depth_demo_P_limb"], units)
    out["rgba_double_precision_P_limb_planet_atmosphere_clouds"] = rgba
    # BASELINE.json sizes (configs[2], configs[3]), whole rows, the bench's textures
    big = demo_textures()
    out["crc_shape_full"], out["crc_cubemap_full"] = np.uint32(S.checksum(big["shape"])), np.uint32(S.checksum(big["cubemap"]))
    padded_big = T.pad_cubemap(big["cubemap"])
    units = dict(u_optical_depth_texture=T.LutTexture(out["lut_demo"]), u_blue_noise_texture=T.ByteTexture2D(blue),
                 u_cloud_shape_texture=T.ShapeTexture(big["shape"]), u_cloud_coverage_cubemap=T.CubeTexture(padded_big))
    for shader, w, h, pose, rows in FULL_SIZE:
        cam = S.Camera.from_pose(w, h, pose)
        out[f"cam_{w}x{h}_{pose}"] = camera_matrices(cam)
        depth = S.depth_ground_sphere(cam)
        rgba, disc, _, _ = run_frame(shader, None, params, np.eye(4), model_matrix, cam, depth, units, rows=rows)
        key = f"full_{w}x{h}_{pose}_{shader}"
        out[f"rgba_{key}"], out[f"rows_{key}"], out[f"depth_{key}"] = rgba, np.asarray(rows), depth[list(rows)]
        print(f"{time.time() - t0:6.1f}s {key}: rows {rows}, {int((~disc).sum())} of {disc.size} fragments kept", flush=True)
    out["called_functions"] = np.array(sorted(calls))
    path = os.path.join(HERE, "reference_exec.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "functions executed:", ", ".join(sorted(calls)))


def with_partner_rows(rows, height):
    """rows plus each row's vertical quad partner, sorted; and the positions of the requested rows in that list."""
    full = sorted(set(rows) | {r ^ 1 for r in rows if (r ^ 1) < height})
    return full, [full.index(r) for r in rows]


def main_round3():
    """tests/golden/reference_exec_r3.npz: the reference text executed (a) with the samplerCube it declares -- linear-mipmap,
    implicit LOD from the 2 x 2 pixel quads -- and (b) at 32 and 64 view steps."""
    t0 = time.time()
    blue = S.make_blue_noise()
    shape = S.make_shape_texture(SHAPE_N)
    cube = S.make_coverage_cubemap(CUBE_N)
    chain = T.mip_chain(cube)
    z = np.load(os.path.join(HERE, "reference_exec.npz"))
    out = {"crc_cubemap": np.uint32(S.checksum(cube)), "cube_levels": np.int64(len(chain)),
           "crc_mips": np.array([S.checksum(lv) for lv in chain], dtype=np.uint32)}
    params, model_matrix = scenes()["demo"]
    lut = z["lut_demo"]
    base_units = dict(u_optical_depth_texture=T.LutTexture(lut), u_blue_noise_texture=T.ByteTexture2D(blue),
                      u_cloud_shape_texture=T.ShapeTexture(shape))
    calls = {}
    # (a) implicit LOD, 48 x 27 (height 27: the last row has no vertical partner)
    for pose in RS.LOD_POSES:
        cam = RS.camera_from_fixture(z, W, H, pose)
        depth = z[f"depth_demo_{pose}"]
        for shader in RS.LOD_VARIANTS:
            rgba, disc, _, c = run_frame(shader, None, params, np.eye(4), model_matrix, cam, depth, base_units, cube_chain=chain)
            calls.update(c)
            out[f"lod_rgba_{pose}_{shader}"] = rgba
            lod0 = z[f"rgba_demo_{pose}_{shader}"]
            print(f"{time.time() - t0:6.1f}s lod {pose} {shader}: {int((~disc).sum())} kept, max |LOD - LOD0| = {np.abs(rgba - lod0).max():.3e}", flush=True)
    # ... at BASELINE sizes with the bench's textures
    big = demo_textures()
    big_chain = T.mip_chain(big["cubemap"])
    out["crc_cubemap_full"] = np.uint32(S.checksum(big["cubemap"]))
    units = dict(base_units, u_cloud_shape_texture=T.ShapeTexture(big["shape"]))
    for shader, w, h, pose, rows in RS.LOD_FULL_SIZE:
        cam = RS.camera_from_fixture(z, w, h, pose)
        depth = S.depth_ground_sphere(cam)
        full, keep = with_partner_rows(rows, h)
        rgba, disc, _, _ = run_frame(shader, None, params, np.eye(4), model_matrix, cam, depth, units, rows=full, cube_chain=big_chain)
        key = f"lodfull_{w}x{h}_{pose}_{shader}"
        out[f"rgba_{key}"], out[f"rows_{key}"], out[f"depth_{key}"] = rgba[keep], np.asarray(rows), depth[list(full)]
        out[f"depthrows_{key}"] = np.asarray(full)
        print(f"{time.time() - t0:6.1f}s {key}: rows {rows} (+ partners), {int((~disc).sum())} of {disc.size} kept", flush=True)
    # (b) 32 and 64 view steps
    for steps in RS.VIEW_STEP_COUNTS:
        force = {"ATMOSPHERE_RAYMARCH_STEPS": steps}
        for pose in POSES:
            cam = RS.camera_from_fixture(z, W, H, pose)
            rgba, disc, _, _ = run_frame("planet_atmosphere_no_clouds", None, params, np.eye(4), model_matrix, cam,
                                         z[f"depth_demo_{pose}"], base_units, force_defines=force)
            out[f"steps{steps}_rgba_{pose}"] = rgba
            print(f"{time.time() - t0:6.1f}s no_clouds view steps {steps} {pose}: max |rgba - 8 steps| = "
                  f"{np.abs(rgba - z[f'rgba_demo_{pose}_planet_atmosphere_no_clouds']).max():.3e}", flush=True)
        shader, w, h, pose, rows = RS.VIEW_STEP_ROWS
        cam = RS.camera_from_fixture(z, w, h, pose)
        depth = S.depth_ground_sphere(cam)
        rgba, disc, _, _ = run_frame(shader, None, params, np.eye(4), model_matrix, cam, depth, base_units, rows=rows, force_defines=force)
        out[f"steps{steps}_rgba_full"], out[f"steps{steps}_depth_full"] = rgba, depth[list(rows)]
        print(f"{time.time() - t0:6.1f}s no_clouds view steps {steps} 1920x1080 rows {rows}", flush=True)
    path = os.path.join(HERE, "reference_exec_r3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_fuzz_lod():
    """tests/golden/reference_exec_fuzz_lod.npz: the random scenes of reference_exec_fuzz.npz that have a cubemap and a cloud variant,
    executed again with the declared linear-mipmap cubemap sampler (implicit LOD): other planet scales, cube sizes 17 .. 128 (17 is
    not a power of two: the kernels' general LOD path), moved and rotated planets, cameras inside the layer."""
    import json

    t0 = time.time()
    fz = np.load(os.path.join(HERE, "reference_exec_fuzz.npz"))
    out = {}
    for k in range(RS.FUZZ_SEEDS):
        tex = RS.fuzz_textures(k)
        shaders = [sh for sh in RS.fuzz_variants(k) if VARIANTS[sh].get("cloud_steps")]
        if tex["cubemap"] is None or not shaders:
            continue
        shader = shaders[0]
        params = {kk: (tuple(v) if isinstance(v, list) else v) for kk, v in json.loads(str(fz[f"params_{k}"])).items()}
        _, cam_args, sun, model, _ = RS.random_scene(k)
        cam = S.Camera(RS.FUZZ_W, RS.FUZZ_H, **cam_args)
        m = fz[f"cam_{k}"]
        cam.inv_projection, cam.inv_view, cam.view = m[0].copy(), m[1].copy(), m[2].copy()
        lut, _ = run_bake({kk: params[kk] for kk in ("u_planet_radius", "u_atmosphere_height", "u_density")})
        units = dict(u_optical_depth_texture=T.LutTexture(lut), u_blue_noise_texture=T.ByteTexture2D(tex["blue_noise"]),
                     u_cloud_shape_texture=T.ShapeTexture(tex["shape"]))
        rgba, disc, _, _ = run_frame(shader, None, params, np.linalg.inv(model), model, cam, fz[f"depth_{k}"], units, sun=sun,
                                     cube_chain=T.mip_chain(tex["cubemap"]))
        out[f"rgba_{k}_{shader}"] = rgba
        d0 = float(np.nanmax(np.abs(rgba - fz[f"rgba_{k}_{shader}"])))
        print(f"{time.time() - t0:6.1f}s seed {k} {shader} (cube {tex['cubemap'].shape[1]}): {int((~disc).sum())} kept, max |LOD - LOD0| = {d0:.3e}", flush=True)
    out["cases"] = np.array(sorted(out))
    path = os.path.join(HERE, "reference_exec_fuzz_lod.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_round4():
    """tests/golden/reference_exec_r4.npz: `planet_atmosphere_no_clouds` executed with 64 view steps (ATMOSPHERE_RAYMARCH_STEPS forced) on four
    THIN atmospheres (H / R = 0.047 .. 0.064) -- seeds 55 and 91 of tests/test_gpu_parity.py::_random_scene are the two of 252 random scenes on
    which the default kernels' running position sum had drifted to 1.07e-4 / 1.08e-4 of alpha in round 3; 13 and 43 are the
    two thinnest 64-step scenes of that fuzz.  Inputs (parameters, camera matrices, sun, depth) are stored beside the outputs."""
    import json

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_parity as TG

    t0 = time.time()
    out = {"seeds": np.array(RS.R4_THIN_SEEDS), "view_steps": np.int64(64)}
    for seed in RS.R4_THIN_SEEDS:
        rng = np.random.default_rng(1000 + seed)
        params, cam, sun = TG._random_scene(rng, seed)
        depth = S.depth_ground_sphere(cam, radius=params["u_planet_radius"]) if seed % 3 else S.depth_far(cam)
        blue = S.make_blue_noise(seed + 1)
        lut, _ = run_bake({k: params[k] for k in ("u_planet_radius", "u_atmosphere_height", "u_density")})
        # the scene dictionaries hold the colours as the inspector does (sRGB); the engine uploads `source_color` uniforms linear
        lin = dict(params, u_atmosphere_modulate=tuple(S.srgb_to_linear(params["u_atmosphere_modulate"]).tolist()),
                   u_atmosphere_ambient_color=tuple(S.srgb_to_linear(params["u_atmosphere_ambient_color"]).tolist()))
        units = dict(u_optical_depth_texture=T.LutTexture(lut), u_blue_noise_texture=T.ByteTexture2D(blue))
        rgba, disc, _, _ = run_frame("planet_atmosphere_no_clouds", None, lin, np.eye(4), np.eye(4), cam, depth, units, sun=sun,
                                     force_defines={"ATMOSPHERE_RAYMARCH_STEPS": 64})
        out[f"params_{seed}"] = np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()}))
        out[f"cam_{seed}"], out[f"sun_{seed}"], out[f"depth_{seed}"] = camera_matrices(cam), np.array(sun), depth
        out[f"viewport_{seed}"] = np.array([cam.width, cam.height])
        out[f"lut_crc_{seed}"], out[f"blue_crc_{seed}"] = np.uint32(S.checksum(lut)), np.uint32(S.checksum(blue))
        out[f"rgba_{seed}"] = rgba
        print(f"{time.time() - t0:6.1f}s seed {seed}: H/R {params['u_atmosphere_height'] / params['u_planet_radius']:.4f}, "
              f"{int((~disc).sum())} of {disc.size} kept, max alpha {rgba[..., 3].max():.4f}", flush=True)
    path = os.path.join(HERE, "reference_exec_r4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_round5():
    """tests/golden/reference_exec_r5.npz: the reference text executed with the samplerCube it declares (linear-mipmap, implicit LOD from the
    2 x 2 pixel quads) on every cloud row of reference_scenes.FULL_SIZE -- whole rows of configs[2] (clouds_high, 1920x1080) and configs[3]
    (clouds_high_rm, 3840x2160) at poses P_space (limb rows included) and P_clouds, with the bench's 256^2 cubemap and 64^3 shape volume:
    what round 3's LOD_FULL_SIZE did for 7 rows, for all 25."""
    t0 = time.time()
    z = np.load(os.path.join(HERE, "reference_exec.npz"))
    params, model_matrix = scenes()["demo"]
    big = demo_textures()
    big_chain = T.mip_chain(big["cubemap"])
    out = {"crc_cubemap_full": np.uint32(S.checksum(big["cubemap"])), "crc_shape_full": np.uint32(S.checksum(big["shape"]))}
    units = dict(u_optical_depth_texture=T.LutTexture(z["lut_demo"]), u_blue_noise_texture=T.ByteTexture2D(S.make_blue_noise()),
                 u_cloud_shape_texture=T.ShapeTexture(big["shape"]))
    for shader, w, h, pose, rows in RS.LOD_FULL_SIZE_R5:
        cam = RS.camera_from_fixture(z, w, h, pose)
        depth = S.depth_ground_sphere(cam)
        full, keep = with_partner_rows(rows, h)
        rgba, disc, _, _ = run_frame(shader, None, params, np.eye(4), model_matrix, cam, depth, units, rows=full, cube_chain=big_chain)
        key = f"lodfull_{w}x{h}_{pose}_{shader}"
        out[f"rgba_{key}"], out[f"rows_{key}"], out[f"depth_{key}"] = rgba[keep], np.asarray(rows), depth[list(full)]
        out[f"depthrows_{key}"] = np.asarray(full)
        lod0 = z[f"rgba_full_{w}x{h}_{pose}_{shader}"] if f"rgba_full_{w}x{h}_{pose}_{shader}" in z.files else None
        extra = "" if lod0 is None else f", max |LOD - LOD0| = {np.abs(rgba[keep] - lod0).max():.3e}"
        print(f"{time.time() - t0:6.1f}s {key}: rows {rows} (+ partners), {int((~disc).sum())} of {disc.size} kept{extra}", flush=True)
    path = os.path.join(HERE, "reference_exec_r5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


class UnsetCube:
    """An unbound samplerCube: the engine's default white texture (README.md:46 "cover uniformly")."""

    def texture(self, d):
        return np.ones(d.shape[1], dtype=F32)


def main_fuzz():
    import json

    t0 = time.time()
    out = {"seeds": np.int64(RS.FUZZ_SEEDS), "viewport": np.array([RS.FUZZ_W, RS.FUZZ_H])}
    for k in range(RS.FUZZ_SEEDS):
        params, cam_args, sun, model, depth_kind = RS.random_scene(k)
        tex = RS.fuzz_textures(k)
        cam = S.Camera(RS.FUZZ_W, RS.FUZZ_H, **cam_args)
        depth = (S.depth_ground_sphere(cam, center_world=model[:3, 3], radius=params["u_planet_radius"]) if depth_kind == "ground"
                 else S.depth_far(cam))
        bake_params = {kk: params[kk] for kk in ("u_planet_radius", "u_atmosphere_height", "u_density")}
        lut, _ = run_bake(bake_params)
        cube = tex["cubemap"]
        cube_unit = UnsetCube() if cube is None else T.CubeTexture(T.pad_cubemap(cube))
        units = dict(u_optical_depth_texture=T.LutTexture(lut), u_blue_noise_texture=T.ByteTexture2D(tex["blue_noise"]),
                     u_cloud_shape_texture=T.ShapeTexture(tex["shape"]), u_cloud_coverage_cubemap=cube_unit)
        out[f"params_{k}"] = np.array(json.dumps({kk: (list(v) if isinstance(v, tuple) else v) for kk, v in params.items()}))
        out[f"cam_{k}"], out[f"sun_{k}"], out[f"model_{k}"], out[f"depth_{k}"] = camera_matrices(cam), np.array(sun), model, depth
        out[f"lut_crc_{k}"] = np.uint32(S.checksum(lut))
        out[f"tex_crc_{k}"] = np.array([S.checksum(tex["blue_noise"]), S.checksum(tex["shape"]),
                                        0 if cube is None else S.checksum(cube)], dtype=np.uint32)
        for shader in RS.fuzz_variants(k):
            rgba, disc, vary, _ = run_frame(shader, None, params, np.linalg.inv(model), model, cam, depth, units, sun=sun)
            out[f"rgba_{k}_{shader}"] = rgba
            out[f"planet_vs_{k}"], out[f"sun_vs_{k}"] = vary
            print(f"{time.time() - t0:6.1f}s seed {k} {shader}: {int((~disc).sum())} of {disc.size} kept, max {np.nanmax(rgba):.3f}", flush=True)
    path = os.path.join(HERE, "reference_exec_fuzz.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--round5-only" in sys.argv:
        main_round5()
    elif "--round4-only" in sys.argv:
        main_round4()
    elif "--fuzz-lod-only" in sys.argv:
        main_fuzz_lod()
    elif "--round3-only" in sys.argv:
        main_round3()
        main_fuzz_lod()
    else:
        if "--fuzz-only" not in sys.argv:
            main()
            main_round3()
        main_fuzz()
        main_fuzz_lod()
        main_round4()
        main_round5()
