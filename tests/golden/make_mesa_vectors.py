#!/usr/bin/env python3
"""Generates tests/golden/reference_exec_mesa.npz: the reference's shader text compiled by MESA'S GLSL compiler and executed by llvmpipe
(tests/golden/mesa_exec.py, mesa_glsl_runner.c) -- a pin of the oracle that the author of this repository did not write -- and prints the
report kept as profiles/round5/mesa_pin.txt.  Build container only (needs /root/reference and the image's Mesa); about five minutes.

    python tests/golden/make_mesa_vectors.py | tee profiles/round5/mesa_pin.txt

What the file holds (inputs are those of reference_exec.npz: same scenes, cameras, depth buffers, textures from the same seeds):
  * every fragment of all 7 shader variants x 2 scenes x 5 poses at 48 x 27, the discard masks, the vertex-stage varyings;
  * the 256 x 256 optical-depth bake of both scenes through the RGBA8 packing;
  * planet_atmosphere_no_clouds at 32 and 64 view steps (macro forced over the file's #define);
  * whole rows of BASELINE.json's configs[1..3] at 1920x1080 / 3840x2160, with the level-0 cubemap sampler and with the sampler the reference
    declares (linear-mipmap, llvmpipe's OWN level-of-detail selection);
  * llvmpipe's measured accuracy of exp / exp2 / log2 / pow / sqrt / inversesqrt / sin / cos, which is what bounds the agreement on the
    cloud variants (exp: 1.1e-6 relative; a cloud pixel's light is a product of up to 64 of them times an optical thickness).
tests/test_reference_mesa.py holds the CPU oracle (-m "not gpu") and the HIP path (-m gpu) to these vectors."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import mesa_exec as M  # noqa: E402
import reference_scenes as RS  # noqa: E402
import vm_textures as T  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.demo import demo_textures  # noqa: E402
from godot_atmosphere_shader_amd.planet_atmosphere import make_frame  # noqa: E402

F32 = np.float32
# whole rows at the BASELINE sizes: (shader, w, h, pose, rows) -- a subset of reference_scenes.FULL_SIZE / LOD_FULL_SIZE_R5 (fixture size)
MESA_ROWS = [
    ("planet_atmosphere_no_clouds", 1920, 1080, "P_space", (330, 539)),
    ("planet_atmosphere_clouds_high", 1920, 1080, "P_space", (200, 540, 900)),
    ("planet_atmosphere_clouds_high", 1920, 1080, "P_clouds", (300, 800)),
    ("planet_atmosphere_clouds_high_rm", 3840, 2160, "P_space", (800, 1400)),
    ("planet_atmosphere_clouds_high_rm", 3840, 2160, "P_clouds", (600, 1600)),
]


def relerr(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


BLOCK = 16


def block_means(rgba, discarded):
    """(H / 16, W / 16, 4) float32 means (accumulated in float64; the frame sizes used are multiples of 16 in width, the last block row of 1080 is 8 high)
    and (H / 16, W / 16) uint16 counts of kept fragments."""
    h, w = discarded.shape
    by, bx = (h + BLOCK - 1) // BLOCK, (w + BLOCK - 1) // BLOCK
    pad = np.zeros((by * BLOCK, bx * BLOCK, 4), dtype=np.float64)
    pad[:h, :w] = rgba
    cnt = np.zeros((by * BLOCK, bx * BLOCK), dtype=np.float64)
    cnt[:h, :w] = 1.0
    kept = np.zeros((by * BLOCK, bx * BLOCK), dtype=np.int64)
    kept[:h, :w] = ~discarded
    n = cnt.reshape(by, BLOCK, bx, BLOCK).sum((1, 3))
    mean = pad.reshape(by, BLOCK, bx, BLOCK, 4).sum((1, 3)) / n[..., None]
    return mean.astype(np.float32), kept.reshape(by, BLOCK, bx, BLOCK).sum((1, 3)).astype(np.uint16)


def stats(a, b):
    e = relerr(a, b)
    return f"max {e.max():.2e}  p99 {np.percentile(e, 99):.1e}  p99.9 {np.percentile(e, 99.9):.1e}  beyond 1e-4: {100.0 * np.mean(e > 1e-4):.3f} %"


def function_accuracy():
    src = """#version 450 core
uniform vec2 VIEWPORT_SIZE; out vec4 o; uniform int mode;
void main() { float t = (gl_FragCoord.x - 0.5 + (gl_FragCoord.y - 0.5) * VIEWPORT_SIZE.x) / (VIEWPORT_SIZE.x * VIEWPORT_SIZE.y);
 float x = mix(-20.0, 20.0, t); float p = mix(1e-3, 50.0, t);
 if (mode == 0) o = vec4(x, exp(x), exp2(x), 0.0);
 else if (mode == 1) o = vec4(p, log2(p), sqrt(p), inversesqrt(p));
 else if (mode == 2) o = vec4(p, pow(p, 1.7), 1.0 / p, 3.0 / (p + 30.0));
 else o = vec4(x, sin(x), cos(x), 0.0); }"""
    p = M.Program(src)
    p.set("VIEWPORT_SIZE", "vec2", [256, 256])

    def rel(a, b):
        return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1e-30)))

    acc = {}
    for mode in range(4):
        p.set("mode", "int", [mode])
        o = p.draw(256, 256).reshape(-1, 4)
        x = o[:, 0].astype(np.float64)
        if mode == 0:
            acc["exp (relative)"], acc["exp2 (relative)"] = rel(o[:, 1], np.exp(x)), rel(o[:, 2], np.exp2(x))
        elif mode == 1:
            acc["log2 (absolute)"] = float(np.max(np.abs(o[:, 1] - np.log2(x))))
            acc["sqrt (relative)"], acc["inversesqrt (relative)"] = rel(o[:, 2], np.sqrt(x)), rel(o[:, 3], 1.0 / np.sqrt(x))
        elif mode == 2:
            acc["pow(x, 1.7) (relative)"], acc["1 / x (relative)"], acc["a / b (relative)"] = rel(o[:, 1], x ** 1.7), rel(o[:, 2], 1.0 / x), rel(o[:, 3], 3.0 / (x + 30.0))
        else:
            acc["sin (absolute)"], acc["cos (absolute)"] = float(np.max(np.abs(o[:, 1] - np.sin(x)))), float(np.max(np.abs(o[:, 2] - np.cos(x))))
    p.close()
    return acc


def main():
    from oracle.oracle import Oracle

    oracle = Oracle("f32")
    t0 = time.time()
    z = np.load(os.path.join(HERE, "reference_exec.npz"))
    r3 = np.load(os.path.join(HERE, "reference_exec_r3.npz"))
    r5 = np.load(os.path.join(HERE, "reference_exec_r5.npz"))
    W, H = RS.W, RS.H
    blue, shape, cube = S.make_blue_noise(), S.make_shape_texture(RS.SHAPE_N), S.make_coverage_cubemap(RS.CUBE_N)
    assert S.checksum(blue) == int(z["crc_blue_noise"]) and S.checksum(shape) == int(z["crc_shape"]) and S.checksum(cube) == int(z["crc_cubemap"])
    out = {"mesa_info": np.array(M.info()), "gallivm_perf": np.array(os.environ["GALLIVM_PERF"]),
           "crc_blue_noise": np.uint32(S.checksum(blue)), "crc_shape": np.uint32(S.checksum(shape)), "crc_cubemap": np.uint32(S.checksum(cube))}
    print(f"# The reference's shader text on a GLSL implementation this repository's author did not write.\n# {M.info()}\n"
          f"# GALLIVM_PERF={os.environ['GALLIVM_PERF']} (per-pixel level of detail, float filter weights); generated by tests/golden/make_mesa_vectors.py\n"
          "# errors: |a - b| / max(1, |b|) per channel (absolute up to |value| = 1, relative above: cloud light is unclamped HDR)\n")
    acc = function_accuracy()
    out["accuracy_names"], out["accuracy_values"] = np.array(list(acc)), np.array(list(acc.values()))
    print("## 1. llvmpipe's built-in functions against float64 (256 x 256 arguments each; the interpreter and the oracle are correctly rounded to <= 1 ulp = 6e-8)")
    for k, v in acc.items():
        print(f"   {k:28s} {v:.2e}")
    print("\n## 2. all 7 variants x 2 scenes x 5 poses, 48 x 27, level-0 cubemap sampler: Mesa against the interpreter's vectors (reference_exec.npz)\n"
          "##    and against the CPU oracle; discard masks and vertex-stage varyings compared exactly")
    worst = {}
    for sname, (params, model) in RS.scenes().items():
        w2m = np.linalg.inv(model)
        tex = dict(lut=z[f"lut_{sname}"], blue=blue, shape=shape, cubemap=cube)
        otex = dict(blue_noise=blue, shape=shape, cubemap=cube, optical_depth=z[f"lut_{sname}"])
        oparams = dict(params, u_world_to_model_matrix=S.col_major(w2m))
        for pose in RS.POSES:
            cam = RS.camera_from_fixture(z, W, H, pose)
            depth = z[f"depth_{sname}_{pose}"]
            for shader in RS.VARIANTS:
                rgba, disc, vary = M.run_frame(shader, None, params, w2m, model, cam, depth, tex)
                key = f"{sname}_{pose}_{shader}"
                want = z[f"rgba_{key}"]
                wd = np.unpackbits(z[f"discard_{key}"])[:W * H].reshape(H, W).astype(bool)
                assert np.array_equal(disc, wd), key
                assert np.array_equal(vary[0], z[f"planet_vs_{sname}_{pose}"]) and np.array_equal(vary[1], z[f"sun_vs_{sname}_{pose}"]), key
                orc, _ = oracle.render(oparams, otex, RS.VARIANTS[shader], make_frame(cam, model, S.DEMO_SUN_POSITION, 0.0), depth, nthreads=8)
                out[f"rgba_{key}"], out[f"discard_{key}"] = rgba, np.packbits(disc)
                e_vm, e_or = relerr(rgba, want), relerr(rgba, orc)
                fam = shader.replace("planet_atmosphere_", "")
                w = worst.setdefault(fam, [0.0, 0.0, 0.0, 0])
                w[0], w[1], w[2], w[3] = max(w[0], e_vm.max()), max(w[1], e_or.max()), max(w[2], float(np.mean(e_or > 1e-4))), w[3] + 1
            out[f"planet_vs_{sname}_{pose}"], out[f"sun_vs_{sname}_{pose}"] = vary
    print(f"   {'variant':18s} frames   max vs interpreter   max vs oracle   largest share of a frame's values beyond 1e-4 (vs oracle)")
    for fam, (a, b, c, n) in worst.items():
        print(f"   {fam:18s} {n:4d}     {a:10.2e}        {b:10.2e}       {100.0 * c:6.3f} %")
    print("   70 of 70 discard masks identical; 10 of 10 pairs of varyings bit-identical")
    print("\n## 3. optical_depth.gdshader, 256 x 256 texels through the RGBA8 packing: Mesa against the interpreter (= the oracle's bake, bit for bit)")
    for sname, (params, _) in RS.scenes().items():
        lut, _ = M.run_bake(params)
        want = z[f"lut_{sname}"]
        same = int((lut.view(np.uint32) == want.view(np.uint32)).sum())
        out[f"lut_{sname}"] = lut
        print(f"   scene {sname}: {same} of {lut.size} texels bit-identical; max relative difference {float(np.max(np.abs(lut - want) / np.maximum(np.abs(want), 1e-30))):.2e}")
    print("\n## 4. planet_atmosphere_no_clouds at 32 and 64 view steps (ATMOSPHERE_RAYMARCH_STEPS forced), 5 poses: Mesa against the interpreter (reference_exec_r3.npz)")
    params, model = RS.scenes()["demo"]
    tex = dict(lut=z["lut_demo"], blue=blue, shape=shape, cubemap=cube)
    for steps in RS.VIEW_STEP_COUNTS:
        worst_s = 0.0
        for pose in RS.POSES:
            cam = RS.camera_from_fixture(z, W, H, pose)
            rgba, _, _ = M.run_frame("planet_atmosphere_no_clouds", None, params, np.eye(4), model, cam, z[f"depth_demo_{pose}"], tex,
                                     force_defines={"ATMOSPHERE_RAYMARCH_STEPS": steps})
            out[f"steps{steps}_rgba_{pose}"] = rgba
            worst_s = max(worst_s, float(np.abs(rgba - r3[f"steps{steps}_rgba_{pose}"]).max()))
        print(f"   {steps} view steps: max {worst_s:.2e}")
    # the DOUBLE_PRECISION compile switch (planet_atmosphere_main.gdshaderinc:25,118-125): the text assigns to a component of its `mat4` PARAMETER
    cam = RS.camera_from_fixture(z, W, H, "P_limb")
    neg = RS.camera_from_fixture(z, W, H, "P_limb")
    neg.inv_view = cam.inv_view.copy()
    neg.inv_view[:3, 3] *= -1.0  # what a double-precision engine build hands the shader
    rgba, _, _ = M.run_frame("planet_atmosphere_clouds", {"DOUBLE_PRECISION": ""}, params, np.eye(4), model, neg, z["depth_demo_P_limb"], tex)
    out["rgba_double_precision_P_limb_planet_atmosphere_clouds"] = rgba
    print(f"   #define DOUBLE_PRECISION, clouds, P_limb: {stats(rgba, z['rgba_double_precision_P_limb_planet_atmosphere_clouds'])}")
    print("\n## 5. the sampler the reference declares (linear-mipmap, implicit level of detail), 48 x 27 -- frames whose cubemap is MINIFIED 4-8x.  Here the level\n"
          "##    of detail is llvmpipe's own; where lambda > 0 the two implementations blend different mip levels (why: section 11 -- the fetch sits in non-uniform\n"
          "##    control flow, where GLSL leaves implicit derivatives undefined, and each implementation resolves that its own way).  Recorded, not a test:")
    chain = T.mip_chain(cube)
    for pose in RS.LOD_POSES:
        cam = RS.camera_from_fixture(z, W, H, pose)
        for shader in RS.LOD_VARIANTS:
            rgba, _, _ = M.run_frame(shader, None, params, np.eye(4), model, cam, z[f"depth_demo_{pose}"], tex, cube_chain=chain)
            want = r3[f"lod_rgba_{pose}_{shader}"]
            print(f"   {pose:8s} {shader.replace('planet_atmosphere_', ''):16s} {stats(rgba, want)}   (declared against level 0 in the interpreter: {np.abs(want - z[f'rgba_demo_{pose}_{shader}']).max():.2f})")
    print("\n## 6. BASELINE.json's configs[1..3] at their sizes, FULL FRAMES on llvmpipe against the CPU oracle (every pixel), and the committed rows against the\n"
          "##    interpreter's rows; both cubemap samplers (at these sizes 97-100 % of the coverage samples are magnified: lambda = 0 under either rule)")
    big = demo_textures()
    big_chain = T.mip_chain(big["cubemap"])
    btex = dict(lut=z["lut_demo"], blue=blue, shape=big["shape"], cubemap=big["cubemap"])
    out["crc_shape_full"], out["crc_cubemap_full"] = np.uint32(S.checksum(big["shape"])), np.uint32(S.checksum(big["cubemap"]))
    oparams = dict(params, u_world_to_model_matrix=S.col_major(np.eye(4)))
    for shader, w, h, pose, rows in MESA_ROWS:
        cam = RS.camera_from_fixture(z, w, h, pose)
        depth = S.depth_ground_sphere(cam)
        frame = make_frame(cam, model, S.DEMO_SUN_POSITION, 0.0)
        for sampler in (("lod0", "declared") if "clouds" in shader.replace("no_clouds", "") else ("lod0",)):
            declared = sampler == "declared"
            rgba, disc, _ = M.run_frame(shader, None, params, np.eye(4), model, cam, depth, btex, cube_chain=big_chain if declared else None)
            otex = dict(blue_noise=blue, shape=big["shape"], optical_depth=z["lut_demo"],
                        cubemap=oracle.cubemap_mip_chain(big["cubemap"]) if declared else big["cubemap"])
            cfg = dict(RS.VARIANTS[shader], cube_lod=1) if declared else RS.VARIANTS[shader]
            orc, hits = oracle.render(oparams, otex, cfg, frame, depth, nthreads=8)
            assert np.array_equal(disc, np.all(orc == 0.0, axis=-1)) or int(disc.sum()) == orc.shape[0] * orc.shape[1] - hits
            key = f"rows_{sampler}_{w}x{h}_{pose}_{shader}"
            out[f"rgba_{key}"], out[f"which_{key}"], out[f"depth_{key}"] = rgba[list(rows)], np.asarray(rows), depth[list(rows)]
            if pose == "P_space" and (declared or not ("clouds" in shader.replace("no_clouds", ""))):
                # the WHOLE frame, compactly: per 16 x 16 block the mean of every channel and the number of kept fragments -- every pixel of a frame drawn
                # elsewhere enters a comparison against Mesa's (tests/test_reference_mesa.py::test_*_whole_frame_blocks)
                bm, bk = block_means(rgba, disc)
                out[f"blockmean_{sampler}_{w}x{h}_{pose}_{shader}"], out[f"blockkept_{sampler}_{w}x{h}_{pose}_{shader}"] = bm, bk
                om, ok = block_means(orc, np.all(orc == 0.0, axis=-1))
                line_b = f"   {'':14s} {'':9s} {'':8s} {'':8s} 16 x 16 block means vs oracle: max {np.abs(bm - om).max():.2e}; kept-fragment counts identical in {int((bk == ok).sum())} of {bk.size} blocks"
            else:
                line_b = None
            line = f"   {shader.replace('planet_atmosphere_', ''):14s} {w}x{h} {pose:8s} {sampler:8s} full frame vs oracle: {stats(rgba, orc)}"
            src, rk = (r5, f"lodfull_{w}x{h}_{pose}_{shader}") if declared else (z, f"full_{w}x{h}_{pose}_{shader}")
            if f"rgba_{rk}" in src.files:
                have = [int(r) for r in src[f"rows_{rk}"]]
                common = [r for r in rows if r in have]
                if common:
                    line += f" | rows {common} vs interpreter: max {relerr(rgba[common], src[f'rgba_{rk}'][[have.index(r) for r in common]]).max():.2e}"
            print(line + f"   [{time.time() - t0:.0f} s]", flush=True)
            if line_b:
                print(line_b, flush=True)
    print("##    the other poses, clouds_high_rm 1920x1080, declared sampler, full frames against the oracle (report only, no vectors):")
    for pose in ("P_limb", "P_ground", "P_night"):
        cam = S.Camera.from_pose(1920, 1080, pose)
        depth = S.depth_ground_sphere(cam)
        shader = "planet_atmosphere_clouds_high_rm"
        rgba, disc, _ = M.run_frame(shader, None, params, np.eye(4), model, cam, depth, btex, cube_chain=big_chain)
        otex = dict(blue_noise=blue, shape=big["shape"], optical_depth=z["lut_demo"], cubemap=oracle.cubemap_mip_chain(big["cubemap"]))
        orc, hits = oracle.render(oparams, otex, dict(RS.VARIANTS[shader], cube_lod=1), make_frame(cam, model, S.DEMO_SUN_POSITION, 0.0), depth, nthreads=8)
        assert int((~disc).sum()) == hits, pose
        print(f"   clouds_high_rm 1920x1080 {pose:8s} declared full frame vs oracle: {stats(rgba, orc)}; {hits} fragments kept by both   [{time.time() - t0:.0f} s]", flush=True)
    headline(out, oracle, z)
    composite(out, z)
    path = os.path.join(HERE, "reference_exec_mesa.npz")
    np.savez_compressed(path, **out)
    print(f"\n# wrote {os.path.relpath(path, ROOT)}: {os.path.getsize(path)} bytes, {len(out)} arrays")


def headline(out, oracle, z):
    """Section 13: BASELINE.json's headline configuration -- planet_atmosphere_no_clouds at 32 view steps with the DIRECT 8-step light march.  It has no shader
    file of its own in the reference; mesa_exec.DIRECT_LIGHT_GLUE is the composition (the reference's ray_sphere and get_atmosphere_density in the loop of
    optical_depth.gdshader:17-31, spliced in front of the reference's compute_atmosphere_v2, whose one LUT fetch a macro redirects)."""
    params, model = RS.scenes()["demo"]
    blue = S.make_blue_noise()
    tex = dict(lut=z["lut_demo"], blue=blue)
    oparams = dict(params, u_world_to_model_matrix=S.col_major(np.eye(4)))
    ocfg = dict(view_steps=32, light_steps=8)
    force = {"ATMOSPHERE_RAYMARCH_STEPS": 32}
    print("\n## 13. BASELINE's headline configuration (32 view x 8 light steps, direct light march: the kernel bench.py's `value` is measured on), composed from the\n"
          "##     reference's own functions (mesa_exec.DIRECT_LIGHT_GLUE: 14 lines of glue) and compiled by Mesa, against the CPU oracle's direct mode")
    for pose in RS.POSES:
        cam = RS.camera_from_fixture(z, RS.W, RS.H, pose)
        depth = z[f"depth_demo_{pose}"]
        rgba, disc, _ = M.run_frame("planet_atmosphere_no_clouds", None, params, np.eye(4), model, cam, depth, tex, force_defines=force, direct_light_steps=8)
        orc, hits = oracle.render(oparams, dict(blue_noise=blue), ocfg, make_frame(cam, model, S.DEMO_SUN_POSITION, 0.0), depth, nthreads=8)
        assert int((~disc).sum()) == hits
        out[f"direct32x8_rgba_{pose}"] = rgba
        print(f"   48x27 {pose:8s}: {hits} fragments kept by both; max |Mesa - oracle| {np.abs(rgba - orc).max():.2e}")
    w, h, pose, rows = 1920, 1080, "P_space", (330, 539)
    cam = RS.camera_from_fixture(z, w, h, pose)
    depth = S.depth_ground_sphere(cam)
    rgba, disc, _ = M.run_frame("planet_atmosphere_no_clouds", None, params, np.eye(4), model, cam, depth, tex, force_defines=force, direct_light_steps=8)
    orc, hits = oracle.render(oparams, dict(blue_noise=blue), ocfg, make_frame(cam, model, S.DEMO_SUN_POSITION, 0.0), depth, nthreads=8)
    assert int((~disc).sum()) == hits
    bm, bk = block_means(rgba, disc)
    om, ok = block_means(orc, np.all(orc == 0.0, axis=-1))
    out["direct32x8_rows_rgba"], out["direct32x8_rows_which"] = rgba[list(rows)], np.asarray(rows)
    out["direct32x8_blockmean"], out["direct32x8_blockkept"] = bm, bk
    print(f"   1920x1080 P_space (BASELINE configs[1] as benchmarked), full frame vs oracle: {stats(rgba, orc)}; {hits} fragments kept by both;\n"
          f"      16 x 16 block means: max {np.abs(bm - om).max():.2e}, kept-fragment counts identical in {int((bk == ok).sum())} of {bk.size} blocks")


COMPOSITE = [("P_limb", "planet_atmosphere_clouds_high_rm"), ("P_space", "planet_atmosphere_no_clouds")]


def composite_scene():
    """the scene colour buffer the atmosphere is blended onto: seeded noise, regenerated by the tests"""
    return np.random.default_rng(7).uniform(0.0, 1.0, (RS.H, RS.W, 4)).astype(F32)


def composite(out, z):
    """Section 14: the draw WITH the renderer's blend stage -- the shader's (ALBEDO, ALPHA) through llvmpipe's fixed-function blender
    set as a blend_mix material's is (colour SRC_ALPHA / ONE_MINUS_SRC_ALPHA, alpha ONE / ONE_MINUS_SRC_ALPHA), onto a colour buffer holding a scene."""
    params, model = RS.scenes()["demo"]
    tex = dict(lut=z["lut_demo"], blue=S.make_blue_noise(), shape=S.make_shape_texture(RS.SHAPE_N), cubemap=S.make_coverage_cubemap(RS.CUBE_N))
    scene = composite_scene()
    print("\n## 14. the draw with the blend stage: llvmpipe's fixed-function blender (blend_mix state) over a colour buffer of seeded noise, 48 x 27")
    for pose, shader in COMPOSITE:
        cam = RS.camera_from_fixture(z, RS.W, RS.H, pose)
        blended, _, _ = M.run_frame(shader, None, params, np.eye(4), model, cam, z[f"depth_demo_{pose}"], tex, over=scene)
        src = out[f"rgba_demo_{pose}_{shader}"]
        disc = np.unpackbits(out[f"discard_demo_{pose}_{shader}"])[:RS.W * RS.H].reshape(RS.H, RS.W).astype(bool)
        a = src[..., 3:4]
        want = np.concatenate([src[..., :3] * a + scene[..., :3] * (F32(1.0) - a), a + scene[..., 3:] * (F32(1.0) - a)], axis=-1).astype(F32)
        want[disc] = scene[disc]
        out[f"composite_{pose}_{shader}"] = blended
        print(f"   {pose:8s} {shader.replace('planet_atmosphere_', ''):16s}: discarded fragments leave the buffer untouched: {np.array_equal(blended[disc], scene[disc])};"
              f" blender == src * a + dst * (1 - a), products rounded on their own: {int((blended == want).all(-1).sum())} of {disc.size} pixels bit-identical")


def main_headline():
    from oracle.oracle import Oracle

    path = os.path.join(HERE, "reference_exec_mesa.npz")
    old = np.load(path)
    out = {k: old[k] for k in old.files}
    zz = np.load(os.path.join(HERE, "reference_exec.npz"))
    if "--composite-only" not in sys.argv:
        headline(out, Oracle("f32"), zz)
    composite(out, zz)
    np.savez_compressed(path, **out)
    print(f"\n# wrote {os.path.relpath(path, ROOT)}: {os.path.getsize(path)} bytes, {len(out)} arrays")


def fuzz_report(store=True):
    """Section 10 of the report: the 24 random scenes of reference_exec_fuzz.npz (other planet scales, moved and rotated planets, cameras inside the
    layer, cube sizes 17 .. 128, shape volumes 24 .. 64, scenes without a cubemap) on Mesa against the interpreter's committed vectors.  No new vectors:
    the interpreter's are what the oracle and the HIP path are held to (tests/test_reference_exec.py)."""
    import json

    fz = np.load(os.path.join(HERE, "reference_exec_fuzz.npz"))
    print("\n## 10. the 24 random scenes of reference_exec_fuzz.npz, 40 x 24, level-0 sampler: Mesa against the interpreter's vectors")
    worst = {}
    fout = {"mesa_info": np.array(M.info()), "gallivm_perf": np.array(os.environ["GALLIVM_PERF"])}
    for k in range(RS.FUZZ_SEEDS):
        params = {kk: (tuple(v) if isinstance(v, list) else v) for kk, v in json.loads(str(fz[f"params_{k}"])).items()}
        _, cam_args, _, _, _ = RS.random_scene(k)
        cam = S.Camera(RS.FUZZ_W, RS.FUZZ_H, **cam_args)
        m = fz[f"cam_{k}"]
        cam.inv_projection, cam.inv_view, cam.view = m[0].copy(), m[1].copy(), m[2].copy()
        model, sun, depth = fz[f"model_{k}"], tuple(fz[f"sun_{k}"].tolist()), fz[f"depth_{k}"]
        tex = RS.fuzz_textures(k)
        lut, _ = M.run_bake({kk: params[kk] for kk in ("u_planet_radius", "u_atmosphere_height", "u_density")})
        mtex = dict(lut=lut, blue=tex["blue_noise"], shape=tex["shape"],
                    cubemap=tex["cubemap"] if tex["cubemap"] is not None else np.full((6, 1, 1), 255, dtype=np.uint8))  # unbound: the engine's white
        for shader in RS.fuzz_variants(k):
            rgba, disc, vary = M.run_frame(shader, None, params, np.linalg.inv(model), model, cam, depth, mtex, sun=sun)
            want = fz[f"rgba_{k}_{shader}"]
            fout[f"rgba_{k}_{shader}"] = rgba
            # (this fixture stores no discard mask: a discarded fragment and a kept one that evaluates to (0, 0, 0, 0) are the same pixel in it)
            same_disc = np.array_equal(np.all(rgba == 0.0, axis=-1), np.all(want == 0.0, axis=-1))
            note = "" if same_disc else "   ZERO SETS DIFFER"
            fin = np.isfinite(want)
            e = relerr(rgba[fin], want[fin]) if fin.any() else np.zeros(1)
            fam = shader.replace("planet_atmosphere_", "")
            w = worst.setdefault(fam, [0.0, 0.0, 0, 0])
            w[0], w[1], w[2], w[3] = max(w[0], float(e.max())), max(w[1], float(np.mean(e > 1e-4))), w[2] + 1, w[3] + int(same_disc and np.array_equal(np.isfinite(rgba), fin))
            print(f"   seed {k:2d} {fam:16s} R = {params['u_planet_radius']:8.2f} cube {0 if tex['cubemap'] is None else tex['cubemap'].shape[1]:4d}: {stats(rgba[fin], want[fin])}"
                  f"{note}", flush=True)
    print("   (seeds 15, 17: the v1 model's products reach 1e14 before its clamp; the fp32 oracle itself is 1e-2 / 1e-1 from the fp64 evaluation of those frames.)")
    print(f"   {'variant':18s} scenes   zero + finite sets identical      max      largest share beyond 1e-4")
    for fam, (a, b, n, ok) in worst.items():
        print(f"   {fam:18s} {n:4d}     {ok:4d}                            {a:8.2e}   {100 * b:6.3f} %")
    if store:   # tests/golden/reference_exec_mesa_fuzz.npz: Mesa's frames of the random scenes (inputs: reference_exec_fuzz.npz)
        path = os.path.join(HERE, "reference_exec_mesa_fuzz.npz")
        np.savez_compressed(path, **fout)
        print(f"   wrote {os.path.relpath(path, ROOT)}: {os.path.getsize(path)} bytes, {len(fout) - 2} frames")


if __name__ == "__main__":
    if "--fuzz-report" in sys.argv:
        fuzz_report()
    elif "--headline-only" in sys.argv or "--composite-only" in sys.argv:
        main_headline()
    else:
        main()
        fuzz_report()
