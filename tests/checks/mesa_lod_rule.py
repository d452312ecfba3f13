#!/usr/bin/env python3
"""the 12 MINIFIED frames of profiles/round5/mesa_pin.txt section 5 (48 x 27, the demo scene, the sampler the reference declares)
executed by Mesa / llvmpipe HERE, against the CPU oracle under
  cube_lod = 1   the stated convention (no derivative from a quad partner that does not reach the fetch)
  cube_lod = 2   lock-step quads: the partner contributes the coordinate its lane would hold (what mesa_pin 11e isolated); 3: only helper pixels below the viewport
and the CHECKER options of OracleConfig.lod_log2_fast: bit 0 = lambda's log2 piecewise linear (llvmpipe's level-of-detail unit), bit 1 = rho as a 3-D distance on
the cube, bit 2 = the quotient RULE d(s / ma) = (ds ma - s dma) / ma^2 with the lane's own ma instead of the exact difference of the two projections.
--save writes Mesa's 12 frames (tests/golden/reference_exec_mesa_lod.npz: the fixture of tests/test_reference_mesa.py::test_minified_frames_...).
Build container only (needs /root/reference and the image's Mesa).  python tests/checks/mesa_lod_rule.py [--save tests/golden/reference_exec_mesa_lod.npz]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("GALLIVM_PERF", "no_quad_lod,no_aos_sampling")

import mesa_exec as M  # noqa: E402
import reference_scenes as RS  # noqa: E402
import vm_textures as T  # noqa: E402
from godot_atmosphere_shader_amd import scene as S  # noqa: E402
from godot_atmosphere_shader_amd.planet_atmosphere import make_frame  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def relerr(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def stats(a, b):
    e = relerr(a, b)
    return f"max {e.max():.2e} p99 {np.percentile(e, 99):.1e} beyond 1e-4 {100.0 * np.mean(e > 1e-4):6.3f} %"


def main():
    oracle = Oracle("f32")
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_exec.npz"))
    W, H = RS.W, RS.H
    blue, shape, cube = S.make_blue_noise(), S.make_shape_texture(RS.SHAPE_N), S.make_coverage_cubemap(RS.CUBE_N)
    params, model = RS.scenes()["demo"]
    w2m = np.linalg.inv(model)
    tex = dict(lut=z["lut_demo"], blue=blue, shape=shape, cubemap=cube)
    chain = T.mip_chain(cube)
    otex = dict(blue_noise=blue, shape=shape, cubemap=oracle.cubemap_mip_chain(cube), optical_depth=z["lut_demo"])
    oparams = dict(params, u_world_to_model_matrix=S.col_major(w2m))
    save = {}
    print(f"# {M.info()}  GALLIVM_PERF={os.environ['GALLIVM_PERF']}")
    print("# Mesa's frame against the oracle's, |a - b| / max(1, |b|) per channel")
    for pose in RS.LOD_POSES:
        cam = RS.camera_from_fixture(z, W, H, pose)
        depth = z[f"depth_demo_{pose}"]
        frame = make_frame(cam, model, S.DEMO_SUN_POSITION, 0.0)
        for shader in RS.LOD_VARIANTS:
            rgba, disc, _ = M.run_frame(shader, None, params, w2m, model, cam, depth, tex, cube_chain=chain)
            save[f"mesa_lod_rgba_{pose}_{shader}"] = rgba
            row = f"   {pose:8s} {shader.replace('planet_atmosphere_', ''):16s}"
            for name, cfg in (("stated (1)", dict(cube_lod=1)), ("+ quotient rule", dict(cube_lod=1, lod_log2_fast=4)), ("+ fast log2", dict(cube_lod=1, lod_log2_fast=1)),
                              ("+ BOTH", dict(cube_lod=1, lod_log2_fast=5)), ("lock-step (2) + both", dict(cube_lod=2, lod_log2_fast=5)),
                              ("helpers below the viewport lock-step (3) + both", dict(cube_lod=3, lod_log2_fast=5)), ("stated + 3-D rho + fast log2", dict(cube_lod=1, lod_log2_fast=3))):
                orc, _ = oracle.render(oparams, otex, dict(RS.VARIANTS[shader], **cfg), frame, depth, nthreads=8)
                row += f"\n        {name:48s} {stats(rgba, orc)}"
                if name == "+ BOTH":
                    e = relerr(rgba, orc)
                    pix = e.max(axis=-1)
                    bad = np.argwhere(pix[:H - 1] > 1e-4)
                    row += (f"\n        {'   of which rows 0..%d' % (H - 2):48s} max {e[:H - 1].max():.2e} beyond 1e-4 {100.0 * np.mean(e[:H - 1] > 1e-4):6.3f} %  pixels "
                            f"{[(int(y), int(x)) for y, x in bad]};  last row (its vertical partners are helper pixels BELOW the viewport): max {e[H - 1].max():.2e}, "
                            f"{int((pix[H - 1] > 1e-4).sum())} pixels beyond 1e-4")
            print(row, flush=True)
    if "--save" in sys.argv:
        path = sys.argv[sys.argv.index("--save") + 1]
        np.savez_compressed(path, mesa_info=np.array(M.info()), gallivm_perf=np.array(os.environ["GALLIVM_PERF"]), **save)
        print("saved", path)


if __name__ == "__main__":
    main()
