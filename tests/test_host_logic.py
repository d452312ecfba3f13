"""CPU-only tests of the host side: the C ABI library loads and exports every symbol include/atmo.h declares
(no compute without a GPU), it fails loudly without a device, sharding index math, scene generators, and the
pieces of the PlanetAtmosphere mirror that need no GPU."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from godot_atmosphere_shader_amd import scene as S
from godot_atmosphere_shader_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    from godot_atmosphere_shader_amd import _native as N
    return N.load().atmo_device_count() > 0


def test_library_exports_every_declared_symbol():
    from godot_atmosphere_shader_amd import _native as N
    from godot_atmosphere_shader_amd.build import build_native

    build_native()
    lib = N.load()
    declared = set()
    for name, want in (("atmo.h", N.CORE_SYMBOLS), ("atmo_debug.h", N.DEBUG_SYMBOLS)):
        header = open(os.path.join(ROOT, "include", name)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
        header = re.sub(r"#ifdef ATMO_WAVE_TRACE.*?#endif", "", header, flags=re.S)   # declared for the diagnostic build only; not exported otherwise
        syms = set(re.findall(r"\b(atmo_[a-z0-9_]+)\s*\(", header))
        assert syms == set(want), name
        declared |= syms
    # the header a host binds holds the calls that replace reference interfaces, not the experiment knobs (22 since round 6: atmo_render_tiles_split is
    # product API of the tile-sharded path)
    assert len(N.CORE_SYMBOLS) <= 22 and not set(N.CORE_SYMBOLS) & set(N.DEBUG_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.atmo_abi_version() == N.ABI_VERSION


def test_struct_layout_matches_header():
    from godot_atmosphere_shader_amd import _native as N
    # 16+16 floats, 2 ints, 3+3+1 floats, 4 ints
    assert C.sizeof(N.AtmoFrame) == (32 + 2 + 7 + 4) * 4


def test_no_gpu_fails_loudly():
    """Without an MI355X there is no fallback: create fails with ATMO_E_NO_DEVICE and says why."""
    from godot_atmosphere_shader_amd import _native as N
    if _has_gpu():
        pytest.skip("GPU present")
    lib = N.load()
    assert lib.atmo_device_count() < 0
    ctx = C.c_void_p()
    rc = lib.atmo_create(0, N.VARIANT_NO_CLOUDS, 0, 0, N.LIGHT_LUT, 0, C.byref(ctx))
    assert rc == N.ATMO_E_NO_DEVICE and not ctx.value
    assert b"no HIP device" in lib.atmo_last_error_string(None)
    from godot_atmosphere_shader_amd import PlanetAtmosphere
    with pytest.raises(N.AtmoError):
        PlanetAtmosphere()


def test_argument_validation_needs_no_gpu():
    from godot_atmosphere_shader_amd import _native as N
    lib = N.load()
    ctx = C.c_void_p()
    assert lib.atmo_create(0, 17, 0, 0, 0, 0, C.byref(ctx)) == N.ATMO_E_ARG
    assert lib.atmo_create(0, 0, 0, 0, N.LIGHT_DIRECT, 0, C.byref(ctx)) == N.ATMO_E_ARG  # direct needs light_steps
    assert lib.atmo_create(0, 0, -1, 0, 0, 0, C.byref(ctx)) == N.ATMO_E_ARG
    assert lib.atmo_create(0, 0, 0, 0, 0, 0, None) == N.ATMO_E_ARG
    assert lib.atmo_destroy(None) == N.ATMO_OK
    assert lib.atmo_set_param_f32(None, b"u_density", (C.c_float * 1)(1.0), 1) == N.ATMO_E_ARG
    # round 4 entry points: a null context is an argument error, not a crash
    assert lib.atmo_set_target_cleared(None, 1) == N.ATMO_E_ARG
    assert lib.atmo_set_sampler_lod(None, -1) == N.ATMO_E_ARG
    assert lib.atmo_render_tiles(None, None, None, None, None, 0, None) == N.ATMO_E_ARG


def test_row_bands_cover_and_partition():
    for h in (1, 7, 1080, 2160, 1081):
        for n in (1, 2, 3, 4, 8):
            bands = sharding.row_bands(h, n)
            assert len(bands) == n and bands[0][0] == 0 and bands[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
            sizes = [b[1] - b[0] for b in bands]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.row_bands(2, 4) == [(0, 0), (0, 1), (1, 1), (1, 2)]
    assert sharding.band_rect(1920, (135, 270)) == (0, 135, 1920, 270)
    with pytest.raises(ValueError):
        sharding.row_bands(10, 0)


def test_balanced_row_bands_equalise_hits():
    # P_space-like profile: empty rows top and bottom, a disc in the middle
    h = 1080
    y = np.arange(h) - h / 2
    cost = np.sqrt(np.maximum(430.0 ** 2 - y ** 2, 0.0))
    bands = sharding.balanced_row_bands(cost, 8)
    assert bands[0][0] == 0 and bands[-1][1] == h and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
    sums = [cost[a:b].sum() for a, b in bands]
    assert max(sums) / (sum(sums) / 8) < 1.02
    equal = [cost[a:b].sum() for a, b in sharding.row_bands(h, 8)]
    assert max(equal) / (sum(equal) / 8) > 1.5  # what balancing fixes
    assert sharding.balanced_row_bands(np.zeros(16), 4) == sharding.row_bands(16, 4)


def test_scene_generators_are_deterministic_and_well_formed():
    bn = S.make_blue_noise()
    assert bn.shape == (256, 256) and bn.dtype == np.uint8
    assert np.array_equal(np.bincount(bn.ravel(), minlength=256), np.full(256, 256))  # flat histogram, mean 127.5
    assert S.checksum(bn) == S.checksum(S.make_blue_noise())
    sh = S.make_shape_texture(16, cells=4)
    assert sh.shape == (16, 16, 16) and sh.std() > 10
    cm = S.make_coverage_cubemap(16)
    assert cm.shape == (6, 16, 16)
    d = S.cube_texel_directions(4)
    assert np.allclose(np.linalg.norm(d, axis=-1), 1.0)
    # Vulkan table: +X face has x as the major axis, s grows with -z, t grows with -y
    assert np.all(d[0, ..., 0] > 0.5) and d[0, 0, 0, 2] > d[0, 0, 3, 2] and d[0, 0, 0, 1] > d[0, 3, 0, 1]
    assert np.all(d[3, ..., 1] < -0.5) and np.all(d[4, ..., 2] > 0.5)


def test_camera_conventions():
    cam = S.Camera.from_pose(64, 36, "P_space")
    # centre pixel looks down -Z in view space; reversed-Z: depth 1 at near, 0 at far
    v = cam.inv_projection @ np.array([0.0, 0.0, 1.0, 1.0])
    assert (v[:3] / v[3])[2] == pytest.approx(-cam.near)
    v = cam.inv_projection @ np.array([0.0, 0.0, 0.0, 1.0])
    assert (v[:3] / v[3])[2] == pytest.approx(-cam.far, rel=1e-6)
    # SCREEN_UV origin top-left: ndc y = -1 is the top of the view (+Y up)
    top = cam.inv_projection @ np.array([0.0, -1.0, 1.0, 1.0])
    assert top[1] / top[3] > 0
    depth = S.depth_ground_sphere(cam)
    assert depth.dtype == np.float32 and depth.max() < 1.0 and depth.min() == 0.0
    hit = depth > 0
    assert 0.3 < hit.mean() < 0.6  # the ground sphere covers about half of the 16:9 frame from the demo camera


def test_srgb_to_linear_and_demo_params():
    assert S.srgb_to_linear(1.0) == pytest.approx(1.0)
    assert S.srgb_to_linear(0.0196078) == pytest.approx(0.0196078 / 12.92)
    assert S.srgb_to_linear(0.980392) == pytest.approx(((0.980392 + 0.055) / 1.055) ** 2.4)
    assert S.DEMO_SHADER_PARAMS["u_atmosphere_ambient_color"][2] == pytest.approx(((0.0431373 + 0.055) / 1.055) ** 2.4)


def test_shader_resources_and_transform2d():
    from godot_atmosphere_shader_amd import SHADERS, Transform2D, atmosphere_vertex, load_shader
    assert load_shader("res://addons/zylann.atmosphere/shaders/planet_atmosphere_clouds_high_m.gdshader") is SHADERS["planet_atmosphere_clouds_high_rm"]
    assert load_shader("planet_atmosphere_clouds").cloud_steps == 32
    assert load_shader("planet_atmosphere_clouds_high.gdshader").cloud_steps == 64
    assert all(s.view_steps == (16 if s.lite else 8) for s in SHADERS.values())
    assert len(SHADERS) == 7  # the reference's seven spatial shader variants
    with pytest.raises(FileNotFoundError):
        load_shader("nope.gdshader")
    names = [u["name"] for u in SHADERS["planet_atmosphere_no_clouds"].get_shader_uniform_list()]
    assert "u_optical_depth_texture" in names and "u_cloud_blend" not in names
    t = Transform2D().rotated(0.3).as_mat2_col_major()
    assert np.allclose(t, [math.cos(0.3), math.sin(0.3), -math.sin(0.3), math.cos(0.3)])
    cam = S.Camera.from_pose(16, 9, "P_space")
    planet, sun = atmosphere_vertex(cam.view, np.eye(4), S.DEMO_SUN_POSITION)
    assert planet.dtype == np.float32
    assert np.allclose(planet, [-0.357289, -0.105603, -157.92054], atol=1e-4)
    assert np.allclose(sun, [-0.357289, -0.105603, 478.677 - 157.92054], atol=1e-3)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "godot_atmosphere_shader_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
                assert "atmo_oracle" not in text, f
    # the developer tools and the example host stay on the product side as well (checks that need the oracle live under tests/checks/)
    for sub in ("tools", "examples"):
        for f in os.listdir(os.path.join(ROOT, sub)):
            if f.endswith((".py", ".sh", ".cpp", ".hip", ".c")):
                text = open(os.path.join(ROOT, sub, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f"{sub}/{f}"
    # bench.py: only the cpu_baseline leg (and the NoiseCubemap extra's one-core reference figure) may reach for it
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            assert "oracle" not in ast.dump(node), "bench.py imports the oracle at module level"
        if isinstance(node, ast.FunctionDef) and "oracle" in ast.dump(node):
            assert node.name in ("cpu_baseline", "bench_noise_cubemap"), f"bench.{node.name} touches the oracle"


def test_host_texture_layouts_against_the_oracle(oracle32):
    """The device layouts atmo_set_texture uploads (built on the host, no GPU needed): cubemap footprints carry the
    oracle's seamless texels (incl. the folded apron and the rounded-mean corners), shape footprints the repeat
    wrap, the LUT a clamp-to-edge apron."""
    from godot_atmosphere_shader_amd import _native as N

    lib = N.load()
    rng = np.random.default_rng(12)
    for n in (1, 2, 5, 16):
        cube = rng.integers(0, 256, (6, n, n), dtype=np.uint8)
        fp = np.zeros((6, n + 1, n + 1), dtype=np.uint32)
        assert lib.atmo_host_layout_cubemap(cube.ctypes.data_as(C.c_void_p), n, fp.ctypes.data_as(C.c_void_p)) == N.ATMO_OK
        for f in range(6):
            for j in range(n + 1):
                for i in range(n + 1):
                    w = int(fp[f, j, i])
                    got = [(w >> (8 * k)) & 255 for k in range(4)]
                    # word (i,j) starts at padded (i,j) = texel (i-1, j-1)
                    want = [oracle32.cube_texel(cube, f, i - 1 + di, j - 1 + dj) for (di, dj) in ((0, 0), (1, 0), (0, 1), (1, 1))]
                    assert got == want, (n, f, i, j)
    for n in (1, 3, 8):
        tex = rng.integers(0, 256, (n, n, n), dtype=np.uint8)
        fp = np.zeros((n, n, n), dtype=np.uint32)
        assert lib.atmo_host_layout_shape(tex.ctypes.data_as(C.c_void_p), n, fp.ctypes.data_as(C.c_void_p)) == N.ATMO_OK
        for k in range(n):
            for j in range(n):
                for i in range(n):
                    w = int(fp[k, j, i])
                    want = [tex[k, j, i], tex[k, j, (i + 1) % n], tex[k, (j + 1) % n, i], tex[k, (j + 1) % n, (i + 1) % n]]
                    assert [(w >> (8 * b)) & 255 for b in range(4)] == [int(v) for v in want]
    lut = rng.random((5, 7), dtype=np.float32)
    ap = np.zeros((7, 9), dtype=np.float32)
    assert lib.atmo_host_layout_lut(lut.ctypes.data_as(C.c_void_p), 7, 5, ap.ctypes.data_as(C.c_void_p)) == N.ATMO_OK
    assert np.array_equal(ap, np.pad(lut, 1, mode="edge"))
    assert lib.atmo_host_layout_cubemap(None, 4, fp.ctypes.data_as(C.c_void_p)) == N.ATMO_E_ARG


class _RecordingLib:
    """Stands in for libatmo_hip.so in the CPU tests of the node mirror: records what `_forward` uploads."""

    def __init__(self):
        self.uploads = {}

    def atmo_set_param_f32(self, ctx, name, ptr, n):
        self.uploads[name.decode()] = [float(ptr[i]) for i in range(n)]
        return 0


def _bare_node():
    from godot_atmosphere_shader_amd.planet_atmosphere import DefaultShader, PlanetAtmosphere

    node = object.__new__(PlanetAtmosphere)  # no context: the GPU is not needed for the upload logic
    node._lib = _RecordingLib()
    node._ctx = C.c_void_p()
    node._params = {}
    node._shader = DefaultShader
    node._uses_baked_optical_depth = False
    node._bake_pending = False
    return node


def test_source_color_uniforms_are_converted_on_upload_and_round_trip():
    """`source_color` uniforms hold sRGB values on the node (what the inspector shows) and are
    converted to linear when the material uploads them, as the engine does; `set(k, get(k))` -- the inspector's round
    trip over get_property_list -- does not change the uploaded floats; LinearColor opts out of the conversion."""
    from godot_atmosphere_shader_amd.planet_atmosphere import SHADER_DEFAULTS, LinearColor, load_shader

    node = _bare_node()
    node._shader = load_shader("planet_atmosphere_v1_clouds")
    up = node._lib.uploads
    node.set("shader_params/u_atmosphere_modulate", (1.0, 0.980392, 0.964706))
    assert up["u_atmosphere_modulate"] == pytest.approx(S.srgb_to_linear((1.0, 0.980392, 0.964706)).tolist(), rel=1e-6)
    # defaults as the inspector shows them (sRGB) upload as the library's linear defaults
    node.set("shader_params/u_day_color0", node.get("shader_params/u_day_color0"))
    assert node.get("shader_params/u_day_color0") == SHADER_DEFAULTS["u_day_color0"] == (0.5, 0.8, 1.0, 1.0)
    assert up["u_day_color0"] == pytest.approx([0.21404114, 0.60382734, 1.0, 1.0], rel=1e-6)  # alpha not converted
    node.set("shader_params/u_atmosphere_ambient_color", node.get("shader_params/u_atmosphere_ambient_color"))
    assert up["u_atmosphere_ambient_color"] == pytest.approx([0.0, 0.0, 0.002 / 12.92], rel=1e-6)
    # the round trip is a fixed point: set(k, get(k)) twice uploads the same floats
    for prop in node.get_property_list():
        k = prop["name"]
        v = node.get(k)
        if v is None or k.endswith("_texture") or k.endswith("_cubemap"):
            continue
        node.set(k, v)
        first = list(up[k[len("shader_params/"):]])
        node.set(k, node.get(k))
        assert up[k[len("shader_params/"):]] == first, k
    # hosts that keep linear colours opt out
    node.set_shader_parameter("u_atmosphere_modulate", LinearColor(0.25, 0.5, 0.75))
    assert up["u_atmosphere_modulate"] == [0.25, 0.5, 0.75]
    assert node.get_shader_parameter("u_atmosphere_modulate") == (0.25, 0.5, 0.75)
    node.set("shader_params/u_atmosphere_modulate", node.get("shader_params/u_atmosphere_modulate"))
    assert up["u_atmosphere_modulate"] == [0.25, 0.5, 0.75]  # the wrapper survives get(): still a no-op
    # non-colour uniforms are never touched
    node.set("shader_params/u_scattering_wavelengths", (0.5, 0.5, 0.5))
    assert up["u_scattering_wavelengths"] == [0.5, 0.5, 0.5]


def test_forward_z_camera_conventions():
    cam = S.Camera.from_pose(64, 36, "P_space", reverse_z=False)
    v = cam.inv_projection @ np.array([0.0, 0.0, 0.0, 1.0])
    assert (v[:3] / v[3])[2] == pytest.approx(-cam.near)
    v = cam.inv_projection @ np.array([0.0, 0.0, 1.0, 1.0])
    assert (v[:3] / v[3])[2] == pytest.approx(-cam.far, rel=1e-4)
    d = S.depth_ground_sphere(cam)
    assert d.max() == 1.0 and 0.9 < d.min() < 1.0 and np.array_equal(S.depth_far(cam), np.ones_like(d))
    rev = S.Camera.from_pose(64, 36, "P_space")
    assert np.allclose(cam.pixel_view_dirs(), rev.pixel_view_dirs())
    assert np.array_equal(d < 1.0, S.depth_ground_sphere(rev) > 0.0)


def test_pixel_coordinate_division_is_exact(tmp_path):
    """pixel_coord() of csrc/atmo_kernels.hip replaces (i + 0.5) / n by one Markstein correction: the exhaustive check
    (tools/uv_division.c, every n <= 65536 when run without argument) on the prefix n <= 6000, plus the large power-of-two
    and display sizes in numpy float32 with an exact FMA emulated in float64 (a * r and the residual fit 53 bits)."""
    import subprocess

    exe = tmp_path / "uv_division"
    subprocess.check_call(["gcc", "-O2", "-o", str(exe), os.path.join(ROOT, "tools", "uv_division.c"), "-lm"])
    out = subprocess.run([str(exe), "6000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "pixel_coord: 0 mismatches" in out.stdout and "unorm8_exact: 0 mismatches" in out.stdout
    f32, f64 = np.float32, np.float64
    for n in (7680, 8192, 15360, 16384, 32768, 65535, 65536):
        a = np.arange(n, dtype=f32) + f32(0.5)
        c = f32(n)
        r = f32(1.0) / c
        q0 = a * r
        # fma(-q0, c, a): the product of two float32 is exact in float64, and so is its difference to a here (|.| < 2^-20 |a|)
        res = (a.astype(f64) - q0.astype(f64) * f64(c)).astype(f32)
        q1 = (q0.astype(f64) + res.astype(f64) * f64(r)).astype(f32)  # float64 sum, then one float32 rounding
        assert np.array_equal(q1, a / c), n


def test_feedback_motion_estimate_matches_the_camera_motion():
    """The tile-order feedback predicts, from the matrices of consecutive frames, how far the picture's cost features move
    (atmo_api.hip feedback_motion_px; host arithmetic, no device): a pan of 1 degree moves everything by focal * 1 degree =
    12.3 px at 1920x1080 / 75 degrees; an orbit around the planet leaves the SILHOUETTE where it is (what a cloudless kernel's
    cost map is) but moves points fixed on the planet (the cloud pattern); a still camera moves nothing."""
    import bench
    from godot_atmosphere_shader_amd import _native as N
    from godot_atmosphere_shader_amd import scene as S
    from godot_atmosphere_shader_amd.planet_atmosphere import _to_native_frame, make_frame

    lib = N.load()
    w, h = 1920, 1080

    def frames(motion):
        cams = bench.motion_cameras(S, w, h, motion, 3)
        return [_to_native_frame(make_frame(c, np.eye(4), S.DEMO_SUN_POSITION, 0.0)) for c in cams]

    def px(a, b, surface):
        return lib.atmo_debug_motion_px(C.byref(a), C.byref(b), 108.0, surface)

    focal = 0.5 * h / np.tan(np.radians(37.5))
    f = frames(("pan", 1.0))
    assert abs(px(f[1], f[0], 0) - focal * np.radians(1.0)) < 0.15 * focal * np.radians(1.0) + 1.0   # larger off-axis: sec^2
    assert px(f[1], f[0], 0) >= focal * np.radians(1.0) * 0.99
    assert px(f[0], f[0], 1) == 0.0
    f = frames(("orbit", 1.0))
    assert px(f[1], f[0], 0) < 0.05                       # the disc stays where it is
    surf = px(f[1], f[0], 1)
    assert 15.0 < surf < 50.0                              # the sub-camera point moves R * 1 degree at 50 units' distance: ~ 26 px
    f = frames(("orbit", 0.0))
    assert px(f[1], f[0], 1) == 0.0


def test_bench_motion_helpers():
    """bench.py --motion: pose sequences are continuous (no jump when replayed ping-pong), a pan stays inside +-25 degrees and advances
    by the requested angle per frame, an orbit keeps its distance; the workload string names the cubemap sampler for cloud workloads only."""
    import bench
    from godot_atmosphere_shader_amd import scene as S

    assert bench.parse_motion("") is None and bench.parse_motion("static") is None
    assert bench.parse_motion("orbit:0.5") == ("orbit", 0.5) and bench.parse_motion("pan:5") == ("pan", 5.0)
    with pytest.raises(SystemExit):
        bench.parse_motion("spin:1")
    assert [bench.pingpong(i, 4) for i in range(9)] == [0, 1, 2, 3, 2, 1, 0, 1, 2] and bench.pingpong(5, 1) == 0
    cams = bench.motion_cameras(S, 192, 108, ("pan", 5.0), 24)
    yaw = [np.degrees(np.arctan2(c.inv_view[0, 2], c.inv_view[2, 2])) for c in cams]
    assert max(abs(y) for y in yaw) <= 25.0 + 1e-6
    steps = np.abs(np.diff(yaw))
    assert np.all(steps <= 5.0 + 1e-6) and np.median(steps) > 4.99                     # 5 degrees per frame except at the turning points
    assert all(np.allclose(c.inv_view[:3, 3], cams[0].inv_view[:3, 3]) for c in cams)  # the eye stays where it is
    cams = bench.motion_cameras(S, 192, 108, ("orbit", 1.0), 8)
    r = [np.hypot(c.inv_view[0, 3], c.inv_view[2, 3]) for c in cams]
    assert np.allclose(r, r[0]) and not np.allclose(cams[1].inv_view[:3, 3], cams[0].inv_view[:3, 3])
    assert bench.workload_suffix("no_clouds_32x8_direct", None) == "" and bench.workload_suffix("no_clouds_8", "lod0") == ""
    assert "LOD 0" in bench.workload_suffix("clouds_high", "lod0") and "linear-mipmap" in bench.workload_suffix("clouds_high_rm", None)


def test_bench_final_line_is_compact():
    """bench.py's LAST stdout line is the record the driver parses, and the driver keeps the last 8 KB of stdout: round 3's single line
    had grown to 41.7 KB (13 extras with prose) and BENCH_r03.parsed came back null.  compact_record() of that very run -- and of a run with
    ten times as many extras -- stays under 6 KB, keeps the contract's keys and both mandatory objects, and drops nothing it must carry."""
    import json

    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "round3", "bench_default.json")))
    assert len(json.dumps(full)) > 30000  # the line that broke the driver's parser
    full["config"]["mrays_per_s_feedback_off"] = 20400.0
    rec = bench.compact_record(full, "bench_detail.json")
    line = json.dumps(rec)
    assert len(line) < bench.COMPACT_LIMIT < 8192, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5) and rec["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert set(rec["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_avg_ms", "algorithmic_bytes_per_launch"}
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["achieved"] / 8000.0, rel=1e-5)
    assert set(rec["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert rec["config"]["workload"] and rec["config"]["mrays_per_s_feedback_off"] == 20400.0 and "model" not in rec["config"]
    assert all(isinstance(v, list) and len(v) == 2 for v in rec["extra"].values())   # name -> [Mrays/s, hbm fraction], no prose
    assert rec["extra"]["lut32"][0] == pytest.approx(full["extra"]["lut32"]["Mrays/s"], rel=1e-4)
    assert "clouds_high_rm@moving/pan:1" in rec["extra"] and "extra_truncated" not in rec
    # an --also list of any length cannot push the line over the limit
    many = dict(full, extra={f"{k}#{i}": v for i in range(10) for k, v in full["extra"].items()})
    rec2 = bench.compact_record(many, "bench_detail.json")
    assert len(json.dumps(rec2)) <= bench.COMPACT_LIMIT and rec2.get("extra_truncated") and rec2["value"] == rec["value"]
    # multi-GPU shape: the gather modes' rates and the configs[4] block survive
    multi = dict(full, n_gpus=8, extra={"config4_clouds_high_rm_3840x2160": {"Mrays/s_final_gather": 1.0, "Mrays/s_no_gather": 2.0, "workload": "x" * 500}})
    multi["config"] = dict(full["config"], mrays_per_s_final_gather=1.5, mrays_per_s_gather_every=1.25, shard="y" * 2000)
    rec3 = bench.compact_record(multi)
    assert len(json.dumps(rec3)) < bench.COMPACT_LIMIT and rec3["extra"]["config4_clouds_high_rm_3840x2160"]["Mrays/s_no_gather"] == 2.0
    assert rec3["config"]["mrays_per_s_gather_every"] == 1.25


def test_bench_marks_counters_of_another_build_as_stale(tmp_path, monkeypatch):
    """roofline.traffic and valu_roofline are READ from committed rocprofv3 summaries (profiles/round<N>/pmc_*.json), not measured in the
    run that prints them.  Every summary now carries the build id of the library that was profiled (tools/profile.sh -> atmo_build_id: sha256 of the
    kernel sources, headers and flags); when it differs from the loaded library's -- a kernel changed and nobody re-profiled -- the line says
    traffic_stale: true and carries no valu_roofline."""
    import json

    import bench
    from godot_atmosphere_shader_amd import _native as N, build

    assert N.load().atmo_build_id().decode() == build.source_id() == bench.loaded_build_id()
    d = tmp_path / "profiles" / "roundX"
    d.mkdir(parents=True)
    counters = {k: {"dispatches": 6, "mean_per_launch": v} for k, v in dict(FETCH_SIZE=8100.0, WRITE_SIZE=32400.0, SQ_INSTS_VALU=5.0e7, SQ_INSTS_VALU_TRANS_F32=4.0e6).items()}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PROFILE_DIR", "profiles/roundX")
    for stamp, stale in ((build.source_id(), False), ("0123456789abcdef", True), (None, True)):
        doc = {"pmc_per_launch": counters, "kernel_stats": {"avg_ns": 87000.0}}
        if stamp:
            doc["build_id"] = stamp
        (d / "pmc_direct32x8_1920x1080.json").write_text(json.dumps(doc))
        pmc = bench.pmc_summary("direct32x8", 1920, 1080)
        assert pmc["stale"] is stale and pmc["hbm_bytes"] == (8100.0 + 32400.0) * 1024.0
        rf = bench.hbm_roofline(0.087, 20, 1920 * 1080, pmc)
        assert rf["traffic_stale"] is stale and rf["build_id"] == build.source_id() and rf["traffic"] == pmc["hbm_bytes"]
        vr = bench.valu_roofline(pmc, 0.087)
        assert (vr is None) == stale
        rec = bench.compact_record({"metric": "m", "value": 1.0, "config": {}, "roofline": rf, **({"valu_roofline": vr} if vr else {})})
        assert rec["roofline"]["traffic_stale"] is stale and ("valu_roofline" in rec) == (not stale)
    assert bench.pmc_summary("no_such_workload", 1, 1) is None


def test_bench_detail_file(tmp_path, monkeypatch):
    import json

    import bench

    monkeypatch.setenv("ATMO_BENCH_DETAIL", str(tmp_path / "d.json"))
    assert bench.write_detail({"a": 1}) == str(tmp_path / "d.json") and json.load(open(tmp_path / "d.json")) == {"a": 1}
    monkeypatch.setenv("ATMO_BENCH_DETAIL", "/nonexistent-dir/x.json")
    assert bench.write_detail({"a": 1}) is None   # an unwritable place does not fail the bench
    monkeypatch.setenv("ATMO_BENCH_DETAIL", "")
    assert bench.write_detail({"a": 1}) is None


def _run_bench_as_typed(tmp_path, *argv):
    """`python bench.py <argv>` exactly as a driver would type it: no launcher, no WORLD_SIZE / RANK in the environment."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    env["ATMO_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p, (json.loads(lines[-1]) if lines and lines[-1].lstrip().startswith("{") else None)


def test_bench_gpus_n_launches_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus N` typed WITHOUT torch.distributed.run used to exit with a message for N > 1, so the first
    8-GPU node would have produced no scaling curve.  Now it starts the N ranks itself as a child process (bench.self_launch; before torch is
    imported, never an exec).  Run here on CPU through the test switch --backend gloo --stub-renderer (no kernel, frames are torch fills): two
    ranks rendezvous on 127.0.0.1, run the barrier / max-over-ranks timed loop with the final gather, the configs[4]-shaped loop after it,
    and rank 0's JSON record is the LAST stdout line of the command the user typed -- saying what ran (ranks, backend, gather mode) and that
    it was the stub."""
    p, rec = _run_bench_as_typed(tmp_path, "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--stub-renderer",
                                 "--width", "64", "--height", "36")
    assert p.returncode == 0, p.stderr[-2000:]
    assert rec is not None, p.stdout[-2000:]
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["backend"] == "gloo" and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["gather_mode"] == "final" and rec["shard_mode"] == "viewports" and rec["scaling"] == "weak"
    assert rec["stub"] is True and rec["data"] == "stub" and "STUB" in rec["config"]["workload"]
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["value"] == pytest.approx(2 * 64 * 36 * 3 / (rec["timed_region_ms"] * 1e-3) / 1e6, rel=1e-3)
    assert rec["config"]["mrays_per_s_no_gather"] > 0 and rec["config"]["mrays_per_s_gather_every"] > 0
    assert set(rec["extra"]["config4_clouds_high_rm_3840x2160"]) == {"Mrays/s_final_gather", "Mrays/s_gather_every_frame", "Mrays/s_no_gather"}
    assert "torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1" in p.stderr   # it says what it started
    # one frame in row bands, gathered every frame (strong scaling): same launch path
    p, rec = _run_bench_as_typed(tmp_path, "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--stub-renderer",
                                 "--width", "64", "--height", "36", "--shard", "bands")
    assert p.returncode == 0 and rec["ranks"] == 2 and rec["scaling"] == "strong" and rec["gather_mode"] == "every" and rec["shard_mode"] == "bands"


def test_bench_launch_errors_are_loud(tmp_path):
    """The child's exit code is the command's exit code; the test switches cannot reach the product path."""
    import bench

    p, rec = _run_bench_as_typed(tmp_path, "--gpus", "2", "--backend", "gloo")           # gloo without the stub: refused before anything starts
    assert p.returncode != 0 and rec is None and "--stub-renderer" in p.stderr
    p, rec = _run_bench_as_typed(tmp_path, "--gpus", "1", "--stub-renderer", "--backend", "gloo")
    assert p.returncode != 0 and rec is None
    if not _has_gpu():
        # the real N > 1 command on a host without GPUs: every rank fails loudly, the launcher's code comes back, no JSON line
        p, rec = _run_bench_as_typed(tmp_path, "--gpus", "2", "--steps", "1", "--warmup", "0")
        assert p.returncode != 0 and rec is None and "needs an MI355X" in p.stderr and "torch.distributed.run" not in p.stderr   # refused before anything is launched
    # under a launcher (WORLD_SIZE set) nothing is launched again
    os.environ["WORLD_SIZE"] = "2"
    try:
        assert bench.self_launch(type("A", (), {"gpus": 2})()) is None
    finally:
        del os.environ["WORLD_SIZE"]
    assert bench.self_launch(type("A", (), {"gpus": 1})()) is None


def test_whole_quad_exchange_registers_are_private():
    """The declared-sampler kernels read their quad partners' cube coordinates from the partner LANES inside inline-asm blocks that run in
    whole-quad mode (s_wqm_b64): lanes the compiler believes inactive write the blocks' registers.  tools/check_quad_regs.py compiles the
    kernels to ISA (hipcc -S, ~20 s) and verifies that in every kernel with an exchange block no instruction outside the blocks writes
    those VGPRs -- the property the source's QuadRegs (read-write operands spanning the kernel) is there to guarantee."""
    import subprocess
    import sys as _sys

    p = subprocess.run([_sys.executable, os.path.join(ROOT, "tools", "check_quad_regs.py")], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if "exchange blocks" in ln]
    assert p.returncode == 0, p.stdout + p.stderr
    assert len(lines) >= 11 and all(": ok;" in ln for ln in lines), p.stdout   # every KF_CUBE_LOD instantiation
    # round 5: the registers the blocks READ -- the march position the helper lanes read too -- are written inside the march's loop
    # nest by the position update alone (no copy made under a narrowed EXEC)
    assert all(ln.rstrip().endswith("only the position update writes them in the march") for ln in lines), p.stdout
    assert any("<49, 0, 1>" in ln and " 1 exchange" in ln for ln in lines) and any("<51, 0, 1>" in ln and " 2 exchange" in ln for ln in lines)
    assert any("0 with a stack frame, 0 scratch instructions" in ln for ln in p.stdout.splitlines()), p.stdout   # and no kernel spills


def test_headline_view_loop_sits_at_its_fast_position():
    """Round 6 (profiles/round6/ab_loop_phase.txt): the 32 x 8 direct-light kernel <4, 8, 1> -- the BASELINE configs[1] headline -- draws in 0.0865 ms when
    the first instruction of its 436-byte view loop lies 12 bytes into a 32-byte block of the instruction stream and in 0.094-0.096 ms at each of the other seven
    positions.  A change anywhere in front of that loop can move it by four bytes (that is what made earlier rounds' preamble and SGPR-cap experiments lose 8-10 %):
    this test reads the position from the library as built (tools/loop_phase.py) and fails until ATMO_LOOP_PAD (atmo_kernels.hip, march_atmosphere) puts it back."""
    import sys as _sys

    from godot_atmosphere_shader_amd.build import build_native

    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    _sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import loop_phase
    finally:
        _sys.path.pop(0)
    lib = build_native()
    for pattern, knob in (("atmo_render_kernelILi4ELi8ELi1E", "ATMO_LOOP_PAD"), (loop_phase.GEO_TWIN, "ATMO_LOOP_PAD_GEO")):   # ... and its twin behind the geometric order's lookup
        rows = loop_phase.view_loops(lib, pattern)
        assert len(rows) == 1, rows                      # one kernel matches, with one loop holding the seven-root cluster of the light march
        name, offset, phase, size = rows[0]
        assert phase == loop_phase.FAST_PHASE, (f"{name}: the view loop starts {phase} bytes into its 32-byte block (at +0x{offset:x}, {size} bytes); the measured-fast "
                                                f"position is {loop_phase.FAST_PHASE}: move {knob} by {((loop_phase.FAST_PHASE - phase) % 32) // 4}")


# ---- round 5: the host side of the C ABI without a device (atmo_debug_create_host_only; also what `make sanitize-host` runs) ---------------
def _host_ctx(variant, view_steps=0, cloud_steps=0, light_mode=0, light_steps=0):
    from godot_atmosphere_shader_amd import _native as N

    lib = N.load()
    ctx = C.c_void_p()
    assert lib.atmo_debug_create_host_only(variant, view_steps, cloud_steps, light_mode, light_steps, C.byref(ctx)) == N.ATMO_OK and ctx.value
    return lib, ctx


def test_uniform_table_on_a_host_only_context():
    """Every uniform of SURVEY.md 8b by the reference's name: shader defaults (`source_color` ones linear), set / get round trip, the ABI's
    error codes for unknown names and wrong counts -- on a context that owns no GPU, so the uniform table runs on the CPU box (and under
    ASan / UBSan: make sanitize-host).  Entry points that need the device fail loudly on such a context."""
    from godot_atmosphere_shader_amd import _native as N

    lib, ctx = _host_ctx(N.VARIANT_CLOUDS_HIGH_RM)
    defaults = {"u_planet_radius": [1.0], "u_atmosphere_height": [0.1], "u_density": [0.2], "u_scattering_strength": [20.0],
                "u_scattering_wavelengths": [700.0, 530.0, 440.0], "u_atmosphere_modulate": [1.0, 1.0, 1.0],
                "u_atmosphere_ambient_color": [0.0, 0.0, 0.002 / 12.92], "u_sphere_depth_factor": [0.0], "u_cloud_density_scale": [50.0],
                "u_cloud_bottom": [0.2], "u_cloud_top": [0.5], "u_cloud_blend": [0.5], "u_cloud_shape_invert": [0.0], "u_cloud_coverage_bias": [0.0],
                "u_cloud_shape_factor": [0.8], "u_cloud_shape_scale": [1.0], "u_cloud_coverage_rotation": [1.0, 0.0, 0.0, 1.0],
                "u_world_to_model_matrix": list(np.eye(4).reshape(-1)), "u_day_night_transition_scale": [2.0], "u_sun_position": [0.0, 0.0, 0.0],
                "u_clip_mode": [0.0], "u_day_color0": [0.21404114, 0.60382734, 1.0, 1.0], "u_night_color1": [0.03310477, 0.13286832, 0.60382734, 1.0]}
    rng = np.random.default_rng(2)
    for name, want in defaults.items():
        n = len(want)
        buf = (C.c_float * n)()
        assert lib.atmo_get_param_f32(ctx, name.encode(), buf, n) == N.ATMO_OK, name
        assert list(buf) == pytest.approx(want, rel=1e-6, abs=0), name
        new = rng.random(n).astype(np.float32)
        assert lib.atmo_set_param_f32(ctx, name.encode(), (C.c_float * n)(*new), n) == N.ATMO_OK
        assert lib.atmo_get_param_f32(ctx, name.encode(), buf, n) == N.ATMO_OK and np.array_equal(np.array(buf, dtype=np.float32), new)
        for wrong in {1, 2, 3, 4, 16} - {n}:
            assert lib.atmo_set_param_f32(ctx, name.encode(), (C.c_float * 16)(), wrong) == N.ATMO_E_ARG
            assert lib.atmo_get_param_f32(ctx, name.encode(), (C.c_float * 16)(), wrong) == N.ATMO_E_ARG
    one = (C.c_float * 1)(1.0)
    assert lib.atmo_set_param_f32(ctx, b"u_nope", one, 1) == N.ATMO_E_NAME and b"u_nope" in lib.atmo_last_error_string(ctx)
    assert lib.atmo_get_param_f32(ctx, b"", one, 1) == N.ATMO_E_NAME and lib.atmo_set_param_f32(ctx, None, one, 1) == N.ATMO_E_NAME
    assert lib.atmo_set_param_f32(ctx, b"u_density", None, 1) == N.ATMO_E_ARG
    # switches that live on the host
    assert lib.atmo_set_precision(ctx, 3) == N.ATMO_E_ARG and lib.atmo_set_precision(ctx, 0) == N.ATMO_OK and lib.atmo_set_precision(ctx, 1) == N.ATMO_OK
    assert lib.atmo_set_sampler_lod(ctx, 2) == N.ATMO_E_ARG and lib.atmo_set_sampler_lod(ctx, 0) == N.ATMO_OK
    assert lib.atmo_set_tile_feedback(ctx, 5) == N.ATMO_E_ARG and lib.atmo_set_tile_feedback(ctx, 0) == N.ATMO_OK
    assert lib.atmo_set_target_cleared(ctx, 1) == N.ATMO_OK and lib.atmo_set_host_double_precision(ctx, 1) == N.ATMO_OK
    w = C.c_int(-1)
    assert lib.atmo_get_texture_size(ctx, b"u_cloud_shape_texture", C.byref(w), None, None, None) == N.ATMO_OK and w.value == 0
    assert lib.atmo_get_texture_size(ctx, b"u_bogus", C.byref(w), None, None, None) == N.ATMO_E_NAME
    n = C.c_uint(7)
    assert lib.atmo_get_host_wait_stats(ctx, C.byref(n)) == N.ATMO_OK and n.value == 0
    # anything that needs the device fails loudly (no fallback)
    f = N.AtmoFrame()
    f.viewport_w = f.viewport_h = f.x1 = f.y1 = 8
    if not _has_gpu():
        assert lib.atmo_bake_optical_depth(ctx, None) in (N.ATMO_E_HIP, N.ATMO_E_NO_DEVICE)
    assert lib.atmo_render(ctx, C.byref(f), None, None, None) in (N.ATMO_E_ARG, N.ATMO_E_STATE)
    assert lib.atmo_destroy(ctx) == N.ATMO_OK
    bad = C.c_void_p()
    assert lib.atmo_debug_create_host_only(9, 0, 0, 0, 0, C.byref(bad)) == N.ATMO_E_ARG and not bad.value
    assert lib.atmo_debug_create_host_only(0, 0, 0, N.LIGHT_DIRECT, 0, C.byref(bad)) == N.ATMO_E_ARG


def test_per_frame_constants_follow_the_shader_text():
    """fill_consts (csrc/atmo_api.hip) evaluates the pixel-independent expressions of the shader once per draw, in fp32 and in the reference's
    operation order: held here, bit for bit, against a numpy float32 restatement written from the shader text -- camera position (main:136),
    sun direction (main:164), Rayleigh coefficients (v2:47-51), the cloud shell radii (clouds:260-261), view -> model transform (clouds:285-288),
    the march-distance cap (clouds:186-202), the raymarched light's tap schedule (clouds:108-115,129,138,143) -- and the level-0 certificate's
    constant against its formula, including ADVICE r4's contracting matrix."""
    from godot_atmosphere_shader_amd import _native as N
    from godot_atmosphere_shader_amd.planet_atmosphere import _to_native_frame, make_frame

    F = np.float32
    rng = np.random.default_rng(11)
    for trial in range(12):
        lib, ctx = _host_ctx(N.VARIANT_CLOUDS_HIGH_RM, cloud_steps=[0, 64, 200, 1][trial % 4])
        R, H = F(rng.choice([1.0, 100.0, 637.1])), None
        H = F(R * F(rng.uniform(0.03, 0.3)))
        cb, ct = F(rng.uniform(0.05, 0.3)), F(rng.uniform(0.4, 0.9))
        lam = rng.uniform(400, 750, 3).astype(F)
        strength, dscale = F(rng.uniform(0.5, 30)), F(rng.uniform(1, 80))
        a = rng.uniform(0, 6.28)
        rot = [np.array([np.cos(a), np.sin(a), -np.sin(a), np.cos(a)]), np.array([0.6, 0.0, 0.0, 0.6]), np.array([1.0, 0.25, 0.0, 1.0]),
               np.array([0.3, 0.0, 0.0, 3.0])][trial % 4].astype(F)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        model = np.eye(4)
        model[:3, :3], model[:3, 3] = q, rng.normal(size=3) * float(R)
        w2m = np.linalg.inv(model)
        for name, v in (("u_planet_radius", [R]), ("u_atmosphere_height", [H]), ("u_cloud_bottom", [cb]), ("u_cloud_top", [ct]),
                        ("u_scattering_wavelengths", lam), ("u_scattering_strength", [strength]), ("u_cloud_density_scale", [dscale]),
                        ("u_cloud_coverage_rotation", rot), ("u_world_to_model_matrix", S.col_major(w2m))):
            v = np.asarray(v, dtype=F)
            assert lib.atmo_set_param_f32(ctx, name.encode(), (C.c_float * len(v))(*v), len(v)) == N.ATMO_OK
        eye = rng.normal(size=3)
        eye = eye / np.linalg.norm(eye) * float(R) * rng.uniform(1.02, 3.0)
        cam = S.Camera(160, 90, eye=eye, target=model[:3, 3] * 0.1, near=0.05, far=50.0 * float(R))
        sun = tuple((rng.normal(size=3) * 50 * float(R)).tolist())
        frame = _to_native_frame(make_frame(cam, model, sun))
        cnt = C.c_int()
        cube_n = [256, 1024, 17, 4][trial % 4]
        assert lib.atmo_debug_frame_constants(ctx, C.byref(frame), cube_n, None, 0, C.byref(cnt)) == N.ATMO_OK and cnt.value == 81
        out = (C.c_float * 81)()
        assert lib.atmo_debug_frame_constants(ctx, C.byref(frame), cube_n, out, 80, None) == N.ATMO_E_ARG
        assert lib.atmo_debug_frame_constants(ctx, C.byref(frame), cube_n, out, 81, None) == N.ATMO_OK
        got = np.array(out, dtype=F)
        V = np.array(frame.inv_view_matrix, dtype=F)            # column-major
        pc, sc = np.array(frame.planet_center_viewspace, dtype=F), np.array(frame.sun_center_viewspace, dtype=F)
        # main:136  inv_view * vec4(0, 0, 0, 1), summed left to right
        cam_pos = np.array([F(F(F(V[r] * F(0)) + F(V[4 + r] * F(0))) + F(V[8 + r] * F(0))) + F(V[12 + r] * F(1)) for r in range(3)], dtype=F)
        assert np.array_equal(got[0:3], cam_pos)
        d = (sc - pc).astype(F)                                   # main:164 normalize = v * (1 / sqrt(dot))
        inv = F(1) / np.sqrt(F(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])))
        sun_dir = (d * inv).astype(F)
        assert np.array_equal(got[3:6], sun_dir)
        assert got[6] == F(R + H)
        pow4 = lambda x: F(F(F(x * x) * x) * x)                   # util.gdshaderinc pow4
        assert np.array_equal(got[7:10], np.array([F(pow4(F(F(400) / l)) * strength) for l in lam], dtype=F))
        bottom, top = F(R + F(cb * H)), F(R + F(ct * H))          # clouds:260-261
        assert (got[10], got[11], got[12], got[13]) == (bottom, top, F(top - bottom), F(F(1) / F(top - bottom)))
        A = np.asarray(S.col_major(w2m), dtype=F)                 # clouds:285  u_world_to_model_matrix * INV_VIEW_MATRIX
        M = np.zeros(16, dtype=F)
        for col in range(4):
            for row in range(4):
                M[col * 4 + row] = F(F(F(A[row] * V[col * 4]) + F(A[4 + row] * V[col * 4 + 1])) + F(A[8 + row] * V[col * 4 + 2])) + F(A[12 + row] * V[col * 4 + 3])
        assert np.array_equal(got[18:34], M)
        origin = np.array([F(F(F(M[r] * F(0)) + F(M[4 + r] * F(0))) + F(M[8 + r] * F(0))) + F(M[12 + r] * F(1)) for r in range(3)], dtype=F)
        sun_m = np.array([F(F(F(M[r] * sun_dir[0]) + F(M[4 + r] * sun_dir[1])) + F(M[8 + r] * sun_dir[2])) + F(M[12 + r] * F(0)) for r in range(3)], dtype=F)
        assert np.array_equal(got[34:37], origin) and np.array_equal(got[37:40], sun_m)
        # clouds:186-202  the march-distance cap
        gt = F(R / top)
        space = F(F(F(0.5) * np.sqrt(F(F(1) - F(gt * gt)))) * bottom)
        groundd = F(F(3) * space)
        ln = np.sqrt(F(F(F(origin[0] * origin[0]) + F(origin[1] * origin[1])) + F(origin[2] * origin[2])))
        t = np.clip(F(F(ln - bottom) / F(F(top * F(1.05)) - bottom)), F(0), F(1))
        sm = F(F(t * t) * F(F(3) - F(F(2) * t)))
        assert got[40] == F(F(groundd * F(F(1) - sm)) + F(space * sm))
        steps = [64, 64, 200, 1][trial % 4]
        assert got[41] == F(F(1) / F(steps))
        # clouds:104-151  get_light_raymarched: step_len grows x 1.2 after every tap
        step_len = F(F(F(top - bottom) * F(0.15)) * F(F(1) / F(6)))
        for i in range(6):
            off = F(F(i) * step_len)
            assert got[42 + i] == off and got[48 + i] == F(step_len * dscale)
            assert np.array_equal(got[54 + 3 * i:57 + 3 * i], (off * sun_m).astype(F))
            step_len = F(step_len * F(1.2))
        # the level-0 certificate: 1 / C, C = 0.97 * 4 (1 - 2/n)^2 / (n max(sigma, 1))^2; withheld (inf) off the fast path, for tiny faces,
        # and when a singular value of the coverage matrix leaves [0.5, 2]
        sv = np.linalg.svd(np.array([[rot[0], rot[2]], [rot[1], rot[3]]], dtype=np.float64), compute_uv=False)
        offered = cube_n in (256, 1024, 4) and sv.max() <= 2.0 and sv.min() >= 0.5
        if offered:
            want = 1.0 / (0.97 * 4.0 * (1.0 - 2.0 / cube_n) ** 2 / (cube_n * max(sv.max(), 1.0)) ** 2)
            assert got[72] == pytest.approx(want, rel=3e-6), (trial, got[72], want)
        else:
            assert np.isinf(got[72])
        assert got[73] == F(steps - 1) and got[74] == pytest.approx((steps + 1) * 2.07e-7, rel=1e-6)
        assert got[76] == F(F(1) / F(160)) and got[77] == F(F(1) / F(90))
        assert lib.atmo_destroy(ctx) == N.ATMO_OK
