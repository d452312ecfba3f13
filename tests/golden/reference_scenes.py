"""The scenes tests/golden/reference_exec.npz was generated for (shared by make_reference_vectors.py, which needs the
reference tree, and tests/test_reference_exec.py, which does not)."""
import numpy as np

from godot_atmosphere_shader_amd import scene as S
from godot_atmosphere_shader_amd.demo import demo_params

W, H = 48, 27
SHAPE_N, CUBE_N = 32, 64

# reference shader file -> the step-count configuration it #defines (the checker's config dict)
VARIANTS = {
    "planet_atmosphere_no_clouds": dict(view_steps=8),
    "planet_atmosphere_clouds": dict(view_steps=8, cloud_steps=32),
    "planet_atmosphere_clouds_high": dict(view_steps=8, cloud_steps=64),
    "planet_atmosphere_clouds_high_rm": dict(view_steps=8, cloud_steps=64, cloud_light_rm=1),
    "planet_atmosphere_v1_no_clouds": dict(view_steps=16, lite=1),
    "planet_atmosphere_v1_clouds": dict(view_steps=16, lite=1, cloud_steps=32),
    "planet_atmosphere_v1_clouds_high": dict(view_steps=16, lite=1, cloud_steps=64),
}
POSES = ["P_space", "P_ground", "P_limb", "P_clouds", "P_night"]

# parameter sets: the demo scene, and a second scene that takes the branches the demo leaves alone
ALT_MODEL = np.array([[0.96, 0.0, 0.28, 0.0], [0.0, 1.0, 0.0, 0.0], [-0.28, 0.0, 0.96, 0.0], [1.5, -2.0, 0.75, 1.0]],
                     dtype=np.float64)  # column-major rows below are COLUMNS: rotation about y + translation


def scenes():
    demo = demo_params()
    alt = demo_params(u_cloud_shape_invert=0.0, u_sphere_depth_factor=0.35, u_cloud_blend=0.2, u_cloud_coverage_bias=0.08,
                      u_cloud_density_scale=35.0, u_cloud_shape_factor=0.6, u_cloud_shape_scale=0.37, u_density=0.17,
                      u_scattering_strength=14.0, u_scattering_wavelengths=(680.0, 550.0, 450.0),
                      u_atmosphere_modulate=(0.9, 1.0, 0.8), u_atmosphere_ambient_color=(0.01, 0.0, 0.02),
                      u_cloud_bottom=0.25, u_cloud_top=0.6, u_day_night_transition_scale=1.4,
                      u_day_color0=(0.3, 0.7, 1.0, 1.0), u_day_color1=(0.9, 0.6, 0.3, 1.0),
                      u_night_color0=(0.02, 0.05, 0.3, 1.0), u_night_color1=(0.1, 0.0, 0.2, 1.0))
    return {"demo": (demo, np.eye(4)), "alt": (alt, ALT_MODEL.T.copy())}




# BASELINE.json sizes: (reference shader file, viewport, pose, rows executed).  configs[2] = clouds_high 1920x1080,
# configs[3] = clouds_high_rm 3840x2160; whole rows spread over the frame incl. the limb rows of P_space.
FULL_SIZE = [
    ("planet_atmosphere_no_clouds", 1920, 1080, "P_space", (40, 330, 539, 1000)),
    ("planet_atmosphere_clouds_high", 1920, 1080, "P_space", (40, 200, 400, 539, 540, 700, 900, 1040)),
    ("planet_atmosphere_clouds_high", 1920, 1080, "P_clouds", (0, 300, 539, 800, 1079)),
    ("planet_atmosphere_clouds_high_rm", 3840, 2160, "P_space", (80, 400, 800, 1079, 1400, 1800, 2080)),
    ("planet_atmosphere_clouds_high_rm", 3840, 2160, "P_clouds", (0, 600, 1079, 1600, 2159)),
]


# ---- round 3 vectors (reference_exec_r3.npz) -------------------------------------------------------------------------------
# the reference's DECLARED cubemap sampler (linear-mipmap, implicit LOD: cloud_funcs.gdshaderinc:15,45) on the four cloud variants
LOD_VARIANTS = ["planet_atmosphere_clouds", "planet_atmosphere_clouds_high", "planet_atmosphere_clouds_high_rm",
                "planet_atmosphere_v1_clouds_high"]
LOD_POSES = ["P_space", "P_clouds", "P_limb"]
# ... and at BASELINE.json's sizes with the bench's 256^2 cubemap: (shader, w, h, pose, rows); the executed rows include each
# row's vertical quad partner (row ^ 1)
LOD_FULL_SIZE = [
    ("planet_atmosphere_clouds_high", 1920, 1080, "P_space", (200, 540, 901)),
    ("planet_atmosphere_clouds_high", 1920, 1080, "P_clouds", (300, 801)),
    ("planet_atmosphere_clouds_high_rm", 3840, 2160, "P_space", (801, 1400)),
]
# round 5 (reference_exec_r5.npz): the declared sampler on EVERY cloud row of FULL_SIZE above (25 rows, limb rows of P_space included; the
# executed rows again include each row's vertical quad partner) -- the kernels bench.py reports for configs[2] / configs[3] since round 4
LOD_FULL_SIZE_R5 = [c for c in FULL_SIZE if "clouds" in c[0].replace("no_clouds", "")]
# north_star's 32 view steps and the 64 of atmosphere_funcs_v2.gdshaderinc:42-43, through the reference text with
# ATMOSPHERE_RAYMARCH_STEPS forced over the file's own #define (planet_atmosphere_no_clouds.gdshader:4)
VIEW_STEP_COUNTS = [32, 64]
VIEW_STEP_ROWS = ("planet_atmosphere_no_clouds", 1920, 1080, "P_space", (40, 330, 539, 1000))


def quad_partners(width, rows, height):
    """Lane indices of the horizontal / vertical 2 x 2 quad partner of every pixel of the given rows (row-major lanes), -1
    where the partner pixel is outside the viewport or its row is not executed."""
    rows = list(rows)
    pos = {r: k for k, r in enumerate(rows)}
    px, py = np.meshgrid(np.arange(width), np.asarray(rows))
    lane = np.arange(width * len(rows)).reshape(len(rows), width)
    qx = np.where((px ^ 1) < width, lane // width * width + (px ^ 1), -1)
    prow = np.array([pos.get(r ^ 1, -1) if (r ^ 1) < height else -1 for r in rows])
    qy = np.where(prow[:, None] >= 0, prow[:, None] * width + px, -1)
    return qx.reshape(-1), qy.reshape(-1)


def camera_matrices(cam):
    """What a fixture stores of a camera (numpy's 4x4 inverse may differ in the last bit between hosts)."""
    return np.stack([cam.inv_projection, cam.inv_view, cam.view]).astype(np.float64)


def camera_from_fixture(z, w, h, pose):
    cam = S.Camera.from_pose(w, h, pose)
    m = z[f"cam_{w}x{h}_{pose}"]
    cam.inv_projection, cam.inv_view, cam.view = m[0].copy(), m[1].copy(), m[2].copy()
    return cam


# ---- random scenes (reference_exec_fuzz.npz): planets, cameras, suns and shader parameters the demo does not cover -------
FUZZ_SEEDS = 24
FUZZ_W, FUZZ_H = 40, 24


def fuzz_variants(k):
    """Three of the seven reference shader files per seed, rotating so that each is used six times."""
    names = list(VARIANTS)
    return [names[(3 * k + j) % len(names)] for j in range(3)]


def fuzz_textures(k):
    shape_n = [32, 24, 16, 48][k % 4]          # 24 and 48 are not powers of two
    cube_n = [64, 32, 17, 128][k % 4]
    return dict(blue_noise=S.make_blue_noise(k + 1), shape=S.make_shape_texture(shape_n, seed=k, cells=4),
                cubemap=None if k % 5 == 4 else S.make_coverage_cubemap(cube_n, seed=k))


def random_scene(k):
    """Returns (params, camera arguments, sun position, planet model matrix, depth kind) for seed k."""
    rng = np.random.default_rng(7000 + k)
    R = float(rng.choice([1.0, 10.0, 100.0, 637.1]))
    H = R * float(rng.uniform(0.03, 0.25))
    dens_target = float(rng.uniform(0.3, 3.0))  # vertical optical depth rho^2 * H / 4 of order one
    a = float(rng.uniform(0, 2 * np.pi))
    params = demo_params(
        u_planet_radius=R, u_atmosphere_height=H, u_density=float(np.sqrt(4.0 * dens_target / H)),
        u_scattering_strength=float(rng.uniform(0.3, 3.0)),
        u_scattering_wavelengths=(float(rng.uniform(600, 750)), float(rng.uniform(500, 580)), float(rng.uniform(400, 480))),
        u_atmosphere_modulate=tuple(rng.uniform(0.5, 1.0, 3).tolist()),
        u_atmosphere_ambient_color=tuple(rng.uniform(0.0, 0.01, 3).tolist()),
        u_sphere_depth_factor=float(rng.choice([0.0, 0.35, 1.0])),
        u_cloud_density_scale=float(rng.uniform(0.5, 60.0) / H * 8.0),
        u_cloud_bottom=float(rng.uniform(0.05, 0.3)), u_cloud_top=float(rng.uniform(0.4, 0.9)),
        u_cloud_blend=float(rng.uniform(0.0, 1.0)), u_cloud_shape_invert=float(rng.choice([0.0, 1.0])),
        u_cloud_coverage_bias=float(rng.uniform(-0.2, 0.2)), u_cloud_shape_factor=float(rng.uniform(0.0, 1.2)),
        u_cloud_shape_scale=float(rng.uniform(0.05, 0.4) * 100.0 / R),
        u_day_night_transition_scale=float(rng.uniform(0.5, 3.0)),
        u_day_color0=tuple(rng.uniform(0, 1, 3).tolist()) + (1.0,), u_day_color1=tuple(rng.uniform(0, 1, 3).tolist()) + (1.0,),
        u_night_color0=tuple(rng.uniform(0, 0.5, 3).tolist()) + (1.0,), u_night_color1=tuple(rng.uniform(0, 0.5, 3).tolist()) + (1.0,),
        u_cloud_coverage_rotation=(float(np.cos(a)), float(np.sin(a)), float(-np.sin(a)), float(np.cos(a))),
    )
    alt = float(rng.choice([0.02 * H, 0.4 * H, 0.9 * H, 1.5 * H, 0.6 * R, 2.0 * R]))
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    center = rng.normal(size=3) * R * 0.05 if k % 2 else np.zeros(3)   # every second planet is off the origin ...
    yaw = float(rng.uniform(0, 2 * np.pi)) if k % 2 else 0.0           # ... and rotated about y
    model = np.array([[np.cos(yaw), 0, np.sin(yaw), center[0]], [0, 1, 0, center[1]], [-np.sin(yaw), 0, np.cos(yaw), center[2]],
                      [0, 0, 0, 1]], dtype=np.float64)
    eye = center + d * (R + alt)
    tangent = np.cross(d, rng.normal(size=3))
    tangent /= np.linalg.norm(tangent)
    target = eye + tangent * R * 0.5 - d * R * float(rng.uniform(-0.1, 0.6))
    sun = rng.normal(size=3)
    sun = center + sun / np.linalg.norm(sun) * R * 50.0
    cam_args = dict(eye=eye, target=target, fovy_deg=float(rng.uniform(40, 90)), near=0.05 * H, far=20.0 * R)
    return params, cam_args, tuple(float(v) for v in sun), model, ("ground" if k % 3 else "far")


# ---- round 4 vectors (reference_exec_r4.npz): thin atmospheres at 64 view steps; seeds of tests/test_gpu_parity.py::_random_scene ---------
R4_THIN_SEEDS = (55, 91, 13, 43)
