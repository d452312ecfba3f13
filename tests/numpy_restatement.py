"""A second, independent restatement of the reference's fragment path -- vectorised numpy, float64 -- written
directly from the GDShader text (paths under /root/reference/addons/zylann.atmosphere/shaders/).  TEST
INFRASTRUCTURE: it exists to catch transcription errors in oracle/atmo_oracle.c (tests/test_oracle_kat.py compares
the two); it shares no code with the oracle or the product.

Scope: atmosphere_fragment (include/planet_atmosphere_main.gdshaderinc:106-197), compute_atmosphere_v2
(include/atmosphere_funcs_v2.gdshaderinc:32-101) with the baked LUT, compute_atmosphere (v1,
include/atmosphere_funcs_v1.gdshaderinc:15-63), render_clouds / raymarch_cloud / get_density / get_light*
(include/cloud_funcs.gdshaderinc) with an UNSET coverage cubemap (coverage texel = 1; the cubemap sampler has its own
known-answer tests) and the trilinear repeat shape texture.
"""
import numpy as np


def _mix(a, b, t):
    return a * (1.0 - t) + b * t


def _clamp(x, lo, hi):
    return np.minimum(np.maximum(x, lo), hi)


def _smoothstep(e0, e1, x):
    t = _clamp((x - e0) / (e1 - e0), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def _dot(a, b):
    return (a * b).sum(axis=-1)


def _length(a):
    return np.sqrt(_dot(a, a))


def _normalize(a):
    return a / _length(a)[..., None]


def ray_sphere(center, radius, origin, direction):
    """util.gdshaderinc:20-40.  Returns (x, y) arrays; (1e6, 1e6) where missed."""
    oc = origin - center
    b = _dot(oc, direction)
    qc = oc - b[..., None] * direction
    h = radius * radius - _dot(qc, qc)
    miss = h < 0.0
    hs = np.sqrt(np.where(miss, 0.0, h))
    return np.where(miss, 1000000.0, -b - hs), np.where(miss, 1000000.0, -b + hs)


def atmosphere_density(p, height):
    """atmosphere_common.gdshaderinc:12-24"""
    sd = height - p["u_planet_radius"]
    h = _clamp(sd / p["u_atmosphere_height"], 0.0, 1.0)
    y = 1.0 - h
    return y * y * y * p["u_density"]


def sample_lut(lut, u, v):
    """texture(sampler2D, repeat_disable): bilinear, clamp to edge, texel centres at (i + 0.5) / N."""
    h, w = lut.shape
    x = u * w - 0.5
    y = v * h - 0.5
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = x - x0
    fy = y - y0
    i0 = np.clip(x0.astype(int), 0, w - 1)
    i1 = np.clip(x0.astype(int) + 1, 0, w - 1)
    j0 = np.clip(y0.astype(int), 0, h - 1)
    j1 = np.clip(y0.astype(int) + 1, 0, h - 1)
    lut = lut.astype(np.float64)
    top = _mix(lut[j0, i0], lut[j0, i1], fx)
    bot = _mix(lut[j1, i0], lut[j1, i1], fx)
    return _mix(top, bot, fy)


def sample_shape(tex, p):
    """texture(sampler3D, repeat_enable): trilinear, wrap.  tex indexed [z, y, x], values byte / 255."""
    n = tex.shape[0]
    q = p * n - 0.5
    q0 = np.floor(q)
    f = q - q0
    i0 = np.mod(q0.astype(int), n)
    i1 = np.mod(q0.astype(int) + 1, n)
    t = tex.astype(np.float64) / 255.0
    x0, y0, z0 = i0[..., 0], i0[..., 1], i0[..., 2]
    x1, y1, z1 = i1[..., 0], i1[..., 1], i1[..., 2]
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    c00 = _mix(t[z0, y0, x0], t[z0, y0, x1], fx)
    c10 = _mix(t[z0, y1, x0], t[z0, y1, x1], fx)
    c01 = _mix(t[z1, y0, x0], t[z1, y0, x1], fx)
    c11 = _mix(t[z1, y1, x0], t[z1, y1, x1], fx)
    return _mix(_mix(c00, c10, fy), _mix(c01, c11, fy), fz)


def compute_atmosphere_v2(p, lut, steps, ray_dir, center, t_begin, t_end, sun_dir, jitter):
    """atmosphere_funcs_v2.gdshaderinc:32-101 (ray_origin = 0)."""
    lam = np.asarray(p["u_scattering_wavelengths"], dtype=np.float64)
    coeff = (400.0 / lam) ** 4 * p["u_scattering_strength"]
    step_len = (t_end - t_begin) / float(steps)
    total = np.zeros(ray_dir.shape)
    view_od = np.zeros(ray_dir.shape[:-1])
    alpha = np.zeros(ray_dir.shape[:-1])
    pos = ray_dir * t_begin[..., None]
    for _ in range(steps):
        # get_baked_optical_depth, v2:14-29
        rel = pos - center
        dist = _length(rel)
        height_ratio = _clamp((dist - p["u_planet_radius"]) / p["u_atmosphere_height"], 0.0, 1.0)
        up = rel / dist[..., None]
        uvx = 0.5 + 0.5 * _dot(up, np.broadcast_to(sun_dir, up.shape))
        sun_od = sample_lut(lut, uvx, height_ratio)
        local_density = atmosphere_density(p, dist) * p["u_density"]
        view_od = view_od + local_density * step_len
        transmittance = np.exp(-(sun_od + view_od)[..., None] * coeff)
        total = total + (local_density * step_len)[..., None] * transmittance * coeff
        vt = np.exp(-local_density * step_len)
        alpha = alpha + (1.0 - vt) * (1.0 - alpha)
        pos = pos + ray_dir * step_len[..., None]
    total = _clamp(total + np.asarray(p["u_atmosphere_ambient_color"], dtype=np.float64), 0.0, 1.0)
    alpha = _clamp(alpha + jitter * 0.02, 0.0, 0.99)
    total = total * np.asarray(p["u_atmosphere_modulate"], dtype=np.float64)
    return total, alpha


def compute_atmosphere_v1(p, steps, ray_dir, center, t_begin, t_end, sun_dir):
    """atmosphere_funcs_v1.gdshaderinc:15-63"""
    inv_steps = 1.0 / float(steps)
    step_len = (t_end - t_begin) * inv_steps
    stepv = ray_dir * step_len[..., None]
    pos = ray_dir * t_begin[..., None]
    factor = np.ones(ray_dir.shape[:-1])
    light_sum = np.zeros(ray_dir.shape[:-1])
    for _ in range(steps):
        rel = pos - center
        d = _length(rel)
        up = rel / d[..., None]
        density = atmosphere_density(p, d)
        light = _clamp(1.2 * _dot(np.broadcast_to(sun_dir, up.shape), up) + 0.5, 0.0, 1.0)
        light = light * light
        light_sum = light_sum + light * inv_steps
        factor = factor * (1.0 - density * step_len)
        pos = pos + stepv
    atmo = 1.0 - factor
    d0, d1 = np.asarray(p["u_day_color0"][:3], dtype=np.float64), np.asarray(p["u_day_color1"][:3], dtype=np.float64)
    n0, n1 = np.asarray(p["u_night_color0"][:3], dtype=np.float64), np.asarray(p["u_night_color1"][:3], dtype=np.float64)
    night = _mix(n0, n1, atmo[..., None])
    day = _mix(d0, d1, atmo[..., None])
    day_factor = _clamp(light_sum * p["u_day_night_transition_scale"], 0.0, 1.0)
    col = _mix(night, day, day_factor[..., None])
    return col, _clamp(atmo, 0.0, 1.0)


def _cloud_density(p, shape_tex, pos, bottom, top):
    """get_density_full, cloud_funcs.gdshaderinc:31-68, low quality (detail 0.5), coverage cubemap unset (= 1)."""
    height = _length(pos) - bottom
    height_ratio = height / (top - bottom)
    hc = np.maximum(1.0 - (2.0 * height_ratio - 1.0) ** 2, 0.0)
    coverage = 1.0 - 0.25 * height_ratio + p["u_cloud_coverage_bias"]
    shape = _mix(0.5, sample_shape(shape_tex, pos * p["u_cloud_shape_scale"]), p["u_cloud_shape_factor"])
    if p["u_cloud_shape_invert"] == 1.0:
        shape = 1.0 - shape
    density = (shape - 0.2 * 0.5 + _mix(-1.2, 1.5, coverage)) * hc
    return _clamp(density * 50.0 - 20.0, 0.0, 1.0), height_ratio


def _light_raymarched(p, shape_tex, pos0, sun_dir, bottom, top):
    """cloud_funcs.gdshaderinc:104-151"""
    steps = 6
    reach = (top - bottom) * 0.15
    h0 = (_length(pos0) - bottom) / (top - bottom)
    step_len = reach * (1.0 / steps)
    alpha = np.zeros(pos0.shape[:-1])
    for i in range(steps):
        pos = pos0 + float(i) * step_len * sun_dir
        density, _ = _cloud_density(p, shape_tex, pos, bottom, top)
        density = density * (step_len * p["u_cloud_density_scale"])
        tr = np.exp(-density)
        alpha = alpha + (1.0 - tr) * (1.0 - alpha)
        step_len = step_len * 1.2
    return _mix(1.0, h0 * 0.2, alpha)


def raymarch_cloud(p, shape_tex, steps, rm, origin, ray_dir, t_begin, t_end, jitter, sun_dir):
    """cloud_funcs.gdshaderinc:175-247"""
    R, H = p["u_planet_radius"], p["u_atmosphere_height"]
    bottom, top = R + p["u_cloud_bottom"] * H, R + p["u_cloud_top"] * H
    space = 0.5 * np.sqrt(1.0 - (R / top) ** 2) * bottom
    ground = 3.0 * space
    max_d = _mix(ground, space, _smoothstep(bottom, top * 1.05, np.sqrt((origin * origin).sum())))
    t_end = t_begin + np.minimum(t_end - t_begin, max_d)
    step_len = (t_end - t_begin) * (1.0 / float(steps))
    tt = np.ones(ray_dir.shape[:-1])
    total_light = np.zeros(ray_dir.shape[:-1])
    alpha = np.zeros(ray_dir.shape[:-1])
    pos = origin + (jitter * step_len)[..., None] * ray_dir + ray_dir * t_begin[..., None]
    for _ in range(steps):
        if rm:
            light = _light_raymarched(p, shape_tex, pos, sun_dir, bottom, top)
        else:  # get_light_cheap, :92-102
            hr = (_length(pos) - bottom) / (top - bottom)
            dp = _dot(ray_dir, np.broadcast_to(sun_dir, ray_dir.shape))
            p16 = np.where(dp > 0.0, np.abs(dp) ** 16, 0.0)
            light = hr + np.maximum(p16, 0.0) * (1.0 - alpha)
        # get_planet_shadow, :78-90
        shadow = _smoothstep(-0.3, 0.3, _dot(_normalize(pos), np.broadcast_to(-sun_dir, pos.shape)))
        light = light * _mix(1.0, 0.002, shadow)
        density, _ = _cloud_density(p, shape_tex, pos, bottom, top)
        density = density * p["u_cloud_density_scale"]
        tr = np.exp(-density * step_len)
        tt = np.maximum(tt * tr, 0.005)
        total_light = total_light + light * density * step_len * tt
        alpha = alpha + (1.0 - tr) * (1.0 - alpha)
        pos = pos + ray_dir * step_len[..., None]
    return total_light, alpha


def render(p, tex, cfg, frame, depth):
    """atmosphere_fragment for every pixel.  p: uniform dict, tex: dict(optical_depth, blue_noise, shape), cfg: oracle
    config dict, frame: dict from make_frame (column-major flat matrices).  Returns float64 (H, W, 4)."""
    w, h = frame["viewport_w"], frame["viewport_h"]
    inv_p = np.asarray(frame["inv_projection_matrix"], dtype=np.float64).reshape(4, 4).T
    inv_v = np.asarray(frame["inv_view_matrix"], dtype=np.float64).reshape(4, 4).T
    center = np.asarray(frame["planet_center_viewspace"], dtype=np.float64)
    sun_c = np.asarray(frame["sun_center_viewspace"], dtype=np.float64)
    px, py = np.meshgrid(np.arange(w), np.arange(h))
    uv = np.stack([(px + 0.5) / w, (py + 0.5) / h], axis=-1)
    ndc = np.concatenate([uv * 2.0 - 1.0, depth.astype(np.float64)[..., None], np.ones((h, w, 1))], axis=-1)
    view = ndc @ inv_p.T
    world = view @ inv_v.T
    pos_world = world[..., :3] / world[..., 3:4]
    cam = inv_v[:3, 3]
    linear_depth = _length(cam - pos_world)
    ray_dir = _normalize(view[..., :3])
    R, H = p["u_planet_radius"], p["u_atmosphere_height"]
    zero = np.zeros(3)
    ax, ay = ray_sphere(center, R + H, zero, ray_dir)
    hit = ax != ay
    t_begin = np.maximum(ax, 0.0)
    t_end = np.maximum(ay, 0.0)
    gx, gy = ray_sphere(center, R, zero, ray_dir)
    gd = np.where(gx != gy, gx, 10000000.0)
    linear_depth = _mix(linear_depth, gd, p["u_sphere_depth_factor"])
    t_end = np.minimum(t_end, linear_depth)
    sun_dir = (sun_c - center) / np.sqrt(((sun_c - center) ** 2).sum())
    jx = (w * uv[..., 0]).astype(int) & 0xFF
    jy = (h * uv[..., 1]).astype(int) & 0xFF
    jitter = tex["blue_noise"][jy, jx].astype(np.float64) / 255.0

    if cfg.get("lite"):
        rgb, alpha = compute_atmosphere_v1(p, cfg["view_steps"], ray_dir, center, t_begin, t_end, sun_dir)
    else:
        rgb, alpha = compute_atmosphere_v2(p, tex["optical_depth"], cfg["view_steps"], ray_dir, center, t_begin, t_end, sun_dir, jitter)

    if cfg.get("cloud_steps", 0) > 0:  # render_clouds, cloud_funcs.gdshaderinc:249-324
        bottom, top = R + p["u_cloud_bottom"] * H, R + p["u_cloud_top"] * H
        tx, ty = ray_sphere(center, top, zero, ray_dir)
        bx, by = ray_sphere(center, bottom, zero, ray_dir)
        c0 = np.maximum(tx, 0.0)
        c1 = np.minimum(ty, linear_depth)
        gate = (tx != ty) & (c0 < linear_depth) & ((linear_depth > by) | (bx > 0.0))
        m = np.asarray(p["u_world_to_model_matrix"], dtype=np.float64).reshape(4, 4).T @ inv_v
        origin_m = m[:3, 3]
        dir_m = ray_dir @ m[:3, :3].T
        sun_m = m[:3, :3] @ sun_dir
        cl, ca = raymarch_cloud(p, tex["shape"], cfg["cloud_steps"], bool(cfg.get("cloud_light_rm")), origin_m, dir_m,
                                np.where(gate, c0, 0.0), np.where(gate, c1, 0.0), jitter, sun_m)
        # blend_colors(self = atmosphere, over = cloud), util.gdshaderinc:61-69
        sa = 1.0 - ca
        a = alpha * sa + ca
        safe = np.where(a == 0.0, 1.0, a)
        ab_rgb = np.where((a == 0.0)[..., None], 0.0, (rgb * (alpha * sa)[..., None] + (cl * ca)[..., None]) / safe[..., None])
        ab_a = np.where(a == 0.0, 0.0, a)
        add_rgb = rgb + (cl * ca)[..., None]
        add_a = np.maximum(alpha, ca)
        k = p["u_cloud_blend"]
        rgb = np.where(gate[..., None], _mix(ab_rgb, add_rgb, k), rgb)
        alpha = np.where(gate, _mix(ab_a, add_a, k), alpha)

    out = np.concatenate([rgb, alpha[..., None]], axis=-1)
    out[~hit] = 0.0
    return out
