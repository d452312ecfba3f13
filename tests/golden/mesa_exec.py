#!/usr/bin/env python3
"""The reference's GDShader text compiled by Mesa's GLSL compiler and executed by llvmpipe -- a GLSL implementation this repository's
author did not write -- as a second pin of the oracle beside tests/golden/gdshader_vm.py.  BUILD CONTAINER ONLY (needs /root/reference and
the image's Mesa 23.2.1 swrast_dri.so); the vectors it produces are committed (tests/golden/make_mesa_vectors.py).

What is done to the text (nothing else; the functions' bodies reach Mesa's compiler as the reference wrote them):
  * `#include "x"` is resolved textually (GLSL has no #include); every other directive -- #define, #ifdef, the include guards -- is left
    to Mesa's preprocessor; a forced macro (another step count) is a `#define` line in place of the file's own;
  * a trailing comma in a parameter or argument list (GDShader accepts it) is removed; `shader_type` / `render_mode` lines are dropped; `uniform T name : hints = default;` becomes `uniform T name;` (the host uploads the
    default, sRGB -> linear for `source_color`, and applies the hints' sampler state); `varying T name;` becomes a global;
  * the engine's built-ins the text names (SCREEN_UV, INV_PROJECTION_MATRIX, ALBEDO, ...) are declared as uniforms / globals and a main()
    calls fragment() (or vertex()) -- the same values tests/golden/make_reference_vectors.py hands the interpreter.
Sampler state as DESIGN.md section 2 states it for the engine: optical-depth LUT R32F linear clamp; blue noise R8 nearest repeat (texelFetch);
shape volume R8 linear repeat; coverage cubemap R8, seamless, linear (level 0) or linear-mipmap-linear with the mip chain uploaded level by
level (2 x 2 box, as noise_cubemap.gd builds it); depth R32F nearest.
llvmpipe is run with GALLIVM_PERF=no_quad_lod,no_aos_sampling (per-pixel level of detail, float filter weights: its fast paths -- one level
of detail per quad, 8-bit weights -- are what a real GPU's texture unit does and are measured separately, see make_mesa_vectors.py)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SHADERS = "/root/reference/addons/zylann.atmosphere/shaders"
F32 = np.float32
os.environ.setdefault("GALLIVM_PERF", "no_quad_lod,no_aos_sampling")
os.environ.setdefault("LP_NUM_THREADS", "8")

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "_mesa", "libmesa_glsl_runner.so")
        src = os.path.join(HERE, "mesa_glsl_runner.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-ldl"], check=True)
        L = C.CDLL(so)
        L.mgl_error.restype = C.c_char_p
        L.mgl_info.restype = C.c_char_p
        L.mgl_program.restype = C.c_uint
        L.mgl_program.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.mgl_uniform.argtypes = [C.c_uint, C.c_char_p, C.c_int, C.c_void_p]
        L.mgl_texture.restype = C.c_uint
        L.mgl_texture.argtypes = [C.c_uint, C.c_char_p] + [C.c_int] * 8 + [C.c_void_p]
        L.mgl_draw.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.mgl_draw_over.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.mgl_delete_texture.argtypes = [C.c_uint]
        L.mgl_delete_program.argtypes = [C.c_uint]
        if L.mgl_init(None) != 0:
            raise RuntimeError("mesa runner: " + L.mgl_error().decode())
        _lib = L
    return _lib


def info():
    return lib().mgl_info().decode()


# ---------------------------------------------------------------------------------------------------- text -> GLSL
def flatten(path, forced):
    """The file with its #include lines replaced by the included files' text; a `#define NAME ...` of a forced macro is replaced by ours."""
    out = []
    with open(path, "r", encoding="utf-8") as fh:
        for line in fh.read().split("\n"):
            m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
            if m:
                out.append(flatten(os.path.join(os.path.dirname(path), m.group(1)), forced))
                continue
            m = re.match(r"\s*#\s*define\s+(\w+)", line)
            if m and m.group(1) in forced:
                out.append(f"#define {m.group(1)} {forced[m.group(1)]}")
                continue
            out.append(line)
    return "\n".join(out)


_UNIFORM = re.compile(r"^([ \t]*)uniform[ \t]+(\w+)[ \t]+(\w+)[ \t]*(?::[ \t]*([^=;\n]+?))?[ \t]*(?:=[ \t]*([^;\n]+?))?[ \t]*;", re.M)


def _default_value(ty, expr):
    """`0.2`, `false`, `vec3(1.0)`, `vec4(0.5, 0.8, 1.0, 1.0)` -> floats"""
    n = {"float": 1, "bool": 1, "int": 1, "vec2": 2, "vec3": 3, "vec4": 4}[ty]
    expr = expr.strip()
    if expr in ("true", "false"):
        return [1.0 if expr == "true" else 0.0]
    m = re.match(r"^(\w+)\((.*)\)$", expr)
    vals = [float(v) for v in (m.group(2).split(",") if m else [expr])]
    return vals * n if len(vals) == 1 else vals


# exp and pow, correctly rounded, written in GLSL with double arithmetic (GL 4.x fp64, which llvmpipe executes natively) and substituted for llvmpipe's own
# (18 ulp / 1e-6) by two #defines in front of the reference's text: the `exact` configuration of the Mesa pin.  GLSL leaves the accuracy of exp and
# pow to the implementation; this one is the best the type allows, which is also what the interpreter and (to an ulp) the oracle use.  Everything else --
# the compiler, the control flow, sqrt / division / normalize / mix / clamp / dot, the texture units -- stays Mesa's.
EXACT_BUILTINS = """
double mgl_exp_d(double x) {   // e^x, |error| < 1e-15lf relative for |x| < 700
    const double LN2_HI = 0.693147180369123816490lf, LN2_LO = 1.90821492927058770002e-10lf, INV_LN2 = 1.44269504088896338700lf;
    double n = round(x * INV_LN2);
    double r = (x - n * LN2_HI) - n * LN2_LO;          // |r| <= 0.3466lf
    double p = 1.0lf / 6227020800.0lf;                      // Taylor to r^13 (0.3466lf^14 / 14! = 4e-18lf)
    p = p * r + 1.0lf / 479001600.0lf;  p = p * r + 1.0lf / 39916800.0lf;  p = p * r + 1.0lf / 3628800.0lf;  p = p * r + 1.0lf / 362880.0lf;
    p = p * r + 1.0lf / 40320.0lf;  p = p * r + 1.0lf / 5040.0lf;  p = p * r + 1.0lf / 720.0lf;  p = p * r + 1.0lf / 120.0lf;
    p = p * r + 1.0lf / 24.0lf;  p = p * r + 1.0lf / 6.0lf;  p = p * r + 0.5lf;  p = p * r + 1.0lf;  p = p * r + 1.0lf;
    return ldexp(p, int(n));
}
double mgl_log_d(double x) {   // ln x for x > 0
    int e;
    double m = frexp(x, e);                             // x = m 2^e, m in [0.5lf, 1)
    if (m < 0.70710678118654752440lf) { m *= 2.0lf; e -= 1; }
    double s = (m - 1.0lf) / (m + 1.0lf), s2 = s * s;       // |s| <= 0.1716lf; ln m = 2 atanh s
    double p = 1.0lf / 27.0lf;
    p = p * s2 + 1.0lf / 25.0lf;  p = p * s2 + 1.0lf / 23.0lf;  p = p * s2 + 1.0lf / 21.0lf;  p = p * s2 + 1.0lf / 19.0lf;  p = p * s2 + 1.0lf / 17.0lf;
    p = p * s2 + 1.0lf / 15.0lf;  p = p * s2 + 1.0lf / 13.0lf;  p = p * s2 + 1.0lf / 11.0lf;  p = p * s2 + 1.0lf / 9.0lf;  p = p * s2 + 1.0lf / 7.0lf;
    p = p * s2 + 1.0lf / 5.0lf;  p = p * s2 + 1.0lf / 3.0lf;  p = p * s2 + 1.0lf;
    return 2.0lf * s * p + double(e) * 0.69314718055994530942lf;
}
float mgl_exp(float x) { return (x == x && abs(x) < 100.0) ? float(mgl_exp_d(double(x))) : exp(x); }
vec2 mgl_exp(vec2 x) { return vec2(mgl_exp(x.x), mgl_exp(x.y)); }
vec3 mgl_exp(vec3 x) { return vec3(mgl_exp(x.x), mgl_exp(x.y), mgl_exp(x.z)); }
vec4 mgl_exp(vec4 x) { return vec4(mgl_exp(x.x), mgl_exp(x.y), mgl_exp(x.z), mgl_exp(x.w)); }
float mgl_pow(float x, float y) { return (x > 0.0 && x < 3.0e38 && abs(y) < 1.0e4) ? float(mgl_exp_d(double(y) * mgl_log_d(double(x)))) : pow(x, y); }
#define exp mgl_exp
#define pow mgl_pow
"""


# BASELINE.json's headline configuration ("32 view x 8 light steps") has no shader file of its own in the reference: its light march is this repository's
# composition -- the LUT fetch of compute_atmosphere_v2 replaced by the quantity the LUT tabulates (optical_depth.gdshader:17-31
# with the chord of :56-65), evaluated at the 3-D sample position with N left-Riemann samples.  For Mesa that composition is these lines, spliced in front of
# the reference's compute_atmosphere_v2 (whose body stays untouched: a macro redirects its one call) and built from the reference's own ray_sphere and
# get_atmosphere_density:
DIRECT_LIGHT_GLUE = """
float mgl_marched_optical_depth(vec3 pos, vec3 dir, vec3 planet_center) {
    vec2 rs = ray_sphere(planet_center, u_planet_radius + u_atmosphere_height, pos, dir);
    float ray_len = rs.y - max(rs.x, 0.0);
    float step_len = ray_len / float(MGL_LIGHT_STEPS);
    float optical_depth = 0.0;
    for (int i = 0; i < MGL_LIGHT_STEPS; ++i) {
        vec3 p = pos + dir * step_len * float(i);
        float d = length(p - planet_center);
        float density = get_atmosphere_density(d);
        optical_depth += density * step_len * u_density;
    }
    return optical_depth;
}
#define get_baked_optical_depth(p, d, c, t) mgl_marched_optical_depth(p, d, c)
"""


def translate(shader_file, defines=None, force_defines=None, stage="fragment", exact=False, direct_light_steps=None):
    """-> (GLSL 4.50 fragment-shader source, {uniform: (type, [hints], default floats or None)})"""
    forced = {k: str(v) for k, v in (force_defines or {}).items()}
    text = flatten(shader_file, forced)
    text = re.sub(r"^[ \t]*(shader_type|render_mode)\b[^;\n]*;", "", text, flags=re.M)
    uniforms = {}

    def uni(m):
        ty, name, hints, default = m.group(2), m.group(3), m.group(4), m.group(5)
        uniforms[name] = (ty, [h.strip() for h in hints.split(",")] if hints else [],
                          _default_value(ty, default) if default is not None else None)
        return f"{m.group(1)}uniform {ty} {name};"

    text = _UNIFORM.sub(uni, text)
    text = re.sub(r"^([ \t]*)varying[ \t]+", r"\1", text, flags=re.M)
    text = re.sub(r",(\s*)\)", r"\1)", text)  # GDShader accepts a trailing comma in parameter and argument lists, GLSL does not
    if direct_light_steps:
        anchor = "vec4 compute_atmosphere_v2("
        assert text.count(anchor) == 1
        text = text.replace(anchor, f"#define MGL_LIGHT_STEPS {int(direct_light_steps)}\n" + DIRECT_LIGHT_GLUE + anchor)
    head = ["#version 450 core"]
    for k, v in list((defines or {}).items()) + [(k, v) for k, v in forced.items() if not re.search(r"#\s*define\s+" + k + r"\b", text)]:
        head.append(f"#define {k} {v}")
    builtins = """
uniform vec2 VIEWPORT_SIZE;
uniform mat4 INV_PROJECTION_MATRIX, INV_VIEW_MATRIX, PROJECTION_MATRIX, MODELVIEW_MATRIX, MODEL_MATRIX, VIEW_MATRIX;
uniform float TIME;
uniform vec3 MGL_planet_center_viewspace, MGL_sun_center_viewspace;
vec2 SCREEN_UV, UV;
vec3 ALBEDO, VERTEX;
float ALPHA;
vec4 POSITION, COLOR;
out vec4 MGL_out;
"""
    if stage == "fragment":
        main = """
void main() {
    SCREEN_UV = gl_FragCoord.xy / VIEWPORT_SIZE;
    v_planet_center_viewspace = MGL_planet_center_viewspace;
    v_sun_center_viewspace = MGL_sun_center_viewspace;
    ALBEDO = vec3(0.0);
    ALPHA = 0.0;
    fragment();
    MGL_out = vec4(ALBEDO, ALPHA);
}
"""
    elif stage == "vertex":  # the varyings of the vertex stage, pixel 0 and pixel 1 of a 2 x 1 target
        main = """
void main() {
    VERTEX = vec3(0.0);
    POSITION = vec4(0.0);
    vertex();
    MGL_out = gl_FragCoord.x < 1.0 ? vec4(v_planet_center_viewspace, 0.0) : vec4(v_sun_center_viewspace, 0.0);
}
"""
    else:  # canvas_item (the LUT bake): UV over the target, COLOR out
        main = """
void main() {
    UV = gl_FragCoord.xy / VIEWPORT_SIZE;
    COLOR = vec4(0.0);
    fragment();
    MGL_out = COLOR;
}
"""
    return "\n".join(head) + builtins + (EXACT_BUILTINS if exact else "") + text + main, uniforms


# ---------------------------------------------------------------------------------------------------- running
class Program:
    def __init__(self, source):
        L = lib()
        log = C.create_string_buffer(1 << 16)
        self.id = L.mgl_program(source.encode(), log, len(log))
        if not self.id:
            lines = source.split("\n")
            raise RuntimeError("Mesa rejected the shader:\n" + log.value.decode() + "\n" +
                               "\n".join(f"{i + 1:4d} {ln}" for i, ln in enumerate(lines) if re.search(rf"\b0:{i + 1}\(", log.value.decode())))
        self.textures = []

    def set(self, name, ty, value):
        kind = {"float": 1, "vec2": 2, "vec3": 3, "vec4": 4, "mat2": 22, "mat3": 9, "mat4": 16, "bool": 0, "int": 0}[ty]
        a = np.ascontiguousarray(np.asarray(value, dtype=F32).reshape(-1))
        return lib().mgl_uniform(self.id, name.encode(), kind, a.ctypes.data_as(C.c_void_p))

    def texture(self, sampler, unit, dims, shape, fmt, filt, repeat, data):
        data = np.ascontiguousarray(data)
        w, h, d = shape
        t = lib().mgl_texture(self.id, sampler.encode(), unit, dims, w, h, d, fmt, filt, int(repeat), data.ctypes.data_as(C.c_void_p))
        if not t:
            raise RuntimeError(f"texture {sampler} failed")
        self.textures.append(t)

    def draw(self, w, h, clear=-1.0, over=None):
        """over: (h, w, 4) float32 scene colours the draw is alpha-blended onto by the fixed-function blend stage (blend_mix); None: a cleared target"""
        out = np.empty((h, w, 4), dtype=F32)
        dst = None if over is None else np.ascontiguousarray(over, dtype=F32)
        if lib().mgl_draw_over(self.id, w, h, C.c_float(clear), None if dst is None else dst.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError("draw: " + lib().mgl_error().decode())
        return out

    def close(self):
        for t in self.textures:
            lib().mgl_delete_texture(t)
        lib().mgl_delete_program(self.id)


def _upload_uniforms(prog, uniforms, params, world_to_model, sun, source_color):
    from godot_atmosphere_shader_amd import scene as S

    for name, (ty, hints, default) in uniforms.items():
        if ty.startswith("sampler"):
            continue
        if name == "u_world_to_model_matrix":
            val = S.col_major(world_to_model)
        elif name == "u_sun_position":
            val = sun
        elif name in params:   # scene parameters are the values the shader sees (linear colours), as in make_reference_vectors.uniforms_for
            val = params[name]
        elif default is not None:
            val = np.asarray(default, dtype=np.float64).reshape(-1)
            if "source_color" in hints:  # the engine converts sRGB -> linear when it uploads a source_color value: the text's defaults are sRGB
                val = np.concatenate([np.asarray(source_color(val[:3])), val[3:]])
        else:
            continue
        prog.set(name, ty, np.asarray(val, dtype=np.float64).reshape(-1))


def run_frame(shader, defines, params, world_to_model, model_matrix, cam, depth, textures, time_s=0.0, sun=None, force_defines=None,
              cube_chain=None, exact=False, direct_light_steps=None, over=None):
    """As make_reference_vectors.run_frame: vertex() once, fragment() for every pixel.  textures: dict(lut (H, W) f32, blue (256, 256) u8,
    shape (n, n, n) u8 [z, y, x], cubemap (6, n, n) u8); cube_chain: [(6, n, n), (6, n/2, n/2), ...] -> the declared linear-mipmap sampler.
    Returns rgba (H, W, 4) with discarded fragments zeroed, the discard mask, the two varyings."""
    from godot_atmosphere_shader_amd import scene as S

    sun = S.DEMO_SUN_POSITION if sun is None else sun
    path = os.path.join(SHADERS, shader + ".gdshader")
    # vertex stage
    vsrc, uniforms = translate(path, defines, force_defines, stage="vertex", exact=exact, direct_light_steps=direct_light_steps)
    vp = Program(vsrc)
    _upload_uniforms(vp, uniforms, params, world_to_model, sun, S.srgb_to_linear)
    ident = np.eye(4, dtype=F32).reshape(-1)
    vp.set("PROJECTION_MATRIX", "mat4", ident)
    vp.set("MODELVIEW_MATRIX", "mat4", ident)
    vp.set("MODEL_MATRIX", "mat4", S.col_major(model_matrix))
    vp.set("VIEW_MATRIX", "mat4", S.col_major(cam.view))
    vp.set("TIME", "float", [time_s])
    v = vp.draw(2, 1)
    vp.close()
    varyings = (v[0, 0, :3].copy(), v[0, 1, :3].copy())
    # fragment stage
    fsrc, uniforms = translate(path, defines, force_defines, stage="fragment", exact=exact, direct_light_steps=direct_light_steps)
    fp = Program(fsrc)
    _upload_uniforms(fp, uniforms, params, world_to_model, sun, S.srgb_to_linear)
    fp.set("VIEWPORT_SIZE", "vec2", [cam.width, cam.height])
    fp.set("INV_PROJECTION_MATRIX", "mat4", S.col_major(cam.inv_projection))
    fp.set("INV_VIEW_MATRIX", "mat4", S.col_major(cam.inv_view))
    fp.set("TIME", "float", [time_s])
    fp.set("MGL_planet_center_viewspace", "vec3", varyings[0])
    fp.set("MGL_sun_center_viewspace", "vec3", varyings[1])
    unit = 0
    for name, (ty, hints, _) in uniforms.items():
        if not ty.startswith("sampler"):
            continue
        if fp.set(name, "int", [unit]) == -1:   # not active in this variant (declared by an include whose functions the variant does not call)
            continue
        if name == "u_depth_texture":
            d = np.ascontiguousarray(depth, dtype=F32)
            fp.texture(name, unit, 2, (d.shape[1], d.shape[0], 1), 2, 0, False, d)
        elif name == "u_optical_depth_texture":
            t = np.ascontiguousarray(textures["lut"], dtype=F32)
            fp.texture(name, unit, 2, (t.shape[1], t.shape[0], 1), 2, 1, "repeat_enable" in hints, t)
        elif name == "u_blue_noise_texture":
            t = np.ascontiguousarray(textures["blue"], dtype=np.uint8)
            fp.texture(name, unit, 2, (t.shape[1], t.shape[0], 1), 0, 0 if "filter_nearest" in hints else 1, "repeat_enable" in hints, t)
        elif name == "u_cloud_shape_texture":
            t = np.ascontiguousarray(textures["shape"], dtype=np.uint8)
            fp.texture(name, unit, 3, (t.shape[2], t.shape[1], t.shape[0]), 0, 1, "repeat_enable" in hints, t)
        elif name == "u_cloud_coverage_cubemap":
            if cube_chain is not None:
                data = np.concatenate([np.ascontiguousarray(lv, dtype=np.uint8).reshape(-1) for lv in cube_chain])
                fp.texture(name, unit, 6, (cube_chain[0].shape[1], cube_chain[0].shape[1], len(cube_chain)), 0, 2, False, data)
            else:
                t = np.ascontiguousarray(textures["cubemap"], dtype=np.uint8)
                fp.texture(name, unit, 6, (t.shape[1], t.shape[1], 1), 0, 1, False, t)
        else:
            raise RuntimeError(f"no texture for sampler {name}")
        unit += 1
    if over is not None:   # the draw with the renderer's blend stage: the frame is what the colour buffer holds afterwards
        out = fp.draw(cam.width, cam.height, over=over)
        fp.close()
        return out, None, varyings
    out = fp.draw(cam.width, cam.height, clear=-1.0)
    fp.close()
    disc = np.all(out == -1.0, axis=-1)
    out[disc] = 0.0
    return out, disc, varyings


def run_bake(params, n=256, exact=False):
    """optical_depth.gdshader over the n x n target, then what the baker does with it (RGBA8 viewport -> bytes reinterpreted as R32F)."""
    src, uniforms = translate(os.path.join(SHADERS, "optical_depth.gdshader"), stage="canvas", exact=exact)
    p = Program(src)
    for k in ("u_planet_radius", "u_atmosphere_height", "u_density"):
        p.set(k, "float", [params[k]])
    p.set("VIEWPORT_SIZE", "vec2", [n, n])
    color = p.draw(n, n, clear=0.0)
    p.close()
    b = np.rint(color.astype(np.float64) * 255.0).astype(np.uint32)
    bits = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | (b[..., 3] << 24)
    return bits.astype(np.uint32).view(F32).reshape(n, n), color
