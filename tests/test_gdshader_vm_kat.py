"""Known-answer tests for the GDShader interpreter (tests/golden/gdshader_vm.py), written from the GLSL ES 3.00 specification -- NOT from the
reference's shaders and not from the oracle (the interpreter is what pins the oracle to the reference text, and it shares
an author with the oracle; these cases pin the interpreter to the language).

Every case is a few lines of shader text written for the purpose and the values the specification prescribes for them, worked out by hand
(section numbers refer to "The OpenGL ES Shading Language 3.00", the language Godot's shader dialect is defined against):
  5.1  operator precedence and associativity            5.4  constructors (scalar conversion, vector, matrix: column-major)
  5.5  vector components and swizzles (also as l-values) 5.9  expressions (component-wise operators, scalar-vector, comparison)
  5.10/5.11 vector and matrix operations                  6.1  function calls: in / out / inout are copy-in / copy-out
  6.3 / 6.4 selection, iteration, return, discard         8.1-8.5 built-in functions
Where the specification leaves a result undefined or to the implementation (pow of a non-positive base, the rounding of a built-in), the case
states the interpreter's documented convention and says so.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gdshader_vm as VM  # noqa: E402

F = np.float32


def run(tmp_path, body, lanes=1, inputs=None, outs=("OUT",), uniforms=None, pre="", out_types=None, samplers=None):
    """Executes `pre` + void main() { body } on `lanes` lanes; inputs: name -> (type, values); returns the named outputs (lane axis last)."""
    path = tmp_path / "t.gdshader"
    path.write_text(pre + "\nvoid main() {\n" + body + "\n}\n")
    m = VM.Machine(VM.load(str(path)), lanes, samplers or {}, uniforms or {})
    for k, (ty, val) in (inputs or {}).items():
        m.globals[k] = m.from_host(ty, val)
    for k in outs:
        ty = (out_types or {}).get(k, "vec4")
        n = {"float": None, "vec2": 2, "vec3": 3, "vec4": 4}[ty]
        m.globals[k] = m.from_host(ty, np.zeros(lanes) if n is None else np.zeros((n, lanes)))
    m.run("main")
    res = [np.asarray(m.globals[k].a) for k in outs]
    return res[0] if len(res) == 1 else res


def col(x):
    return np.asarray(x, dtype=F).reshape(-1, 1)


# ---- 5.1 precedence and associativity -----------------------------------------------------------------------------------------------------
def test_multiplicative_binds_tighter_than_additive(tmp_path):
    out = run(tmp_path, "OUT = vec4(1.0 + 2.0 * 3.0, (1.0 + 2.0) * 3.0, 10.0 - 4.0 / 2.0, 2.0 * 3.0 + 4.0 * 5.0);")
    assert out[:, 0].tolist() == [7.0, 9.0, 8.0, 26.0]


def test_subtraction_and_division_associate_left_to_right(tmp_path):
    out = run(tmp_path, "OUT = vec4(8.0 - 3.0 - 2.0, 8.0 / 4.0 / 2.0, 8.0 - (3.0 - 2.0), 8.0 / (4.0 / 2.0));")
    assert out[:, 0].tolist() == [3.0, 1.0, 7.0, 4.0]


def test_unary_minus_binds_tighter_than_multiplication(tmp_path):
    out = run(tmp_path, "float a = 2.0; OUT = vec4(-a * 3.0, 3.0 * -a, -(a * 3.0) + 1.0, - -a);")
    assert out[:, 0].tolist() == [-6.0, -6.0, -5.0, 2.0]


def test_relational_below_additive_and_logical_and_above_or(tmp_path):
    body = """
    bool p = 1.0 + 2.0 < 4.0;            // (1 + 2) < 4
    bool q = true || false && false;     // true || (false && false)
    bool r = !(2.0 >= 2.0) || 3.0 != 3.0;
    bool s = 1.0 < 2.0 == true;          // relational above equality: (1 < 2) == true
    OUT = vec4(p ? 1.0 : 0.0, q ? 1.0 : 0.0, r ? 1.0 : 0.0, s ? 1.0 : 0.0);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [1.0, 1.0, 0.0, 1.0]


def test_conditional_operator_associates_right_to_left(tmp_path):
    body = "OUT = vec4(IN.x > 0.0 ? 1.0 : IN.x < 0.0 ? 2.0 : 3.0, 0.0, 0.0, 0.0);"
    out = run(tmp_path, body, lanes=3, inputs={"IN": ("vec2", np.array([[5.0, -5.0, 0.0], [0, 0, 0]], dtype=F))})
    assert out[0].tolist() == [1.0, 2.0, 3.0]


def test_conditional_operator_selects_per_lane_and_binds_loosest(tmp_path):
    body = "float k = IN.x < 1.0 ? IN.x + 10.0 : IN.x * 2.0; OUT = vec4(k);"   # ?: below + and *: whole sums are the arms
    out = run(tmp_path, body, lanes=2, inputs={"IN": ("vec2", np.array([[0.5, 4.0], [0, 0]], dtype=F))})
    assert out[0].tolist() == [10.5, 8.0]


def test_integer_arithmetic_division_and_bit_operators(tmp_path):
    body = """
    int a = 7; int b = 2;
    int sh = (1 << 4) | 3;               // shift above bitwise-or
    int m = 260 & 0xff;                  // the jitter index of the reference: & 0xff
    OUT = vec4(float(a / b), float(a - a / b * b), float(sh), float(m));
    """
    assert run(tmp_path, body)[:, 0].tolist() == [3.0, 1.0, 19.0, 4.0]


def test_shift_and_mask_of_unsigned_values(tmp_path):
    body = """
    uint u = floatBitsToUint(1.0);       // 0x3f800000
    uint e = (u >> 23u) & 255u;
    uint lo = floatBitsToUint(-2.0) >> 31u;
    OUT = vec4(float(e), float(lo), float(floatBitsToUint(0.5) >> 23u), float((floatBitsToUint(3.0) >> 22u) & 1u));
    """
    assert run(tmp_path, body)[:, 0].tolist() == [127.0, 1.0, 126.0, 1.0]   # 3.0 = 1.1b x 2^1: top mantissa bit set


# ---- 5.4 constructors -----------------------------------------------------------------------------------------------------------------------
def test_float_to_int_conversion_drops_the_fraction(tmp_path):
    body = "OUT = vec4(float(int(2.9)), float(int(-2.9)), float(int(0.999)), float(int(255.0 * 0.5)));"
    assert run(tmp_path, body)[:, 0].tolist() == [2.0, -2.0, 0.0, 127.0]


def test_int_to_float_and_bool_free_mixed_expressions(tmp_path):
    body = "int i = 3; float f = float(i) * 0.5 + float(i / 2); OUT = vec4(f, float(i * 2 + 1), float(-i), float(i) / 2.0);"
    assert run(tmp_path, body)[:, 0].tolist() == [2.5, 7.0, -3.0, 1.5]


def test_vector_constructors_concatenate_and_splat(tmp_path):
    body = """
    vec2 a = vec2(1.0, 2.0);
    vec3 b = vec3(a, 3.0);
    vec3 c = vec3(0.5, a);
    vec4 d = vec4(b, 4.0);
    vec4 e = vec4(7.0);
    OUT = vec4(b.z + c.x, c.y + c.z, d.w + d.x, e.y + e.w);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [3.5, 3.0, 5.0, 14.0]


def test_vector_constructor_from_a_longer_vector_drops_components(tmp_path):
    body = "vec4 v = vec4(1.0, 2.0, 3.0, 4.0); vec3 a = vec3(v); vec2 b = vec2(a); OUT = vec4(a.z, b.y, b.x, float(ivec2(vec2(-2.7, 260.9)).y));"
    assert run(tmp_path, body)[:, 0].tolist() == [3.0, 2.0, 1.0, 260.0]


def test_matrix_constructors_are_column_major(tmp_path):
    body = """
    mat2 m = mat2(1.0, 2.0, 3.0, 4.0);           // columns (1, 2) and (3, 4)
    mat2 n = mat2(vec2(5.0, 6.0), vec2(7.0, 8.0));
    OUT = vec4(m[0].y, m[1].x, n[1][0], n[0][1]);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [2.0, 3.0, 7.0, 6.0]


def test_mat4_constructor_from_sixteen_scalars_and_column_access(tmp_path):
    vals = ", ".join(f"{k}.0" for k in range(16))
    body = f"mat4 m = mat4({vals}); vec4 c2 = m[2]; OUT = vec4(c2.x, c2.w, m[3][1], m[0][3]);"
    assert run(tmp_path, body)[:, 0].tolist() == [8.0, 11.0, 13.0, 3.0]


# ---- 5.5 components and swizzles --------------------------------------------------------------------------------------------------------------
def test_swizzle_reads_reorder_and_repeat(tmp_path):
    body = "vec4 v = vec4(1.0, 2.0, 3.0, 4.0); vec3 a = v.zyx; vec2 b = v.ww; OUT = vec4(a.x, a.z, b.x + b.y, v.xyz.y);"
    assert run(tmp_path, body)[:, 0].tolist() == [3.0, 1.0, 8.0, 2.0]


def test_the_three_component_name_sets_are_aliases(tmp_path):
    body = "vec4 v = vec4(1.0, 2.0, 3.0, 4.0); OUT = vec4(v.r + v.x, v.g + v.t, v.b + v.p, v.a + v.q);"
    assert run(tmp_path, body)[:, 0].tolist() == [2.0, 4.0, 6.0, 8.0]


def test_swizzle_as_an_l_value_writes_the_named_components_only(tmp_path):
    body = """
    vec4 v = vec4(1.0, 2.0, 3.0, 4.0);
    v.yx = vec2(10.0, 20.0);     // y = 10, x = 20
    v.w += 0.5;
    v.z = v.x - v.y;
    OUT = v;
    """
    assert run(tmp_path, body)[:, 0].tolist() == [20.0, 10.0, 10.0, 4.5]


def test_swapping_through_a_swizzle_reads_before_it_writes(tmp_path):
    body = "vec4 v = vec4(1.0, 2.0, 3.0, 4.0); v.xy = v.yx; v.zw = v.wz; OUT = v;"
    assert run(tmp_path, body)[:, 0].tolist() == [2.0, 1.0, 4.0, 3.0]


# ---- 5.9 - 5.11 operators on vectors and matrices -----------------------------------------------------------------------------------------------
def test_arithmetic_operators_are_component_wise_and_broadcast_scalars(tmp_path):
    body = """
    vec3 a = vec3(1.0, 2.0, 3.0); vec3 b = vec3(4.0, 5.0, 6.0);
    vec3 p = a * b;          // NOT a dot product
    vec3 q = 2.0 * a - b / 2.0;
    vec3 r = a + 1.0;
    OUT = vec4(p.z, q.x, q.z, r.y);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [18.0, 0.0, 3.0, 3.0]


def test_compound_assignment_operators(tmp_path):
    body = "vec2 v = vec2(3.0, 4.0); v *= 2.0; v -= vec2(1.0, 2.0); v /= 5.0; float s = 1.0; s += v.x; s *= v.y; OUT = vec4(v, s, 0.0);"
    out = run(tmp_path, body)[:, 0]
    assert out.tolist() == [1.0, F(6.0) / F(5.0), F(2.0) * (F(6.0) / F(5.0)), 0.0]


def test_matrix_times_vector_is_a_linear_combination_of_columns(tmp_path):
    body = "mat2 m = mat2(1.0, 2.0, 3.0, 4.0); vec2 a = m * vec2(1.0, 1.0); vec2 b = m * vec2(2.0, -1.0); OUT = vec4(a, b);"
    assert run(tmp_path, body)[:, 0].tolist() == [4.0, 6.0, -1.0, 0.0]      # col0 * x + col1 * y


def test_vector_times_matrix_dots_the_vector_with_each_column(tmp_path):
    body = "mat2 m = mat2(1.0, 2.0, 3.0, 4.0); vec2 a = vec2(1.0, 1.0) * m; vec2 b = vec2(2.0, -1.0) * m; OUT = vec4(a, b);"
    assert run(tmp_path, body)[:, 0].tolist() == [3.0, 7.0, 0.0, 2.0]


def test_matrix_product_applies_the_right_factor_first(tmp_path):
    body = """
    mat2 a = mat2(1.0, 2.0, 3.0, 4.0); mat2 b = mat2(0.0, 1.0, 1.0, 0.0);   // b swaps the components
    mat2 c = a * b;                                                         // (a * b)[j] = a * b[j]: the columns of a, swapped
    vec2 v = c * vec2(1.0, 0.0);
    vec2 w = (a * b) * vec2(5.0, 7.0);
    vec2 u = a * (b * vec2(5.0, 7.0));
    OUT = vec4(v, w - u);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [3.0, 4.0, 0.0, 0.0]


def test_mat4_times_vec4_carries_the_translation_column(tmp_path):
    mat = np.eye(4, dtype=F)
    mat[:3, 3] = [10.0, 20.0, 30.0]                                         # translation
    uni = {"M": np.ascontiguousarray(mat.T).reshape(-1)}                     # column-major memory order
    body = "vec4 p = M * vec4(1.0, 2.0, 3.0, 1.0); vec4 d = M * vec4(1.0, 2.0, 3.0, 0.0); OUT = vec4(p.xy, d.z, (M * vec4(0.0, 0.0, 0.0, 1.0)).z);"
    assert run(tmp_path, body, uniforms=uni, pre="uniform mat4 M;")[:, 0].tolist() == [11.0, 22.0, 3.0, 30.0]


def test_mat4_times_mat4_composes_transforms(tmp_path):
    t = np.eye(4, dtype=F); t[:3, 3] = [1.0, 2.0, 3.0]
    s = np.diag(np.array([2.0, 2.0, 2.0, 1.0], dtype=F))
    uni = {"T": np.ascontiguousarray(t.T).reshape(-1), "S": np.ascontiguousarray(s.T).reshape(-1)}
    body = "vec4 a = (T * S) * vec4(1.0, 1.0, 1.0, 1.0); vec4 b = (S * T) * vec4(1.0, 1.0, 1.0, 1.0); OUT = vec4(a.x, a.z, b.x, b.z);"
    # T * S: scale first, then translate = (3, 4, 5);  S * T: translate first, then scale = (4, 6, 8)
    assert run(tmp_path, body, uniforms=uni, pre="uniform mat4 T; uniform mat4 S;")[:, 0].tolist() == [3.0, 5.0, 4.0, 8.0]


def test_sums_of_products_round_every_operation_in_binary32(tmp_path):
    body = """
    float big = 16777216.0;                       // 2^24: big + 1 is not representable
    float a = (big + 1.0) - big;
    float b = 0.1 + 0.2;
    float c = 1.0 / 3.0;
    float d = 1.0 + 1e-8;
    OUT = vec4(a, b, c, d);
    """
    out = run(tmp_path, body)[:, 0]
    assert out[0] == 0.0 and out[1] == F(0.1) + F(0.2) and out[1] != np.float64(0.1) + np.float64(0.2)
    assert out[2] == F(1.0) / F(3.0) and out[3] == 1.0


# ---- 6.1 functions: copy-in / copy-out ------------------------------------------------------------------------------------------------------------
def test_out_and_inout_parameters_are_copied_back_at_return(tmp_path):
    pre = """
    void f(inout float a, out float b) { a += 1.0; b = a * 2.0; }
    float g(float a, out float b) { b = 1.0; b += a; return a * 10.0; }   // `a` was copied in before `b` is written
    """
    body = "float x = 3.0; float y = 0.0; f(x, y); float z = 5.0; float r = g(z, z); OUT = vec4(x, y, z, r);"
    assert run(tmp_path, body, pre=pre)[:, 0].tolist() == [4.0, 8.0, 6.0, 50.0]


def test_in_parameters_are_copies_the_caller_keeps_its_value(tmp_path):
    pre = "float twice(float a) { a = a * 2.0; return a; } vec2 bump(vec2 v) { v.x += 1.0; return v; }"
    body = "float x = 3.0; float y = twice(x); vec2 p = vec2(1.0, 2.0); vec2 q = bump(p); OUT = vec4(x, y, p.x, q.x);"
    assert run(tmp_path, body, pre=pre)[:, 0].tolist() == [3.0, 6.0, 1.0, 2.0]


def test_out_parameter_through_a_swizzle_and_a_struct_member(tmp_path):
    pre = "struct S { float a; vec2 b; }; void set2(out vec2 o) { o = vec2(7.0, 8.0); } void inc(inout float o) { o += 1.0; }"
    body = "vec4 v = vec4(0.0); set2(v.zw); S s; s.a = 1.0; s.b = vec2(2.0, 3.0); inc(s.a); set2(s.b); OUT = vec4(v.z, v.w, s.a, s.b.y);"
    assert run(tmp_path, body, pre=pre)[:, 0].tolist() == [7.0, 8.0, 2.0, 8.0]


def test_structs_are_passed_and_returned_by_value(tmp_path):
    pre = """
    struct Hit { float t; vec3 n; };
    Hit make(float t) { Hit h; h.t = t; h.n = vec3(0.0, t, 0.0); return h; }
    float use(Hit h) { h.t = 100.0; return h.n.y; }
    """
    body = "Hit h = make(2.5); float y = use(h); OUT = vec4(h.t, y, h.n.y, 0.0);"
    assert run(tmp_path, body, pre=pre)[:, 0].tolist() == [2.5, 2.5, 2.5, 0.0]


def test_nested_calls_evaluate_arguments_before_the_call(tmp_path):
    pre = "float sq(float x) { return x * x; } float add(float a, float b) { return a + b; }"
    body = "OUT = vec4(add(sq(2.0), sq(3.0)), sq(add(1.0, 2.0)), add(sq(sq(2.0)), 1.0), sq(-3.0));"
    assert run(tmp_path, body, pre=pre)[:, 0].tolist() == [13.0, 9.0, 17.0, 9.0]


# ---- 6.3 / 6.4 control flow under divergence --------------------------------------------------------------------------------------------------------
def test_if_else_chains_take_one_branch_per_lane(tmp_path):
    body = """
    float r;
    if (IN.x < 1.0) { r = 10.0; } else if (IN.x < 2.0) { r = 20.0; } else { r = 30.0; }
    float s = 0.0;
    if (IN.x > 1.5) s = 1.0;                  // a branch without braces
    OUT = vec4(r, s, 0.0, 0.0);
    """
    out = run(tmp_path, body, lanes=3, inputs={"IN": ("vec2", np.array([[0.5, 1.5, 2.5], [0, 0, 0]], dtype=F))})
    assert out[0].tolist() == [10.0, 20.0, 30.0] and out[1].tolist() == [0.0, 0.0, 1.0]


def test_for_loops_run_per_lane_trip_counts_through_a_uniform_bound_and_a_lane_mask(tmp_path):
    body = """
    float acc = 0.0;
    for (int i = 0; i < 5; ++i) {
        if (float(i) < IN.x) { acc += float(i); }
    }
    int n = 0;
    for (int j = 10; j > 4; j -= 2) { n += 1; }      // 10, 8, 6
    OUT = vec4(acc, float(n), 0.0, 0.0);
    """
    out = run(tmp_path, body, lanes=3, inputs={"IN": ("vec2", np.array([[0.0, 2.5, 9.0], [0, 0, 0]], dtype=F))})
    assert out[0].tolist() == [0.0, 3.0, 10.0] and np.broadcast_to(out[1], (3,)).tolist() == [3.0, 3.0, 3.0]   # 0 + 1 + 2 below 2.5


def test_return_inside_a_loop_ends_the_function_for_that_lane_only(tmp_path):
    pre = """
    float first_above(float limit) {
        for (int i = 0; i < 8; ++i) {
            float v = float(i) * 1.5;
            if (v > limit) { return v; }
        }
        return -1.0;
    }
    """
    out = run(tmp_path, "OUT = vec4(first_above(IN.x));", lanes=4, pre=pre,
              inputs={"IN": ("vec2", np.array([[-1.0, 2.0, 4.5, 99.0], [0, 0, 0, 0]], dtype=F))})
    assert out[0].tolist() == [0.0, 3.0, 6.0, -1.0]


def test_statements_after_a_divergent_return_do_not_touch_returned_lanes(tmp_path):
    pre = """
    void shade(float x, out float a, out float b) {
        a = 1.0; b = 1.0;
        if (x < 0.0) { a = 2.0; return; }
        b = 3.0;
        if (x > 1.0) { return; }
        a = 4.0;
    }
    """
    out = run(tmp_path, "float a; float b; shade(IN.x, a, b); OUT = vec4(a, b, 0.0, 0.0);", lanes=3, pre=pre,
              inputs={"IN": ("vec2", np.array([[-1.0, 0.5, 2.0], [0, 0, 0]], dtype=F))})
    assert out[0].tolist() == [2.0, 4.0, 1.0] and out[1].tolist() == [1.0, 3.0, 3.0]


def test_loop_variables_and_block_scopes_shadow_outer_names(tmp_path):
    body = """
    float x = 1.0;
    { float x = 5.0; x += 1.0; }                   // an inner block's x
    float sum = 0.0;
    for (int i = 0; i < 3; ++i) { float x = float(i); sum += x; }
    OUT = vec4(x, sum, 0.0, 0.0);
    """
    assert run(tmp_path, body)[:2, 0].tolist() == [1.0, 3.0]


# ---- 8 built-in functions -----------------------------------------------------------------------------------------------------------------------------
def test_clamp_min_max_abs_on_scalars_and_vectors(tmp_path):
    body = """
    vec3 v = clamp(vec3(-1.0, 0.5, 7.0), 0.0, 1.0);
    vec2 lohi = clamp(vec2(5.0, 5.0), vec2(0.0, 6.0), vec2(4.0, 9.0));
    OUT = vec4(v.x + v.y + v.z, lohi.x * 10.0 + lohi.y, max(-3.0, -4.0) + min(2.0, 8.0), abs(-2.5) + abs(vec2(-1.0, 1.0)).x);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [1.5, 46.0, -1.0, 3.5]


def test_floor_and_fract_of_negative_numbers(tmp_path):
    body = "OUT = vec4(floor(-0.25), fract(-0.25), floor(2.75) + fract(2.75), fract(vec2(1.5, -1.5)).y);"   # fract = x - floor(x)
    assert run(tmp_path, body)[:, 0].tolist() == [-1.0, 0.75, 2.75, 0.5]


def test_mix_is_exact_at_both_ends_and_linear_in_between(tmp_path):
    body = "OUT = vec4(mix(2.0, 10.0, 0.0), mix(2.0, 10.0, 1.0), mix(2.0, 10.0, 0.25), mix(vec2(0.0, 4.0), vec2(8.0, 0.0), 0.5).y);"
    assert run(tmp_path, body)[:, 0].tolist() == [2.0, 10.0, 4.0, 2.0]


def test_smoothstep_saturates_outside_the_edges_and_is_hermite_inside(tmp_path):
    body = "OUT = vec4(smoothstep(1.0, 3.0, 0.0), smoothstep(1.0, 3.0, 5.0), smoothstep(1.0, 3.0, 2.0), smoothstep(0.0, 1.0, 0.25));"
    out = run(tmp_path, body)[:, 0]
    assert out[:3].tolist() == [0.0, 1.0, 0.5] and out[3] == F(0.25) * F(0.25) * (F(3.0) - F(2.0) * F(0.25))   # t t (3 - 2 t)


def test_geometric_built_ins(tmp_path):
    body = """
    vec3 a = vec3(3.0, 4.0, 12.0);
    vec3 n = normalize(vec3(0.0, 0.0, -5.0));
    OUT = vec4(length(a), distance(vec2(1.0, 1.0), vec2(4.0, 5.0)), dot(a, vec3(1.0, 2.0, 0.5)), n.z);
    """
    assert run(tmp_path, body)[:, 0].tolist() == [13.0, 5.0, 17.0, -1.0]


def test_sqrt_exp_and_pow(tmp_path):
    body = "OUT = vec4(sqrt(2.25), exp(0.0), pow(2.0, 10.0), pow(9.0, 0.5));"
    assert run(tmp_path, body)[:, 0].tolist() == [1.5, 1.0, 1024.0, 3.0]


def test_exp_and_pow_are_correctly_rounded_to_binary32(tmp_path):
    out = run(tmp_path, "OUT = vec4(exp(1.0), exp(-2.5), pow(1.5, 16.0), pow(0.3, 4.0));")[:, 0]
    # the specification bounds these to a few ulp; the interpreter's convention: evaluated in binary64, rounded once
    want = [F(np.exp(np.float64(1.0))), F(np.exp(np.float64(-2.5))), F(np.float64(1.5) ** 16), F(np.float64(F(0.3)) ** 4)]
    assert out.tolist() == [float(w) for w in want]


def test_pow_of_zero_and_of_a_negative_base_follow_the_stated_convention(tmp_path):
    # 8.2: pow(x, y) is undefined for x < 0, and for x = 0 with y <= 0.  The interpreter returns 0 for a non-positive base with a positive
    # exponent ("pow(dp, 16) of a non-positive base = 0"; the reference guards the call with clamp(dp, 0, 1) anyway)
    out = run(tmp_path, "OUT = vec4(pow(0.0, 2.0), pow(0.0, 16.0), pow(1.0, 0.0), pow(4.0, -1.0));")[:, 0]
    assert out.tolist() == [0.0, 0.0, 1.0, 0.25]


def test_built_ins_apply_component_wise_to_vectors(tmp_path):
    body = "vec3 v = sqrt(vec3(4.0, 9.0, 16.0)); vec2 m = max(vec2(1.0, 5.0), vec2(3.0, 2.0)); vec2 f = floor(vec2(1.5, -1.5)); OUT = vec4(v.y, m.x + m.y, f.x, f.y);"
    assert run(tmp_path, body)[:, 0].tolist() == [3.0, 8.0, 1.0, -2.0]


def test_min_max_with_mixed_scalar_bounds_and_clamp_order(tmp_path):
    body = "vec3 v = max(vec3(-1.0, 2.0, 0.5), 0.0); vec3 w = min(vec3(-1.0, 2.0, 0.5), 1.0); OUT = vec4(v.x + v.y, w.x + w.y, clamp(5.0, 0.0, 0.99), clamp(-5.0, 0.0, 0.99));"
    out = run(tmp_path, body)[:, 0]
    assert out.tolist() == [2.0, 0.0, float(F(0.99)), 0.0]


# ---- discard, uniforms, constants, the texture call ---------------------------------------------------------------------------------------------------
def test_discard_marks_the_lane_and_the_others_run_on(tmp_path):
    # 6.4: a discarded fragment updates no buffer -- what it computes afterwards is unobservable; the interpreter records the lane in
    # `discarded` (the caller stores nothing for it, and texture units do not count it as reaching later calls) and lets the others finish
    body = "OUT = vec4(1.0); if (IN.x < 0.0) { discard; } OUT = vec4(IN.x + 1.0);"
    path = tmp_path / "d.gdshader"
    path.write_text("void main() {\n" + body + "\n}\n")
    m = VM.Machine(VM.load(str(path)), 3, {}, {})
    m.globals["IN"] = m.from_host("vec2", np.array([[-1.0, 1.0, -2.0], [0, 0, 0]], dtype=F))
    m.globals["OUT"] = m.from_host("vec4", np.zeros((4, 3)))
    m.run("main")
    assert np.asarray(m.discarded).tolist() == [True, False, True]
    assert np.broadcast_to(np.asarray(m.globals["OUT"].a)[0], (3,))[1] == 2.0


def test_uniforms_defaults_and_global_constants(tmp_path):
    pre = "uniform float u_k = 2.5; uniform vec3 u_v = vec3(1.0, 2.0, 3.0); uniform float u_set = 1.0; const float HALF = 0.5; const vec2 C = vec2(3.0, 4.0);"
    out = run(tmp_path, "OUT = vec4(u_k * HALF, u_v.z, u_set, length(C));", pre=pre, uniforms={"u_set": np.array([9.0], dtype=F)})[:, 0]
    assert out.tolist() == [1.25, 3.0, 9.0, 5.0]


def test_texture_calls_hand_every_lanes_coordinates_to_the_unit(tmp_path):
    class Unit:
        def __init__(self):
            self.seen = None

        def texture(self, uv):   # a single-channel unit (every texture of the path is R8 / R32F): .r per lane; the call yields (r, 0, 0, 1)
            self.seen = np.array(uv)
            return (uv[0] + uv[1]).astype(F)

    unit = Unit()
    out = run(tmp_path, "OUT = vec4(texture(tex, IN * 2.0).r, texture(tex, vec2(0.25, 0.5)).a, 0.0, 0.0);", lanes=2, pre="uniform sampler2D tex;",
              inputs={"IN": ("vec2", np.array([[1.0, 2.0], [3.0, 4.0]], dtype=F))}, samplers={"tex": unit})
    assert out[0].tolist() == [8.0, 12.0] and np.broadcast_to(out[1], (2,)).tolist() == [1.0, 1.0]
    assert unit.seen.shape == (2, 2)


def test_preprocessor_conditionals_nest_and_macros_expand_inside_expressions(tmp_path):
    pre = """
    #define STEPS 4
    #define SCALE (1.0 + 1.0)
    #define ENABLED
    #ifdef ENABLED
      #ifndef MISSING
        #define PICK 7.0
      #else
        #define PICK 8.0
      #endif
    #else
      #define PICK 9.0
    #endif
    """
    body = "float s = 0.0; for (int i = 0; i < STEPS; ++i) { s += SCALE; } OUT = vec4(s, PICK, SCALE * 3.0, float(STEPS / 3));"
    assert run(tmp_path, body, pre=pre)[:, 0].tolist() == [8.0, 7.0, 6.0, 1.0]
