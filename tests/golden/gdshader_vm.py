"""A small SIMT interpreter for the GDShader subset the reference's atmosphere shaders are written in.

TEST INFRASTRUCTURE, fixture generation only.  `make_reference_vectors.py` feeds it the reference's OWN shader source
text (read from /root/reference at generation time, never copied into this repo), executes `vertex()` / `fragment()`
for a set of pixels, and commits inputs + outputs as `tests/golden/reference_exec_*.npz`.  Those vectors are outputs
of the reference itself, run here: they pin oracle/atmo_oracle.c (and through it the HIP path) to the reference text
rather than to this repo's reading of it.

What the interpreter supplies is only what the shading LANGUAGE and the ENGINE define, not the reference:
  * the preprocessor (#include, #define NAME [tokens], #ifdef / #ifndef / #else / #endif), comments;
  * GLSL ES 3.0 expression / statement semantics for the constructs the files use: functions with in / out / inout
    parameters, structs, const, if / else, for, return, discard, swizzles, constructors, matrix / vector algebra;
  * IEEE binary32 arithmetic (numpy float32; `exp` and `pow` are evaluated in binary64 and rounded once);
  * the built-in functions, spelled as DESIGN.md section 2 states them (normalize = v * (1 / sqrt(dot)), mix(a, b, t) =
    a * (1 - t) + b * t, sums left to right, max / min ignore a NaN operand);
  * the texture units (`texture`, `texelFetch`) as callables given by the caller (tests/golden/vm_textures.py).

All lanes (pixels) execute together: a value is a numpy array whose LAST axis is the lane axis (length 1 = uniform);
control flow is executed under lane masks, the way the GPU runs it.
"""
from __future__ import annotations

import math
import os
import re

import numpy as np

F32 = np.float32
_VEC = {"vec2": 2, "vec3": 3, "vec4": 4}
_IVEC = {"ivec2": 2, "ivec3": 3, "ivec4": 4}
_UVEC = {"uvec2": 2, "uvec3": 3, "uvec4": 4}
_MAT = {"mat2": 2, "mat3": 3, "mat4": 4}
_SAMPLERS = ("sampler2D", "sampler3D", "samplerCube")
_SCALARS = ("float", "int", "uint", "bool")
_BUILTIN_TYPES = set(_VEC) | set(_IVEC) | set(_UVEC) | set(_MAT) | set(_SAMPLERS) | set(_SCALARS) | {"void"}
_SWZ = {c: i for s in ("xyzw", "rgba", "stpq") for i, c in enumerate(s)}


class ShaderError(Exception):
    pass


# ----------------------------------------------------------------------------------------------- preprocessor + lexer
_TOKEN = re.compile(r"""
    (?P<float>(?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)
  | (?P<hex>0[xX][0-9a-fA-F]+[uU]?)
  | (?P<int>\d+[uU]?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op>\+\+|--|\+=|-=|\*=|/=|==|!=|<=|>=|&&|\|\||<<|>>|[-+*/%<>=!&|^~?:;,.(){}\[\]])
  | (?P<ws>\s+)
""", re.X)


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _lex(line: str):
    pos, out = 0, []
    while pos < len(line):
        m = _TOKEN.match(line, pos)
        if not m:
            raise ShaderError(f"cannot tokenise: {line[pos:pos + 20]!r}")
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group(0)))
    return out


def preprocess(path: str, defines: dict | None = None, force_defines: dict | None = None):
    """Token list of `path` after includes, conditionals and object-like macros.  `defines`: predefined macros (an in-file
    #define of the same name replaces them, as in C); `force_defines`: macros that WIN over in-file #defines of the same
    name -- how a step count other than the shipped one is run through the reference text unchanged (the variant files
    #define ATMOSPHERE_RAYMARCH_STEPS themselves, planet_atmosphere_no_clouds.gdshader:4)."""
    macros = {k: _lex(str(v)) for k, v in (defines or {}).items()}
    forced = {k: _lex(str(v)) for k, v in (force_defines or {}).items()}
    macros.update(forced)
    tokens = []

    def expand(toks, depth=0):
        for kind, text in toks:
            if kind == "id" and text in macros and depth < 16:
                expand(macros[text], depth + 1)
            else:
                tokens.append((kind, text))

    def run(p):
        with open(p, "r", encoding="utf-8") as fh:
            src = _strip_comments(fh.read())
        stack = []  # (this_branch_active, parent_active)
        active = True
        for line in src.split("\n"):
            s = line.strip()
            if s.startswith("#"):
                m = re.match(r"#\s*(\w+)\s*(.*)", s)
                d, rest = m.group(1), m.group(2).strip()
                if d in ("ifdef", "ifndef"):
                    cond = (rest.split()[0] in macros) == (d == "ifdef")
                    stack.append((active, cond))
                    active = active and cond
                elif d == "else":
                    parent, cond = stack[-1]
                    active = parent and not cond
                elif d == "endif":
                    active, _ = stack.pop()
                elif not active:
                    pass
                elif d == "define":
                    mm = re.match(r"(\w+)\s*(.*)", rest)
                    if mm.group(1) not in forced:
                        macros[mm.group(1)] = _lex(mm.group(2))
                elif d == "include":
                    run(os.path.join(os.path.dirname(p), rest.strip('"')))
                else:
                    raise ShaderError(f"unsupported directive #{d}")
            elif active:
                expand(_lex(line))
        if stack:
            raise ShaderError(f"unterminated conditional in {p}")

    run(path)
    return tokens, set(macros)


# ------------------------------------------------------------------------------------------------------------ parser
class Parser:
    def __init__(self, tokens):
        self.t = tokens
        self.i = 0
        self.structs = {}
        self.functions = {}
        self.uniforms = {}   # name -> (type, hints, default expr or None)
        self.varyings = {}
        self.consts = []     # (type, name, expr)

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, text):
        if self.peek()[1] == text:
            self.i += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            raise ShaderError(f"expected {text!r}, got {self.peek()[1]!r} (token {self.i})")

    def is_type(self, text):
        return text in _BUILTIN_TYPES or text in self.structs

    def ident(self):
        kind, text = self.next()
        if kind != "id":
            raise ShaderError(f"expected identifier, got {text!r}")
        return text

    # -- top level
    def parse(self):
        while self.peek()[0] != "eof":
            kind, text = self.peek()
            if text in ("shader_type", "render_mode"):
                while self.next()[1] != ";":
                    pass
            elif text == "uniform":
                self.next()
                ty, name, hints, default = self.ident(), self.ident(), [], None
                if self.accept(":"):
                    hints.append(self.ident())
                    while self.accept(","):
                        hints.append(self.ident())
                if self.accept("="):
                    default = self.expr()
                self.expect(";")
                self.uniforms[name] = (ty, hints, default)
            elif text == "varying":
                self.next()
                ty, name = self.ident(), self.ident()
                self.expect(";")
                self.varyings[name] = ty
            elif text == "struct":
                self.next()
                name = self.ident()
                self.expect("{")
                fields = []
                while not self.accept("}"):
                    fty, fname = self.ident(), self.ident()
                    self.expect(";")
                    fields.append((fty, fname))
                self.expect(";")
                self.structs[name] = fields
            elif text == "const":
                self.next()
                ty, name = self.ident(), self.ident()
                self.expect("=")
                e = self.expr()
                self.expect(";")
                self.consts.append((ty, name, e))
            else:
                ty, name = self.ident(), self.ident()
                self.expect("(")
                params = []
                while not self.accept(")"):
                    qual = "in"
                    while self.peek()[1] in ("in", "out", "inout", "const"):
                        q = self.next()[1]
                        qual = q if q != "const" else qual
                    params.append((qual, self.ident(), self.ident()))
                    if not self.accept(","):
                        self.expect(")")
                        break
                body = self.block()
                self.functions[name] = (ty, params, body)
        return self

    # -- statements
    def block(self):
        self.expect("{")
        stmts = []
        while not self.accept("}"):
            stmts.append(self.statement())
        return ("block", stmts)

    def statement(self):
        kind, text = self.peek()
        if text == "{":
            return self.block()
        if text == "if":
            self.next()
            self.expect("(")
            c = self.expr()
            self.expect(")")
            a = self.statement()
            b = self.statement() if self.accept("else") else None
            return ("if", c, a, b)
        if text == "for":
            self.next()
            self.expect("(")
            init = self.statement()  # consumes its ';'
            cond = self.expr()
            self.expect(";")
            it = self.expr()
            self.expect(")")
            return ("for", init, cond, it, self.statement())
        if text == "return":
            self.next()
            e = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", e)
        if text == "discard":
            self.next()
            self.expect(";")
            return ("discard",)
        if text in ("break", "continue", "while", "do", "switch"):
            raise ShaderError(f"statement {text!r} is outside the supported subset")
        if text == "const" or (kind == "id" and self.is_type(text) and self.peek(1)[0] == "id"):
            if text == "const":
                self.next()
            ty = self.ident()
            decls = []
            while True:
                name = self.ident()
                init = self.assign() if self.accept("=") else None
                decls.append((name, init))
                if not self.accept(","):
                    break
            self.expect(";")
            return ("decl", ty, decls)
        e = self.expr()
        self.expect(";")
        return ("expr", e)

    # -- expressions
    def expr(self):
        return self.assign()

    def assign(self):
        lhs = self.ternary()
        if self.peek()[1] in ("=", "+=", "-=", "*=", "/="):
            op = self.next()[1]
            return ("assign", op, lhs, self.assign())
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.assign()
            self.expect(":")
            return ("ternary", c, a, self.assign())
        return c

    _LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"),
               ("*", "/", "%")]

    def binary(self, level):
        if level == len(self._LEVELS):
            return self.unary()
        lhs = self.binary(level + 1)
        while self.peek()[1] in self._LEVELS[level] and self.peek()[0] == "op":
            op = self.next()[1]
            lhs = ("bin", op, lhs, self.binary(level + 1))
        return lhs

    def unary(self):
        text = self.peek()[1]
        if text in ("-", "+", "!", "~"):
            self.next()
            return ("un", text, self.unary())
        if text in ("++", "--"):
            self.next()
            return ("preinc", text, self.unary())
        return self.postfix()

    def postfix(self):
        kind, text = self.next()
        if kind == "float":
            e = ("lit", "float", float(text))
        elif kind == "hex":
            u = text[-1] in "uU"
            e = ("lit", "uint" if u else "int", int(text.rstrip("uU"), 16))
        elif kind == "int":
            u = text[-1] in "uU"
            e = ("lit", "uint" if u else "int", int(text.rstrip("uU")))
        elif text in ("true", "false"):
            e = ("lit", "bool", text == "true")
        elif kind == "id":
            if self.peek()[1] == "(":
                self.next()
                args = []
                while not self.accept(")"):
                    args.append(self.assign())
                    if not self.accept(","):
                        self.expect(")")
                        break
                e = ("call", text, args)
            else:
                e = ("var", text)
        elif text == "(":
            e = self.expr()
            self.expect(")")
        else:
            raise ShaderError(f"unexpected token {text!r}")
        while True:
            if self.accept("."):
                e = ("member", e, self.ident())
            elif self.accept("["):
                idx = self.expr()
                self.expect("]")
                e = ("index", e, idx)
            elif self.peek()[1] in ("++", "--"):
                e = ("postinc", self.next()[1], e)
            else:
                return e


# ------------------------------------------------------------------------------------------------------------ values
class V:
    """A typed SIMT value.  `a`: numpy array with the lane axis last (float32 / int32 / uint32 / bool), a dict of V for a
    struct, or the caller's object for a sampler."""
    __slots__ = ("t", "a")

    def __init__(self, t, a):
        self.t, self.a = t, a

    def __repr__(self):
        return f"V({self.t}, {self.a!r})"


def _f(x):
    return V("float", np.asarray(x, dtype=F32).reshape(-1))


def _base(t):
    if t in _VEC or t in _MAT or t == "float":
        return "float"
    if t in _IVEC or t == "int":
        return "int"
    if t in _UVEC or t == "uint":
        return "uint"
    return t


_DT = {"float": F32, "int": np.int32, "uint": np.uint32, "bool": np.bool_}


def _vec_type(base, n):
    if n == 1:
        return base
    return {"float": "vec", "int": "ivec", "uint": "uvec"}[base] + str(n)


def _where(mask, new, old):
    """Lane-masked merge of two arrays (lane axis last)."""
    return np.where(mask, new, old)


def _exp32(x):
    with np.errstate(all="ignore"):
        return np.exp(x.astype(np.float64)).astype(F32)


def _pow32(x, y):
    with np.errstate(all="ignore"):
        return np.power(x.astype(np.float64), y.astype(np.float64)).astype(F32)  # NaN for x < 0 with fractional y only
        # (GLSL leaves x < 0 undefined; hardware computes exp2(y * log2(x)) = NaN, see _pow below)


def _fma(a, b, c):
    """a * b + c rounded once (the float32 product is exact in float64; the second rounding float64 -> float32 can differ
    from a true fused operation only in half-way cases -- good enough for the sensitivity runs this is for)."""
    with np.errstate(all="ignore"):
        return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(F32)


_CONTRACT = [False]  # sensitivity runs only (Machine(conventions={"fma": True})): sums of products contracted into FMAs


def _dot(a, b):
    r = a[0] * b[0]
    for k in range(1, a.shape[0]):
        r = _fma(a[k], b[k], r) if _CONTRACT[0] else r + a[k] * b[k]
    return r


def _twin_calls(a, b):
    """Both statements are `X = f(args);` (possibly wrapped in a one-statement block) with the same X and the same args."""
    def only(st):
        while st[0] == "block" and len(st[1]) == 1:
            st = st[1][0]
        if st[0] == "expr" and st[1][0] == "assign" and st[1][1] == "=" and st[1][3][0] == "call":
            return st[1][2], st[1][3][2]
        return None
    x, y = only(a), only(b)
    return x is not None and y is not None and x == y


class Frame:
    __slots__ = ("scopes", "masks", "ret", "returned", "ret_type")

    def __init__(self, ret_type, mask):
        self.scopes = [{}]
        self.masks = [mask]  # the lane mask each scope was opened under: its variables are live for those lanes only
        self.ret = None
        self.returned = None  # None = no lane has returned; else bool (N,)
        self.ret_type = ret_type


class Machine:
    """Executes the functions of one parsed shader over `lanes` lanes."""

    def __init__(self, parser: Parser, lanes: int, samplers: dict, uniforms: dict | None = None,
                 source_color=lambda c: c, merge_twin_calls: bool = False, conventions: dict | None = None):
        """merge_twin_calls: an if / else whose two branches assign the same variable from calls with the same argument
        list (cloud_funcs.gdshaderinc:132-136: get_density(pos, time, settings) / get_density_low(pos, time, settings), the
        same function of the same arguments once CLOUDS_ALWAYS_LOW_QUALITY is defined, main:49) counts as ONE call site for
        the quad derivatives of an implicit-LOD texture fetch inside: a quad partner that took the other branch still
        contributes its coordinate, as it does after a compiler has inlined and merged the two branches.  False = literal:
        only lanes active at this very execution of the call contribute.
        conventions: sensitivity runs only (tests/golden/sensitivity.py) -- {"normalize": "div"} for v / sqrt(dot) instead
        of v * (1 / sqrt(dot)), {"mix": "lerp"} for a + (b - a) t instead of a (1 - t) + b t, {"fma": True} for contracted
        sums of products."""
        self.p = parser
        self.n = lanes
        self.merge_twin_calls = merge_twin_calls
        self.conv = dict(conventions or {})
        _CONTRACT[0] = bool(self.conv.get("fma"))
        self.reach_stack = []
        self.globals = {}
        self.frames = []
        self.discarded = np.zeros(lanes, dtype=bool)
        self.calls = {}
        for name, (ty, hints, default) in parser.uniforms.items():
            if ty in _SAMPLERS:
                self.globals[name] = V(ty, samplers.get(name))
                continue
            if uniforms is not None and name in uniforms:
                val = self.from_host(ty, uniforms[name])
            elif default is not None:
                val = self.convert(ty, self.eval(default, None))
                if "source_color" in hints:  # the engine converts sRGB -> linear when it uploads a source_color value
                    rgb = source_color(val.a[:3, 0].astype(np.float64))
                    val = V(ty, np.concatenate([np.asarray(rgb, dtype=F32), val.a[3:, 0]]).reshape(-1, 1))
            else:
                val = self.zero(ty)
            self.globals[name] = val
        for name, ty in parser.varyings.items():
            self.globals[name] = self.zero(ty)
        for ty, name, e in parser.consts:
            self.globals[name] = self.convert(ty, self.eval(e, None))

    # -- host <-> value
    def from_host(self, ty, x):
        if isinstance(x, V):
            return x
        a = np.asarray(x)
        if ty in _MAT:
            n = _MAT[ty]
            a = a.astype(F32)
            if a.ndim == 1:      # flat column-major, uniform
                a = a.reshape(n, n, 1)
            elif a.ndim == 2:    # (n*n, lanes) or (n, n): treat (n, n) as [col][row]
                a = a.reshape(n, n, 1) if a.shape == (n, n) else a.reshape(n, n, -1)
            return V(ty, a)
        dt = _DT[_base(ty)]
        if ty in _SCALARS:
            return V(ty, a.astype(dt).reshape(-1))
        n = (_VEC.get(ty) or _IVEC.get(ty) or _UVEC.get(ty))
        a = a.astype(dt)
        return V(ty, a.reshape(n, -1))

    def zero(self, ty):
        if ty in self.p.structs:
            return V(ty, {fn: self.zero(ft) for ft, fn in self.p.structs[ty]})
        if ty in _MAT:
            return V(ty, np.zeros((_MAT[ty], _MAT[ty], 1), dtype=F32))
        if ty in _SCALARS:
            return V(ty, np.zeros(1, dtype=_DT[ty]))
        n = _VEC.get(ty) or _IVEC.get(ty) or _UVEC.get(ty)
        if n is None:
            raise ShaderError(f"cannot zero-initialise {ty}")
        return V(ty, np.zeros((n, 1), dtype=_DT[_base(ty)]))

    def convert(self, ty, v):
        """Initialiser conversion: the language is strict, only int literal -> declared scalar type is tolerated."""
        if v.t == ty:
            return v
        if ty in _SCALARS and v.t in _SCALARS:
            return V(ty, v.a.astype(_DT[ty]))
        raise ShaderError(f"type mismatch: {v.t} where {ty} is expected")

    # -- variables
    def lookup(self, name, with_mask=False):
        if self.frames:
            fr = self.frames[-1]
            for k in range(len(fr.scopes) - 1, -1, -1):
                if name in fr.scopes[k]:
                    return (fr.scopes[k], fr.scopes[k][name], fr.masks[k]) if with_mask else (fr.scopes[k], fr.scopes[k][name])
        if name in self.globals:
            return (self.globals, self.globals[name], None) if with_mask else (self.globals, self.globals[name])
        raise ShaderError(f"unknown identifier {name!r}")

    def push_scope(self, mask):
        self.frames[-1].scopes.append({})
        self.frames[-1].masks.append(mask)

    def pop_scope(self):
        self.frames[-1].scopes.pop()
        self.frames[-1].masks.pop()

    def merge(self, old: V, new: V, mask):
        if mask is None:
            return new
        if isinstance(new.a, dict):
            return V(new.t, {k: self.merge(old.a[k], new.a[k], mask) for k in new.a})
        return V(new.t, _where(mask, new.a, old.a))

    def store(self, lv, val: V, mask):
        kind = lv[0]
        if kind == "var":
            scope, old, scope_mask = self.lookup(lv[1], with_mask=True)
            if old.t != val.t:
                val = self.convert(old.t, val)
            if mask is not None and scope is not self.globals:
                # every lane the variable is live for is active: plain overwrite (keeps loop counters uniform).  For a PARAMETER "live"
                # includes the lanes that have returned from the function since: an `out` / `inout` parameter of a lane that returned
                # early is still copied back to the caller and must keep what that lane wrote (round 5: found by
                # tests/test_gdshader_vm_kat.py -- the shortcut used to look at the still-running lanes only, for every variable)
                params = self.frames[-1].scopes[0] if self.frames else None
                need = scope_mask if scope is params else self.live(scope_mask)
                if need is not None and not (need & ~mask).any():
                    mask = None
            scope[lv[1]] = self.merge(old, val, mask)
        elif kind == "member":
            base = self.eval(lv[1], mask)
            if isinstance(base.a, dict):
                fields = dict(base.a)
                fields[lv[2]] = self.merge(fields[lv[2]], self.convert(fields[lv[2]].t, val), None)
                self.store(lv[1], V(base.t, fields), mask)
            else:
                idx = [_SWZ[c] for c in lv[2]]
                lanes = max(base.a.shape[-1], val.a.shape[-1])
                arr = np.broadcast_to(base.a, base.a.shape[:-1] + (lanes,)).copy()
                if len(idx) == 1:
                    arr[idx[0]] = val.a
                else:
                    for k, i in enumerate(idx):
                        arr[i] = val.a[k]
                self.store(lv[1], V(base.t, arr), mask)
        elif kind == "index":
            base = self.eval(lv[1], mask)
            i = self.uniform_int(self.eval(lv[2], mask))
            lanes = max(base.a.shape[-1], val.a.shape[-1])
            arr = np.broadcast_to(base.a, base.a.shape[:-1] + (lanes,)).copy()
            arr[i] = val.a
            self.store(lv[1], V(base.t, arr), mask)
        else:
            raise ShaderError(f"not an l-value: {kind}")

    @staticmethod
    def uniform_int(v):
        if v.a.shape[-1] != 1:
            raise ShaderError("a varying index / loop bound is outside the supported subset")
        return int(v.a.reshape(-1)[0])

    # -- masks
    def live(self, mask):
        """`mask` minus the lanes that already returned from the current function or were discarded."""
        fr = self.frames[-1] if self.frames else None
        dead = None
        if fr is not None and fr.returned is not None:
            dead = fr.returned
        if dead is None:
            return mask
        return ~dead if mask is None else (mask & ~dead)

    @staticmethod
    def _and(mask, c):
        return c if mask is None else (mask & c)

    # -- statements
    def exec(self, st, mask):
        mask = self.live(mask)
        if mask is not None and not mask.any():
            return
        kind = st[0]
        if kind == "block":
            self.push_scope(mask)
            for s in st[1]:
                self.exec(s, mask)
            self.pop_scope()
        elif kind == "decl":
            for name, init in st[2]:
                v = self.zero(st[1]) if init is None else self.convert(st[1], self.eval(init, mask))
                self.frames[-1].scopes[-1][name] = v
        elif kind == "expr":
            self.eval(st[1], mask)
        elif kind == "if":
            c = self.eval(st[1], mask)
            if c.t != "bool":
                raise ShaderError("if condition must be bool")
            if c.a.shape[-1] == 1:
                if bool(c.a[0]):
                    self.exec(st[2], mask)
                elif st[3] is not None:
                    self.exec(st[3], mask)
            else:
                twin = self.merge_twin_calls and st[3] is not None and _twin_calls(st[2], st[3])
                if twin:
                    self.reach_stack.append(self._lanes(mask))
                self.exec(st[2], self._and(mask, c.a))
                if st[3] is not None:
                    self.exec(st[3], self._and(mask, ~c.a))
                if twin:
                    self.reach_stack.pop()
        elif kind == "for":
            self.push_scope(mask)
            self.exec(st[1], mask)
            guard = 0
            while True:
                c = self.eval(st[2], mask)
                if c.a.shape[-1] != 1:
                    raise ShaderError("a varying loop condition is outside the supported subset")
                if not bool(c.a[0]):
                    break
                self.exec(st[4], mask)
                self.eval(st[3], mask)
                guard += 1
                if guard > 100000:
                    raise ShaderError("loop does not terminate")
            self.pop_scope()
        elif kind == "return":
            fr = self.frames[-1]
            if st[1] is not None:
                v = self.convert(fr.ret_type, self.eval(st[1], mask))
                fr.ret = v if fr.ret is None else self.merge(fr.ret, v, mask)
            done = np.ones(self.n, dtype=bool) if mask is None else mask
            fr.returned = done if fr.returned is None else (fr.returned | done)
        elif kind == "discard":
            self.discarded |= np.ones(self.n, dtype=bool) if mask is None else mask
        else:
            raise ShaderError(f"unknown statement {kind}")

    # -- expressions
    def eval(self, e, mask) -> V:
        kind = e[0]
        if kind == "lit":
            return V(e[1], np.asarray([e[2]], dtype=_DT[e[1]]))
        if kind == "var":
            return self.lookup(e[1])[1]
        if kind == "bin":
            return self.binary(e[1], e[2], e[3], mask)
        if kind == "un":
            v = self.eval(e[2], mask)
            if e[1] == "-":
                return V(v.t, -v.a)
            if e[1] == "+":
                return v
            if e[1] == "!":
                return V("bool", ~v.a)
            return V(v.t, ~v.a)
        if kind == "assign":
            rhs = self.eval(e[3], mask)
            if e[1] != "=":
                cur = self.eval(e[2], mask)
                rhs = self.arith(e[1][0], cur, rhs)
            self.store(e[2], rhs, self.live(mask))
            return rhs
        if kind in ("preinc", "postinc"):
            cur = self.eval(e[2], mask)
            one = V(cur.t, np.asarray([1], dtype=cur.a.dtype))
            new = self.arith("+" if e[1] == "++" else "-", cur, one)
            self.store(e[2], new, self.live(mask))
            return new if kind == "preinc" else cur
        if kind == "ternary":
            c = self.eval(e[1], mask)
            if c.a.shape[-1] == 1:
                return self.eval(e[2] if bool(c.a[0]) else e[3], mask)
            a, b = self.eval(e[2], self._and(mask, c.a)), self.eval(e[3], self._and(mask, ~c.a))
            return V(a.t, _where(c.a, a.a, b.a))
        if kind == "member":
            base = self.eval(e[1], mask)
            if isinstance(base.a, dict):
                return base.a[e[2]]
            idx = [_SWZ[c] for c in e[2]]
            b = _base(base.t)
            if len(idx) == 1:
                return V(b, base.a[idx[0]])
            return V(_vec_type(b, len(idx)), base.a[idx])
        if kind == "index":
            base = self.eval(e[1], mask)
            i = self.uniform_int(self.eval(e[2], mask))
            if base.t in _MAT:
                return V("vec" + str(_MAT[base.t]), base.a[i])
            return V(_base(base.t), base.a[i])
        if kind == "call":
            return self.call(e[1], e[2], mask)
        raise ShaderError(f"unknown expression {kind}")

    def binary(self, op, le, re_, mask):
        a = self.eval(le, mask)
        if op in ("&&", "||"):
            if a.a.shape[-1] == 1:  # uniform short circuit
                if bool(a.a[0]) == (op == "||"):
                    return a
                return self.eval(re_, mask)
            b = self.eval(re_, mask)
            return V("bool", (a.a & b.a) if op == "&&" else (a.a | b.a))
        if _CONTRACT[0] and op in ("+", "-"):
            fused = self._contract(op, le, re_, a, mask)
            if fused is not None:
                return fused
        b = self.eval(re_, mask)
        if op in ("==", "!=", "<", ">", "<=", ">="):
            if a.t != b.t or a.t not in _SCALARS:
                raise ShaderError(f"comparison {a.t} {op} {b.t}")
            with np.errstate(all="ignore"):
                r = {"==": np.equal, "!=": np.not_equal, "<": np.less, ">": np.greater, "<=": np.less_equal,
                     ">=": np.greater_equal}[op](a.a, b.a)
            return V("bool", r)
        return self.arith(op, a, b)

    def _contract(self, op, le, re_, a, mask):
        """x * y + c / c + x * y / x * y - c / c - x * y as one rounding, for float scalars and vectors (sensitivity runs)."""
        def product(e):
            return e[0] == "bin" and e[1] == "*"
        if product(re_):
            x, y = self.eval(re_[2], mask), self.eval(re_[3], mask)
            c, sign_p, sign_c = a, (1.0 if op == "+" else -1.0), 1.0
        elif product(le):
            x, y = self.eval(le[2], mask), self.eval(le[3], mask)
            c, sign_p, sign_c = self.eval(re_, mask), 1.0, (1.0 if op == "+" else -1.0)
        else:
            return None
        if any(_base(v.t) != "float" or v.t in _MAT for v in (x, y, c)):
            return None
        t = self.arith("+", self.arith("*", x, y), c).t
        return V(t, _fma(F32(sign_p) * x.a, y.a, F32(sign_c) * c.a))

    def arith(self, op, a: V, b: V) -> V:
        ba, bb = _base(a.t), _base(b.t)
        if ba != bb or ba not in ("float", "int", "uint"):
            raise ShaderError(f"operator {op} on {a.t} and {b.t}")
        with np.errstate(all="ignore"):
            if op == "*" and (a.t in _MAT or b.t in _MAT):
                if a.t in _MAT and b.t in _VEC:      # M v, summed left to right over the columns
                    n = _MAT[a.t]
                    r = a.a[0] * b.a[0]
                    for k in range(1, n):
                        r = _fma(a.a[k], b.a[k], r) if _CONTRACT[0] else r + a.a[k] * b.a[k]
                    return V(b.t, r)
                if a.t in _VEC and b.t in _MAT:      # v M (section 5.10: the vector is a row vector): component c = dot(v, M[c]), left to right
                    n = _MAT[b.t]
                    comps = []
                    for c in range(n):
                        r = a.a[0] * b.a[c][0]
                        for k in range(1, n):
                            r = r + a.a[k] * b.a[c][k]
                        comps.append(r)
                    lanes = max(x.shape[-1] for x in comps)
                    return V(a.t, np.stack([np.broadcast_to(x, (lanes,)) for x in comps]))
                if a.t in _MAT and b.t in _MAT:      # (A B)[col] = A B[col]
                    n = _MAT[a.t]
                    cols = []
                    for c in range(n):
                        r = a.a[0] * b.a[c][0]
                        for k in range(1, n):
                            r = r + a.a[k] * b.a[c][k]
                        cols.append(r)
                    lanes = max(x.shape[-1] for x in cols)
                    return V(a.t, np.stack([np.broadcast_to(x, (n, lanes)) for x in cols]))
                if b.t == "float":
                    return V(a.t, a.a * b.a)
                if a.t == "float":
                    return V(b.t, a.a * b.a)
                raise ShaderError(f"operator * on {a.t} and {b.t}")
            if a.t != b.t and a.t not in _SCALARS and b.t not in _SCALARS:
                raise ShaderError(f"operator {op} on {a.t} and {b.t}")
            t = a.t if a.t not in _SCALARS else b.t
            x, y = a.a, b.a
            if op == "+":
                r = x + y
            elif op == "-":
                r = x - y
            elif op == "*":
                r = x * y
            elif op == "/":
                if ba == "float":
                    r = x / y
                else:
                    r = (np.trunc(x.astype(np.float64) / np.where(y == 0, 1, y))).astype(x.dtype)
            elif op == "%":
                r = np.fmod(x, y)
            elif op == "&":
                r = x & y
            elif op == "|":
                r = x | y
            elif op == "^":
                r = x ^ y
            elif op == "<<":
                r = x << y
            elif op == ">>":
                r = x >> y
            else:
                raise ShaderError(f"operator {op}")
        if r.dtype != _DT[ba]:
            r = r.astype(_DT[ba])
        return V(t, r)

    # -- calls
    def call(self, name, arg_exprs, mask) -> V:
        self.calls[name] = self.calls.get(name, 0) + 1
        if name in _BUILTIN_TYPES:
            return self.construct(name, [self.eval(x, mask) for x in arg_exprs])
        if name in self.p.functions:
            return self.call_user(name, arg_exprs, mask)
        args = [self.eval(x, mask) for x in arg_exprs]
        fn = getattr(self, "bi_" + name, None)
        if fn is None:
            raise ShaderError(f"unknown function {name!r}")
        return fn(mask, *args)

    def call_user(self, name, arg_exprs, mask, host_args=None) -> V:
        ret_type, params, body = self.p.functions[name]
        if len(arg_exprs) != len(params):
            raise ShaderError(f"{name}: {len(arg_exprs)} arguments for {len(params)} parameters")
        fr = Frame(ret_type, self.live(mask))
        for (qual, ty, pname), ae in zip(params, arg_exprs):
            if qual == "out":
                v = self.zero(ty)
            else:
                v = host_args[pname] if host_args is not None else self.eval(ae, mask)
                if v.t != ty:
                    raise ShaderError(f"{name}: argument {pname} is {v.t}, expected {ty}")
            fr.scopes[0][pname] = v
        self.frames.append(fr)
        self.exec(body, mask)
        self.frames.pop()
        for (qual, ty, pname), ae in zip(params, arg_exprs):
            if qual in ("out", "inout") and host_args is None:
                self.store(ae, fr.scopes[0][pname], self.live(mask))
        if host_args is not None:
            return fr.scopes[0]
        if ret_type == "void":
            return V("void", None)
        if fr.ret is None:
            raise ShaderError(f"{name}: no lane returned a value")
        return fr.ret

    def run(self, name, mask=None):
        """Execute a parameterless entry point (`vertex`, `fragment`)."""
        fr = Frame("void", mask)
        self.frames.append(fr)
        self.exec(self.p.functions[name][2], mask)
        self.frames.pop()

    def construct(self, ty, args):
        if ty in _SCALARS:
            (v,) = args
            if v.t not in _SCALARS:
                raise ShaderError(f"{ty}({v.t})")
            if ty in ("int", "uint") and v.t == "float":
                with np.errstate(all="ignore"):
                    return V(ty, np.trunc(v.a).astype(np.int64).astype(_DT[ty]))
            return V(ty, v.a.astype(_DT[ty]))
        if ty in _MAT:
            n = _MAT[ty]
            if len(args) == 1 and args[0].t == "float":
                a = np.zeros((n, n, args[0].a.shape[-1]), dtype=F32)
                for k in range(n):
                    a[k, k] = args[0].a
                return V(ty, a)
            if len(args) == n and all(x.t == "vec" + str(n) for x in args):
                lanes = max(x.a.shape[-1] for x in args)
                return V(ty, np.stack([np.broadcast_to(x.a, (n, lanes)) for x in args]))
            # GLSL ES 3.00 section 5.4.2: scalars and vectors are consumed left to right and fill the matrix in column-major order
            comps = []
            for v in args:
                if v.t in _MAT or isinstance(v.a, dict) or _base(v.t) != "float":
                    raise ShaderError(f"{ty} constructor form outside the supported subset")
                src = v.a if v.a.ndim == 2 else v.a[None, :]
                comps.extend(src[k] for k in range(src.shape[0]))
            if len(comps) != n * n:
                raise ShaderError(f"{ty} constructed from {len(comps)} components")
            lanes = max(c.shape[-1] for c in comps)
            flat = np.stack([np.broadcast_to(c, (lanes,)) for c in comps]).astype(F32)
            return V(ty, flat.reshape(n, n, lanes))
        n = _VEC.get(ty) or _IVEC.get(ty) or _UVEC.get(ty)
        base = _base(ty)
        comps = []
        for v in args:
            if v.t in _MAT or isinstance(v.a, dict):
                raise ShaderError(f"{ty}({v.t})")
            src = v.a if v.a.ndim == 2 else v.a[None, :]
            if _base(v.t) == "float" and base != "float":
                with np.errstate(all="ignore"):
                    src = np.trunc(src).astype(np.int64)
            comps.extend(src[k].astype(_DT[base]) for k in range(src.shape[0]))
        if len(comps) == 1:
            comps = comps * n
        if len(comps) > n and len(args) >= 1:
            # section 5.4.2: "it is an error to provide extra arguments beyond this last used argument" -- but the last USED argument may
            # be longer than what is left to fill (vec3(vec4), vec2(vec3)): its remaining components are dropped
            last = args[-1]
            last_n = last.a.shape[0] if last.a.ndim == 2 else 1
            if len(comps) - last_n < n:
                comps = comps[:n]
        if len(comps) != n:
            raise ShaderError(f"{ty} constructed from {len(comps)} components")
        lanes = max(c.shape[-1] for c in comps)
        return V(ty, np.stack([np.broadcast_to(c, (lanes,)) for c in comps]))

    # -- built-in functions (GLSL ES 3.0 section 8, spelled as DESIGN.md section 2 states them)
    @staticmethod
    def _same(*vs):
        t = next((v.t for v in vs if v.t not in _SCALARS), vs[0].t)
        for v in vs:
            if _base(v.t) != _base(t):
                raise ShaderError("mixed base types in a built-in call")
        return t

    def bi_length(self, m, v):
        return V("float", np.sqrt(_dot(v.a, v.a)) if v.a.ndim == 2 else np.abs(v.a))

    def bi_distance(self, m, a, b):
        d = a.a - b.a
        return V("float", np.sqrt(_dot(d, d)))

    def bi_dot(self, m, a, b):
        return V("float", _dot(a.a, b.a))

    def bi_normalize(self, m, v):
        with np.errstate(all="ignore"):
            if self.conv.get("normalize") == "div":
                return V(v.t, v.a / np.sqrt(_dot(v.a, v.a)))
            return V(v.t, v.a * (F32(1.0) / np.sqrt(_dot(v.a, v.a))))

    def bi_sqrt(self, m, v):
        with np.errstate(all="ignore"):
            return V(v.t, np.sqrt(v.a))

    def bi_exp(self, m, v):
        return V(v.t, _exp32(v.a))

    def bi_pow(self, m, a, b):
        t = self._same(a, b)
        with np.errstate(all="ignore"):
            r = _pow32(a.a, b.a)
            r = np.where(a.a < 0, F32(np.nan), r)  # exp2(y log2 x): NaN for a negative base
        return V(t, r.astype(F32))

    def bi_abs(self, m, v):
        return V(v.t, np.abs(v.a))

    def bi_floor(self, m, v):
        return V(v.t, np.floor(v.a))

    def bi_fract(self, m, v):
        return V(v.t, v.a - np.floor(v.a))

    def bi_max(self, m, a, b):
        return V(self._same(a, b), np.fmax(a.a, b.a))

    def bi_min(self, m, a, b):
        return V(self._same(a, b), np.fmin(a.a, b.a))

    def bi_clamp(self, m, x, lo, hi):
        return V(self._same(x, lo, hi), np.fmin(np.fmax(x.a, lo.a), hi.a))

    def bi_mix(self, m, a, b, t):
        with np.errstate(all="ignore"):
            if self.conv.get("mix") == "lerp":
                return V(self._same(a, b, t), a.a + (b.a - a.a) * t.a)
            if _CONTRACT[0]:
                return V(self._same(a, b, t), _fma(b.a, t.a, a.a * (F32(1.0) - t.a)))
            return V(self._same(a, b, t), a.a * (F32(1.0) - t.a) + b.a * t.a)

    def bi_smoothstep(self, m, e0, e1, x):
        with np.errstate(all="ignore"):
            t = np.fmin(np.fmax((x.a - e0.a) / (e1.a - e0.a), F32(0.0)), F32(1.0))
            return V(self._same(e0, e1, x), t * t * (F32(3.0) - F32(2.0) * t))

    def bi_floatBitsToUint(self, m, v):
        return V("uint" if v.t == "float" else "uvec" + v.t[-1], np.ascontiguousarray(v.a).view(np.uint32))

    def _lanes(self, mask):
        return np.ones(self.n, dtype=bool) if mask is None else mask

    def bi_texture(self, mask, s, coord):
        if s.a is None:
            raise ShaderError(f"{s.t} used but no texture unit was bound")
        c = coord.a if coord.a.ndim == 2 else coord.a[None, :]
        lanes = self._lanes(self.live(mask))
        c = np.broadcast_to(c, (c.shape[0], self.n))
        if getattr(s.a, "needs_quad", False):
            # Implicit LOD: the unit sees the coordinate every lane holds at this call (the interpreter evaluates expressions
            # for all lanes; masks only gate stores) and which lanes REACH the call: the lanes active here, or -- inside an
            # if / else of twin calls with merge_twin_calls -- the lanes that entered that if / else.
            reach = self.reach_stack[-1] & ~self.discarded if self.reach_stack else lanes
            with np.errstate(all="ignore"):
                r = np.asarray(s.a.texture_quad(np.where(np.isfinite(c), c, F32(0.0)).astype(F32), reach), dtype=F32)
            r = np.where(lanes, r, F32(0.0))
        else:
            r = np.asarray(s.a.texture(np.where(lanes, c, F32(0.0)).astype(F32)), dtype=F32)
        z = np.zeros_like(r)
        return V("vec4", np.stack([r, z, z, z + F32(1.0)]))

    def bi_texelFetch(self, mask, s, coord, lod):
        lanes = self._lanes(self.live(mask))
        c = np.broadcast_to(coord.a, (coord.a.shape[0], self.n))
        r = np.asarray(s.a.texel_fetch(np.where(lanes, c, 0), self.uniform_int(lod)), dtype=F32)
        z = np.zeros_like(r)
        return V("vec4", np.stack([r, z, z, z + F32(1.0)]))


def load(path: str, defines: dict | None = None, force_defines: dict | None = None) -> Parser:
    tokens, _ = preprocess(path, defines, force_defines)
    return Parser(tokens).parse()
