"""A small interpreter for the HLSL compute subset the reference's SSAO shaders are written in.

TEST INFRASTRUCTURE ONLY.  Purpose: execute the reference's *own shader source text*
(/root/reference/Assets/MiniEngineAO/Shaders/*.compute, read at generation time, never copied
into this repository) so that the oracle can be pinned against outputs of the reference's code
rather than only against restatements of it.  tests/golden/make_reference_goldens.py drives it the
way AmbientOcclusion.cs drives Unity (bind textures, set constants, dispatch) and commits the
resulting buffers as fixtures; /root/reference does not exist on the GPU box.

What is taken from the source text: every declaration, function, expression, index computation,
swizzle, branch, groupshared array, barrier and [numthreads].  What this file defines (the parts a
D3D driver/GPU would supply): the numerics contract of DESIGN.md (binary32 RNE, IEEE divide, an
Add/Sub with exactly one float Mul operand is one fused mad, lerp = mad(s, y-x, x), dot = mul +
mad chain, saturate/min/max with D3D NaN rules), resource semantics (out-of-range load = 0,
out-of-range store dropped, Gather on a point/clamp sampler with component order x=(0,1) y=(1,1)
z=(1,0) w=(0,0)) and the storage conversions of the render-target formats (shared with the oracle).

Supported subset: #pragma kernel variants, #if/#ifdef/#ifndef/#else/#endif, object-like #define,
CBUFFER_START/END, scalar/vector types of float/int/uint/bool, Texture2D / Texture2DArray /
RWTexture2D / RWTexture2DArray / SamplerState, groupshared arrays, functions, if/else, return,
compound assignment, the C operator set incl. shifts, bit ops, ternary, constructors, swizzles,
indexing, .Gather, and the intrinsics abs min max saturate clamp lerp dot.
"""
from __future__ import annotations

import ctypes
import re
from dataclasses import dataclass, field

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.fmaf.restype = ctypes.c_float
_libm.fmaf.argtypes = [ctypes.c_float] * 3
F = np.float32


def fmaf(a, b, c):
    return F(_libm.fmaf(float(a), float(b), float(c)))


# ------------------------------------------------------------------------------------------------
# preprocessing


def kernel_variants(text):
    """#pragma kernel NAME [MACRO[=VALUE]]...  ->  {name: {macro: value}}"""
    out = {}
    for m in re.finditer(r"^\s*#pragma\s+kernel\s+(\w+)(.*)$", text, re.M):
        defs = {}
        for tok in m.group(2).split():
            k, _, v = tok.partition("=")
            defs[k] = v if v else "1"
        out[m.group(1)] = defs
    return out


def preprocess(text, defines):
    defines = dict(defines)
    defines.setdefault("CBUFFER_END", "")
    out, stack = [], []           # stack of (taking, taken_before)
    for raw in text.split("\n"):
        line = re.sub(r"//.*", "", raw)
        s = line.strip()
        active = all(t for t, _ in stack)
        if s.startswith("#"):
            d = s[1:].split(None, 1)
            name, rest = d[0], (d[1].strip() if len(d) > 1 else "")
            if name in ("ifdef", "ifndef", "if"):
                if name == "if":
                    val = defines.get(rest, "0")
                    cond = val.strip() not in ("", "0")
                else:
                    cond = (rest in defines) == (name == "ifdef")
                stack.append((cond, cond))
            elif name == "else":
                taking, taken = stack.pop()
                stack.append((not taken, True))
            elif name == "endif":
                stack.pop()
            elif active and name == "define":
                k, _, v = rest.partition(" ")
                defines[k] = v.strip()
            # #pragma / #include: ignored
            continue
        if active:
            out.append(line)
    src = "\n".join(out)
    src = re.sub(r"CBUFFER_START\s*\(\s*\w+\s*\)", "", src)
    for _ in range(4):                       # object-like macros (may nest: TILE_DIM * TILE_DIM)
        for k, v in defines.items():
            src = re.sub(r"\b%s\b" % re.escape(k), v, src)
    return src


# ------------------------------------------------------------------------------------------------
# lexer / parser

TOKEN = re.compile(r"""
    (?P<num>(?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?[fF]?|\d+[eE][+-]?\d+[fF]?|0[xX][0-9a-fA-F]+|\d+[uU]?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op><<=|>>=|\+=|-=|\*=|/=|\|=|&=|<<|>>|<=|>=|==|!=|&&|\|\||[-+*/%<>=!&|^~?:;,.(){}\[\]])
  | (?P<ws>\s+)
""", re.X)

TYPE_RE = re.compile(r"^(float|int|uint|bool|half)([1-4])?$")
RESOURCE_TYPES = ("Texture2D", "Texture2DArray", "RWTexture2D", "RWTexture2DArray", "SamplerState")


def lex(src):
    toks, i = [], 0
    while i < len(src):
        m = TOKEN.match(src, i)
        if not m:
            raise SyntaxError("unexpected character %r" % src[i:i + 20])
        i = m.end()
        if m.lastgroup != "ws":
            toks.append((m.lastgroup, m.group(m.lastgroup)))
    toks.append(("eof", ""))
    return toks


@dataclass
class Func:
    name: str
    ret: str
    params: list          # (type, name)
    body: list
    numthreads: tuple = None


@dataclass
class Program:
    resources: dict = field(default_factory=dict)     # name -> declared type string
    uniforms: dict = field(default_factory=dict)      # name -> (type, array_len or None)
    shared: dict = field(default_factory=dict)        # name -> length
    funcs: dict = field(default_factory=dict)


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, val):
        if self.peek()[1] == val and self.peek()[0] != "eof":
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise SyntaxError("expected %r, got %r (token %d)" % (val, self.peek()[1], self.i))

    def is_type(self, s):
        return bool(TYPE_RE.match(s))

    # -------- top level
    def program(self):
        prog = Program()
        while self.peek()[0] != "eof":
            numthreads = None
            if self.accept("["):
                assert self.next()[1] == "numthreads"
                self.expect("(")
                dims = []
                while True:
                    dims.append(self.expr())
                    if not self.accept(","):
                        break
                self.expect(")")
                self.expect("]")
                numthreads = tuple(dims)
            tok = self.peek()[1]
            if tok in RESOURCE_TYPES:
                self.next()
                rtype = tok
                if self.accept("<"):
                    rtype += "<" + self.next()[1] + ">"
                    self.expect(">")
                prog.resources[self.next()[1]] = rtype
                self.expect(";")
            elif tok == "groupshared":
                self.next()
                self.next()                       # element type (float)
                name = self.next()[1]
                self.expect("[")
                prog.shared[name] = self.expr()
                self.expect("]")
                self.expect(";")
            elif tok in ("static", "const"):
                self.next()
            elif self.is_type(tok) or tok == "void":
                typ = self.next()[1]
                name = self.next()[1]
                if self.accept("("):
                    params = []
                    while not self.accept(")"):
                        while self.peek()[1] in ("in", "const"):
                            self.next()
                        pt, pn = self.next()[1], self.next()[1]
                        if self.accept(":"):
                            self.next()           # semantic
                        params.append((pt, pn))
                        self.accept(",")
                    body = self.block()
                    prog.funcs[name] = Func(name, typ, params, body, numthreads)
                else:
                    length = None
                    if self.accept("["):
                        length = self.expr()
                        self.expect("]")
                    prog.uniforms[name] = (typ, length)
                    self.expect(";")
            else:
                raise SyntaxError("unexpected top-level token %r" % tok)
        return prog

    # -------- statements
    def block(self):
        self.expect("{")
        out = []
        while not self.accept("}"):
            out.append(self.statement())
        return out

    def statement(self):
        tok = self.peek()[1]
        if tok == "{":
            return ("block", self.block())
        if tok == "if":
            self.next()
            self.expect("(")
            cond = self.expr()
            self.expect(")")
            then = self.statement()
            els = self.statement() if self.accept("else") else None
            return ("if", cond, then, els)
        if tok == "return":
            self.next()
            val = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", val)
        if tok == "const":
            self.next()
            tok = self.peek()[1]
        if self.is_type(tok) and self.peek(1)[0] == "id":
            typ = self.next()[1]
            name = self.next()[1]
            init = self.expr() if self.accept("=") else None
            self.expect(";")
            return ("decl", typ, name, init)
        e = self.expr()
        self.expect(";")
        return ("expr", e)

    # -------- expressions (C precedence)
    def expr(self):
        return self.assignment()

    def assignment(self):
        lhs = self.ternary()
        tok = self.peek()[1]
        if tok in ("=", "+=", "-=", "*=", "/=", "<<=", ">>=", "|=", "&="):
            self.next()
            rhs = self.assignment()
            if tok != "=":
                rhs = ("bin", tok[:-1], lhs, rhs)
            return ("assign", lhs, rhs)
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.assignment()
            self.expect(":")
            b = self.assignment()
            return ("ternary", c, a, b)
        return c

    LEVELS = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="),
              ("<<", ">>"), ("+", "-"), ("*", "/", "%")]

    def binary(self, level):
        if level == len(self.LEVELS):
            return self.unary()
        lhs = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[level]:
            op = self.next()[1]
            lhs = ("bin", op, lhs, self.binary(level + 1))
        return lhs

    def unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "+", "!", "~"):
            op = self.next()[1]
            return ("un", op, self.unary())
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.accept("["):
                idx = self.expr()
                self.expect("]")
                e = ("index", e, idx)
            elif self.accept("."):
                name = self.next()[1]
                if self.accept("("):
                    e = ("method", e, name, self.args())
                else:
                    e = ("member", e, name)
            else:
                return e

    def args(self):
        out = []
        while not self.accept(")"):
            out.append(self.assignment())
            self.accept(",")
        return out

    def primary(self):
        kind, val = self.next()
        if kind == "num":
            v = val.rstrip("fFuU")
            if re.match(r"^0[xX]", v):
                return ("lit", "i", int(v, 16))
            if re.match(r"^\d+$", v):
                return ("lit", "i", int(v, 8) if len(v) > 1 and v[0] == "0" else int(v))   # C octal
            return ("lit", "f", F(float(v)))
        if kind == "id":
            if self.peek()[1] == "(":
                self.next()
                return ("call", val, self.args())
            return ("var", val)
        if val == "(":
            e = self.expr()
            self.expect(")")
            return e
        raise SyntaxError("unexpected token %r" % val)


# ------------------------------------------------------------------------------------------------
# values: (kind, [components]); kind in 'f' 'i' 'u' 'b'

RANK = {"b": 0, "i": 1, "u": 2, "f": 3}
MASK = 0xFFFFFFFF


def wrap(kind, v):
    if kind == "f":
        return F(v)
    if kind == "u":
        return int(v) & MASK
    if kind == "i":
        v = int(v) & MASK
        return v - (1 << 32) if v & 0x80000000 else v
    return bool(v)


def convert(val, kind):
    k, comps = val
    if k == kind:
        return val
    if kind == "f":
        return ("f", [F(int(c)) if k != "f" else c for c in comps])
    if k == "f":                                    # float -> int: truncate
        return (kind, [wrap(kind, int(np.trunc(c))) if kind != "b" else bool(c != 0) for c in comps])
    return (kind, [wrap(kind, int(c)) for c in comps])


def broadcast(a, b):
    na, nb = len(a[1]), len(b[1])
    if na == nb:
        return a, b
    if na == 1:
        return (a[0], a[1] * nb), b
    if nb == 1:
        return a, (b[0], b[1] * na)
    n = min(na, nb)                                 # HLSL truncates the longer vector
    return (a[0], a[1][:n]), (b[0], b[1][:n])


def parse_type(t):
    m = TYPE_RE.match(t)
    base = {"float": "f", "half": "f", "int": "i", "uint": "u", "bool": "b"}[m.group(1)]
    return base, int(m.group(2) or 1)


SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


class Return(Exception):
    def __init__(self, value):
        self.value = value


class Texture:
    """A bound render texture.  load/store/gather convert through `codec` (format of the RT)."""

    def __init__(self, array, decode=None, encode=None):
        self.a = array                               # [slices][h][w] (slices = 1 for 2D)
        self.decode = decode or (lambda v: F(v))
        self.encode = encode or (lambda v: F(v))

    @property
    def dims(self):
        return self.a.shape[2], self.a.shape[1]

    def load(self, x, y, s=0):
        w, h = self.dims
        if 0 <= x < w and 0 <= y < h and 0 <= s < self.a.shape[0]:
            return self.decode(self.a[s, y, x])
        return F(0.0)                                # D3D: out-of-range resource reads return 0

    def store(self, x, y, s, v):
        w, h = self.dims
        if 0 <= x < w and 0 <= y < h and 0 <= s < self.a.shape[0]:
            self.a[s, y, x] = self.encode(v)         # out-of-range UAV writes are dropped

    def gather(self, u, v, s=0):
        w, h = self.dims
        px = F(F(u * F(w)) - F(0.5))
        py = F(F(v * F(h)) - F(0.5))
        i0, j0 = int(np.floor(px)), int(np.floor(py))
        cl = lambda a, n: min(max(a, 0), n - 1)      # noqa: E731  clamp sampler
        i0, i1, j0, j1 = cl(i0, w), cl(i0 + 1, w), cl(j0, h), cl(j0 + 1, h)
        t = lambda i, j: self.decode(self.a[s, j, i])    # noqa: E731
        return ("f", [t(i0, j1), t(i1, j1), t(i1, j0), t(i0, j0)])


class Machine:
    def __init__(self, prog: Program):
        self.p = prog
        self.bind = {}            # resource name -> Texture
        self.const = {}           # uniform name -> value or list of values
        self.lds = {}

    # ---- float arithmetic under the numerics contract
    def fbin(self, op, a, b):
        if op == "+":
            return F(a + b)
        if op == "-":
            return F(a - b)
        if op == "*":
            return F(a * b)
        if op == "/":
            with np.errstate(divide="ignore", invalid="ignore"):
                return F(a / b)
        raise ValueError(op)

    def mul_operands(self, node, env):
        """For a '*' node: its two evaluated operands and whether the product is a float product."""
        if node[0] == "bin" and node[1] == "*":
            x, y = self.eval(node[2], env), self.eval(node[3], env)
            return x, y, (x[0] == "f" or y[0] == "f")
        return None

    # ---- evaluation
    def eval(self, n, env):
        tag = n[0]
        if tag == "lit":
            return (n[1], [n[2]])
        if tag == "var":
            return self.lookup(n[1], env)
        if tag == "un":
            v = self.eval(n[2], env)
            if n[1] == "-":
                return (v[0], [wrap(v[0], -c) if v[0] != "f" else F(-c) for c in v[1]])
            if n[1] == "!":
                return ("b", [not bool(c) for c in v[1]])
            if n[1] == "+":
                return v
            return (v[0], [wrap(v[0], ~int(c)) for c in v[1]])
        if tag == "bin":
            return self.binop(n, env)
        if tag == "ternary":
            c = self.eval(n[1], env)
            a, b = self.eval(n[2], env), self.eval(n[3], env)
            kind = a[0] if RANK[a[0]] >= RANK[b[0]] else b[0]
            a, b = convert(a, kind), convert(b, kind)
            a, b = broadcast(a, b)
            cc = c[1] * len(a[1]) if len(c[1]) == 1 else c[1]
            return (kind, [x if bool(k) else y for k, x, y in zip(cc, a[1], b[1])])
        if tag == "member":
            v = self.eval(n[1], env)
            return (v[0], [v[1][SWZ[ch]] for ch in n[2]])
        if tag == "index":
            return self.index(n, env)
        if tag == "call":
            return self.call(n[1], n[2], env)
        if tag == "method":
            assert n[2] == "Gather"
            tex = self.bind[n[1][1]]
            uv = convert(self.eval(n[3][1], env), "f")[1]
            s = int(np.trunc(uv[2])) if len(uv) > 2 else 0
            return tex.gather(uv[0], uv[1], s)
        if tag == "assign":
            val = self.eval(n[2], env)
            self.assign(n[1], val, env)
            return val
        raise ValueError(tag)

    def lookup(self, name, env):
        for scope in reversed(env):
            if name in scope:
                return scope[name]
        if name in self.const:
            return self.const[name]
        raise NameError(name)

    def binop(self, n, env):
        op, ln, rn = n[1], n[2], n[3]
        if op in ("+", "-"):
            # mad contraction: an Add/Sub with exactly one float Mul operand is a single fused mad
            lp, rp = self.mul_operands(ln, env), self.mul_operands(rn, env)
            lm, rm = bool(lp and lp[2]), bool(rp and rp[2])
            if lm != rm:
                (a, b, _), other = (lp, rn) if lm else (rp, ln)
                c = (self.arith("*", *rp[:2]) if rp else self.eval(rn, env)) if lm else \
                    (self.arith("*", *lp[:2]) if lp else self.eval(ln, env))
                del other
                a, b, c = convert(a, "f"), convert(b, "f"), convert(c, "f")
                a, b = broadcast(a, b)
                a, c = broadcast(a, c)
                a, b = broadcast(a, b)
                out = []
                for x, y, z in zip(a[1], b[1], c[1]):
                    if op == "+":
                        out.append(fmaf(x, y, z))              # x*y + z
                    elif lm:
                        out.append(fmaf(x, y, F(-z)))          # x*y - z
                    else:
                        out.append(fmaf(F(-x), y, z))          # z - x*y
                return ("f", out)
            a = self.arith("*", *lp[:2]) if lp else self.eval(ln, env)
            b = self.arith("*", *rp[:2]) if rp else self.eval(rn, env)
            return self.arith(op, a, b)
        return self.arith(op, self.eval(ln, env), self.eval(rn, env))

    def arith(self, op, a, b):
        if op in ("&&", "||"):
            a, b = broadcast(convert(a, "b"), convert(b, "b"))
            f = (lambda x, y: x and y) if op == "&&" else (lambda x, y: x or y)
            return ("b", [f(x, y) for x, y in zip(a[1], b[1])])
        if op in ("<<", ">>"):
            b = convert(b, "u")
            a, b = broadcast(a, b)
            if op == "<<":
                return (a[0], [wrap(a[0], int(x) << (int(y) & 31)) for x, y in zip(a[1], b[1])])
            return (a[0], [wrap(a[0], int(x) >> (int(y) & 31)) for x, y in zip(a[1], b[1])])
        kind = a[0] if RANK[a[0]] >= RANK[b[0]] else b[0]
        if op in ("|", "&", "^") and kind == "b":
            a, b = broadcast(a, b)
            f = {"|": lambda x, y: x or y, "&": lambda x, y: x and y, "^": lambda x, y: x != y}[op]
            return ("b", [bool(f(x, y)) for x, y in zip(a[1], b[1])])
        if kind == "b":
            kind = "i"
        a, b = broadcast(convert(a, kind), convert(b, kind))
        if op in ("==", "!=", "<", ">", "<=", ">="):
            f = {"==": lambda x, y: x == y, "!=": lambda x, y: x != y, "<": lambda x, y: x < y,
                 ">": lambda x, y: x > y, "<=": lambda x, y: x <= y, ">=": lambda x, y: x >= y}[op]
            return ("b", [bool(f(x, y)) for x, y in zip(a[1], b[1])])
        if kind == "f":
            return ("f", [self.fbin(op, x, y) for x, y in zip(a[1], b[1])])

        def iop(x, y):
            x, y = int(x), int(y)
            if op == "+":
                return x + y
            if op == "-":
                return x - y
            if op == "*":
                return x * y
            if op in ("/", "%"):
                q = abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)     # C truncation
                return q if op == "/" else x - y * q
            return {"|": x | y, "&": x & y, "^": x ^ y}[op]
        return (kind, [wrap(kind, iop(x, y)) for x, y in zip(a[1], b[1])])

    def index(self, n, env):
        base, idx = n[1], self.eval(n[2], env)
        if base[0] == "var":
            name = base[1]
            if name in self.bind:                              # Texture[uintN]
                tex = self.bind[name]
                # loads use unsigned coordinates (Texture2D<float>[uint2]); negative ints wrap high
                c = [int(v) for v in idx[1]]
                return ("f", [tex.load(c[0], c[1], c[2] if len(c) > 2 else 0)])
            if name in self.lds:
                return ("f", [self.lds[name][int(idx[1][0]) & MASK]])
            if name in self.const and isinstance(self.const[name], list):
                return self.const[name][int(idx[1][0])]
        v = self.eval(base, env)
        return (v[0], [v[1][int(idx[1][0])]])

    def assign(self, target, val, env):
        tag = target[0]
        if tag == "var":
            name = target[1]
            for scope in reversed(env):
                if name in scope:
                    old = scope[name]
                    v = convert(val, old[0])
                    if len(v[1]) == 1 and len(old[1]) > 1:
                        v = (v[0], v[1] * len(old[1]))
                    scope[name] = (v[0], v[1][:len(old[1])])
                    return
            raise NameError(name)
        if tag == "index":
            name = target[1][1]
            idx = self.eval(target[2], env)
            fv = convert(val, "f")[1][0]
            if name in self.bind:
                c = [wrap("i", v) for v in idx[1]]             # RW indices behave as signed here:
                self.bind[name].store(c[0], c[1], c[2] if len(c) > 2 else 0, fv)   # negatives are dropped
                return
            if name in self.lds:
                self.lds[name][int(idx[1][0]) & MASK] = fv
                return
        raise ValueError("unsupported assignment target %r" % (target,))

    def call(self, name, args, env):
        m = TYPE_RE.match(name)
        if m:                                                   # constructor / cast
            kind, n = parse_type(name)
            comps = []
            for a in args:
                comps += convert(self.eval(a, env), kind)[1]
            if len(comps) == 1 and n > 1:
                comps = comps * n
            assert len(comps) == n, (name, comps)
            return (kind, comps)
        if name in self.p.funcs:
            return self.invoke(self.p.funcs[name], [self.eval(a, env) for a in args])
        vals = [self.eval(a, env) for a in args]
        if name == "abs":
            v = vals[0]
            return (v[0], [F(abs(c)) if v[0] == "f" else abs(c) for c in v[1]])
        if name in ("min", "max"):
            a, b = vals
            kind = a[0] if RANK[a[0]] >= RANK[b[0]] else b[0]
            a, b = broadcast(convert(a, kind), convert(b, kind))
            f = np.fmin if name == "min" else np.fmax          # D3D: return the non-NaN operand
            return (kind, [F(f(x, y)) if kind == "f" else (min(x, y) if name == "min" else max(x, y))
                           for x, y in zip(a[1], b[1])])
        if name == "saturate":
            v = convert(vals[0], "f")
            return ("f", [F(np.fmin(np.fmax(c, F(0)), F(1))) for c in v[1]])
        if name == "clamp":
            x, lo, hi = (convert(v, "f") for v in vals)
            x, lo = broadcast(x, lo)
            x, hi = broadcast(x, hi)
            x, lo = broadcast(x, lo)
            return ("f", [F(np.fmin(np.fmax(a, l), h)) for a, l, h in zip(x[1], lo[1], hi[1])])
        if name == "lerp":                                      # x + s*(y - x) as one mad
            x, y, s = (convert(v, "f") for v in vals)
            x, y = broadcast(x, y)
            x, s = broadcast(x, s)
            x, y = broadcast(x, y)
            return ("f", [fmaf(c, F(b - a), a) for a, b, c in zip(x[1], y[1], s[1])])
        if name == "dot":                                       # mul, then a mad chain
            a, b = broadcast(convert(vals[0], "f"), convert(vals[1], "f"))
            acc = F(a[1][0] * b[1][0])
            for x, y in zip(a[1][1:], b[1][1:]):
                acc = fmaf(x, y, acc)
            return ("f", [acc])
        raise NameError("unknown function %s" % name)

    def invoke(self, fn, argvals):
        scope = {}
        for (ptype, pname), v in zip(fn.params, argvals):
            kind, n = parse_type(ptype)
            v = convert(v, kind)
            if len(v[1]) == 1 and n > 1:
                v = (kind, v[1] * n)
            scope[pname] = (kind, list(v[1][:n]))
        try:
            self.run(fn.body, [scope])
        except Return as r:
            if fn.ret == "void" or r.value is None:
                return None
            kind, n = parse_type(fn.ret)
            return convert(r.value, kind)
        return None

    def run(self, stmts, env):
        for s in stmts:
            self.exec(s, env)

    def exec(self, s, env):
        tag = s[0]
        if tag == "expr":
            self.eval(s[1], env)
        elif tag == "decl":
            kind, n = parse_type(s[1])
            if s[3] is None:
                v = (kind, [wrap(kind, 0)] * n)
            else:
                v = convert(self.eval(s[3], env), kind)
                if len(v[1]) == 1 and n > 1:
                    v = (kind, v[1] * n)
                v = (kind, list(v[1][:n]))
            env[-1][s[2]] = v
        elif tag == "if":
            c = self.eval(s[1], env)
            if bool(c[1][0]):
                self.exec_scoped(s[2], env)
            elif s[3] is not None:
                self.exec_scoped(s[3], env)
        elif tag == "block":
            self.run(s[1], env + [{}])
        elif tag == "return":
            raise Return(None if s[1] is None else self.eval(s[1], env))
        else:
            raise ValueError(tag)

    def exec_scoped(self, s, env):
        if s[0] == "block":
            self.run(s[1], env + [{}])
        else:
            self.exec(s, env + [{}])

    # ---- dispatch
    def dispatch(self, entry, groups):
        """Run entry over groups = (gx, gy, gz).  Threads of a group advance phase by phase, a phase
        ending at each top-level GroupMemoryBarrierWithGroupSync(); locals persist across phases."""
        fn = self.p.funcs[entry]
        tx, ty, tz = (int(self.eval(d, [])[1][0]) for d in fn.numthreads)
        phases, cur = [], []
        for s in fn.body:
            if s[0] == "expr" and s[1][0] == "call" and s[1][1] == "GroupMemoryBarrierWithGroupSync":
                phases.append(cur)
                cur = []
            else:
                cur.append(s)
        phases.append(cur)
        gx, gy, gz = groups
        for gzz in range(gz):
            for gyy in range(gy):
                for gxx in range(gx):
                    self.lds = {n: [F(0.0)] * int(self.eval(l, [])[1][0]) for n, l in self.p.shared.items()}
                    threads = []
                    for lz in range(tz):
                        for ly in range(ty):
                            for lx in range(tx):
                                vals = {"Gid": ("u", [gxx, gyy, gzz]),
                                        "GI": ("u", [lz * tx * ty + ly * tx + lx]),
                                        "GTid": ("u", [lx, ly, lz]),
                                        "DTid": ("u", [gxx * tx + lx, gyy * ty + ly, gzz * tz + lz])}
                                scope = {}
                                for (ptype, pname), sem in zip(fn.params, fn.semantics):
                                    kind, n = parse_type(ptype)
                                    scope[pname] = (kind, list(vals[sem][1][:n]))
                                threads.append([scope])
                    for ph in phases:
                        for env in threads:
                            try:
                                self.run(ph, env)
                            except Return:
                                pass


SEMANTIC_TO_VALUE = {"SV_GroupID": "Gid", "SV_GroupIndex": "GI", "SV_GroupThreadID": "GTid",
                     "SV_DispatchThreadID": "DTid"}


def attach_semantics(prog, entry, preprocessed_src):
    """System-value semantics of the entry point's parameters, read from the source text."""
    sig = re.search(r"void\s+%s\s*\((.*?)\)" % re.escape(entry), preprocessed_src, re.S).group(1)
    prog.funcs[entry].semantics = [SEMANTIC_TO_VALUE[s] for s in re.findall(r":\s*(SV_\w+)", sig)]


def compile_kernel(source_text, kernel_name):
    """Preprocess + parse one #pragma kernel variant.  Returns (Program, entry function name)."""
    variants = kernel_variants(source_text)
    defines = dict(variants[kernel_name])
    src = preprocess(source_text, defines)
    # system-value semantics of the entry point parameters, read from the source text
    entry = defines.get("MAIN", kernel_name)
    prog = Parser(lex(src)).program()
    fn = prog.funcs[entry]
    sig = re.search(r"void\s+%s\s*\((.*?)\)" % re.escape(entry), src, re.S).group(1)
    sems = re.findall(r":\s*(SV_\w+)", sig)
    fn.semantics = [SEMANTIC_TO_VALUE[s] for s in sems]
    return prog, entry
