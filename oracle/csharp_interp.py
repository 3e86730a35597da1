"""A small interpreter for the C# subset used by the host side of the reference's SSAO path.

TEST INFRASTRUCTURE ONLY.  Purpose: execute the *reference's own C# source text*
(/root/reference/Assets/MiniEngineAO/AmbientOcclusion.cs, read at generation time, never copied
into this repository) for everything that feeds the compute dispatches: the RTHandle buffer
table and ceil-div sizing, CalculateZBufferParams, CalculateTanHalfFovHeight, the SampleThickness
table, PushRenderCommands and PushUpsampleCommands (constant blocks, kernel choice, dispatch
sizes) and the wiring in RebuildCommandBuffers.  Unity itself is replaced by recording mocks
(Camera, SystemInfo, CommandBuffer, ComputeShader, Shader, Mathf, Vector2/4) -- see
tests/golden/make_reference_goldens.py -- so what comes out is the list of dispatches the
reference would record, with the constants it would upload.

Numerics: C# `float` arithmetic is evaluated in binary32 with one rounding per operation
(numpy.float32), int arithmetic with C# truncating division; Mathf.Sqrt / Mathf.Pow go through
double and round once, like UnityEngine.Mathf.  (Old Mono may keep float temporaries in double
precision -- unverifiable here, noted in DESIGN.md.)

Supported subset: classes with fields / auto-less properties (get/set bodies) / methods /
constructors, enums, static members, var and typed locals, const, if/else, for, foreach, switch
with case/default/break, return, compound assignment, ++/--, casts, ternary, C operator set,
`new T(...)`, `new T[n]`, `new T[n][]`, hex literals, byte/ushort casts, array initialisers, indexers `a[i]` and `m[i, j]`, `out` arguments.
"""
from __future__ import annotations

import math
import re

import numpy as np

F = np.float32

TOKEN = re.compile(r"""
    (?P<num>0[xX][0-9a-fA-F]+|(?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?[fFdD]?|\d+[eE][+-]?\d+[fFdD]?|\d+[fFuU]?)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<id>[A-Za-z_]\w*)
  | (?P<op>\+\+|--|<<=|>>=|\+=|-=|\*=|/=|%=|\|=|&=|<<|>>|<=|>=|==|!=|&&|\|\||[-+*/%<>=!&|^~?:;,.(){}\[\]])
  | (?P<ws>\s+)
""", re.X)

MODIFIERS = {"public", "private", "protected", "internal", "static", "readonly", "sealed", "const", "override", "virtual"}
CAST_TYPES = ("int", "uint", "float", "double", "byte", "ushort", "short", "long")
INT_TYPES = ("int", "uint", "byte", "ushort", "short", "long")
PRIMITIVES = {"int", "uint", "float", "double", "bool", "string", "void", "var", "object"}


def strip_noise(text):
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#(region|endregion|if|endif|else|define)[^\n]*$", "", text, flags=re.M)
    return text


def lex(src):
    toks, i = [], 0
    while i < len(src):
        m = TOKEN.match(src, i)
        if not m:
            raise SyntaxError("unexpected character %r" % src[i:i + 30])
        i = m.end()
        if m.lastgroup != "ws":
            toks.append((m.lastgroup, m.group(m.lastgroup)))
    toks.append(("eof", ""))
    return toks


class ClassDef:
    def __init__(self, name):
        self.name = name
        self.fields = {}        # name -> (is_static, init expr or None, declared type)
        self.props = {}         # name -> (is_static, getter body, setter body)
        self.methods = {}       # name -> (is_static, params [(is_out, name)], body)
        self.ctor = None        # (params, body)
        self.enums = {}         # nested enums: name -> {member: int}
        self.classes = {}       # nested classes


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[min(self.i + k, len(self.t) - 1)]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, v):
        if self.peek()[0] != "eof" and self.peek()[1] == v and self.peek()[0] != "str":
            self.i += 1
            return True
        return False

    def expect(self, v):
        if not self.accept(v):
            raise SyntaxError("expected %r, got %r near token %d" % (v, self.peek()[1], self.i))

    def skip_attributes(self):
        while self.peek()[1] == "[" and self.peek()[0] == "op":
            depth = 0
            while True:
                tok = self.next()[1]
                if tok == "[":
                    depth += 1
                elif tok == "]":
                    depth -= 1
                    if depth == 0:
                        break

    def type_name(self):
        """identifier(.identifier)* (<...>)? ([])*  -> string (only its spelling matters)"""
        name = self.next()[1]
        while self.peek()[1] == "." and self.peek(1)[0] == "id":
            self.next()
            name += "." + self.next()[1]
        if self.accept("<"):
            depth = 1
            while depth:
                tok = self.next()[1]
                depth += tok == "<"
                depth -= tok == ">"
        while self.peek()[1] == "[" and self.peek(1)[1] == "]":
            self.next()
            self.next()
            name += "[]"
        return name

    # ---- declarations
    def compilation_unit(self):
        classes = {}
        while self.peek()[0] != "eof":
            tok = self.peek()[1]
            if tok == "using":
                while self.next()[1] != ";":
                    pass
            elif tok == "namespace":
                self.next()
                self.type_name()
                self.expect("{")
                while not self.accept("}"):
                    c = self.member_container()
                    if c:
                        classes[c.name] = c
            else:
                c = self.member_container()
                if c:
                    classes[c.name] = c
        return classes

    def member_container(self):
        self.skip_attributes()
        while self.peek()[1] in MODIFIERS:
            self.next()
        if self.accept("class"):
            return self.class_body(self.next()[1])
        raise SyntaxError("expected class, got %r" % self.peek()[1])

    def class_body(self, name):
        c = ClassDef(name)
        if self.accept(":"):
            self.type_name()
            while self.accept(","):
                self.type_name()
        self.expect("{")
        while not self.accept("}"):
            self.skip_attributes()
            mods = set()
            while self.peek()[1] in MODIFIERS:
                mods.add(self.next()[1])
            static = "static" in mods or "const" in mods
            if self.accept("enum"):
                ename = self.next()[1]
                self.expect("{")
                members, val = {}, 0
                while not self.accept("}"):
                    m = self.next()[1]
                    if self.accept("="):
                        val = int(self.next()[1])
                    members[m] = val
                    val += 1
                    self.accept(",")
                c.enums[ename] = members
                continue
            if self.accept("class"):
                nested = self.class_body(self.next()[1])
                c.classes[nested.name] = nested
                continue
            # constructor: Name(
            if self.peek()[1] == name and self.peek(1)[1] == "(":
                self.next()
                params = self.params()
                c.ctor = (params, self.block())
                continue
            ftype = self.type_name()
            mname = self.next()[1]
            if self.peek()[1] == "<":                 # generic method: skip <T>
                self.type_name_generic_tail()
            if self.peek()[1] == "(":
                params = self.params()
                while self.peek()[1] == "where":      # generic constraint
                    while self.peek()[1] != "{":
                        self.next()
                c.methods[mname] = (static, params, self.block())
            elif self.peek()[1] == "{":               # property
                self.next()
                getter = setter = None
                while not self.accept("}"):
                    acc = self.next()[1]
                    body = self.block()
                    if acc == "get":
                        getter = body
                    else:
                        setter = body
                c.props[mname] = (static, getter, setter)
            else:                                     # field(s)
                while True:
                    init = None
                    if self.accept("="):
                        init = self.array_init() if self.peek()[1] == "{" else self.expr()
                    c.fields[mname] = (static, init, ftype)
                    if self.accept(","):
                        mname = self.next()[1]
                        continue
                    break
                self.expect(";")
        return c

    def type_name_generic_tail(self):
        self.expect("<")
        depth = 1
        while depth:
            tok = self.next()[1]
            depth += tok == "<"
            depth -= tok == ">"

    def params(self):
        self.expect("(")
        out = []
        while not self.accept(")"):
            is_out = False
            while self.peek()[1] in ("out", "ref", "this", "params"):
                is_out = self.next()[1] in ("out", "ref") or is_out
            ptype = self.type_name()
            out.append((is_out, self.next()[1], ptype))
            self.accept(",")
        return out

    def array_init(self):
        self.expect("{")
        items = []
        while not self.accept("}"):
            items.append(self.expr())
            self.accept(",")
        return ("arrayinit", items)

    # ---- statements
    def block(self):
        self.expect("{")
        out = []
        while not self.accept("}"):
            out.append(self.statement())
        return ("block", out)

    def looks_like_decl(self):
        """type ident [=;,]   (type may be dotted / generic / array)"""
        save = self.i
        try:
            if self.peek()[0] != "id":
                return False
            self.type_name()
            ok = self.peek()[0] == "id" and self.peek(1)[1] in ("=", ";", ",") or \
                (self.peek()[0] == "id" and self.peek(1)[1] == "in")
            return ok
        except Exception:
            return False
        finally:
            self.i = save

    def statement(self):
        tok = self.peek()[1]
        if tok == "{":
            return self.block()
        if tok == "if":
            self.next()
            self.expect("(")
            c = self.expr()
            self.expect(")")
            t = self.statement()
            e = self.statement() if self.accept("else") else None
            return ("if", c, t, e)
        if tok == "for":
            self.next()
            self.expect("(")
            init = self.statement()                  # consumes ';'
            cond = self.expr()
            self.expect(";")
            step = self.expr()
            self.expect(")")
            return ("for", init, cond, step, self.statement())
        if tok == "foreach":
            self.next()
            self.expect("(")
            self.type_name()
            var = self.next()[1]
            self.expect("in")
            seq = self.expr()
            self.expect(")")
            return ("foreach", var, seq, self.statement())
        if tok == "switch":
            self.next()
            self.expect("(")
            subj = self.expr()
            self.expect(")")
            self.expect("{")
            cases = []                                # (value expr or None, [stmts])
            while not self.accept("}"):
                if self.accept("default"):
                    val = None
                else:
                    self.expect("case")
                    val = self.expr()
                self.expect(":")
                body = []
                while self.peek()[1] not in ("case", "default", "}"):
                    body.append(self.statement())
                cases.append((val, body))
            return ("switch", subj, cases)
        if tok == "return":
            self.next()
            v = None if self.peek()[1] == ";" else self.expr()
            self.expect(";")
            return ("return", v)
        if tok == "break":
            self.next()
            self.expect(";")
            return ("break",)
        if tok == "const":
            self.next()
        if self.looks_like_decl():
            dtype = self.type_name()
            decls = []
            while True:
                name = self.next()[1]
                init = None
                if self.accept("="):
                    init = self.array_init() if self.peek()[1] == "{" else self.expr()
                decls.append((name, init))
                if not self.accept(","):
                    break
            self.expect(";")
            return ("decl", decls, dtype)
        e = self.expr()
        self.expect(";")
        return ("expr", e)

    # ---- expressions
    def expr(self):
        lhs = self.ternary()
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("=", "+=", "-=", "*=", "/=", "%=", "<<=", ">>=", "|=", "&="):
            self.next()
            rhs = self.expr()
            if tok[1] != "=":
                rhs = ("bin", tok[1][:-1], lhs, rhs)
            return ("assign", lhs, rhs)
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.expr()
            self.expect(":")
            b = self.expr()
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
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+", "!", "~"):
            self.next()
            return ("un", tok[1], self.unary())
        # cast: ( type ) unary      -- only primitive casts occur in the subset
        if tok[1] == "(" and self.peek(1)[1] in CAST_TYPES and self.peek(2)[1] == ")":
            self.next()
            typ = self.next()[1]
            self.next()
            return ("cast", typ, self.unary())
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.accept("["):
                idx = [self.expr()]
                while self.accept(","):
                    idx.append(self.expr())
                self.expect("]")
                e = ("index", e, idx)
            elif self.peek()[1] == "." and self.peek()[0] == "op":
                self.next()
                name = self.next()[1]
                if self.peek()[1] == "<" and self.peek(2)[1] == ">" and self.peek(3)[1] == "(":   # Foo<T>(
                    self.type_name_generic_tail()
                if self.peek()[1] == "(":
                    e = ("mcall", e, name, self.args())
                else:
                    e = ("member", e, name)
            elif self.peek()[1] in ("++", "--") and self.peek()[0] == "op":
                op = self.next()[1]
                e = ("assign", e, ("bin", op[0], e, ("lit", 1)))
            else:
                return e

    def args(self):
        self.expect("(")
        out = []
        while not self.accept(")"):
            is_out = False
            if self.peek()[1] in ("out", "ref"):
                self.next()
                is_out = True
            out.append((is_out, self.expr()))
            self.accept(",")
        return out

    def primary(self):
        kind, val = self.next()
        if kind == "num":
            if val[:2] in ("0x", "0X"):
                return ("lit", int(val, 16))
            if re.search(r"[.eE]", val) or val[-1] in "fFdD":
                return ("lit", float(val.rstrip("fFdD")) if val[-1] in "dD" else F(float(val.rstrip("fFdD"))))
            return ("lit", int(val.rstrip("uU")))
        if kind == "str":
            return ("lit", val[1:-1])
        if kind == "id":
            if val == "new":
                typ = self.type_name_no_array()
                if self.accept("["):
                    n = self.expr()
                    self.expect("]")
                    while self.peek()[1] == "[" and self.peek(1)[1] == "]":   # jagged: new T[n][]
                        self.next()
                        self.next()
                        typ += "[]"
                    return ("newarray", typ, n)
                if self.peek()[1] == "(":
                    return ("new", typ, self.args())
                raise SyntaxError("unsupported new-expression")
            if val in ("true", "false"):
                return ("lit", val == "true")
            if val == "null":
                return ("lit", None)
            if self.peek()[1] == "<" and self.peek(1)[0] == "id" and self.peek(2)[1] == ">" and self.peek(3)[1] == "(":
                self.type_name_generic_tail()             # Foo<T>(...)
            if self.peek()[1] == "(":
                return ("call", val, self.args())
            return ("var", val)
        if val == "(":
            e = self.expr()
            self.expect(")")
            return e
        raise SyntaxError("unexpected token %r" % val)

    def type_name_no_array(self):
        name = self.next()[1]
        while self.peek()[1] == "." and self.peek(1)[0] == "id":
            self.next()
            name += "." + self.next()[1]
        return name


class EnumValue(int):
    """An enum member: behaves as its integer for (int) casts and comparisons."""


def coerce(value, typ):
    """Implicit conversion to a declared C# type (only the numeric primitives matter here)."""
    if value is None or isinstance(value, (bool, str)):
        return value
    if typ.endswith("[]") and isinstance(value, list):
        if typ[:-2] in ("float", "double"):
            value[:] = [coerce(v, typ[:-2]) for v in value]
        return value
    if typ == "float" and isinstance(value, (int, float, np.floating)):
        return F(value)
    if typ == "double" and isinstance(value, (int, float, np.floating)):
        return float(value)
    if typ in ("int", "uint") and isinstance(value, (int, np.integer)) and not isinstance(value, EnumValue):
        return int(value)
    return value


class Return(Exception):
    def __init__(self, v):
        self.v = v


class Break(Exception):
    pass


class Instance:
    def __init__(self, cls):
        self.cls = cls
        self.f = {}


class Ref:
    """An lvalue handed to an `out` parameter of a native (mock) method."""

    def __init__(self, setter):
        self.set = setter


def mathf_sqrt(v):
    return F(math.sqrt(float(v)))


def mathf_pow(a, b):
    return F(math.pow(float(a), float(b)))


class Interp:
    """globals_: name -> Python object for everything Unity provides (mocks, Mathf, enums...)."""

    def __init__(self, classes, globals_):
        self.classes = classes
        self.g = dict(globals_)
        self.statics = {}                       # (class name, field) -> value

    # ---- helpers
    def find_class(self, name, within=None):
        if within is not None and name in within.classes:
            return within.classes[name]
        for c in self.classes.values():
            if c.name == name:
                return c
            if name in c.classes:
                return c.classes[name]
        return None

    def owner_of(self, cls):
        for c in self.classes.values():
            if cls.name in c.classes and c.classes[cls.name] is cls:
                return c
        return None

    def enum_lookup(self, name, cls):
        for c in filter(None, (cls, self.owner_of(cls) if cls else None, *self.classes.values())):
            if name in c.enums:
                return c.enums[name]
        return None

    def new_instance(self, cls, args):
        inst = Instance(cls)
        for fname, (static, init, ftype) in cls.fields.items():
            if not static:
                zero = F(0) if ftype == "float" else (0 if ftype in ("int", "uint") else (False if ftype == "bool" else None))
                inst.f[fname] = coerce(self.eval(init, [{}], inst), ftype) if init is not None else zero
        if cls.ctor:
            params, body = cls.ctor
            scope = {p[1]: coerce(a, p[2]) for p, a in zip(params, args)}
            try:
                self.exec(body, [scope], inst)
            except Return:
                pass
        return inst

    def static_field(self, cls, name):
        key = (cls.name, name)
        if key not in self.statics:
            init, ftype = cls.fields[name][1], cls.fields[name][2]
            zero = F(0) if ftype == "float" else (0 if ftype in ("int", "uint") else None)
            v = self.eval(init, [{}], Instance(cls)) if init is not None else zero
            if isinstance(v, list) and ftype.startswith("float"):
                v = [F(x) for x in v]
            self.statics[key] = coerce(v, ftype)
        return self.statics[key]

    def call_method(self, inst, cls, name, args):
        static, params, body = cls.methods[name]
        if inst is None:
            inst = Instance(cls)                # static call: class context without instance fields
        scope = {}
        refs = []
        for (is_out, pname, ptype), a in zip(params, args):
            if is_out and isinstance(a, Ref):
                refs.append((pname, a))
                scope[pname] = None
            else:
                scope[pname] = coerce(a, ptype)
        try:
            self.exec(body, [scope], inst)
            ret = None
        except Return as r:
            ret = r.v
        for pname, ref in refs:
            ref.set(scope[pname])
        return ret

    # ---- arithmetic with C# numeric promotion
    @staticmethod
    def arith(op, a, b):
        if op in ("&&", "||"):
            return (bool(a) and bool(b)) if op == "&&" else (bool(a) or bool(b))
        if op in ("==", "!="):
            eq = (a is b) if (a is None or b is None) else (a == b)
            return bool(eq) if op == "==" else not bool(eq)
        if op in ("<", ">", "<=", ">="):
            return bool({"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op])
        if isinstance(a, bool) and isinstance(b, bool) and op in ("|", "&", "^"):
            return {"|": a or b, "&": a and b, "^": a != b}[op]
        is_f32 = isinstance(a, np.float32) or isinstance(b, np.float32)
        is_f64 = (isinstance(a, float) and not isinstance(a, np.floating)) or \
                 (isinstance(b, float) and not isinstance(b, np.floating))
        if is_f64:
            a, b = float(a), float(b)
            return {"+": a + b, "-": a - b, "*": a * b, "/": a / b}[op]
        if is_f32:
            a, b = F(a), F(b)
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                return F({"+": a + b, "-": a - b, "*": a * b, "/": a / b}[op])
        a, b = int(a), int(b)
        if op in ("/", "%"):
            q = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)
            return q if op == "/" else a - b * q
        if op == "<<":
            return a << b
        if op == ">>":
            return a >> b
        return {"+": a + b, "-": a - b, "*": a * b, "|": a | b, "&": a & b, "^": a ^ b}[op]

    # ---- evaluation
    def lookup(self, name, env, this):
        for scope in reversed(env):
            if name in scope:
                return scope[name]
        if this is not None:
            if name in this.f:
                return this.f[name]
            cls = this.cls
            for c in filter(None, (cls, self.owner_of(cls))):
                if name in c.fields and c.fields[name][0]:
                    return self.static_field(c, name)
                if name in c.props:
                    return self.get_prop(this if c is cls else None, c, name)
                if name in c.enums:
                    return ("enum", c.enums[name])
                if name in c.classes:
                    return ("class", c.classes[name])
        if name in self.g:
            return self.g[name]
        c = self.find_class(name)
        if c:
            return ("class", c)
        e = self.enum_lookup(name, this.cls if this else None)
        if e:
            return ("enum", e)
        raise NameError(name)

    def get_prop(self, inst, cls, name):
        static, getter, _ = cls.props[name]
        try:
            self.exec(getter, [{}], inst if inst is not None else Instance(cls))
        except Return as r:
            return r.v
        return None

    def member(self, obj, name):
        if isinstance(obj, tuple) and obj and obj[0] == "enum":
            return EnumValue(obj[1][name])
        if isinstance(obj, tuple) and obj and obj[0] == "class":
            cls = obj[1]
            if name in cls.fields:
                return self.static_field(cls, name)
            if name in cls.props:
                return self.get_prop(None, cls, name)
            if name in cls.enums:
                return ("enum", cls.enums[name])
            raise AttributeError(name)
        if isinstance(obj, Instance):
            if name in obj.f:
                return obj.f[name]
            if name in obj.cls.props:
                return self.get_prop(obj, obj.cls, name)
            if name in obj.cls.fields:
                return self.static_field(obj.cls, name)
            raise AttributeError("%s.%s" % (obj.cls.name, name))
        if isinstance(obj, (list, tuple)) and name == "Length":
            return len(obj)
        return getattr(obj, name)

    def eval(self, n, env, this):
        tag = n[0]
        if tag == "lit":
            return n[1]
        if tag == "var":
            return self.lookup(n[1], env, this)
        if tag == "un":
            v = self.eval(n[2], env, this)
            if n[1] == "-":
                return F(-v) if isinstance(v, np.float32) else -v
            if n[1] == "!":
                return not bool(v)
            return v
        if tag == "bin":
            a = self.eval(n[2], env, this)
            if n[1] == "&&" and not a:
                return False
            if n[1] == "||" and a:
                return True
            return self.arith(n[1], a, self.eval(n[3], env, this))
        if tag == "ternary":
            return self.eval(n[2] if self.eval(n[1], env, this) else n[3], env, this)
        if tag == "cast":
            v = self.eval(n[2], env, this)
            if n[1] in INT_TYPES:
                mask = {"byte": 0xFF, "ushort": 0xFFFF}.get(n[1])
                return int(v) & mask if mask else int(v)
            return F(v) if n[1] == "float" else float(v)
        if tag == "member":
            return self.member(self.eval(n[1], env, this), n[2])
        if tag == "index":
            base = self.eval(n[1], env, this)
            idx = [self.eval(i, env, this) for i in n[2]]
            return base[idx[0]] if len(idx) == 1 else base[tuple(idx)]
        if tag == "arrayinit":
            return [self.eval(i, env, this) for i in n[1]]
        if tag == "newarray":
            count = int(self.eval(n[2], env, this))
            zero = F(0) if n[1] == "float" else (0 if n[1] in INT_TYPES else None)
            return [zero] * count
        if tag == "new":
            args = [self.eval(a, env, this) for _, a in n[2]]
            cls = self.find_class(n[1].split(".")[-1], this.cls if this else None)
            if cls:
                return self.new_instance(cls, args)
            return self.g[n[1].split(".")[-1]](*args)
        if tag == "call":
            args = self.eval_args(n[2], env, this)
            cls = this.cls if this else None
            for c in filter(None, (cls, self.owner_of(cls) if cls else None)):
                if n[1] in c.methods:
                    return self.call_method(this if c is cls else None, c, n[1], args)
            return self.g[n[1]](*args)
        if tag == "mcall":
            obj = self.eval(n[1], env, this)
            args = self.eval_args(n[3], env, this)
            if isinstance(obj, Instance) and n[2] in obj.cls.methods:
                return self.call_method(obj, obj.cls, n[2], args)
            if isinstance(obj, tuple) and obj and obj[0] == "class" and n[2] in obj[1].methods:
                return self.call_method(None, obj[1], n[2], args)
            return getattr(obj, n[2])(*args)
        if tag == "assign":
            val = self.eval(n[2], env, this)
            self.assign(n[1], val, env, this)
            return val
        raise ValueError(tag)

    def eval_args(self, args, env, this):
        out = []
        for is_out, a in args:
            if is_out:
                out.append(Ref(lambda v, a=a: self.assign(a, v, env, this)))
            else:
                out.append(self.eval(a, env, this))
        return out

    def assign(self, target, val, env, this):
        tag = target[0]
        if tag == "var":
            name = target[1]
            for scope in reversed(env):
                if name in scope:
                    old = scope[name]
                    scope[name] = F(val) if isinstance(old, np.float32) and not isinstance(val, np.float32) else val
                    return
            if this is not None and name in this.f:
                this.f[name] = val
                return
            if this is not None:
                for c in filter(None, (this.cls, self.owner_of(this.cls))):
                    if name in c.fields and c.fields[name][0]:
                        self.static_field(c, name)
                        self.statics[(c.name, name)] = val
                        return
                    if name in c.props and c.props[name][2] is not None:
                        try:
                            self.exec(c.props[name][2], [{"value": val}], this)
                        except Return:
                            pass
                        return
            raise NameError(name)
        if tag == "index":
            base = self.eval(target[1], env, this)
            idx = [self.eval(i, env, this) for i in target[2]]
            old = base[idx[0]]
            base[idx[0]] = F(val) if isinstance(old, np.float32) else val
            return
        if tag == "member":
            obj = self.eval(target[1], env, this)
            if isinstance(obj, Instance):
                obj.f[target[2]] = val
            else:
                setattr(obj, target[2], val)
            return
        raise ValueError("bad assignment target")

    def exec(self, s, env, this):
        tag = s[0]
        if tag == "block":
            env = env + [{}]
            for st in s[1]:
                self.exec(st, env, this)
        elif tag == "expr":
            self.eval(s[1], env, this)
        elif tag == "decl":
            for name, init in s[1]:
                zero = F(0) if s[2] == "float" else (0 if s[2] in ("int", "uint") else None)
                env[-1][name] = coerce(self.eval(init, env, this), s[2]) if init is not None else zero
        elif tag == "if":
            if self.eval(s[1], env, this):
                self.exec(s[2], env, this)
            elif s[3] is not None:
                self.exec(s[3], env, this)
        elif tag == "for":
            env = env + [{}]
            self.exec(s[1], env, this)
            while self.eval(s[2], env, this):
                self.exec(s[4], env, this)
                self.eval(s[3], env, this)
        elif tag == "foreach":
            for v in list(self.eval(s[2], env, this)):
                self.exec(s[3], env + [{s[1]: v}], this)
        elif tag == "switch":
            v = self.eval(s[1], env, this)
            run = False
            try:
                for val, body in s[2]:
                    if not run and (val is None or self.eval(val, env, this) == v):
                        run = True
                    if run:
                        for st in body:
                            self.exec(st, env, this)
            except Break:
                pass
        elif tag == "return":
            raise Return(None if s[1] is None else self.eval(s[1], env, this))
        elif tag == "break":
            raise Break()
        else:
            raise ValueError(tag)


def load(path):
    with open(path) as f:
        return Parser(lex(strip_noise(f.read()))).compilation_unit()
