"""Executes the raster passes of the reference's Blit.shader from their source text.

TEST INFRASTRUCTURE ONLY (like oracle/hlsl_interp.py, which it extends).  Purpose: pin the
composite and the de-tile debug view against the reference's own
ShaderLab/Cg text instead of only against a restatement.  /root/reference/Assets/MiniEngineAO/
Shaders/Blit.shader is read at generation time (tests/golden/make_reference_goldens.py) and never
copied into this repository.

Taken from the source text: the pass list, each pass's `Blend` statement, its fragment program
(declarations, struct outputs, every expression).  Supplied here, because a GPU's raster pipeline
would: a full-screen primitive puts one fragment at every destination pixel centre with
uv = ((x + 0.5) / W, (y + 0.5) / H) (what vert_procedural / vert_img2 interpolate to); tex2D on a
point-filtered texture returns the texel containing uv; the output merger evaluates
dst * DstFactor (+ src * SrcFactor when SrcFactor is not Zero) in f32 and rounds once to the
target format (f16 round-to-nearest-even, UNORM8 like the AO stores) -- DESIGN.md section 8.
"""
from __future__ import annotations

import re

import numpy as np

from . import hlsl_interp as HI

F = np.float32


# ------------------------------------------------------------------------------------------------
# ShaderLab text -> passes

def _matching_brace(text, open_at):
    depth = 0
    for i in range(open_at, len(text)):
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return i
    raise SyntaxError("unbalanced braces")


def passes(shader_text):
    """[{blend: str | None, name: str | None, cg: str}] in SubShader order."""
    sub = shader_text.index("SubShader")
    out, pos = [], sub
    for m in re.finditer(r"\bPass\s*\{", shader_text[sub:]):
        start = sub + m.end() - 1
        if start < pos:
            continue
        end = _matching_brace(shader_text, start)
        body = shader_text[start + 1:end]
        pos = end
        cg = re.search(r"CGPROGRAM(.*?)ENDCG", body, re.S)
        blend = re.search(r"^\s*Blend\s+(.+?)\s*$", body, re.M)
        name = re.search(r'^\s*Name\s+"(\w+)"', body, re.M)
        out.append({"blend": blend.group(1) if blend else None, "name": name.group(1) if name else None,
                    "cg": cg.group(1) if cg else ""})
    return out


_PRELUDE = "struct v2f_img { float4 pos; float2 uv; };\n"


def _to_hlsl(cg):
    src = re.sub(r"^\s*#pragma.*$", "", cg, flags=re.M)
    src = re.sub(r"\bsampler2D(_float)?\s+(\w+)\s*;", r"Texture2D \2;", src)
    src = re.sub(r"UNITY_DECLARE_TEX2DARRAY\s*\(\s*(\w+)\s*\)\s*;", r"Texture2DArray \1;", src)
    src = re.sub(r":\s*SV_Target\d*", "", src)        # output semantics carry no arithmetic
    return _PRELUDE + src


# ------------------------------------------------------------------------------------------------
# the HLSL interpreter + structs, tex2D, frac, floor

class _Parser(HI.Parser):
    def __init__(self, toks):
        super().__init__(toks)
        self.structs = {}

    def is_type(self, s):
        return s in self.structs or super().is_type(s)

    def program(self):
        # structs are split off first; the rest is the base grammar
        toks, keep, i = self.t, [], 0
        while i < len(toks):
            if toks[i][1] == "struct":
                name = toks[i + 1][1]
                assert toks[i + 2][1] == "{"
                j, fields = i + 3, []
                while toks[j][1] != "}":
                    fields.append((toks[j][1], toks[j + 1][1]))
                    j += 2
                    while toks[j][1] != ";":
                        j += 1
                    j += 1
                self.structs[name] = fields
                i = j + 2                      # } ;
            else:
                keep.append(toks[i])
                i += 1
        self.t, self.i = keep, 0
        prog = super().program()
        prog.structs = self.structs
        return prog


class _Machine(HI.Machine):
    """Values of struct type are ('struct', {field: value})."""

    def __init__(self, prog, textures):
        super().__init__(prog)
        self.textures = textures           # name -> (array [slices?, h, w] of f32 components, is_array)

    def _zero(self, typ):
        if typ in self.p.structs:
            return ("struct", {f: self._zero(t) for t, f in self.p.structs[typ]})
        kind, n = HI.parse_type(typ)
        return (kind, [HI.wrap(kind, 0)] * n)

    def exec(self, s, env):
        if s[0] == "decl" and s[1] in self.p.structs:
            env[-1][s[2]] = self._zero(s[1]) if s[3] is None else self.eval(s[3], env)
            return
        super().exec(s, env)

    def eval(self, n, env):
        if n[0] == "member":
            base = self.eval(n[1], env)
            if base[0] == "struct":
                return base[1][n[2]]
        return super().eval(n, env)

    def assign(self, target, val, env):
        if target[0] == "member":
            base = self.eval(target[1], env)
            if base[0] == "struct":
                old = base[1][target[2]]
                v = HI.convert(val, old[0])
                if len(v[1]) == 1 and len(old[1]) > 1:
                    v = (v[0], v[1] * len(old[1]))
                base[1][target[2]] = (v[0], list(v[1][:len(old[1])]))
                return
        super().assign(target, val, env)

    def _texel(self, name, uv, slice_index=0):
        tex = self.textures[name]
        h, w = tex.shape[-2], tex.shape[-1]
        x = min(max(int(np.floor(F(uv[0]) * F(w))), 0), w - 1)       # point filter: the texel containing uv
        y = min(max(int(np.floor(F(uv[1]) * F(h))), 0), h - 1)
        v = tex[slice_index, y, x] if tex.ndim == 3 else tex[y, x]
        return F(v)

    def call(self, name, args, env):
        if name in ("tex2D", "SAMPLE_DEPTH_TEXTURE"):
            uv = HI.convert(self.eval(args[1], env), "f")[1]
            r = self._texel(args[0][1], uv)
            return ("f", [r]) if name == "SAMPLE_DEPTH_TEXTURE" else ("f", [r, F(0), F(0), F(1)])   # single-channel textures
        if name == "UNITY_SAMPLE_TEX2DARRAY":
            uvw = HI.convert(self.eval(args[1], env), "f")[1]
            r = self._texel(args[0][1], uvw, int(uvw[2]))
            return ("f", [r, F(0), F(0), F(1)])
        if name in ("frac", "floor"):
            v = HI.convert(self.eval(args[0], env), "f")
            fl = [F(np.floor(c)) for c in v[1]]
            return ("f", fl if name == "floor" else [F(c - f) for c, f in zip(v[1], fl)])
        return super().call(name, args, env)

    def invoke(self, fn, argvals):
        scope = {}
        for (ptype, pname), v in zip(fn.params, argvals):
            if ptype in self.p.structs:
                scope[pname] = v
            else:
                kind, n = HI.parse_type(ptype)
                v = HI.convert(v, kind)
                if len(v[1]) == 1 and n > 1:
                    v = (kind, v[1] * n)
                scope[pname] = (kind, list(v[1][:n]))
        try:
            self.run(fn.body, [scope])
        except HI.Return as r:
            if r.value is None:
                return None
            if fn.ret in self.p.structs:
                return r.value
            kind, n = HI.parse_type(fn.ret)
            v = HI.convert(r.value, kind)
            if len(v[1]) == 1 and n > 1:
                v = (kind, v[1] * n)                       # `return tex2D(...).r;` from a float4 function
            return (kind, list(v[1][:n]))
        return None


def fragment_outputs(pass_info, textures, width, height):
    """Runs the pass's `frag` at every pixel centre.  Returns {target name or 'SV_Target': [h, w, 4] f32}."""
    prog = _Parser(HI.lex(_to_hlsl(pass_info["cg"]))).program()
    m = _Machine(prog, textures)
    fn = prog.funcs["frag"]
    out = {}
    for y in range(height):
        for x in range(width):
            frag_in = ("struct", {"pos": ("f", [F(0)] * 4),
                                  "uv": ("f", [F((x + 0.5) / width), F((y + 0.5) / height)])})
            res = m.invoke(fn, [frag_in])
            items = res[1].items() if res[0] == "struct" else [("SV_Target", res)]
            for key, val in items:
                out.setdefault(key, np.zeros((height, width, 4), np.float32))[y, x] = val[1]
    return out


# ------------------------------------------------------------------------------------------------
# output merger

def _factor(name, src):
    """Blend factor as a 4-vector (rgb from the colour form, a from the alpha form is handled by the caller)."""
    one = np.ones(4, np.float32)
    if name == "Zero":
        return None
    if name == "One":
        return one
    if name == "SrcAlpha":
        return np.full(4, src[3], np.float32)
    if name == "OneMinusSrcAlpha":
        return np.full(4, F(1) - src[3], np.float32)
    if name == "SrcColor":
        return src.astype(np.float32)
    if name == "OneMinusSrcColor":
        return (one - src).astype(np.float32)
    raise ValueError("blend factor %s" % name)


def blend(blend_statement, src, dst):
    """dst' per the ShaderLab `Blend SrcFactor DstFactor [, SrcFactorA DstFactorA]` statement; f32 in, f32 out
    (one rounding per multiply; a Zero source factor means the source term is not evaluated)."""
    if blend_statement is None:
        return src.astype(np.float32)
    parts = [p.split() for p in blend_statement.split(",")]
    rgb = parts[0]
    alpha = parts[1] if len(parts) > 1 else parts[0]
    out = np.zeros(4, np.float32)
    for chans, (sf, df) in (((0, 1, 2), rgb), ((3,), alpha)):
        fs, fd = _factor(sf, src), _factor(df, src)
        for c in chans:
            term_d = F(0) if fd is None else F(dst[c] * fd[c])
            out[c] = term_d if fs is None else F(F(src[c] * fs[c]) + term_d)
    return out
