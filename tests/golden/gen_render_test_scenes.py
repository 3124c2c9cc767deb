#!/usr/bin/env python3
"""Extracts the scene corpus of the reference's render tests into tests/golden/render_test_scenes.json.

  python tests/golden/gen_render_test_scenes.py [--reference /root/reference]

The reference keeps ~130 scene definitions as Rust code (integration-tests/src/render_tests/{view,rescaler,tiles,
tiles_transitions,transition}.rs: `runner.update_scene(Component::View(ViewComponent { .. }))` + `runner.snapshot(pts)`); their
PNG snapshots live in an un-vendored submodule, so the *scenes* are what can travel.  This script evaluates those test
functions with a small interpreter for the Rust subset they use (struct literals with `..Default::default()`, enum paths,
closures, ranges + map/collect, helper fns, consts, `format!`) against a recording TestRunner, and writes every test as
  {"module", "name", "resolution": [w, h], "inputs": [{"id", "index", "width", "height", "kind"}], "mode",
   "steps": [{"update": <smelter-api JSON>} | {"snapshot_ms": t}]}
with the scene::Component tree converted to the JSON the HTTP API takes (smelter-api/src/video/component.rs), i.e. what
smr_renderer_update_scene / smr_scene_update consume.  Nothing is copied: the output is data derived by running the tests'
own scene-building code.  Tests whose components this library does not draw (Text / Image / Shader / WebView) are skipped
and listed.  tests/test_gpu_reference_scenes.py renders the corpus on the GPU against the oracle.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FILES = ["view", "rescaler", "tiles", "tiles_transitions", "transition", "simple", "text", "shader", "image"]

# ----------------------------------------------------------------------------------------------- tokens
TOK = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0x[0-9A-Fa-f_]+|\d[\d_]*(?:\.\d[\d_]*)?(?:[eE][+-]?\d+)?(?:_?(?:f32|f64|u8|u16|u32|u64|usize|i32|i64|isize))?)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*!?)
  | (?P<op>::|\.\.=|\.\.|=>|->|==|!=|<=|>=|&&|\|\||[{}()\[\],;:.=|&!?<>+\-*/#%])
""", re.X | re.S)


def tokenize(src):
    out, i = [], 0
    while i < len(src):
        m = TOK.match(src, i)
        if not m:
            raise SyntaxError(f"bad character {src[i]!r} at {i}")
        i = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group()))
    out.append(("eof", ""))
    return out


# ----------------------------------------------------------------------------------------------- AST (tuples)
class P:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def at(self, v, k=0):
        return self.t[self.i + k][1] == v

    def eat(self, v=None):
        tok = self.t[self.i]
        if v is not None and tok[1] != v:
            raise SyntaxError(f"expected {v!r}, got {tok[1]!r} near token {self.i}: {[x[1] for x in self.t[max(0, self.i - 8):self.i + 4]]}")
        self.i += 1
        return tok

    # ---- types are skipped
    def skip_type(self, stop):
        depth = 0
        while True:
            v = self.peek()[1]
            if depth == 0 and v in stop:
                return
            if v in "<([":
                depth += 1
            elif v in ">)]":
                depth -= 1
            elif v == "->" or v == "eof":
                if v == "eof":
                    raise SyntaxError("type runs to eof")
            self.i += 1

    def items(self):
        out = []
        while not self.at("") or self.peek()[0] != "eof":
            if self.peek()[0] == "eof":
                break
            if self.at("#"):  # attribute
                self.eat("#")
                if self.at("!"):
                    self.eat()
                self.eat("[")
                depth = 1
                while depth:
                    v = self.eat()[1]
                    depth += v == "["
                    depth -= v == "]"
                continue
            if self.at("pub"):
                self.eat()
                if self.at("("):
                    while not self.at(")"):
                        self.eat()
                    self.eat(")")
                continue
            if self.at("use") or self.at("mod"):
                while not self.at(";"):
                    self.eat()
                self.eat(";")
                continue
            if self.at("const") or self.at("static"):
                self.eat()
                name = self.eat()[1]
                self.eat(":")
                self.skip_type(("=",))
                self.eat("=")
                e = self.expr()
                self.eat(";")
                out.append(("const", name, e))
                continue
            if self.at("fn"):
                out.append(self.fn())
                continue
            if self.at("struct"):  # a plain data struct of the test file: its literals evaluate to dicts, nothing to record
                self.eat()
                self.eat()
                depth = 0
                while True:
                    v = self.eat()[1]
                    depth += v == "{"
                    depth -= v == "}"
                    if depth == 0 and v in ("}", ";"):
                        break
                continue
            if self.at("impl"):  # methods of such a struct: ("impl", Type, [fns])
                self.eat()
                ty = self.eat()[1]
                self.eat("{")
                fns = []
                while not self.at("}"):
                    if self.at("pub"):
                        self.eat()
                        continue
                    if self.at("#"):
                        self.eat("#")
                        self.eat("[")
                        while not self.at("]"):
                            self.eat()
                        self.eat("]")
                        continue
                    fns.append(self.fn())
                self.eat("}")
                out.append(("impl", ty, fns))
                continue
            raise SyntaxError(f"item? {self.peek()}")
        return out

    def fn(self):
        self.eat("fn")
        name = self.eat()[1]
        if self.at("<"):
            self.skip_type(("(",))
        self.eat("(")
        params = []
        while not self.at(")"):
            if self.at("&"):
                self.eat()
            if self.at("mut"):
                self.eat()
            pname = self.eat()[1]
            if pname == "self" and not self.at(":"):
                params.append(pname)
                if self.at(","):
                    self.eat()
                continue
            self.eat(":")
            self.skip_type((",", ")"))
            params.append(pname)
            if self.at(","):
                self.eat()
        self.eat(")")
        if self.at("->"):
            self.eat()
            self.skip_type(("{",))
        return ("fn", name, params, self.block())

    def block(self):
        self.eat("{")
        stmts = []
        while not self.at("}"):
            if self.at("const"):
                self.eat()
                name = self.eat()[1]
                self.eat(":")
                self.skip_type(("=",))
                self.eat("=")
                e = self.expr()
                self.eat(";")
                stmts.append(("let", name, e))
                continue
            if self.at("let"):
                self.eat()
                if self.at("mut"):
                    self.eat()
                name = self.eat()[1]
                if self.at(":"):
                    self.eat()
                    self.skip_type(("=",))
                self.eat("=")
                e = self.expr()
                self.eat(";")
                stmts.append(("let", name, e))
                continue
            if self.at("for"):
                self.eat()
                var = self.eat()[1]
                self.eat("in")
                it = self.expr(no_struct=True)
                body = self.block()
                stmts.append(("for", var, it, body))
                continue
            e = self.expr()
            if self.at(";"):
                self.eat()
                stmts.append(("expr", e))
            else:
                stmts.append(("ret", e))
                break
        self.eat("}")
        return ("block", stmts)

    BIN = [("||",), ("&&",), ("==", "!=", "<", ">", "<=", ">="), ("+", "-"), ("*", "/", "%")]

    def expr(self, no_struct=False, lvl=0):
        if lvl == 0 and self.at("|"):
            return self.closure()
        if lvl == 0 and self.at("||"):
            self.eat()
            return ("closure", [], self.expr())
        if lvl == 0 and self.at("move"):
            self.eat()
            return self.closure()
        if lvl == len(self.BIN):
            return self.range_(no_struct)
        lhs = self.expr(no_struct, lvl + 1)
        while self.peek()[1] in self.BIN[lvl]:
            op = self.eat()[1]
            rhs = self.expr(no_struct, lvl + 1)
            lhs = ("bin", op, lhs, rhs)
        return lhs

    def range_(self, no_struct):
        lhs = self.unary(no_struct)
        if self.at("..=") or self.at(".."):
            op = self.eat()[1]
            rhs = self.unary(no_struct)
            return ("range", lhs, rhs, op == "..=")
        return lhs

    def closure(self):
        self.eat("|")
        params = []
        while not self.at("|"):
            if self.at("&"):
                self.eat()
            if self.at("mut"):
                self.eat()
            params.append(self.eat()[1])
            if self.at(":"):
                self.eat()
                self.skip_type((",", "|"))
            if self.at(","):
                self.eat()
        self.eat("|")
        body = self.block() if self.at("{") else self.expr()
        return ("closure", params, body)

    def unary(self, no_struct):
        if self.at("-"):
            self.eat()
            return ("neg", self.unary(no_struct))
        if self.at("!"):
            self.eat()
            return ("not", self.unary(no_struct))
        if self.at("&") or self.at("*"):
            self.eat()
            if self.at("mut"):
                self.eat()
            return self.unary(no_struct)
        return self.postfix(no_struct)

    def args(self):
        self.eat("(")
        out = []
        while not self.at(")"):
            out.append(self.expr())
            if self.at(","):
                self.eat()
        self.eat(")")
        return out

    def postfix(self, no_struct):
        e = self.primary(no_struct)
        while True:
            if self.at("("):
                e = ("call", e, self.args())
            elif self.at("."):
                self.eat()
                name = self.eat()[1]
                if self.at("::"):  # turbofish
                    self.eat()
                    self.skip_type(("(",))
                if self.at("("):
                    e = ("method", e, name, self.args())
                else:
                    e = ("field", e, name)
            elif self.at("?"):
                self.eat()
            elif self.at("["):
                self.eat()
                ix = self.expr()
                self.eat("]")
                e = ("index", e, ix)
            elif self.at("as"):
                self.eat()
                self.skip_type((",", ")", ";", "}", "]", "+", "-", "*", "/"))
            else:
                return e

    def primary(self, no_struct):
        kind, v = self.peek()
        if kind == "num":
            self.eat()
            if v.startswith("0x"):
                return ("lit", int(v.replace("_", ""), 16))
            txt = re.sub(r"_?(f32|f64|u8|u16|u32|u64|usize|i32|i64|isize)$", "", v).replace("_", "")
            return ("lit", float(txt) if ("." in txt or "e" in txt.lower()) else int(txt))
        if kind == "str":
            self.eat()
            return ("lit", bytes(v[1:-1], "utf-8").decode("unicode_escape"))
        if v == "(":
            self.eat()
            elems = []
            while not self.at(")"):
                elems.append(self.expr())
                if self.at(","):
                    self.eat()
                    if self.at(")"):
                        elems.append(None)  # marker: trailing comma => tuple
            self.eat(")")
            if len(elems) == 1:
                return elems[0]
            return ("tuple", [e for e in elems if e is not None])
        if v == "[":
            self.eat()
            elems = []
            while not self.at("]"):
                elems.append(self.expr())
                if self.at(";"):  # [x; n]
                    self.eat()
                    n = self.expr()
                    self.eat("]")
                    return ("repeat", elems[0], n)
                if self.at(","):
                    self.eat()
            self.eat("]")
            return ("list", elems)
        if v == "{":
            return self.block()
        if v == "if":
            self.eat()
            c = self.expr(no_struct=True)
            a = self.block()
            b = None
            if self.at("else"):
                self.eat()
                b = self.primary(False) if self.at("if") else self.block()
            return ("if", c, a, b)
        if v == "match":
            raise SyntaxError("match is not supported")
        if v in ("vec!", "format!", "assert!", "assert_eq!", "println!", "matches!", "include_str!"):
            self.eat()
            open_ = self.eat()[1]
            close = {"[": "]", "(": ")", "{": "}"}[open_]
            elems = []
            while not self.at(close):
                elems.append(self.expr())
                if self.at(";"):
                    self.eat()
                    n = self.expr()
                    self.eat(close)
                    return ("repeat", elems[0], n)
                if self.at(","):
                    self.eat()
            self.eat(close)
            return ("macro", v, elems)
        if kind == "id":
            path = [self.eat()[1]]
            while self.at("::"):
                self.eat()
                if self.at("<"):
                    self.skip_type(("(", "{", ";", ","))
                    continue
                path.append(self.eat()[1])
            if self.at("{") and not no_struct and path[-1][0].isupper() and self._looks_like_struct():
                self.eat("{")
                fields, base = [], None
                while not self.at("}"):
                    if self.at(".."):
                        self.eat()
                        base = self.expr()
                        break
                    name = self.eat()[1]
                    if self.at(":"):
                        self.eat()
                        fields.append((name, self.expr()))
                    else:
                        fields.append((name, ("path", [name])))
                    if self.at(","):
                        self.eat()
                self.eat("}")
                return ("struct", path, fields, base)
            return ("path", path)
        raise SyntaxError(f"primary? {self.peek()} near {[x[1] for x in self.t[max(0, self.i - 6):self.i + 4]]}")

    def _looks_like_struct(self):
        a, b = self.peek(1), self.peek(2)
        return a[1] == "}" or a[1] == ".." or (a[0] == "id" and b[1] in (":", ",", "}"))


# ----------------------------------------------------------------------------------------------- evaluation
class Closure:
    def __init__(self, params, body, env):
        self.params, self.body, self.env = params, body, env


class Runner:
    def __init__(self, module, name):
        self.module, self.name = module, name
        self.inputs, self.resolution, self.mode, self.steps = [], [640, 360], "gpu_optimized", []
        self.renderers = []  # with_renderers: {"id", "kind": "shader" | "image", "wgsl": file name | "image_type", "source"}
        self.unsupported = None


class Unsupported(Exception):
    pass


def call(f, args):
    if isinstance(f, Closure):
        env = dict(f.env)
        env.update(zip(f.params, args))
        return ev(f.body, env)
    if callable(f):
        return f(*args)
    raise TypeError(f"not callable: {f!r}")


def struct(name, **kw):
    d = {"__struct__": name}
    d.update(kw)
    return d


def test_input(index, resolution=None, kind="solid"):
    r = resolution or struct("Resolution", width=640, height=360)
    return {"__input__": True, "index": index, "id": f"input_{index}", "width": r["width"], "height": r["height"], "kind": kind}


def ms(x):
    return {"__duration_ms__": x}


BUILTINS = {
    ("Some",): lambda x: x,
    ("Box", "new"): lambda x: x,
    ("Arc", "new"): lambda x: x,
    ("Arc", "from"): lambda x: x,
    ("Duration", "from_millis"): lambda x: ms(float(x)),
    ("Duration", "from_secs"): lambda x: ms(1000.0 * x),
    ("Duration", "from_secs_f64"): lambda x: ms(1000.0 * x),
    ("Duration", "from_secs_f32"): lambda x: ms(1000.0 * x),
    ("BorderRadius", "new_with_radius"): lambda x: struct("BorderRadius", radius=x),
    ("TestInput", "new"): lambda i: test_input(i),
    ("TestInput", "new_with_resolution"): lambda i, r: test_input(i, r),
    ("TestInput", "new_multiscale_grid"): lambda i, r: test_input(i, r, kind="multiscale_grid"),
    ("TestRunner", "new"): lambda m, n: Runner(m, n),
    ("RGBAColor",): lambda r, g, b, a: struct("RGBAColor", r=r, g=g, b=b, a=a),
    ("InputId",): lambda s: s,
    ("ComponentId",): lambda s: s,
    ("RendererId",): lambda s: s,
    ("Ok",): lambda x=None: x,
    ("integration_tests_root",): lambda: {"__path__": ""},
    ("submodule_root_path",): lambda: {"__path__": "snapshot_tests_submodule"},
}
IMPLS = {}  # struct name -> {method name: fn item} of the file being evaluated


def ev(n, env):
    k = n[0]
    if k == "lit":
        return n[1]
    if k == "path":
        p = tuple(n[1])
        if len(p) == 1 and p[0] in env:
            return env[p[0]]
        if p == ("None",):
            return None
        if p in (("true",), ("false",)):
            return p[0] == "true"
        if p == ("Duration", "ZERO"):
            return ms(0.0)
        if p == ("Default", "default"):
            return lambda: {"__default__": True}
        if p[-1] == "default":  # ViewComponent::default()
            return (lambda t: lambda: {"__struct__": t})(p[-2])
        if p in BUILTINS:
            return BUILTINS[p]
        if p[0] in ("f32", "f64", "u32", "usize") and len(p) == 2:
            return {"MAX": 3.4e38, "MIN": -3.4e38, "EPSILON": 1.1920929e-07}.get(p[1], 0)
        if len(p) >= 2 and p[-1][0].isupper():  # enum unit variant or tuple-variant constructor
            name = "::".join(p[-2:])
            return EnumCtor(name)
        if len(p) == 1 and p[0][0].isupper():
            return EnumCtor(p[0])
        raise NameError("::".join(p))
    if k == "struct":
        name = n[1][-1] if len(n[1]) == 1 or not n[1][-2][0].isupper() or n[1][-2] in ("scene",) else "::".join(n[1][-2:])
        d = {"__struct__": name}
        for f, e in n[2]:
            d[f] = ev(e, env)
        return d
    if k == "call":
        f = ev(n[1], env)
        args = [ev(a, env) for a in n[2]]
        if isinstance(f, EnumCtor):
            return f(*args)
        return call(f, args)
    if k == "method":
        return method(ev(n[1], env), n[2], [ev(a, env) for a in n[3]], env)
    if k == "field":
        o = ev(n[1], env)
        if isinstance(o, dict):
            if n[2].isdigit():
                return o[["r", "g", "b", "a"][int(n[2])]] if o.get("__struct__") == "RGBAColor" else o["args"][int(n[2])]
            return o[n[2]]
        if isinstance(o, (tuple, list)):
            return o[int(n[2])]
        return getattr(o, n[2])
    if k == "index":
        return ev(n[1], env)[ev(n[2], env)]
    if k == "closure":
        return Closure(n[1], n[2], env)
    if k == "block":
        env = dict(env)
        val = None
        for s in n[1]:
            if s[0] == "let":
                env[s[1]] = ev(s[2], env)
            elif s[0] == "for":
                for x in ev(s[2], env):
                    e2 = dict(env)
                    e2[s[1]] = x
                    ev(s[3], e2)
            elif s[0] == "expr":
                ev(s[1], env)
            else:
                val = ev(s[1], env)
        return val
    if k == "if":
        if ev(n[1], env):
            return ev(n[2], env)
        return ev(n[3], env) if n[3] is not None else None
    if k == "bin":
        a, b, op = ev(n[2], env), ev(n[3], env), n[1]
        if isinstance(a, dict) and "__duration_ms__" in a:
            return ms({"+": a["__duration_ms__"] + (b["__duration_ms__"] if isinstance(b, dict) else b), "*": a["__duration_ms__"] * (b if not isinstance(b, dict) else 1),
                       "/": a["__duration_ms__"] / (b if not isinstance(b, dict) else 1), "-": a["__duration_ms__"] - (b["__duration_ms__"] if isinstance(b, dict) else b)}[op])
        return {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "/": lambda: (a // b if isinstance(a, int) and isinstance(b, int) else a / b),
                "%": lambda: a % b, "==": lambda: a == b, "!=": lambda: a != b, "<": lambda: a < b, ">": lambda: a > b, "<=": lambda: a <= b,
                ">=": lambda: a >= b, "&&": lambda: a and b, "||": lambda: a or b}[op]()
    if k == "neg":
        return -ev(n[1], env)
    if k == "not":
        return not ev(n[1], env)
    if k == "range":
        a, b = ev(n[1], env), ev(n[2], env)
        return list(range(a, b + 1 if n[3] else b))
    if k == "tuple":
        return tuple(ev(e, env) for e in n[1])
    if k == "list":
        return [ev(e, env) for e in n[1]]
    if k == "repeat":
        return [ev(n[1], env)] * ev(n[2], env)
    if k == "macro":
        if n[1] == "vec!":
            return [ev(e, env) for e in n[2]]
        if n[1] == "include_str!":
            return {"__include_str__": ev(n[2][0], env)}
        if n[1] == "format!":
            fmt = ev(n[2][0], env)
            rest = [ev(e, env) for e in n[2][1:]]
            it = iter(rest)
            return re.sub(r"\{(\w*)(?::[^}]*)?\}", lambda m: str(env[m.group(1)] if m.group(1) else next(it)), fmt)
        return None
    raise NotImplementedError(k)


class EnumCtor:
    """Path that names an enum variant: used bare (unit variant), called (tuple variant) — struct variants come as ("struct", ..)."""

    def __init__(self, name):
        self.name = name

    def __call__(self, *args):
        if self.name.startswith("Component::"):
            return args[0]
        return {"__enum__": self.name, "args": list(args)}


def method(o, name, args, env):
    if isinstance(o, Runner):
        if name == "with_inputs":
            o.inputs = list(args[0])
        elif name == "with_resolution":
            o.resolution = [args[0]["width"], args[0]["height"]]
        elif name == "with_rendering_mode":
            o.mode = "cpu_optimized" if "Cpu" in unit(args[0]) else "gpu_optimized"
        elif name == "with_renderers":
            for rid, spec in args[0]:
                kind, inner = spec["__enum__"], spec["args"][0]
                if kind.endswith("Shader"):
                    o.renderers.append({"id": rid, "kind": "shader", "wgsl": os.path.basename(inner["source"]["__include_str__"])})
                elif kind.endswith("Image"):
                    src = inner["src"]
                    where = src.get("url") or src.get("path")
                    o.renderers.append({"id": rid, "kind": "image", "image_type": unit(inner["image_type"]).split("::")[-1].lower(),
                                        "source": where if isinstance(where, str) else os.path.basename(str(where.get("__path__", where)))})
                else:
                    o.unsupported = f"registers a {kind} renderer"
        elif name == "update_scene":
            o.steps.append({"update": component(args[0])})
        elif name in ("snapshot", "render"):
            o.steps.append({"snapshot_ms" if name == "snapshot" else "render_ms": args[0]["__duration_ms__"]})
        elif name == "finish":
            return None
        else:
            raise Unsupported(f"TestRunner::{name}")
        return o
    if isinstance(o, dict) and "__path__" in o and name == "join":
        return {"__path__": (o["__path__"] + "/" if o["__path__"] else "") + args[0]}
    if isinstance(o, dict) and o.get("__struct__") in IMPLS and name in IMPLS[o["__struct__"]]:
        it = IMPLS[o["__struct__"]][name]
        return ev(it[3], {**env, **dict(zip(it[2], [o] + list(args)))})
    if name in ("into", "clone", "to_string", "to_owned", "collect", "iter", "into_iter", "cloned", "copied", "unwrap", "as_str", "to_vec", "as_ref"):
        return o
    if name == "map":
        if o is None:
            return None
        if isinstance(o, list):
            return [call(args[0], [x]) for x in o]
        return call(args[0], [o])
    if name == "then":  # bool::then
        return call(args[0], []) if o else None
    if name == "then_some":
        return args[0] if o else None
    if name == "rev":
        return list(reversed(o))
    if name == "enumerate":
        return [(i, x) for i, x in enumerate(o)]
    if name == "chain":
        return list(o) + list(args[0])
    if name == "filter":
        return [x for x in o if call(args[0], [x])]
    if name == "push":
        o.append(args[0])
        return None
    if name == "len":
        return len(o)
    if name in ("min", "max"):
        return min(o, args[0]) if name == "min" else max(o, args[0])
    if name == "unwrap_or":
        return args[0] if o is None else o
    if name == "is_some":
        return o is not None
    raise Unsupported(f"method .{name}()")


def unit(v):
    if isinstance(v, EnumCtor):
        return v.name
    if isinstance(v, dict) and "__enum__" in v:
        return v["__enum__"]
    if isinstance(v, dict) and "__struct__" in v:
        return v["__struct__"]
    return str(v)


# ----------------------------------------------------------------------------------------------- scene::Component -> smelter-api JSON
def color(c):
    return "#%02X%02X%02X%02X" % (c["r"], c["g"], c["b"], c["a"])


def snake(v):
    return re.sub(r"(?<!^)(?=[A-Z])", "_", unit(v).split("::")[-1]).lower()


def radius(b):
    if "radius" in b:
        return float(b["radius"])
    vals = [b[k] for k in ("top_left", "top_right", "bottom_right", "bottom_left")]
    if len(set(vals)) != 1:
        raise Unsupported("per-corner border radius has no JSON form")
    return float(vals[0])


def transition(t):
    if t is None:
        return None
    out = {"duration_ms": t["duration"]["__duration_ms__"]}
    kind = t.get("interpolation_kind")
    name = unit(kind)
    if "CubicBezier" in name:
        out["easing_function"] = {"function_name": "cubic_bezier", "points": [float(kind[k]) for k in ("x1", "y1", "x2", "y2")]}
    elif "Bounce" in name:
        out["easing_function"] = {"function_name": "bounce"}
    else:
        out["easing_function"] = {"function_name": "linear"}
    if t.get("should_interrupt"):
        out["should_interrupt"] = True
    return out


def position(p, out):
    if p is None or (isinstance(p, dict) and p.get("__default__")):
        return
    name = unit(p)
    if name.endswith("Static"):
        for k in ("width", "height"):
            if p.get(k) is not None:
                out[k] = float(p[k])
        return
    a = p["args"][0]
    for k in ("width", "height"):
        if a.get(k) is not None:
            out[k] = float(a[k])
    h, v = a["position_horizontal"], a["position_vertical"]
    out["left" if "Left" in unit(h) else "right"] = float(h["args"][0])
    out["top" if "Top" in unit(v) else "bottom"] = float(v["args"][0])
    if a.get("rotation_degrees"):
        out["rotation"] = float(a["rotation_degrees"])


def common(c, out):
    if c.get("id") is not None:
        out["id"] = c["id"]
    if "position" in c:
        position(c["position"], out)
    if c.get("transition") is not None:
        out["transition"] = transition(c["transition"])
    if "border_radius" in c:
        out["border_radius"] = radius(c["border_radius"])
    if "border_width" in c:
        out["border_width"] = float(c["border_width"])
    if "border_color" in c:
        out["border_color"] = color(c["border_color"])
    if c.get("box_shadow"):
        out["box_shadow"] = [{"offset_x": float(b["offset_x"]), "offset_y": float(b["offset_y"]), "blur_radius": float(b["blur_radius"]), "color": color(b["color"])}
                             for b in c["box_shadow"]]


def component(c):
    s = c.get("__struct__")
    if s == "InputStreamComponent":
        out = {"type": "input_stream", "input_id": c["input_id"]}
        if c.get("id") is not None:
            out["id"] = c["id"]
        return out
    if s == "ViewComponent":
        out = {"type": "view"}
        common(c, out)
        if "children" in c:
            out["children"] = [component(k) for k in c["children"]]
        if "direction" in c:
            out["direction"] = snake(c["direction"])
        if "overflow" in c:
            out["overflow"] = snake(c["overflow"])
        if "background_color" in c:
            out["background_color"] = color(c["background_color"])
        if "padding" in c:
            p = c["padding"]
            for k in ("top", "right", "bottom", "left"):
                if p.get(k):
                    out[f"padding_{k}"] = float(p[k])
        return out
    if s == "RescalerComponent":
        out = {"type": "rescaler", "child": component(c["child"])}
        common(c, out)
        if "mode" in c:
            out["mode"] = snake(c["mode"])
        for k in ("horizontal_align", "vertical_align"):
            if k in c:
                out[k] = snake(c[k])
        return out
    if s == "TilesComponent":
        out = {"type": "tiles"}
        if c.get("id") is not None:
            out["id"] = c["id"]
        if "children" in c:
            out["children"] = [component(k) for k in c["children"]]
        for k in ("width", "height"):
            if c.get(k) is not None:
                out[k] = float(c[k])
        if "background_color" in c:
            out["background_color"] = color(c["background_color"])
        if "tile_aspect_ratio" in c:
            out["tile_aspect_ratio"] = "%d:%d" % tuple(c["tile_aspect_ratio"])
        for k in ("margin", "padding"):
            if k in c:
                out[k] = float(c[k])
        for k in ("horizontal_align", "vertical_align"):
            if k in c:
                out[k] = snake(c[k])
        if c.get("transition") is not None:
            out["transition"] = transition(c["transition"])
        return out
    if s == "TextComponent":
        out = {"type": "text", "text": c["text"], "font_size": float(c["font_size"])}
        if c.get("id") is not None:
            out["id"] = c["id"]
        d = c.get("dimensions")
        if d is not None:
            kind = d["__struct__"].split("::")[-1] if "__struct__" in d else unit(d).split("::")[-1]
            if kind == "Fixed":
                out["width"], out["height"] = float(d["width"]), float(d["height"])
            elif kind == "FittedColumn":
                out["width"] = float(d["width"])
                if d.get("max_height") is not None:
                    out["max_height"] = float(d["max_height"])
            else:
                for k in ("max_width", "max_height"):
                    if d.get(k) is not None:
                        out[k] = float(d[k])
        if "line_height" in c:
            out["line_height"] = float(c["line_height"])
        if "color" in c:
            out["color"] = color(c["color"])
        if "background_color" in c:
            out["background_color"] = color(c["background_color"])
        if "font_family" in c:
            out["font_family"] = c["font_family"]
        for k in ("style", "align", "wrap", "weight"):
            if k in c:
                out[k] = snake(c[k])
        return out
    if s == "ImageComponent":
        out = {"type": "image", "image_id": c["image_id"]}
        if c.get("id") is not None:
            out["id"] = c["id"]
        for k in ("width", "height"):
            if c.get(k) is not None:
                out[k] = float(c[k])
        return out
    if s == "ShaderComponent":
        out = {"type": "shader", "shader_id": c["shader_id"], "resolution": {"width": int(c["size"]["width"]), "height": int(c["size"]["height"])}}
        if c.get("id") is not None:
            out["id"] = c["id"]
        if c.get("children"):
            out["children"] = [component(k) for k in c["children"]]
        if c.get("shader_param") is not None:
            out["shader_param"] = shader_param(c["shader_param"])
        return out
    raise Unsupported(f"{s} component")


def shader_param(p):
    kind, args = p["__enum__"].split("::")[-1], p["args"]
    if kind in ("F32", "U32", "I32"):
        return {"type": kind.lower(), "value": float(args[0]) if kind == "F32" else int(args[0])}
    if kind == "List":
        return {"type": "list", "value": [shader_param(x) for x in args[0]]}
    if kind == "Struct":
        return {"type": "struct", "value": [dict(shader_param(f["value"]), field_name=f["field_name"]) for f in args[0]]}
    raise Unsupported(f"ShaderParam::{kind}")


# ----------------------------------------------------------------------------------------------- driver
def extract(reference):
    tests, skipped = [], []
    for mod in FILES:
        src = open(os.path.join(reference, "integration-tests", "src", "render_tests", mod + ".rs")).read()
        items = P(tokenize(src)).items()
        env = {"MODULE": mod, "DEFAULT_RESOLUTION": struct("Resolution", width=640, height=360)}  # harness/mod.rs:13-17
        fns = {}
        for it in items:
            if it[0] == "const":
                try:
                    env[it[1]] = ev(it[2], env)
                except Exception as ex:
                    if it[1] != "TESTS":  # (the TESTS tables name generated statics)
                        print(f"  {mod}: const {it[1]} not evaluated: {ex!r}", file=sys.stderr)
        IMPLS.clear()
        for it in items:
            if it[0] == "fn":
                fns[it[1]] = it
            if it[0] == "impl":
                IMPLS.setdefault(it[1], {}).update({f[1]: f for f in it[2]})
        for name, it in fns.items():
            env[name] = (lambda it: lambda *a: ev(it[3], {**env, **dict(zip(it[2], a))}))(it)
        is_test = re.compile(r"#\[render_test[^\]]*\]\s*fn\s+(\w+)")
        for name in is_test.findall(src):
            e = dict(env)
            e["TEST_NAME"] = name
            runners = []
            real = BUILTINS[("TestRunner", "new")]
            BUILTINS[("TestRunner", "new")] = lambda m, n: runners.append(Runner(m, n)) or runners[-1]
            try:
                ev(fns[name][3], e)
                r = runners[0]
                if r.unsupported:
                    raise Unsupported(r.unsupported)
                t = {"module": mod, "name": name, "resolution": r.resolution, "mode": r.mode,
                     "inputs": [{k: i[k] for k in ("id", "index", "width", "height", "kind")} for i in r.inputs], "steps": r.steps}
                if r.renderers:
                    t["renderers"] = r.renderers
                tests.append(t)
            except Unsupported as ex:
                skipped.append((mod, name, str(ex)))
            except Exception as ex:  # a construct the interpreter does not know: reported, never silently dropped
                skipped.append((mod, name, f"interpreter: {type(ex).__name__}: {ex}"))
            finally:
                BUILTINS[("TestRunner", "new")] = real
    return tests, skipped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    tests, skipped = extract(args.reference)
    out = os.path.join(ROOT, "tests", "golden", "render_test_scenes.json")
    json.dump({"source": "integration-tests/src/render_tests/{" + ",".join(FILES) + "}.rs, evaluated by tests/golden/gen_render_test_scenes.py",
               "tests": tests, "skipped": [{"module": m, "name": n, "why": w} for m, n, w in skipped]}, open(out, "w"), indent=1)
    snaps = sum(1 for t in tests for s in t["steps"] if "snapshot_ms" in s)
    print(f"{out}: {len(tests)} tests, {snaps} snapshots, {len(skipped)} skipped")
    for m, n, w in skipped:
        print("  skipped", m, n, "-", w)


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
