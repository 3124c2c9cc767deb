#!/usr/bin/env python3
"""Extracts the reference's scene-deserialisation test vectors into tests/golden/scene_api_vectors.json.

Source: /root/reference/smelter-api/tests/scene_deserialization.rs — each #[test] there is
  check(json!(..), <scene::Component literal>)   -> {"kind": "ok",  "scene": .., "expected": <canonical tree>}
  check_err(json!(..), "<message>")              -> {"kind": "err", "scene": .., "message": ..}
  check_serde_err(json!(..))                     -> {"kind": "serde_err", "scene": ..}
The Rust struct literals are parsed with a small expression parser and folded (with the `Default` impls of
smelter-render/src/scene/components.rs:266-347) into the canonical tree `smr_scene_parse` emits.
Run in the build container only (the reference is not on the GPU box); the JSON output is committed.
"""
import json
import os
import re
import sys

import numpy as np

SRC = "/root/reference/smelter-api/tests/scene_deserialization.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_api_vectors.json")


# ----------------------------------------------------------------------------- tokenizer / parser for the Rust subset
TOKEN = re.compile(r"""\s*(?:(//[^\n]*)|("(?:\\.|[^"\\])*")|(0x[0-9a-fA-F]+|-?\d[\d_]*(?:\.\d+)?(?:e-?\d+)?(?:_?[fiu]\d+)?)|([A-Za-z_][A-Za-z0-9_]*(?:::[A-Za-z_][A-Za-z0-9_]*)*)|(\.\.|[(){}\[\],:!.\-&]))""")


def tokenize(text):
    pos, out = 0, []
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise SyntaxError(f"cannot tokenize at {text[pos:pos + 40]!r}")
        pos = m.end()
        if m.group(1):
            continue
        if m.group(2):
            out.append(("str", rust_string(m.group(2))))
        elif m.group(3) and m.group(3).startswith("0x"):
            out.append(("num", float(int(m.group(3), 16))))
        elif m.group(3):
            out.append(("num", float(re.sub(r"_?[fiu]\d+$", "", m.group(3)).replace("_", ""))))
        elif m.group(4):
            out.append(("id", m.group(4)))
        else:
            out.append(("p", m.group(5)))
    return out


def rust_string(lit):
    body = lit[1:-1]
    body = re.sub(r"\\\n\s*", "", body)  # line continuation
    return body.replace('\\"', '"').replace("\\\\", "\\").replace("\\n", "\n")


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", None)

    def take(self, kind=None, val=None):
        tok = self.peek()
        if (kind and tok[0] != kind) or (val is not None and tok[1] != val):
            raise SyntaxError(f"expected {kind} {val}, got {tok} at {self.i}")
        self.i += 1
        return tok

    def args(self, close):
        items = []
        while self.peek() != ("p", close):
            items.append(self.expr())
            if self.peek() == ("p", ","):
                self.take()
        self.take("p", close)
        return items

    def expr(self):
        node = self.primary()
        while True:
            if self.peek() == ("p", ".") and self.peek(1)[0] == "id":
                self.take()
                name = self.take("id")[1]
                if self.peek() == ("p", "("):
                    self.take()
                    node = ("method", name, node, self.args(")"))
                else:
                    node = ("field", node, name)
            elif self.peek() == ("id", "as"):
                self.take()
                self.take("id")
            else:
                return node

    def primary(self):
        kind, val = self.peek()
        if kind == "num" or kind == "str":
            self.take()
            return val
        if (kind, val) == ("p", "-"):
            self.take()
            return -self.take("num")[1]
        if (kind, val) == ("p", "&"):
            self.take()
            return self.primary()
        if (kind, val) == ("p", "("):
            self.take()
            return ("tuple", self.args(")"))
        if kind == "id":
            self.take()
            if val in ("true", "false"):
                return val == "true"
            nxt = self.peek()
            if nxt == ("p", "("):
                self.take()
                return ("call", val, self.args(")"))
            if nxt == ("p", "!"):
                self.take()
                self.take("p", "[")
                return ("vec", self.args("]"))
            if nxt == ("p", "{"):
                self.take()
                fields, base = {}, None
                while self.peek() != ("p", "}"):
                    if self.peek() == ("p", ".."):
                        self.take()
                        base = self.expr()
                    else:
                        name = self.take("id")[1]
                        self.take("p", ":")
                        fields[name] = self.expr()
                    if self.peek() == ("p", ","):
                        self.take()
                self.take("p", "}")
                return ("struct", val, fields, base)
            return ("path", val)
        raise SyntaxError(f"unexpected token {kind} {val}")


# ----------------------------------------------------------------------------- canonical tree
f32 = lambda v: float(np.float32(v))  # noqa: E731
MAX_W, MAX_H = 7682.0, 4320.0  # MAX_NODE_RESOLUTION (smelter-render/src/types.rs:146-149)


def last(path):
    return path.split("::")[-1]


def val(n):
    """Scalars, Option, ids, strings."""
    if isinstance(n, (int, float, str, bool)) or n is None:
        return n
    tag = n[0]
    if tag == "path":
        return None if n[1] == "None" else last(n[1])
    if tag == "call":
        name = last(n[1])
        if name in ("Some", "component_id", "renderer_id", "from", "RendererId", "ComponentId", "InputId", "new"):
            return val(n[2][0])
        raise ValueError(f"val: call {n[1]}")
    if tag == "method":  # "x".into() / .to_string()
        return val(n[2])
    if tag == "field":   # smelter_render::MAX_NODE_RESOLUTION.width
        return {"width": MAX_W, "height": MAX_H}[n[2]]
    raise ValueError(f"val: {n}")


def opt_f32(n):
    v = val(n)
    return None if v is None else f32(v)


def color(n):
    assert n[0] == "call" and last(n[1]) == "RGBAColor", n
    return [int(x) for x in n[2]]


def radius(n):
    if n[0] == "call" and last(n[1]) == "new_with_radius":
        return [f32(n[2][0])] * 4
    if n[0] == "path" and last(n[1]) == "ZERO":
        return [0.0] * 4
    f = n[2]
    return [f32(f["top_left"]), f32(f["top_right"]), f32(f["bottom_right"]), f32(f["bottom_left"])]


def position(n):
    if n[0] == "struct" and last(n[1]) == "Static":
        return {"kind": "Static", "width": opt_f32(n[2]["width"]), "height": opt_f32(n[2]["height"])}
    assert n[0] == "call" and last(n[1]) == "Absolute", n
    f = n[2][0][2]
    h, v = f["position_horizontal"], f["position_vertical"]
    return {"kind": "Absolute", "width": opt_f32(f["width"]), "height": opt_f32(f["height"]),
            "horizontal": ["Right" if last(h[1]) == "RightOffset" else "Left", f32(h[2][0])],
            "vertical": ["Bottom" if last(v[1]) == "BottomOffset" else "Top", f32(v[2][0])],
            "rotation_degrees": f32(f["rotation_degrees"])}


def transition(n):
    if n == ("path", "None"):
        return None
    f = n[2][0][2]
    d = f["duration"]
    assert d[0] == "call" and last(d[1]) in ("from_millis", "from_secs"), d
    ns = int(d[2][0]) * (1_000_000 if last(d[1]) == "from_millis" else 1_000_000_000)
    k = f["interpolation_kind"]
    if k[0] == "struct":
        interp = ["CubicBezier"] + [float(k[2][x]) for x in ("x1", "y1", "x2", "y2")]
    else:
        interp = last(k[1])
    return {"duration_ns": ns, "interpolation": interp, "should_interrupt": bool(val(f["should_interrupt"]))}


def shadows(n):
    return [{"offset_x": f32(s[2]["offset_x"]), "offset_y": f32(s[2]["offset_y"]), "blur_radius": f32(s[2]["blur_radius"]),
             "color": color(s[2]["color"])} for s in n[1]]


def padding(n):
    if n[0] == "path":
        return [0.0] * 4
    f = n[2]
    return [f32(f["top"]), f32(f["right"]), f32(f["bottom"]), f32(f["left"])]


def shader_param(n):
    if n == ("path", "None"):
        return None
    if n[0] == "call" and last(n[1]) == "Some":
        n = n[2][0]
    name = last(n[1])
    if name == "F32":
        return ["F32", f32(n[2][0])]
    if name in ("U32", "I32"):
        return [name, float(n[2][0])]
    if name == "List":
        return ["List", [shader_param(x) for x in n[2][0][1]]]
    assert name == "Struct", n
    return ["Struct", [[val(x[2]["field_name"]), shader_param(x[2]["value"])] for x in n[2][0][1]]]


VIEW_DEFAULT = {"type": "View", "id": None, "children": [], "direction": "Row", "position": {"kind": "Static", "width": None, "height": None},
                "transition": None, "overflow": "Hidden", "background_color": [0, 0, 0, 0], "border_radius": [0.0] * 4, "border_width": 0.0,
                "border_color": [0, 0, 0, 0], "box_shadow": [], "padding": [0.0] * 4}
RESCALER_DEFAULT = {"type": "Rescaler", "id": None, "child": None, "position": {"kind": "Static", "width": None, "height": None},
                    "transition": None, "mode": "Fit", "horizontal_align": "Center", "vertical_align": "Center", "border_radius": [0.0] * 4,
                    "border_width": 0.0, "border_color": [0, 0, 0, 0], "box_shadow": []}
TILES_DEFAULT = {"type": "Tiles", "id": None, "width": None, "height": None, "margin": 0.0, "padding": 0.0, "children": [], "transition": None,
                 "vertical_align": "Center", "horizontal_align": "Center", "background_color": [0, 0, 0, 0], "tile_aspect_ratio": [16.0, 9.0]}


def text_default(text, font_size):
    return {"type": "Text", "id": None, "text": text, "font_size": f32(font_size), "line_height": f32(font_size), "color": [255] * 4,
            "font_family": "Verdana", "style": "Normal", "align": "Left", "weight": "Normal", "wrap": "None", "background_color": [0, 0, 0, 0],
            "dimensions": ["Fitted", MAX_W, MAX_H]}


def dimensions(n):
    f = n[2]
    name = last(n[1])
    if name == "Fixed":
        return ["Fixed", f32(val(f["width"])), f32(val(f["height"]))]
    if name == "FittedColumn":
        return ["FittedColumn", f32(val(f["width"])), f32(val(f["max_height"]))]
    return ["Fitted", f32(val(f["max_width"])), f32(val(f["max_height"]))]


CONVERT = {
    "id": val, "children": lambda n: [component(x) for x in n[1]], "direction": val, "position": position, "transition": transition,
    "overflow": val, "background_color": color, "border_radius": radius, "border_width": lambda n: f32(val(n)), "border_color": color,
    "box_shadow": shadows, "child": lambda n: component(n), "mode": val, "horizontal_align": val, "vertical_align": val,
    "width": opt_f32, "height": opt_f32, "margin": lambda n: f32(val(n)), "tile_aspect_ratio": lambda n: [float(x) for x in n[1]],
    "text": val, "font_size": lambda n: f32(val(n)), "line_height": lambda n: f32(val(n)), "color": color, "font_family": val, "style": val,
    "align": val, "weight": val, "wrap": val, "dimensions": dimensions, "image_id": val, "shader_id": val, "shader_param": shader_param,
    "size": lambda n: [f32(n[2]["width"]), f32(n[2]["height"])], "instance_id": val, "input_id": val,
}


def component(n):
    tag = n[0]
    if tag == "call":
        name = last(n[1])
        if name == "input_stream":
            return {"type": "InputStream", "id": val(n[2][0]), "input_id": val(n[2][1])}
        if name in ("View", "Rescaler", "Tiles", "Text", "Image", "Shader", "WebView", "InputStream", "new"):
            return component(n[2][0])
        if name in ("view_default", "default") and "View" in n[1] + "View":
            return dict(VIEW_DEFAULT)
        raise ValueError(f"component: call {n[1]}")
    if tag == "method" and n[1] == "into":
        return component(n[2])
    assert tag == "struct", n
    kind = last(n[1]).replace("Component", "")
    fields, base = n[2], n[3]
    if base is not None:
        bname = last(base[1])
        if bname == "view_default" or (bname == "default" and "View" in base[1]):
            out = dict(VIEW_DEFAULT)
        elif bname == "rescaler_default":
            out = dict(RESCALER_DEFAULT)
            out["child"] = component(base[2][0])
        elif bname == "tiles_default" or (bname == "default" and "Tiles" in base[1]):
            out = dict(TILES_DEFAULT)
        elif bname == "text_default":
            out = text_default(val(base[2][0]), val(base[2][1]))
        elif bname == "default" and "Rescaler" in base[1]:
            out = dict(RESCALER_DEFAULT)
            out["child"] = dict(VIEW_DEFAULT)
        else:
            raise ValueError(f"unknown base {base}")
    else:
        out = {"type": kind}
    for k, v in fields.items():
        if kind == "View" and k == "padding":
            out[k] = padding(v)
        elif kind == "Tiles" and k == "padding":
            out[k] = f32(val(v))
        else:
            out[k] = CONVERT[k](v)
    return out


# ----------------------------------------------------------------------------- extraction
def balanced(text, start, open_c="(", close_c=")"):
    depth, i, in_str = 0, start, False
    while i < len(text):
        c = text[i]
        if in_str:
            if c == "\\":
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c == open_c:
            depth += 1
        elif c == close_c:
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise SyntaxError("unbalanced")


def main():
    src = open(SRC).read()
    vectors = []
    for m in re.finditer(r"#\[test\]\s*fn (\w+)\(\) \{", src):
        name = m.group(1)
        body_end = balanced(src, m.end() - 1, "{", "}")
        body = src[m.end():body_end]
        for call in re.finditer(r"\b(check|check_err|check_serde_err)\(", body):
            end = balanced(body, call.end() - 1)
            inner = body[call.end():end]
            jm = re.search(r"json!\(", inner)
            jend = balanced(inner, jm.end() - 1)
            scene = json.loads(inner[jm.end():jend])["video"]["root"]
            rest = inner[jend + 1:].strip().lstrip(",").strip().rstrip(",").strip()
            if call.group(1) == "check_serde_err":
                vectors.append({"name": name, "kind": "serde_err", "scene": scene})
            elif call.group(1) == "check_err":
                vectors.append({"name": name, "kind": "err", "scene": scene, "message": Parser(tokenize(rest)).expr()})
            else:
                vectors.append({"name": name, "kind": "ok", "scene": scene, "expected": component(Parser(tokenize(rest)).expr())})
    json.dump({"source": "smelter-api/tests/scene_deserialization.rs", "vectors": vectors}, open(OUT, "w"), indent=1)
    print(f"{len(vectors)} vectors -> {OUT}", {k: sum(v['kind'] == k for v in vectors) for k in ('ok', 'err', 'serde_err')})


if __name__ == "__main__":
    sys.exit(main())
