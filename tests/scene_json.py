"""Scene JSON (the smelter API) -> oracle/scene.py component objects.  TEST INFRASTRUCTURE.

The product parses scene JSON in C++ (smelter_amd/csrc/host/scene_*.cpp).  This is an independent reading of the same conversion,
written from the reference's smelter-api/src/video/component_into.rs, so that the layouts the oracle renders for the reference's
render-test scenes do not come out of the C++ engine under test:
  View      component_into.rs:36-148   (absolute iff top / bottom / left / right / rotation is given; padding_* > padding_vertical /
                                        padding_horizontal > padding; defaults: row, overflow hidden, transparent colours)
  Rescaler  component_into.rs:150-226  (defaults: fit, centre / centre)
  Tiles     component_into.rs:382-416  (defaults: 16:9, margin 0, padding 0, centre / centre)
  BoxShadow component_into.rs:418-432  (defaults: offsets 0, blur 0, white)
Input streams are numbered in order of appearance (depth first): the n-th one is child n of the root layout node.
Component ids and `transition` objects (smelter-api/src/video/transition.rs:11-69: duration_ms, easing_function {function_name, points},
should_interrupt) are carried onto the oracle's components; oracle/transition.py runs the scene's history from them.
Node kinds other than input streams (text, image, shader, web view) are outside what oracle/scene.py restates: `Unsupported` is raised.
"""
from __future__ import annotations

from oracle import scene as S
from oracle import transition as T


class Unsupported(Exception):
    pass


def colour(text):
    """#RRGGBB / #RRGGBBAA (smelter-api/src/common_core/... RGBAColor: what the reference's render tests use)."""
    if not isinstance(text, str) or not text.startswith("#") or len(text) not in (7, 9):
        raise Unsupported(f"colour {text!r}")
    v = [int(text[i:i + 2], 16) for i in range(1, len(text), 2)]
    return tuple(v) if len(v) == 4 else (v[0], v[1], v[2], 255)


def _position(js, kw):
    absolute = any(k in js for k in ("top", "bottom", "left", "right", "rotation"))
    if not absolute:
        kw["width"], kw["height"] = js.get("width"), js.get("height")
        return
    if ("top" in js) == ("bottom" in js) or ("left" in js) == ("right" in js):
        raise Unsupported("absolute position needs exactly one of top / bottom and one of left / right")
    kw["absolute"] = S.AbsolutePosition(width=js.get("width"), height=js.get("height"), top=js.get("top"), bottom=js.get("bottom"),
                                        left=js.get("left"), right=js.get("right"), rotation_degrees=js.get("rotation", 0.0))


def _decor(js, kw):
    kw["border_radius"] = js.get("border_radius", 0.0)
    kw["border_width"] = js.get("border_width", 0.0)
    kw["border_color"] = colour(js["border_color"]) if "border_color" in js else (0, 0, 0, 0)
    kw["box_shadow"] = [S.BoxShadow(b.get("offset_x", 0.0), b.get("offset_y", 0.0), b.get("blur_radius", 0.0),
                                    colour(b["color"]) if "color" in b else (255, 255, 255, 255)) for b in js.get("box_shadow", [])]


def transition(js):
    """Transition -> scene::Transition (transition.rs:33-69)."""
    if js is None:
        return None
    ef = js.get("easing_function") or {"function_name": "linear"}
    name = ef.get("function_name", "linear")
    if name == "cubic_bezier":
        pts = tuple(float(x) for x in ef["points"])
        if not (0.0 <= pts[0] <= 1.0 and 0.0 <= pts[2] <= 1.0):
            raise Unsupported("cubic bezier control points outside [0, 1]")
        easing = T.Easing("cubic_bezier", pts)
    elif name in ("linear", "bounce"):
        easing = T.Easing(name)
    else:
        raise Unsupported(f"easing function {name!r}")
    return T.TransitionOptions(duration_ns=int(round(float(js["duration_ms"]) / 1000.0 * 1e9)), easing=easing, should_interrupt=bool(js.get("should_interrupt", False)))


class Converter:
    def __init__(self):
        self.input_ids = []  # input_id of every input stream component, in order of appearance

    def component(self, js):
        kind = js.get("type")
        if kind == "input_stream":
            self.input_ids.append(js["input_id"])
            return S.InputStream(len(self.input_ids) - 1, id=js.get("id"))
        kw = {"id": js.get("id"), "transition": transition(js.get("transition"))}
        if kind == "view":
            _position(js, kw)
            _decor(js, kw)
            kw["direction"] = js.get("direction", "row")
            kw["overflow"] = js.get("overflow", "hidden")
            kw["background_color"] = colour(js["background_color"]) if "background_color" in js else (0, 0, 0, 0)
            pick = lambda side, axis: js.get("padding_" + side, js.get("padding_" + axis, js.get("padding", 0.0)))  # noqa: E731
            kw["padding"] = S.Padding(top=pick("top", "vertical"), right=pick("right", "horizontal"), bottom=pick("bottom", "vertical"),
                                      left=pick("left", "horizontal"))
            return S.View(children=[self.component(c) for c in js.get("children", [])], **kw)
        if kind == "rescaler":
            _position(js, kw)
            _decor(js, kw)
            kw["mode"] = js.get("mode", "fit")
            kw["horizontal_align"] = js.get("horizontal_align", "center")
            kw["vertical_align"] = js.get("vertical_align", "center")
            return S.Rescaler(child=self.component(js["child"]), **kw)
        if kind == "tiles":
            kw["width"], kw["height"] = js.get("width"), js.get("height")
            kw["background_color"] = colour(js["background_color"]) if "background_color" in js else (0, 0, 0, 0)
            if "tile_aspect_ratio" in js:
                a, b = js["tile_aspect_ratio"].split(":")
                kw["tile_aspect_ratio"] = (int(a), int(b))
            kw["margin"] = js.get("margin", 0.0)
            kw["padding"] = js.get("padding", 0.0)
            kw["horizontal_align"] = js.get("horizontal_align", "center")
            kw["vertical_align"] = js.get("vertical_align", "center")
            return S.Tiles(children=[self.component(c) for c in js.get("children", [])], **kw)
        raise Unsupported(f"component type {kind!r}")


def to_oracle(scene_json):
    """-> (oracle.scene root component, [input_id of child 0, child 1, ...])"""
    c = Converter()
    root = c.component(scene_json)
    if isinstance(root, S.InputStream):
        raise Unsupported("a bare input stream is not a layout tree")
    return root, c.input_ids
