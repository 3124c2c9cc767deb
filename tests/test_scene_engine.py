"""Host scene engine (smelter_amd/csrc/host/scene*.cpp) — CPU only.

The C++ engine and oracle/scene.py are two independent restatements of smelter-render/src/scene/* + layout/flatten.rs;
they must agree word for word on the flattened smr_layout list.  Transition timing, easing and colour parsing are checked
against the reference's own test vector (scene/transition/cubic_bezier.rs:140-147) and against closed forms.
"""
import ctypes as C
import json
import math

import numpy as np
import pytest

import scenes
from oracle import oracle as orc
from oracle import scene as S
from smelter_amd import _ffi
from smelter_amd.scene import Scene, SceneError, bounce_easing, cubic_bezier_easing, parse_color

WORDS = C.sizeof(_ffi.Layout) // 4
pytestmark = pytest.mark.filterwarnings("ignore::RuntimeWarning")  # 0x0 inputs put inf/NaN through the f32 layout maths on purpose


def _words(arr, n):
    raw = bytes(arr)[: n * C.sizeof(_ffi.Layout)]
    return np.frombuffer(raw, np.uint32).reshape(n, WORDS), np.frombuffer(raw, np.float32).reshape(n, WORDS)


def assert_same_layouts(engine, n_engine, oracle_layouts):
    assert n_engine == len(oracle_layouts), (n_engine, len(oracle_layouts))
    if n_engine == 0:
        return
    eu, ef = _words(engine, n_engine)
    ou, of = _words(orc.pack_layouts(oracle_layouts), n_engine)
    with np.errstate(invalid="ignore"):
        same = (eu == ou) | (ef == of)  # +0 / -0 are the same coordinate
    bad = np.argwhere(~same)
    assert bad.size == 0, f"first mismatch layout {bad[0][0]} word {bad[0][1]}: {ef[tuple(bad[0])]} vs {of[tuple(bad[0])]}"


# ----------------------------------------------------------------------------- BASELINE scenes
def test_cfg2_matches_oracle():
    want, res = scenes.cfg2_scene()
    arr, n = scenes.engine_layouts(scenes.cfg2_scene_json(), 1920, 1080, res)
    assert_same_layouts(arr, n, want)


@pytest.mark.parametrize("with_text", [True, False])
def test_cfg3_matches_oracle(with_text):
    want, res = scenes.cfg3_scene(with_text=with_text)
    arr, n = scenes.engine_layouts(scenes.cfg3_scene_json(with_text=with_text), 3840, 2160, res)
    assert_same_layouts(arr, n, want)


def test_cpu_optimized_colors():
    want = S.scene_layouts(S.View(background_color=(10, 200, 30, 128), width=None), 64, 64, [], srgb=False)
    arr, n = scenes.engine_layouts({"type": "view", "background_color": "#0AC81E80"}, 64, 64, [], mode=1)
    assert_same_layouts(arr, n, want)


# ----------------------------------------------------------------------------- the reference's own API test vectors
def _api_vectors():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_api_vectors.json")
    return json.load(open(path))["vectors"]


@pytest.mark.parametrize("vec", _api_vectors(), ids=lambda v: v["name"])
def test_scene_api_reference_vectors(vec):
    """smelter-api/tests/scene_deserialization.rs, extracted by tests/golden/gen_scene_api_vectors.py: the JSON -> scene::Component
    conversion (defaults, validation, exact error strings)."""
    sc = Scene()
    if vec["kind"] == "ok":
        assert sc.parse(vec["scene"]) == vec["expected"]
    elif vec["kind"] == "err":
        with pytest.raises(SceneError) as e:
            sc.parse(vec["scene"])
        assert str(e.value) == vec["message"]
    else:
        with pytest.raises(SceneError):
            sc.parse(vec["scene"])


# ----------------------------------------------------------------------------- random trees, both restatements
def _hex(c):
    return "#%02X%02X%02X%02X" % tuple(c)


class TreeGen:
    """Random component trees as scene JSON plus the equivalent oracle.scene objects."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.inputs = 0
        self.resolutions = []

    def num(self, lo, hi):
        return float(np.float32(self.rng.uniform(lo, hi))) if self.rng.random() < 0.5 else float(self.rng.integers(int(lo), int(hi) + 1))

    def color(self):
        return tuple(int(x) for x in self.rng.integers(0, 256, 4))

    def shadows(self):
        return [(self.num(-20, 20), self.num(-20, 20), self.num(0, 30), self.color()) for _ in range(int(self.rng.integers(0, 3)))]

    def position(self, js, kw):
        r = self.rng
        if r.random() < 0.6:
            js["width"] = kw["width"] = self.num(20, 900)
        if r.random() < 0.6:
            js["height"] = kw["height"] = self.num(20, 600)
        if r.random() < 0.35:
            ab = S.AbsolutePosition(width=kw.pop("width", None), height=kw.pop("height", None))
            if r.random() < 0.5:
                js["top"] = ab.top = self.num(-50, 300)
            else:
                js["bottom"] = ab.bottom = self.num(-50, 300)
            if r.random() < 0.5:
                js["left"] = ab.left = self.num(-50, 300)
            else:
                js["right"] = ab.right = self.num(-50, 300)
            if r.random() < 0.5:
                js["rotation"] = ab.rotation_degrees = self.num(-180, 180)
            kw["absolute"] = ab

    def leaf(self):
        if self.rng.random() < 0.7:
            i = self.inputs
            self.inputs += 1
            w, h = int(self.rng.integers(16, 2000)), int(self.rng.integers(16, 1200))
            self.resolutions.append((w, h) if self.rng.random() < 0.9 else None)
            return {"type": "input_stream", "input_id": f"in{i}"}, S.InputStream(len(self.resolutions) - 1)
        w, h = int(self.rng.integers(8, 400)), int(self.rng.integers(8, 200))
        self.resolutions.append((w, h))
        return {"type": "text", "text": "x", "font_size": 10.0, "width": float(w), "height": float(h)}, S.NodeChild(w, h)

    def node(self, depth):
        r = self.rng
        kind = r.choice(["view", "rescaler", "tiles", "leaf"], p=[0.4, 0.2, 0.15, 0.25]) if depth < 4 else "leaf"
        if kind == "leaf":
            return self.leaf()
        js, kw = {"type": str(kind)}, {}
        if kind in ("view", "rescaler"):
            self.position(js, kw)
            if r.random() < 0.5:
                js["border_radius"] = kw["border_radius"] = self.num(0, 60)
            if r.random() < 0.5:
                js["border_width"] = kw["border_width"] = self.num(0, 12)
                c = self.color()
                js["border_color"], kw["border_color"] = _hex(c), c
            sh = self.shadows()
            if sh:
                js["box_shadow"] = [{"offset_x": a, "offset_y": b, "blur_radius": c, "color": _hex(d)} for a, b, c, d in sh]
                kw["box_shadow"] = [S.BoxShadow(*s) for s in sh]
        if kind == "view":
            if r.random() < 0.5:
                js["direction"] = kw["direction"] = str(r.choice(["row", "column"]))
            if r.random() < 0.6:
                js["overflow"] = kw["overflow"] = str(r.choice(["visible", "hidden", "fit"]))
            if r.random() < 0.7:
                c = self.color()
                js["background_color"], kw["background_color"] = _hex(c), c
            if r.random() < 0.4:
                p = [self.num(0, 20) for _ in range(4)]
                js.update(padding_top=p[0], padding_right=p[1], padding_bottom=p[2], padding_left=p[3])
                kw["padding"] = S.Padding(*p)
            kids = [self.node(depth + 1) for _ in range(int(r.integers(0, 4)))]
            js["children"] = [k[0] for k in kids]
            return js, S.View(children=[k[1] for k in kids], **kw)
        if kind == "rescaler":
            if r.random() < 0.5:
                js["mode"] = kw["mode"] = str(r.choice(["fit", "fill"]))
            if r.random() < 0.5:
                js["horizontal_align"] = kw["horizontal_align"] = str(r.choice(["left", "right", "center", "justified"]))
            if r.random() < 0.5:
                js["vertical_align"] = kw["vertical_align"] = str(r.choice(["top", "bottom", "center", "justified"]))
            kj, ko = self.node(depth + 1)
            js["child"] = kj
            return js, S.Rescaler(child=ko, **kw)
        if r.random() < 0.5:
            js["width"] = kw["width"] = self.num(100, 900)
            js["height"] = kw["height"] = self.num(100, 600)
        if r.random() < 0.5:
            c = self.color()
            js["background_color"], kw["background_color"] = _hex(c), c
        if r.random() < 0.5:
            a = (int(r.integers(1, 22)), int(r.integers(1, 22)))
            js["tile_aspect_ratio"], kw["tile_aspect_ratio"] = f"{a[0]}:{a[1]}", a
        if r.random() < 0.5:
            js["margin"] = kw["margin"] = self.num(0, 20)
        if r.random() < 0.5:
            js["padding"] = kw["padding"] = self.num(0, 20)
        if r.random() < 0.5:
            js["horizontal_align"] = kw["horizontal_align"] = str(r.choice(["left", "right", "center", "justified"]))
        if r.random() < 0.5:
            js["vertical_align"] = kw["vertical_align"] = str(r.choice(["top", "bottom", "center", "justified"]))
        kids = [self.node(depth + 1) for _ in range(int(r.integers(0, 6)))]
        js["children"] = [k[0] for k in kids]
        return js, S.Tiles(children=[k[1] for k in kids], **kw)

    def root(self):
        while True:
            js, obj = self.node(0)
            if js["type"] != "input_stream" and js["type"] != "text":
                return js, obj
            self.inputs, self.resolutions = 0, []


@pytest.mark.parametrize("seed", range(300))
def test_random_tree_matches_oracle(seed):
    g = TreeGen(seed)
    js, obj = g.root()
    W, H = int(g.rng.integers(64, 2000)), int(g.rng.integers(64, 1200))
    # the forced root size is the output resolution; a root with its own width/height reports that as the node resolution
    want = S.scene_layouts(obj, W, H, g.resolutions)
    sc = Scene()
    sc.update(js, W, H)
    arr, n, w, h = sc.node_layouts(0, 0, g.resolutions)
    if (w, h) != (W, H):  # flatten() culls against the node resolution (layout.rs:181-184)
        want = S.to_render_layouts(S.flatten(S.layout(obj, np.float32(W), np.float32(H)), list(g.resolutions), w, h))
    assert_same_layouts(arr, n, want)


# ----------------------------------------------------------------------------- easing / colours
def test_cubic_bezier_reference_vectors():
    close = lambda a, b: abs(a - b) < 1e-7  # noqa: E731  (ALLOWED_FLOATING_ERROR, cubic_bezier.rs:3)
    assert close(cubic_bezier_easing(0.0, 0.0, 0.0, 1.0, 1.0), 0.0)
    assert close(cubic_bezier_easing(1.0, 0.0, 0.0, 1.0, 1.0), 1.0)
    assert close(cubic_bezier_easing(0.5, 0.0, 0.0, 1.0, 1.0), 0.5)
    assert close(cubic_bezier_easing(0.294, 0.25, 0.1, 0.25, 1.0), 0.5014012915764126)
    assert close(cubic_bezier_easing(0.5, 0.85, 0.0, 0.15, 1.0), 0.5)


@pytest.mark.parametrize("pts", [(0.25, 0.1, 0.25, 1.0), (0.42, 0.0, 0.58, 1.0), (0.0, 0.0, 0.58, 1.0), (0.17, 0.67, 0.83, 0.67), (1.0, 0.3, 0.0, 0.7)])
def test_cubic_bezier_solves_the_curve(pts):
    x1, y1, x2, y2 = pts
    bez = lambda t, a, b: 3 * (1 - t) ** 2 * t * a + 3 * (1 - t) * t * t * b + t ** 3  # noqa: E731
    ts = np.linspace(0.0, 1.0, 20001)
    xs = bez(ts, x1, x2)
    prev = 0.0
    for p in np.linspace(0.01, 0.99, 37):
        y = cubic_bezier_easing(float(p), x1, y1, x2, y2)
        t = ts[np.argmin(np.abs(xs - p))]
        assert abs(y - min(1.0, max(0.0, bez(t, y1, y2)))) < 2e-3
        assert y >= prev - 1e-9 or y1 > 1 or y2 < 0
        prev = y


def test_bounce_easing():
    assert bounce_easing(0.0) == 0.0
    assert abs(bounce_easing(1.0) - 1.0) < 1e-12
    assert abs(bounce_easing(1 / 2.75) - 1.0) < 1e-12           # first touch-down
    assert abs(bounce_easing(0.5) - (7.5625 * (0.5 - 1.5 / 2.75) ** 2 + 0.75)) < 1e-15


def test_parse_color():
    assert parse_color("#FF000080") == (255, 0, 0, 128)
    assert parse_color("#10ff7A") == (16, 255, 122, 255)
    assert parse_color(" rebeccapurple ") == (102, 51, 153, 255)
    assert parse_color("burntsienna") == (234, 126, 93, 255)
    assert parse_color("rgb(1, 2,3)") == (1, 2, 3, 255)
    assert parse_color("rgba(255, 0, 10,0.5)") == (255, 0, 10, 128)   # (0.5 * 255).round()
    assert parse_color("rgba(0,0,0,1)") == (0, 0, 0, 255)
    for bad in ["#FFF", "#GGGGGG", "rgb(1,2)", "rgba(1,2,3, 0.5)", "rgba(1,2,3,1.5)", "rgb(256,0,0)", "hsl(1,2,3)", "Red", ""]:
        with pytest.raises(SceneError):
            parse_color(bad)


# ----------------------------------------------------------------------------- transitions
def _root_rect(sc, pts_ns, res=()):
    arr, n, _, _ = sc.node_layouts(0, pts_ns, list(res))
    return arr, n


def _view_scene(width, transition=None, **extra):
    inner = {"type": "view", "id": "box", "width": width, "height": 100.0, "top": 10.0, "left": 20.0, "background_color": "#FF0000FF"}
    if transition:
        inner["transition"] = transition
    inner.update(extra)
    return {"type": "view", "children": [inner]}


def _box(sc, pts_ns):
    arr, n = _root_rect(sc, pts_ns)
    assert n == 1  # the transparent root is culled
    return arr[0]


def test_view_transition_linear():
    sc = Scene()
    sc.update(_view_scene(100.0), 1280, 720)
    assert _box(sc, 1_000_000_000).width == 100.0
    sc.update(_view_scene(300.0, {"duration_ms": 2000}), 1280, 720)   # starts at the last rendered pts (1 s)
    for pts_s, want in [(1.0, 100.0), (1.5, 150.0), (2.0, 200.0), (2.75, 275.0), (3.0, 300.0), (10.0, 300.0)]:
        assert _box(sc, int(pts_s * 1e9)).width == np.float32(want), pts_s
    # no id match -> no transition
    sc2 = Scene()
    sc2.update(_view_scene(100.0), 1280, 720)
    _box(sc2, 0)
    other = _view_scene(300.0, {"duration_ms": 2000})
    other["children"][0]["id"] = "different"
    sc2.update(other, 1280, 720)
    assert _box(sc2, 1).width == 300.0


def test_layouts_of_a_scene_at_rest_are_served_from_the_cache_and_never_stale():
    """node_layouts keeps its result while nothing under the node is in transition (smr_host::Scene::node_layouts): the list must
    still follow a transition frame by frame, settle when it ends, change with the input resolutions and with the next update."""
    sc = Scene()
    scene = {"type": "view", "children": [{"type": "rescaler", "id": "r", "width": 400.0, "height": 300.0, "top": 10.0, "left": 10.0,
                                           "child": {"type": "input_stream", "input_id": "a"}}]}
    sc.update(scene, 1280, 720)

    def first(pts_ns, res):
        arr, n, _, _ = sc.node_layouts(0, pts_ns, [res])
        assert n == 1
        return (arr[0].top, arr[0].left, arr[0].width, arr[0].height)

    at_rest = first(0, (640, 360))
    assert first(1_000_000_000, (640, 360)) == at_rest            # cached
    other_input = first(2_000_000_000, (360, 640))                  # another resolution: not the cached list
    assert other_input != at_rest
    assert first(3_000_000_000, (640, 360)) == at_rest
    moved = dict(scene["children"][0], width=800.0, transition={"duration_ms": 1000})
    sc.update({"type": "view", "children": [moved]}, 1280, 720)     # starts at the last rendered pts (3 s)
    seen = [first(int(t * 1e9), (640, 360)) for t in (3.0, 3.25, 3.5, 3.75, 4.0, 5.0, 6.0)]
    assert seen[0] == at_rest and len(set(seen[:5])) == 5           # every frame of the transition is its own list
    assert seen[4] == seen[5] == seen[6] != at_rest                 # settled: served from the cache from here on
    sc.update(scene, 1280, 720)                                     # no transition on the new scene: jumps back
    assert first(7_000_000_000, (640, 360)) == at_rest


def test_view_transition_easing_and_offsets():
    tr = {"duration_ms": 1000, "easing_function": {"function_name": "cubic_bezier", "points": [0.25, 0.1, 0.25, 1.0]}}
    sc = Scene()
    sc.update(_view_scene(100.0, left=None, right=40.0), 1280, 720)
    _box(sc, 0)
    sc.update(_view_scene(200.0, tr, left=None, right=140.0, rotation=90.0, border_radius=20.0), 1280, 720)
    s = cubic_bezier_easing(0.294, 0.25, 0.1, 0.25, 1.0)
    b = _box(sc, 294_000_000)
    lerp = lambda a, c: np.float32(a + (c - a) * s)  # noqa: E731  (types/interpolation.rs: f64 lerp, cast to f32)
    width = lerp(100.0, 200.0)
    assert b.width == width
    assert b.rotation_degrees == lerp(0.0, 90.0)
    assert b.border_radius[0] == lerp(0.0, 20.0)
    assert b.left == np.float32(np.float32(1280.0) - lerp(40.0, 140.0) - width)  # right offset -> left edge


def test_transition_interruption_rules():
    tr = {"duration_ms": 1000}
    sc = Scene()
    sc.update(_view_scene(100.0), 1280, 720)
    _box(sc, 0)
    sc.update(_view_scene(200.0, tr), 1280, 720)
    assert _box(sc, 500_000_000).width == 150.0
    # a new update without should_interrupt keeps the running transition's remaining time (transition.rs:58-75):
    # it continues from the interpolated state (150) to the new end over the remaining 0.5 s
    sc.update(_view_scene(400.0, tr), 1280, 720)
    assert _box(sc, 500_000_000).width == 150.0
    assert _box(sc, 750_000_000).width == 275.0
    assert _box(sc, 1_000_000_000).width == 400.0
    # should_interrupt restarts with the full duration
    sc.update(_view_scene(0.0 + 100.0, {"duration_ms": 1000, "should_interrupt": True}), 1280, 720)
    assert _box(sc, 1_500_000_000).width == 250.0
    assert _box(sc, 2_000_000_000).width == 100.0


def test_rescaler_transition_and_bounce():
    def scene(w, tr=None):
        r = {"type": "rescaler", "id": "r", "width": w, "height": 90.0, "top": 0.0, "left": 0.0,
             "child": {"type": "view", "width": 160.0, "height": 90.0, "background_color": "#00FF00FF"}}
        if tr:
            r["transition"] = tr
        return {"type": "view", "children": [r]}
    sc = Scene()
    sc.update(scene(160.0), 640, 360)
    arr, n = _root_rect(sc, 0)
    assert n == 1 and arr[0].width == 160.0
    sc.update(scene(320.0, {"duration_ms": 1000, "easing_function": {"function_name": "bounce"}}), 640, 360)
    arr, n = _root_rect(sc, 400_000_000)
    w = np.float32(160.0 + 160.0 * bounce_easing(0.4))
    assert n == 1 and arr[0].width == min(w / np.float32(160.0), np.float32(1.0)) * np.float32(160.0)  # fit: limited by the height
    assert arr[0].left == (w - arr[0].width) / np.float32(2)


def test_tiles_transition_moves_tiles_by_id():
    def scene(order, tr=True):
        t = {"type": "tiles", "id": "t", "children": [{"type": "view", "id": k, "background_color": c} for k, c in order]}
        if tr:
            t["transition"] = {"duration_ms": 1000}
        return t
    a, b, c = ("a", "#FF0000FF"), ("b", "#00FF00FF"), ("c", "#0000FFFF")
    sc = Scene()
    sc.update(scene([a, b]), 1280, 720)
    arr, n = _root_rect(sc, 0)
    assert n == 2
    first = [(arr[i].left, arr[i].top, arr[i].width, arr[i].height) for i in range(2)]
    sc.update(scene([b, a]), 1280, 720)
    arr, n = _root_rect(sc, 500_000_000)
    mid = np.float32((first[0][0] + first[1][0]) / 2)
    assert n == 2 and arr[0].left == mid and arr[1].left == mid       # both halfway through swapping places
    arr, n = _root_rect(sc, 1_000_000_000)
    assert [arr[i].left for i in range(2)] == [first[0][0], first[1][0]]
    assert arr[0].color[1] > 0.9 and arr[1].color[0] > 0.9            # b is now first
    # a new tile appears only once the others have made room (tiles_component/interpolation.rs:38-62)
    sc.update(scene([b, a, c]), 1280, 720)
    arr, n = _root_rect(sc, 1_500_000_000)
    assert n == 2
    arr, n = _root_rect(sc, 2_000_000_000)
    assert n == 3


# ----------------------------------------------------------------------------- render graph
def test_render_graph_nodes():
    sc = Scene()
    sc.register_image("logo", 300, 100)
    scene = {"type": "view", "children": [
        {"type": "input_stream", "input_id": "cam"},
        {"type": "view", "children": [{"type": "image", "image_id": "logo", "width": 150.0},
                                      {"type": "text", "text": "hello", "font_size": 20.0, "width": 99.9, "height": 30.2}]},
        {"type": "shader", "shader_id": "blur", "resolution": {"width": 640, "height": 360},
         "shader_param": {"type": "f32", "value": 1.5},
         "children": [{"type": "view", "id": "nested", "width": 320.0, "height": 180.0, "children": [{"type": "input_stream", "input_id": "cam2"}]}]},
    ]}
    nodes = sc.update(scene, 1920, 1080)
    kinds = [n.kind for n in nodes]
    assert kinds == [_ffi.NODE_LAYOUT, _ffi.NODE_INPUT_STREAM, _ffi.NODE_IMAGE, _ffi.NODE_TEXT, _ffi.NODE_SHADER, _ffi.NODE_LAYOUT,
                     _ffi.NODE_INPUT_STREAM]
    root = nodes[0]
    assert root.children == [1, 2, 3, 4] and (root.width, root.height) == (1920, 1080)   # nested layouts are merged into the root node
    assert nodes[1].ref_id == "cam" and nodes[6].ref_id == "cam2"
    # image_component.rs: width given -> height from the image's (integer!) aspect ratio 300/100 = 3
    assert (nodes[2].width, nodes[2].height) == (150, 50)
    assert (nodes[3].width, nodes[3].height, nodes[3].payload) == (99, 30, "hello")        # `as usize` truncation
    assert (nodes[4].width, nodes[4].height, nodes[4].ref_id, nodes[4].children) == (640, 360, "blur", [5])
    assert (nodes[5].width, nodes[5].height, nodes[5].id, nodes[5].children) == (320, 180, "nested", [6])
    arr, n, w, h = sc.node_layouts(5, 0, [(1280, 720)])
    assert (w, h) == (320, 180) and n == 1 and arr[0].type == 0 and tuple(arr[0].crop) == (0.0, 0.0, 1280.0, 720.0)


@pytest.mark.parametrize("scene, fragment", [
    ({"type": "view", "id": "a", "children": [{"type": "view", "id": "a"}]}, "More than one component has an id \"a\""),
    ({"type": "view", "colour": "red"}, "unknown field `colour`"),
    ({"type": "view", "top": 1.0, "bottom": 2.0, "left": 0.0}, "\"top\" and \"bottom\" are mutually exclusive"),
    ({"type": "view", "top": 1.0}, "requires either \"left\" or \"right\""),
    ({"type": "rescaler", "rotation": 10.0, "child": {"type": "view"}}, "requires either \"top\" or \"bottom\""),
    ({"type": "view", "padding": -1.0}, "Padding values cannot be negative."),
    ({"type": "image", "image_id": "nope"}, "Image \"nope\" does not exist"),
    ({"type": "tiles", "tile_aspect_ratio": "16x9"}, "Aspect ratio needs to be a string"),
    ({"type": "text", "text": "a", "font_size": 0.0, "width": 1.0, "height": 1.0}, "\"font_size\" property has to be larger than 0"),
    ({"type": "text", "text": "a", "font_size": 1.0, "height": 1.0}, "can only be provided if \"width\" is also defined"),
    ({"type": "view", "transition": {"duration_ms": 10, "easing_function": {"function_name": "cubic_bezier", "points": [1.5, 0, 0.5, 1]}}},
     "Control point x1 has to be in the range [0, 1]."),
    ({"type": "shader", "shader_id": "s", "resolution": {"width": 8, "height": 8}, "children": [{"type": "view"}]},
     "need to have known size. Please provide width and height values."),
    ({"type": "view", "background_color": "#12345"}, "Color has to be in #RRGGBB or #RRGGBBAA format"),
    ({"type": "mystery"}, "unknown variant `mystery`"),
])
def test_scene_errors(scene, fragment):
    sc = Scene()
    with pytest.raises(SceneError) as e:
        sc.update(scene, 640, 360)
    assert fragment in str(e.value)


def test_failed_update_keeps_previous_scene():
    sc = Scene()
    sc.update({"type": "view", "background_color": "#FFFFFFFF"}, 64, 64)
    with pytest.raises(SceneError):
        sc.update("{not json", 64, 64)
    arr, n, _, _ = sc.node_layouts(0, 0, [])
    assert n == 1 and tuple(arr[0].color) == (1.0, 1.0, 1.0, 1.0)


def test_rgba_deserialization_reference_vectors():
    """smelter-api/src/video/color.rs:264-288 (test_rgba_deserialization), values and error strings."""
    assert parse_color("#00000000") == (0, 0, 0, 0)
    assert parse_color("#01020304") == (1, 2, 3, 4)
    assert parse_color("#01FF0304") == (1, 255, 3, 4)
    assert parse_color("#FFffFFff") == (255, 255, 255, 255)
    for bad, msg in (("#0000000G", "Invalid format. Color representation is not a valid number."),
                     ("#000", "Invalid format. Color has to be in #RRGGBB or #RRGGBBAA format.")):
        with pytest.raises(SceneError):
            parse_color(bad)
        with pytest.raises(SceneError) as e:  # (the message travels with a scene update, as the TypeError does in the reference)
            Scene().update({"type": "view", "background_color": bad}, 64, 64)
        assert msg in str(e.value)
