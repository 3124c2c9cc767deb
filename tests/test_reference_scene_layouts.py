"""Layouts of the reference's own render-test scenes (tests/golden/render_test_scenes.json), three independent witnesses:

* the C++ scene engine of the product (smelter_amd/csrc/host/scene_*.cpp, through smr_scene_*),
* oracle/scene.py fed by tests/scene_json.py — a separate reading of smelter-api's JSON conversion (component_into.rs) and a
  separate restatement of scene/* + layout/flatten.rs; this is what tests/test_gpu_reference_scenes.py renders the oracle image from,
* known answers worked out by hand from the reference's rules, arithmetic in the comments below (tile grids: tiles.rs:59-166 and
  tiles_component/layout.rs:107-129; view rows / columns, padding and borders: view_component/layout.rs:31-201; rescaler fit:
  rescaler_component/layout.rs:14-165; culling: layout/flatten.rs:121-165).

The first two are transliterations of the same Rust; the third is not a transliteration of anything."""
import json
import os

import numpy as np
import pytest

from oracle import scene as S
from smelter_amd import _ffi
from smelter_amd.scene import Scene
from tests import scene_json
from tests.test_scene_engine import assert_same_layouts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the layout scenes of the corpus — trees of View / Rescaler / Tiles over input streams; the 33 scenes with Text / Image / Shader nodes are rendered
#  node by node in tests/test_gpu_reference_scenes.py)
def _only_layout_components(test):
    """True for the corpus scenes that are trees of View / Rescaler / Tiles over input streams (110 of the reference's 143 render tests)."""
    def ok(c):
        if c["type"] == "input_stream":
            return True
        if c["type"] not in ("view", "rescaler", "tiles"):
            return False
        return all(ok(k) for k in c.get("children", [])) and ("child" not in c or ok(c["child"]))
    return all(ok(s["update"]) for s in test["steps"] if "update" in s)


CORPUS = [t for t in json.load(open(os.path.join(ROOT, "tests", "golden", "render_test_scenes.json")))["tests"] if _only_layout_components(t)]
BY_NAME = {(t["module"], t["name"]): t for t in CORPUS}


def _has_transition(js):
    if isinstance(js, dict):
        return "transition" in js or any(_has_transition(v) for v in js.values())
    return isinstance(js, list) and any(_has_transition(v) for v in js)


def _static(case):
    """No update of the test carries a transition (those run through oracle/transition.py: tests/test_oracle_transition.py)."""
    try:
        for s in case["steps"]:
            if "update" in s:
                if _has_transition(s["update"]):
                    return False
                scene_json.to_oracle(s["update"])
        return True
    except scene_json.Unsupported:
        return False


STATIC = [t for t in CORPUS if _static(t)]


def _resolutions(case, input_ids):
    by_id = {i["id"]: (i["width"], i["height"]) for i in case["inputs"]}
    return [by_id.get(i) for i in input_ids]


def test_most_of_the_corpus_is_static():
    assert len(STATIC) >= 80 and {t["module"] for t in STATIC} >= {"view", "rescaler", "tiles"}


@pytest.mark.parametrize("case", STATIC, ids=[f'{t["module"]}.{t["name"]}' for t in STATIC])
def test_engine_and_oracle_agree_on_every_static_reference_scene(case):
    W, H = case["resolution"]
    mode = _ffi.MODE_GPU_OPTIMIZED if case["mode"] == "gpu_optimized" else _ffi.MODE_CPU_OPTIMIZED
    sc = Scene()
    n_updates = 0
    for step in case["steps"]:
        if "update" not in step:
            continue
        root, input_ids = scene_json.to_oracle(step["update"])
        res = _resolutions(case, input_ids)
        graph = sc.update(step["update"], W, H)
        # the engine's node graph: the root layout node with one input-stream child per occurrence, in the same order
        kids = [graph[k] for k in graph[0].children]
        assert [k.ref_id for k in kids] == input_ids
        want = S.scene_layouts(root, W, H, res, srgb=case["mode"] == "gpu_optimized")
        arr, n, w, h = sc.node_layouts(0, 0, res, mode)
        assert (w, h) == (W, H)
        assert_same_layouts(arr, n, want)
        n_updates += 1
    assert n_updates >= 1


# ----------------------------------------------------------------------------------------------------- hand-computed known answers
# (type, top, left, width, height[, source_index]) of the flattened list, in order; type 1 = colour rect, 0 = child texture.
# Exact, except for the two scenes whose tile size is a third of something (THIRDS): there the f32 rounding of every step is not
# worked out by hand and all fields get 2e-3 px.
THIRDS = {("tiles", "tiles_05_inputs"), ("tiles", "margin_with_03_inputs")}
KNOWN = {
    # root row 640x360; red view width 100 (height = parent's), green view width 300 at x = 100 with three 180x200 input streams in a
    # row at x = 100 + 180 k (static children with a known size keep it; the third sticks out of the green view: masked, not culled,
    # its left edge 460 is inside the 640-wide output); the transparent root is culled (flatten.rs:134-138)
    ("view", "overflow_hidden_with_input_stream_children"): [
        (1, 0, 0, 100, 360), (1, 0, 100, 300, 360), (0, 0, 100, 180, 200, 0), (0, 0, 280, 180, 200, 1), (0, 0, 460, 180, 200, 2)],
    # column root (blue); two children without a size share the height: 360 / 2 = 180 each, full width.  A: only a border -> still
    # rendered.  B: border 10 -> content 620x160; row direction; its child C has no size: width = 620 - (20 + 20) = 580, height =
    # 160 - (20 + 40) = 100, at top = border 10 + padding_top 20 = 30, left = border 10 + padding_left 20 = 30 inside B (y = 180)
    ("view", "unsized_view_padding_static_children"): [
        (1, 0, 0, 640, 360), (1, 0, 0, 640, 180), (1, 180, 0, 640, 180), (1, 210, 30, 580, 100)],
    # padding_top 360 leaves (360 - 360 - 0) / 2 = 0 px of height per child: culled (height <= 0); their parent has no colour
    ("view", "view_padding_overflow_children"): [(1, 0, 0, 640, 360)],
    # 2 inputs, 16:9 tiles on 640x360: one row of two -> x scale 640/2/16 = 20 < y scale 360/1/9 = 40 -> 320x180 (two rows of one give
    # the same width: the first maximum wins); centred vertically: (360 - 180) / 2 = 90
    ("tiles", "tiles_02_inputs"): [(1, 0, 0, 640, 360), (0, 90, 0, 320, 180, 0), (0, 90, 320, 320, 180, 1)],
    # 3 inputs: 1x3 -> 213.3 wide, 2x2 -> 320 wide, 3x1 -> 213.3: 2 rows x 2 columns of 320x180; last row holds one tile, centred:
    # (640 - 320) / 2 = 160
    ("tiles", "tiles_03_inputs"): [(1, 0, 0, 640, 360), (0, 0, 0, 320, 180, 0), (0, 0, 320, 320, 180, 1), (0, 180, 160, 320, 180, 2)],
    ("tiles", "align_top_left_with_03_inputs"): [(1, 0, 0, 640, 360), (0, 0, 0, 320, 180, 0), (0, 0, 320, 320, 180, 1), (0, 180, 0, 320, 180, 2)],
    # 5 inputs: 2 rows x 3 columns: scale = min(640/3/16, 360/2/9) = 13.33 -> 213.33 x 120; vertical slack 360 - 240 = 120 -> top 60;
    # first row fills the width, second row (two tiles) has 640 - 426.67 = 213.33 of slack -> left 106.67, then + 213.33 = 320
    ("tiles", "tiles_05_inputs"): [(1, 0, 0, 640, 360), (0, 60, 0, 213.3333, 120, 0), (0, 60, 213.3333, 213.3333, 120, 1),
                                   (0, 60, 426.6667, 213.3333, 120, 2), (0, 180, 106.6667, 213.3333, 120, 3), (0, 180, 320, 213.3333, 120, 4)],
    # margin 50, 3 inputs: 2x2 wins — x scale (640 - 3*50)/2/16 = 15.31, y scale (360 - 3*50)/2/9 = 11.67 -> 186.67 x 105; no vertical
    # slack -> top = margin 50, second row 50 + 105 + 50 = 205; first row slack 640 - 373.33 - 150 = 116.67 -> left 58.33 + 50 = 108.33,
    # next + 186.67 + 50 = 345; second row slack 640 - 186.67 - 100 = 353.33 -> left 176.67 + 50 = 226.67
    ("tiles", "margin_with_03_inputs"): [(1, 0, 0, 640, 360), (0, 50, 108.3333, 186.6667, 105, 0), (0, 50, 345, 186.6667, 105, 1),
                                         (0, 205, 226.6667, 186.6667, 105, 2)],
    # 1:2 tiles, one 360x640 input on 640x360: scale = min(640/1/1, 360/1/2) = 180 -> tile 180x360 centred at left (640-180)/2 = 230;
    # the input fits with min(180/360, 360/640) = 0.5 -> 180x320, centred in the tile: top (360-320)/2 = 20
    ("tiles", "tiles_01_portrait_inputs"): [(1, 0, 0, 640, 360), (0, 20, 230, 180, 320, 0)],
    # red 160x90 view at the origin; rescaler 320x180 at (160, 90), fit: scale = min(320/640, 180/260) = 0.5 -> 320x130, centred:
    # top = 90 + (180 - 130) / 2 = 115
    ("rescaler", "fit_input_stream_lower_aspect_ratio"): [(1, 0, 0, 160, 90), (0, 115, 160, 320, 130, 0)],
}


def _records(arr, n):
    out = []
    for i in range(n):
        L = arr[i]
        rec = (int(L.type), L.top, L.left, L.width, L.height)
        out.append(rec + ((int(L.source_index),) if L.type == 0 else ()))
    return out


def _assert_known(got, want, what, tol):
    assert len(got) == len(want), (what, got, want)
    for g, w in zip(got, want):
        assert g[0] == w[0], (what, g, w)
        for a, b in zip(g[1:5], w[1:5]):
            assert abs(a - b) <= tol, (what, g, w)
        if w[0] == 0:
            assert g[5] == w[5], (what, g, w)


@pytest.mark.parametrize("key", sorted(KNOWN), ids=[".".join(k) for k in sorted(KNOWN)])
def test_known_answers(key):
    case = BY_NAME[key]
    W, H = case["resolution"]
    update = next(s["update"] for s in case["steps"] if "update" in s)
    root, input_ids = scene_json.to_oracle(update)
    res = _resolutions(case, input_ids)
    # the oracle
    oracle = [(l.type, l.top, l.left, l.width, l.height) + ((l.source_index,) if l.type == 0 else ()) for l in S.scene_layouts(root, W, H, res)]
    tol = 2e-3 if key in THIRDS else 0.0
    _assert_known(oracle, KNOWN[key], "oracle/scene.py", tol)
    # the product's engine
    sc = Scene()
    sc.update(update, W, H)
    arr, n, _, _ = sc.node_layouts(0, 0, res)
    _assert_known(_records(arr, n), KNOWN[key], "C++ scene engine", tol)
    # a texture layout of these scenes shows its whole source (crop = the input's size)
    for i in range(n):
        if arr[i].type == 0:
            w, h = res[arr[i].source_index]
            assert list(arr[i].crop) == [0.0, 0.0, float(w), float(h)]
