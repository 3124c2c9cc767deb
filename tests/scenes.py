"""Synthetic inputs and the BASELINE.json scenes, shared by tests/ and bench.py.

Inputs restate integration-tests/src/render_tests/harness/input.rs:58-151 (TestInput pattern and
the multiscale grid) with seeded noise on top.  Scenes are built with the scene
oracle (oracle/scene.py) — for bench.py the resulting POD layout list is only *data* handed to the
HIP renderer, nothing from oracle/ runs inside the timed region.
"""
from __future__ import annotations

import numpy as np

from oracle import scene as S

from smelter_amd.synth import COLOR_VARIANTS, label_glyphs, multiscale_grid, random_yuv420, test_input  # noqa: F401


# ----------------------------------------------------------------------------- BASELINE configs
def cfg2_scene(in_w=1920, in_h=1080, out_w=1920, out_h=1080, n=4):
    """configs[1]: 4x1080p YUV420 -> 1080p, Tiles, rescale + blend only."""
    root = S.Tiles(children=[S.InputStream(i) for i in range(n)], background_color=(0, 0, 0, 255))
    res = [(in_w, in_h)] * n
    return S.scene_layouts(root, out_w, out_h, res), res


LABEL_W, LABEL_H = 176, 32


def cfg3_scene(in_w=1920, in_h=1080, out_w=3840, out_h=2160, n=8, with_text=True):
    """configs[2]: 8x1080p tiled -> 4K with text overlay + rounded corners.

    Tiles{ n x View{ Rescaler{InputStream, border_radius 24}, Text label (absolute, bottom-left) } }.
    Node order per tile: [input_i, text_i]."""
    kids = []
    res = []
    for i in range(n):
        children = [S.Rescaler(child=S.InputStream(i), border_radius=24.0, border_width=4.0, border_color=(255, 255, 255, 255),
                               box_shadow=[S.BoxShadow(0.0, 8.0, 16.0, (0, 0, 0, 160))] if i % 2 == 0 else [])]
        res.append((in_w, in_h))
        if with_text:
            label = S.View(children=[S.NodeChild(LABEL_W, LABEL_H)], background_color=(0, 0, 0, 128), border_radius=8.0,
                           absolute=S.AbsolutePosition(width=float(LABEL_W), height=float(LABEL_H), left=24.0, bottom=24.0),
                           padding=S.Padding(4.0, 8.0, 4.0, 8.0))
            children.append(label)
            res.append((LABEL_W, LABEL_H))
        kids.append(S.View(children=children, background_color=(16, 16, 24, 255)))
    root = S.Tiles(children=kids, background_color=(32, 32, 48, 255), margin=0.0)
    return S.scene_layouts(root, out_w, out_h, res), res


def cfg2_scene_json(n=4) -> dict:
    """cfg2_scene as the scene JSON the reference's API takes (smelter-api/src/video/component.rs)."""
    return {"type": "tiles", "background_color": "#000000FF",
            "children": [{"type": "input_stream", "input_id": f"in{i}"} for i in range(n)]}


def cfg3_scene_json(n=8, with_text=True) -> dict:
    """cfg3_scene as scene JSON.  Node order per tile: [input_i, text_i]."""
    kids = []
    for i in range(n):
        resc = {"type": "rescaler", "child": {"type": "input_stream", "input_id": f"in{i}"}, "border_radius": 24.0,
                "border_width": 4.0, "border_color": "#FFFFFFFF"}
        if i % 2 == 0:
            resc["box_shadow"] = [{"offset_x": 0.0, "offset_y": 8.0, "blur_radius": 16.0, "color": "#000000A0"}]
        children = [resc]
        if with_text:
            children.append({"type": "view", "background_color": "#00000080", "border_radius": 8.0, "width": float(LABEL_W),
                             "height": float(LABEL_H), "left": 24.0, "bottom": 24.0, "padding_vertical": 4.0, "padding_horizontal": 8.0,
                             "children": [{"type": "text", "text": f"input {i}", "font_size": 24.0, "width": float(LABEL_W),
                                           "height": float(LABEL_H)}]})
        kids.append({"type": "view", "background_color": "#101018FF", "children": children})
    return {"type": "tiles", "background_color": "#202030FF", "margin": 0.0, "children": kids}


def engine_layouts(scene_json: dict, out_w: int, out_h: int, resolutions, pts_ns: int = 0, mode: int = 0):
    """Scene JSON -> (smr_layout ctypes array, count) through the C++ scene engine (root layout node)."""
    from smelter_amd.scene import Scene
    sc = Scene()
    sc.update(scene_json, out_w, out_h)
    arr, n, w, h = sc.node_layouts(0, pts_ns, list(resolutions), mode)
    assert (w, h) == (out_w, out_h)
    return arr, n


