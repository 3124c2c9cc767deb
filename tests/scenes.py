"""Synthetic inputs and the BASELINE.json scenes, shared by tests/ and bench.py.

Inputs restate integration-tests/src/render_tests/harness/input.rs:58-151 (TestInput pattern and
the multiscale grid) with seeded noise on top.  Scenes are built with the scene
oracle (oracle/scene.py) — for bench.py the resulting POD layout list is only *data* handed to the
HIP renderer, nothing from oracle/ runs inside the timed region.
"""
from __future__ import annotations

import numpy as np

from oracle import scene as S

COLOR_VARIANTS = [(255, 0, 0), (0, 255, 0), (255, 255, 0), (255, 0, 255), (0, 0, 255), (0, 255, 255), (255, 165, 0),
                  (255, 255, 255), (128, 128, 128), (255, 128, 128), (128, 128, 255), (128, 255, 128), (255, 192, 203),
                  (128, 0, 128), (165, 42, 42), (154, 205, 50), (255, 255, 224)]


def _rgb_to_yuv_f32(rgb):
    # RGBColor::to_yuv, smelter-render/src/scene/types.rs:28-41
    r, g, b = [np.float32(c) / np.float32(255) for c in rgb]
    y = r * np.float32(0.2126) + g * np.float32(0.7152) + b * np.float32(0.0722)
    u = r * np.float32(-0.1146) + g * np.float32(-0.3854) + b * np.float32(0.5)
    v = r * np.float32(0.5) + g * np.float32(-0.4542) + b * np.float32(-0.0458)
    cl = lambda x: min(max(x, np.float32(0)), np.float32(1))
    return (cl(y * np.float32(0.85882354) + np.float32(16.0 / 255.0)),
            cl((u + np.float32(0.5)) * np.float32(0.8784314) + np.float32(16.0 / 255.0)),
            cl((v + np.float32(0.5)) * np.float32(0.8784314) + np.float32(16.0 / 255.0)))


def test_input(index: int, w: int = 640, h: int = 360, noise_seed=None, shift: int = 0):
    """TestInput::new_with_resolution (input.rs:58-110), vectorised. Returns (Y, U, V) uint8 planes (4:2:0)."""
    yc, uc, vc = _rgb_to_yuv_f32(COLOR_VARIANTS[index % len(COLOR_VARIANTS)])
    xs = (np.arange(w) + shift) % w
    ys = np.arange(h)
    border_x = (xs <= 18) | ((xs <= w) & (xs >= w - 18))
    border_y = (ys <= 18) | ((ys <= h) & (ys >= h - 18))
    grid = ((xs[None, :] // 72 + ys[:, None] // 72) % 2) == 0
    dark = border_x[None, :] | border_y[:, None] | grid
    yv = np.where(dark, np.float32(yc) - np.float32(0.2), np.float32(yc)).astype(np.float32)
    yv = np.clip(yv, 0, 1)
    Y = (yv * np.float32(255.0)).astype(np.uint8)
    if noise_seed is not None:
        rng = np.random.default_rng(noise_seed)
        Y = np.clip(Y.astype(np.int16) + rng.integers(-8, 9, size=Y.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    U = np.full((h // 2, w // 2), np.uint8(np.float32(uc * 4) * np.float32(64.0)), np.uint8)
    V = np.full((h // 2, w // 2), np.uint8(np.float32(vc * 4) * np.float32(64.0)), np.uint8)
    return Y, U, V


def multiscale_grid(w: int, h: int):
    """TestInput::new_multiscale_grid (input.rs:116-151)."""
    periods = [21, 15, 12, 9, 7, 5, 4, 3]
    band_w = w // len(periods)
    xs, ys = np.arange(w), np.arange(h)
    band = np.minimum(xs // band_w, len(periods) - 1)
    per = np.array(periods)[band]
    on_v = (xs % per) < 2
    on_h = (ys[:, None] % per[None, :]) < 2
    Y = np.where(on_v[None, :] | on_h, 30, 200).astype(np.uint8)
    U = np.full((h // 2, w // 2), 128, np.uint8)
    V = np.full((h // 2, w // 2), 128, np.uint8)
    return Y, U, V


def random_yuv420(w, h, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8),
            rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8))


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


def label_glyphs(text: str, scale: int = 3):
    """A procedural 5x7 bitmap font -> (atlas A8, glyph quads) for synthetic labels. Glyph *shapes* are
    not a parity subject (third-party rasteriser in the reference); the blit arithmetic is."""
    from oracle.oracle import Glyph
    rng = np.random.default_rng(77)
    gw, gh = 5 * scale, 7 * scale
    chars = sorted(set(text))
    atlas = np.zeros((gh, gw * len(chars)), np.uint8)
    for ci, ch in enumerate(chars):
        bits = rng.integers(0, 2, (7, 5), dtype=np.uint8) if ch != " " else np.zeros((7, 5), np.uint8)
        cov = np.kron(bits, np.ones((scale, scale), np.uint8)).astype(np.float32) * 255
        # soften edges so coverage takes intermediate values like a real rasteriser's AA
        pad = np.pad(cov, 1, mode="edge")
        cov = (pad[:-2, 1:-1] + pad[2:, 1:-1] + pad[1:-1, :-2] + pad[1:-1, 2:] + 4 * cov) / 8
        atlas[:, ci * gw:(ci + 1) * gw] = cov.astype(np.uint8)
    glyphs = []
    x = 4
    for ch in text:
        ci = chars.index(ch)
        glyphs.append(Glyph(x, 5, gw, gh, ci * gw, 0, (1.0, 1.0, 1.0, 1.0)))
        x += gw + scale
    return atlas, glyphs
