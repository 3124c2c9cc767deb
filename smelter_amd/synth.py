"""Synthetic inputs for bench.py, smoke() and the tests (no datasets are reachable from the build hosts).

Restates integration-tests/src/render_tests/harness/input.rs:58-151 (the TestInput pattern and the multiscale grid) with
seeded noise on top, plus a procedural bitmap font for text-node payloads.
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

COLOR_VARIANTS = [(255, 0, 0), (0, 255, 0), (255, 255, 0), (255, 0, 255), (0, 0, 255), (0, 255, 255), (255, 165, 0),
                  (255, 255, 255), (128, 128, 128), (255, 128, 128), (128, 128, 255), (128, 255, 128), (255, 192, 203),
                  (128, 0, 128), (165, 42, 42), (154, 205, 50), (255, 255, 224)]

Glyph = namedtuple("Glyph", "dst_x dst_y w h atlas_x atlas_y color")  # field names of smr_glyph (include/smr.h)


def _rgb_to_yuv_f32(rgb):
    # RGBColor::to_yuv, smelter-render/src/scene/types.rs:28-41
    r, g, b = [np.float32(c) / np.float32(255) for c in rgb]
    y = r * np.float32(0.2126) + g * np.float32(0.7152) + b * np.float32(0.0722)
    u = r * np.float32(-0.1146) + g * np.float32(-0.3854) + b * np.float32(0.5)
    v = r * np.float32(0.5) + g * np.float32(-0.4542) + b * np.float32(-0.0458)
    cl = lambda x: min(max(x, np.float32(0)), np.float32(1))  # noqa: E731
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


test_input.__test__ = False  # not a pytest test


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


def label_glyphs(text: str, scale: int = 3):
    """A procedural 5x7 bitmap font -> (atlas A8, glyph quads) for synthetic labels. Glyph *shapes* are
    not a parity subject (third-party rasteriser in the reference); the blit arithmetic is."""
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


def animated_grid_scene(n: int = 16, rotate: int = 0, layer_w: int = 960, layer_h: int = 540, transition_ms: int = 500,
                        sigma: float = 3.0) -> dict:
    """BASELINE.json configs[4] as scene JSON: `n` input streams in a Tiles grid whose children swap places between
    scene updates (`rotate` shifts the child order; Tiles animates children matched by id, tiles_component/interpolation.rs:17-64)
    with a 500 ms cubic-bezier transition, plus one layer that goes through the built-in gaussian-blur shader (a Shader node
    with its own layout sub-tree) and sits on top of the grid with rounded corners."""
    order = [(i + rotate) % n for i in range(n)]
    grid = {"type": "tiles", "id": "grid", "background_color": "#101018FF",
            "transition": {"duration_ms": transition_ms, "easing_function": {"function_name": "cubic_bezier", "points": [0.65, 0.0, 0.35, 1.0]}},
            "children": [{"type": "input_stream", "id": f"tile_{i}", "input_id": f"input_{i}"} for i in order]}
    layer = {"type": "view", "width": layer_w, "height": layer_h, "top": round(layer_h / 8), "left": round(layer_w / 8), "border_radius": 16,
             "children": [{"type": "shader", "shader_id": "soften", "resolution": {"width": layer_w, "height": layer_h},
                           "shader_param": {"type": "f32", "value": sigma},
                           "children": [{"type": "view", "width": layer_w, "height": layer_h,
                                         "children": [{"type": "rescaler", "child": {"type": "input_stream", "input_id": "input_0"}}]}]}]}
    return {"type": "view", "children": [grid, layer]}
